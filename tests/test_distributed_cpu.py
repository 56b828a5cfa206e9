"""world_size-2 gloo test of the multi-GPU host logic (node sharding + moment all-reduce).
No GPU here: each rank computes the partial moment of ITS nodes with the oracle's per-node body
(test infrastructure), the sum goes through the same `allreduce_moment` the RCCL path uses, and
must equal the single-process moment."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    import importlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = graft.load_package()
        fo = graft.load_oracle()
        dm = importlib.import_module(graft.PKG_NAME + ".distributed")
        wl = importlib.import_module(graft.PKG_NAME + ".workloads")
        w = wl.c3_sparse_generalized(8, 19, 16)
        con = pkg.circular_contour_trapezoidal(w["c"], w["r"], 16)
        A, B = w["A"].astype(np.complex128), w["B"].astype(np.complex128)
        X = w["X0"] / np.linalg.norm(w["X0"], axis=0)
        lam = np.linspace(0.1, 1.2, 16).astype(np.complex128)
        R = A @ X - (B @ X) * lam[None, :]
        mine = dm.shard_nodes(16, world, rank)

        def partial(nodes):
            Q = np.zeros_like(X)
            for k in nodes:                                   # src/feast.jl:59-71
                Y = fo.lu(A - B * con.nodes[k]).solve(R)
                Q += (X - Y) * (con.weights[k] / (con.nodes[k] - lam))[None, :]
            return Q

        qt = torch.from_numpy(np.ascontiguousarray(partial(mine)).view(np.float64).copy())
        dm.allreduce_moment(qt)
        full = partial(range(16))
        err = np.abs(qt.numpy().view(np.complex128).reshape(full.shape) - full).max() / np.abs(full).max()
        q.put((rank, mine, float(err)))
    finally:
        dist.destroy_process_group()


def test_shard_nodes_partition(pkg):
    import importlib
    import __graft_entry__ as graft
    dm = importlib.import_module(graft.PKG_NAME + ".distributed")
    for K in (1, 8, 16, 32, 33):
        for G in (1, 2, 3, 4, 8):
            parts = [dm.shard_nodes(K, G, r) for r in range(G)]
            assert sorted(sum(parts, [])) == list(range(K))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        dm.shard_nodes(8, 2, 2)


def test_two_rank_moment_allreduce_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] == list(range(0, 16, 2)) and res[1][1] == list(range(1, 16, 2))
    assert all(r[2] < 1e-12 for r in res)
