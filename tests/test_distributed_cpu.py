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


def test_shard_plan_deals_out_conjugate_pairs(pkg):
    """feasthip_set_node_shard with the conjugate fold (host-only query of the library's logic): every rank gets whole
    conjugate pairs, the ranks together cover every node exactly once, and the fold is dropped consistently on all ranks
    when it does not apply (complex centre, more ranks than pairs, fold not allowed)."""
    import importlib
    import __graft_entry__ as graft
    dm = importlib.import_module(graft.PKG_NAME + ".distributed")
    for K in (8, 16, 32):
        con = pkg.circular_contour_trapezoidal(0.3, 0.7, K)
        z = np.asarray(con.nodes)
        for G in (1, 2, 3, 4, 8, 16):
            plans = [dm.shard_plan(z, G, r) for r in range(G)]
            folded = {p[2] for p in plans}
            assert len(folded) == 1                                   # the same decision on every rank
            covered = sorted(sum((p[0] + [q for q in p[1] if q >= 0] for p in plans), []))
            assert covered == list(range(K))
            if K // 2 >= G:
                assert folded == {True}
                for loc, par, _ in plans:
                    assert all(z[a].imag > 0 and abs(z[b] - np.conj(z[a])) <= 1e-14 for a, b in zip(loc, par))
                assert max(len(p[0]) for p in plans) - min(len(p[0]) for p in plans) <= 1
            else:
                assert folded == {False} and all(p[0] == dm.shard_nodes(K, G, r) for r, p in enumerate(plans))
        # not allowed (complex matrices / per-node overrides) or no mirror symmetry: single nodes, round robin
        assert dm.shard_plan(z, 2, 1, fold_allowed=False) == (dm.shard_nodes(K, 2, 1), [-1] * (K // 2), False)
        zc = np.asarray(pkg.circular_contour_trapezoidal(0.3 + 0.1j, 0.7, K).nodes)
        assert dm.shard_plan(zc, 2, 0)[2] is False
    zr = np.asarray(pkg.circular_contour_trapezoidal(0.0, 1.0, 8).nodes).copy()
    zr[0] = zr[0].real                                               # a node on the real axis has no partner
    assert dm.shard_plan(zr, 1, 0)[2] is False


def test_conjugate_fold_identity_with_the_oracle(pkg, fo, wl):
    """The identity behind the fold, checked on the CPU with the oracle's LU: for real A, B
    T(conj z)^-1 R = conj(T(z)^-1 conj(R)), so a pair's contribution to the moment needs one factorisation."""
    w = wl.c3_sparse_generalized(6, 19, 8)
    con = pkg.circular_contour_trapezoidal(w["c"], w["r"], 8)
    A, B = w["A"].astype(np.complex128), w["B"].astype(np.complex128)
    rng = np.random.default_rng(3)
    X = rng.standard_normal(w["X0"].shape) + 1j * rng.standard_normal(w["X0"].shape)
    lam = np.linspace(0.1, 1.2, X.shape[1]).astype(np.complex128)
    R = A @ X - (B @ X) * lam[None, :]
    import importlib
    import __graft_entry__ as graft
    dm = importlib.import_module(graft.PKG_NAME + ".distributed")
    loc, par, folded = dm.shard_plan(con.nodes, 1, 0)
    assert folded and len(loc) == 4
    Qf = np.zeros_like(X)
    for a, b in zip(loc, par):
        F = fo.lu(A - B * con.nodes[a])
        Y, Y2 = F.solve(R), F.solve(R.conj())
        Qf += (X - Y) * (con.weights[a] / (con.nodes[a] - lam))[None, :] + (X - Y2.conj()) * (con.weights[b] / (con.nodes[b] - lam))[None, :]
    Q = np.zeros_like(X)
    for k in range(8):                                                 # src/feast.jl:59-71
        Q += (X - fo.lu(A - B * con.nodes[k]).solve(R)) * (con.weights[k] / (con.nodes[k] - lam))[None, :]
    assert np.abs(Qf - Q).max() <= 1e-12 * np.abs(Q).max()


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


def test_bench_gpus_n_never_reports_a_single_rank():
    """`python bench.py --gpus 2` without a launcher starts its two ranks itself and must fail loudly when they cannot
    form a 2-rank job (here: no GPU) -- it must never print a line with n_gpus = 1."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--grid", "8",
                        "--cpu-baseline", "off"], env=env, capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.device_count() == 0:
        assert r.returncode != 0
        assert '"n_gpus": 1' not in r.stdout
