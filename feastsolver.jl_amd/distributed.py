"""Multi-GPU driver: contour nodes sharded over ranks, one RCCL sum of Q per iteration.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Rank g owns
the nodes {k : k mod G = g} -- or, for real matrices on a mirror-symmetric contour, every G-th conjugate
node PAIR (shard_plan) -- including their factorisations -- the memory-heavy
state is what gets sharded; A, B, X, R are replicated.  After the all-reduce every rank
runs the small Rayleigh-Ritz step redundantly, so no broadcast is needed.  The reference has
no distributed path (Distributed/SharedArrays are imported but unused,
src/FEASTSolver.jl:10-11); its node loop (src/feast.jl:59-71) is what is sharded here.
"""
from __future__ import annotations

import numpy as np


def shard_nodes(K, world_size, rank):
    """Indices of the contour nodes owned by `rank` when single nodes are dealt out (round robin: what
    feasthip_set_node_shard does without the conjugate fold)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return list(range(rank, K, world_size))


def shard_plan(nodes, world_size, rank, fold_allowed=True):
    """What feasthip_set_node_shard(rank, world_size) makes of a contour (host-only query of the library's own logic,
    feasthip_shard_plan): (local, partner, folded).  With the conjugate fold (real matrices, every node has its mirror
    image on the contour, at least world_size pairs) the units dealt out are conjugate PAIRS: local[i] is factorised,
    partner[i] rides along as extra right-hand-side columns; this rank integrates over local + partner."""
    import ctypes as C
    from . import _lib
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    z = np.ascontiguousarray(nodes, dtype=np.complex128)
    K = len(z)
    loc = np.zeros(K, dtype=np.int32); par = np.zeros(K, dtype=np.int32)
    cnt, fol = C.c_int(0), C.c_int(0)
    rc = _lib.load_library().feasthip_shard_plan(K, z.ctypes.data, 1 if fold_allowed else 0, int(rank), int(world_size),
                                                 loc.ctypes.data, par.ctypes.data, C.byref(cnt), C.byref(fol))
    if rc != 0:
        raise ValueError("feasthip_shard_plan failed")
    return loc[:cnt.value].tolist(), par[:cnt.value].tolist(), bool(fol.value)


def allreduce_moment(q, group=None):
    """Sum the partial moment(s) Q over all ranks in place.  `q` is a torch tensor (device
    memory for nccl/RCCL, host memory for gloo); a no-op without an initialised process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if q.is_cuda and dist.get_backend(group) == "gloo":      # rehearsal without RCCL (several ranks on one GPU): stage through the host
            tmp = q.cpu()
            dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group)
            q.copy_(tmp)
        else:
            dist.all_reduce(q, op=dist.ReduceOp.SUM, group=group)
    return q


class MomentReducer:
    """One contour step of this rank's node shard + the sum of the partial moments over the ranks.

    overlap=True: the library produces the moment in `chunks` contiguous row ranges (feasthip_contour_step_begin); as soon as
    a range is final its all-reduce is issued (async), so the collective of range c runs while the ranges after it are still
    being accumulated, and only the last range's collective is exposed.  overlap=False: contour step, then one all-reduce of
    the whole n x ldw block.  Either way every element is the sum of the same partial moments; the ORDER in which a ring or tree
    collective adds them may depend on the element's offset and the buffer size, so the two settings agree to rounding (a few ulp of
    the moment: what the tests against the single-process fixture bound at 1e-10), not bit by bit.
    `q` is the torch tensor handed to feasthip_set_q_buffer (float64 view of the complex128 moment)."""

    def __init__(self, handle, q, n, ldw, overlap=True, chunks=4, group=None):
        self.h, self.q, self.n, self.ldw, self.group = handle, q, int(n), int(ldw), group
        self.overlap = bool(overlap)
        self.chunks = max(1, int(chunks)) if self.overlap else 1

    def contour_step_and_reduce(self, refactor=False):
        """Returns the host time at which this rank's own contour step had finished (for the Amdahl breakdown)."""
        import time
        import torch
        import torch.distributed as dist
        live = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1
        gloo = live and dist.get_backend(self.group) == "gloo"
        row = self.ldw * 2                                   # float64 per row of the complex128 block
        self.h.contour_step_begin(refactor, self.chunks)
        works = []
        for c in range(self.chunks):
            r0, rows = self.h.moment_chunk_wait(c)          # the library's stream is not torch's: host-side wait, then enqueue
            if not live:
                continue
            part = self.q[r0 * row:(r0 + rows) * row]
            if gloo and part.is_cuda:                        # rehearsal without RCCL (ranks sharing one GPU): stage through the host
                tmp = part.cpu()
                works.append((dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.group, async_op=True), part, tmp))
            else:
                works.append((dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group, async_op=True), None, None))
        self.h.contour_step_end()
        t_done = time.perf_counter()
        for w, part, tmp in works:
            w.wait()
            if tmp is not None:
                part.copy_(tmp)
        if live:
            torch.cuda.synchronize()                         # the next Rayleigh-Ritz step reads Q on the library's stream
        return t_done


class LibraryReducer:
    """The same step with the collective INSIDE the library (feasthip_comm_init, round 4): rank 0 draws RCCL's unique id, torch.distributed
    only carries those 128 bytes to the other ranks (any backend: it is host data), every rank joins the communicator, and from then on
    feasthip_contour_step returns with the summed moment -- the all-reduce of a row range is enqueued on the library's communication
    stream behind the event of that range's accumulation, without a host wait (MomentReducer waits on the host per range and enqueues
    on torch's stream).  What a Julia host would use (julia/FEASTHip.jl); executed with one rank only so far."""

    def __init__(self, handle, group=None):
        import torch.distributed as dist
        from . import api
        self.h = handle
        live = dist.is_available() and dist.is_initialized()
        rank = dist.get_rank(group) if live else 0
        world = dist.get_world_size(group) if live else 1
        box = [api.comm_unique_id() if rank == 0 else None]
        if live and world > 1:
            # src is a GLOBAL rank: the first member of a sub-group need not be global rank 0
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
        if not isinstance(box[0], (bytes, bytearray)) or len(box[0]) != 128:
            raise RuntimeError("LibraryReducer: the RCCL unique id did not arrive (128 bytes expected)")
        handle.comm_init(box[0], rank, world)

    def contour_step_and_reduce(self, refactor=False):
        import time
        self.h.contour_step(refactor)
        return time.perf_counter()


def gen_feast_distributed(X, A, B, contour, *, iter=10, eps=1e-12, store=True, device=None, group=None, info=None, overlap=True, reduce="torch"):
    """gen_feast! (src/feast.jl:89-156) with the node loop sharded over the ranks of `group`.
    Every rank passes the same X, A, B, contour and receives the same result.  B=None: feast!.
    reduce: "torch" (torch.distributed all-reduce of the moment, MomentReducer) or "library" (RCCL inside libfeasthip, LibraryReducer)."""
    import torch
    import torch.distributed as dist
    from . import api
    from .contour import in_contour
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if device is None:
        device = torch.cuda.current_device()
    if A.shape[0] != A.shape[1]:
        raise ValueError("Incorrect dimensions of A, must be square")
    if X.shape[0] != A.shape[0]:
        raise ValueError("Incorrect dimensions of X, must match A")
    n, m0 = X.shape
    ldw = ((m0 + 15) // 16) * 16
    h = api.FeastHandle(device)
    try:
        h.set_terms([A, B], scales=[1.0, -1.0])
        h.set_contour(contour)
        h.set_node_shard(rank, world)
        q = torch.zeros(2 * n * ldw * 2, dtype=torch.float64, device=f"cuda:{device}")
        h.set_q_buffer(q.data_ptr(), q.numel() * 8)
        if store:
            h.factorize()
        h.set_x(np.asfortranarray(X, dtype=np.complex128))
        red = LibraryReducer(h, group=group) if reduce == "library" else MomentReducer(h, q, n, ldw, overlap=overlap, group=group)
        nit = 0
        for nit in range(iter + 1):                                   # src/feast.jl:40
            lam, res, ins = h.rr_step()
            if ins.any() and res[ins].max() < eps:                    # :53
                break
            if nit < iter:                                            # :57
                red.contour_step_and_reduce(refactor=not store)
        Xo = h.get_block(0)
        if info is not None:
            info.update(h.stats())
            info.update(nit=nit, rank=rank, world=world)
    finally:
        h.close()
    if isinstance(X, np.ndarray) and np.issubdtype(X.dtype, np.complexfloating):
        X[...] = Xo
    inside = in_contour(lam, contour)
    return lam[inside], Xo[:, inside], res[inside]
