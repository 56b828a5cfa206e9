#!/usr/bin/env python3
"""FEAST iterations/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[2] = C3, sparse generalized
A x = lambda B x with A the 19-point 3-D Laplacian on a 46^3 grid (n = 97,336,
nnz = 1,786,456), B = diag(U(0.5,1.5)), circular contour with 16 trapezoid
nodes, m0 = 64 -- the configuration the metric is quoted on (it fits one GPU).
One *step* = one FEAST iteration with store=true semantics (src/feast.jl:40-73):
Rayleigh-Ritz block + 16 shifted block solves + moment accumulation (+ the RCCL
sum of Q when N > 1).  Ordering, symbolic analysis and the 16 numeric
factorisations are setup and reported separately.  For N > 1 the contour nodes
are sharded over ranks (one process per GPU, launched by torch.distributed.run)
and the partial moments are summed with one all-reduce per iteration; total work
is fixed, so scaling is "strong".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F64_MFMA_PEAK_TF = 78.6      # spec; 77.2 measured (profiles/r01_probe_gfx950.log)


def cpu_baseline(pkg, wl, fo, args):
    """Oracle (numpy/scipy restatement of the reference loop) timed on the host cores of
    this box on a bounded sample: the same operator family on a 24^3 grid, ONE contour node
    factorised and solved (the reference's per-node body, src/feast.jl:59-71), the
    Rayleigh-Ritz block once; iteration time = 16 x node + RR."""
    import scipy.sparse as sp
    g = args.cpu_grid
    w = wl.c3_sparse_generalized(g, args.stencil, 64)
    A = sp.csc_matrix(w["A"], dtype=np.complex128)
    B = sp.csc_matrix(w["B"], dtype=np.complex128)
    con = fo.circular_contour_trapezoidal(w["c"], w["r"], 16)
    n, m0 = A.shape[0], 64
    X = w["X0"].copy()
    t0 = time.perf_counter()
    Q, _ = np.linalg.qr(X, mode="reduced")
    R = A @ Q
    Aq = Q.conj().T @ R
    Bq = Q.conj().T @ (B @ Q)
    lam, Xq = fo._sorted_eig(Aq, Bq)
    Xr = Q @ Xq
    fo.update_R(Xr, R, lam, A, B)
    res = fo.residuals_linear(R)
    t_rr = time.perf_counter() - t0
    t0 = time.perf_counter()
    F = fo.lu(A - B * con.nodes[0])
    t_fact = time.perf_counter() - t0
    temp = np.zeros((n, m0), dtype=np.complex128)
    t0 = time.perf_counter()
    fo.ldiv(temp, F, R)
    temp = (Xr - temp) * ((1.0 / (con.nodes[0] - lam)) * con.weights[0])[None, :]
    t_node = time.perf_counter() - t0
    it_s = 1.0 / (16 * t_node + t_rr)
    # the HIP path on the very same sample, for an apples-to-apples ratio
    h = pkg.FeastHandle(0)
    h.set_terms([w["A"], w["B"]], [1.0, -1.0])
    h.set_contour(pkg.circular_contour_trapezoidal(w["c"], w["r"], 16))
    h.analyze(); h.factorize(); h.set_x(w["X0"])
    for _ in range(2):
        h.rr_step(); h.contour_step()
    t0 = time.perf_counter()
    for _ in range(5):
        h.rr_step(); h.contour_step()
    gpu_it_s = 5.0 / (time.perf_counter() - t0)
    st = h.stats()
    h.close()
    return {
        "value": it_s, "unit": "iterations/s", "cores": 1, "kind": "port",
        "sample": f"same pencil family on a {g}^3 grid (n={n}), 16 nodes, m0=64: one node factorised "
                  f"(SuperLU, {t_fact:.2f} s, setup) and block-solved ({t_node:.3f} s), RR block once ({t_rr:.3f} s); "
                  f"iteration = 16 x node + RR; SuperLU is serial like the reference's UMFPACK solves",
        "host_cpus": os.cpu_count(), "store_false_value": 1.0 / (16 * (t_node + t_fact) + t_rr),
        "gpu_same_sample_value": gpu_it_s, "gpu_same_sample_factor_ms": st["factor_ms"],
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", type=int, default=46)
    ap.add_argument("--stencil", type=int, default=19)
    ap.add_argument("--m0", type=int, default=64)
    ap.add_argument("--nodes", type=int, default=16)
    ap.add_argument("--cpu-baseline", choices=["auto", "off"], default="auto")
    ap.add_argument("--cpu-grid", type=int, default=24)
    ap.add_argument("--no-fold", action="store_true",
                    help="factorise every contour node separately (the conjugate fold of real pencils off): the reference's literal schedule")
    ap.add_argument("--mixed-prec", action="store_true",
                    help="secondary measurement: complex64 shifted solves with complex128 accumulation (feast!'s mixed_prec); not the headline number")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch  # first: libfeasthip.so then shares torch's HIP runtime (same SONAME)
    dist = None
    use_dist = world > 1 or os.environ.get("FEASTHIP_FORCE_DIST") == "1"    # the flag rehearses the RCCL path on one GPU
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    pkg = graft.load_package()
    import importlib
    wl = importlib.import_module(graft.PKG_NAME + ".workloads")
    w = wl.c3_sparse_generalized(args.grid, args.stencil, args.m0)
    n = w["A"].shape[0]
    con = pkg.circular_contour_trapezoidal(w["c"], w["r"], args.nodes)

    h = pkg.FeastHandle(local_rank)
    h.set_terms([w["A"], w["B"]], [1.0, -1.0])
    h.set_contour(con)
    h.set_node_shard(rank, world)
    if args.no_fold:
        h.set_conjugate_fold(False)
    if args.mixed_prec:
        h.set_precision(True)
    ldw = ((args.m0 + 15) // 16) * 16
    qbuf = None
    if use_dist:
        qbuf = torch.zeros(2 * n * ldw * 2, dtype=torch.float64, device=f"cuda:{local_rank}")
        h.set_q_buffer(qbuf.data_ptr(), qbuf.numel() * 8)
    t0 = time.perf_counter()
    h.analyze()
    t_analyze = time.perf_counter() - t0
    warn = h.factorize()
    st = h.stats()
    h.set_x(w["X0"])

    phase = {"rr": 0.0, "contour": 0.0, "allreduce": 0.0}      # host-synchronous phases: the Amdahl breakdown per N
    conv = {"count": 0, "at": None}                              # first Rayleigh-Ritz step whose residuals inside are below eps

    def step():
        t0 = time.perf_counter()
        lam, res, ins = h.rr_step()
        if conv["at"] is None and ins.any() and res[ins].max() < w["eps"]:
            conv["at"] = conv["count"]
        conv["count"] += 1
        t1 = time.perf_counter()
        h.contour_step()
        t2 = time.perf_counter()
        if use_dist:
            dist.all_reduce(qbuf[: n * ldw * 2])
            torch.cuda.synchronize()
        t3 = time.perf_counter()
        phase["rr"] += t1 - t0; phase["contour"] += t2 - t1; phase["allreduce"] += t3 - t2
        return lam, res, ins

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    for k in phase:
        phase[k] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lam, res, ins = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    lam, res, ins = h.rr_step()
    st2 = h.stats()
    prof = h.profile_solve() if rank == 0 else None

    if rank == 0:
        pa = prof["panel_apply"]
        traffic = None          # HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes (profiles/)
        tpath = os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")
        if os.path.exists(tpath) and args.grid == 46 and args.stencil == 19 and world == 1:
            # the PMC run uses two streams (twice the launches of the single-stream profiling sweep): normalise per step
            per_step = json.load(open(tpath)).get("traffic_bytes_per_contour_step")
            traffic = per_step / max(1, pa["launches"]) if per_step else None
        dom_gbs = pa["bytes"] / (pa["ms"] * 1e-3) / 1e9
        dom_tf = pa["flops"] / (pa["ms"] * 1e-3) / 1e12
        out = {
            "metric": "FEAST iterations/sec (16 contour nodes, m0=64)",
            "value": args.steps / dt, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "c64 solves + c128 accumulation (mixed_prec)" if args.mixed_prec else "c128 (f64 complex)", "data": "synthetic",
            "config": {
                "workload": f"C3 sparse generalized: {args.stencil}-point 3-D Laplacian {args.grid}^3, B=diag(U(0.5,1.5)), "
                            f"circular contour {args.nodes} trapezoid nodes, m0={args.m0}, store=true",
                "n": n, "nnz_A": int(w["A"].nnz), "nodes": args.nodes, "m0": args.m0, "nodes_per_gpu": st["local_nodes"],
                "conjugate_fold": bool(st["folded"]), "factorisations_per_gpu": st["factor_units"],
                "nnz_L": st["nnz_l"], "nnz_U": st["nnz_u"], "supernodes": st["nsuper"], "max_front": st["max_front"],
            },
            "roofline": {
                "kernel": "k_apply (supernodal block forward/back substitution, all kinds)",
                "bound": "hbm", "achieved": dom_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom_gbs / HBM_PEAK_GBS,
                "traffic": traffic, "algorithmic_bytes_per_launch": pa["bytes"] / max(1, pa["launches"]),
                "launches_per_step": pa["launches"], "avg_launch_ms": pa["ms"] / max(1, pa["launches"]),
                "f64_mfma": {"achieved": dom_tf, "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": dom_tf / F64_MFMA_PEAK_TF,
                             "note": "m0=64 complex f64 is 32 flop/B (24 with the 3M product): the f64 matrix cores bind before HBM (ceiling 3.2 TB/s of factor bytes); achieved counts 8 flops per complex multiply-add"},
                "contour_step": {"ms": st2["solve_ms"], "algorithmic_GB": st2["solve_bytes"] / 1e9, "streamed_GB_model": st2["solve_bytes_streamed"] / 1e9,
                                 "GBps": st2["solve_bytes"] / (st2["solve_ms"] * 1e-3) / 1e9,
                                 "TFLOPs": st2["solve_flops"] / (st2["solve_ms"] * 1e-3) / 1e12},
                "kernel_classes_ms": {k: v["ms"] for k, v in prof.items()},
            },
            "breakdown_ms_per_step": {k: 1e3 * v / args.steps for k, v in phase.items()},      # rank 0; rr is replicated, contour is this rank's node shard
            # analyze_s: ordering + symbolic analysis + plan upload; alloc_s: hipMalloc of the arenas (0.03 .. 2 s for the same
            # request depending on what the previous process left to scrub) -- reported separately, both count in time_to_solution_s
            "setup": {"analyze_s": st["analyze_seconds"], "alloc_s": st["alloc_seconds"], "analyze_call_s": t_analyze, "factor_ms": st["factor_ms"], "factor_TFLOPs": st["flops_factor"] * st["factor_units"] / (st["factor_ms"] * 1e-3) / 1e12,
                      "perturbed_pivots": int(warn), "store_false_iterations_per_s": 1.0 / (dt / args.steps + st["factor_ms"] * 1e-3)},
            "result": {"inside": int(ins.sum()), "expected_inside": w["n_inside"], "max_res_inside": float(res[ins].max()) if ins.any() else None,
                       "rr_ms": st2["rr_ms"],
                       # eigenpairs/s = converged pairs inside / wall time to convergence of a cold run
                       # (ordering + symbolic analysis, the factorisations, and the iterations up to residual < eps)
                       "iterations_to_eps": conv["at"], "eps": w["eps"],
                       "time_to_solution_s": (t_analyze + st["factor_ms"] * 1e-3 + conv["at"] * dt / args.steps) if conv["at"] is not None else None,
                       "eigenpairs_per_s": (float(ins.sum()) / (t_analyze + st["factor_ms"] * 1e-3 + conv["at"] * dt / args.steps)) if conv["at"] is not None else None},
        }
        if world == 1 and args.cpu_baseline == "auto":
            h.close()
            out["cpu_baseline"] = cpu_baseline(pkg, wl, graft.load_oracle(), args)
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
