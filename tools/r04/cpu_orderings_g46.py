"""SuperLU column orderings on the metric's own 46^3 pencil, one contour node (the CPU baseline's factorisation + 64-column solve):
COLAMD (scipy's default) against MMD(A^T + A).  Run in the BUILD container (8 host cores; the GPU box's host measured COLAMD in round 3:
profiles/r03_cpu_baseline_c3_g46.json) -- what matters here is the ratio and the factor entries."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "feastsolver.jl_amd"))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import workloads as wl
g = int(sys.argv[1]) if len(sys.argv) > 1 else 46
w = wl.c3_sparse_generalized(g, 19, 64)
z = w["c"] + w["r"] * np.exp(1j * np.pi / 16)
C = sp.csc_matrix(w["A"].astype(complex) - z * w["B"])
R = np.random.default_rng(0).standard_normal((g ** 3, 64)) + 0j
out = {"grid": g, "n": g ** 3, "host": "build container", "host_cpus": os.cpu_count(), "orderings": {}}
for spec in ("MMD_AT_PLUS_A", "COLAMD"):
    t = time.perf_counter(); f = spla.splu(C, permc_spec=spec); t1 = time.perf_counter() - t
    t = time.perf_counter(); f.solve(R); t2 = time.perf_counter() - t
    out["orderings"][spec] = {"t_factor_node_s": t1, "t_solve_node_s": t2, "nnz_LU": int(f.L.nnz + f.U.nnz)}
    print(spec, out["orderings"][spec], flush=True)
    del f
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "profiles", f"r04_cpu_orderings_g{g}_build_container.json"), "w"), indent=1)
