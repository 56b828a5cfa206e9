"""Wall time of the user-level call gen_feast(X, A, B, contour; store=true) on C3, first and later calls of a process (the later
ones find the ordering + symbolic analysis of the pattern in the process-wide cache)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
w = wl.c3_sparse_generalized(46, 19, 64)
con = pkg.circular_contour_trapezoidal(w["c"], w["r"], 16)
for rep in range(4):
    info = {}
    X = w["X0"].copy(order="F")         # column-major like the reference's Matrix{ComplexF64} (a C-ordered block costs two more 100 MB copies per call: 35 ms)
    t0 = time.perf_counter()
    lam, V, res = pkg.gen_feast(X, w["A"], w["B"], con, iter=w["iter"], eps=w["eps"], store=True, info=info)
    dt = time.perf_counter() - t0
    print(f"call {rep}: {dt:.3f} s, {len(lam)} eigenpairs, max residual {res.max():.2e}, {info['nit']} iterations, "
          f"analysis {info.get('analyze_seconds', float('nan')):.3f} s, factorisation {info.get('factor_ms', float('nan')):.1f} ms -> {len(lam) / dt:.0f} eigenpairs/s")
