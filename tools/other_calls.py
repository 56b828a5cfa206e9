"""Wall time of the other user-level calls of the boundary (second call of each in the process): nlfeast_moments on system5 (sparse
quadratic, n = 1000), dual_gen_feast and contour_estimate_eig on C3, nlfeast on a closure NEP (hadeler, split form)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")

def timed(name, fn, reps=2):
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); dt = time.perf_counter() - t0
    print(f"{name}: {dt * 1e3:.1f} ms", flush=True)
    return out

coeffs = wl.system5_coeffs(sparse=True)
T = pkg.PolynomialNEP(coeffs)
X0 = wl.rand_x0(1000, 32, 77)
info = {}
out = timed("nlfeast_moments system5 (n=1000 sparse quadratic, 16 nodes, m0=32, 2 moments)",
            lambda: pkg.nlfeast_moments(T, X0.copy(), 16, 10, c=-1.55 + 0j, r=0.05, eps=1e-10, moments=2, info=info))
print("   iterations", info.get("nit"), "inside", int(np.sum(np.abs(out[0] - (-1.55)) <= 0.05)))
w = wl.c3_sparse_generalized(46, 19, 64)
con = pkg.circular_contour_trapezoidal(w["c"], w["r"], 16)
timed("contour_estimate_eig C3 (n=97,336, 16 nodes, 64 samples)", lambda: pkg.contour_estimate_eig(w["A"], con, w["B"], samples=64, seed=1))
info = {}
timed("dual_gen_feast C3 (right + left subspaces)", lambda: pkg.dual_gen_feast(w["X0"].copy(), w["X0"].copy(), w["A"], w["B"], con, iter=20, eps=1e-10, store=True, info=info))
print("   iterations", info.get("nit"), {k: info.get(k) for k in ("factor_units", "factor_ms", "solve_ms", "rr_ms", "analyze_seconds", "alloc_seconds", "symmetric", "folded")})
