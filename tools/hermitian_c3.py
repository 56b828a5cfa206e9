"""The C3 pencil with a complex Hermitian A (workloads.c3_hermitian: random phases on the couplings, a gauge field on the same grid).
The shifted matrices are neither symmetric nor conjugate pairs; round 4: elimination without interchanges (path 2) and the ADJOINT fold
(T(conj z) = T(z)^H: 8 factorisations for the 16 nodes, the mirror nodes solved by U^H / L^H sweeps).  FEASTHIP_ADJ_FOLD=0 for the A/B."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
w = wl.c3_hermitian(46, 19, 64)
con = pkg.circular_contour_trapezoidal(w["c"], w["r"], 16)
for rep in range(2):
    info = {}
    t0 = time.perf_counter()
    lam, V, res = pkg.gen_feast(w["X0"].copy(), w["A"], w["B"], con, iter=20, eps=1e-10, store=True, info=info)
    dt = time.perf_counter() - t0
    print(f"Hermitian C3: call {dt:.3f} s, {len(lam)} eigenpairs ({w['n_inside']} expected), {info['nit']} iterations, max residual {res.max():.1e}, "
          f"{info['factor_units']} factorisations in {info['factor_ms']:.0f} ms (path {info['symmetric']}, fold {info['folded']}), "
          f"contour step {info['solve_ms']:.1f} ms, Rayleigh-Ritz {info['rr_ms']:.2f} ms, max |Im lam| {np.abs(lam.imag).max():.1e}", flush=True)
