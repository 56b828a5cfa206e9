"""Is nlfeast_moments on the deficient quadratic (n = 15, dense, pivoted LU) bit-reproducible from call to call?  Prints a digest of
the eigenvalues, candidates and residuals of four calls in one process (other handles created and closed in between)."""
import hashlib, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
coeffs = wl.deficient_quadratic_coeffs()
T = pkg.PolynomialNEP(coeffs)
for rep in range(4):
    X = wl.rand_x0(15, 4, 6001)
    info = {}
    lam, Y, res = pkg.nlfeast_moments(T, X, 16, 10, c=0j, r=0.25, eps=1e-13, moments=2, info=info)
    d = hashlib.sha256(np.ascontiguousarray(lam).tobytes() + np.ascontiguousarray(Y).tobytes() + np.ascontiguousarray(res).tobytes()).hexdigest()[:16]
    print(rep, info["nit"], d, np.sort(res)[:5], flush=True)
    if rep == 1:      # another problem in between: different allocations, different leftovers in recycled blocks
        w = wl.c3_sparse_generalized(8, 19, 16)
        pkg.gen_feast(w["X0"].copy(), w["A"], w["B"], nodes=8, iter=3, c=w["c"], r=w["r"], store=True)
