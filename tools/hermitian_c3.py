"""The C3 pencil with a complex Hermitian A (random phases on the off-diagonal entries: a gauge field on the same grid): the shifted
matrices are neither symmetric nor conjugate pairs, so the path is 16 pivoted factorisations -- what such a problem costs today."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
w = wl.c3_sparse_generalized(46, 19, 64)
A = sp.coo_matrix(w["A"]).astype(np.complex128)
rng = np.random.default_rng(3)
theta = rng.uniform(-0.3, 0.3, A.nnz)
up = A.row < A.col
data = A.data.copy()
data[up] *= np.exp(1j * theta[up])
Au = sp.coo_matrix((np.where(up, data, 0), (A.row, A.col)), shape=A.shape).tocsc()
Ah = (Au + Au.conj().T + sp.diags(w["A"].diagonal())).tocsc()
assert abs(Ah - Ah.conj().T).max() < 1e-14
con = pkg.circular_contour_trapezoidal(w["c"], w["r"], 16)
for rep in range(2):
    info = {}
    t0 = time.perf_counter()
    lam, V, res = pkg.gen_feast(w["X0"].copy(), Ah, w["B"], con, iter=20, eps=1e-10, store=True, info=info)
    dt = time.perf_counter() - t0
    print(f"Hermitian C3: call {dt:.3f} s, {len(lam)} eigenpairs, {info['nit']} iterations, max residual {res.max():.1e}, factorisations {info['factor_units'] if 'factor_units' in info else '?'} "
          f"in {info['factor_ms']:.0f} ms (symmetric path {info['symmetric']}), contour step {info['solve_ms']:.1f} ms, Rayleigh-Ritz {info['rr_ms']:.2f} ms, max |Im lam| {np.abs(lam.imag).max():.1e}")
