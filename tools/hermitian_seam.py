"""lu of one assembled shifted matrix z D - A of a complex Hermitian problem at the plug-in seam (dense, and one node of the Hermitian
C3 pencil of tools/hermitian_c3.py): elimination without interchanges (path 2) against partial pivoting (FEASTHIP_NOPIV_GEN=0)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
rng = np.random.default_rng(0)
for n in (512, 2100, 4096):
    G = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    C = np.asfortranarray((G + G.conj().T) / 2 / np.sqrt(n) - (0.3 + 0.4j) * np.eye(n))
    R = np.asfortranarray(rng.standard_normal((n, 7)) + 1j * rng.standard_normal((n, 7)))
    ts = []
    for rep in range(4):
        t0 = time.perf_counter(); F = pkg.lu(C); ts.append(time.perf_counter() - t0)
        Y = np.zeros_like(R); pkg.ldiv(Y, F, R); path, gr = F.path, F.growth; pkg.finalize(F)
    ref = np.linalg.solve(C, R)
    print(f"dense n={n}: path {path} growth {gr:.2f} error {np.abs(Y - ref).max() / np.abs(ref).max():.1e} lu {min(ts) * 1e3:.2f} ms")
w = wl.c3_sparse_generalized(46, 19, 64)
A = sp.coo_matrix(w["A"]).astype(np.complex128)
up = A.row < A.col
data = A.data.copy(); data[up] *= np.exp(1j * rng.uniform(-0.3, 0.3, A.nnz)[up])
Au = sp.coo_matrix((np.where(up, data, 0), (A.row, A.col)), shape=A.shape).tocsc()
Ah = (Au + Au.conj().T + sp.diags(w["A"].diagonal())).tocsc()
Cm = sp.csc_matrix((w["c"] + w["r"] * np.exp(1j * np.pi / 16)) * w["B"] - Ah)
R = np.asfortranarray(rng.standard_normal((Cm.shape[0], 8)) + 0j)
ts = []
for rep in range(4):
    t0 = time.perf_counter(); F = pkg.lu(Cm); ts.append(time.perf_counter() - t0)
    Y = np.zeros_like(R); pkg.ldiv(Y, F, R); path, gr = F.path, F.growth; pkg.finalize(F)
print(f"one node of the Hermitian C3 pencil: path {path} growth {gr:.2f} residual {np.linalg.norm(Cm @ Y - R) / (spla.norm(Cm) * np.linalg.norm(Y)):.1e} lu {min(ts) * 1e3:.1f} ms")
