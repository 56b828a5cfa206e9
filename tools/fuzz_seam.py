"""Randomised run of the factorizer / left_divider seam: lu + ldiv of random matrices (dense and sparse, symmetric and not, complex128 and
complex64, sizes around the 32 / 64 / 128 / 256 / 512 block edges) against a host solve.  Usage: python tools/fuzz_seam.py [cases] [seed]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import __graft_entry__ as graft
pkg = graft.load_package()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
edges = [1, 2, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 191, 192, 193, 255, 256, 257, 383, 385, 511, 512, 513, 640, 767, 769, 1023, 1025, 1300]
bad = 0
t0 = time.time()
_nan = None
if os.environ.get("FUZZ_LDS_NAN"):      # leave NaN in the LDS of every CU before each case (stale-LDS reads: tests/test_gpu_parity.py, test_narrow_blocks_...)
    import torch
    _nan = torch.full((2048, 2048), float("nan"), device="cuda", dtype=torch.float64)
for case in range(ncases):
    if _nan is not None: (_nan @ _nan).sum().item(); torch.softmax(_nan, 1).sum().item()
    n = int(rng.choice(edges)) if rng.integers(0, 3) else int(rng.integers(1, 1500))
    sparse = bool(rng.integers(0, 2))
    sym = bool(rng.integers(0, 2))
    c64 = bool(rng.integers(0, 4) == 0)
    m = int(rng.choice([1, 3, 16, 37, 64, 100]))
    shift = complex(rng.uniform(-0.5, 0.5), rng.uniform(0.2, 0.6))
    if sparse:
        dens = min(1.0, float(rng.uniform(1.0, 10.0)) / max(n, 1))
        S = sp.random(n, n, density=dens, random_state=np.random.RandomState(int(rng.integers(1 << 30))), data_rvs=lambda k: rng.standard_normal(k))
        A = (S + S.T) if sym else (S + 0.5 * S.T + sp.random(n, n, density=dens / 2, random_state=np.random.RandomState(int(rng.integers(1 << 30)))))
        dom = 1.0 + np.abs(A).sum(axis=1).A1
        Cm = sp.csc_matrix(A.astype(np.complex128) + sp.diags(dom) - shift * sp.identity(n))
        Cd = Cm.toarray()
    else:
        G = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        G = (G + G.T) / 2 if sym else G
        Cd = G / np.sqrt(max(n, 1)) + np.diag(rng.uniform(1.0, 2.0, n)) - shift * np.eye(n)
        Cm = np.asfortranarray(Cd)
    if c64:
        Cm = Cm.astype(np.complex64)
        Cd = np.asarray(Cm.toarray() if sparse else Cm, dtype=np.complex128)
    R = np.asfortranarray(rng.standard_normal((n, m)) + 1j * rng.standard_normal((n, m)))
    Y = np.zeros((n, m), dtype=np.complex64 if c64 else np.complex128, order="F")
    try:
        F = pkg.lu(Cm)
        pkg.ldiv(Y, F, R)
        symflag = F.symmetric
        pkg.finalize(F)
    except Exception as ex:
        print(f"case {case}: n={n} sparse={sparse} sym={sym} c64={c64} m={m} EXCEPTION {ex}"); bad += 1; continue
    ref = np.linalg.solve(Cd, R)
    err = np.abs(Y - ref).max() / max(1e-300, np.abs(ref).max())
    tol = 2e-4 if c64 else 1e-10
    ok = err < tol
    if not ok or case % 10 == 0:
        print(f"case {case}: n={n} sparse={sparse} sym={sym} c64={c64} m={m} symmetric_path {symflag} rel err {err:.1e} {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print(f"{ncases} cases, {bad} bad, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
