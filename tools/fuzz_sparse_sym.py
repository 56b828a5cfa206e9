"""Stress run of the pivot-light sparse factorisation: random complex symmetric matrices (grid-like band structure plus random long-range
couplings, n = 2,000 .. 60,000) through lu + ldiv, checked by the residual.  Usage: python tools/fuzz_sparse_sym.py [cases] [seed]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import __graft_entry__ as graft
pkg = graft.load_package()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 21)
bad = 0
t0 = time.time()
_nan = None
if os.environ.get("FUZZ_LDS_NAN"):      # leave NaN in the LDS of every CU before each case (stale-LDS reads: tests/test_gpu_parity.py, test_narrow_blocks_...)
    import torch
    _nan = torch.full((2048, 2048), float("nan"), device="cuda", dtype=torch.float64)
for case in range(ncases):
    if _nan is not None: (_nan @ _nan).sum().item(); torch.softmax(_nan, 1).sum().item()
    n = int(rng.integers(2000, 60000))
    offs = sorted(set([1] + [int(v) for v in rng.integers(2, max(3, int(n ** 0.67)), size=int(rng.integers(1, 4)))]))
    diags = [rng.standard_normal(n - o) for o in offs]
    S = sp.diags(diags, offs, shape=(n, n), format="csc")
    extra = int(rng.integers(0, n // 10))
    if extra:
        i = rng.integers(0, n, extra); j = rng.integers(0, n, extra)
        S = S + sp.coo_matrix((rng.standard_normal(extra) * 0.3, (i, j)), shape=(n, n)).tocsc()
    A = (S + S.T).tocsc()
    dom = 0.5 + 0.6 * np.abs(A).sum(axis=1).A1 * rng.uniform(0.5, 1.5)
    Cm = sp.csc_matrix(A.astype(np.complex128) + sp.diags(dom) - complex(rng.uniform(-0.3, 0.3), rng.uniform(0.1, 0.5)) * sp.identity(n))
    m = int(rng.choice([4, 16, 64]))
    R = np.asfortranarray(rng.standard_normal((n, m)) + 1j * rng.standard_normal((n, m)))
    Y = np.zeros_like(R)
    try:
        t1 = time.time(); F = pkg.lu(Cm); pkg.ldiv(Y, F, R); symf, gr = F.symmetric, F.growth; pkg.finalize(F); dt = time.time() - t1
    except Exception as ex:
        print(f"case {case}: n={n} offsets={offs} extra={extra} EXCEPTION {ex}"); bad += 1; continue
    rel = np.linalg.norm(Cm @ Y - R) / (spla.norm(Cm) * np.linalg.norm(Y))
    ok = rel < 1e-12
    print(f"case {case}: n={n} offsets={offs} extra={extra} m={m} symmetric_path {symf} growth {gr:.2f} residual {rel:.1e} {dt * 1e3:.0f} ms {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print(f"{ncases} cases, {bad} bad, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
