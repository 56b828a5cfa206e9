"""Randomised run of nlfeast on polynomial eigenvalue problems T(z) = sum_j z^j M_j (degree 2 or 3, dense or sparse coefficients)
against the spectrum of the companion pencil (src/companion.jl:1-28).  Usage: python tools/fuzz_nlfeast.py [cases] [seed]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, scipy.linalg as sla
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = 0
t0 = time.time()
_nan = None
if os.environ.get("FUZZ_LDS_NAN"):      # leave NaN in the LDS of every CU before each case (stale-LDS reads: tests/test_gpu_parity.py, test_narrow_blocks_...)
    import torch
    _nan = torch.full((2048, 2048), float("nan"), device="cuda", dtype=torch.float64)
for case in range(ncases):
    if _nan is not None: (_nan @ _nan).sum().item(); torch.softmax(_nan, 1).sum().item()
    n = int(rng.choice([8, 20, 33, 64, 100, 150]))
    deg = int(rng.choice([2, 3]))
    sparse = bool(rng.integers(0, 2)) and n >= 20
    cplx = bool(rng.integers(0, 2))
    Ms = []
    for j in range(deg + 1):
        if sparse:
            M = sp.random(n, n, density=min(1.0, 6.0 / n), random_state=np.random.RandomState(int(rng.integers(1 << 30))), data_rvs=lambda k: rng.standard_normal(k)).toarray()
        else:
            M = rng.standard_normal((n, n)) / np.sqrt(n)
        if cplx: M = M + 1j * rng.standard_normal((n, n)) * (M != 0)
        M = M + np.eye(n) * (1.0 + j)
        Ms.append(M.astype(np.complex128))
    # companion pencil C1 x = lam C2 x (first-companion form): eigenvalues of the NEP
    N = n * deg
    C1 = np.zeros((N, N), complex); C2 = np.zeros((N, N), complex)
    C1[:n, :n] = Ms[0]
    for i in range(n, N): C1[i, i] = 1; C2[i, i - n] = 1
    for i in range(deg): C2[:n, n * i:n * (i + 1)] = -Ms[i + 1]
    lam_all = sla.eigvals(C1, C2)
    lam_all = lam_all[np.isfinite(lam_all)]
    c = lam_all[int(rng.integers(0, len(lam_all)))]
    dist = np.sort(np.abs(lam_all - c))
    kin = int(rng.integers(1, 5))
    if kin >= len(dist): continue
    r = 0.5 * (dist[kin - 1] + dist[kin])
    if dist[kin] - dist[kin - 1] < 0.25 * r or r <= 0: continue
    m0 = min(n, 2 * kin + 2)
    X0 = wl.rand_x0(n, m0, int(rng.integers(1 << 30)))
    coeffs = [sp.csc_matrix(M) for M in Ms] if sparse else Ms
    T = pkg.PolynomialNEP(coeffs)
    info = {}
    try:
        lam, X, res = pkg.nlfeast(T, X0.copy(), 16, 30, c=complex(c), r=float(r), eps=1e-10, spurious=1e-5, info=info)
    except Exception as ex:
        print(f"case {case}: n={n} deg={deg} sparse={sparse} cplx={cplx} EXCEPTION {ex}"); bad += 1; continue
    ins = (np.abs(lam - c) <= r) & (res < 1e-7)
    ref = lam_all[np.abs(lam_all - c) <= r]
    ok = ins.sum() == len(ref) and max(np.abs(lam[ins] - x).min() for x in ref) < 1e-6 * max(1.0, abs(c))
    okm, fm = True, -1
    if os.environ.get("FUZZ_MOMENTS"):     # the same problem through nlfeast_moments (src/nlfeast.jl:173-318): a smaller subspace (2 moments), candidates by residual
        mm = max(2, kin + 1)
        try:
            lm, Ym, rm = pkg.nlfeast_moments(T, wl.rand_x0(n, mm, int(rng.integers(1 << 30))), 16, 30, c=complex(c), r=float(r), eps=1e-10, moments=2, spurious=1e-5)
            im = (np.abs(lm - c) <= r) & (rm < 1e-7)
            fm = int(im.sum())
            okm = fm == len(ref) and max(np.abs(lm[im] - x).min() for x in ref) < 1e-6 * max(1.0, abs(c))
        except Exception as ex:
            print(f"case {case}: nlfeast_moments EXCEPTION {ex}"); okm = False
    print(f"case {case}: n={n} deg={deg} sparse={sparse} cplx={cplx} m0={m0} inside {len(ref)} found {int(ins.sum())} nit {info.get('nit')} moments found {fm} {'ok' if ok and okm else 'MISMATCH'}", flush=True)
    bad += 0 if ok and okm else 1
print(f"{ncases} cases, {bad} bad, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
