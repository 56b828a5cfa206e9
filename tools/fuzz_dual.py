"""Randomised differential run of dual_gen_feast (src/feast.jl:158-257): random non-Hermitian pencils (dense and sparse, real and complex,
B = I / diagonal / general) against scipy's two-sided eigendecomposition: eigenvalues inside the contour, right residuals A x - lam B x and
left residuals y' A - lam y' B.  Usage: python tools/fuzz_dual.py [cases] [seed]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, scipy.linalg as sla
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 31)
bad = 0; t0 = time.time()
for case in range(ncases):
    n = int(rng.choice([17, 64, 100, 129, 257, 400, 777]))
    sparse = bool(rng.integers(0, 2))
    cplx = bool(rng.integers(0, 2))
    bkind = str(rng.choice(["identity", "diagonal", "general"]))
    K = int(rng.choice([8, 16]))
    if sparse:
        dens = min(1.0, float(rng.uniform(3.0, 10.0)) / n)
        S = sp.random(n, n, density=dens, random_state=np.random.RandomState(int(rng.integers(1 << 30))), data_rvs=lambda k: rng.standard_normal(k))
        if cplx: S = S + 1j * sp.random(n, n, density=dens, random_state=np.random.RandomState(int(rng.integers(1 << 30))), data_rvs=lambda k: rng.standard_normal(k))
        A = (S + 0.3 * S.T + sp.diags(np.arange(n) * 0.05)).tocsc(); Ad = A.toarray()
    else:
        G = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cplx else 0)
        Ad = G / np.sqrt(n) + np.diag(np.arange(n) * (2.0 / n)); A = Ad
    if bkind == "identity": Bd = np.eye(n)
    elif bkind == "diagonal": Bd = np.diag(rng.uniform(0.5, 1.5, n))
    else: Bd = np.eye(n) + 0.1 * (rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cplx else 0)) / np.sqrt(n)
    B = sp.csc_matrix(Bd) if (sparse and bkind != "general") else Bd
    if sparse and bkind == "general": A = Ad          # dense B: dense pencil
    lam_all = sla.eigvals(Ad, Bd)
    c = lam_all[int(rng.integers(0, n))]
    dist = np.sort(np.abs(lam_all - c))
    kin = int(rng.integers(1, min(5, n - 1) + 1))
    r = 0.5 * (dist[kin - 1] + dist[kin])
    if (dist[kin] - dist[kin - 1]) < 0.2 * r:
        continue
    m0 = min(n, int(kin * rng.uniform(1.5, 3.0)) + 2)
    Xr, Xl = wl.rand_x0(n, m0, int(rng.integers(1 << 30))), wl.rand_x0(n, m0, int(rng.integers(1 << 30)))
    con = pkg.circular_contour_trapezoidal(complex(c), float(r), K)
    info = {}
    store = bool(rng.integers(0, 2))
    try:
        lam, Vr, Vl, resr = pkg.dual_gen_feast(Xr.copy(), Xl.copy(), A, B, con, iter=40, eps=1e-9, store=store, info=info)
    except Exception as ex:
        print(f"case {case}: n={n} sparse={sparse} cplx={cplx} B={bkind} EXCEPTION {ex}"); bad += 1; continue
    ref = lam_all[np.abs(lam_all - c) <= r]
    ok = len(lam) == len(ref) and (len(ref) == 0 or max(np.abs(lam - x).min() for x in ref) < 1e-6 * max(1.0, abs(c)))
    rr = rl = float("nan")
    if len(lam):
        sc = max(1.0, np.abs(Ad).max())
        rr = (np.linalg.norm(Ad @ Vr - (Bd @ Vr) * lam[None, :], axis=0) / np.linalg.norm(Vr, axis=0)).max()
        rl = (np.linalg.norm(Ad.conj().T @ Vl - (Bd.conj().T @ Vl) * lam.conj()[None, :], axis=0) / np.linalg.norm(Vl, axis=0)).max()
        ok = ok and rr < 1e-6 * sc and rl < 1e-6 * sc
    print(f"case {case}: n={n} sparse={sparse} cplx={cplx} B={bkind} K={K} m0={m0} store={store} inside {len(ref)} found {len(lam)} nit {info.get('nit')} right res {rr:.1e} left res {rl:.1e} {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print(f"{ncases} cases, {bad} bad, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
