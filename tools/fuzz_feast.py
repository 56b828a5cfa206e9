"""Randomised differential run: gen_feast on random small / medium pencils (sparse and dense, symmetric and not, real and complex,
standard and generalized, odd sizes and block widths) against a dense host eigensolve.  Prints one line per case and a summary;
exit status 1 if any case disagrees.  Usage: python tools/fuzz_feast.py [cases] [seed] [max n]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, scipy.linalg as sla
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
sizes = [v for v in (3, 17, 33, 64, 100, 129, 257, 400, 777, 1200, 2000) if v <= nmax]
bad = 0
t_start = time.time()
_nan = None
if os.environ.get("FUZZ_LDS_NAN"):      # leave NaN in the LDS of every CU before each case (stale-LDS reads: tests/test_gpu_parity.py, test_narrow_blocks_...)
    import torch
    _nan = torch.full((2048, 2048), float("nan"), device="cuda", dtype=torch.float64)
for case in range(ncases):
    if _nan is not None: (_nan @ _nan).sum().item(); torch.softmax(_nan, 1).sum().item()
    n = int(rng.choice(sizes))
    sparse = bool(rng.integers(0, 2)) and n >= 17
    sym = bool(rng.integers(0, 2))
    cplx = bool(rng.integers(0, 2))
    gen = bool(rng.integers(0, 2))
    K = int(rng.choice([4, 8, 16]))
    if sparse:
        dens = min(1.0, float(rng.uniform(2.0, 12.0)) / n)
        S = sp.random(n, n, density=dens, random_state=np.random.RandomState(int(rng.integers(1 << 30))), data_rvs=lambda k: rng.standard_normal(k))
        if cplx: S = S + 1j * sp.random(n, n, density=dens, random_state=np.random.RandomState(int(rng.integers(1 << 30))), data_rvs=lambda k: rng.standard_normal(k))
        A = (S + (S.T if not cplx or not sym else S.conj().T)) if sym else (S + 0.3 * S.T)
        A = (A + sp.diags(np.arange(n) * 0.05)).tocsc()
        Ad = A.toarray()
    else:
        G = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cplx else 0)
        Ad = (G + G.conj().T) / 2 if sym else G
        Ad = Ad / np.sqrt(n) + np.diag(np.arange(n) * (2.0 / n))
        A = Ad
    if gen:
        bd = rng.uniform(0.5, 1.5, n)
        B = sp.diags(bd).tocsc() if sparse else np.diag(bd)
        Bd = np.diag(bd)
    else:
        B, Bd = None, np.eye(n)
    lam_all = sla.eigvals(Ad, Bd)
    # a disc around a random eigenvalue holding 1..6 eigenvalues, with a gap to the next one
    c = lam_all[int(rng.integers(0, n))]
    dist = np.sort(np.abs(lam_all - c))
    kin = int(rng.integers(1, min(6, n) + 1))
    if kin >= n: kin = n - 1 if n > 1 else 1
    r = 0.5 * (dist[kin - 1] + dist[kin]) if kin < n else dist[-1] * 1.5 + 0.1
    if kin < n and (dist[kin] - dist[kin - 1]) < 0.2 * r:
        continue                                    # eigenvalues within 10 % of the contour on either side: FEAST itself converges slowly or not at
                                                    # all there with few nodes (the CPU oracle shows the same histories), nothing to compare
    m0 = min(n, int(kin * rng.uniform(1.5, 3.0)) + 2)
    X0 = wl.rand_x0(n, m0, int(rng.integers(1 << 30)))
    # FUZZ_CONTOURS=1: also the reference's other contour constructors (src/contour.jl:33-86), a square of half width r around c for the
    # rectangular ones (open rectangle: eigenvalues within 10 % of r of an edge: skipped)
    ckind = "circ_trap"
    if os.environ.get("FUZZ_CONTOURS"):
        ckind = str(rng.choice(["circ_trap", "circ_gauss", "rect_trap", "rect_gauss"]))
        if ckind.startswith("rect"): K = 16
    if ckind.startswith("rect"):
        dre, dim = np.abs(lam_all.real - c.real), np.abs(lam_all.imag - c.imag)
        near = (np.maximum(dre, dim) > 0.9 * r) & (np.maximum(dre, dim) < 1.1 * r)
        if near.any():
            continue
        con = (pkg.rectangular_contour_trapezoidal if ckind == "rect_trap" else pkg.rectangular_contour_gauss)(complex(c) - r * (1 + 1j), complex(c) + r * (1 + 1j), K)
    else:
        con = (pkg.circular_contour_gauss if ckind == "circ_gauss" else pkg.circular_contour_trapezoidal)(complex(c), float(r), K)
    info = {}
    mixed = (not gen) and bool(rng.integers(0, 4) == 0)          # feast!'s mixed_prec: complex64 shifted solves, complex128 accumulation
    store = bool(rng.integers(0, 2))
    if os.environ.get("FUZZ_ONLY") and case != int(os.environ["FUZZ_ONLY"]):
        continue                                    # (replays one case of a run: the random stream is consumed as in the full run)
    if os.environ.get("FUZZ_DUMP"):                # inputs of the replayed case for a stand-alone reproducer
        np.savez(os.environ["FUZZ_DUMP"], A=Ad, B=Bd, c=complex(c), r=float(r), K=K, m0=m0, X0=X0, gen=gen, mixed=mixed, store=store, sparse=sparse)
    try:
        lam, V, res = pkg.gen_feast(X0.copy(), A, B, con, iter=int(os.environ.get("FUZZ_ITER", "40")), eps=1e-9, store=store, info=info) if gen else \
                      pkg.feast(X0.copy(), A, con, iter=int(os.environ.get("FUZZ_ITER", "40")), eps=1e-9, store=store, mixed_prec=mixed, info=info)
    except Exception as ex:
        print(f"case {case}: n={n} sparse={sparse} sym={sym} cplx={cplx} gen={gen} K={K} m0={m0} EXCEPTION {ex}")
        bad += 1
        continue
    ref = lam_all[(np.maximum(np.abs(lam_all.real - c.real), np.abs(lam_all.imag - c.imag)) < r) if ckind.startswith("rect") else (np.abs(lam_all - c) <= r)]
    ok = len(lam) == len(ref) and (len(ref) == 0 or max(np.abs(lam - x).min() for x in ref) < 1e-6 * max(1.0, abs(c)))
    if len(lam):
        Rres = np.linalg.norm(Ad @ V - (Bd @ V) * lam[None, :], axis=0).max()
        ok = ok and Rres < 1e-6 * max(1.0, np.abs(Ad).max())
    else:
        Rres = float("nan")
    niter = int(os.environ.get("FUZZ_ITER", "40"))
    if not ok and info.get("nit") == niter and len(lam) == len(ref) and ckind == "circ_trap" and not mixed:
        # not converged after all its iterations: a slow case of the algorithm (few nodes, a non-normal matrix, an eigenvalue near the contour)
        # or a wrong sweep?  The CPU oracle (checker) on the same inputs tells: the same residual after the same number of iterations = slow.
        fo = graft.load_oracle()
        lo, Vo, ro = fo.gen_feast(X0.copy(), Ad, (Bd if gen else None), fo.circular_contour_trapezoidal(complex(c), float(r), K), iter=niter, eps=1e-9, store=True)
        if len(lo) == len(lam) and len(ro) and len(res) and 0.1 < res.max() / ro.max() < 10.0:
            print(f"case {case}: not converged after {niter} iterations, residual {res.max():.3e}; the oracle stands at {ro.max():.3e}: slow convergence, not a difference", flush=True)
            ok = True
    print(f"case {case}: n={n} sparse={sparse} sym={sym} cplx={cplx} gen={gen} mixed={mixed} K={K} m0={m0} inside {len(ref)} found {len(lam)} nit {info.get('nit')} "
          f"res {Rres:.1e} symmetric_path {info.get('symmetric')} store {store} {ckind} {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print(f"{ncases} cases, {bad} bad, {time.time() - t_start:.0f} s")
sys.exit(1 if bad else 0)
