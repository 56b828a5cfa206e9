"""Stress run of the early forward sweep (engine.hpp: Engine::early_rhs_; store=false of src/feast.jl:64-65): random real symmetric sparse pencils
(band structure plus random long-range couplings, n = 2,000 .. 40,000, so that the elimination trees have many levels), a contour around
a piece of the spectrum, and per case
  * one contour step that factorises first, with the early sweep and -- in a second handle created under FEASTHIP_EARLY_SWEEP=0 -- without:
    the two moments must be bit for bit the same, on fully and on lazily converted factors, with and without the conjugate fold;
  * the moment against host solves: || T(z_k) Y_k - R || over the nodes, through the moment's defining sum (a direct check that the sweep that
    started under the factorisation solved the right systems).
Usage: python tools/fuzz_early.py [cases] [seed]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 77)
bad = 0
t0 = time.time()


def step(terms, scales, con, X0, lazy, fold, first, mixed=False):
    h = pkg.FeastHandle(0)
    try:
        h.set_terms(terms, scales); h.set_contour(con); h.set_conjugate_fold(fold); h.set_lazy_conversion(lazy)
        if mixed:
            h.set_precision(1)
        if not first:
            h.factorize()                      # the step below factorises AGAIN (every iteration of store=false)
        h.set_x(X0); h.rr_step(); h.contour_step(True)
        return h.get_block(2).copy(), h.get_block(1).copy(), h.get_block(0).copy(), h.stats()      # moment Q, right-hand sides R, Ritz block X
    finally:
        h.close()


for case in range(ncases):
    n = int(rng.integers(2000, 40000))
    offs = sorted(set([1] + [int(v) for v in rng.integers(2, max(3, int(n ** 0.67)), size=int(rng.integers(1, 4)))]))
    S = sp.diags([rng.standard_normal(n - o) for o in offs], offs, shape=(n, n), format="csc")
    extra = int(rng.integers(0, n // 10))
    if extra:
        S = S + sp.coo_matrix((rng.standard_normal(extra) * 0.3, (rng.integers(0, n, extra), rng.integers(0, n, extra))), shape=(n, n)).tocsc()
    A = (S + S.T).tocsc()
    dom = 0.5 + 0.6 * np.abs(A).sum(axis=1).A1 * rng.uniform(0.5, 1.5)          # (as tools/fuzz_sparse_sym.py: the elimination without interchanges is accepted on these)
    A = (A + sp.diags(dom)).tocsc()
    gen = bool(rng.integers(0, 2))
    B = sp.diags(rng.uniform(0.5, 1.5, n)).tocsc() if gen else sp.identity(n, format="csc")
    K = int(rng.choice([4, 8, 16]))
    m0 = int(rng.choice([5, 16, 40, 64]))
    c, r = float(np.quantile(dom, rng.uniform(0.0, 0.05))), float(rng.uniform(0.2, 0.6))          # a disc at the lower end of the diagonal's range
    con = pkg.circular_contour_trapezoidal(c, r, K)
    X0 = wl.rand_x0(n, m0, int(rng.integers(1 << 30)))
    lazy, fold, first = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    # FUZZ_KINDS=1: also unsymmetric values on the same pattern (fronts with row interchanges), Hermitian phases (elimination with L and U apart,
    # adjoint fold) and complex64 sweeps (mixed_prec)
    kind, mixed = "sym", False
    if os.environ.get("FUZZ_KINDS"):
        kind = str(rng.choice(["sym", "unsym", "herm", "sym_mixed"]))
        if kind == "unsym":
            A = sp.csc_matrix(A, dtype=np.complex128)
            A = A + sp.csc_matrix((0.05 * rng.standard_normal(A.nnz), A.indices, A.indptr), shape=A.shape)
        elif kind == "herm":
            A = wl.hermitian_phases(A, seed=int(rng.integers(1 << 20)))
        mixed = kind == "sym_mixed"
    terms, scales = [A, B], [1.0, -1.0]
    try:
        q_e, R, X, st_e = step(terms, scales, con, X0, lazy, fold, first, mixed)
        os.environ["FEASTHIP_EARLY_SWEEP"] = "0"
        q_l, _, _, st_l = step(terms, scales, con, X0, lazy, fold, first, mixed)
        del os.environ["FEASTHIP_EARLY_SWEEP"]
    except Exception as ex:
        os.environ.pop("FEASTHIP_EARLY_SWEEP", None)
        print(f"case {case}: n={n} EXCEPTION {ex}"); bad += 1; continue
    same = np.array_equal(q_e, q_l) and np.isfinite(q_e).all()
    # direct check on the host: Q e_j = sum_k (X e_j - T(z_k)^-1 R e_j) w_k / (z_k - lambda_j) (include/feasthip.h: feasthip_contour_step) lies in the
    # span of X e_j and the K host solutions y_k = T(z_k)^-1 R e_j -- least-squares residual of that (small cases: K sparse host solves)
    err = float("nan")
    if n <= 12000 and not os.environ.get("FUZZ_KINDS"):
        j = int(rng.integers(0, m0))
        Ys = np.stack([spla.spsolve((A - z * B).tocsc().astype(np.complex128), R[:, j]) for z in con.nodes] + [X[:, j]], axis=1)
        coef, *_ = np.linalg.lstsq(Ys, q_e[:, j], rcond=None)
        err = np.linalg.norm(Ys @ coef - q_e[:, j]) / max(np.linalg.norm(q_e[:, j]), 1e-300)
    ok = same and st_l["early_sweep_levels"] == 0 and (not (err == err) or err < (1e-4 if mixed else 1e-9))
    print(f"case {case}: n={n} offsets={offs} extra={extra} gen={gen} kind={kind} K={K} m0={m0} lazy={lazy} fold={fold} first={first} symmetric_path {st_e['symmetric']} "
          f"early levels {st_e['early_sweep_levels']} depth {st_e['depth']} bitwise {same} span residual {err:.1e} {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print(f"{ncases} cases, {bad} bad, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
