"""Pins the CPU oracle (oracle/feast_oracle.py) against the reference's own known
answers (test/runtests.jl) and against companion-linearisation spectra of the
reference's data files (src/companion.jl).  CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp


def c2n(pairs):
    return np.array([complex(a, b) for a, b in pairs])


def test_runtests_dense_standard(fo, wl):
    # test/runtests.jl:16-20
    A = np.diag(np.arange(1, 26))
    e, v, res = fo.feast(wl.rand_x0(25, 5, 101), A, fo.circular_contour_trapezoidal(1.5, 2.0, 8), iter=10)
    assert np.allclose(np.sort(e.real), [1, 2, 3])
    assert np.sort(res)[:3].max() < 1e-12


def test_runtests_dense_generalized(fo, wl):
    # test/runtests.jl:21-23
    A = np.diag(np.arange(1, 26)).astype(float)
    e, v, res = fo.gen_feast(wl.rand_x0(25, 5, 102), A, np.eye(25), fo.circular_contour_trapezoidal(1.5, 2, 8), iter=100)
    assert np.sort(res).max() < 1e-12
    assert np.allclose(np.sort(e.real), [1, 2, 3])


def test_runtests_sparse_contours(fo, wl, golden):
    # test/runtests.jl:33-49
    A = sp.diags([-np.ones(99), 2 * np.ones(100), -np.ones(99)], [-1, 0, 1], format="csc")
    Cc, R = 0.05, 0.05
    ana = np.array(golden["runtests_laplace1d_analytic"])
    for k, con in enumerate([fo.circular_contour_trapezoidal(Cc, R, 8),
                             fo.rectangular_contour_trapezoidal(0.0 - R * 1j, 2 * R + R * 1j, 8),
                             fo.rectangular_contour_gauss(0.0 - R * 1j, 2 * R + R * 1j, 8)]):
        e, v, res = fo.feast(wl.rand_x0(100, 20, 110 + k), A, con, eps=10e-15)
        assert e.shape[0] == 10
        assert res.max() < 10e-15
        assert np.abs(np.sort(e.real) - ana).max() < 1e-13


def test_contour_weights_integrate_resolvent(fo):
    # sum_k w_k/(z_k - z) ~ 1 inside, ~ 0 outside (src/contour.jl:102-108 rational_func)
    for con in [fo.circular_contour_trapezoidal(0.2, 1.0, 32), fo.circular_contour_gauss(0.2, 1.0, 32),
                fo.rectangular_contour_trapezoidal(-1 - 1j, 1 + 1j, 64), fo.rectangular_contour_gauss(-1 - 1j, 1 + 1j, 64)]:
        f = lambda z: np.sum(con.weights / (con.nodes - z))
        assert abs(f(0.1 + 0.05j) - 1) < 1e-3
        assert abs(f(5.0)) < 1e-3


def test_butterfly_companion_and_nlfeast(fo, wl, golden):
    w = wl.c4_butterfly()
    cl, cx, cr = fo.companion(w["coeffs"])
    ins = np.abs(cl - w["c"]) <= w["r"]
    assert ins.sum() == 13
    assert cr[ins].max() < 1e-10
    lam, X, res = fo.nlfeast(fo.poly_T(w["coeffs"]), w["X0"].copy(), 16, 10, c=w["c"], r=w["r"], eps=w["eps"])
    li = np.abs(lam - w["c"]) <= w["r"]
    assert li.sum() == 13 and res[li].max() < 1e-10
    d = [np.abs(cl[ins] - l).min() for l in lam[li]]
    assert max(d) < 1e-11
    g = c2n(golden["c4_butterfly"]["companion_inside"])
    assert np.abs(np.sort_complex(cl[ins]) - g).max() < 1e-10


def test_deficient_and_system5_counts(fo, wl, golden):
    cl, _, _ = fo.companion(wl.deficient_quadratic_coeffs())
    fin = np.isfinite(cl)
    assert (np.abs(cl[fin]) <= 0.25).sum() == 4
    assert len(golden["system5"]["companion_inside"]) == 50


@pytest.mark.parametrize("case,comp_key", [("moments_deficient_quadratic", "deficient_quadratic"),
                                           ("moments_butterfly", "c4_butterfly"), ("moments_system5", "system5")])
def test_nlfeast_moments_matches_companion(fo, wl, golden, case, comp_key):
    """nlfeast_moments! (src/nlfeast.jl:173-318) on the reference's data files finds exactly the eigenvalues the
    companion linearisation (src/companion.jl) has inside the contour, and reproduces its committed fixture."""
    g = golden[case]
    coeffs = {"moments_deficient_quadratic": wl.deficient_quadratic_coeffs, "moments_butterfly": lambda: wl.c4_butterfly()["coeffs"],
              "moments_system5": lambda: wl.system5_coeffs(sparse=True)}[case]()
    X = wl.rand_x0(g["n"], g["m0"], g["seed"])
    c = complex(*g["c"])
    hist = []
    lam, Y, res = fo.nlfeast_moments(fo.poly_T(coeffs), X, g["nodes"], g["iter"], c=c, r=g["r"], eps=g["eps"], moments=g["moments"],
                                     history=hist)
    M = g["moments"] * g["m0"]
    assert lam.shape == (M,) and Y.shape == (g["n"], M) and np.all(np.diff(res) >= 0)       # all candidates, sorted by residual
    ins = (np.abs(lam - c) <= g["r"]) & (res < 1e-9)
    comp = np.array([complex(a, b) for a, b in golden[comp_key]["companion_inside"]])
    assert ins.sum() == len(comp)
    assert max(np.abs(comp - x).min() for x in lam[ins]) < 1e-9 and max(np.abs(lam[ins] - x).min() for x in comp) < 1e-9
    glam = np.array([complex(a, b) for a, b in g["lam_inside"]])
    assert max(np.abs(glam - x).min() for x in lam[ins]) < 1e-10
    assert hist[-1][0] == g["history"][-1][0]
    assert np.allclose(np.linalg.norm(X, axis=0), 1.0)


def test_hadeler_split_form_fixture(fo, wl, golden):
    """non-polynomial NEP written out in test/hadeler.jl:35-52: the oracle's nlfeast!/nlfeast_moments! reproduce the
    committed eigenvalues, and each of them makes T(lam) singular (smallest singular value ~ 0)."""
    w = wl.hadeler_split()
    T = lambda z: sum(M * f(z) for M, f in zip(w["mats"], w["funcs"]))      # noqa: E731
    lam, X, res = fo.nlfeast(T, wl.rand_x0(200, w["m0"], w["seed"]), w["nodes"], w["iter"], c=w["c"], r=w["r"], eps=w["eps"],
                             spurious=w["spurious"])
    ins = np.abs(lam - w["c"]) <= w["r"]
    g = c2n(golden["hadeler_nlfeast"]["lam_inside"])
    assert ins.sum() == len(g) == 12 and np.abs(np.sort_complex(lam[ins]) - g).max() < 1e-9
    assert np.abs(c2n(golden["hadeler_moments"]["lam_inside"]) - g).max() < 1e-9
    for z in g[:4]:
        Tz = T(z)
        assert np.linalg.svd(Tz, compute_uv=False)[-1] < 1e-9 * np.linalg.norm(Tz)


def test_mixed_prec_restatement_floor(fo, wl):
    """feast!(...; mixed_prec=true) restated literally (src/feast.jl:19-25,68): ComplexF32 factors and a ComplexF32
    `temp` put a floor of a few 1e-7 under the residual, the fp64 run goes on to 1e-13."""
    w = wl.c1_dense_hermitian(256, 16)
    con = fo.circular_contour_trapezoidal(w["c"], w["r"], w["nodes"])
    h64, h32 = [], []
    fo.gen_feast(w["X0"].copy(), w["A"], None, con, iter=10, eps=1e-12, store=True, history=h64)
    fo.gen_feast(w["X0"].copy(), w["A"], None, con, iter=10, eps=1e-12, store=True, history=h32, mixed_prec=True)
    assert h64[-1][2] < 1e-12 and 1e-8 < h32[-1][2] < 1e-5
    assert abs(np.log10(h32[1][2] / h64[1][2])) < 0.5


def test_dual_gen_feast_restatement(fo, wl, golden):
    """dual_gen_feast! (src/feast.jl:165-257, repaired as documented in the oracle): right and left eigenvectors of a
    non-Hermitian pencil, checked against scipy's generalized eigenvalues and their defining equations."""
    import scipy.linalg as sla
    g = golden["dual_c2_n256"]
    w = wl.c2_dense_generalized(256, 32, n_inside=12, radius=g["r"])
    con = fo.circular_contour_trapezoidal(w["c"], w["r"], 16)
    lam, xr, xl, res = fo.dual_gen_feast(wl.rand_x0(256, 32, 8001), wl.rand_x0(256, 32, 8002), w["A"], w["B"], con, iter=20, eps=1e-11, store=True)
    ev = sla.eigvals(w["A"], w["B"])
    ev = ev[np.abs(ev - w["c"]) <= w["r"]]
    assert len(lam) == len(ev) == 12 and max(np.abs(ev - x).min() for x in lam) < 1e-10
    assert np.abs(np.sort_complex(lam) - c2n(g["lam"])).max() < 1e-10
    A, B = w["A"], w["B"]
    for i in range(len(lam)):
        assert np.linalg.norm(A @ xr[:, i] - lam[i] * (B @ xr[:, i])) < 1e-10
        assert np.linalg.norm(xl[:, i].conj() @ A - lam[i] * (xl[:, i].conj() @ B)) < 1e-10
    G = xl.conj().T @ B @ xr                                            # bi-orthogonality of left and right vectors
    assert np.abs(G - np.diag(np.diag(G))).max() < 1e-9


def test_golden_c3_small_reproduces(fo, wl, golden):
    g = golden["c3_g8_s19_m16"]
    w = wl.c3_sparse_generalized(8, 19, 16)
    hist = []
    lam, X, res = fo.gen_feast(w["X0"].copy(), w["A"], w["B"], fo.circular_contour_trapezoidal(w["c"], w["r"], 16),
                               iter=w["iter"], eps=w["eps"], store=True, history=hist)
    assert len(lam) == len(g["lam"]) == w["n_inside"]
    assert np.abs(np.sort_complex(lam) - np.sort_complex(c2n(g["lam"]))).max() < 1e-10
    assert len(hist) == len(g["history"])


def test_store_false_equals_store_true(fo, wl):
    w = wl.c3_sparse_generalized(8, 19, 16)
    con = fo.circular_contour_trapezoidal(w["c"], w["r"], 16)
    a = fo.gen_feast(w["X0"].copy(), w["A"], w["B"], con, iter=3, eps=0.0, store=True)
    b = fo.gen_feast(w["X0"].copy(), w["A"], w["B"], con, iter=3, eps=0.0, store=False)
    assert np.abs(np.sort_complex(a[0]) - np.sort_complex(b[0])).max() < 1e-10


def test_contour_estimate_counts_eigenvalues(fo):
    # src/stochastic.jl: with enough samples the estimate approaches the number of eigenvalues inside
    A = np.diag(np.arange(1, 41)).astype(np.complex128)
    est = fo.contour_estimate_eig(A, fo.circular_contour_trapezoidal(5.5, 5.0, 32), samples=40, rng=np.random.default_rng(1))
    assert abs(est - 10) < 2.5


def test_dimension_errors(fo, wl):
    import pytest
    with pytest.raises(ValueError, match="must be square"):
        fo.feast(wl.rand_x0(5, 2, 1), np.zeros((5, 4)), fo.circular_contour_trapezoidal(0, 1, 8))
    with pytest.raises(ValueError, match="must match A"):
        fo.feast(wl.rand_x0(6, 2, 1), np.eye(5), fo.circular_contour_trapezoidal(0, 1, 8))
