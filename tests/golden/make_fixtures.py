"""Generate tests/golden/*.json from the CPU oracle (oracle/feast_oracle.py).

Run in the build container:  python tests/golden/make_fixtures.py [--full]

The reference (Julia) cannot run here or on the GPU box, so these vectors come
from the numpy/scipy restatement of its algorithm, which tests/test_oracle.py
pins against the reference's own known answers (test/runtests.jl) and against
companion-linearisation spectra (src/companion.jl) of the reference's data
files.  Inputs are regenerated from seeds by feastsolver.jl_amd/workloads.py,
so only outputs and contour parameters are stored.
``--full`` also computes the contour for the 46^3 bench problem (eigsh with a
SuperLU shift-invert, about two minutes).
"""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "feastsolver.jl_amd"))
import feast_oracle as fo  # noqa: E402
import workloads as wl  # noqa: E402


def cplx(a):
    a = np.asarray(a)
    return [[float(x.real), float(x.imag)] for x in a.ravel()]


def c3_contour(g, stencil, n_inside=32, k=44):
    A = wl.laplacian_3d(g, stencil)
    rng = np.random.default_rng(3001)
    B = sp.diags(rng.uniform(0.5, 1.5, g ** 3), format="csc")
    ev = np.sort(spla.eigsh(A, k=k, M=B, sigma=0.0, which="LM", return_eigenvectors=False))
    # pick the count closest to n_inside that has the largest relative gap
    best = max(range(n_inside - 4, n_inside + 5), key=lambda j: (ev[j] - ev[j - 1]) / ev[j])
    hi = 0.5 * (ev[best - 1] + ev[best])
    lo = ev[0] - (ev[1] - ev[0])
    return dict(c=0.5 * (lo + hi), r=0.5 * (hi - lo), n_inside=int(best),
                eigs=[float(x) for x in ev[:best]], n=g ** 3, nnz=int(A.nnz))


def run_linear(w, store=True):
    hist = []
    X = w["X0"].copy()
    con = fo.circular_contour_trapezoidal(w["c"], w["r"], w["nodes"])
    lam, Xo, res = fo.gen_feast(X, w["A"], w["B"], con, iter=w["iter"], eps=w["eps"], store=store, history=hist)
    order = np.lexsort((lam.imag, lam.real))
    return dict(lam=cplx(lam[order]), res=[float(x) for x in res[order]], history=hist,
                c=[w["c"].real, w["c"].imag], r=w["r"], nodes=w["nodes"], m0=w["m0"],
                iter=w["iter"], eps=w["eps"])


def moments_fixtures():
    """nlfeast_moments! (src/nlfeast.jl:173-318) on the reference's own data: the configurations of
    test/paper.jl:147 (deficient quadratic), test/butterfly.jl:73 and test/polynomial.jl:20 (system5, with
    m0 = 60 instead of 80 to keep the fixture small).  Stored: inputs by seed, the per-iteration history and the
    converged candidates inside the contour."""
    out = {}
    cases = {
        "moments_deficient_quadratic": dict(coeffs=wl.deficient_quadratic_coeffs(), n=15, m0=4, seed=6001, nodes=16, iter=10,
                                            c=0j, r=0.25, moments=2, eps=1e-13),
        "moments_butterfly": dict(coeffs=wl.c4_butterfly()["coeffs"], n=64, m0=15, seed=6002, nodes=16, iter=40,
                                  c=1 + 1j, r=0.5, moments=2, eps=1e-13),
        "moments_system5": dict(coeffs=wl.system5_coeffs(sparse=True), n=1000, m0=60, seed=6003, nodes=32, iter=10,
                                c=-1.55 + 0j, r=0.05, moments=2, eps=1e-13),
    }
    for name, cs in cases.items():
        hist = []
        X = wl.rand_x0(cs["n"], cs["m0"], cs["seed"])
        lam, Y, res = fo.nlfeast_moments(fo.poly_T(cs["coeffs"]), X, cs["nodes"], cs["iter"], c=cs["c"], r=cs["r"], eps=cs["eps"],
                                         moments=cs["moments"], history=hist)
        ins = (np.abs(lam - cs["c"]) <= cs["r"]) & (res < 1e-9)
        out[name] = dict(n=cs["n"], m0=cs["m0"], seed=cs["seed"], nodes=cs["nodes"], iter=cs["iter"], c=[cs["c"].real, cs["c"].imag],
                         r=cs["r"], moments=cs["moments"], eps=cs["eps"], history=hist,
                         lam_inside=cplx(np.sort_complex(lam[ins])), res_inside=[float(x) for x in res[ins]])
    # non-polynomial NEP in split form (test/hadeler.jl:27-59), plain nlfeast! and the moments variant
    w = wl.hadeler_split()
    T = lambda z: sum(M * f(z) for M, f in zip(w["mats"], w["funcs"]))      # noqa: E731
    hist = []
    lam, _, res = fo.nlfeast(T, wl.rand_x0(200, w["m0"], w["seed"]), w["nodes"], w["iter"], c=w["c"], r=w["r"], eps=w["eps"],
                             spurious=w["spurious"], history=hist)
    ins = np.abs(lam - w["c"]) <= w["r"]
    out["hadeler_nlfeast"] = dict(history=hist, lam_inside=cplx(np.sort_complex(lam[ins])), res_inside=[float(x) for x in res[ins]])
    hist = []
    lam, _, res = fo.nlfeast_moments(T, wl.rand_x0(200, w["m0"], w["seed"]), w["nodes"], w["iter"], c=w["c"], r=w["r"], eps=w["eps"],
                                     moments=2, spurious=w["spurious"], history=hist)
    ins = (np.abs(lam - w["c"]) <= w["r"]) & (res < 1e-9)
    out["hadeler_moments"] = dict(history=hist, lam_inside=cplx(np.sort_complex(lam[ins])), res_inside=[float(x) for x in res[ins]])
    # dual_gen_feast! (src/feast.jl:165-257) on a C2-type dense non-Hermitian pencil: 12 eigenvalues inside
    w = wl.c2_dense_generalized(256, 32, n_inside=12)
    hist = []
    lam, _, _, res = fo.dual_gen_feast(wl.rand_x0(256, 32, 8001), wl.rand_x0(256, 32, 8002), w["A"], w["B"],
                                       fo.circular_contour_trapezoidal(w["c"], w["r"], 16), iter=20, eps=1e-11, store=True, history=hist)
    out["dual_c2_n256"] = dict(r=w["r"], history=hist, lam=cplx(np.sort_complex(lam)), res=[float(x) for x in res])
    return out


def main():
    if "--moments-only" in sys.argv:       # add / refresh the nlfeast_moments! cases without touching the others
        gp = os.path.join(HERE, "golden.json")
        out = json.load(open(gp))
        out.update(moments_fixtures())
        json.dump(out, open(gp, "w"), indent=1)
        print({k: len(v["lam_inside"]) for k, v in out.items() if k.startswith("moments_") or k.startswith("hadeler_")}, len(out["dual_c2_n256"]["lam"]))
        return
    full = "--full" in sys.argv
    out = {}

    # ---- contours for the C3 family -------------------------------------
    cpath = os.path.join(HERE, "c3_contours.json")
    contours = json.load(open(cpath)) if os.path.exists(cpath) else {}
    for g, s, ni in [(8, 19, 12), (12, 19, 32), (12, 7, 32), (16, 19, 32), (24, 19, 32)]:
        contours[f"g{g}_s{s}"] = c3_contour(g, s, ni)
    if full:
        for g, s in [(46, 19), (46, 7), (32, 19)]:
            contours[f"g{g}_s{s}"] = c3_contour(g, s, 32)
    json.dump(contours, open(cpath, "w"), indent=1)

    # ---- runtests.jl known answers with seeded X0 --------------------------
    A = np.diag(np.arange(1, 26)).astype(float)
    hist = []
    lam, _, res = fo.feast(wl.rand_x0(25, 5, 11), A, fo.circular_contour_trapezoidal(1.5, 2.0, 8), iter=10, history=hist)
    out["runtests_dense_standard"] = dict(lam=cplx(np.sort_complex(lam)), res=[float(x) for x in res], history=hist)
    A1 = sp.diags([-np.ones(99), 2 * np.ones(100), -np.ones(99)], [-1, 0, 1], format="csc")
    ana = 4 * np.sin(np.arange(1, 11) * np.pi / 202) ** 2
    out["runtests_laplace1d_analytic"] = [float(x) for x in ana]

    # ---- BASELINE configs at oracle-friendly sizes -------------------------
    out["c1_n512"] = run_linear(wl.c1_dense_hermitian(512, 16))
    w2 = wl.c2_dense_generalized(512, 64)
    out["c2_n512"] = run_linear(w2)
    out["c2_n512"]["radius"] = w2["r"]
    out["c3_g12_s19"] = run_linear(wl.c3_sparse_generalized(12, 19, 64, fixture=contours["g12_s19"]))
    out["c3_g8_s19_m16"] = run_linear(wl.c3_sparse_generalized(8, 19, 16, fixture=contours["g8_s19"]))

    # ---- NEP: companion spectra + nlfeast runs -----------------------------
    w4 = wl.c4_butterfly()
    cl, _, _ = fo.companion(w4["coeffs"])
    ins = np.abs(cl - w4["c"]) <= w4["r"]
    hist = []
    lam, _, res = fo.nlfeast(fo.poly_T(w4["coeffs"]), w4["X0"].copy(), 16, 10, c=w4["c"], r=w4["r"],
                             eps=w4["eps"], history=hist)
    li = np.abs(lam - w4["c"]) <= w4["r"]
    out["c4_butterfly"] = dict(companion_inside=cplx(np.sort_complex(cl[ins])), lam_inside=cplx(np.sort_complex(lam[li])),
                               res_inside=[float(x) for x in res[li]], history=hist)

    dq = wl.deficient_quadratic_coeffs()
    cl, _, _ = fo.companion(dq)
    fin = np.isfinite(cl)
    ins = fin & (np.abs(np.where(fin, cl, 1e9)) <= 0.25)
    out["deficient_quadratic"] = dict(companion_inside=cplx(np.sort_complex(cl[ins])))
    C1, C2 = fo.companion_pencil(dq)
    hist = []
    lam, _, res = fo.gen_feast(wl.rand_x0(30, 30, 5002), C1, C2, fo.circular_contour_trapezoidal(0j, 0.25, 32),
                               iter=20, eps=1e-10, history=hist)
    out["c5_literal_m30"] = dict(lam=cplx(np.sort_complex(lam)), res=[float(x) for x in res], history=hist)

    s5 = wl.system5_coeffs(sparse=False)
    cl, _, _ = fo.companion(s5)
    ins = np.abs(cl + 1.55) <= 0.05
    out["system5"] = dict(companion_inside=cplx(np.sort_complex(cl[ins])))

    # C5 (ii): system5 companion pencil 2000 x 2000, 32 nodes, m0 = 128
    out["c5_system5_companion"] = run_linear(wl.c5_system5_companion(128))

    out.update(moments_fixtures())
    json.dump(out, open(os.path.join(HERE, "golden.json"), "w"), indent=1)
    print({k: (len(v["lam"]) if isinstance(v, dict) and "lam" in v else "-") for k, v in out.items()})


if __name__ == "__main__":
    main()
