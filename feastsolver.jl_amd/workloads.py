"""Synthetic inputs for the BASELINE.json configs.

Pure numpy/scipy data generation: no solver code lives here.  Every generator is
seeded so the HIP path, the oracle and the fixtures see the same bytes.
"""
from __future__ import annotations

import json
import os

import numpy as np
import scipy.sparse as sp

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def rand_x0(n, m0, seed):
    """rand(ComplexF64, n, m0) as in test/runtests.jl:18."""
    rng = np.random.default_rng(seed)
    return np.asfortranarray(rng.random((n, m0)) + 1j * rng.random((n, m0)))


def complex_normal(n, m, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((n, m)) + 1j * rng.standard_normal((n, m))) / np.sqrt(2.0)


def c1_dense_hermitian(n=512, m0=16, seed=1551):
    """C1: dense Hermitian, standard feast!, circular contour 8 nodes."""
    G = complex_normal(n, n, seed)
    A = (G + G.conj().T) / 2
    ev = np.linalg.eigvalsh(A)
    a = np.sort(np.abs(ev))
    r = 0.5 * (a[7] + a[8])
    return dict(A=A, B=None, c=0j, r=float(r), nodes=8, m0=m0, X0=rand_x0(n, m0, seed + 1),
                iter=20, eps=1e-12, n_inside=8)


def c2_dense_generalized(n=4096, m0=64, seeds=(2001, 2002, 2003), n_inside=None, radius=None):
    """C2: dense generalized complex non-Hermitian, 16 nodes.  ``radius`` may
    be supplied (from the fixture) to skip the O(n^3) host eigensolve."""
    A = complex_normal(n, n, seeds[0]) / np.sqrt(n)
    B = np.eye(n) + 0.1 * complex_normal(n, n, seeds[1]) / np.sqrt(n)
    if n_inside is None:
        n_inside = m0 // 2
    if radius is None:
        import scipy.linalg as sla
        ev = sla.eigvals(A, B)
        a = np.sort(np.abs(ev))
        radius = 0.5 * (a[n_inside - 1] + a[n_inside])
    return dict(A=A, B=B, c=0j, r=float(radius), nodes=16, m0=m0, X0=rand_x0(n, m0, seeds[2]),
                iter=20, eps=1e-10, n_inside=n_inside)


def laplacian_3d(g, stencil=19):
    """3-D Dirichlet Laplacian on a g x g x g grid (n = g^3), CSC float64.
    stencil 7: (6, -1 faces).  19: (4, -1/3 faces, -1/6 edges).
    27: (64/15 c, -7/15 faces, -1/10 edges, -1/30 corners) -- all zero-row-sum
    in the interior and symmetric positive definite with Dirichlet boundary."""
    if stencil == 7:
        wc, wf, we, wv = 6.0, -1.0, 0.0, 0.0
    elif stencil == 19:
        wc, wf, we, wv = 4.0, -1.0 / 3.0, -1.0 / 6.0, 0.0
    elif stencil == 27:
        wc, wf, we, wv = 64.0 / 15.0, -7.0 / 15.0, -1.0 / 10.0, -1.0 / 30.0
    else:
        raise ValueError("stencil must be 7, 19 or 27")
    idx = np.arange(g ** 3).reshape(g, g, g)
    rows, cols, vals = [idx.ravel()], [idx.ravel()], [np.full(g ** 3, wc)]
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                nz = abs(dx) + abs(dy) + abs(dz)
                w = (0.0, wf, we, wv)[nz]
                if nz == 0 or w == 0.0:
                    continue
                sx = slice(max(0, -dx), g - max(0, dx))
                sy = slice(max(0, -dy), g - max(0, dy))
                sz = slice(max(0, -dz), g - max(0, dz))
                tx = slice(max(0, dx), g - max(0, -dx))
                ty = slice(max(0, dy), g - max(0, -dy))
                tz = slice(max(0, dz), g - max(0, -dz))
                r_ = idx[sx, sy, sz].ravel()
                c_ = idx[tx, ty, tz].ravel()
                rows.append(r_)
                cols.append(c_)
                vals.append(np.full(r_.shape[0], w))
    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(g ** 3, g ** 3)).tocsc()
    A.sort_indices()
    return A


def c3_sparse_generalized(g=46, stencil=19, m0=64, seeds=(3001, 3002), fixture=None):
    """C3: A = 3-D Laplacian, B = diag(U(0.5,1.5)), 16 nodes, m0 = 64.
    The contour (c, r) comes from tests/golden/c3_contours.json (computed once
    by tests/golden/make_fixtures.py with eigsh) so no eigensolve runs here."""
    A = laplacian_3d(g, stencil)
    n = g ** 3
    rng = np.random.default_rng(seeds[0])
    B = sp.diags(rng.uniform(0.5, 1.5, n), format="csc")
    if fixture is None:
        with open(os.path.join(GOLDEN_DIR, "c3_contours.json")) as f:
            table = json.load(f)
        key = f"g{g}_s{stencil}"
        if key in table:
            fixture = table[key]
        else:   # no eigensolve available for this size: scale the 46^3 contour with h^2 (count is then approximate)
            ref = table[f"g46_s{stencil}"]
            sc = ((46 + 1.0) / (g + 1.0)) ** 2
            fixture = dict(c=ref["c"] * sc, r=ref["r"] * sc, n_inside=-1)
    return dict(A=A, B=B, c=complex(fixture["c"], 0.0), r=float(fixture["r"]), nodes=16, m0=m0,
                X0=rand_x0(n, m0, seeds[1]), iter=20, eps=1e-10, n_inside=int(fixture["n_inside"]))


def hermitian_phases(A, seed=3, amplitude=0.3):
    """A real symmetric sparse matrix with random phases e^{i theta} on the couplings above the diagonal and the conjugate
    ones below (a gauge field on the same graph): complex Hermitian, same pattern, same diagonal."""
    C = sp.coo_matrix(A).astype(np.complex128)
    rng = np.random.default_rng(seed)
    theta = rng.uniform(-amplitude, amplitude, C.nnz)
    up = C.row < C.col
    data = C.data.copy()
    data[up] *= np.exp(1j * theta[up])
    Au = sp.coo_matrix((np.where(up, data, 0), (C.row, C.col)), shape=C.shape).tocsc()
    Ah = (Au + Au.conj().T + sp.diags(A.diagonal())).tocsc()
    Ah.sort_indices()
    return Ah


def c3_hermitian(g=46, stencil=19, m0=64, seed=3):
    """The C3 pencil with a complex Hermitian A (`hermitian_phases` of the grid Laplacian), B = the same diagonal, a contour around
    its ~32 lowest eigenvalues (from the fixture; the real pencil's contour without one): the home ground of FEAST that gets neither the `U = D L^T` elimination nor the conjugate fold.  The expected
    eigenvalues come from tests/golden/c3_hermitian_g<g>.json (make_hermitian_fixture.py: shift-invert Lanczos on the host)."""
    w = c3_sparse_generalized(g, stencil, m0)
    w["A"] = hermitian_phases(w["A"], seed)
    path = os.path.join(GOLDEN_DIR, f"c3_hermitian_g{g}.json")
    if os.path.exists(path):
        with open(path) as f:
            fx = json.load(f)
        w["c"], w["r"] = complex(fx["c"], 0.0), float(fx["r"])        # chosen on this pencil's own spectrum (the gauge field lifts it)
        w["eigs"] = np.array(fx["eigs_inside"])
        w["n_inside"] = len(w["eigs"])
        w["eigs_near"] = np.array(fx["eigs_lowest"])
    else:
        w["n_inside"] = -1
    return w


def load_mtx(name):
    import scipy.io
    return scipy.io.mmread(os.path.join(GOLDEN_DIR, "data", name))


def c4_butterfly(m0=32, seed=4001):
    """C4: quartic butterfly NEP, data/butterflyM0..M4.mtx (test/butterfly.jl:67-72)."""
    coeffs = [np.asarray(load_mtx(f"butterflyM{j}.mtx").todense(), dtype=np.complex128) for j in range(5)]
    return dict(coeffs=coeffs, c=1 + 1j, r=0.5, nodes=16, m0=m0, X0=rand_x0(64, m0, seed),
                iter=10, eps=1e-11, spurious=1e-5, n_inside=13)


def deficient_quadratic_coeffs():
    """T(z) = (z+0.2)(z-0.1) A1 + A0 (test/paper.jl:38-40) as polynomial
    coefficients [A0 - 0.02 A1, 0.1 A1, A1]."""
    A0 = np.asarray(load_mtx("quadraticM0.mtx").todense(), dtype=np.complex128)
    A1 = np.asarray(load_mtx("quadraticM1.mtx").todense(), dtype=np.complex128)
    return [A0 - 0.02 * A1, 0.1 * A1, A1]


def system5_coeffs(sparse=True):
    """data/system5A0..A2.mtx quadratic (test/polynomial.jl:5-11)."""
    Ms = [load_mtx(f"system5A{j}.mtx") for j in range(3)]
    if sparse:
        return [sp.csc_matrix(M, dtype=np.complex128) for M in Ms]
    return [np.asarray(M.todense(), dtype=np.complex128) for M in Ms]


def c5_system5_companion(m0=128, seed=5001):
    """C5 (ii): system5 quadratic companion-linearised to a 2000 x 2000 dense
    pencil (src/companion.jl:4-14), gen_feast!, 32 nodes."""
    A = system5_coeffs(sparse=False)
    N, L = 1000, 2
    C1 = np.zeros((N * L, N * L), dtype=np.complex128)
    C2 = np.zeros((N * L, N * L), dtype=np.complex128)
    C1[:N, :N] = A[0]
    for i in range(N, N * L):
        C1[i, i] = 1
        C2[i, i - N] = 1
    for i in range(L):
        C2[:N, N * i:N * (i + 1)] = -A[i + 1]
    return dict(A=C1, B=C2, c=-1.55 + 0j, r=0.05, nodes=32, m0=m0, X0=rand_x0(N * L, m0, seed),
                iter=20, eps=1e-10, n_inside=50)


def hadeler_split(n=200, b0=100.0):
    """Hadeler NEP T(z) = (e^z - 1) M1 + z^2 M2 - M0 as written out in test/hadeler.jl:35-52 (the nlevp_native_hadeler
    gallery problem the script solves): M0 = b0 I, M1[i,j] = (n+1-max(i,j)) i j, M2[i,j] = n d_ij + 1/(i+j).
    Returns (mats, funcs) of the split form sum_t f_t(z) M_t, and the contour of test/hadeler.jl:56-57."""
    i = np.arange(1, n + 1)[:, None]
    j = np.arange(1, n + 1)[None, :]
    M1 = ((n + 1 - np.maximum(i, j)) * i * j).astype(np.complex128)
    M2 = (n * (i == j) + 1.0 / (i + j)).astype(np.complex128)
    M0 = b0 * np.eye(n, dtype=np.complex128)
    funcs = [lambda z: np.exp(z) - 1.0, lambda z: z * z, lambda z: -1.0 + 0.0 * z]
    return dict(mats=[M1, M2, M0], funcs=funcs, c=-30.0 + 0j, r=10.0, nodes=16, m0=15, iter=20, eps=1e-13, spurious=1e-2,
                seed=7001)
