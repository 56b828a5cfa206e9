"""CPU oracle for the FEAST contour-integral hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product path (``feastsolver.jl_amd``) never does and fails loudly when
the HIP library is missing.

It is a numpy/scipy restatement of the reference algorithm (spacedome/
FEASTSolver.jl @ 2024-12-18).  Each function cites the reference file:line it
follows.  The reference is Julia and there is no Julia runtime in the build
container or on the GPU box, so the reference itself can never be executed:
the oracle is pinned by the reference's own known-answer tests
(``test/runtests.jl``, replayed in ``tests/test_oracle.py``) and by companion-
linearisation spectra of the reference's ``data/*.mtx`` files.  ``nlfeast!``
has no assertion anywhere in the reference, so for that entry point parity is
pinned only through the companion spectrum ("Oracle").

Third-party arithmetic the reference reaches and that is NOT under
/root/reference (Manifest.toml): Julia 1.9.1 LinearAlgebra -> OpenBLAS 0.3.21
(zgetrf/zgetrs/zgeqrt/zgemm/zgeevx/zggev/zgesdd), SparseArrays -> SuiteSparse
5.10.1 UMFPACK.  Here the same LAPACK routines are reached through scipy's
bundled OpenBLAS and sparse LU through SuperLU (``splu``); pivot orders differ
from UMFPACK so agreement is to rounding, not bitwise.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
import scipy.sparse.linalg as spla


# --------------------------------------------------------------------------
# contours                                                   src/contour.jl
# --------------------------------------------------------------------------
class CircularContour:
    """src/contour.jl:3-8"""

    def __init__(self, c, r, nodes, weights):
        self.c, self.r = complex(c), float(r)
        self.nodes = np.asarray(nodes, dtype=np.complex128)
        self.weights = np.asarray(weights, dtype=np.complex128)


class RectangularContour:
    """src/contour.jl:10-16 (constructor rejects unordered corners)"""

    def __init__(self, bl, tr, nodes, weights):
        bl, tr = complex(bl), complex(tr)
        if not (bl.real < tr.real and bl.imag < tr.imag):
            raise ValueError("Invalid corners")
        self.bottom_left, self.top_right = bl, tr
        self.nodes = np.asarray(nodes, dtype=np.complex128)
        self.weights = np.asarray(weights, dtype=np.complex128)


def circular_contour_trapezoidal(c, r, N=16):
    """src/contour.jl:26-31  theta = LinRange(pi/N, 2pi-pi/N, N)"""
    theta = np.linspace(np.pi / N, 2 * np.pi - np.pi / N, N)
    e = r * np.exp(1j * theta)
    return CircularContour(c, r, e + c, e / N)


def circular_contour_gauss(c, r, N=16):
    """src/contour.jl:33-45"""
    if N % 2 != 0:
        raise ValueError("Number of nodes must be multiple of 2")
    n = N // 2
    x, w = np.polynomial.legendre.leggauss(n)
    x = (np.pi / 2.0) * (x + 1.0)
    nodes = np.concatenate([r * np.exp(1j * x) + c, r * np.exp(1j * (x + np.pi)) + c])
    weights = np.concatenate([r * np.exp(1j * x) * w / 4.0, r * np.exp(1j * (x + np.pi)) * w / 4.0])
    return CircularContour(c, r, nodes, weights)


def rectangular_contour_gauss(bl, tr, N=16):
    """src/contour.jl:47-65 (nodes clockwise: top, right, bottom, left)"""
    bl, tr = complex(bl), complex(tr)
    if N % 4 != 0:
        raise ValueError("Number of nodes must be multiple of 4")
    n = N // 4
    x, w = np.polynomial.legendre.leggauss(n)
    top_len, side_len = tr.real - bl.real, tr.imag - bl.imag
    nodes = np.zeros(N, dtype=np.complex128)
    weights = np.zeros(N, dtype=np.complex128)
    nodes[0:n] = (x + 1) * (top_len / 2) + (tr.imag * 1j + bl.real)
    nodes[n:2 * n] = (x + 1) * (1j * side_len / 2) + (bl.imag * 1j + tr.real)
    nodes[2 * n:3 * n] = (x[::-1] + 1) * (top_len / 2) + (bl.imag * 1j + bl.real)
    nodes[3 * n:4 * n] = (x[::-1] + 1) * (1j * side_len / 2) + (bl.imag * 1j + bl.real)
    weights[0:n] = w * top_len
    weights[n:2 * n] = -1j * w * side_len
    weights[2 * n:3 * n] = -w * top_len
    weights[3 * n:4 * n] = 1j * w * side_len
    return RectangularContour(bl, tr, nodes, weights / (-4.0 * np.pi * 1j))


def rectangular_contour_trapezoidal(bl, tr, N=16):
    """src/contour.jl:67-86"""
    bl, tr = complex(bl), complex(tr)
    if N % 4 != 0:
        raise ValueError("Number of nodes must be multiple of 4")
    n = N // 4
    nodes = np.zeros(N, dtype=np.complex128)
    weights = np.zeros(N, dtype=np.complex128)
    nodes[0:n] = np.linspace(bl.real, tr.real, n + 1)[:n] + tr.imag * 1j
    nodes[n:2 * n] = np.linspace(tr.imag, bl.imag, n + 1)[:n] * 1j + tr.real
    nodes[2 * n:3 * n] = np.linspace(tr.real, bl.real, n + 1)[:n] + bl.imag * 1j
    nodes[3 * n:4 * n] = np.linspace(bl.imag, tr.imag, n + 1)[:n] * 1j + bl.real
    top_len, side_len = tr.real - bl.real, tr.imag - bl.imag
    weights[0] = 1j * side_len / (2 * n) + top_len / (2 * n)
    weights[1:n] = top_len / n
    weights[n] = top_len / (2 * n) - 1j * side_len / (2 * n)
    weights[n + 1:2 * n] = -1j * side_len / n
    weights[2 * n] = -1j * side_len / (2 * n) - top_len / (2 * n)
    weights[2 * n + 1:3 * n] = -top_len / n
    weights[3 * n] = -top_len / (2 * n) + 1j * side_len / (2 * n)
    weights[3 * n + 1:4 * n] = 1j * side_len / n
    return RectangularContour(bl, tr, nodes, weights / (-2.0 * np.pi * 1j))


def in_contour(lam, contour, r=None):
    """src/contour.jl:88-100: closed disc / open rectangle."""
    lam = np.asarray(lam)
    if r is not None:  # in_contour(lam, c, r)
        return np.abs(lam - contour) <= r
    if isinstance(contour, CircularContour):
        return np.abs(lam - contour.c) <= contour.r
    bl, tr = contour.bottom_left, contour.top_right
    return (bl.real < lam.real) & (lam.real < tr.real) & (bl.imag < lam.imag) & (lam.imag < tr.imag)


# --------------------------------------------------------------------------
# the factorizer / left_divider / finalize! seam            src/utils.jl:173-179
# --------------------------------------------------------------------------
class _DenseLU:
    """LU in the element type of C: complex128, or complex64 on the mixed_prec path (Julia's lu of a
    Matrix{ComplexF32}, src/feast.jl:25-36)."""

    def __init__(self, C):
        self.dtype = np.complex64 if np.asarray(C).dtype == np.complex64 else np.complex128
        self.lu, self.piv = sla.lu_factor(np.array(C, dtype=self.dtype), check_finite=False)

    def solve(self, R):
        return sla.lu_solve((self.lu, self.piv), np.asarray(R).astype(self.dtype), check_finite=False)


class _SparseLU:
    """SuperLU through scipy (the reference: UMFPACK through Julia's `lu`).  `permc_spec`: SuperLU's column ordering -- scipy's default
    COLAMD, or MMD_AT_PLUS_A, the closest SuperLU setting to the symmetric strategy UMFPACK picks for patterns like the grid
    Laplacians of the benchmark (2-3 x fewer factor entries there; bench.py times both and reports the faster)."""

    def __init__(self, C, permc_spec=None):
        self.dtype = np.complex64 if C.dtype == np.complex64 else np.complex128
        self.f = spla.splu(sp.csc_matrix(C, dtype=self.dtype), **({} if permc_spec is None else {"permc_spec": permc_spec}))
        self.nnz_lu = int(self.f.L.nnz + self.f.U.nnz)

    def solve(self, R):
        return self.f.solve(np.ascontiguousarray(R, dtype=self.dtype))


def lu(C):
    """default ``factorizer`` (Julia ``lu``: LAPACK zgetrf / UMFPACK)."""
    return _SparseLU(C) if sp.issparse(C) else _DenseLU(C)


def lu_mmd(C):
    """``factorizer`` with SuperLU's MMD(A^T + A) column ordering for sparse C (see _SparseLU)."""
    return _SparseLU(C, "MMD_AT_PLUS_A") if sp.issparse(C) else _DenseLU(C)


def ldiv(Y, F, R):
    """default ``left_divider`` (3-arg ``ldiv!``): Y .= F \\ R, R untouched."""
    Y[...] = F.solve(R)
    return Y


def linsolve(Y, C, X, factorizer=lu, left_divider=ldiv):
    """src/utils.jl:175-179"""
    F = factorizer(C)
    left_divider(Y, F, X)


def _matmul(A, X):
    return A @ X


def _eye_like(A, n):
    return sp.identity(n, dtype=np.complex128, format="csc") if sp.issparse(A) else np.eye(n, dtype=np.complex128)


def _sorted_eig(Aq, Bq=None):
    """Julia ``eigen!(A)`` / ``eigen!(A,B)``: LAPACK zgeevx / zggev, unit-norm
    vectors, eigenvalues sorted by (real, imag) (stdlib default ``sortby``)."""
    if Bq is None:
        w, v = sla.eig(Aq, check_finite=False)
    else:
        w, v = sla.eig(Aq, Bq, check_finite=False)
    v = v / np.linalg.norm(v, axis=0, keepdims=True)
    order = np.lexsort((w.imag, w.real))
    return w[order], v[:, order]


# --------------------------------------------------------------------------
# iteration helpers                                           src/utils.jl
# --------------------------------------------------------------------------
def update_R(X, R, lam, A, B=None):
    """src/utils.jl:111-116: normalise columns of X in place, R[:,i] =
    (A - lam_i B) x_i.  Restated as A X - (B X) diag(lam) (identical algebra;
    the reference forms an n x n matrix per column)."""
    X /= np.linalg.norm(X, axis=0, keepdims=True)
    AX = _matmul(A, X)
    BX = X if B is None else _matmul(B, X)
    R[...] = AX - BX * lam[None, :]


def update_R_nl(X, R, lam, T):
    """src/utils.jl:104-109"""
    X /= np.linalg.norm(X, axis=0, keepdims=True)
    for i in range(X.shape[1]):
        R[:, i] = T(lam[i]) @ X[:, i]


def residuals_linear(R):
    """src/utils.jl:166-171: absolute 2-norms."""
    return np.linalg.norm(R, axis=0)


def _fro(M):
    return spla.norm(M) if sp.issparse(M) else np.linalg.norm(M)


def residuals_nl(R, lam, T):
    """src/utils.jl:151-157: ||R_i||_2 / ||T(lam_i)||_F."""
    return np.array([np.linalg.norm(R[:, i]) / _fro(T(lam[i])) for i in range(lam.shape[0])])


def beyn_svd_step(Q0, Q1, X, lam):
    """src/utils.jl:69-77 (no rank truncation, 1 ./ S)."""
    U, S, Vh = sla.svd(Q0, full_matrices=False, check_finite=False)
    A = U.conj().T @ Q1
    Bm = A @ Vh.conj().T
    A = Bm * (1.0 / S)[None, :]
    w, v = _sorted_eig(A)
    X[...] = U @ v
    lam[...] = w


# --------------------------------------------------------------------------
# solvers                                        src/feast.jl, src/nlfeast.jl
# --------------------------------------------------------------------------
def gen_feast(X, A, B, contour, iter=10, eps=1e-12, store=False,
              factorizer=lu, left_divider=ldiv, history=None, mixed_prec=False):
    """``gen_feast!`` src/feast.jl:89-156; with ``B=None`` it is ``feast!``
    src/feast.jl:10-80 (the two loops are identical up to B).

    X is overwritten in place (the ``!`` convention).  Returns
    (lam[inside], X[:, inside], res[inside]).  ``history`` (a list) receives
    per-iteration (nit, count_inside, max_res_inside).  ``mixed_prec`` (feast! only in the
    reference, src/feast.jl:10,19-25): the shifted matrix, its factors and ``temp`` are ComplexF32,
    everything else ComplexF64.
    """
    X = np.asarray(X)
    N, m0 = X.shape
    if A.shape[0] != A.shape[1]:
        raise ValueError("Incorrect dimensions of A, must be square")
    if A.shape[0] != N:
        raise ValueError("Incorrect dimensions of X, must match A")
    generalized = B is not None
    if sp.issparse(A):
        A = sp.csc_matrix(A, dtype=np.complex128)
    else:
        A = np.asarray(A).astype(np.complex128)
    if generalized:
        B = sp.csc_matrix(B, dtype=np.complex128) if sp.issparse(B) else np.asarray(B).astype(np.complex128)
        if sp.issparse(A) != sp.issparse(B):  # mixed: densify the sparse one
            if sp.issparse(A):
                A = A.toarray()
            else:
                B = B.toarray()
    Bs = B if generalized else _eye_like(A, N)
    lam = np.zeros(m0, dtype=np.complex128)
    res = np.zeros(m0)
    ctype = np.complex64 if mixed_prec else np.complex128     # :19
    temp = np.zeros((N, m0), dtype=ctype)                     # :22
    R = np.zeros((N, m0), dtype=np.complex128)
    Q = X.astype(np.complex128).copy()
    nodes, weights = contour.nodes, contour.weights
    K = nodes.shape[0]

    def shifted(k):                                           # ZmA .= A - I*z  into a Ctype buffer  :25-29
        M = A - Bs * nodes[k]
        return M.astype(ctype) if mixed_prec else M

    facts = None
    if store:  # src/feast.jl:28-38 / :104-114
        facts = [factorizer(shifted(k)) for k in range(K)]

    for nit in range(iter + 1):
        Q, _ = np.linalg.qr(Q, mode="reduced")               # :41 / :117
        R[...] = _matmul(A, Q)                                # :42
        Aq = Q.conj().T @ R                                   # :43
        if generalized:
            R[...] = _matmul(B, Q)                            # :120
            Bq = Q.conj().T @ R                               # :121
            lam[...], Xq = _sorted_eig(Aq, Bq)                # :122
        else:
            lam[...], Xq = _sorted_eig(Aq)                    # :45
        X[...] = Q @ Xq                                       # :48
        update_R(X, R, lam, A, B if generalized else None)    # :49 / :126
        res[...] = residuals_linear(R)                        # :50
        inside = in_contour(lam, contour)
        if history is not None:
            history.append((nit, int(inside.sum()), float(res[inside].max()) if inside.any() else float("nan")))
        if inside.any() and res[inside].max() < eps:          # :53
            break
        if nit < iter:                                        # :57
            Q = np.zeros((N, m0), dtype=np.complex128)
            for k in range(K):
                resolvent = 1.0 / (nodes[k] - lam)            # :60
                if store:
                    left_divider(temp, facts[k], R)
                else:
                    linsolve(temp, shifted(k), R, factorizer, left_divider)
                temp[...] = X - temp                          # :68  (rounded to Ctype: temp is a Ctype array)
                temp *= (resolvent * weights[k])[None, :].astype(ctype)     # :69
                Q += temp                                     # :70
    inside = in_contour(lam, contour)
    return lam[inside].copy(), X[:, inside].copy(), res[inside].copy()


def dual_gen_feast(Xr, Xl, A, B, contour, iter=10, eps=1e-12, store=False, factorizer=lu, left_divider=ldiv, history=None):
    """``dual_gen_feast!`` src/feast.jl:165-257: two-sided FEAST, right (Xr) and left (Xl) eigenvectors.
    Restated with the three slips of the reference repaired ("fix rather than reproduce"):
    (i) :200-201 ``Diagonal(1.0/S.S)`` -> both bases scaled by S^-1/2, so that Ql' B Qr = I;
    (ii) :214 the left residual is A' y - conj(lam) B' y (the reference passes lam itself);
    (iii) :210-212 the left Ritz vectors come from eigen(Aq', Bq') in ITS eigenvalue order; here they are paired
    with the right ones by eigenvalue (conj).  Returns (lam[in], Xr[:, in], Xl[:, in], resr[in])."""
    Xr, Xl = np.asarray(Xr), np.asarray(Xl)
    N, m0 = Xl.shape
    if A.shape[0] != A.shape[1]:
        raise ValueError("Incorrect dimensions of A, must be square")
    if A.shape[0] != N:
        raise ValueError("Incorrect dimensions of X, must match A")
    A = sp.csc_matrix(A, dtype=np.complex128) if sp.issparse(A) else np.asarray(A).astype(np.complex128)
    B = sp.csc_matrix(B, dtype=np.complex128) if sp.issparse(B) else np.asarray(B).astype(np.complex128)
    Ah, Bh = A.conj().T, B.conj().T
    lam = np.zeros(m0, dtype=np.complex128)
    resr = np.zeros(m0)
    temp = np.zeros((N, m0), dtype=np.complex128)
    Rr = np.zeros((N, m0), dtype=np.complex128)
    Rl = np.zeros((N, m0), dtype=np.complex128)
    Qr, Ql = Xr.astype(np.complex128).copy(), Xl.astype(np.complex128).copy()
    nodes, weights = contour.nodes, contour.weights
    K = nodes.shape[0]
    rfacts = lfacts = None
    if store:                                                  # :178-197
        rfacts = [factorizer(A - B * nodes[k]) for k in range(K)]
        lfacts = [factorizer((A - B * nodes[k]).conj().T) for k in range(K)]
    for nit in range(iter + 1):
        U, S, Vh = np.linalg.svd(Ql.conj().T @ _matmul(B, Qr))          # :199
        Qr = (Qr @ Vh.conj().T) / np.sqrt(S)[None, :]                   # :200  (repaired)
        Ql = (Ql @ U) / np.sqrt(S)[None, :]                             # :201
        Aq = Ql.conj().T @ _matmul(A, Qr)                               # :202-203
        Bq = Ql.conj().T @ _matmul(B, Qr)                               # :204-205
        lam[...], Xq = _sorted_eig(Aq, Bq)                              # :206-208
        mu, Yq = _sorted_eig(Aq.conj().T, Bq.conj().T)                  # :210-211
        used = np.zeros(m0, dtype=bool)
        Ym = np.zeros_like(Yq)
        for i in range(m0):
            d = np.where(used, np.inf, np.abs(mu - np.conj(lam[i])))
            j = int(np.argmin(d))
            used[j] = True
            Ym[:, i] = Yq[:, j]
        Xr[...] = Qr @ Xq                                               # :209
        Xl[...] = Ql @ Ym                                               # :212
        update_R(Xr, Rr, lam, A, B)                                     # :213
        update_R(Xl, Rl, np.conj(lam), Ah, Bh)                          # :214  (repaired)
        resr[...] = residuals_linear(Rr)                                # :215
        inside = in_contour(lam, contour)
        if history is not None:
            history.append((nit, int(inside.sum()), float(resr[inside].max()) if inside.any() else float("nan")))
        if inside.any() and resr[inside].max() < eps:                   # :218
            break
        if nit < iter:                                                  # :222
            Qr = np.zeros((N, m0), dtype=np.complex128)
            Ql = np.zeros((N, m0), dtype=np.complex128)
            for k in range(K):
                resolvent = 1.0 / (nodes[k] - lam)                      # :226
                if store:
                    left_divider(temp, rfacts[k], Rr)
                else:
                    linsolve(temp, A - B * nodes[k], Rr, factorizer, left_divider)
                Qr += (Xr - temp) * (resolvent * weights[k])[None, :]   # :232-234
                resolvent = 1.0 / (np.conj(nodes[k]) - np.conj(lam))    # :236
                if store:
                    left_divider(temp, lfacts[k], Rl)
                else:
                    linsolve(temp, (A - B * nodes[k]).conj().T, Rl, factorizer, left_divider)
                Ql += (Xl - temp) * (resolvent * np.conj(weights[k]))[None, :]   # :242-244
    inside = in_contour(lam, contour)
    return lam[inside].copy(), Xr[:, inside].copy(), Xl[:, inside].copy(), resr[inside].copy()


def feast(X, A, contour, **kw):
    """``feast!`` src/feast.jl:10-80"""
    return gen_feast(X, A, None, contour, **kw)


def nlfeast(T, X, nodes, iter, c=0j, r=1.0, eps=10e-12, store=True, spurious=1e-5,
            factorizer=lu, left_divider=ldiv, history=None):
    """``nlfeast!`` src/nlfeast.jl:2-84.  Returns (lam, X, res) for ALL m0
    pairs (unfiltered), X column-normalised."""
    X = np.asarray(X)
    assert X.dtype == np.complex128
    N, m0 = X.shape
    lam = np.zeros(m0, dtype=np.complex128)
    res = np.zeros(m0)
    theta = np.linspace(np.pi / nodes, 2 * np.pi - np.pi / nodes, nodes)   # :8
    R = np.zeros((N, m0), dtype=np.complex128)
    q, _ = np.linalg.qr(X, mode="reduced")                                  # :12-13
    X[...] = q
    zs = r * np.exp(1j * theta) + c
    ws = r * np.exp(1j * theta) / nodes
    facts = [factorizer(T(zs[k])) for k in range(nodes)] if store else None
    tinv = np.zeros((N, m0), dtype=np.complex128)
    for nit in range(iter + 1):
        Q0 = np.zeros((N, m0), dtype=np.complex128)
        Q1 = np.zeros((N, m0), dtype=np.complex128)
        for k in range(nodes):
            F = facts[k] if store else factorizer(T(zs[k]))
            if nit == 0:                                                    # :39-45
                left_divider(tinv, F, X)
                tinv *= ws[k]
            else:                                                           # :46-55
                resolvent = (1.0 / (zs[k] - lam)) * ws[k]
                left_divider(tinv, F, R)
                tinv[...] = X - tinv
                tinv *= resolvent[None, :]
            Q0 += tinv                                                      # :56-59
            Q1 += tinv * zs[k]
        beyn_svd_step(Q0, Q1, X, lam)                                       # :64
        update_R_nl(X, R, lam, T)                                           # :66
        res[...] = residuals_nl(R, lam, T)                                  # :67
        ins = in_contour(lam, c, r)
        res_inside = res[ins]
        if history is not None:
            history.append((nit, int(ins.sum()), float(res_inside.max()) if ins.any() else float("nan")))
        if res_inside.size > 0 and res_inside.max() < eps:                  # :73-76
            break
        if nit > 1 and (res_inside < spurious).sum() > 0 and res_inside[res_inside < spurious].max() < eps:
            break                                                           # :77-79
    X /= np.linalg.norm(X, axis=0, keepdims=True)                           # :82
    return lam, X, res


def update_R_moments(Y, R, lam, res, T):
    """``update_R_moments!`` src/utils.jl:118-134: normalise, R[:,i] = T(lam_i) y_i, relative
    residuals, then everything sorted by residual (ascending, stable like Julia's sortperm)."""
    for i in range(Y.shape[1]):
        Y[:, i] /= np.linalg.norm(Y[:, i])
        R[:, i] = _matmul(T(lam[i]), Y[:, i])
    res[...] = residuals_nl(R, lam, T)
    p = np.argsort(res, kind="stable")
    res[...] = res[p]
    Y[...] = Y[:, p]
    lam[...] = lam[p]
    R[...] = R[:, p]


def nlfeast_moments(T, X, nodes, iter, c=0j, r=1.0, eps=10e-12, moments=2, store=True, spurious=1e-5,
                    factorizer=lu, left_divider=ldiv, history=None):
    """``nlfeast_moments!`` src/nlfeast.jl:173-318.  2*moments contour moments
    Q_j = sum_k w_k z_k^j T(z_k)^-1 (.) , block-Hankel matrices of moments x moments blocks,
    Beyn linearisation through the thin SVD of the first one.  Returns (lam, Y, res) for all
    moments*m0 candidates sorted by residual; X is overwritten with the m0 best, normalised."""
    X = np.asarray(X)
    assert X.dtype == np.complex128
    N, m0 = X.shape
    M = moments * m0
    theta = np.linspace(np.pi / nodes, 2 * np.pi - np.pi / nodes, nodes)    # :178
    zs = r * np.exp(1j * theta) + c
    ws = r * np.exp(1j * theta) / nodes
    lam = np.zeros(M, dtype=np.complex128)
    res = np.zeros(M)
    Y = np.zeros((N, M), dtype=np.complex128)
    R = np.zeros((N, M), dtype=np.complex128)
    facts = [factorizer(T(zs[k])) for k in range(nodes)] if store else None
    tinv = np.zeros((N, m0), dtype=np.complex128)

    def hankel_step(Q):                                                     # :213-224 == :270-281
        Q0 = np.zeros((moments * N, M), dtype=np.complex128)
        Q1 = np.zeros((moments * N, M), dtype=np.complex128)
        for i in range(moments):
            for j in range(moments):
                Q0[i * N:(i + 1) * N, j * m0:(j + 1) * m0] = Q[i + j]
                Q1[i * N:(i + 1) * N, j * m0:(j + 1) * m0] = Q[i + j + 1]
        U, S, Vh = np.linalg.svd(Q0, full_matrices=False)
        A = ((U.conj().T @ Q1) @ Vh.conj().T) * (1.0 / S)[None, :]
        vals, vecs = np.linalg.eig(A)
        Y[...] = U[:N, :] @ vecs
        lam[...] = vals
        update_R_moments(Y, R, lam, res, T)                                 # :229 / :286
        X[...] = Y[:, :m0]

    for nit in range(iter + 1):
        Q = np.zeros((2 * moments, N, m0), dtype=np.complex128)
        for k in range(nodes):
            F = facts[k] if store else factorizer(T(zs[k]))
            if nit == 0:                                                    # :193-212
                left_divider(tinv, F, X)
                temp = tinv * ws[k]
            else:                                                           # :246-268
                resolvent = (1.0 / (zs[k] - lam[:m0])) * ws[k]
                left_divider(tinv, F, R[:, :m0])
                temp = (X - tinv) * resolvent[None, :]
            for j in range(2 * moments):
                Q[j] += temp * zs[k] ** j
        hankel_step(Q)
        ins = in_contour(lam[:m0], c, r)
        res_inside = res[:m0][ins]
        if history is not None:
            history.append((nit, int(ins.sum()), float(res_inside.max()) if ins.any() else float("nan")))
        if nit >= 1:                                                        # the reference tests only inside its loop
            if res_inside.size > 0 and res_inside.max() < eps:              # :296-299
                break
            if nit > 1 and (res_inside < spurious).sum() > 0 and res_inside[res_inside < spurious].max() < eps:
                break                                                       # :300-302
    X /= np.linalg.norm(X, axis=0, keepdims=True)                           # :305-311 (res is already sorted)
    return lam, Y, res


def contour_estimate_eig(A, contour, B=None, X=None, samples=None, rng=None):
    """src/stochastic.jl:2-33: est = real(sum_k w_k tr(X' (z_k B - A)^-1 X)) / samples."""
    n = A.shape[0]
    if X is None:
        m = min(100, n) if samples is None else samples
        rng = np.random.default_rng() if rng is None else rng
        X = (rng.standard_normal((n, m)) + 1j * rng.standard_normal((n, m))) / np.sqrt(2.0)
    Bs = _eye_like(A, n) if B is None else B
    est = 0.0
    for z, w in zip(contour.nodes, contour.weights):
        Y = lu(Bs * z - A).solve(X)                         # :24-25
        est += np.trace(X.conj().T @ Y) * w / X.shape[1]    # :26-27
    return float(np.real(est))


def companion(coeffs):
    """src/companion.jl:1-28: dense companion linearisation of
    sum_j z^j coeffs[j]; returns (lam, X, res) for all N*L pairs."""
    A = [np.asarray(M.toarray() if sp.issparse(M) else M, dtype=np.complex128) for M in coeffs]
    N, L = A[0].shape[0], len(A) - 1
    C1 = np.zeros((N * L, N * L), dtype=np.complex128)
    C2 = np.zeros((N * L, N * L), dtype=np.complex128)
    C1[:N, :N] = A[0]
    for i in range(N, N * L):
        C1[i, i] = 1
        C2[i, i - N] = 1
    for i in range(L):
        C2[:N, N * i:N * (i + 1)] = -A[i + 1]
    w, v = sla.eig(C1, C2, check_finite=False)
    X = v[(L - 1) * N:L * N, :].copy()
    res = np.zeros(N * L)
    for i in range(N * L):
        X[:, i] /= np.linalg.norm(X[:, i])
        Rv = sum(A[j] @ X[:, i] * w[i] ** j for j in range(L + 1))
        res[i] = np.linalg.norm(Rv) / np.linalg.norm(sum(A[j] * w[i] ** j for j in range(L + 1)))
    return w, X, res


def companion_pencil(coeffs):
    """The (C1, C2) pencil of src/companion.jl:4-14 without solving it."""
    A = [np.asarray(M.toarray() if sp.issparse(M) else M, dtype=np.complex128) for M in coeffs]
    N, L = A[0].shape[0], len(A) - 1
    C1 = np.zeros((N * L, N * L), dtype=np.complex128)
    C2 = np.zeros((N * L, N * L), dtype=np.complex128)
    C1[:N, :N] = A[0]
    for i in range(N, N * L):
        C1[i, i] = 1
        C2[i, i - N] = 1
    for i in range(L):
        C2[:N, N * i:N * (i + 1)] = -A[i + 1]
    return C1, C2


def poly_T(coeffs):
    """Matrix polynomial closure z -> sum_j z^j M_j (test/butterfly.jl:60-62,
    test/paper.jl:50-52)."""
    def T(z):
        out = coeffs[0] * 1.0
        zz = 1.0
        for M in coeffs[1:]:
            zz = zz * z
            out = out + M * zz
        return out
    return T
