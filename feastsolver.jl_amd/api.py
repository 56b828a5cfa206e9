"""Host-side mirror of the reference's solver interface (src/feast.jl,
src/nlfeast.jl, src/utils.jl:173-179) on top of the C ABI.

    feast(X, A, contour)            <->  feast!(X, A, contour; ...)       src/feast.jl:10-80
    gen_feast(X, A, B, contour)     <->  gen_feast!(X, A, B, contour; ...) src/feast.jl:89-156
    nlfeast(T, X, nodes, iter)      <->  nlfeast!(T, X, nodes, iter; ...)  src/nlfeast.jl:2-84
    lu / ldiv / finalize            <->  factorizer / left_divider / finalize!

X is overwritten in place (the `!` convention); return values are the reference's
3-tuples.  Keyword names follow the reference (eps for the Julia epsilon).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

try:
    import scipy.sparse as sp
except Exception:  # pragma: no cover
    sp = None

from . import _lib
from .contour import CircularContour, circular_contour_trapezoidal

F64, C128, I64T, C64T = 0, 1, 2, 3
ROW_MAJOR = 0x100       # feasthip.h: FEASTHIP_ROW_MAJOR (dense matrices in C order)


def _issparse(M):
    return sp is not None and sp.issparse(M)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def _dense_arg(M):
    """-> (array kept alive, type tag).  The bytes are handed over in the order they are stored in (C-ordered arrays with the
    row-major flag: element type and storage order are converted on the device); only an unsupported dtype or a strided view
    costs a host copy."""
    M = np.asarray(M)
    if M.dtype == np.complex128:
        ty = C128
    elif M.dtype == np.float64:
        ty = F64
    elif M.dtype == np.int64:
        ty = I64T
    elif M.dtype == np.complex64:
        ty = C64T
    elif np.issubdtype(M.dtype, np.integer) or M.dtype == np.bool_:
        M, ty = M.astype(np.int64), I64T
    elif np.issubdtype(M.dtype, np.complexfloating):
        M, ty = M.astype(np.complex128), C128
    else:
        M, ty = M.astype(np.float64), F64
    if M.flags.f_contiguous:
        return M, ty
    return np.ascontiguousarray(M), ty | ROW_MAJOR


def _csc_arg(M):
    M = sp.csc_matrix(M)
    M.sort_indices()
    M.sum_duplicates()
    if np.issubdtype(M.dtype, np.complexfloating):
        vals, ty = np.ascontiguousarray(M.data, dtype=np.complex128), C128
    else:
        vals, ty = np.ascontiguousarray(M.data, dtype=np.float64), F64
    return (np.ascontiguousarray(M.indptr, dtype=np.int64), np.ascontiguousarray(M.indices, dtype=np.int64), vals), ty


class PolynomialNEP:
    """T(z) = sum_j z^j M_j.  Callable like the reference's closures
    (test/butterfly.jl:60-62, test/paper.jl:38-52) and understood by ``nlfeast``'s
    device fast path."""

    def __init__(self, coeffs):
        self.coeffs = list(coeffs)

    def __call__(self, z):
        out = self.coeffs[0] * 1.0
        zz = 1.0
        for M in self.coeffs[1:]:
            zz = zz * z
            out = out + M * zz
        return out


class SplitNEP:
    """T(z) = sum_t f_t(z) M_t (sum of products of matrices and scalar functions): the form of the reference's
    non-polynomial test problems (test/gun2.jl:30-37, test/hadeler.jl:27-59).  Callable like a reference closure;
    ``nlfeast`` / ``nlfeast_moments`` run it on the device (the M_t are resident, the f_t are evaluated on the host
    through one callback, include/feasthip.h: feasthip_set_coef_functions)."""

    def __init__(self, mats, funcs):
        if len(mats) != len(funcs):
            raise ValueError("one scalar function per matrix")
        self.coeffs = list(mats)
        self.funcs = list(funcs)

    def __call__(self, z):
        out = self.coeffs[0] * complex(self.funcs[0](z))
        for M, f in zip(self.coeffs[1:], self.funcs[1:]):
            out = out + M * complex(f(z))
        return out


class FeastHandle:
    """Owns one feasthip_handle_t (one GPU).  Used by feast/gen_feast/nlfeast and by the
    multi-GPU driver (distributed.py)."""

    def __init__(self, device=0):
        self.lib = _lib.load_library()
        h = C.c_void_p()
        code = self.lib.feasthip_create(C.byref(h), int(device))
        if code != 0:
            msg = self.lib.feasthip_last_error(None)
            raise _lib.FeastHipError(code, msg.decode() if msg else "")
        self.h = h
        self._keep = []
        self.n = 0
        self.m0 = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.feasthip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, code):
        return _lib.check(self.lib, self.h, code)

    # ---- problem -------------------------------------------------------------
    def set_terms(self, mats, scales=None):
        """mats[j] is M_j (dense ndarray, scipy sparse, or None for the identity)."""
        nterms = len(mats)
        n = None
        for M in mats:
            if M is not None:
                if M.shape[0] != M.shape[1]:
                    raise ValueError("Incorrect dimensions of A, must be square")
                n = M.shape[0]
        if n is None:
            raise ValueError("at least one explicit matrix is required")
        sc = np.ones(nterms, dtype=np.complex128) if scales is None else np.asarray(scales, dtype=np.complex128)
        scd = np.ascontiguousarray(sc.view(np.float64))
        types = (C.c_int * nterms)()
        any_sparse = any(_issparse(M) for M in mats if M is not None)
        keep = []
        if any_sparse:
            cps, rvs, vls = (C.c_void_p * nterms)(), (C.c_void_p * nterms)(), (C.c_void_p * nterms)()
            for j, M in enumerate(mats):
                if M is None:
                    cps[j] = rvs[j] = vls[j] = None
                    continue
                (cp, rv, vl), ty = _csc_arg(M)
                keep += [cp, rv, vl]
                cps[j], rvs[j], vls[j], types[j] = cp.ctypes.data, rv.ctypes.data, vl.ctypes.data, ty
            self._ck(self.lib.feasthip_set_csc(self.h, n, nterms, cps, rvs, vls, types,
                                               scd.ctypes.data_as(C.POINTER(C.c_double)), 0))
        else:
            ptrs = (C.c_void_p * nterms)()
            for j, M in enumerate(mats):
                if M is None:
                    ptrs[j] = None
                    continue
                arr, ty = _dense_arg(M)
                keep.append(arr)
                ptrs[j], types[j] = arr.ctypes.data, ty
            self._ck(self.lib.feasthip_set_dense(self.h, n, nterms, ptrs, types,
                                                 scd.ctypes.data_as(C.POINTER(C.c_double))))
        self.n = n
        self.nterms = nterms

    def set_contour(self, contour):
        nodes = np.ascontiguousarray(contour.nodes, dtype=np.complex128)
        weights = np.ascontiguousarray(contour.weights, dtype=np.complex128)
        kind, region = contour.region()
        region = np.ascontiguousarray(region, dtype=np.float64)
        self._ck(self.lib.feasthip_set_contour(self.h, len(nodes), _vp(nodes), _vp(weights), kind, _vp(region)))
        self.K = len(nodes)

    def set_stream(self, stream_ptr):
        """Run the library's launches on a caller-owned HIP stream (e.g. ``torch.cuda.Stream().cuda_stream``); 0 / None
        selects the null stream."""
        self._ck(self.lib.feasthip_set_stream(self.h, C.c_void_p(int(stream_ptr) if stream_ptr else 0)))

    def set_node_shard(self, first, stride):
        self._ck(self.lib.feasthip_set_node_shard(self.h, int(first), int(stride)))

    def set_precision(self, mixed):
        """mixed_prec of feast! (src/feast.jl:19-25): complex64 shifted solves, complex128 accumulation."""
        self._ck(self.lib.feasthip_set_precision(self.h, 1 if mixed else 0))

    def set_conjugate_fold(self, on):
        """Conjugate fold (default on): real coefficient matrices + mirror-symmetric contour => one factorisation per
        conjugate node pair, the mirror system as m0 more right-hand-side columns (feasthip_set_conjugate_fold)."""
        self._ck(self.lib.feasthip_set_conjugate_fold(self.h, 1 if on else 0))

    def set_lazy_conversion(self, on):
        """Factorisations that serve one contour step (store=false) keep their update rows raw (feasthip_set_lazy_conversion); the
        whole-loop calls switch it on by themselves for store=False."""
        self._ck(self.lib.feasthip_set_lazy_conversion(self.h, 1 if on else 0))

    def set_coef_functions(self, funcs):
        """f_t(z) of a split-form NEP (None restores the polynomial default z^t)."""
        if funcs is None:
            self._coef_cb = None
            self._ck(self.lib.feasthip_set_coef_functions(self.h, None, None))
            return
        funcs = list(funcs)
        nt = len(funcs)

        def cb(zp, count, fp, _user):
            for i in range(count):
                z = complex(zp[2 * i], zp[2 * i + 1])
                for t in range(nt):
                    v = complex(funcs[t](z))
                    fp[2 * (i * nt + t)] = v.real
                    fp[2 * (i * nt + t) + 1] = v.imag

        self._coef_cb = _lib.COEF_FN(cb)          # keep the trampoline alive as long as the handle
        self._ck(self.lib.feasthip_set_coef_functions(self.h, C.cast(self._coef_cb, C.c_void_p), None))

    def set_node_coefficients(self, table):
        if table is None:
            self._ck(self.lib.feasthip_set_node_coefficients(self.h, None, 0, 0))
            return
        tab = np.ascontiguousarray(table, dtype=np.complex128)
        self._ck(self.lib.feasthip_set_node_coefficients(self.h, _vp(tab), tab.shape[0], tab.shape[1]))

    def nl_beyn_step(self):
        lam, _, ins = self._outs()
        self._ck(self.lib.feasthip_nl_beyn_step(self.h, _vp(lam), _vp(ins)))
        return lam, ins.astype(bool)

    def nl_set_r(self, R):
        R = np.asfortranarray(R, dtype=np.complex128)
        self._ck(self.lib.feasthip_nl_set_r(self.h, _vp(R)))

    def analyze(self):
        self._ck(self.lib.feasthip_analyze(self.h))

    def factorize(self):
        return self._ck(self.lib.feasthip_factorize(self.h))

    # ---- iteration -------------------------------------------------------------
    def set_x(self, X):
        X = np.asfortranarray(X, dtype=np.complex128)
        if X.shape[0] != self.n:
            raise ValueError("Incorrect dimensions of X, must match A")
        self.m0 = X.shape[1]
        self._ck(self.lib.feasthip_set_x(self.h, _vp(X), X.shape[0], X.shape[1]))

    def get_block(self, which=0):
        if self.n <= 0 or self.m0 <= 0:
            raise _lib.FeastHipError(-4, "get_block: no iteration state on this handle (set_x / *_run first)")
        out = np.empty((self.n, self.m0), dtype=np.complex128, order="F")
        self._ck(self.lib.feasthip_get_block(self.h, which, _vp(out)))
        return out

    @staticmethod
    def _outs_m(m0):
        return np.zeros(m0, dtype=np.complex128), np.zeros(m0), np.zeros(m0, dtype=np.uint8)

    def _outs(self):
        return (np.zeros(self.m0, dtype=np.complex128), np.zeros(self.m0), np.zeros(self.m0, dtype=np.uint8))

    def rr_step(self):
        lam, res, ins = self._outs()
        self._ck(self.lib.feasthip_rr_step(self.h, _vp(lam), _vp(res), _vp(ins)))
        return lam, res, ins.astype(bool)

    def small_eig_herm(self, A, B=None):
        """feasthip_small_eig_herm: the device's reduced Hermitian(-definite) eigensolver alone (m <= 64).  Returns (lambda ascending,
        X with X' B X = I, status[8])."""
        A = np.asfortranarray(A, dtype=np.complex128)
        m = A.shape[0]
        Bf = None if B is None else np.asfortranarray(B, dtype=np.complex128)
        lam = np.zeros(m)
        X = np.zeros((m, m), dtype=np.complex128, order="F")
        st = np.zeros(8)
        self._ck(self.lib.feasthip_small_eig_herm(self.h, m, _vp(A), None if Bf is None else _vp(Bf), _vp(lam), _vp(X), _vp(st)))
        return lam, X, st

    def contour_step(self, refactor=False):
        return self._ck(self.lib.feasthip_contour_step(self.h, 1 if refactor else 0))

    def set_rr_only(self, on=True):
        self._ck(self.lib.feasthip_set_rr_only(self.h, 1 if on else 0))

    def comm_init(self, uid, rank, world):
        """feasthip_comm_init: join the RCCL communicator drawn by `comm_unique_id()` on rank 0; the contour steps of this handle then
        return with the moment summed over the ranks (include/feasthip.h)."""
        buf = (C.c_char * 128).from_buffer_copy(bytes(uid))
        self._ck(self.lib.feasthip_comm_init(self.h, C.cast(buf, C.c_void_p), int(rank), int(world)))

    def comm_destroy(self):
        self._ck(self.lib.feasthip_comm_destroy(self.h))

    def contour_step_begin(self, refactor=False, nchunks=1):
        """feasthip_contour_step_begin: the step enqueued, the moment produced in `nchunks` row ranges (multi-GPU driver)."""
        return self._ck(self.lib.feasthip_contour_step_begin(self.h, 1 if refactor else 0, int(nchunks)))

    def moment_chunk_wait(self, c):
        """Blocks until row range `c` of the moment is final; returns (row0, rows)."""
        r0, rows = C.c_int64(0), C.c_int64(0)
        self._ck(self.lib.feasthip_moment_chunk_wait(self.h, int(c), C.byref(r0), C.byref(rows)))
        return r0.value, rows.value

    def contour_step_end(self):
        return self._ck(self.lib.feasthip_contour_step_end(self.h))

    def nl_orthonormalize_x(self):
        self._ck(self.lib.feasthip_nl_orthonormalize_x(self.h))

    def nl_contour_step(self, first):
        return self._ck(self.lib.feasthip_nl_contour_step(self.h, 1 if first else 0))

    def nl_rr_step(self):
        lam, res, ins = self._outs()
        self._ck(self.lib.feasthip_nl_rr_step(self.h, _vp(lam), _vp(res), _vp(ins)))
        return lam, res, ins.astype(bool)

    def nl_finalize_x(self):
        self._ck(self.lib.feasthip_nl_finalize_x(self.h))

    def q_device(self):
        p, n, ld, nb = C.c_void_p(), C.c_int64(), C.c_int64(), C.c_int()
        self._ck(self.lib.feasthip_q_device(self.h, C.byref(p), C.byref(n), C.byref(ld), C.byref(nb)))
        return p.value, n.value, ld.value, nb.value

    def set_q_buffer(self, dptr, nbytes):
        self._ck(self.lib.feasthip_set_q_buffer(self.h, C.c_void_p(dptr), nbytes))

    def feast_run(self, X, iter, eps, store):
        """feasthip_feast_run: whole `for nit=0:iter` loop in the library.  X (F-ordered
        complex128) is overwritten."""
        if X.shape[0] != self.n:
            raise ValueError("Incorrect dimensions of X, must match A")
        self.m0 = X.shape[1]
        lam, res, ins = self._outs()
        nit = C.c_int(0)
        code = self._ck(self.lib.feasthip_feast_run(self.h, _vp(X), X.shape[0], X.shape[1], int(iter), float(eps),
                                                    1 if store else 0, _vp(lam), _vp(res), _vp(ins), C.byref(nit)))
        return lam, res, ins.astype(bool), nit.value, code

    def nlfeast_run(self, X, iter, eps, spurious):
        self.m0 = X.shape[1]
        lam, res, ins = self._outs()
        nit = C.c_int(0)
        code = self._ck(self.lib.feasthip_nlfeast_run(self.h, _vp(X), X.shape[0], X.shape[1], int(iter), float(eps),
                                                      float(spurious), _vp(lam), _vp(res), _vp(ins), C.byref(nit)))
        return lam, res, ins.astype(bool), nit.value, code

    def nlfeast_moments_run(self, X, moments, iter, eps, spurious):
        self.m0 = X.shape[1]
        M = int(moments) * X.shape[1]
        lam = np.zeros(M, dtype=np.complex128)
        res = np.zeros(M)
        ins = np.zeros(M, dtype=np.uint8)
        Y = np.zeros((X.shape[0], M), dtype=np.complex128, order="F")
        nit = C.c_int(0)
        code = self._ck(self.lib.feasthip_nlfeast_moments_run(self.h, _vp(X), X.shape[0], X.shape[1], int(moments), int(iter), float(eps),
                                                              float(spurious), _vp(lam), _vp(res), _vp(ins), _vp(Y), C.byref(nit)))
        return lam, Y, res, ins.astype(bool), nit.value, code

    def stats(self):
        s = _lib.Stats()
        self._ck(self.lib.feasthip_get_stats(self.h, C.byref(s)))
        return {name: getattr(s, name) for name, _ in _lib.Stats._fields_}

    def level_table(self):
        """Per elimination-tree depth (root = 0): fronts, pivot columns, update rows, factor entries, widest pivot block."""
        t = np.zeros((256, 5))
        nl = self.lib.feasthip_level_table(self.h, _vp(t), 256)
        return t[:max(0, min(nl, 256))]

    def profile_solve(self):
        out = np.zeros((8, 4))
        self._ck(self.lib.feasthip_profile_solve(self.h, _vp(out)))
        det = np.zeros((4096, 6))        # class, ms, bytes, flops, tiles, tree depth
        nd = self.lib.feasthip_profile_detail(self.h, _vp(det), 4096)
        self.last_profile_detail = det[:max(0, min(nd, 4096))]
        names = ["gather_rhs", "fwd_assemble", "diag_solve", "panel_apply", "accumulate"]
        return {nm: dict(ms=out[i, 0], bytes=out[i, 1], flops=out[i, 2], launches=int(out[i, 3])) for i, nm in enumerate(names)}


# ---------------------------------------------------------------------------
def _inplace_target(X):
    X_in = X
    Xw = X if (isinstance(X, np.ndarray) and X.dtype == np.complex128 and X.flags.f_contiguous) else \
        np.asfortranarray(np.asarray(X.toarray() if _issparse(X) else X), dtype=np.complex128)
    return X_in, Xw


def _writeback(X_in, Xw):
    if X_in is not Xw and isinstance(X_in, np.ndarray) and np.issubdtype(X_in.dtype, np.complexfloating):
        X_in[...] = Xw


def gen_feast(X, A, B, contour=None, *, nodes=8, iter=10, c=0j, r=1.0, debug=False, store=False, eps=1e-12,
              factorizer=None, left_divider=None, device=0, info=None, _mixed_prec=False, conjugate_fold=None):
    """gen_feast!(X, A, B[, contour]; ...)  src/feast.jl:82-156.  B=None gives feast!.
    ``conjugate_fold=False`` factorises every contour node separately like the reference even when the matrices are real
    and the contour is mirror-symmetric (FeastHandle.set_conjugate_fold); ``True`` asks for the fold, ``None`` leaves the
    library's default (on, unless FEASTHIP_FOLD=0).

    ``factorizer``/``left_divider`` (src/feast.jl:5,11,84,90): left at their defaults (None, or this module's ``lu`` /
    ``ldiv``) the whole loop runs inside the library (batched factorisation of all nodes, fused sweeps).  Anything else is
    honoured the way the reference does: the node loop (src/feast.jl:59-71 / 134-148) then runs on the host through
    ``F = factorizer(ZmA)``, ``left_divider(temp, F, R)``, ``finalize(F)`` -- exactly the calls, order and arguments of the
    reference -- while the Rayleigh-Ritz half stays on the GPU (``_gen_feast_plugin``).  ``info`` (a dict) receives
    iteration count, library statistics and the static-pivot warning count."""
    if A.shape[0] != A.shape[1]:
        raise ValueError("Incorrect dimensions of A, must be square")
    if X.shape[0] != A.shape[0]:
        raise ValueError("Incorrect dimensions of X, must match A")
    if contour is None:
        contour = circular_contour_trapezoidal(c, r, nodes)
    X_in, Xw = _inplace_target(X)
    if (factorizer is not None and factorizer is not lu) or (left_divider is not None and left_divider is not ldiv):
        lam, res, ins = _gen_feast_plugin(Xw, A, B, contour, iter, eps, debug, store, bool(_mixed_prec), factorizer or lu,
                                          left_divider or ldiv, device, info)
        _writeback(X_in, Xw)
        if not ins.any():
            print("no eigenvalues found in contour!")            # src/feast.jl:78
        return lam[ins], Xw[:, ins], res[ins]
    h = FeastHandle(device)
    try:
        h.set_terms([A, B], scales=[1.0, -1.0])          # T(z) = A - z B
        h.set_contour(contour)
        if conjugate_fold is not None:
            h.set_conjugate_fold(bool(conjugate_fold))
        if _mixed_prec:
            h.set_precision(True)
        if debug:
            # same loop, driven from the host so that every iteration can be reported like
            # iter_debug_print (src/utils.jl:2-42): nit, #inside, #inside below 1e-5, max residual inside
            h.set_lazy_conversion((not store) or iter <= 64)      # what feasthip_feast_run decides for itself: same factors, same bits
            warn = h.factorize() if store else 0
            h.set_x(Xw)
            nit = 0
            for nit in range(iter + 1):
                lam, res, ins = h.rr_step()
                inside_res = res[ins]
                print(f"{nit:3d}: {int(ins.sum()):4d} ({int((inside_res < 1e-5).sum())})  "
                      f"{(inside_res.max() if ins.any() else float('nan')):.3e}")
                if ins.any() and inside_res.max() < eps:
                    print(f"converged in {nit} iteration")
                    break
                if nit < iter:
                    warn = max(warn, h.contour_step(refactor=not store))
            Xw[...] = h.get_block(0)
        else:
            lam, res, ins, nit, warn = h.feast_run(Xw, iter, eps, store)
        if info is not None:
            info.update(h.stats())
            info.update(nit=nit, warn=warn, lam_all=lam.copy(), res_all=res.copy(), inside=ins.copy())
    finally:
        h.close()
    _writeback(X_in, Xw)
    if not ins.any():
        print("no eigenvalues found in contour!")                # src/feast.jl:78
    return lam[ins], Xw[:, ins], res[ins]


def _gen_feast_plugin(Xw, A, B, contour, iter, eps, debug, store, mixed_prec, factorizer, left_divider, device, info):
    """feast!/gen_feast! with a caller-supplied factorizer / left_divider: the reference's loop statement by statement
    (src/feast.jl:19-76 / 98-152).  The shifted matrix is assembled on the host in the caller's matrix type (scipy sparse
    stays sparse, arrays stay arrays; ComplexF32 when mixed_prec), every node goes through the plug-ins in the reference's
    order (store: all factorisations first, :28-38; otherwise linsolve! = factor, solve, finalize per node and iteration,
    src/utils.jl:175-179), Q is accumulated on the host; the Rayleigh-Ritz half (:41-50) is the library's rr_step."""
    n, m0 = Xw.shape
    ctype = np.complex64 if mixed_prec else np.complex128
    z, w = np.asarray(contour.nodes, dtype=np.complex128), np.asarray(contour.weights, dtype=np.complex128)
    K = len(z)
    sparse = _issparse(A)
    Ac = sp.csc_matrix(A, dtype=np.complex128) if sparse else np.asarray(A, dtype=np.complex128)
    if B is None:
        Bc = sp.identity(n, dtype=np.complex128, format="csc") if sparse else np.eye(n, dtype=np.complex128)
    elif sparse:
        Bc = sp.csc_matrix(B, dtype=np.complex128)
    else:
        Bc = np.asarray(B.toarray() if _issparse(B) else B, dtype=np.complex128)

    def shifted(k):                                                   # ZmA .= (A - B*contour.nodes[i])   :29,35,65 / 105,111,141
        M = Ac - Bc * z[k]
        return sp.csc_matrix(M, dtype=ctype) if sparse else np.asfortranarray(M, dtype=ctype)

    h = FeastHandle(device)
    facts = []
    try:
        h.set_terms([A, B], scales=[1.0, -1.0])
        h.set_contour(contour)
        h.set_rr_only(True)
        if store:
            facts = [factorizer(shifted(k)) for k in range(K)]       # :28-38
        temp = np.zeros((n, m0), dtype=ctype, order="F")
        Q = Xw.copy(order="F")
        nit = 0
        for nit in range(iter + 1):                                   # :40
            h.set_x(Q)                                                # Q into the handle; rr_step orthonormalises it   :41
            lam, res, ins = h.rr_step()                               # :41-50
            if debug:
                ir = res[ins]
                print(f"{nit:3d}: {int(ins.sum()):4d} ({int((ir < 1e-5).sum())})  {(ir.max() if ins.any() else float('nan')):.3e}")
            if ins.any() and res[ins].max() < eps:                    # :53-56
                break
            if nit < iter:                                            # :57
                Xh, R = h.get_block(0), h.get_block(1)
                Q[...] = 0.0
                for k in range(K):
                    resolvent = 1.0 / (z[k] - lam)                    # :60
                    if store:
                        left_divider(temp, facts[k], R)               # :62
                    else:                                             # linsolve!  src/utils.jl:175-179
                        F = factorizer(shifted(k))
                        left_divider(temp, F, R)
                        finalize(F)
                    Q += (Xh - temp) * (resolvent * w[k])[None, :]    # :68-70
        Xw[...] = h.get_block(0)
        if info is not None:
            info.update(h.stats())
            info.update(nit=nit, warn=0, lam_all=lam.copy(), res_all=res.copy(), inside=ins.copy(), plugin_seam=True)
    finally:
        for F in facts:                                               # foreach(finalize!, facts)   :74-76
            finalize(F)
        h.close()
    return lam, res, ins


def trim_memory():
    """Return the library's cache of released device blocks to the driver (feasthip_trim_memory): handles that are closed
    leave their large buffers (factor arena, block vectors) for the next handle of the process."""
    _lib.load_library().feasthip_trim_memory()


def comm_unique_id():
    """feasthip_comm_unique_id: 128 bytes (RCCL's ncclUniqueId) for rank 0 to hand to the other ranks."""
    lib = _lib.load_library()
    buf = (C.c_char * 128)()
    code = lib.feasthip_comm_unique_id(C.cast(buf, C.c_void_p))
    if code != 0:
        msg = lib.feasthip_last_error(None)
        raise _lib.FeastHipError(code, msg.decode() if msg else "feasthip_comm_unique_id failed")
    return bytes(buf)


def mmread(path):
    """MatrixMarket coordinate file -> scipy CSC complex128 through the library's own reader (feasthip_mm_read_csc);
    what MatrixMarket.mmread does in the reference's scripts (test/butterfly.jl:57-59, test/polynomial.jl:5-11)."""
    lib = _lib.load_library()
    rows, cols, nnz, field = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int(0)
    bp = str(path).encode()
    if lib.feasthip_mm_info(bp, C.byref(rows), C.byref(cols), C.byref(nnz), C.byref(field)) != 0:
        raise ValueError("mmread: " + (lib.feasthip_mm_last_error() or b"").decode())
    cp = np.zeros(cols.value + 1, dtype=np.int64)
    rv = np.zeros(max(1, nnz.value), dtype=np.int64)
    vl = np.zeros(max(1, nnz.value), dtype=np.complex128)
    if lib.feasthip_mm_read_csc(bp, _vp(cp), _vp(rv), _vp(vl), nnz.value) != 0:
        raise ValueError("mmread: " + (lib.feasthip_mm_last_error() or b"").decode())
    return sp.csc_matrix((vl[:nnz.value], rv[:nnz.value], cp), shape=(rows.value, cols.value))


def mmwrite(path, M):
    """scipy sparse / dense matrix -> MatrixMarket `coordinate <field> general` (the reference's writer, test/gen_butterfly.jl:6-41)."""
    lib = _lib.load_library()
    (cp, rv, vl), ty = _csc_arg(sp.csc_matrix(M))
    if lib.feasthip_mm_write_csc(str(path).encode(), M.shape[0], M.shape[1], _vp(cp), _vp(rv), _vp(vl), ty, 0) != 0:
        raise ValueError("mmwrite: " + (lib.feasthip_mm_last_error() or b"").decode())


def _adjoint(M):
    return M.conj().T.tocsc() if _issparse(M) else np.ascontiguousarray(np.asarray(M).conj().T)


def dual_gen_feast(Xr, Xl, A, B, contour=None, *, nodes=8, iter=10, c=0j, r=1.0, debug=False, store=False, eps=1e-12,
                   factorizer=None, left_divider=None, device=0, info=None):
    """dual_gen_feast!(Xr, Xl, A, B[, contour]; ...)  src/feast.jl:158-257: right and left eigenvectors of A x = lam B x
    (y' A = lam y' B).  Returns (lam[inside], Xr[:, inside], Xl[:, inside], resr[inside]); Xr, Xl are overwritten."""
    if A.shape[0] != A.shape[1]:
        raise ValueError("Incorrect dimensions of A, must be square")
    if Xl.shape[0] != A.shape[0] or Xr.shape != Xl.shape:
        raise ValueError("Incorrect dimensions of X, must match A")
    if contour is None:
        contour = circular_contour_trapezoidal(c, r, nodes)
    if B is None:
        B = sp.identity(A.shape[0], format="csc", dtype=np.complex128) if _issparse(A) else np.eye(A.shape[0], dtype=np.complex128)
    (Xr_in, Xrw), (Xl_in, Xlw) = _inplace_target(Xr), _inplace_target(Xl)
    K = len(contour.nodes)
    z = np.asarray(contour.nodes, dtype=np.complex128)
    w = np.asarray(contour.weights, dtype=np.complex128)
    import copy
    # rounds 2-3: the adjoint matrices factorised as K more nodes; also the layout under FEASTHIP_FWD_SPLITK=1, whose forward plans the
    # adjoint sweeps cannot walk (engine_solve.hip: solve_range_adj)
    doubled = os.environ.get("FEASTHIP_DUAL_DOUBLED", "0") == "1" or os.environ.get("FEASTHIP_FWD_SPLITK", "0") not in ("", "0")
    dual_contour = copy.copy(contour)
    if doubled:
        dual_contour.nodes, dual_contour.weights = np.concatenate([z, z.conj()]), np.concatenate([w, w.conj()])
        table = np.zeros((2 * K, 4), dtype=np.complex128)
        table[:K, 0], table[:K, 1] = 1.0, z                        # A - z_k B          (scales 1, -1)
        table[K:, 2], table[K:, 3] = 1.0, z.conj()                 # A' - conj(z_k) B'  (the reference's lfacts, :185-197)
    else:
        # K factorisations of A - z_k B; the left systems (A - z_k B)' y = r (lfacts, :185-197, :241) are adjoint sweeps over the same
        # factors (the terms A', B' stay for the left residual A' y - conj(lam) B' y, :214)
        table = np.zeros((K, 4), dtype=np.complex128)
        table[:, 0], table[:, 1] = 1.0, z
    h = FeastHandle(device)
    try:
        h.set_terms([A, B, _adjoint(A), _adjoint(B)], scales=[1.0, -1.0, 1.0, -1.0])
        h.set_contour(dual_contour)
        h.set_node_coefficients(table)
        m0 = Xrw.shape[1]
        lam, resr, ins = h._outs_m(m0)
        resl = np.zeros(m0)
        nit = C.c_int(0)
        warn = h._ck(h.lib.feasthip_dual_feast_run(h.h, _vp(Xrw), _vp(Xlw), Xrw.shape[0], m0, int(iter), float(eps), 1 if store else 0,
                                                   _vp(lam), _vp(resr), _vp(resl), _vp(ins), C.byref(nit)))
        ins = ins.astype(bool)
        if info is not None:
            info.update(h.stats())
            info.update(nit=nit.value, warn=warn, lam_all=lam.copy(), resr_all=resr.copy(), resl_all=resl.copy(), inside=ins.copy())
    finally:
        h.close()
    _writeback(Xr_in, Xrw)
    _writeback(Xl_in, Xlw)
    if not ins.any():
        print("no eigenvalues found in contour!")                # src/feast.jl:255
    return lam[ins], Xrw[:, ins], Xlw[:, ins], resr[ins]


def feast(X, A, contour=None, *, nodes=8, iter=10, c=0j, r=1.0, eps=1e-12, debug=False, store=False,
          mixed_prec=False, factorizer=None, left_divider=None, device=0, info=None):
    """feast!(X, A[, contour]; ...)  src/feast.jl:3-80."""
    # mixed_prec (src/feast.jl:10,19-25): complex64 shifted solves with complex128 accumulation (feasthip_set_precision)
    return gen_feast(X, A, None, contour, nodes=nodes, iter=iter, c=c, r=r, debug=debug, store=store, eps=eps,
                     factorizer=factorizer, left_divider=left_divider, device=device, info=info, _mixed_prec=bool(mixed_prec))


def _fro(M):
    return float(np.sqrt((abs(M) ** 2).sum())) if _issparse(M) else float(np.linalg.norm(M))


def _nlfeast_closure(T, Xw, nodes, iter, c, r, eps, spurious, device, info):
    """nlfeast! for an arbitrary closure z -> matrix (the reference's interface, src/nlfeast.jl:2): T(z_k) for every
    node goes to the device once (one term per node), the loop runs from the host with the phase calls and the
    residual block R[:,i] = T(lam_i) x_i is formed on the host like the reference's update_R! (src/utils.jl:104-109)."""
    con = circular_contour_trapezoidal(c, r, nodes)
    h = FeastHandle(device)
    try:
        h.set_terms([T(z) for z in con.nodes])
        h.set_contour(con)
        h.set_node_coefficients(np.eye(nodes, dtype=np.complex128))
        h.set_x(Xw)
        h.nl_orthonormalize_x()                                            # src/nlfeast.jl:12-13
        warn = h.factorize()                                               # :17-28
        m0 = Xw.shape[1]
        res = np.zeros(m0)
        R = np.zeros(Xw.shape, dtype=np.complex128, order="F")
        nit = 0
        for nit in range(iter + 1):
            h.nl_contour_step(nit == 0)                                    # :36-61
            lam, ins = h.nl_beyn_step()                                    # :64
            Xc = h.get_block(0)
            for i in range(m0):                                            # update_R!, residuals  :66-67
                Tl = T(lam[i])
                R[:, i] = Tl @ Xc[:, i]
                res[i] = np.linalg.norm(R[:, i]) / _fro(Tl)
            h.nl_set_r(R)
            ri = res[ins]
            if ri.size > 0 and ri.max() < eps:                             # :73-76
                break
            if nit > 1 and (ri < spurious).sum() > 0 and ri[ri < spurious].max() < eps:   # :77-79
                break
        h.nl_finalize_x()                                                  # :82
        Xw[...] = h.get_block(0)
        if info is not None:
            info.update(h.stats())
            info.update(nit=min(nit, iter), warn=warn, inside=ins.copy())
    finally:
        h.close()
    return lam, res


def nlfeast(T, X, nodes, iter, *, c=0j, r=1.0, debug=False, eps=10e-12, store=True, spurious=1e-5,
            factorizer=None, left_divider=None, device=0, info=None):
    """nlfeast!(T, X, nodes, iter; ...)  src/nlfeast.jl:2-84.  Returns (lam, X, res) for ALL m0
    pairs (unfiltered).  ``T`` is a PolynomialNEP or SplitNEP (device fast path: matrices resident, residuals on
    the GPU) or any callable z -> matrix (node matrices on the device, residual block formed on the host)."""
    if not (isinstance(X, np.ndarray) and X.dtype == np.complex128):
        raise TypeError("X must be an AbstractMatrix{ComplexF64}")       # src/nlfeast.jl:2
    if not callable(T):
        raise TypeError("T must be callable: z -> matrix")
    X_in, Xw = _inplace_target(X)
    if not isinstance(T, (PolynomialNEP, SplitNEP)):
        lam, res = _nlfeast_closure(T, Xw, nodes, iter, c, r, eps, spurious, device, info)
        _writeback(X_in, Xw)
        return lam, Xw, res
    h = FeastHandle(device)
    try:
        h.set_terms(T.coeffs)
        if isinstance(T, SplitNEP):
            h.set_coef_functions(T.funcs)
        h.set_contour(circular_contour_trapezoidal(c, r, nodes))           # src/nlfeast.jl:8
        lam, res, ins, nit, warn = h.nlfeast_run(Xw, iter, eps, spurious)
        if info is not None:
            info.update(h.stats())
            info.update(nit=nit, warn=warn, inside=ins.copy())
    finally:
        h.close()
    _writeback(X_in, Xw)
    return lam, Xw, res


def _nlfeast_moments_closure(T, Xw, nodes, iter, c, r, eps, moments, spurious, device, info):
    """nlfeast_moments! for an arbitrary closure z -> matrix (test/gun.jl:36, test/fiber.jl:33): node matrices on the
    device, the loop of src/nlfeast.jl:236-303 on the host with the phase calls, residual block of the candidates formed
    on the host like update_R_moments! (src/utils.jl:118-123)."""
    con = circular_contour_trapezoidal(c, r, nodes)
    n, m0 = Xw.shape
    M = int(moments) * m0
    h = FeastHandle(device)
    try:
        h.set_terms([T(z) for z in con.nodes])
        h.set_contour(con)
        h.set_node_coefficients(np.eye(nodes, dtype=np.complex128))
        warn = h._ck(h.lib.feasthip_nlm_begin(h.h, _vp(Xw), n, m0, int(moments)))
        h.m0 = m0                                                          # get_block sizes its output from it
        lam = np.zeros(M, dtype=np.complex128)
        lam_s = np.zeros(M, dtype=np.complex128)
        res = np.zeros(M)
        ins = np.zeros(M, dtype=np.uint8)
        Y = np.zeros((n, M), dtype=np.complex128, order="F")
        R = np.zeros((n, M), dtype=np.complex128, order="F")
        res_in = np.zeros(M)
        nit = 0
        for nit in range(iter + 1):
            h._ck(h.lib.feasthip_nlm_contour_step(h.h, 1 if nit == 0 else 0))
            h._ck(h.lib.feasthip_nlm_candidates(h.h, _vp(lam), _vp(Y)))
            for i in range(M):
                if not np.isfinite(lam[i]):                               # beyond the numerical rank of the Hankel matrix
                    R[:, i] = 0.0
                    res_in[i] = np.inf
                    continue
                Tl = T(lam[i])
                R[:, i] = Tl @ Y[:, i]
                res_in[i] = np.linalg.norm(R[:, i]) / _fro(Tl)
            h._ck(h.lib.feasthip_nlm_finish(h.h, _vp(R), _vp(res_in), _vp(lam_s), _vp(res), _vp(ins)))
            if nit == 0:
                continue                                                   # the reference tests inside `for nit = 1:iter` only
            i0 = ins[:m0].astype(bool)
            ri = res[:m0][i0]
            if ri.size > 0 and ri.max() < eps:                             # :296-299
                break
            if nit > 1 and (ri < spurious).sum() > 0 and ri[ri < spurious].max() < eps:   # :300-302
                break
        h.nl_finalize_x()                                                  # normalize!(X) :311
        Xw[...] = h.get_block(0)
        h._ck(h.lib.feasthip_nlm_get_y(h.h, _vp(Y)))
        if info is not None:
            info.update(h.stats())
            info.update(nit=min(nit, iter), warn=warn, inside=ins.astype(bool))
    finally:
        h.close()
    return lam_s, Y, res


def nlfeast_moments(T, X, nodes, iter, *, c=0j, r=1.0, debug=False, eps=10e-12, moments=2, store=True, spurious=1e-5,
                    device=0, info=None):
    """nlfeast_moments!(T, X, nodes, iter; c, r, debug, eps, moments, store, spurious)  src/nlfeast.jl:173-318.
    Returns (lam, Y, res) for all moments*m0 candidates sorted by relative residual; X is overwritten with the m0
    best candidates, column-normalised.  T must be a PolynomialNEP.  ``store`` is accepted for signature
    compatibility: the factors of the nodes are always kept on the device (the reference's store=true)."""
    if not (isinstance(X, np.ndarray) and X.dtype == np.complex128):
        raise TypeError("X must be an AbstractMatrix{ComplexF64}")       # src/nlfeast.jl:173
    if not callable(T):
        raise TypeError("T must be callable: z -> matrix")
    X_in, Xw = _inplace_target(X)
    if not isinstance(T, (PolynomialNEP, SplitNEP)):
        lam, Y, res = _nlfeast_moments_closure(T, Xw, nodes, iter, c, r, eps, moments, spurious, device, info)
        _writeback(X_in, Xw)
        return lam, Y, res
    h = FeastHandle(device)
    try:
        h.set_terms(T.coeffs)
        if isinstance(T, SplitNEP):
            h.set_coef_functions(T.funcs)
        h.set_contour(circular_contour_trapezoidal(c, r, nodes))           # src/nlfeast.jl:178
        lam, Y, res, ins, nit, warn = h.nlfeast_moments_run(Xw, moments, iter, eps, spurious)
        if debug:
            m0 = Xw.shape[1]
            print(f"nit={nit} inside(first m0)={int(ins[:m0].sum())} max_res_inside={res[:m0][ins[:m0]].max() if ins[:m0].any() else float('nan'):.3e}")
        if info is not None:
            info.update(h.stats())
            info.update(nit=nit, warn=warn, inside=ins.copy())
    finally:
        h.close()
    _writeback(X_in, Xw)
    return lam, Y, res


def contour_estimate_eig(A, contour, B=None, *, samples=None, X=None, device=0, seed=None):
    """contour_estimate_eig(A, contour, B=I; samples)  src/stochastic.jl:2-33: stochastic estimate of the number of
    eigenvalues inside the contour.  ``X`` (n x samples complex Gaussian block) may be passed for reproducibility."""
    n = A.shape[0]
    if X is None:
        m = min(100, n) if samples is None else int(samples)
        rng = np.random.default_rng(seed)
        X = (rng.standard_normal((n, m)) + 1j * rng.standard_normal((n, m))) / np.sqrt(2.0)     # randn(ComplexF64, N, m)
    X = np.asfortranarray(X, dtype=np.complex128)
    h = FeastHandle(device)
    try:
        h.set_terms([A, B], scales=[1.0, -1.0])
        h.set_contour(contour)
        est = C.c_double(0.0)
        h._ck(h.lib.feasthip_estimate_count(h.h, _vp(X), X.shape[0], X.shape[1], C.byref(est)))
    finally:
        h.close()
    return est.value


# ---------------------------------------------------------------------------
# factorizer / left_divider / finalize! seam  (src/utils.jl:173-179)
# ---------------------------------------------------------------------------
class HipLU:
    """Factorization object returned by ``lu`` (the reference's `factorizer`)."""

    def __init__(self, handle, n, lib, info=0):
        self._h, self.n, self._lib = handle, n, lib
        self.info = info          # > 0: number of statically perturbed pivots (LAPACK info convention)
        sym, gr = C.c_int(0), C.c_double(0.0)
        lib.feasthip_fact_info(handle, C.byref(sym), C.byref(gr), None)
        # path 1: pivot-light elimination of a symmetric matrix (C == C^T, U = D L^T); 2: elimination without interchanges of a matrix that is
        # Hermitian off its diagonal (L and U apart); 0: partial pivoting.  growth: what the growth check of paths 1 / 2 saw.
        self.path, self.symmetric, self.growth = sym.value, sym.value == 1, gr.value

    def __del__(self):
        try:
            finalize(self)
        except Exception:
            pass


def lu(Cm, device=0):
    """factorizer(C): GPU LU of the assembled shifted matrix; C is copied (src/feast.jl:29-36).  A complex64 matrix
    (the ComplexF32 ZmA of the mixed_prec path, src/feast.jl:19-25) gives a factor object that solves in complex64."""
    lib = _lib.load_library()
    F = C.c_void_p()
    if Cm.shape[0] != Cm.shape[1]:
        raise ValueError("Incorrect dimensions of A, must be square")
    n = Cm.shape[0]
    c64 = Cm.dtype == np.complex64
    if _issparse(Cm):
        (cp, rv, vl), ty = _csc_arg(Cm)
        if c64:
            vl, ty = np.ascontiguousarray(vl, dtype=np.complex64), 3          # FEASTHIP_C64
        code = lib.feasthip_lu_csc(int(device), n, _vp(cp), _vp(rv), _vp(vl), ty, 0, C.byref(F))
    elif c64:
        arr = np.asfortranarray(Cm, dtype=np.complex64)
        code = lib.feasthip_lu_dense_c64(int(device), _vp(arr), n, C.byref(F))
    else:
        arr = np.asfortranarray(Cm, dtype=np.complex128)
        code = lib.feasthip_lu_dense(int(device), _vp(arr), n, C.byref(F))
    if code < 0:
        msg = lib.feasthip_last_error(None)
        raise _lib.FeastHipError(code, msg.decode() if msg else "")
    return HipLU(F, n, lib, code)


def ldiv(Y, F, R, adjoint=False):
    """left_divider(Y, F, R): Y .= C^-1 R, R untouched (src/feast.jl:62).  adjoint=True: Y .= C^-H R over the same factor object
    (Julia's `ldiv!(Y, F', R)`; the reference factorises C' again for it, src/feast.jl:185-197)."""
    Rw = np.asfortranarray(R, dtype=np.complex128)
    Yw = Y if (Y.dtype == np.complex128 and Y.flags.f_contiguous) else np.empty(Rw.shape, dtype=np.complex128, order="F")
    fn = F._lib.feasthip_ldiv_adjoint if adjoint else F._lib.feasthip_ldiv
    code = fn(F._h, _vp(Yw), Yw.shape[0], _vp(Rw), Rw.shape[0], Rw.shape[1])
    if code < 0:
        raise _lib.FeastHipError(code, "feasthip_ldiv failed")
    if Yw is not Y:
        Y[...] = Yw
    return Y


def finalize(F):
    """finalize!(F): release the device memory of a factor object (src/utils.jl:173)."""
    if getattr(F, "_h", None):
        F._lib.feasthip_fact_free(F._h)
        F._h = None
