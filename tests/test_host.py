"""Host-side logic and the C-ABI surface, no GPU needed: the library loads and exports
every symbol of include/feasthip.h, the symbolic analysis predicts a superset of the true
LU fill, the small dense eigensolver matches LAPACK, contours match the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import scipy.linalg as sla
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported(lib, pkg):
    hdr = open(os.path.join(ROOT, "include", "feasthip.h")).read()
    names = sorted(set(re.findall(r"\b(feasthip_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 30
    import importlib
    _lib = importlib.import_module(pkg.__name__ + "._lib")
    for nm in names:
        assert hasattr(lib, nm), f"{nm} declared in feasthip.h but not exported"
        assert nm in _lib.PROTOTYPES, f"{nm} has no ctypes prototype"


def test_no_gpu_fails_loudly(lib, pkg):
    if lib.feasthip_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(pkg.FeastHipError):
        pkg.FeastHandle(0)
    with pytest.raises(pkg.FeastHipError):
        pkg.gen_feast(np.ones((4, 2), dtype=complex), np.eye(4), np.eye(4))
    with pytest.raises(pkg.FeastHipError):
        pkg.lu(np.eye(4, dtype=complex))


def _probe(lib, A, ordering=0):
    A = sp.csc_matrix(A)
    n = A.shape[0]
    cp = np.ascontiguousarray(A.indptr, dtype=np.int64)
    ri = np.ascontiguousarray(A.indices, dtype=np.int64)
    stats = np.zeros(8, dtype=np.int64)
    perm = np.zeros(n, dtype=np.int32)
    snof = np.zeros(n, dtype=np.int32)
    rc = lib.feasthip_symbolic_probe(n, cp.ctypes.data, ri.ctypes.data, 0, ordering, stats.ctypes.data, perm.ctypes.data,
                                     snof.ctypes.data, None)
    assert rc == 0
    return stats, perm, snof, cp, ri


def _lu_nopivot_fill(M):
    """pattern of L+U of a dense matrix eliminated without pivoting"""
    M = M.copy()
    n = M.shape[0]
    for k in range(n):
        M[k + 1:, k] /= M[k, k]
        M[k + 1:, k + 1:] -= np.outer(M[k + 1:, k], M[k, k + 1:])
    return np.abs(M) > 1e-13


@pytest.mark.parametrize("case", ["lap1d", "grid2d", "grid3d_19", "random", "arrow", "blockdiag"])
def test_symbolic_structure_contains_fill(lib, wl, case):
    rng = np.random.default_rng(7)
    if case == "lap1d":
        A = sp.diags([-np.ones(99), 2 * np.ones(100), -np.ones(99)], [-1, 0, 1], format="csc")
    elif case == "grid2d":
        g = 12
        T = sp.diags([-np.ones(g - 1), 2 * np.ones(g), -np.ones(g - 1)], [-1, 0, 1])
        A = sp.csc_matrix(sp.kron(sp.identity(g), T) + sp.kron(T, sp.identity(g)))
    elif case == "grid3d_19":
        A = wl.laplacian_3d(6, 19)
    elif case == "random":
        A = sp.random(150, 150, density=0.02, random_state=3, format="csc") + sp.identity(150, format="csc")
    elif case == "arrow":
        A = sp.lil_matrix((60, 60)); A.setdiag(1.0); A[0, :] = 1.0; A[:, 0] = 1.0; A[30, :] = 1.0; A = sp.csc_matrix(A)
    else:
        A = sp.block_diag([sp.random(20, 20, density=0.3, random_state=k) + sp.identity(20) for k in range(4)], format="csc")
    n = A.shape[0]
    stats, perm, snof, cp, ri = _probe(lib, A)
    assert sorted(perm.tolist()) == list(range(n))
    # numeric no-pivot LU of the symmetrised, permuted, diagonally dominant pattern
    P = (abs(A) + abs(A).T).toarray() > 0
    M = np.where(P, rng.uniform(0.5, 1.5, (n, n)), 0.0)
    M[np.arange(n), np.arange(n)] = n * 2.0
    Mp = M[np.ix_(perm, perm)]
    fill = _lu_nopivot_fill(Mp)
    ii, jj = np.nonzero(fill)
    oi, oj = perm[ii].astype(np.int32), perm[jj].astype(np.int32)   # back to original labels
    bad = lib.feasthip_symbolic_contains(n, cp.ctypes.data, ri.ctypes.data, 0, 0, np.ascontiguousarray(oi).ctypes.data,
                                         np.ascontiguousarray(oj).ctypes.data, len(oi))
    assert bad == 0
    # stored entries >= true fill, and not absurdly larger
    assert stats[3] >= fill.sum() - n
    assert stats[1] + stats[2] <= 6 * fill.sum() + 64 * n


def test_symbolic_quality_3d(lib, wl):
    A = wl.laplacian_3d(16, 7)
    stats, perm, *_ = _probe(lib, A)
    n = A.shape[0]
    # natural (banded) ordering would give about n * g^2 = 1.05e6 per factor; ND must beat it clearly
    assert stats[1] < 0.6 * n * 16 * 16
    assert stats[6] < 40      # tree depth stays logarithmic


def test_symmetric_fast_path_gives_the_general_path_s_graph(lib, wl):
    """analyze_pattern takes the adjacency of a structurally symmetric pattern with ascending row indices straight from the pattern
    (no transposition); the general path (here: the same columns with their row indices reversed, which only costs it a sort) must see
    the same graph: same ordering, same supernodes, same fill.  Above 20,000 vertices both paths run on host threads."""
    for g in (9, 28):
        A = sp.csc_matrix(wl.laplacian_3d(g, 19))
        A.sort_indices()
        s1, p1, sn1, cp, ri = _probe(lib, A)
        B = A.copy()
        for j in range(B.shape[0]):
            B.indices[B.indptr[j]:B.indptr[j + 1]] = B.indices[B.indptr[j]:B.indptr[j + 1]][::-1].copy()
        B.has_sorted_indices = False
        n = B.shape[0]
        cpb = np.ascontiguousarray(B.indptr, dtype=np.int64); rib = np.ascontiguousarray(B.indices, dtype=np.int64)
        s2 = np.zeros(8, dtype=np.int64); p2 = np.zeros(n, dtype=np.int32); sn2 = np.zeros(n, dtype=np.int32)
        assert lib.feasthip_symbolic_probe(n, cpb.ctypes.data, rib.ctypes.data, 0, 0, s2.ctypes.data, p2.ctypes.data, sn2.ctypes.data, None) == 0
        assert np.array_equal(s1, s2) and np.array_equal(p1, p2) and np.array_equal(sn1, sn2)


def test_python_constants_match_the_header(pkg):
    """element-type tags and the row-major flag of feasthip_set_dense: the Python mirror uses the header's values"""
    import importlib
    import __graft_entry__ as graft
    api = importlib.import_module(graft.PKG_NAME + ".api")
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "feasthip.h")).read()
    m = re.search(r"enum\s*\{\s*FEASTHIP_F64\s*=\s*(\d+),\s*FEASTHIP_C128\s*=\s*(\d+),\s*FEASTHIP_I64\s*=\s*(\d+),\s*FEASTHIP_C64\s*=\s*(\d+)", hdr)
    assert m and tuple(int(v) for v in m.groups()) == (api.F64, api.C128, api.I64T, api.C64T)
    m = re.search(r"#define\s+FEASTHIP_ROW_MAJOR\s+(0x[0-9a-fA-F]+)", hdr)
    assert m and int(m.group(1), 16) == api.ROW_MAJOR


def test_ordering_does_not_depend_on_the_trace_switch(lib, wl, capfd):
    """FEASTHIP_ANALYZE_TRACE only prints: the ordering and the supernodes are the same with and without it, on a graph large enough
    for the threaded top levels of the dissection (the switch once decided whether those levels ran their multilevel separator once
    or twice), and the analysis is reproducible run to run."""
    A = wl.laplacian_3d(28, 7)          # 21,952 vertices: the asynchronous multilevel path of the top level
    runs = []
    for trace in (None, "1", None):
        old = os.environ.pop("FEASTHIP_ANALYZE_TRACE", None)
        if trace:
            os.environ["FEASTHIP_ANALYZE_TRACE"] = trace
        try:
            stats, perm, snof, *_ = _probe(lib, A)
        finally:
            os.environ.pop("FEASTHIP_ANALYZE_TRACE", None)
            if old is not None:
                os.environ["FEASTHIP_ANALYZE_TRACE"] = old
        runs.append((stats.copy(), perm.copy(), snof.copy()))
    capfd.readouterr()
    for st, pm, sn in runs[1:]:
        assert (st == runs[0][0]).all() and (pm == runs[0][1]).all() and (sn == runs[0][2]).all()


def test_small_eig_matches_lapack(lib):
    rng = np.random.default_rng(11)
    for m in (1, 2, 5, 16, 33, 64, 81, 100, 128):      # from 80 on the QR iteration applies the column rotations in groups of 8
        A = rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m))
        B = np.eye(m) + 0.2 * (rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m)))
        for Bm in (None, B):
            lam = np.zeros(m, dtype=np.complex128)
            V = np.zeros((m, m), dtype=np.complex128, order="F")
            Af = np.asfortranarray(A)
            Bf = None if Bm is None else np.asfortranarray(Bm)
            rc = lib.feasthip_small_eig(m, Af.ctypes.data, None if Bf is None else Bf.ctypes.data, lam.ctypes.data, V.ctypes.data)
            assert rc == 0
            ref = sla.eigvals(A) if Bm is None else sla.eigvals(A, Bm)
            assert np.abs(np.sort_complex(lam) - np.sort_complex(ref)).max() < 1e-9 * max(1, np.abs(ref).max())
            # sorted by (re, im) like Julia's eigen
            assert all((lam[i].real, lam[i].imag) <= (lam[i + 1].real, lam[i + 1].imag) for i in range(m - 1))
            Bx = V if Bm is None else Bm @ V
            resid = np.linalg.norm(A @ V - Bx * lam[None, :], axis=0)
            assert resid.max() < 1e-9 * np.linalg.norm(A)
            assert np.allclose(np.linalg.norm(V, axis=0), 1.0)


def test_small_eig_hermitian_and_defective(lib):
    # Hermitian input: real eigenvalues; Jordan block: still converges
    rng = np.random.default_rng(5)
    G = rng.standard_normal((20, 20)) + 1j * rng.standard_normal((20, 20))
    H = np.asfortranarray(G + G.conj().T)
    lam = np.zeros(20, dtype=np.complex128); V = np.zeros((20, 20), dtype=np.complex128, order="F")
    assert lib.feasthip_small_eig(20, H.ctypes.data, None, lam.ctypes.data, V.ctypes.data) == 0
    assert np.abs(lam.imag).max() < 1e-10 and np.abs(np.sort(lam.real) - np.linalg.eigvalsh(H)).max() < 1e-10
    J = np.asfortranarray(np.diag(np.full(6, 2.0 + 0j)) + np.diag(np.ones(5), 1))
    lam = np.zeros(6, dtype=np.complex128); V = np.zeros((6, 6), dtype=np.complex128, order="F")
    assert lib.feasthip_small_eig(6, J.ctypes.data, None, lam.ctypes.data, V.ctypes.data) == 0
    assert np.abs(lam - 2.0).max() < 1e-6


def test_small_eig_hermitian_definite_pencil(lib):
    """the path every FEAST iteration of a Hermitian(-definite) problem takes (Julia's eigen! on Hermitian input):
    eigenvalues and B-normalised residuals against LAPACK zhegv, odd and even sizes."""
    rng = np.random.default_rng(23)
    for m in (1, 2, 3, 17, 64, 100):
        G = rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m))
        A = np.asfortranarray(G + G.conj().T)
        H = rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m))
        B = np.asfortranarray(H @ H.conj().T + m * np.eye(m))
        lam = np.zeros(m, dtype=np.complex128)
        V = np.zeros((m, m), dtype=np.complex128, order="F")
        assert lib.feasthip_small_eig(m, A.ctypes.data, B.ctypes.data, lam.ctypes.data, V.ctypes.data) == 0
        ref = sla.eigh(A, B, eigvals_only=True)
        assert np.abs(lam.imag).max() == 0 and np.abs(lam.real - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())
        assert np.linalg.norm(A @ V - (B @ V) * lam[None, :], axis=0).max() < 1e-11 * np.linalg.norm(A)
        assert np.allclose(np.linalg.norm(V, axis=0), 1.0)
        off = V.conj().T @ B @ V
        assert np.abs(off - np.diag(np.diag(off))).max() < 1e-10 * np.abs(np.diag(off)).max()     # B-orthogonal eigenvectors


def test_contours_match_oracle(pkg, fo):
    for N in (4, 8, 16, 32):
        for name in ("circular_contour_trapezoidal", "circular_contour_gauss"):
            a, b = getattr(pkg, name)(0.3 + 0.1j, 0.7, N), getattr(fo, name)(0.3 + 0.1j, 0.7, N)
            assert np.abs(a.nodes - b.nodes).max() < 1e-15 and np.abs(a.weights - b.weights).max() < 1e-15
        for name in ("rectangular_contour_trapezoidal", "rectangular_contour_gauss"):
            a, b = getattr(pkg, name)(-0.1 - 0.2j, 0.5 + 0.3j, N), getattr(fo, name)(-0.1 - 0.2j, 0.5 + 0.3j, N)
            assert np.abs(a.nodes - b.nodes).max() < 1e-15 and np.abs(a.weights - b.weights).max() < 1e-15
    with pytest.raises(ValueError):
        pkg.RectangularContour(1 + 1j, 0, [], [])
    lam = np.array([0.1, 0.75, 0.3 + 0.69j, 2.0])
    c = pkg.circular_contour_trapezoidal(0.05, 0.7, 8)
    assert pkg.in_contour(lam, c).tolist() == fo.in_contour(lam, fo.circular_contour_trapezoidal(0.05, 0.7, 8)).tolist()


def test_argument_validation_without_gpu(pkg):
    with pytest.raises(ValueError, match="must be square"):
        pkg.gen_feast(np.ones((4, 2), dtype=complex), np.ones((4, 3)), None)
    with pytest.raises(ValueError, match="must match A"):
        pkg.gen_feast(np.ones((5, 2), dtype=complex), np.eye(4), None)
    with pytest.raises(TypeError):                                     # X must be ComplexF64 (src/nlfeast.jl:2)
        pkg.nlfeast(lambda z: np.eye(3), np.ones((3, 2)), 8, 2)
    with pytest.raises(TypeError):
        pkg.nlfeast_moments(lambda z: np.eye(3), np.ones((3, 2)), 8, 2)
    with pytest.raises(TypeError):
        pkg.nlfeast_moments("not callable", np.ones((3, 2), dtype=complex), 8, 2)
    nep = pkg.SplitNEP([np.eye(2), np.ones((2, 2))], [lambda z: z * z, np.exp])
    assert np.allclose(nep(0.5), 0.25 * np.eye(2) + np.exp(0.5) * np.ones((2, 2)))


def test_matrixmarket_io_matches_scipy(pkg, wl, tmp_path):
    """the library's MatrixMarket reader/writer against scipy.io on the reference's data files and on
    symmetric / hermitian / pattern storage; host only."""
    import os
    import scipy.io
    for name in ["butterflyM1.mtx", "quadraticM0.mtx", "system5A1.mtx"]:
        path = os.path.join(wl.GOLDEN_DIR, "data", name)
        M = pkg.mmread(path)
        ref = sp.csc_matrix(scipy.io.mmread(path), dtype=np.complex128)
        assert M.shape == ref.shape and abs(M - ref).max() == 0
    rng = np.random.default_rng(0)
    A = sp.random(30, 30, density=0.2, random_state=1, format="csc") + 1j * sp.random(30, 30, density=0.2, random_state=2, format="csc")
    H = sp.csc_matrix(A + A.conj().T)
    for mat, kw in [(H, dict(symmetry="hermitian")), (sp.csc_matrix((A + A.T).real), dict(symmetry="symmetric")),
                    (sp.csc_matrix(A.real != 0, dtype=np.int64), dict(field="pattern"))]:
        p = str(tmp_path / "m.mtx")
        scipy.io.mmwrite(p, mat, **kw)
        got = pkg.mmread(p)
        assert abs(got - sp.csc_matrix(mat, dtype=np.complex128)).max() < 1e-15
    for mat in [A, sp.csc_matrix(A.real), sp.csc_matrix(np.arange(12).reshape(3, 4))]:
        p = str(tmp_path / "w.mtx")
        pkg.mmwrite(p, mat)                                   # round trip through scipy's reader
        back = sp.csc_matrix(scipy.io.mmread(p), dtype=np.complex128)
        assert back.shape == mat.shape and abs(back - sp.csc_matrix(mat, dtype=np.complex128)).max() < 1e-15
    with pytest.raises(ValueError):
        pkg.mmread(str(tmp_path / "missing.mtx"))


def test_missing_rccl_library_is_a_status_not_a_crash(pkg):
    """feasthip_comm_unique_id on a host whose RCCL cannot be loaded returns FEASTHIP_ECOMM (-8, include/feasthip.h:43) with a message (the round-4 code called
    dlerror() twice and ran strlen(NULL)).  Child process: the library caches the outcome of its first lookup."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import __graft_entry__ as g; p = g.load_package()\n"
            "try:\n    p.comm_unique_id(); print('NOERROR')\n"
            "except p.FeastHipError as e:\n    print('CODE', e.code, str(e))\n") % ROOT
    env = dict(os.environ, FEASTHIP_RCCL_LIB="libdoes_not_exist_feasthip.so.9")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "CODE -8" in out.stdout and "RCCL not found" in out.stdout, out.stdout


_THREAD_INDEPENDENCE_CHILD = r"""
import hashlib, importlib, os, sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as graft
pkg = graft.load_package(); lib = pkg.load_library()
wl = importlib.import_module(graft.PKG_NAME + ".workloads")
h = hashlib.sha256()
for g, st in ((12, 7), (30, 19)):          # 27,000 vertices: the threaded matching / contraction / row-structure paths
    A = sp.csc_matrix(wl.laplacian_3d(g, st)); n = A.shape[0]
    cp = np.ascontiguousarray(A.indptr, dtype=np.int64); ri = np.ascontiguousarray(A.indices, dtype=np.int64)
    stats = np.zeros(8, dtype=np.int64); perm = np.zeros(n, dtype=np.int32); snof = np.zeros(n, dtype=np.int32); fr = np.zeros(4 * n, dtype=np.int32)
    assert lib.feasthip_symbolic_probe(n, cp.ctypes.data, ri.ctypes.data, 0, 0, stats.ctypes.data, perm.ctypes.data, snof.ctypes.data, fr.ctypes.data) == 0
    for a in (stats, perm, snof, fr): h.update(a.tobytes())
rng = np.random.default_rng(5)
for m in (64, 128):
    A = np.asfortranarray(rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m)))
    B = np.asfortranarray(np.eye(m) + 0.2 * (rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m))))
    lam = np.zeros(m, dtype=np.complex128); V = np.zeros((m, m), dtype=np.complex128, order="F")
    assert lib.feasthip_small_eig(m, A.ctypes.data, B.ctypes.data, lam.ctypes.data, V.ctypes.data) == 0
    h.update(lam.tobytes()); h.update(V.tobytes())
print(h.hexdigest())
"""


def test_results_do_not_depend_on_the_host_thread_count():
    """The analysis (range-parallel matching, contraction, FM set-up, row structures of independent subtrees, launch plans) and the
    reduced general eigensolver (eigenvector and right-hand-side loops) run parallel regions on the host pool (csrc/hostpool.cpp).
    The ranges are fixed, only who executes them varies: one thread, three, or the default must give bit-identical orderings,
    supernodes, front tables, eigenvalues and eigenvectors."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for nthreads in ("1", "3", None):
        env = dict(os.environ)
        env.pop("FEASTHIP_HOST_THREADS", None)
        if nthreads:
            env["FEASTHIP_HOST_THREADS"] = nthreads
        r = subprocess.run([sys.executable, "-c", _THREAD_INDEPENDENCE_CHILD, root], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] == outs[2], outs


def test_analysis_in_a_forked_child_has_no_pool_and_the_same_answer(lib, wl):
    """The host pool's threads do not survive fork(): a child process (multiprocessing's fork start method) runs every parallel
    region on its calling thread and must get the parent's ordering."""
    A = wl.laplacian_3d(24, 7)
    s0, p0, sn0, *_ = _probe(lib, A)          # creates the pool in this process
    r, w_ = os.pipe()
    pid = os.fork()
    if pid == 0:
        try:
            s1, p1, sn1, *_ = _probe(lib, A)
            ok = np.array_equal(s0, s1) and np.array_equal(p0, p1) and np.array_equal(sn0, sn1)
            os.write(w_, b"1" if ok else b"0")
        finally:
            os._exit(0)
    os.close(w_)
    out = os.read(r, 1)
    os.close(r)
    os.waitpid(pid, 0)
    assert out == b"1"
