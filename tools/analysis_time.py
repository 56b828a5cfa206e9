"""Host-only: wall time of the symbolic analysis of the C3 pattern (median of N runs in fresh processes is what to quote; this prints every run of one process).
python tools/analysis_time.py [grid] [runs]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import __graft_entry__ as graft
pkg = graft.load_package(); lib = pkg.load_library()
wl = importlib.import_module(graft.PKG_NAME + ".workloads")
g = int(sys.argv[1]) if len(sys.argv) > 1 else 46
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
w = wl.c3_sparse_generalized(g, 19, 64)
A = sp.csc_matrix(w["A"]); n = A.shape[0]
cp = np.ascontiguousarray(A.indptr, dtype=np.int64); ri = np.ascontiguousarray(A.indices, dtype=np.int64)
ts = []
for r in range(runs):
    stats = np.zeros(8, dtype=np.int64); perm = np.zeros(n, dtype=np.int32); snof = np.zeros(n, dtype=np.int32); fr = np.zeros(4 * n, dtype=np.int32)
    t0 = time.perf_counter()
    rc = lib.feasthip_symbolic_probe(n, cp.ctypes.data, ri.ctypes.data, 0, 0, stats.ctypes.data, perm.ctypes.data, snof.ctypes.data, fr.ctypes.data)
    ts.append((time.perf_counter() - t0) * 1e3)
print("rc", rc, "analysis ms", " ".join(f"{t:.1f}" for t in ts), "| supernodes %d nnz_L %d max front %d flops %.4g" % (stats[0], stats[1], stats[4], stats[7]), os.environ.get("FEASTHIP_ND_PARMIN", ""), os.environ.get("FEASTHIP_ND_TRIALS2_MIN", ""), os.environ.get("FEASTHIP_HOST_THREADS", ""))
