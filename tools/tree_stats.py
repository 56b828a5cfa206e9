"""Host-only: per-level statistics of the elimination tree the symbolic analysis builds (no GPU needed)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import __graft_entry__ as graft
pkg = graft.load_package(); lib = pkg.load_library()
wl = importlib.import_module(graft.PKG_NAME + ".workloads")
g = int(sys.argv[1]) if len(sys.argv) > 1 else 46
w = wl.c3_sparse_generalized(g, 19, 64)
A = sp.csc_matrix(w["A"]); n = A.shape[0]
cp = np.ascontiguousarray(A.indptr, dtype=np.int64); ri = np.ascontiguousarray(A.indices, dtype=np.int64)
stats = np.zeros(8, dtype=np.int64); perm = np.zeros(n, dtype=np.int32); snof = np.zeros(n, dtype=np.int32)
fr = np.zeros(4 * n, dtype=np.int32)
import time; t0 = time.time()
rc = lib.feasthip_symbolic_probe(n, cp.ctypes.data, ri.ctypes.data, 0, 0, stats.ctypes.data, perm.ctypes.data, snof.ctypes.data, fr.ctypes.data)
print("rc", rc, "analysis s", time.time() - t0, "stats", stats)
ns = int(stats[0]); fr = fr[:4 * ns].reshape(ns, 4)
p, u, d = fr[:, 0].astype(np.int64), fr[:, 1].astype(np.int64), fr[:, 2]
print("depth  nfronts  pmax  pmean  umax  factor_entries(M)  frac   vecrows(k)  in-block-steps")
tot = (p * p + 2 * p * u).sum()
for dd in range(d.max() + 1):
    m = d == dd
    fe = (p[m] * p[m] + 2 * p[m] * u[m]).sum()
    print(f"{dd:4d} {m.sum():7d} {p[m].max():6d} {p[m].mean():7.1f} {u[m].max():6d} {fe/1e6:10.2f} {fe/tot:8.3f} {(p[m]+u[m]).sum()/1e3:9.1f} {(p[m].max()+127)//128 - 1:4d}")
print("total factor entries (M)", tot / 1e6)
np.save("/tmp/fronts.npy", fr)
