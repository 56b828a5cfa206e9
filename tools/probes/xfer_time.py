"""Host <-> device transfer of the iteration block at the boundary (feasthip_set_x / feasthip_get_block: n x m0 complex128 from / to the caller's
pageable memory): milliseconds and GB/s on C3's shape.  python tools/probes/xfer_time.py [grid=46] [m0=64]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
g = int(sys.argv[1]) if len(sys.argv) > 1 else 46
m0 = int(sys.argv[2]) if len(sys.argv) > 2 else 64
w = wl.c3_sparse_generalized(g, 19, m0)
h = pkg.FeastHandle(0)
h.set_rr_only(True)
h.set_terms([w["A"], w["B"]], [1.0, -1.0])
X = np.asfortranarray(w["X0"], dtype=np.complex128)
gb = X.nbytes / 1e9
ts, tg = [], []
for i in range(6):
    t0 = time.perf_counter(); h.set_x(X); t1 = time.perf_counter(); Y = h.get_block(0); t2 = time.perf_counter()
    ts.append(t1 - t0); tg.append(t2 - t1)
assert np.array_equal(X, Y)
# into memory the caller has touched already (feast!'s X is overwritten in place; the Python get_block above allocates its result)
api = importlib.import_module(graft.PKG_NAME + ".api")
out = np.zeros_like(X)          # touched pages
tp = []
for i in range(6):
    t0 = time.perf_counter(); h._ck(h.lib.feasthip_get_block(h.h, 0, api._vp(out))); tp.append(time.perf_counter() - t0)
assert np.array_equal(X, out)
print("get_block into touched memory ms " + " ".join(f"{1e3 * t:.2f}" for t in tp) + f" ({gb / min(tp):.1f} GB/s)")
print(f"n = {X.shape[0]}, m0 = {m0}: {gb * 1e3:.0f} MB | set_x ms " + " ".join(f"{1e3 * t:.2f}" for t in ts) + f" ({gb / min(ts):.1f} GB/s) | get_block ms " + " ".join(f"{1e3 * t:.2f}" for t in tg) + f" ({gb / min(tg):.1f} GB/s)")
h.close()
