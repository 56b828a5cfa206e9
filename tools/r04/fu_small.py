import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
w = wl.c3_sparse_generalized(46, 19, 64)
out = []
for fold, world in ((True, 2), (True, 4), (False, 8), (True, 8)):
    h = pkg.FeastHandle(0); h.set_terms([w["A"], w["B"]], [1.0, -1.0]); h.set_contour(pkg.circular_contour_trapezoidal(w["c"], w["r"], 16))
    h.set_conjugate_fold(fold)
    h.set_node_shard(0, world); h.analyze(); h.factorize(); h.set_x(w["X0"])
    for _ in range(3): h.rr_step(); h.contour_step()
    ts = []
    for _ in range(8):
        h.rr_step(); t1 = time.perf_counter(); h.contour_step(); ts.append(time.perf_counter() - t1)
    out.append(f"{np.median(ts)*1e3:6.2f}")
    h.close()
print(os.environ.get("FEASTHIP_XCD_C0"), os.environ.get("FEASTHIP_XCD_THR"), " ".join(out), flush=True)
