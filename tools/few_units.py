import importlib, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
w = wl.c3_sparse_generalized(46, 19, 64)
for world in (8, 4):
    h = pkg.FeastHandle(0); h.set_terms([w["A"], w["B"]], [1.0, -1.0]); h.set_contour(pkg.circular_contour_trapezoidal(w["c"], w["r"], 16))
    h.set_node_shard(0, world); h.analyze(); h.factorize(); h.set_x(w["X0"])
    for _ in range(3): h.rr_step(); h.contour_step()
    ts = []
    for _ in range(6):
        h.rr_step(); t1 = time.perf_counter(); h.contour_step(); ts.append(time.perf_counter() - t1)
    lam, res, ins = h.rr_step()
    print(f"world {world}: contour {np.median(ts)*1e3:.2f} ms  res {res[ins].max():.1e}", flush=True)
    h.close()
