import importlib, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
w = wl.c3_sparse_generalized(46, 19, 64)
for world in (1, 8):
    h = pkg.FeastHandle(0); h.set_terms([w["A"], w["B"]], [1.0, -1.0]); h.set_contour(pkg.circular_contour_trapezoidal(w["c"], w["r"], 16))
    h.set_node_shard(0, world); h.analyze(); h.factorize(); h.set_x(w["X0"])
    for _ in range(3): h.rr_step(); h.contour_step()
    for mode in ("after contour", "after 20 ms idle", "back to back"):
        tr = []
        for _ in range(6):
            if mode == "after contour": h.contour_step()
            elif mode == "after 20 ms idle": time.sleep(0.02)
            t0 = time.perf_counter(); h.rr_step(); tr.append(time.perf_counter() - t0)
        print(f"world {world}, rr_step {mode}: {np.median(tr)*1e3:.2f} ms (event {h.stats()['rr_ms']:.2f})", flush=True)
    h.close()
