import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))) if "__file__" in dir() else "/root/repo")
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
t0 = time.perf_counter(); w = wl.c3_sparse_generalized(46, 19, 64); print("workload build s", time.perf_counter() - t0)
for rep in range(3):
    t0 = time.perf_counter(); h = pkg.FeastHandle(0); t1 = time.perf_counter()
    h.set_terms([w["A"], w["B"]], [1.0, -1.0]); t2 = time.perf_counter()
    h.set_contour(pkg.circular_contour_trapezoidal(w["c"], w["r"], 16)); t3 = time.perf_counter()
    h.analyze(); t4 = time.perf_counter()
    h.factorize(); t5 = time.perf_counter()
    h.set_x(w["X0"]); t6 = time.perf_counter()
    lam, res, ins = h.rr_step(); t7 = time.perf_counter()
    h.contour_step(); t8 = time.perf_counter()
    X = h.get_block(0); t9 = time.perf_counter()
    h.close(); t10 = time.perf_counter()
    print("create %.1f set_terms %.1f contour %.2f analyze %.1f factorize %.1f set_x %.2f rr %.2f contour_step %.2f get_block %.2f close %.2f  (ms)" %
          tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t8 - t7, t9 - t8, t10 - t9)))
