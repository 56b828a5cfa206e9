"""cProfile of the second gen_feast(...; store=true) call on C3 (the first one pays the HIP runtime): where the host-side milliseconds of a
whole user-level call go."""
import cProfile, importlib, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
w = wl.c3_sparse_generalized(46, 19, 64)
def call():
    X = w["X0"].copy()
    return pkg.gen_feast(X, w["A"], w["B"], nodes=16, iter=10, c=w["c"], r=w["r"], eps=1e-10, store=True)
call()
t0 = time.perf_counter(); call(); print("second call %.1f ms" % (1e3 * (time.perf_counter() - t0)))
pr = cProfile.Profile(); pr.enable(); call(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
