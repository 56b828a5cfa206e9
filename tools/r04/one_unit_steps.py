"""A few contour steps with one factorisation unit on the GPU (C3, set_node_shard(0, 8)): the subject of a rocprofv3 kernel trace that
shows the idle time between the 80-odd dependent launches of a step."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
w = wl.c3_sparse_generalized(46, 19, 64)
h = pkg.FeastHandle(0); h.set_terms([w["A"], w["B"]], [1.0, -1.0]); h.set_contour(pkg.circular_contour_trapezoidal(w["c"], w["r"], 16))
h.set_node_shard(0, 8); h.analyze(); h.factorize(); h.set_x(w["X0"])
for _ in range(6): h.rr_step(); h.contour_step()
h.close()
