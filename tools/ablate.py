"""Timing-only sweep for the ablation builds of k_apply (tools/ablate.sh); results are not checked."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
g = int(sys.argv[1]) if len(sys.argv) > 1 else 46
w = wl.c3_sparse_generalized(g, 19, 64)
h = pkg.FeastHandle(0); h.set_terms([w["A"], w["B"]], [1.0, -1.0]); h.set_contour(pkg.circular_contour_trapezoidal(w["c"], w["r"], 16))
h.analyze(); h.factorize(); h.set_x(w["X0"]); h.rr_step()
for _ in range(2):
    pr = h.profile_solve()
d = h.last_profile_detail
ap = d[d[:, 0] == 3]
print(sys.argv[2] if len(sys.argv) > 2 else "", "apply ms %.3f  TF %.1f" % (ap[:, 1].sum(), ap[:, 3].sum() / ap[:, 1].sum() / 1e9),
      "big launches (>0.5ms) ms %.3f TF %.1f" % (ap[ap[:, 1] > 0.5][:, 1].sum(), ap[ap[:, 1] > 0.5][:, 3].sum() / ap[ap[:, 1] > 0.5][:, 1].sum() / 1e9))
