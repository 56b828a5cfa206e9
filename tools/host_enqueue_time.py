"""How long the host takes to enqueue one contour step (feasthip_contour_step_begin returns when everything is launched) against
the GPU time of the step: the slack that keeps the step from being launch-bound."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
g = int(sys.argv[1]) if len(sys.argv) > 1 else 46
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w = wl.c3_sparse_generalized(g, 19, 64)
h = pkg.FeastHandle(0); h.set_terms([w["A"], w["B"]], [1.0, -1.0]); h.set_contour(pkg.circular_contour_trapezoidal(w["c"], w["r"], 16))
h.set_node_shard(0, world); h.analyze(); h.factorize(); h.set_x(w["X0"])
for _ in range(3): h.rr_step(); h.contour_step()
tb, te, tr = [], [], []
for _ in range(10):
    t0 = time.perf_counter(); h.rr_step(); t1 = time.perf_counter(); h.contour_step_begin(False, 1); t2 = time.perf_counter(); h.contour_step_end(); t3 = time.perf_counter()
    tr.append(t1 - t0); tb.append(t2 - t1); te.append(t3 - t2)
st = h.stats()
print(f"world {world}: rr {np.median(tr)*1e3:.2f} ms | contour step: host enqueue {np.median(tb)*1e3:.2f} ms (max {max(tb)*1e3:.2f}), then wait {np.median(te)*1e3:.2f} ms; GPU events {st['solve_ms']:.2f} ms")
h.close()
