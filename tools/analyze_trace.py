"""Host-side analysis phases (FEASTHIP_ANALYZE_TRACE=1) and the low-node-count contour step (per-rank load at N GPUs)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
w = wl.c3_sparse_generalized(46, 19, 64)
for world in (1, 2, 4, 8):
    h = pkg.FeastHandle(0); h.set_terms([w["A"], w["B"]], [1.0, -1.0]); h.set_contour(pkg.circular_contour_trapezoidal(w["c"], w["r"], 16))
    h.set_node_shard(0, world)
    t0 = time.perf_counter(); h.analyze(); ta = time.perf_counter() - t0
    h.factorize(); st = h.stats(); h.set_x(w["X0"])
    for _ in range(3): h.rr_step(); h.contour_step()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); h.rr_step(); t1 = time.perf_counter(); h.contour_step(); t2 = time.perf_counter(); ts.append((t1 - t0, t2 - t1))
    print(f"world {world}: units {st['factor_units']} nodes {st['local_nodes']} analyze {ta*1e3:.0f} ms factor {st['factor_ms']:.1f} ms  rr {np.median([a for a,b in ts])*1e3:.2f} ms  contour {np.median([b for a,b in ts])*1e3:.2f} ms", flush=True)
    h.close()
