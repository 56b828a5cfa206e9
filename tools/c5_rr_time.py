"""Rayleigh-Ritz step of C5 (2000 x 2000 companion pencil, m0 = 128, non-Hermitian: host eigensolver) timed alone: python tools/c5_rr_time.py [reps]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w = wl.c5_system5_companion()
h = pkg.FeastHandle(0); h.set_terms([w["A"], w["B"]], [1.0, -1.0]); h.set_contour(pkg.circular_contour_trapezoidal(w["c"], w["r"], w["nodes"]))
h.analyze(); h.factorize(); h.set_x(w["X0"])
ts = []
for r in range(reps):
    t0 = time.perf_counter(); h.rr_step(); ts.append((time.perf_counter() - t0) * 1e3); h.contour_step()
print("C5 rr_step ms:", " ".join(f"{t:.2f}" for t in ts), "| median %.2f" % np.median(ts[1:]), os.environ.get("FEASTHIP_EIG_ZASYNC", ""), os.environ.get("FEASTHIP_HOST_THREADS", ""))
h.close()
