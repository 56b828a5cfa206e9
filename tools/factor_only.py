"""Numeric factorisations alone, for kernel traces: python tools/factor_only.py c2|c3|c3h|c5 [reps]
c2: 16 dense 4096^2 LUs (pivoted); c3: 8 units of the 46^3 symmetric pencil; c3h: the Hermitian variant (adjoint fold); c5: 32 LUs of
the 2000^2 companion pencil of system5.  Prints factor_ms of every repetition."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
case = sys.argv[1] if len(sys.argv) > 1 else "c3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
h = pkg.FeastHandle(0)
if case == "c2":
    n = 4096
    w = wl.c2_dense_generalized(n, 64, radius=float(np.sqrt(32.0 / n)) * 1.03)
    h.set_terms([w["A"], w["B"]], [1.0, -1.0]); h.set_contour(pkg.circular_contour_trapezoidal(w["c"], w["r"], 16))
elif case == "c5":
    w = wl.c5_system5_companion()
    h.set_terms([w["A"], w["B"]], [1.0, -1.0]); h.set_contour(pkg.circular_contour_trapezoidal(w["c"], w["r"], w["nodes"]))
else:
    g = int(os.environ.get("GRID", "46"))
    w = wl.c3_sparse_generalized(g, 19, 64)
    if case == "c3h": w["A"] = wl.hermitian_phases(w["A"], seed=12)
    h.set_terms([w["A"], w["B"]], [1.0, -1.0]); h.set_contour(pkg.circular_contour_trapezoidal(w["c"], w["r"], 16))
h.analyze()
for r in range(reps):
    print(f"--- factorisation {r}", file=sys.stderr, flush=True)
    warn = h.factorize(); st = h.stats()
    print(f"{case} rep {r}: factor_ms {st['factor_ms']:.2f} units {st['factor_units']} TFLOP/s {st['flops_factor'] * st['factor_units'] / (st['factor_ms'] * 1e-3) / 1e12:.1f} warn {warn}", flush=True)
h.close()
