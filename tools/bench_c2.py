"""C2 (BASELINE configs[1]): dense generalized complex non-Hermitian 4096 x 4096, 16 nodes, m0 = 64, one MI355X.
Prints setup (16 dense LUs), per-iteration times and the converged result."""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m0 = 64
w = wl.c2_dense_generalized(n, m0, radius=float(np.sqrt(32.0 / n)) * 1.03)
h = pkg.FeastHandle(0)
t0 = time.perf_counter(); h.set_terms([w["A"], w["B"]], [1.0, -1.0]); t_up = time.perf_counter() - t0
h.set_contour(pkg.circular_contour_trapezoidal(w["c"], w["r"], 16))
h.analyze(); warn = h.factorize(); st = h.stats()
h.set_x(w["X0"])
hist = []
t0 = time.perf_counter()
for it in range(12):
    lam, res, ins = h.rr_step()
    hist.append((it, int(ins.sum()), float(res[ins].max()) if ins.any() else None))
    if ins.any() and res[ins].max() < 1e-10: break
    h.contour_step()
dt = time.perf_counter() - t0
st2 = h.stats(); prof = h.profile_solve()
X = h.get_block(0)
r_true = np.linalg.norm(w["A"] @ X[:, ins] - (w["B"] @ X[:, ins]) * lam[ins][None, :], axis=0).max() if ins.any() else None
print(json.dumps({"config": f"C2 dense generalized n={n} m0={m0} 16 nodes", "upload_s": t_up, "factor_ms": st["factor_ms"],
                  "factor_TFLOPs": st["flops_factor"] * 16 / (st["factor_ms"] * 1e-3) / 1e12, "perturbed": int(warn),
                  "iterations": len(hist), "history": hist, "loop_s": dt, "contour_step_ms": st2["solve_ms"], "rr_ms": st2["rr_ms"],
                  "solve_TFLOPs": st2["solve_flops"] / (st2["solve_ms"] * 1e-3) / 1e12, "true_residual": r_true,
                  "kernel_classes_ms": {k: v["ms"] for k, v in prof.items()}}))
