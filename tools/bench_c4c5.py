"""C4 (butterfly quartic NEP through nlfeast, 16 nodes, m0 = 32) and C5 (system5 quadratic, companion-linearised 2000 x 2000 dense
pencil, gen_feast, 32 nodes, m0 = 128 -- SURVEY 8d substitution for the 15 x 15 literal data) on one MI355X: wall time of the whole
call (hand-over, factorisations, iterations), second call of each (the first pays the process's one-off costs)."""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
out = {}
w = wl.c4_butterfly()
T = pkg.PolynomialNEP(w["coeffs"])
for rep in range(3):
    info = {}
    t0 = time.perf_counter()
    lam, X, res = pkg.nlfeast(T, w["X0"].copy(), 16, 10, c=w["c"], r=w["r"], eps=w["eps"], spurious=w["spurious"], info=info)
    dt = time.perf_counter() - t0
ins = np.abs(lam - w["c"]) <= w["r"]
out["C4 butterfly nlfeast"] = {"call_s": dt, "iterations": int(info["nit"]), "inside": int(ins.sum()), "max_res_inside": float(res[ins].max()),
                               "ms_per_iteration": 1e3 * dt / max(1, int(info["nit"]))}
w = wl.c5_system5_companion()
con = pkg.circular_contour_trapezoidal(w["c"], w["r"], w["nodes"])
for rep in range(2):
    info = {}
    t0 = time.perf_counter()
    lam, X, res = pkg.gen_feast(w["X0"].copy(), w["A"], w["B"], con, iter=w["iter"], eps=w["eps"], store=True, info=info)
    dt = time.perf_counter() - t0
out["C5 system5 companion gen_feast (2000 x 2000, 32 nodes, m0 = 128)"] = {"call_s": dt, "iterations": int(info["nit"]), "inside": int(len(lam)),
                                                                        "max_res": float(res.max()) if len(res) else None}
for k in ("factor_ms", "solve_ms", "rr_ms", "analyze_seconds"):
    if k in info: out["C5 system5 companion gen_feast (2000 x 2000, 32 nodes, m0 = 128)"][k] = info[k]
print(json.dumps(out))
