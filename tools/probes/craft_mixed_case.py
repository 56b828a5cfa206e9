import importlib, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, scipy.linalg as sla
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
n = 1200
for seed in range(1, 9):
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((n, n))
    A = (G + G.T) / 2 / np.sqrt(n) + np.diag(np.arange(n) * (2.0 / n))
    lam_all = np.linalg.eigvalsh(A)
    i0 = 300 + 37 * seed
    c = lam_all[i0]; dist = np.sort(np.abs(lam_all - c)); kin = 5
    r = 0.5 * (dist[kin - 1] + dist[kin])
    if dist[kin] - dist[kin - 1] < 0.2 * r: print(seed, "skip"); continue
    con = pkg.circular_contour_trapezoidal(complex(c), float(r), 8)
    info = {}
    lam, V, res = pkg.feast(wl.rand_x0(n, 16, 100 + seed).copy(), A, con, iter=40, eps=1e-9, store=True, mixed_prec=True, info=info)
    print(seed, "inside 5 found", len(lam), "nit", info["nit"], "res", res.max() if len(res) else None, "path", info.get("symmetric"), "growth", round(info.get("growth", 0), 1), flush=True)
