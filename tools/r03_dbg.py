import importlib, os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
    import __graft_entry__ as graft
    pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
    for g in [int(x) for x in sys.argv[2].split(",")]:
        A = wl.laplacian_3d(g, 19); n = A.shape[0]
        rng = np.random.default_rng(1)
        Cm = sp.csc_matrix(A.astype(np.complex128) - (0.17 + 0.17j) * sp.diags(rng.uniform(0.5, 1.5, n)))
        R = np.asfortranarray(rng.standard_normal((n, 8)) + 1j * rng.standard_normal((n, 8)))
        F = pkg.lu(Cm); Y = np.zeros_like(R); pkg.ldiv(Y, F, R)
        rr = np.linalg.norm(Cm @ Y - R) / (spla.norm(Cm) * np.linalg.norm(Y))
        print(f"  g={g} relres={rr:.2e}" + ("   <<<<<< BAD" if rr > 1e-14 else ""), flush=True)
        pkg.finalize(F)
else:
    grids = "22,24,26,22,24,26,22,24,26,22,24,26"
    for env in ({}, {"FEASTHIP_LOOKAHEAD": "0"}, {"FEASTHIP_LOOKAHEAD": "0", "FEASTHIP_FIN_ASYNC": "0"}, {"FEASTHIP_NOPIV_GROUP": "0"}):
        print("==", env, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", grids], env=dict(os.environ, **env))
