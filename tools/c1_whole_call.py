"""C1 (BASELINE configs[0], the reference's own CPU-runnable case: dense Hermitian 512 x 512, feast!, 8 nodes, m0 = 16): wall time
of the user-level call on the GPU and of the CPU oracle (numpy/scipy restatement of src/feast.jl:10-80) on this box's host cores."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
fo = graft.load_oracle()
w = wl.c1_dense_hermitian()
con = pkg.circular_contour_trapezoidal(w["c"], w["r"], w["nodes"])
for store in (True, False):
    for rep in range(3):
        info = {}
        t0 = time.perf_counter()
        lam, V, res = pkg.feast(np.asfortranarray(w["X0"].copy()), w["A"], con, iter=w["iter"], eps=w["eps"], store=store, info=info)
        dt = time.perf_counter() - t0
    print(f"GPU  feast(store={store}): {dt * 1e3:.1f} ms, {len(lam)} eigenpairs, {info['nit']} iterations, max residual {res.max():.1e}")
for store in (True, False):
    for rep in range(2):
        t0 = time.perf_counter()
        out = fo.feast(w["X0"].copy(), w["A"], fo.circular_contour_trapezoidal(w["c"], w["r"], w["nodes"]), iter=w["iter"], eps=w["eps"], store=store)
        dt = time.perf_counter() - t0
    print(f"CPU oracle feast(store={store}): {dt * 1e3:.1f} ms, {len(out[0])} eigenpairs (host threads: {os.cpu_count()})")
