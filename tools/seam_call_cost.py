import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package()
rng = np.random.default_rng(0)
for n in (64, 512):
    A = np.asfortranarray(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    R = np.asfortranarray(rng.standard_normal((n, 16)) + 1j * rng.standard_normal((n, 16)))
    Y = np.zeros_like(R)
    ts = []
    for rep in range(12):
        t0 = time.perf_counter(); F = pkg.lu(A); t1 = time.perf_counter(); pkg.ldiv(Y, F, R); t2 = time.perf_counter(); pkg.finalize(F); t3 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1, t3 - t2))
    ts = np.array(ts[2:]) * 1e3
    print(n, "lu ms %.2f  ldiv ms %.2f  finalize ms %.2f" % tuple(np.median(ts, axis=0)))
t0 = time.perf_counter()
hs = [pkg.FeastHandle(0) for _ in range(5)]
print("FeastHandle create ms", (time.perf_counter() - t0) / 5 * 1e3)
t0 = time.perf_counter()
for h in hs: h.close()
print("FeastHandle close ms", (time.perf_counter() - t0) / 5 * 1e3)
