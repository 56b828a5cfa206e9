"""Would two half-batches of units factorised CONCURRENTLY (their latency chains overlapping each other's products) beat one batch in lockstep?
Two handles with half the nodes each, factorised from two host threads at once, against one handle with all of them.
python tools/two_batches.py c2|c5|c3"""
import importlib, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
case = sys.argv[1] if len(sys.argv) > 1 else "c2"
if case == "c2":
    w = wl.c2_dense_generalized(4096, 64, radius=float(np.sqrt(32.0 / 4096)) * 1.03); nodes = 16
elif case == "c5":
    w = wl.c5_system5_companion(); nodes = w["nodes"]
else:
    w = wl.c3_sparse_generalized(46, 19, 64); nodes = 16
con = pkg.circular_contour_trapezoidal(w["c"], w["r"], nodes)
def mk(shard):
    h = pkg.FeastHandle(0); h.set_terms([w["A"], w["B"]], [1.0, -1.0]); h.set_contour(con)
    if shard: h.set_node_shard(*shard)
    h.analyze(); h.factorize(); return h
h1 = mk(None)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); h1.factorize(); ts.append(time.perf_counter() - t0)
print(f"{case}: one handle, {h1.stats()['factor_units']} units: wall {min(ts) * 1e3:.1f} ms (factor_ms {h1.stats()['factor_ms']:.1f})")
h1.close()
hs = [mk((r, 2)) for r in range(2)]
ts = []
for _ in range(3):
    bar = threading.Barrier(3)
    def run(h):
        bar.wait(); h.factorize()
    th = [threading.Thread(target=run, args=(h,)) for h in hs]
    for t in th: t.start()
    bar.wait(); t0 = time.perf_counter()
    for t in th: t.join()
    ts.append(time.perf_counter() - t0)
print(f"{case}: two handles x {hs[0].stats()['factor_units']} units, concurrent: wall {min(ts) * 1e3:.1f} ms (factor_ms {hs[0].stats()['factor_ms']:.1f} / {hs[1].stats()['factor_ms']:.1f})")
for h in hs: h.close()
