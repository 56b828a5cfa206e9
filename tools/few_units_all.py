"""Per-rank load of the node-sharded run, measured on ONE GPU: the contour step, the factorisation and the replicated Rayleigh-Ritz step
with the 8 / 4 / 2 / 1 factorisation units a rank holds at N = 1 / 2 / 4 / 8 (C3, 16 nodes, m0 = 64); with the conjugate fold (default:
units are node pairs, 128 columns wide) and without it (single nodes, 64 columns: twice the factorisations, faster sweeps at few units).
NOTE on the Rayleigh-Ritz column: a shard's moment is a PARTIAL sum here (no other ranks to add theirs), so the subspace of this emulation
does not converge and the device eigensolver runs cold (5 Jacobi sweeps, 0.5 ms) at N > 1; with the all-reduce every rank sees the N = 1
moment and the N = 1 time."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
w = wl.c3_sparse_generalized(46, 19, 64)
for fold in (True, False):
    for world in (1, 2, 4, 8):
        h = pkg.FeastHandle(0); h.set_terms([w["A"], w["B"]], [1.0, -1.0]); h.set_contour(pkg.circular_contour_trapezoidal(w["c"], w["r"], 16))
        h.set_conjugate_fold(fold)
        h.set_node_shard(0, world); h.analyze(); h.factorize(); h.set_x(w["X0"])
        for _ in range(3): h.rr_step(); h.contour_step()
        ts, tr = [], []
        for _ in range(6):
            t0 = time.perf_counter(); h.rr_step(); t1 = time.perf_counter(); h.contour_step(); ts.append(time.perf_counter() - t1); tr.append(t1 - t0)
        st = h.stats()
        print(f"N = {world} ranks, fold {'on ' if fold else 'off'}: {st['factor_units']} units ({st['local_nodes']} nodes) per rank: contour step {np.median(ts)*1e3:6.2f} ms, "
              f"Rayleigh-Ritz {np.median(tr)*1e3:5.2f} ms (replicated), factorisation {st['factor_ms']:6.1f} ms", flush=True)
        h.close()
