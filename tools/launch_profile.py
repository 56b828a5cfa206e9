"""Per-launch efficiency of the block-substitution kernel (HIP events around every launch)."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
g = int(sys.argv[1]) if len(sys.argv) > 1 else 46
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
w = wl.c3_sparse_generalized(g, 19, 64)
h = pkg.FeastHandle(0); h.set_terms([w["A"], w["B"]], [1.0, -1.0]); h.set_contour(pkg.circular_contour_trapezoidal(w["c"], w["r"], K))
h.analyze(); h.factorize(); h.set_x(w["X0"])
for _ in range(2): h.rr_step(); h.contour_step()
h.rr_step(); pr = h.profile_solve(); d = h.last_profile_detail
ap = d[d[:, 0] == 3]
tot_ms = ap[:, 1].sum(); tot_fl = ap[:, 3].sum()
print("apply launches", len(ap), "ms", tot_ms, "TF", tot_fl / tot_ms / 1e9)
order = np.argsort(-ap[:, 1])
print("   ms     tiles    GF      TF/s   (negative tiles = gather kernel)")
for i in order[:30]:
    print(f"{ap[i,1]:7.3f} {int(ap[i,4]):7d} {ap[i,3]/1e9:8.2f} {ap[i,3]/ap[i,1]/1e9:7.1f}")
small = ap[ap[:, 1] < 0.05]
cl = d[d[:, 0] != 3]
print("other launches", len(cl), "ms", cl[:, 1].sum())
print("all launches sorted by time (top 40):")
for i in np.argsort(-d[:, 1])[:40]: print(f"  class {int(d[i,0])} {d[i,1]*1e3:8.1f} us tiles {int(d[i,4])}")
print("launches < 50us:", len(small), "total ms", small[:, 1].sum(), "GF", small[:, 3].sum() / 1e9)
print("launch sequence (class us tiles GB TB/s TF/s):")
for i in range(len(d)):
    c, ms, by, fl, ti, dep = d[i]
    print(f"  {i:3d} d{int(dep):2d} c{int(c)} {ms*1e3:8.1f}us tiles {int(ti):6d}  {by/1e9:7.3f} GB {by/ms/1e9 if ms>0 else 0:7.2f} TB/s {fl/ms/1e9 if ms>0 else 0:6.1f} TF")
