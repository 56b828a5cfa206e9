"""(needs a library built with `make PROF=1` in feastsolver.jl_amd/csrc)  Phase times of the diagonal-block kernel INSIDE the C3 factorisation: workgroup (0, 0) of the launch of the root front's group that
starts at column K0 (default 1280: mid-front, the trailing product of the other stream at full width next to it).
python tools/diag_phases_insitu.py [k0]"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package(); lib = pkg.load_library(); wl = importlib.import_module(graft.PKG_NAME + ".workloads")
k0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
w = wl.c3_sparse_generalized(46, 19, 64)
h = pkg.FeastHandle(0); h.set_terms([w["A"], w["B"]], [1.0, -1.0]); h.set_contour(pkg.circular_contour_trapezoidal(w["c"], w["r"], 16)); h.analyze()
lib.feasthip_debug_prof.restype = C.c_int; lib.feasthip_debug_prof.argtypes = [C.c_int, C.c_void_p]
h.factorize()
lib.feasthip_debug_prof(2 + k0, None)
h.factorize()
out = (C.c_ulonglong * 64)(); lib.feasthip_debug_prof(0, C.cast(out, C.c_void_p))
t = np.array(out[:17], dtype=np.float64) * 0.01
print(f"C3 root front, group at column {k0}, LU32={os.environ.get('FEASTHIP_LU32', '2')}: total {t[16] - t[0]:.1f} us (factor_ms {h.stats()['factor_ms']:.1f})")
print(f"  block 0: lu32 {t[4] - t[0]:.1f}")
for c in (1, 2, 3):
    b = 1 + 4 * c
    print(f"  block {c}: substitution {t[b + 1] - t[b - 1]:.1f}  update {t[b + 2] - t[b + 1]:.1f}  lu32 {t[b + 3] - t[b + 2]:.1f}")
h.close()
