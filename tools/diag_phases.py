"""(needs a library built with `make PROF=1` in feastsolver.jl_amd/csrc)  Phase times of the diagonal-block kernel of the elimination without interchanges (workgroup (0, 0) of its LAST launch), dense symmetric
n x n, one unit, nothing else on the GPU: python tools/diag_phases.py [n]"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package(); lib = pkg.load_library()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rng = np.random.default_rng(1)
S = rng.standard_normal((n, n)); S = S + S.T
G = rng.standard_normal((n, n)); Cm = (S + 1j * (G @ G.T / n + 0.5 * np.eye(n))).astype(np.complex128)
lib.feasthip_debug_prof.restype = C.c_int
lib.feasthip_debug_prof.argtypes = [C.c_int, C.c_void_p]
F = pkg.lu(Cm); pkg.finalize(F)                      # warm
lib.feasthip_debug_prof(1, None)
F = pkg.lu(Cm)
out = (C.c_ulonglong * 64)()
lib.feasthip_debug_prof(0, C.cast(out, C.c_void_p))
pkg.finalize(F)
t = np.array(out[:17], dtype=np.float64) * 0.01     # us
u = np.array(out[40:44], dtype=np.float64) * 0.01
print(f"  last 32 x 32 block: load + first publish {u[1] - u[0]:.1f}, 32 steps {u[2] - u[1]:.1f}, outputs {u[3] - u[2]:.1f} us")
print(f"n={n} LU32={os.environ.get('FEASTHIP_LU32', '2')}: total {t[16] - t[0]:.1f} us")
print(f"  block 0: lu32 {t[4] - t[0]:.1f}")
for c in (1, 2, 3):
    b = 1 + 4 * c
    print(f"  block {c}: substitution {t[b + 1] - t[b - 1]:.1f}  update {t[b + 2] - t[b + 1]:.1f}  lu32 {t[b + 3] - t[b + 2]:.1f}")
