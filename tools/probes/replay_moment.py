"""One Rayleigh-Ritz step + one contour step of a dumped fuzz case in complex128 and with complex64 sweeps: relative difference of the moments,
column by column.  python tools/probes/replay_moment.py case.npz"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package()
d = np.load(sys.argv[1])
A = d["A"]; n = A.shape[0]
con = pkg.circular_contour_trapezoidal(complex(d["c"]), float(d["r"]), int(d["K"]))
Q = {}
for mixed in (False, True):
    h = pkg.FeastHandle(0)
    h.set_terms([A, np.eye(n)], [1.0, -1.0]); h.set_contour(con)
    if mixed: h.set_precision(True)
    h.factorize(); h.set_x(d["X0"])
    lam, res, ins = h.rr_step()
    h.contour_step()
    Q[mixed] = h.get_block(2).copy()
    st = h.stats(); h.close()
err = np.abs(Q[True] - Q[False]).max(axis=0) / np.abs(Q[False]).max(axis=0)
print("folded", st["folded"], "symmetric", st["symmetric"], "growth", st["growth"], "| per-column relative difference of the moment (complex64 sweeps vs complex128):")
print(" ".join(f"{e:.1e}" for e in err))
print("Ritz values of the start block inside:", int(ins.sum()), "| moment norm", np.abs(Q[False]).max())
