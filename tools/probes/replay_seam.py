"""Per-node check of a dumped fuzz case at the plug-in seam: T(z_k) = A - z_k I factorised in complex128 and complex64, 16 right-hand sides,
relative error against numpy.  python tools/probes/replay_seam.py case.npz"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package()
d = np.load(sys.argv[1])
A = d["A"]; n = A.shape[0]
con = pkg.circular_contour_trapezoidal(complex(d["c"]), float(d["r"]), int(d["K"]))
R = d["X0"].astype(np.complex128)
for z in con.nodes:
    C = (A - z * np.eye(n)).astype(np.complex128)
    ref = np.linalg.solve(C, R)
    out = []
    for dt in (np.complex128, np.complex64):
        F = pkg.lu(C.astype(dt)); Y = np.zeros_like(R, order="F"); pkg.ldiv(Y, F, R)
        out.append((np.abs(Y - ref).max() / np.abs(ref).max(), F.path, round(F.growth, 2))); pkg.finalize(F)
    print(f"z = {z:.5f}: cond {np.linalg.cond(C):.2e} | c128 rel err {out[0][0]:.1e} (path {out[0][1]}, growth {out[0][2]}) | c64 rel err {out[1][0]:.1e} (path {out[1][1]}, growth {out[1][2]})")
