"""Dense symmetric solves under the four combinations of the round-5 group kernels (FEASTHIP_DIAG2 / FEASTHIP_LSCALE2): max difference to the
round-3 kernels and the residual, per size."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package()
for n in [int(a) for a in sys.argv[1:]] or [20, 33, 40, 64, 65, 96, 100, 128, 129, 300]:
    rng = np.random.default_rng(n)
    S = rng.standard_normal((n, n)); S = S + S.T
    G = rng.standard_normal((n, n)); Cm = (S + 1j * (G @ G.T / n + 0.5 * np.eye(n))).astype(np.complex128)
    R = np.asfortranarray(rng.standard_normal((n, 5)) + 1j * rng.standard_normal((n, 5)))
    sols = {}
    os.environ["FEASTHIP_LU32"] = "1"
    for d2, l2 in [("0", "0"), ("1", "0"), ("0", "1"), ("L2", "1")]:
        os.environ["FEASTHIP_DIAG2"], os.environ["FEASTHIP_LSCALE2"] = ("0" if d2 == "L2" else d2), l2
        if d2 == "L2": os.environ["FEASTHIP_LU32"] = "3"
        F = pkg.lu(Cm); Y = np.zeros((n, 5), dtype=np.complex128, order="F"); pkg.ldiv(Y, F, R); pkg.finalize(F)
        sols[(d2, l2)] = Y
    ref = sols[("0", "0")]
    print(n, {k: f"{np.abs(v - ref).max():.2e}" for k, v in sols.items()}, "residual", f"{np.linalg.norm(Cm @ sols[('L2', '1')] - R) / (np.linalg.norm(Cm) * np.linalg.norm(ref)):.2e}", flush=True)
