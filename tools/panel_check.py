"""Pivoted dense LU under FEASTHIP_PANEL_LL=1 / 0 (left-looking panel kernel with implicit interchanges vs the round-1 kernel): difference of the
solutions and residuals per size."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package()
for n in [int(a) for a in sys.argv[1:]] or [5, 31, 33, 64, 100, 129, 257, 700, 1100, 2100, 3100]:
    rng = np.random.default_rng(n)
    Cm = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))).astype(np.complex128)
    if n == 100: Cm = np.round(Cm * 3)            # many exact ties in the pivot search
    R = np.asfortranarray(rng.standard_normal((n, 5)) + 1j * rng.standard_normal((n, 5)))
    sols = {}
    for m in ("0", "1"):
        os.environ["FEASTHIP_PANEL_LL"] = m
        F = pkg.lu(Cm); Y = np.zeros((n, 5), dtype=np.complex128, order="F"); pkg.ldiv(Y, F, R); pkg.finalize(F)
        sols[m] = Y
    print(n, "diff", f"{np.abs(sols['1'] - sols['0']).max():.2e}", "residual", f"{np.linalg.norm(Cm @ sols['1'] - R) / (np.linalg.norm(Cm) * np.linalg.norm(sols['1'])):.2e}", f"{np.linalg.norm(Cm @ sols['0'] - R) / (np.linalg.norm(Cm) * np.linalg.norm(sols['0'])):.2e}", flush=True)
