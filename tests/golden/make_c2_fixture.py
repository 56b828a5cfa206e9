"""One-off host eigensolve of the seeded C2 pencil at its full size (BASELINE configs[1]: dense generalized 4096 x 4096 complex
non-Hermitian, seeds 2001/2002): radius = midpoint between the 32nd and 33rd smallest |lambda|, and the eigenvalues
inside it.  Written to tests/golden/c2_contour_n4096.json; the GPU test then checks the exact count and the eigenvalues.
LAPACK zggev through scipy, a few minutes on 8 cores.  Usage: python tests/golden/make_c2_fixture.py [n]"""
import importlib
import json
import os
import sys
import time

import numpy as np
import scipy.linalg as sla

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
wl = importlib.import_module(graft.PKG_NAME + ".workloads")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w = wl.c2_dense_generalized(n, 64, radius=1.0)
t0 = time.time()
ev = sla.eigvals(w["A"], w["B"])
a = np.sort(np.abs(ev))
r = 0.5 * (a[31] + a[32])
inside = ev[np.abs(ev) <= r]
inside = inside[np.lexsort((inside.imag, inside.real))]
out = {"n": n, "c": [0.0, 0.0], "r": float(r), "n_inside": int(len(inside)), "gap_to_next": float(a[32] - a[31]),
       "eigs": [[float(z.real), float(z.imag)] for z in inside], "seeds": [2001, 2002, 2003],
       "made_by": "tests/golden/make_c2_fixture.py (scipy.linalg.eigvals = LAPACK zggev), %.0f s" % (time.time() - t0)}
json.dump(out, open(os.path.join(HERE, f"c2_contour_n{n}.json"), "w"), indent=1)
print(out["n_inside"], out["r"], out["gap_to_next"], out["made_by"])
