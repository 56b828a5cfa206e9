"""Expected eigenvalues of the complex Hermitian C3 pencil (workloads.c3_hermitian) for the full-size GPU test:
shift-invert Lanczos (scipy eigsh) around sigma = 0 with a SuperLU factorisation ordered by MMD(A^T + A).

Run in the build container:  python tests/golden/make_hermitian_fixture.py [g]   (g = 46: a few minutes, ~6 GB)
Only outputs are stored; the inputs are regenerated from seeds by feastsolver.jl_amd/workloads.py."""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "feastsolver.jl_amd"))
import workloads as wl  # noqa: E402

g = int(sys.argv[1]) if len(sys.argv) > 1 else 46
out_path = os.path.join(HERE, f"c3_hermitian_g{g}.json")
w = wl.c3_hermitian(g, 19, 64)
A, B = w["A"], w["B"]
n = A.shape[0]
if "--recontour" in sys.argv and os.path.exists(out_path):       # the stored lowest eigenvalues, the contour chosen again
    with open(out_path) as f:
        ev = np.array(json.load(f)["eigs_lowest"])
else:
    assert abs(A - A.conj().T).max() < 1e-14
    lu = spla.splu(sp.csc_matrix(A), permc_spec="MMD_AT_PLUS_A")
    print("factor entries", lu.L.nnz + lu.U.nnz, flush=True)
    OPinv = spla.LinearOperator((n, n), matvec=lambda x: lu.solve(np.asarray(x, dtype=np.complex128)), dtype=np.complex128)
    ev = np.sort(spla.eigsh(A, k=48, M=B, sigma=0.0, which="LM", OPinv=OPinv, return_eigenvectors=False, tol=1e-13).real)
# the gauge field lifts the bottom of the spectrum, so the real pencil's contour would hold 10 eigenvalues only: the contour is chosen
# on this pencil's own spectrum like c3_contour() of make_fixtures.py does (about 32 inside, the largest relative gap nearby)
n_inside = 32
best = max(range(n_inside - 4, n_inside + 5), key=lambda j: (ev[j] - ev[j - 1]) / ev[j])
hi = 0.5 * (ev[best - 1] + ev[best])
lo = ev[0] - (ev[1] - ev[0])
c, r = 0.5 * (lo + hi), 0.5 * (hi - lo)
inside = ev[np.abs(ev - c) <= r]
assert len(inside) == best and inside[-1] < ev[-1]
out = dict(g=g, n=n, nnz=int(A.nnz), c=float(c), r=float(r), eigs_inside=[float(x) for x in inside], eigs_lowest=[float(x) for x in ev],
           gap_at_the_contour=float(ev[best] - ev[best - 1]))
with open(out_path, "w") as f:
    json.dump(out, f, indent=0)
print(len(inside), "inside; c, r =", c, r)
