"""Per-level timing of the 16 shifted factorisations of C3 (FEASTHIP_FACTOR_TRACE prints one line per tree level).
Usage: python tools/factor_trace.py [g]"""
import importlib
import os
import sys

os.environ["FEASTHIP_FACTOR_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (first: the library then shares torch's HIP runtime)
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
api = importlib.import_module(graft.PKG_NAME + ".api")
wl = importlib.import_module(graft.PKG_NAME + ".workloads")
contour = importlib.import_module(graft.PKG_NAME + ".contour")

g = int(sys.argv[1]) if len(sys.argv) > 1 else 46
w = wl.c3_sparse_generalized(g=g)
h = api.FeastHandle(0)
h.set_terms([w["A"], w["B"]], scales=[1.0, -1.0])
h.set_contour(contour.circular_contour_trapezoidal(w["c"], w["r"], w["nodes"]))
for rep in range(2):
    print(f"--- factorisation {rep}", file=sys.stderr, flush=True)
    h.factorize()
print(h.stats())
h.close()
