"""Replays one dumped fuzz case (tools/fuzz_feast.py with FUZZ_ONLY / FUZZ_DUMP): python tools/probes/replay_case.py case.npz [store] [mixed] [iter]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package()
d = np.load(sys.argv[1])
store = (sys.argv[2] == "1") if len(sys.argv) > 2 else bool(d["store"])
mixed = (sys.argv[3] == "1") if len(sys.argv) > 3 else bool(d["mixed"])
it = int(sys.argv[4]) if len(sys.argv) > 4 else 40
A = d["A"]
if "sparse" in d and bool(d["sparse"]):
    import scipy.sparse as sp
    A = sp.csc_matrix(A)          # the run took the sparse path
con = pkg.circular_contour_trapezoidal(complex(d["c"]), float(d["r"]), int(d["K"]))
info = {}
lam, V, res = pkg.feast(d["X0"].copy(), A, con, iter=it, eps=1e-9, store=store, mixed_prec=mixed, info=info)
lam_all = np.linalg.eigvals(d["A"])
ref = lam_all[np.abs(lam_all - complex(d["c"])) <= float(d["r"])]
print(f"store {store} mixed {mixed}: inside {len(ref)} found {len(lam)} nit {info.get('nit')} max res {res.max() if len(res) else None} symmetric_path {info.get('symmetric')} "
      f"env NOPIV={os.environ.get('FEASTHIP_NOPIV')} LAZY_OFF={os.environ.get('FEASTHIP_PROBE_NO_LAZY')}")
print({k: info[k] for k in info if k in ("growth", "perturbed", "symmetric", "nopiv", "factor_ms")})
if os.environ.get("REPLAY_ORACLE"):          # the CPU oracle as the checker, on the same inputs: where the reference's algorithm itself stands after `it` iterations
    fo = graft.load_oracle()
    hist = []
    lo, Vo, ro = fo.feast(d["X0"].copy(), d["A"], fo.circular_contour_trapezoidal(complex(d["c"]), float(d["r"]), int(d["K"])), iter=it, eps=1e-9, store=True, history=hist)
    print("oracle: found", len(lo), "max res", (ro.max() if len(ro) else None), "iterations", len(hist))
