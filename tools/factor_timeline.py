"""Timeline of ONE factorisation from a rocprofv3 --kernel-trace CSV: which kernels ran on which queue, how long each queue was busy,
when the device ran only chain kernels, and the launch sequence of the critical (panel) queue.
Usage: python tools/factor_timeline.py <kernel_trace.csv> [nth-factorisation-from-end=1] [--seq N]
A factorisation is delimited by its first assembly kernel (k_dense_assemble / k_scatter) and the next one (or the end of the trace)."""
import csv, re, sys
from collections import defaultdict

path = sys.argv[1]
nth = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else 1
seq = int(sys.argv[sys.argv.index("--seq") + 1]) if "--seq" in sys.argv else 0
rows = []
for r in csv.DictReader(open(path)):
    nm = r["Kernel_Name"]
    m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", nm)
    short = (m.group(1) + (m.group(2) or "")) if m else nm[:30]
    short = re.sub(r"HIP_vector_type<double, 2u>", "cz", short)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Queue_Id", "?"), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0),
                 int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_dense_assemble") or (r[2].startswith("k_scatter") and (i == 0 or not rows[i - 1][2].startswith("k_scatter")))]
if not starts:
    sys.exit("no factorisation found")
i0 = starts[-nth]
i1 = starts[-nth + 1] if nth > 1 else len(rows)
# stop at the first kernel that is not part of a factorisation (sweeps, Rayleigh-Ritz)
fact_names = ("k_dense_assemble", "k_scatter", "k_extend_add", "k_panel", "k_swap", "k_trsm_u", "k_gemm_nn", "k_group_utrsm", "k_diag", "k_lscale", "k_rowperm",
              "k_block_inverse", "k_couple", "k_scale_panels", "k_to_f32", "k_col", "k_fpanel", "k_pivot", "k_dpanel", "k_sympanel", "__amd_rocclr")
for j in range(i0, i1):
    if not rows[j][2].startswith(fact_names):
        i1 = j
        break
R = rows[i0:i1]
t0, t1 = R[0][0], max(r[1] for r in R)
print(f"factorisation: {len(R)} launches, {(t1 - t0) / 1e6:.2f} ms wall")
byq = defaultdict(list)
for r in R: byq[r[3]].append(r)
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
for q, rs in sorted(byq.items(), key=lambda kv: -sum(r[1] - r[0] for r in kv[1])):
    busy = union([(r[0], r[1]) for r in rs])
    agg = defaultdict(lambda: [0, 0])
    for r in rs: agg[r[2]][0] += 1; agg[r[2]][1] += r[1] - r[0]
    top = ", ".join(f"{k} {v[0]}x {v[1] / 1e6:.2f}ms" for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8])
    print(f"queue {q}: {len(rs)} launches, busy {busy / 1e6:.2f} ms ({(rs[0][0] - t0) / 1e6:.2f} .. {(rs[-1][1] - t0) / 1e6:.2f}) | {top}")
agg = defaultdict(lambda: [0, 0])
for r in R: agg[r[2]][0] += 1; agg[r[2]][1] += r[1] - r[0]
print("kernel totals:")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]): print(f"  {k:34s} {v[0]:5d} x {v[1] / v[0] / 1e3:8.1f} us = {v[1] / 1e6:7.2f} ms")
# time with only 'small' kernels running (grid below 64 workgroups) and idle time
ev = []
for r in R:
    big = (r[4] // max(1, r[5])) >= 64
    ev.append((r[0], 1, big)); ev.append((r[1], -1, big))
ev.sort()
nb = ns = 0; last = t0; only_small = idle = 0
for t, d, big in ev:
    if nb == 0 and ns > 0: only_small += t - last
    if nb == 0 and ns == 0: idle += t - last
    last = t
    if big: nb += d
    else: ns += d
print(f"only launches of < 64 workgroups running: {only_small / 1e6:.2f} ms; nothing running: {idle / 1e6:.2f} ms")
if seq:
    q = max(byq.items(), key=lambda kv: sum(1 for r in kv[1] if r[2].startswith(("k_panel", "k_diag", "k_fpanel", "k_sympanel", "k_dpanel"))))[0]
    print(f"last {seq} launches of the panel queue {q} (start ms, duration us, gap to previous end us, kernel, workgroups):")
    rs = byq[q][-seq:]
    for a, b in zip([None] + rs[:-1], rs):
        gap = (b[0] - a[1]) / 1e3 if a else 0.0
        print(f"  {(b[0] - t0) / 1e6:9.3f} {(b[1] - b[0]) / 1e3:8.1f} {gap:8.1f} {b[2]:30s} {b[4] // max(1, b[5])}")
