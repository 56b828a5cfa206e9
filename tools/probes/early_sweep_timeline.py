"""Where the early forward sweep (engine.hpp: early_rhs_) lands inside a store=false iteration: from a rocprofv3 --kernel-trace CSV, the last
factorisation (first k_scatter of a run of them) up to the k_accumulate that ends its contour step: per queue busy time and the sequence of the
sweep kernels and of the conversion kernels whose events they wait for.  Usage: python tools/probes/early_sweep_timeline.py <kernel_trace.csv> [nth factorisation from the end = 2]"""
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:24], r.get("Queue_Id", "?"), int(r.get("Grid_Size_X", 0) or 0) // max(1, int(r.get("Workgroup_Size_X", 1) or 1))))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2] == "k_scatter" and (i == 0 or rows[i - 1][2] != "k_scatter")]
i0 = starts[-(int(sys.argv[2]) if len(sys.argv) > 2 else 2)]      # bench.py factorises once more after its store=false leg: the last but one
i1 = next(j for j in range(i0, len(rows)) if rows[j][2] == "k_accumulate") + 1
R = rows[i0:i1]
t0 = R[0][0]
print(f"iteration window: {len(R)} launches, {(R[-1][1] - t0) / 1e6:.2f} ms")
qs = sorted({r[3] for r in R})
for q in qs:
    rq = [r for r in R if r[3] == q]
    names = {}
    for r in rq: names[r[2]] = names.get(r[2], 0) + (r[1] - r[0])
    print(f"queue {q}: {len(rq)} launches, first {(rq[0][0] - t0) / 1e6:.2f} last end {(max(r[1] for r in rq) - t0) / 1e6:.2f} ms | " + ", ".join(f"{k} {v / 1e6:.2f}" for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:6]))
show = ("k_gather_rhs", "k_fwd_assemble", "k_apply2", "k_scale_panels", "k_couple_level", "k_rowperm", "k_splitk_reduce")
tail_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0      # third argument: every launch of the last so many milliseconds before the root's sweep
print("start ms   dur us  queue kernel workgroups")
part2 = next((r[0] for r in R if r[3] == R[0][3] and r[2] in ("k_fwd_assemble", "k_apply2")), R[-1][0])
for r in R:
    if (r[2] in show and (r[1] - r[0] > 30000 or r[2] in ("k_rowperm", "k_gather_rhs"))) or (tail_ms and part2 - tail_ms * 1e6 <= r[0] < part2):
        print(f"{(r[0] - t0) / 1e6:8.3f} {(r[1] - r[0]) / 1e3:8.1f}  {r[3]:>4} {r[2]:18s} {r[4]}")
