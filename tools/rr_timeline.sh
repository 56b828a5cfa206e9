#!/bin/bash
# kernel timeline of one Rayleigh-Ritz step (between two contour steps) of the default bench
set -o pipefail
O=gpurun_out/r03e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d $O/prof_rr -o rr -- python3 bench.py --steps 6 --warmup 2 --cpu-baseline off > $O/prof_rr.out 2> $O/prof_rr.err
python3 - <<'PY' > $O/rr_timeline.txt 2>&1
import csv, glob, re
f = glob.glob("gpurun_out/r03e/prof_rr/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r'(k_[a-zA-Z_0-9]+|__amd_rocclr_[a-zA-Z]+)', n); return m.group(1) if m else n[:30]
acc = [i for i, r in enumerate(rows) if "k_accumulate" in r["Kernel_Name"]]
gat = [i for i, r in enumerate(rows) if "k_gather_rhs" in r["Kernel_Name"]]
a = acc[-3]; b = [g for g in gat if g > a][0]
t0 = int(rows[a]["End_Timestamp"]); prev = t0
busy = 0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0)/1e3:9.1f} gap {(s - prev)/1e3:7.1f}  {(e - s)/1e3:8.1f}us {short(r['Kernel_Name']):28s} {r.get('Grid_Size_X','')}")
    busy += e - s; prev = e
print("span us", (prev - t0) / 1e3, "busy us", busy / 1e3)
PY
tail -1 $O/prof_rr.out | cut -c1-120
