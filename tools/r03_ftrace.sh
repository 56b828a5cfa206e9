#!/bin/bash
# kernel timeline of the pivot-light factorisation (root front window) + kernel stats
set -o pipefail
O=gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_factor -o pf -- python3 tools/factor_trace.py > $O/prof_factor.out 2> $O/prof_factor.err
python - <<'PY' > gpurun_out/r03/factor_timeline.txt 2>&1
import csv, glob, re
f = glob.glob("gpurun_out/r03/prof_factor/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r'(k_[a-z_0-9]+)', n); return m.group(1) if m else n[:30]
# last factorisation: find the last k_scatter
idx = [i for i, r in enumerate(rows) if "k_scatter" in r["Kernel_Name"]]
start = idx[-2] if len(idx) >= 2 else idx[-1]
t0 = int(rows[start]["Start_Timestamp"])
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    agg[short(r["Kernel_Name"])][0] += 1; agg[short(r["Kernel_Name"])][1] += (e - s) / 1e3
print("kernel totals of the last factorisation (us):")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]): print(f"  {k:24s} {v[0]:6d} calls {v[1]:10.1f} us  avg {v[1]/v[0]:8.1f}")
print("timeline of the last 4 ms (root front):")
tend = max(int(r["End_Timestamp"]) for r in rows[start:])
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s >= tend - 4_000_000:
        print(f"{(s - t0)/1e6:9.3f} {(e - t0)/1e6:9.3f} {(e - s)/1e3:8.1f}us q{r.get('Queue_Id','?')} {short(r['Kernel_Name']):18s} {r.get('Grid_Size_X','')}x{r.get('Grid_Size_Y','')}")
PY
echo done
