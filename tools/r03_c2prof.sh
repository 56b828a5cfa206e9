#!/bin/bash
# kernel statistics of the C2 run (dense 4096 x 4096, 16 pivoted LU factorisations + FEAST loop)
set -o pipefail
O=gpurun_out/r03e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o c2 -- python3 tools/bench_c2.py > $O/prof_c2.out 2> $O/prof_c2.err
python3 - <<'PY' > $O/c2_kernels.txt 2>&1
import csv, glob, re
from collections import defaultdict
f = glob.glob("gpurun_out/r03e/prof_c2/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r'(k_[a-zA-Z_0-9]+)', n); return m.group(1) if m else n[:30]
agg = defaultdict(lambda: [0, 0.0])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    agg[short(r["Kernel_Name"])][0] += 1; agg[short(r["Kernel_Name"])][1] += (e - s) / 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]): print(f"  {k:28s} {v[0]:6d} calls {v[1]:10.1f} us  avg {v[1]/v[0]:8.1f}")
# timeline of panel steps 60..63 (middle of the factorisation)
pan = [i for i, r in enumerate(rows) if "k_panel" in r["Kernel_Name"]]
if len(pan) > 70:
    a, b = pan[60], pan[64]
    t0 = int(rows[a]["Start_Timestamp"])
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0)/1e3:9.1f} {(e - t0)/1e3:9.1f} {(e - s)/1e3:8.1f}us q{r.get('Queue_Id','?')} {short(r['Kernel_Name']):22s} {r.get('Grid_Size_X','')}x{r.get('Grid_Size_Y','')}x{r.get('Grid_Size_Z','')} wg{r.get('Workgroup_Size_X','')}")
PY
tail -3 $O/prof_c2.out
