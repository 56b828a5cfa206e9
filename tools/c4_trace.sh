#!/bin/bash
# kernel timeline of the C4 call (butterfly, nlfeast): where a 64 x 64 problem spends its 6 ms per iteration
set -o pipefail
O=gpurun_out/r03e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/prof_c4 -o c4 -- python3 tools/bench_c4c5.py > $O/prof_c4.out 2> $O/prof_c4.err
python3 - <<'PY' > $O/c4_timeline.txt 2>&1
import csv, glob, re
f = glob.glob("gpurun_out/r03e/prof_c4/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r'(k_[a-zA-Z_0-9]+|__amd_rocclr_[a-zA-Z]+)', n); return m.group(1) if m else n[:30]
# the third nlfeast call: find k_hankel? use the last run of kernels before the first k_dense_import with n=2000 ... simply print the
# 400 kernels before the first kernel whose grid is large (C5 starts)
big = [i for i, r in enumerate(rows) if "k_dense_import" in r["Kernel_Name"] and int(r["Grid_Size_X"]) > 1000]
end = big[0] if big else len(rows)
seg = rows[max(0, end - 420):end]
t0 = int(seg[0]["Start_Timestamp"]); prev = t0; busy = 0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0)/1e3:9.1f} gap {(s - prev)/1e3:7.1f}  {(e - s)/1e3:7.1f}us {short(r['Kernel_Name']):26s} {r.get('Grid_Size_X','')}x{r.get('Grid_Size_Y','')}")
    busy += e - s; prev = e
print("span us", (prev - t0) / 1e3, "busy us", busy / 1e3, "kernels", len(seg))
PY
tail -2 $O/c4_timeline.txt
