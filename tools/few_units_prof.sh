#!/bin/bash
# rocprofv3 kernel stats of the contour steps with one and two factorisation units per GPU (the per-rank load of an 8- / 4-GPU run)
O=gpurun_out/few; mkdir -p $O; rm -rf $O/prof
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 tools/few_units.py > $O/out.txt 2> $O/err.txt
cat $O/out.txt
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    print(r['Name'][:80].ljust(80), r['Calls'].rjust(6), round(int(r['TotalDurationNs'])/1e6,2), 'ms  avg', round(float(r['AverageNs'])/1e3,1), 'us')
PY
