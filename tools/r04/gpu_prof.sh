#!/bin/bash
# kernel-trace profile of the bench command (no counters): per-kernel statistics under gpurun_out/<tag>/
tag=${1:-r04prof}; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-baseline off "$@" > $GRAFT_REPO_ROOT/$out/bench.log 2> $GRAFT_REPO_ROOT/$out/bench.err
rc=$?; echo "rocprof rc=$rc"
cd $GRAFT_REPO_ROOT
if grep -q "Aborted\|Segmentation" $out/bench.err; then echo "ABORT in profiled run"; fi
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); echo $f
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:40]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):6d} total {float(r["TotalDurationNs"])/1e6:9.3f} ms avg {float(r["AverageNs"])/1e3:9.1f} us')
PY
