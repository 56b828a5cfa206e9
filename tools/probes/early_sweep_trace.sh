#!/bin/bash
# kernel trace of a few store=false iterations of C3 and the timeline of the last one (tools/probes/early_sweep_timeline.py)
set -o pipefail
O=gpurun_out/${1:-r05g}; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/$O/tr_early
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr_early -o t -- python3 $R/bench.py --cpu-baseline off --cold-solves 0 --store-false-steps 3 --steps 2 --warmup 1 > $R/$O/early_bench.out 2> $R/$O/early_bench.err) || { tail -5 $R/$O/early_bench.err; exit 1; }
f=$(find $R/$O/tr_early -name '*kernel_trace.csv' | head -1)
python3 $R/tools/probes/early_sweep_timeline.py $f 2 ${3:-0} > $R/$O/early_timeline${2:-}.txt
rm -rf $R/$O/tr_early
cat $R/$O/early_timeline${2:-}.txt
