#!/bin/bash
# kernel traces of the factorisations alone (C2 pivoted dense, C3 symmetric sparse, C5 dense): tools/factor_timeline.py summaries
set -o pipefail
O=gpurun_out/${1:-r05a}; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in ${2:-c3 c2 c5}; do
  rm -rf $R/$O/tr_$c
  (cd /tmp && FEASTHIP_FACTOR_TRACE=1 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr_$c -o t -- python3 $R/tools/factor_only.py $c 3 > $R/$O/factor_$c.out 2> $R/$O/factor_$c.err) || { echo "trace $c failed"; tail -5 $R/$O/factor_$c.err; exit 1; }
  f=$(find $R/$O/tr_$c -name '*kernel_trace.csv' | head -1)
  python3 $R/tools/factor_timeline.py $f 1 --seq 60 > $R/$O/timeline_$c.txt
  cat $R/$O/factor_$c.out
  rm -rf $R/$O/tr_$c
done
echo collected
