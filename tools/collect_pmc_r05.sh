#!/bin/bash
# HBM traffic of the sweep kernel from the PMC counters (separate passes, kernel trace only beside them), final build of round 5
set -o pipefail
O=gpurun_out/${1:-r05pmc}; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_f -o f -- python3 $R/bench.py --cpu-baseline off --steps 2 --warmup 1 --cold-solves 0 --store-false-steps 0 > $R/$O/pmc_f.json 2> $R/$O/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_w -o w -- python3 $R/bench.py --cpu-baseline off --steps 2 --warmup 1 --cold-solves 0 --store-false-steps 0 > $R/$O/pmc_w.json 2> $R/$O/pmc_w.err
cd $R
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w $O/pmc_traffic.json k_apply 4 > $O/pmc_traffic.log 2>&1
python tools/pmc_per_launch.py $O/pmc_f $O/pmc_w > $O/pmc_per_launch_traffic.txt 2>&1
cat $O/pmc_traffic.json
