#!/bin/bash
# HBM traffic of the sweep kernel from the PMC counters (separate passes, kernel trace only beside them), final build of round 3
set -o pipefail
O=gpurun_out/r03h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 bench.py --cpu-baseline off --steps 2 --warmup 1 > $O/pmc_f.json 2> $O/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python3 bench.py --cpu-baseline off --steps 2 --warmup 1 > $O/pmc_w.json 2> $O/pmc_w.err
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w $O/pmc_traffic.json k_apply 4 > $O/pmc_traffic.log 2>&1
python tools/pmc_per_launch.py $O/pmc_f $O/pmc_w > $O/pmc_per_launch_traffic.txt 2>&1
cat $O/pmc_traffic.json
