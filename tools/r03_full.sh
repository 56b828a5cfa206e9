#!/bin/bash
# whole GPU suite + default bench line
set -o pipefail
O=gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests_full.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests_full.txt
tail -5 $O/gpu_tests_full.txt
python bench.py > $O/bench_c3_g46.json 2> $O/bench_c3_g46.err || exit 1
python tools/analyze_trace.py > $O/per_rank_load.txt 2>&1
python tools/launch_profile.py 46 2 > $O/launch_profile_2nodes.txt 2>&1
echo done
