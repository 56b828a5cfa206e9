#!/bin/bash
set -o pipefail
O=gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "splitk or shard or distributed or fold or apply_kernels or mixed or wide_blocks or c3_sparse" > $O/gpu_tests_splitk.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests_splitk.txt; tail -4 $O/gpu_tests_splitk.txt
python tools/analyze_trace.py > $O/per_rank_load.txt 2>&1; tail -4 $O/per_rank_load.txt
FEASTHIP_FWD_SPLITK=0 python tools/analyze_trace.py > $O/per_rank_load_nosplit.txt 2>&1; tail -4 $O/per_rank_load_nosplit.txt
python tools/launch_profile.py 46 2 > $O/launch_profile_2nodes.txt 2>&1
