#!/bin/bash
# quick loop for the factorisation work: targeted tests, factor trace, kernel timeline
set -o pipefail
O=gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "pivot_light or symmetric or seam or c3_sparse or irregular or zero_diagonal or c1_dense or runtests" > $O/gpu_tests_nopiv.txt 2>&1
echo "pytest rc $?" >> $O/gpu_tests_nopiv.txt; tail -4 $O/gpu_tests_nopiv.txt
grep -q "pytest rc 0" $O/gpu_tests_nopiv.txt || exit 1
FEASTHIP_FACTOR_TRACE=1 python tools/factor_trace.py > $O/factor_trace.out 2> $O/factor_trace_nopiv.txt
tail -13 $O/factor_trace_nopiv.txt | cut -c1-200
grep factor_ms $O/factor_trace.out | head -2
bash tools/r03_ftrace.sh
