#!/bin/bash
# pivot-light factorisation: targeted tests, full suite, bench, factor trace, per-rank load
set -o pipefail
O=gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "pivot_light or symmetric or seam or c3_sparse or irregular or zero_diagonal or c1_dense or runtests" > $O/gpu_tests_nopiv.txt 2>&1
echo "pytest rc $?" >> $O/gpu_tests_nopiv.txt; tail -4 $O/gpu_tests_nopiv.txt
grep -q "pytest rc 0" $O/gpu_tests_nopiv.txt || exit 1
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py::test_c2_full_size_dense4096 > $O/gpu_tests_second.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests_second.txt
tail -4 $O/gpu_tests_second.txt
python bench.py > $O/bench_c3_g46_nopiv.json 2> $O/bench_c3_g46_nopiv.err || exit 1
FEASTHIP_NOPIV=0 python bench.py --cpu-baseline off > $O/bench_c3_g46_pivoted.json 2> $O/bench_c3_g46_pivoted.err
FEASTHIP_FACTOR_TRACE=1 python tools/factor_trace.py > $O/factor_trace.out 2> $O/factor_trace_nopiv.txt
python tools/analyze_trace.py > $O/per_rank_load.txt 2>&1
python tools/launch_profile.py 46 2 > $O/launch_profile_2nodes.txt 2>&1
echo done
