#!/bin/bash
# Round-3 first GPU pass: full -m gpu suite with the CPU 46^3 baseline beside it (host cores), then bench + factor trace.
set -o pipefail
O=gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
(python bench.py --cpu-baseline full --record --steps 3 --warmup 1 > $O/bench_cpu_full.json 2> $O/bench_cpu_full.err; cp profiles/r03_cpu_baseline_c3_g46.json $O/ 2>/dev/null) &
CPUJOB=$!
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests_first.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests_first.txt
tail -5 $O/gpu_tests_first.txt
wait $CPUJOB
python bench.py > $O/bench_c3_g46_first.json 2> $O/bench_c3_g46_first.err || exit 1
FEASTHIP_FACTOR_TRACE=1 python tools/factor_trace.py > $O/factor_trace.out 2> $O/factor_trace_first.txt
echo done
