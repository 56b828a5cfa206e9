#!/bin/bash
# Round-3 measurement set (run on the GPU box through gpurun): bench lines, rocprofv3 kernel stats, PMC traffic passes,
# factor trace + kernel totals, secondary configurations, per-rank load.  Outputs under gpurun_out/r03c/; copy the summaries into
# profiles/ afterwards.  The CPU baseline on the metric's own 46^3 configuration (minutes of serial SuperLU) runs beside the
# GPU work on the host cores, first.
set -o pipefail
O=gpurun_out/r03c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py --cpu-baseline full --record --steps 3 --warmup 1 > $O/bench_cpu_full.json 2> $O/bench_cpu_full.err &
CPUJOB=$!
sleep 45     # the GPU part of that run (a 3-step bench) is over by now; the host then factorises for minutes
python bench.py --cpu-baseline off --no-fold > $O/bench_c3_g46_no_fold.json 2>> $O/bench.err
python bench.py --cpu-baseline off --mixed-prec > $O/bench_c3_g46_mixed_prec.json 2>> $O/bench.err
FEASTHIP_NOPIV=0 python bench.py --cpu-baseline off > $O/bench_c3_g46_pivoted_factorisation.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r03 -- python3 bench.py --cpu-baseline off > $O/bench_c3_g46_under_rocprof.json 2> $O/prof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 bench.py --cpu-baseline off --steps 2 --warmup 1 > $O/pmc_f.json 2> $O/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python3 bench.py --cpu-baseline off --steps 2 --warmup 1 > $O/pmc_w.json 2> $O/pmc_w.err
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w $O/pmc_traffic.json k_apply 4 > $O/pmc_traffic.log 2>&1
python tools/factor_trace.py > $O/factor_trace.out 2> $O/factor_trace_c3_g46.txt
python tools/bench_c2.py > $O/bench_c2_dense4096.json 2>> $O/bench.err
python bench.py --cpu-baseline off --grid 64 > $O/bench_c3_g64_scale_check.json 2>> $O/bench.err
echo "gpu part done, waiting for the cpu baseline" >&2
wait $CPUJOB
cp profiles/r03_cpu_baseline_c3_g46.json $O/ 2>/dev/null
python bench.py > $O/bench_c3_g46.json 2> $O/bench_c3_g46.err
echo collected
