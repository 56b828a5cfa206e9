#!/bin/bash
# Round-2 measurement set (run on the GPU box through gpurun): bench lines, rocprofv3 kernel stats, PMC traffic passes,
# factor trace, secondary configurations.  Outputs under gpurun_out/r02/; copy the summaries into profiles/ afterwards.
set -o pipefail
O=gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py > $O/bench_c3_g46.json 2> $O/bench_c3_g46.err || exit 1
python bench.py --cpu-baseline off --no-fold > $O/bench_c3_g46_no_fold.json 2>> $O/bench.err
python bench.py --cpu-baseline off --mixed-prec > $O/bench_c3_g46_mixed_prec.json 2>> $O/bench.err
FEASTHIP_APPLY2=0 python bench.py --cpu-baseline off > $O/bench_c3_g46_k_apply_lds.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r02 -- python3 bench.py --cpu-baseline off > $O/bench_c3_g46_under_rocprof.json 2> $O/prof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 bench.py --cpu-baseline off --steps 2 --warmup 1 > $O/pmc_f.json 2> $O/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python3 bench.py --cpu-baseline off --steps 2 --warmup 1 > $O/pmc_w.json 2> $O/pmc_w.err
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w $O/pmc_traffic.json k_apply 4 > $O/pmc_traffic.log 2>&1
python tools/factor_trace.py > $O/factor_trace.out 2> $O/factor_trace_c3_g46.txt
python tools/bench_c2.py > $O/bench_c2_dense4096.json 2>> $O/bench.err
python bench.py --cpu-baseline off --grid 64 > $O/bench_c3_g64_scale_check.json 2>> $O/bench.err
python tools/analyze_trace.py > $O/per_rank_load.txt 2>&1
echo collected
