#!/bin/bash
# Round-3 measurement set, part 2: everything whose TIMES matter, on a quiet host (part 1, tools/collect_profiles_r03.sh, ran the
# serial CPU baseline of the 46^3 configuration beside the GPU work: 64 BLAS threads on the host tripled the launch-bound contour
# steps of the lines collected there; the counter passes and the CPU line itself are what is kept from part 1).
set -o pipefail
O=gpurun_out/r03g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py > $O/bench_c3_g46.json 2> $O/bench_c3_g46.err || exit 1
python bench.py --cpu-baseline off --no-fold > $O/bench_c3_g46_no_fold.json 2>> $O/bench.err
python bench.py --cpu-baseline off --mixed-prec > $O/bench_c3_g46_mixed_prec.json 2>> $O/bench.err
FEASTHIP_NOPIV=0 python bench.py --cpu-baseline off > $O/bench_c3_g46_pivoted_factorisation.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r03 -- python3 bench.py --cpu-baseline off > $O/bench_c3_g46_under_rocprof.json 2> $O/prof.err
FEASTHIP_ANALYZE_TRACE=1 python tools/factor_trace.py > $O/factor_trace.out 2> $O/factor_trace_c3_g46.txt
python tools/bench_c2.py > $O/bench_c2_dense4096.json 2>> $O/bench.err
python bench.py --cpu-baseline off --grid 64 > $O/bench_c3_g64_scale_check.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_factor -o pf -- python3 tools/factor_trace.py > $O/prof_factor.out 2> $O/prof_factor.err
echo collected
