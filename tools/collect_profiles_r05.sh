#!/bin/bash
# Round-5 measurement set (run on the GPU box through gpurun, on the final build): bench lines, the rocprofv3 kernel-trace summary of
# the bench command (with the check that the profiled process did not abort at exit: DESIGN.md section 6), the secondary
# configurations, the per-rank loads of the node-sharded run on one GPU.  Counter passes: tools/collect_pmc_r05.sh.
set -o pipefail
O=gpurun_out/${1:-r05final}; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
python bench.py > $O/bench_c3_g46.json 2> $O/bench_c3_g46.err || exit 1
python bench.py --cpu-baseline off --no-fold > $O/bench_c3_g46_no_fold.json 2>> $O/bench.err
python bench.py --cpu-baseline off --mixed-prec > $O/bench_c3_g46_mixed_prec.json 2>> $O/bench.err
FEASTHIP_RR_DEVICE=0 python bench.py --cpu-baseline off > $O/bench_c3_g46_host_eigensolver.json 2>> $O/bench.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r05 -- python3 $R/bench.py --cpu-baseline off --cold-solves 0 --store-false-steps 0 > $R/$O/bench_c3_g46_under_rocprof.json 2> $R/$O/prof.err)
if grep -q "Aborted\|Segmentation\|core dumped" $O/prof.err; then echo "ABORT in the profiled bench run"; exit 2; fi
FEASTHIP_ANALYZE_TRACE=1 python tools/factor_trace.py > $O/factor_trace.out 2> $O/factor_trace_c3_g46.txt
python tools/bench_c2.py > $O/bench_c2_dense4096.json 2>> $O/bench.err
python tools/bench_c4c5.py > $O/bench_c4_c5.json 2>> $O/bench.err
python bench.py --cpu-baseline off --grid 64 > $O/bench_c3_g64_scale_check.json 2>> $O/bench.err
python tools/few_units_all.py > $O/per_rank_load_single_gpu.txt 2>> $O/bench.err
python tools/hermitian_c3.py > $O/hermitian_c3.txt 2>> $O/bench.err
FEASTHIP_ADJ_FOLD=0 python tools/hermitian_c3.py > $O/hermitian_c3_no_adjoint_fold.txt 2>> $O/bench.err
python tools/other_calls.py > $O/other_calls.txt 2>> $O/bench.err
python tools/c3_whole_call.py > $O/c3_whole_call.txt 2>> $O/bench.err
python tools/c1_whole_call.py 2>> $O/bench.err | grep -v amdgpu > $O/c1_whole_call.txt
bash tools/collect_factor_traces.sh $(basename $O) "c3 c2 c5" > $O/factor_traces.log 2>&1
python tools/launch_profile.py 46 16 > $O/launch_profile_16_nodes.txt 2>> $O/bench.err
for c in c3 c2 c5; do FEASTHIP_PROBE_SKIP_CONVERSION=1 python tools/factor_only.py $c 3 2>/dev/null; python tools/factor_only.py $c 3 2>/dev/null; done > $O/conversion_exposure.txt
FEASTHIP_ANALYZE_TRACE=1 python tools/factor_trace.py 2>&1 | grep "feasthip analyze\|feasthip nd\] last" > $O/analysis_trace_c3_g46.txt
for i in 1 2 3; do python tools/analysis_time.py 46 5; done > $O/analysis_probe_c3_g46.txt 2>> $O/bench.err
g++ -O3 -mavx2 -mfma -std=c++17 -pthread -Ifeastsolver.jl_amd/csrc -o /tmp/smalleig_bench tools/probes/smalleig_bench.cpp feastsolver.jl_amd/csrc/smalleig.cpp feastsolver.jl_amd/csrc/hostpool.cpp && { FEASTHIP_EIG_TRACE=1 /tmp/smalleig_bench 2>&1 | tail -8; echo "--- FEASTHIP_HOST_THREADS=1"; FEASTHIP_HOST_THREADS=1 /tmp/smalleig_bench; } > $O/smalleig_bench.txt
{ python tools/c5_rr_time.py; FEASTHIP_HOST_THREADS=1 python tools/c5_rr_time.py; } > $O/c5_rr_step.txt 2>> $O/bench.err
echo collected
