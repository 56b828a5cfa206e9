#!/bin/bash
# copies the round-5 measurement set (tools/collect_profiles_r05.sh, tools/collect_pmc_r05.sh) from gpurun_out/ into profiles/ under r05_ names
set -e
O=gpurun_out/${1:-r05final}; P=gpurun_out/${2:-r05pmc}
for f in bench_c3_g46.json bench_c3_g46_no_fold.json bench_c3_g46_mixed_prec.json bench_c3_g46_host_eigensolver.json bench_c3_g46_under_rocprof.json bench_c3_g64_scale_check.json \
         bench_c2_dense4096.json bench_c4_c5.json c3_whole_call.txt c1_whole_call.txt hermitian_c3.txt hermitian_c3_no_adjoint_fold.txt other_calls.txt per_rank_load_single_gpu.txt \
         factor_trace_c3_g46.txt launch_profile_16_nodes.txt conversion_exposure.txt analysis_trace_c3_g46.txt analysis_probe_c3_g46.txt smalleig_bench.txt c5_rr_step.txt; do
  [ -s $O/$f ] && cp $O/$f profiles/r05_$f
done
cp $O/prof/r05_kernel_stats.csv profiles/r05_c3_g46_kernel_stats.csv
for c in c3 c2 c5; do cp $O/timeline_$c.txt profiles/r05_factor_timeline_$c.txt; done
cp $P/pmc_traffic.json profiles/r05_pmc_traffic.json
cp $P/pmc_per_launch_traffic.txt profiles/r05_pmc_per_launch_traffic.txt
ls profiles | grep r05
