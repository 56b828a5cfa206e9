#!/bin/bash
# k_apply2 ablations (FEASTHIP_A2_ABLATE: 1 no factor loads, 2 no block-vector loads, 4 no MFMA) on the few-unit synthetic levels
cd "$(dirname "$0")"
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fcx-limited-range -Wno-unused-result -DFEASTHIP_A2_ABLATE=$v -I../../feastsolver.jl_amd/csrc apply_bench.hip -o apply_bench_a2_$v 2>/dev/null || exit 1
  echo "=== k_apply2 ablate mask $v"
  timeout -k 10 100 ./apply_bench_a2_$v few | grep "kind 1" | grep -v "mid-1" || exit 1
done
