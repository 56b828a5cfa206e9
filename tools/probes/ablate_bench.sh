#!/bin/bash
# Builds tools/probes/apply_bench with each FEASTHIP_ABLATE mask ON THE GPU BOX and times the synthetic levels (timing only).
cd "$(dirname "$0")"
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fcx-limited-range -Wno-unused-result -DFEASTHIP_ABLATE=$v -I../../feastsolver.jl_amd/csrc apply_bench.hip -o apply_bench_$v 2>/dev/null || exit 1
  echo "=== ablate mask $v"
  timeout -k 10 100 ./apply_bench_$v | grep -v "^occupancy\|regs" || exit 1
done
