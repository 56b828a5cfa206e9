#!/bin/bash
# Rebuilds kernels_solve.hip with ring-depth / occupancy variants of k_apply2 and times the default bench with each (run on the GPU box through
# gpurun; it leaves the LAST variant built).  DA4 / DA2 = factor k-steps in flight of the 64-column-per-wave / narrower bodies (must divide 8),
# OCCN / OCCG = workgroups per CU of the 32-row contiguous / gathered instances.  Round 5 on C3: "8 8 3 3" (rounds 2-4) 58.7, "8 4 4 3" 59.7,
# "4 4 4 3" (default now) 60.3, "4 4 4 4" 60.0, OCC 5 spills (26-33 iterations/s).
cd feastsolver.jl_amd/csrc
IFS=';'
for v in ${VARIANTS:-8 8 3 3;8 4 4 3;4 4 4 3;4 4 4 4}; do
  IFS=' ' read -r da4 da2 occn occg <<< "$v"
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function -fcx-limited-range -DFEASTHIP_A2_DA4=$da4 -DFEASTHIP_A2_DA2=$da2 -DFEASTHIP_A2_OCCN=$occn -DFEASTHIP_A2_OCCG=$occg -c kernels_solve.hip -o kernels_solve.o 2>/dev/null && make ARCH=gfx950 > /dev/null 2>&1
  echo "DA4=$da4 DA2=$da2 OCCN=$occn OCCG=$occg"
  (cd ../..; python tools/knob_sweep.py "" "" 2>&1 | cut -c52-140)
done
