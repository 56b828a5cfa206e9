#!/bin/bash
# Rebuilds kernels_solve.o with each FEASTHIP_ABLATE mask (k_apply, the LDS-staged kernel: 1 no factor loads, 2 no LDS-DMA, 4 no epilogue,
# 8 no MFMA, 16 no barrier, 32 no LDS reads) ON THE GPU BOX and times the solve sweep with FEASTHIP_APPLY2=0.  Timing only: results are garbage.
# The same for k_apply2 on synthetic levels: tools/probes/ablate2_bench.sh (FEASTHIP_A2_ABLATE).
set -e
cd "$(dirname "$0")/../feastsolver.jl_amd/csrc"
for v in ${@:-0 1 2 4 8 16}; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function -fcx-limited-range -DFEASTHIP_ABLATE=$v -c kernels_solve.hip -o kernels_solve.o
  make -s ARCH=gfx950 ../libfeasthip.so
  FEASTHIP_APPLY2=0 timeout -k 10 120 python3 ../../tools/ablate.py 46 "ablate=$v"
done
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function -fcx-limited-range -c kernels_solve.hip -o kernels_solve.o
make -s ARCH=gfx950 ../libfeasthip.so
