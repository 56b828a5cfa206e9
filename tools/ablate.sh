#!/bin/bash
# Rebuilds kernels_solve.o with each FEASTHIP_ABLATE variant ON THE GPU BOX and times the solve sweep.
set -e
cd "$(dirname "$0")/../feastsolver.jl_amd/csrc"
for v in 0 1 2 3 4; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -fcx-limited-range -DFEASTHIP_ABLATE=$v -c kernels_solve.hip -o kernels_solve.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -pthread -o ../libfeasthip.so symbolic.o ordering.o smalleig.o kernels_factor.o kernels_solve.o kernels_rr.o engine.o api.o
  timeout -k 10 120 python3 ../../tools/ablate.py 46 "ablate=$v"
done
