#!/bin/bash
# Supernode granularity sweep: leaf size of the nested dissection and the relaxed-amalgamation thresholds
# (leaf, relax_small, relax_mid, zmid, zbig; first line = defaults of symbolic.hpp).
mkdir -p gpurun_out
for cfg in "64 32 64 0.35 0.05" "64 48 96 0.35 0.05" "64 64 128 0.35 0.05" "96 64 128 0.4 0.08" "128 64 128 0.35 0.05" "128 96 160 0.4 0.08" "64 32 128 0.5 0.1" "48 32 64 0.35 0.05" "192 128 192 0.4 0.08" "64 32 64 0.35 0.05"; do
  set -- $cfg
  FEASTHIP_LEAF=$1 FEASTHIP_RELAX_SMALL=$2 FEASTHIP_RELAX_MID=$3 FEASTHIP_ZMID=$4 FEASTHIP_ZBIG=$5 timeout -k 10 200 python3 bench.py --steps 8 --warmup 3 --cpu-baseline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', round(d['value'],2), round(d['roofline']['contour_step']['ms'],2), d['config']['nnz_L'], d['config']['supernodes'], round(d['setup']['factor_ms']), {k:round(v,2) for k,v in d['roofline']['kernel_classes_ms'].items()})"
done
