#!/bin/bash
# Supernode granularity sweep: leaf size of the nested dissection and the relaxed-amalgamation thresholds.
mkdir -p gpurun_out
for cfg in "24 32 96 0.35 0.05" "32 32 96 0.35 0.05" "48 32 96 0.35 0.05" "64 32 96 0.35 0.05" "32 48 128 0.4 0.08" "48 64 128 0.4 0.08" "64 64 160 0.5 0.1" "96 64 160 0.5 0.1"; do
  set -- $cfg
  FEASTHIP_LEAF=$1 FEASTHIP_RELAX_SMALL=$2 FEASTHIP_RELAX_MID=$3 FEASTHIP_ZMID=$4 FEASTHIP_ZBIG=$5 timeout -k 10 200 python3 bench.py --steps 6 --warmup 2 --cpu-baseline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', round(d['value'],2), round(d['roofline']['contour_step']['ms'],2), d['config']['nnz_L'], d['config']['supernodes'], round(d['setup']['factor_ms']), {k:round(v,2) for k,v in d['roofline']['kernel_classes_ms'].items()})"
done
