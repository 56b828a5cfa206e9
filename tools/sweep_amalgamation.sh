mkdir -p gpurun_out
for cfg in "16 96 0.35 0.05" "32 96 0.35 0.05" "32 128 0.5 0.1" "48 160 0.5 0.1" "64 192 0.6 0.15"; do
  set -- $cfg
  FEASTHIP_RELAX_SMALL=$1 FEASTHIP_RELAX_MID=$2 FEASTHIP_ZMID=$3 FEASTHIP_ZBIG=$4 timeout -k 10 200 python bench.py --steps 4 --warmup 1 --cpu-baseline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', round(d['value'],2), round(d['roofline']['contour_step']['ms'],2), d['config']['nnz_L'], d['config']['supernodes'], round(d['setup']['factor_ms']), {k:round(v,2) for k,v in d['roofline']['kernel_classes_ms'].items()})"
done
