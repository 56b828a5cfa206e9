for s in 1 2 3; do
  FEASTHIP_STREAMS=$s timeout -k 10 200 python bench.py --steps 8 --warmup 2 --cpu-baseline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams $s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],2), 'contour', round(d['roofline']['contour_step']['ms'],2), 'rr', round(d['result']['rr_ms'],2))"
done
