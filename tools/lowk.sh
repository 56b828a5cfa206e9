#!/bin/bash
# Per-rank load of the multi-GPU run emulated on one GPU: 8 / 4 / 2 local contour nodes (N = 2 / 4 / 8 ranks).
for nodes in 8 4 2; do
timeout -k 10 200 python3 bench.py --nodes $nodes --steps 10 --warmup 2 --cpu-baseline off 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('nodes',$nodes,'ms/step %.2f contour %.2f rr %.2f'%(d['ms_per_step'], d['roofline']['contour_step']['ms'], d['result']['rr_ms']))"
done
