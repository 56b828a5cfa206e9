#!/bin/bash
# Node groups on separate streams (FEASTHIP_STREAMS): 16 local nodes, C3 46^3.
for st in 1 2 3 4; do
FEASTHIP_STREAMS=$st timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --cpu-baseline off 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('streams',$st,'ms/step %.2f contour %.2f'%(d['ms_per_step'], d['breakdown_ms_per_step']['contour']))"
done
