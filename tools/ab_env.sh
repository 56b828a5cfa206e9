#!/bin/bash
# A/B of the contour step over values of one environment switch, interleaved so that clock drift shows:
#   tools/ab_env.sh FEASTHIP_A2_WGS "0 3 6" [reps] [extra bench.py args]
var=$1; vals=$2; reps=${3:-2}; shift 3
for rep in $(seq 1 $reps); do
for v in $vals; do
  echo -n "rep $rep $var=$v: "
  env $var=$v python bench.py --cpu-baseline off --steps 8 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), 'it/s contour', round(d['breakdown_ms_per_step']['contour'],2), 'ms  k_apply', round(d['roofline']['kernel_classes_ms']['panel_apply'],2), 'ms')"
done; done
