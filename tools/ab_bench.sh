#!/bin/bash
# A/B of the contour step: apply kernel (FEASTHIP_APPLY2) x conjugate fold x node groups, interleaved twice so that clock drift shows
for rep in 1 2; do
for cfg in "1 1 " "0 1 " "1 1 --no-fold" "0 1 --no-fold" "1 2 --no-fold" "0 2 --no-fold" "1 2 " ; do
  set -- $cfg
  echo -n "rep $rep APPLY2=$1 STREAMS=$2 $3: "
  FEASTHIP_APPLY2=$1 FEASTHIP_STREAMS=$2 python bench.py --cpu-baseline off --steps 8 $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), 'it/s contour', round(d['breakdown_ms_per_step']['contour'],2), 'ms  k_apply', round(d['roofline']['kernel_classes_ms']['panel_apply'],2), 'ms')"
done; done
