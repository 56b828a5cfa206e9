#!/bin/bash
# contour-step time for 1 / 2 node groups (streams), with and without the conjugate fold, for the super-block thresholds given in the environment
for f in "" "--no-fold"; do for s in 1 2; do
  echo "== streams $s $f NB256_MINP=${FEASTHIP_NB256_MINP:-default} NB512_MINP=${FEASTHIP_NB512_MINP:-default}"
  FEASTHIP_STREAMS=$s python bench.py --cpu-baseline off --steps 6 $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), d['breakdown_ms_per_step'], d['setup']['factor_ms'])"
done; done
