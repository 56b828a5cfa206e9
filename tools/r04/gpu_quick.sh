#!/bin/bash
# quick loop of round 4: selected GPU tests + the bench line with the Rayleigh-Ritz trace (usage: gpurun -- bash tools/r04/gpu_quick.sh <tag> "<pytest -k expression>")
tag=${1:-r04q}; sel=${2:-"device_reduced or rr_device_path"}
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -q -k "$sel" > $out/tests.log 2>&1; echo "tests rc=$?"; tail -4 $out/tests.log
FEASTHIP_RR_TRACE=1 timeout -k 10 300 python bench.py --cpu-baseline off > $out/bench.log 2> $out/bench.err; echo "bench rc=$?"
python3 - $out/bench.log <<'PY'
import json, sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; d=json.loads(l)
print(d['value'], d['breakdown_ms_per_step'], 'rr_ms(event)', d['result']['rr_ms'], 'frac', d['roofline']['frac'])
PY
grep "on the device" $out/bench.err | tail -2
