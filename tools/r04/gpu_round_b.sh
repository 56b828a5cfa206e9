#!/bin/bash
# full GPU suite + bench + Hermitian C3 with and without the adjoint fold + the other calls (dual_gen_feast on C3)
out=gpurun_out/${1:-r04o}; mkdir -p $out
step() { name=$1; shift; timeout -k 10 "$T" "$@" > $out/$name.log 2> $out/$name.err; rc=$?; echo "$name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed: stopping"; exit 1; fi; }
T=1100 step full_suite python -m pytest tests -m gpu -x -q
tail -3 $out/full_suite.log
T=300 step bench python bench.py --cpu-baseline off
python3 - $out/bench.log <<'PY'
import json, sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; d=json.loads(l)
print(d['value'], d['breakdown_ms_per_step'], 'frac', d['roofline']['frac'], d['setup']['factor_ms'])
PY
T=300 step herm python tools/hermitian_c3.py; cat $out/herm.log
export FEASTHIP_ADJ_FOLD=0
T=300 step herm_nofold python tools/hermitian_c3.py; cat $out/herm_nofold.log
unset FEASTHIP_ADJ_FOLD
T=400 step other python tools/other_calls.py; tail -5 $out/other.log
