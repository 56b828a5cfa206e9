#!/bin/bash
tag=${1:-r04lowk2}; out=gpurun_out/$tag; mkdir -p $out
for env in "FEASTHIP_FOLD=1" "FEASTHIP_FOLD=0" "FEASTHIP_FOLD=0 FEASTHIP_STREAMS=1" "FEASTHIP_FOLD=1 FEASTHIP_STREAMS=1" "FEASTHIP_FOLD=1 FEASTHIP_FWD_SPLITK=1"; do
  echo "== $env"; env $env timeout -k 10 300 python tools/few_units.py 2>&1 | grep world
done
