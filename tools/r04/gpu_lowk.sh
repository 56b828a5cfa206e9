#!/bin/bash
# per-rank load on one GPU (8/4/2/1 units) + per-launch profile at one unit (K = 2 nodes = one folded pair)
tag=${1:-r04lowk}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 300 python tools/few_units.py > $out/few_units.log 2>&1; echo "few_units rc=$?"; cat $out/few_units.log
timeout -k 10 300 python tools/launch_profile.py 46 2 > $out/launch_profile_1unit.log 2>&1; echo "launch_profile rc=$?"; grep -n "apply launches\|other launches" $out/launch_profile_1unit.log
