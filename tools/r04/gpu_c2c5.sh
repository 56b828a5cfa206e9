#!/bin/bash
# C2 and C4/C5 with the Rayleigh-Ritz and host-eigensolver traces
tag=${1:-r04c2c5}; out=gpurun_out/$tag; mkdir -p $out
export FEASTHIP_RR_TRACE=1 FEASTHIP_EIG_TRACE=1
timeout -k 10 300 python tools/bench_c2.py > $out/c2.log 2> $out/c2.err; echo "c2 rc=$?"; tail -2 $out/c2.log | cut -c1-1500; grep -c . $out/c2.err; grep "feasthip" $out/c2.err | tail -6
timeout -k 10 300 python tools/bench_c4c5.py > $out/c4c5.log 2> $out/c4c5.err; echo "c4c5 rc=$?"; tail -1 $out/c4c5.log | cut -c1-1500; grep "feasthip" $out/c4c5.err | tail -8
