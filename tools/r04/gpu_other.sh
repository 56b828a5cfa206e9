#!/bin/bash
out=gpurun_out/${1:-r04p}; mkdir -p $out
timeout -k 10 400 python tools/other_calls.py > $out/other.log 2> $out/other.err; echo rc=$?; tail -4 $out/other.log
FEASTHIP_DUAL_DOUBLED=1 timeout -k 10 400 python tools/other_calls.py > $out/other_doubled.log 2> $out/other_doubled.err; echo rc=$?; tail -2 $out/other_doubled.log
