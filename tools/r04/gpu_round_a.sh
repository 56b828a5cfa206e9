#!/bin/bash
# round 4, first GPU call: the new tests alone, then the whole GPU suite, then the bench line with the Rayleigh-Ritz trace
out=gpurun_out/r04b; mkdir -p $out
step() { name=$1; shift; timeout -k 10 "$T" "$@" > $out/$name.log 2> $out/$name.err; rc=$?; echo "$name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed: stopping"; exit 1; fi; }
T=600 step new_tests python -m pytest tests -m gpu -q -k "device_reduced or rr_device_path or moderate_growth or hermitian_factorisation_is_race or hermitian_c3_full or exits_cleanly"
tail -5 $out/new_tests.log
T=1000 step full_suite python -m pytest tests -m gpu -x -q
tail -3 $out/full_suite.log
export FEASTHIP_RR_TRACE=1
T=300 step bench python bench.py --cpu-baseline off
tail -1 $out/bench.log | cut -c1-600
grep -c "on the device" $out/bench.err; grep "reduced eigenproblem" $out/bench.err | tail -3
