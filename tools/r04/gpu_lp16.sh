#!/bin/bash
out=gpurun_out/${1:-r04lp}; mkdir -p $out
timeout -k 10 300 python tools/launch_profile.py 46 16 > $out/launch_profile_16.log 2>&1; echo rc=$?
grep -n "apply launches\|other launches" $out/launch_profile_16.log
