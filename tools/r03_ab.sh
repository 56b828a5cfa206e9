#!/bin/bash
# A/B of factorisation schedule switches (factor_ms of the second factorisation of tools/factor_trace.py)
O=gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { echo "== $*"; env "$@" python tools/factor_trace.py 2> $O/ab_trace.txt | grep -o "'factor_ms': [0-9.]*"; grep "depth  [0-3] " $O/ab_trace.txt | tail -4 | cut -c1-60,100-200; }
run FEASTHIP_X=0
run FEASTHIP_FIN_MASK=1
run FEASTHIP_FIN_ASYNC=0
run FEASTHIP_RESERVE_CUS=4
run FEASTHIP_RESERVE_CUS=4 FEASTHIP_FIN_MASK=1
run FEASTHIP_LOOKAHEAD=0
