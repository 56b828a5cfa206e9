#!/bin/bash
# A/B of factorisation schedule switches (factor_ms of the second factorisation of tools/factor_trace.py; panel steps of depth 3..0)
O=gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { echo "== $*"; env "$@" python tools/factor_trace.py 2> $O/ab_trace.txt | grep -o "'factor_ms': [0-9.]*"; grep "depth  [0-3] " $O/ab_trace.txt | tail -4 | sed 's/.*panel steps *\([0-9.]*\).*/\1/' | tr '\n' ' '; echo; }
run FEASTHIP_FUSE_EXTEND=128
run FEASTHIP_FUSE_EXTEND=64
run FEASTHIP_FUSE_EXTEND=48
run FEASTHIP_FUSE_EXTEND=1
run FEASTHIP_FUSE_EXTEND=128
run FEASTHIP_NOPIV=0 FEASTHIP_FUSE_EXTEND=128
run FEASTHIP_NOPIV=0 FEASTHIP_FUSE_EXTEND=48
run FEASTHIP_NOPIV=0 FEASTHIP_FUSE_EXTEND=1
