#!/bin/bash
# the GPU suite under the non-default schedules (every combination must give the same results to rounding)
O=gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rc=0
run() { echo "== $*"; env "$@" timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3; [ ${PIPESTATUS[0]} -eq 0 ] || rc=1; }
run FEASTHIP_LOOKAHEAD=0 FEASTHIP_FIN_ASYNC=0
run FEASTHIP_NOPIV=0 FEASTHIP_FUSE_EXTEND=0
run FEASTHIP_FOLD=0 FEASTHIP_STREAMS=2 FEASTHIP_NB256_MINP=1073741824
run FEASTHIP_FWD_SPLITK=1 FEASTHIP_FIN_MASK=1 FEASTHIP_SHORTK_MAXP=0 FEASTHIP_NB512_MINP=300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
exit $rc
