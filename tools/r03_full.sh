#!/bin/bash
# whole GPU suite (default schedule and the fold-off / two-stream / 128-column-block one), smoke, default bench line
set -o pipefail
O=gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests_full.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests_full.txt
tail -4 $O/gpu_tests_full.txt
FEASTHIP_FOLD=0 FEASTHIP_STREAMS=2 FEASTHIP_NB256_MINP=1073741824 timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/bench_c3_g46.json 2> $O/bench_c3_g46.err || exit 1
echo done
