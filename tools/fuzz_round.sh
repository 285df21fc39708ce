#!/bin/bash
# usage: tools/fuzz_round.sh <rNN>   (GPU box, repo root): the randomised sweeps of a round -> gpurun_out/<rNN>_fuzz.txt
tag=${1:-r05}; out=gpurun_out/${tag}_fuzz.txt; mkdir -p gpurun_out; : > $out
run() { echo "== $*" >> $out; "$@" 2>&1 | grep -a "FAIL\|failures" | cut -c1-400 >> $out; }
run timeout 1500 python tools/fuzz_parity.py 1000 9001
PLSX_QUAD_SUMS=1 run timeout 900 python tools/fuzz_parity.py 400 9002
FUZZ_WIDE=1 run timeout 900 python tools/fuzz_parity.py 120 9003
run timeout 1200 python tools/fuzz_graded.py 300 9004
FUZZ_WIDE=1 run timeout 1500 python tools/fuzz_graded.py 120 9005
cat $out
