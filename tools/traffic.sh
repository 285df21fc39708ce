#!/bin/bash
# usage: tools/traffic.sh <cfg> [tag]    (GPU box, repo root; cfg = c4 | c3 | c5 | c4split | c2)
# HBM bytes per launch of the dominant kernel of `bench.py --config <cfg>` from two SEPARATE rocprofv3
# counter passes (FETCH_SIZE, then WRITE_SIZE; counters only, with --kernel-trace -- MI355X_MICROARCH.md,
# section HBM / rocprofv3) -> gpurun_out/traffic_<cfg>.json (copy to profiles/ for bench.py to report it).
cfg=${1:-c4}; tag=${2:-r04}
repo=$(pwd); mkdir -p "$repo/gpurun_out"
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tr_${cfg}_$ctr
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/tr_${cfg}_$ctr -o p -- python "$repo/bench.py" \
      --config $cfg --cpu-sample 0 --steps 1 --warmup 1 --no-primal > /tmp/tr_${cfg}_$ctr.log 2>&1
done
python "$repo/tools/make_traffic.py" $cfg "$(find /tmp/tr_${cfg}_FETCH_SIZE -name '*counter_collection.csv' | head -1)" \
    "$(find /tmp/tr_${cfg}_WRITE_SIZE -name '*counter_collection.csv' | head -1)" "$repo/gpurun_out/traffic_${cfg}.json"
