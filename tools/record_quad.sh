#!/bin/bash
# usage: tools/record_quad.sh <rNN>   (GPU box, repo root): records of the configs the quadratic-form route of the
# bootstrap sums changed (c3, c5): clean bench lines, the same with the route off, rocprofv3 kernel stats, HBM traffic,
# the end-to-end public calls
tag=${1:-r04}
repo=$(pwd); mkdir -p gpurun_out
run() { name=$1; shift; python bench.py "$@" > gpurun_out/${tag}_bench_${name}.json 2> gpurun_out/${tag}_bench_${name}.err; head -c 300 gpurun_out/${tag}_bench_${name}.json; echo; }
python -m pytest tests/test_gpu_multirank.py -q -x -k "two_ranks_equal or config_lines or analysis" 2>&1 | tail -3
run c3 --config c3 --steps 5 --warmup 2
run c5 --config c5 --steps 3 --warmup 1
run c5_1000 --config c5 --perms 1000 --boots 1000 --steps 5 --warmup 2 --cpu-sample 0
PLSX_QUAD_SUMS=-1 run c3_per_bootstrap_pass --config c3 --steps 3 --warmup 1 --cpu-sample 0
PLSX_QUAD_SUMS=-1 run c5_per_bootstrap_pass --config c5 --steps 2 --warmup 1 --cpu-sample 0
run analysis_c3 --mode analysis --config c3 --steps 3
python tools/bench_configs.py c3 c5 > gpurun_out/${tag}_frontend_walltimes_quad.jsonl 2>/dev/null; cat gpurun_out/${tag}_frontend_walltimes_quad.jsonl
cd /tmp && export TMPDIR=/tmp
for cfg in c3 c5; do
  rm -rf /tmp/prof_$cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -o p -- python "$repo/bench.py" --config $cfg --steps 2 --warmup 1 --cpu-sample 0 \
      > "$repo/gpurun_out/${tag}_bench_${cfg}_profiled.json" 2> /dev/null
  f=$(find /tmp/prof_$cfg -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$repo/gpurun_out/${tag}_${cfg}_kernel_stats.csv" && head -8 "$f" | cut -c1-150
done
cd "$repo"; for c in c3 c5; do tools/traffic.sh $c $tag; done
