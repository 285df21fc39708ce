#!/bin/bash
# usage: tools/record_round.sh <rNN>   (GPU box, repo root): clean bench lines of every config for the record
tag=${1:-r04}
mkdir -p gpurun_out
run() { name=$1; shift; python bench.py "$@" > gpurun_out/${tag}_bench_${name}.json 2> gpurun_out/${tag}_bench_${name}.err; head -c 400 gpurun_out/${tag}_bench_${name}.json; echo; }
run c4 --steps 20 --warmup 5
run c4_strong --mode strong --steps 3 --warmup 1
run c4_strong_emulation --mode strong --steps 2 --warmup 1 --emulate-world 1,2,4,8 --cpu-sample 0
run c2 --config c2 --steps 10 --warmup 2
run c3 --config c3 --steps 5 --warmup 2
run c5 --config c5 --steps 3 --warmup 1
run c5_1000 --config c5 --perms 1000 --boots 1000 --steps 5 --warmup 2 --cpu-sample 0
run c4split --config c4split --steps 3 --warmup 1
run analysis_c4_emulation --mode analysis --steps 3 --emulate-world 1,2,4,8
run analysis_c2 --mode analysis --config c2 --steps 5
run analysis_c3 --mode analysis --config c3 --steps 3
python tools/bench_configs.py c2 c3 c5 c4 c4split c4cv > gpurun_out/${tag}_frontend_walltimes.jsonl 2>/dev/null; cat gpurun_out/${tag}_frontend_walltimes.jsonl
rm -f gpurun_out/${tag}_wide.jsonl
for cfg in "400 50000 72 1 1 1024" "400 50000 100 1 1 512" "420 50000 140 1 1 512" "400 50000 200 1 1 256" "100 1000 100 1 4 64"; do
  python tools/bench_wide.py $cfg >> gpurun_out/${tag}_wide.jsonl 2>/dev/null
done
cut -c1-300 gpurun_out/${tag}_wide.jsonl
