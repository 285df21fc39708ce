repo=$(pwd); mkdir -p gpurun_out
python bench.py --config c5 --steps 3 --warmup 1 > gpurun_out/r05_bench_c5.json 2> /dev/null
python bench.py --config c5 --perms 1000 --boots 1000 --steps 3 --warmup 1 > gpurun_out/r05_bench_c5_1000.json 2>/dev/null
python bench.py --mode analysis --config c5 > gpurun_out/r05_bench_analysis_c5.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_c5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -o p -- python $repo/bench.py --config c5 --steps 2 --warmup 1 > $repo/gpurun_out/r05_bench_c5_profiled.json 2>/dev/null
f=$(find /tmp/prof_c5 -name "*kernel_stats.csv" | head -1); cp "$f" $repo/gpurun_out/r05_c5_kernel_stats.csv
head -12 $repo/gpurun_out/r05_c5_kernel_stats.csv | cut -c1-150
cd $repo
for f in r05_bench_c5 r05_bench_c5_1000 r05_bench_analysis_c5 r05_bench_c5_profiled; do python -c "
import json,sys; d=json.load(open('gpurun_out/$f.json')); print('$f', d['value'], d['ms_per_step'])"; done
