#!/bin/bash
# usage: [BENCH_ARGS="--config c4split"] tools/pmc_pass.sh <tag> "<counters>"   -- one rocprofv3 --pmc pass over a short bench run
# (counters only, with --kernel-trace: no sys/hip/hsa traces); per-kernel sums -> gpurun_out/<tag>_pmc.csv
tag=$1; ctr=$2
repo=$(pwd); mkdir -p "$repo/gpurun_out"
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmc_$tag
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$tag -o p -- python "$repo/bench.py" --cpu-sample 0 --steps 1 --warmup 1 ${BENCH_ARGS:---perms 560 --boots 560} > /tmp/pmc_$tag.log 2>&1
f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
[ -n "$f" ] || { echo "no counter csv"; tail -5 /tmp/pmc_$tag.log; exit 1; }
python - "$f" "$repo/gpurun_out/${tag}_pmc.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
with open(sys.argv[2], "w") as f:
    f.write("Kernel,Dispatches,Counter,SumOverDispatches\n")
    for (k, c), v in sorted(acc.items()):
        if any(s in k for s in ("k_xprod", "k_gram4", "k_urot", "k_small", "k_nt_gemm", "k_ucorr", "k_sd_")):
            f.write('"%s",%d,%s,%.1f\n' % (k, len(disp[k]), c, v)); print(k[:40], len(disp[k]), c, v)
PY
