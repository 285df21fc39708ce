#!/bin/bash
# usage: tools/pmc_cfg.sh <cfg> "<counters>" [tag]   (GPU box, repo root): per-kernel PMC sums of `bench.py --config <cfg>`
# (counters only, with --kernel-trace; one pass per call) -> gpurun_out/<tag>_pmc_<cfg>_<counters>.csv
cfg=$1; ctr=$2; tag=${3:-r04}
repo=$(pwd); mkdir -p "$repo/gpurun_out"
n=$(echo $ctr | tr ' ' '_')
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_${cfg}_$n
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_${cfg}_$n -o p -- python "$repo/bench.py" \
    --config $cfg --cpu-sample 0 --steps 1 --warmup 1 --no-primal > /tmp/pmc_${cfg}_$n.log 2>&1
f=$(find /tmp/pmc_${cfg}_$n -name "*counter_collection.csv" | head -1)
[ -n "$f" ] || { echo "no counter csv for $ctr"; tail -3 /tmp/pmc_${cfg}_$n.log; exit 1; }
python - "$f" "$repo/gpurun_out/${tag}_pmc_${cfg}_$n.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
with open(sys.argv[2], "w") as f:
    f.write("Kernel,Dispatches,Counter,SumOverDispatches\n")
    for (k, c), v in sorted(acc.items()):
        if k.startswith("k_"):
            f.write('"%s",%d,%s,%.1f\n' % (k, len(disp[k]), c, v))
            if any(s in k for s in ("k_xprod", "k_nt_gemm", "k_sd_step", "k_dual_gp", "k_split_fused", "k_gram4", "k_ucorr", "k_urot")): print(k[:44], len(disp[k]), c, v)
PY
