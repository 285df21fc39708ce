#!/bin/bash
# usage: tools/prof_round.sh <rNN>      (run on the GPU box from the repo root)
# rocprofv3 evidence of one round, written under gpurun_out/ (copy what is cited into profiles/):
#   <rNN>_bench.json, <rNN>_bench_kernel_stats.csv    default bench.py line + kernel trace stats of the SAME command
#   <rNN>_<cfg>_kernel_stats.csv, <rNN>_bench_<cfg>.json   c4split, c5, c2, c3
#   <rNN>_pmc_<counter>.csv                           per-kernel PMC sums (separate passes, counters only)
tag=${1:-r04}
repo=$(pwd); mkdir -p "$repo/gpurun_out"
cd /tmp && export TMPDIR=/tmp

stats() {   # $1 = name, rest = bench args
  name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o p -- python "$repo/bench.py" "$@" \
      > "$repo/gpurun_out/${tag}_bench_${name}.json" 2> "$repo/gpurun_out/${tag}_bench_${name}.err"
  f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$repo/gpurun_out/${tag}_${name}_kernel_stats.csv"
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:10]:
    print("%-50s calls %5s total %9.2f ms avg %9.3f ms" % (r["Name"][:50], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
}

pmc() {     # $1 = counter list
  ctr=$1; n=$(echo $ctr | tr ' ' '_')
  rm -rf /tmp/pmc_$n
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$n -o p -- python "$repo/bench.py" \
      --cpu-sample 0 --steps 1 --warmup 1 --no-primal --no-configs > /tmp/pmc_$n.log 2>&1
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] || { echo "no counter csv for $ctr"; tail -3 /tmp/pmc_$n.log; return; }
  python - "$f" "$repo/gpurun_out/${tag}_pmc_$n.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
with open(sys.argv[2], "w") as f:
    f.write("Kernel,Dispatches,Counter,SumOverDispatches\n")
    for (k, c), v in sorted(acc.items()):
        if any(s in k for s in ("k_xprod", "k_gram4", "k_urot", "k_small", "k_nt_gemm", "k_split_fused", "k_sd_")):
            f.write('"%s",%d,%s,%.1f\n' % (k, len(disp[k]), c, v)); print(k[:60], len(disp[k]), c, v)
PY
}

echo "== default bench"; stats c4 --steps 5 --warmup 2 --no-configs
echo "== c4split"; stats c4split --config c4split --steps 2 --warmup 1
echo "== c5"; stats c5 --config c5 --steps 2 --warmup 1
echo "== c2"; stats c2 --config c2 --steps 3 --warmup 1
echo "== c3"; stats c3 --config c3 --steps 3 --warmup 1
echo "== pmc"; pmc FETCH_SIZE; pmc WRITE_SIZE; pmc "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
echo "== traffic"; cd "$repo"; for c in c4 c3 c5 c4split c2; do tools/traffic.sh $c $tag; done
