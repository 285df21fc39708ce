#!/bin/bash
# usage: tools/pmc_wide.sh "<counters>" [bench_wide args]   (GPU box, repo root): per-kernel PMC sums of a wide-T' run
ctr=$1; shift
repo=$(pwd); cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcw
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmcw -o p -- python "$repo/tools/bench_wide.py" "$@" > /tmp/pmcw.log 2>&1
f=$(find /tmp/pmcw -name "*counter_collection.csv" | head -1)
[ -n "$f" ] || { echo "no counter csv"; tail -5 /tmp/pmcw.log; exit 1; }
python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for (k, c), v in sorted(acc.items()):
    if any(s in k for s in ("k_xprod", "k_gram", "k_urot", "k_small")):
        print("%-40s %3d %-36s %.4g" % (k[:40], len(disp[k]), c, v))
PY
