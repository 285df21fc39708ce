#!/bin/bash
# usage: tools/prof_stats.sh <tag> [bench.py args...]   (run on the GPU box from the repo root)
# rocprofv3 kernel trace + stats of bench.py; summary -> gpurun_out/<tag>_kernel_stats.csv
tag=$1; shift
repo=$(pwd)
mkdir -p "$repo/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python "$repo/bench.py" "$@" > "$repo/gpurun_out/${tag}_bench.log" 2>&1
tail -1 "$repo/gpurun_out/${tag}_bench.log" | cut -c1-400
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] || { echo "no kernel_stats.csv"; find /tmp/prof_$tag | head; tail -5 "$repo/gpurun_out/${tag}_bench.log"; exit 1; }
cp "$f" "$repo/gpurun_out/${tag}_kernel_stats.csv"
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:16]:
    print("%-44s calls %5s total %9.2f ms avg %9.3f ms" % (r["Name"][:44], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
