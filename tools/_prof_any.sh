#!/bin/bash
repo=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_any
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_any -o p -- python "$repo/$1" > /tmp/any.log 2>&1
tail -2 /tmp/any.log | cut -c1-200
f=$(find /tmp/prof_any -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-44s calls %5s total %9.2f ms avg %9.3f ms" % (r["Name"][:44], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
