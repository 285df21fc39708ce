#!/bin/bash
# usage: CFG=<c2|c3|c5|c4|c4split|c4cv> tools/prof_config.sh   (rocprofv3 kernel stats of one tools/bench_configs.py config)
repo=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_c5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -o p -- python "$repo/tools/bench_configs.py" ${CFG:-c5} > /tmp/c5.log 2>&1
grep "^{" /tmp/c5.log | cut -c1-200
f=$(find /tmp/prof_c5 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-44s calls %5s total %9.2f ms avg %9.3f ms" % (r["Name"][:44], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
