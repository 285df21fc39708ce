#!/bin/bash
# usage (GPU box, repo root): tools/sd_variants_prof.sh <variant> ...   -- per-kernel times (rocprofv3 kernel stats) of
# `bench.py --config c5` under the SIMPLS solver variants of tools/sd_variants.sh
repo=$(pwd); mkdir -p gpurun_out
cp pypyls_amd/libplsx.so /tmp/libplsx_base.so
cd /tmp && export TMPDIR=/tmp
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/libplsx_base.so $repo/pypyls_amd/libplsx.so; else cp $repo/pypyls_amd/variants/libplsx_$v.so $repo/pypyls_amd/libplsx.so; fi
  rm -rf /tmp/prof_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o p -- python $repo/bench.py --config c5 --steps 3 --warmup 1 --cpu-sample 0 > /tmp/prof_$v.json 2> /tmp/prof_$v.err
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    if 'mfma_peak' in r["Name"]: continue
    print("%-46s calls %5s total %9.2f ms avg %9.3f ms" % (r["Name"][:46], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
done
cp /tmp/libplsx_base.so $repo/pypyls_amd/libplsx.so
