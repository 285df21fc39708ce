#!/bin/bash
# usage (GPU box, repo root): tools/sd_variants.sh      -- A/B of the SIMPLS solver's register / occupancy variants
# (pypyls_amd/variants/libplsx_<name>.so, built with -DSD_RC=.. -DSD_WPE=..; see plsx_simpls.h) on `bench.py --config c5`.
mkdir -p gpurun_out
cp pypyls_amd/libplsx.so /tmp/libplsx_base.so
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/libplsx_base.so pypyls_amd/libplsx.so; else cp pypyls_amd/variants/libplsx_$v.so pypyls_amd/libplsx.so; fi
  python bench.py --config c5 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/sd_var_$v.json 2> gpurun_out/sd_var_$v.err
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.load(open('gpurun_out/sd_var_%s.json' % v))
    k = d['config']['kernel_ms_per_step']
    print('%-10s %8.1f resamples/s  %6.2f ms/step  solver %6.2f  nt %6.2f  xprod %6.2f' % (v, d['value'], d['ms_per_step'], k.get('k_simpls_dual', 0), k.get('k_nt_gemm', 0), k.get('k_xprod', 0)))
except Exception as e:
    print(v, 'failed', e)
PY
done
cp /tmp/libplsx_base.so pypyls_amd/libplsx.so
