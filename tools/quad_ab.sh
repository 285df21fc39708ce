#!/bin/bash
# A/B of the quadratic-form route of the bootstrap sums (PLSX_QUAD_SUMS=-1: the per-bootstrap feature pass) on c3 / c5
mkdir -p gpurun_out/r4q
python -m pytest tests/test_gpu_quad.py tests/test_gpu_regression.py -q 2>&1 | tail -3
for cfg in c3 c5; do
  for q in 0 -1; do
    PLSX_QUAD_SUMS=$q python bench.py --config $cfg --steps 3 --warmup 1 --cpu-sample 0 2>gpurun_out/r4q/${cfg}_q${q}.err | tail -1 > gpurun_out/r4q/${cfg}_q${q}.json
  done
done
PLSX_QUAD_SUMS=0 python bench.py --config c5 --perms 1000 --boots 1000 --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 > gpurun_out/r4q/c5_1000_q0.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4q/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f, round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['config'].get('kernel_ms_per_step').items()}, d['config'].get('boot_ms_per_step'))
        print('   ', r.get('kernel')[:60], 'frac', r.get('frac'), 'issued', r.get('frac_issued'), 'pipe', r.get('pipeline_frac_mfma'))
    except Exception as e: print(f, 'ERR', e)
PY
