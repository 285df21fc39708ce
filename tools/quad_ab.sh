mkdir -p gpurun_out/r4q
python -m pytest tests/test_gpu_quad.py -q 2>&1 | tail -3
for cfg in c3 c5; do
  for q in 0 -1; do
    PLSX_QUAD_SUMS=$q python bench.py --config $cfg --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 > gpurun_out/r4q/${cfg}_q${q}.json
  done
done
for q in 0 -1; do
  PLSX_QUAD_SUMS=$q python bench.py --config c5 --perms 5000 --boots 5000 --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 > gpurun_out/r4q/c5full_q${q}.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4q/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value']), round(d['ms_per_step'],2), d['config'].get('kernel_ms_per_step'), d['config'].get('boot_ms_per_step'))
    except Exception as e: print(f, 'ERR', e)
PY
