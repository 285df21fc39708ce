#!/bin/bash
# k_xprod epilogue experiments (PLSX_TUNE bits: 1 skip R stores, 2 skip Rfull loads, 4 de-phase first round,
# 16 no LDS prefetch in the fused split epilogue, 32 non-temporal R stores, bits 8.. sleep units)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for t in ${TUNES:-0 1 32}; do
  for cfg in ${CFGS:-c4 c4split}; do
    extra="--no-primal --cpu-sample 0 --steps 3 --warmup 1"
    PLSX_TUNE=$t python bench.py --config $cfg $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('tune=$t $cfg value=%.1f ms/step=%.2f' % (d['value'], d['ms_per_step']), {k: round(v,2) for k,v in d['config']['kernel_ms_per_step'].items()})
"
  done
done
