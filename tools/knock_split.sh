for k in ${KNOCKS:-0 1 2 4 3 7}; do
  PLSX_SPLIT_KNOCK=$k timeout 200 python bench.py --config c4split --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('knock $k', round(d['config']['kernel_ms_per_step']['k_ucorr_partial']/8,3), 'ms per 100 splits')"
done
