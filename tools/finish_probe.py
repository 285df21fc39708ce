#!/usr/bin/env python3
"""Pieces of the front-end's finish phase at the c4 shape (10 000 bootstraps), each timed alone: the transpose of the
gathered distributions, the percentile intervals, the bootstrap ratios, the D2H copies into pinned memory."""
import sys
import time

import torch

sys.path.insert(0, '.')
from pypyls_amd.engine import Engine               # noqa: E402

eng = Engine()
dev = eng.device
B, L, Tp, n = 200000, 50, 50, 10000
d_full = torch.rand((n, Tp, L), dtype=torch.float64, device=dev)
xw = torch.rand((B, L), dtype=torch.float64, device=dev)
sv = torch.rand((L,), dtype=torch.float64, device=dev)
usum = torch.rand((B, L), dtype=torch.float64, device=dev)
usq = usum * usum + 1
eng.S, eng.B, eng.L, eng.Tp = 500, B, L, Tp
pin = {k: eng.pinned_like(xw) for k in ('xw', 'bsr', 'se')}
pin['dist'] = torch.empty((Tp * L, n), dtype=torch.float64, pin_memory=True)


def t(name, fn, reps=3):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        dt = 1e3 * (time.perf_counter() - t0)
        best = dt if best is None else min(best, dt)
    print('%-28s %.2f ms' % (name, best))
    return r


ser = t('transpose 200MB', lambda: eng.transpose_dev(d_full.reshape(n, Tp * L)))
t('percentile 2500 x 10000', lambda: eng.percentile_ci_dev(ser, ci=95))
bs = t('scale_columns', lambda: eng.scale_columns(xw, sv))
bsr, se = t('boot_rel', lambda: eng.boot_rel_dev(bs, usum, usq, n + 1, add_orig=True))
t('d2h dist 200MB pinned', lambda: eng.to_host_async(ser, pin['dist']))
t('d2h 3 x 80MB pinned', lambda: [eng.to_host_async(a, pin[k]) for a, k in ((xw, 'xw'), (bsr, 'bsr'), (se, 'se'))])
t('perm block cpu() 10000x50', lambda: torch.rand((10000, 50), dtype=torch.float64, device=dev).cpu().numpy().T)
