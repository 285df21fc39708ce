#!/usr/bin/env python3
"""Closing pass of the quadratic-form route (plsx_boot_finish) at the c5 shape, on its own: block height (option
quad_mt), full rows vs the upper triangle (quad_full_rows), one launch per row block (quad_launch_per_block), k_xprod time per series.
    python tools/quad_closing_probe.py"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from pypyls_amd import resampling                    # noqa: E402
from pypyls_amd.engine import Engine                 # noqa: E402

S, B, T, k = 1000, 100000, 20, 15
rs = np.random.RandomState(0)
X = rs.randn(S, B)
Y = rs.randn(S, T) + 0.3 * X[:, :T]
Xc, Yc = X - X.mean(0), Y - Y.mean(0)
rows = []
ref = None
for mt, full, one in ((0, 0, 0), (0, 0, 1), (24, 0, 0), (22, 0, 0), (20, 0, 0), (16, 0, 0), (12, 0, 0), (12, 0, 1), (24, 1, 0), (21, 1, 0)):
    eng = Engine(options={'quad_sums': 1, 'quad_mt': mt, 'quad_full_rows': full, 'quad_launch_per_block': one})
    eng.set_data_regression(Xc, Yc, k)
    W, pct, cvec, _ = eng.simpls_decompose()
    sg = np.sign(W[np.argmax(np.abs(W), axis=0), np.arange(k)])
    eng.simpls_set_original(W * sg)
    idx = eng.index_tensor(resampling.gen_bootsamp([S], 1, 96, seed=3, verbose=False))
    usum = torch.zeros((B, k), dtype=torch.float64, device='cuda'); usq = torch.zeros_like(usum)
    yl = torch.zeros((96, T, k), dtype=torch.float64, device='cuda')
    for rep in range(2):
        usum.zero_(); usq.zero_()
        eng.set_timing(rep == 1)
        assert eng.boot_begin(96) == 1
        eng.simpls_boot_into(idx, usum, usq, yl)
        eng.boot_finish(usum, usq)
        eng.sync()
    kt = eng.kernel_timing()
    tm = eng.last_timing()
    q = usq.cpu().numpy()
    if ref is None:
        ref = q
    rows.append({'quad_mt': mt, 'full_rows': full, 'launch_per_block': one, 'm_tiles': tm['quad_m_tiles'], 'blocks_per_lv': tm['quad_blocks_per_lv'],
                 'k_xprod_ms': kt['k_xprod'][0], 'max_rel_dev_from_first': float(np.max(np.abs(q - ref)) / np.max(np.abs(ref)))})
    print(rows[-1], flush=True)
    del eng
print(json.dumps(rows))
