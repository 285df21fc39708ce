#!/usr/bin/env python3
"""What would compact (distinct-row) cross-product blocks buy the SIMPLS bootstrap at c5 (k = 15 weight rows per
bootstrap = ONE 16-row tile)?  The PLS-C bootstrap with T' = 15 at the c5 shape runs exactly that block --
k_xprod_compact<1, ...>: one bootstrap per block, contraction over the rows it draws, X fragments gathered per block --
so its time per bootstrap, next to the dense grouped blocks (option no_compact_boot: 25 bootstraps on 24 tiles, every X
fragment shared by all of them), is the measurement; no new kernel needed to decide.
    python tools/compact_k15_probe.py [n_boot]"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from pypyls_amd import resampling, hostmath          # noqa: E402
from pypyls_amd.engine import Engine                 # noqa: E402

n_boot = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
S, B, T = 1000, 100000, 15
rs = np.random.RandomState(0)
X = rs.randn(S, B)
Y = rs.randn(S, T) + 0.3 * X[:, :T]
rows = {}
for label, opts in (('compact', {'compact_boot_always': 1}), ('dense', {'no_compact_boot': 1})):
    eng = Engine(options=opts)
    eng.set_data(X, Y, resampling.cell_of_row([S], 1), 1, 1, 0)
    xw, sv, yw = eng.decompose()
    xw, yw = hostmath.sign_convention(xw, yw)
    eng.set_original(xw, sv, yw)
    L, Tp = eng.L, eng.Tp
    idx = eng.index_tensor(resampling.gen_bootsamp([S], 1, n_boot, seed=77, verbose=False))
    usum = torch.zeros((B, L), dtype=torch.float64, device='cuda')
    usq = torch.zeros_like(usum)
    dist = torch.zeros((n_boot, Tp, L), dtype=torch.float64, device='cuda')
    eng.boot_into(idx, usum, usq, dist)
    eng.sync()
    eng.set_timing(True)
    usum.zero_(); usq.zero_()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    eng.boot_into(idx, usum, usq, dist)
    ev[1].record()
    eng.sync()
    kt = eng.kernel_timing()
    tm = eng.last_timing()
    rows[label] = {'total_ms': ev[0].elapsed_time(ev[1]), 'kernels_ms': {k: round(v[0], 3) for k, v in kt.items() if v[0] > 0},
                   'compact_row_fraction': tm.get('compact_row_fraction', 0.0), 'usum_checksum': float(usum.sum().item())}
    eng.set_timing(False)
    del eng
fl = 2.0 * S * 16 * B * n_boot
for label in rows:
    x = rows[label]['kernels_ms'].get('k_xprod', 0.0)
    rows[label]['xprod_dense_equivalent_tflops'] = fl / (x * 1e-3) / 1e12 if x > 0 else None
print(json.dumps({'shape': 'behavioral X(1000x100000) Y(1000x15): T\' = 15 = one 16-row tile per bootstrap', 'n_boot': n_boot,
                  'routes': rows}))
