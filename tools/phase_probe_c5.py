#!/usr/bin/env python3
"""Phases of the public pls_regression call at c5 (5000 + 5000), from a seed (index arrays drawn on the host thread)
and with given arrays: python tools/phase_probe_c5.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import pypyls_amd as pls
from pypyls_amd import resampling
rs = np.random.RandomState(0)
S, B, T, k = 1000, 100000, 20, 15
X = rs.randn(S, B); Y = rs.randn(S, T) + 0.3 * X[:, :T]
pls.pls_regression(X, Y, n_components=k, n_perm=5000, n_boot=5000, seed=1, verbose=False)
for mode in ('seed', 'given'):
    kw = dict(seed=1234)
    if mode == 'given':
        kw = dict(permsamples=resampling.gen_permsamp([S], 1, 5000, seed=3, verbose=False),
                  bootsamples=resampling.gen_bootsamp([S], 1, 5000, seed=4, verbose=False))
    for rep in range(2):
        ph = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pls.pls_regression(X, Y, n_components=k, n_perm=5000, n_boot=5000, verbose=False, _phases=ph, **kw)
        print(mode, round(1e3 * (time.perf_counter() - t0), 1), {a: round(b, 1) for a, b in ph.items()})
