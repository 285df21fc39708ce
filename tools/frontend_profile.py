#!/usr/bin/env python3
"""Where the wall time of the public pls_regression call goes at the c5 shape (the PLS-C front-ends report their
phases themselves: bench.py --mode analysis): every Engine method wrapped with a device sync on both sides.
    python tools/frontend_profile.py [n_perm n_boot]"""
import functools
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import pypyls_amd as pls                               # noqa: E402
from pypyls_amd import engine as E, parallel           # noqa: E402

n_perm = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
n_boot = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
rs = np.random.RandomState(0)
S, B, T, k = 1000, 100000, 20, 15
X = rs.randn(S, B)
Y = rs.randn(S, T) + 0.3 * X[:, :T]
pls.pls_regression(X, Y, n_components=k, n_perm=n_perm, n_boot=n_boot, seed=1, verbose=False)      # (same sizes: scratch mapped)
acc = {}


def timed(obj, name):
    f = getattr(obj, name)

    @functools.wraps(f)
    def w(*a, **kw):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = f(*a, **kw)
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        return r
    setattr(obj, name, w)


for n in ['set_data', 'colmean_dev', 'simpls_decompose', 'simpls_set_original', 'project', 'boot_rel', 'percentile_ci',
          'simpls_perm_into', 'simpls_boot_into', 'boot_begin', 'boot_finish', 'rows_tensor', 'sync', '_dev', 'set_data_regression']:
    timed(E.Engine, n)
timed(parallel, 'collect_slices')
torch.cuda.synchronize()
t = time.perf_counter()
res = pls.pls_regression(X, Y, n_components=k, n_perm=n_perm, n_boot=n_boot, seed=1234, verbose=False)
torch.cuda.synchronize()
tot = time.perf_counter() - t
print('total %.3f s' % tot, {k_: round(v, 4) for k_, v in acc.items()}, 'sum %.3f' % sum(acc.values()))
