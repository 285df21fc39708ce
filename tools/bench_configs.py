#!/usr/bin/env python3
"""
Throughput of the other BASELINE.json configs (parity-test shapes, not the
headline bench line): device-only resamples/s with index arrays pre-generated,
plus end-to-end wall time of the public front-end call where it is cheap.

    python tools/bench_configs.py [c2] [c3] [c5] [c4split] [c4cv]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn):
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0


def synth(S, B, T):
    rs = np.random.RandomState(0)
    X = rs.randn(S, B)
    Y = rs.randn(S, T) + 0.3 * X[:, :T]
    return X, Y


def c2():
    import pypyls_amd as pls
    from pypyls_amd import resampling
    X, Y = synth(80, 10000, 10)
    perms = resampling.gen_permsamp([80], 1, 5000, seed=1234, verbose=False)
    boots = resampling.gen_bootsamp([80], 1, 5000, seed=1235, verbose=False)
    pls.behavioral_pls(X, Y, n_perm=64, n_boot=64, test_split=0, seed=1, verbose=False)   # warm-up
    res, dt = timed(lambda: pls.behavioral_pls(X, Y, n_perm=5000, n_boot=5000, test_split=0,
                                               permsamples=perms, bootsamples=boots, verbose=False))
    return dict(config='c2 behavioral X(80x10000) Y(80x10) 5000+5000', seconds=dt,
                resamples_per_s=10000 / dt, note='front-end call incl. decomposition, H2D/D2H, CIs')


def c3():
    import pypyls_amd as pls
    from pypyls_amd import resampling
    rs = np.random.RandomState(0)
    X = rs.randn(200, 50000)
    groups = [25, 25, 25, 25]
    perms = resampling.gen_permsamp(groups, 2, 10000, seed=1234, verbose=False)
    boots = resampling.gen_bootsamp(groups, 2, 10000, seed=1235, verbose=False)
    pls.meancentered_pls(X, groups=groups, n_cond=2, n_perm=64, n_boot=64, seed=1, verbose=False)
    res, dt = timed(lambda: pls.meancentered_pls(X, groups=groups, n_cond=2, n_perm=10000, n_boot=10000,
                                                 permsamples=perms, bootsamples=boots, verbose=False))
    return dict(config='c3 meancentered X(200x50000) groups=[25]*4 n_cond=2 10000+10000', seconds=dt,
                resamples_per_s=20000 / dt, note='front-end call; groups per SURVEY section 0')


def c5():
    import pypyls_amd as pls
    from pypyls_amd import resampling
    X, Y = synth(1000, 100000, 20)
    perms = resampling.gen_permsamp([1000], 1, 5000, seed=1234, verbose=False)
    boots = resampling.gen_bootsamp([1000], 1, 5000, seed=1235, verbose=False)
    pls.pls_regression(X, Y, n_components=15, n_perm=32, n_boot=32, seed=1, verbose=False)
    res, dt = timed(lambda: pls.pls_regression(X, Y, n_components=15, n_perm=5000, n_boot=5000,
                                               permsamples=perms, bootsamples=boots, verbose=False))
    return dict(config='c5 pls_regression X(1000x100000) Y(1000x20) k=15 5000+5000', seconds=dt,
                resamples_per_s=10000 / dt, note='front-end call; SIMPLS in the dual space')


def c4():
    """End-to-end front-end call at the headline shape (includes index generation,
    H2D of X, original decomposition, D2H of the (B, L) results, percentile CIs)."""
    import pypyls_amd as pls
    X, Y = synth(500, 200000, 50)
    pls.behavioral_pls(X, Y, n_perm=56, n_boot=56, test_split=0, seed=1, verbose=False)
    t0 = time.perf_counter()
    res, dt = timed(lambda: pls.behavioral_pls(X, Y, n_perm=5000, n_boot=5000, test_split=0, seed=1234,
                                               verbose=False))
    return dict(config='c4 behavioral X(500x200000) Y(500x50) 5000+5000, seed only (front-end wall time)',
                seconds=dt, resamples_per_s=10000 / dt, pval0=float(res.permres.pvals[0]))


def _c4_engine():
    from pypyls_amd import resampling, hostmath
    from pypyls_amd.engine import Engine
    X, Y = synth(500, 200000, 50)
    from pypyls_amd.engine import options_from_env
    eng = Engine(**options_from_env())
    eng.set_data(X, Y, resampling.cell_of_row([500], 1), 1, 1, 0)
    xw, sv, yw = eng.decompose()
    xw, yw = hostmath.sign_convention(xw, yw)
    eng.set_original(xw, sv, yw)
    return eng, resampling


def c4split():
    eng, resampling = _c4_engine()
    n_arr, ns = 4, 100
    perms = resampling.gen_permsamp([500], 1, n_arr, seed=3, verbose=False)
    masks = np.stack([resampling.gen_splits([500], 1, ns, seed=i) for i in range(n_arr)])
    eng.split_half(masks[:1], perms=perms[:, :1])          # warm-up at full size (allocations)
    (_, _), dt = timed(lambda: eng.split_half(masks, perms=perms))
    return dict(config='c4 split-half X(500x200000) Y(500x50), n_split=100 per permutation',
                seconds=dt, splits_per_s=n_arr * ns / dt,
                permutations_with_100_splits_per_s=n_arr / dt)


def c4cv():
    eng, resampling = _c4_engine()
    splits = resampling.gen_splits([500], 1, 100, seed=5, test_size=0.25)
    eng.crossval(splits)                                   # warm-up at full size (allocations)
    (_, _), dt = timed(lambda: eng.crossval(splits))
    return dict(config='c4 cross-validation X(500x200000) Y(500x50), test_split=100', seconds=dt,
                splits_per_s=100 / dt)


if __name__ == '__main__':
    which = sys.argv[1:] or ['c2', 'c3', 'c5', 'c4split', 'c4cv']
    for name in which:
        print(json.dumps(globals()[name]()), flush=True)
