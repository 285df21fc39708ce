#!/usr/bin/env python3
"""Wide behaviour matrices: ms per bootstrap / permutation and the per-kernel split.
    python tools/bench_wide.py [S B T n_groups n_cond nres]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from pypyls_amd import resampling, hostmath
    from pypyls_amd.engine import Engine
    a = [int(x) for x in sys.argv[1:]]
    S, B, T, G, C, n = (a + [400, 50000, 200, 1, 1, 256][len(a):])[:6]
    rs = np.random.RandomState(0)
    X = rs.randn(S, B)
    Y = rs.randn(S, T) + 0.3 * X[:, :T]
    groups = [S // G // C] * G
    from pypyls_amd.engine import options_from_env
    eng = Engine(**dict(options_from_env(), scratch_gb=24))
    eng.set_data(X, Y, resampling.cell_of_row(groups, C), G, C, 0)
    xw, sv, yw = eng.decompose()
    xw, yw = hostmath.sign_convention(xw, yw)
    eng.set_original(xw, sv, yw)
    boots = eng.index_tensor(resampling.gen_bootsamp(groups, C, n, seed=1, verbose=False))
    perms = eng.index_tensor(resampling.gen_permsamp(groups, C, n, seed=2, verbose=False))
    dev = boots.device
    usum = torch.zeros((B, eng.L), dtype=torch.float64, device=dev)
    usq = torch.zeros_like(usum)
    dist = torch.zeros((n, eng.Tp, eng.L), dtype=torch.float64, device=dev)
    out = torch.zeros((n, eng.L), dtype=torch.float64, device=dev)
    res = {}
    for name, fn in (('boot', lambda: eng.boot_into(boots, usum, usq, dist)),
                     ('perm_dual', lambda: eng.perm_into(perms, out)),
                     ('perm_primal', lambda: (eng.set_perm_path(False), eng.perm_into(perms, out),
                                              eng.set_perm_path(True)))):
        fn()
        torch.cuda.synchronize()
        eng.set_timing(True)
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kt = eng.kernel_timing()
        eng.set_timing(False)
        res[name] = dict(ms_per_resample=1e3 * dt / n, kernels_ms_per_resample={k: v[0] / n for k, v in kt.items()})
    print(json.dumps(dict(S=S, B=B, T=T, Tp=eng.Tp, n=n, **res)))


if __name__ == '__main__':
    main()
