#!/usr/bin/env python3
"""Wide behaviour matrices, split-half leg: ms per split and the per-kernel split.
    python tools/bench_wide_split.py [S B T n_arr n_split]"""
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
    S, B, T, n_arr, ns = (a + [400, 50000, 100, 4, 50][len(a):])[:5]
    rs = np.random.RandomState(0)
    X = rs.randn(S, B)
    Y = rs.randn(S, T) + 0.3 * X[:, :T]
    from pypyls_amd.engine import options_from_env
    eng = Engine(**dict(options_from_env(), scratch_gb=24))
    eng.set_data(X, Y, resampling.cell_of_row([S], 1), 1, 1, 0)
    xw, sv, yw = eng.decompose()
    xw, yw = hostmath.sign_convention(xw, yw)
    eng.set_original(xw, sv, yw)
    perms = eng.index_tensor(resampling.gen_permsamp([S], 1, n_arr, seed=2, verbose=False))
    masks = np.stack([resampling.gen_splits([S], 1, ns, seed=10 + i) for i in range(n_arr)])     # (n_arr, S, ns)
    dm = torch.from_numpy(np.ascontiguousarray(masks.transpose(0, 2, 1), dtype=np.uint8)).to(perms.device)
    uc = torch.zeros((n_arr, ns, eng.L), dtype=torch.float64, device=perms.device)
    vc = torch.zeros_like(uc)
    eng.split_half_into(perms, dm, uc, vc)
    torch.cuda.synchronize()
    eng.set_timing(True)
    t0 = time.perf_counter()
    eng.split_half_into(perms, dm, uc, vc)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kt = eng.kernel_timing()
    eng.set_timing(False)
    n = n_arr * ns
    print(json.dumps(dict(S=S, B=B, T=T, Tp=eng.Tp, splits=n, ms_per_split=1e3 * dt / n,
                          kernels_ms_per_split={k: v[0] / n for k, v in kt.items()})))


if __name__ == '__main__':
    main()
