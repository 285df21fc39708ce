"""c4 headline step (1008 permutations through the feature pass + 1008 bootstraps at 500 x 200 000, T' = 50) with both
chains on one context and stream vs the permutation chain on a second context of the same device on its own stream.
usage: python tools/c4_overlap_probe.py [steps]"""
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
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    S, B, T, n = 500, 200000, 50, 1008
    rs = np.random.RandomState(0)
    X = rs.randn(S, B)
    Y = rs.randn(S, T) + 0.3 * X[:, :T]
    dev = torch.device('cuda', 0)
    engs = []
    for _ in range(2):
        e = Engine(scratch_gb=24.0)
        e.set_data(X, Y, resampling.cell_of_row([S], 1), 1, 1, 0)
        xw, sv, yw = e.decompose()
        xw, yw = hostmath.sign_convention(xw, yw)
        e.set_original(xw, sv, yw)
        e.set_perm_path(False)                      # permutations through the feature pass (bench.py's `value`)
        engs.append(e)
    L, Tp = engs[0].L, engs[0].Tp
    perm = engs[0].index_tensor(resampling.gen_permsamp([S], 1, n, seed=1, verbose=False))
    boot = engs[0].index_tensor(resampling.gen_bootsamp([S], 1, n, seed=2, verbose=False))
    out = torch.zeros((n, L), dtype=torch.float64, device=dev)
    dist = torch.zeros((n, Tp, L), dtype=torch.float64, device=dev)
    usum = torch.zeros((B, L), dtype=torch.float64, device=dev)
    usq = torch.zeros((B, L), dtype=torch.float64, device=dev)
    s2 = torch.cuda.Stream(device=dev)

    def boots(e):
        usum.zero_(); usq.zero_()
        e.boot_begin(n)
        e.boot_into(boot, usum, usq, dist)
        e.boot_finish(usum, usq)

    def sequential():
        engs[0].perm_into(perm, out, rotate=True)
        boots(engs[0])

    def overlapped():
        s2.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s2):
            engs[1].perm_into(perm, out, rotate=True)
        boots(engs[0])
        torch.cuda.current_stream(dev).wait_stream(s2)

    res = {}
    ref = None
    for name, fn in (('one context, one stream', sequential), ('two contexts, two streams', overlapped),
                     ('one context, one stream (again)', sequential)):
        fn(); fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        snap = (out.cpu().numpy().copy(), usum.cpu().numpy().copy())
        if ref is None:
            ref = snap
        same = all(np.array_equal(a, b) for a, b in zip(ref, snap))
        res[name] = {'ms_per_step': ms, 'resamples_per_s': 2 * n / (ms * 1e-3), 'bit_identical_to_first': same}
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
