"""Does the c4 bootstrap pipeline gain from two sub-batches in flight on two HIP streams?

Two contexts bound to the same data (own scratch each), each fed from its own stream; the chains
xprod -> gram -> small -> urot of the two overlap wherever the hardware finds room.  Timing-only
experiment (the two engines accumulate into separate sums).  Usage: python tools/overlap_probe.py [S B T n]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from pypyls_amd import resampling, hostmath
    from pypyls_amd.engine import Engine
    from bench import synth
    vals = [500, 200000, 50, 1008]
    for i, a in enumerate(sys.argv[1:5]):
        vals[i] = int(a)
    S, B, T, n = vals
    X, Y = synth(S, B, T)
    cells = resampling.cell_of_row([S], 1)
    engs = []
    for k in range(2):
        e = Engine(scratch_gb=48.0)
        e.set_data(X, Y, cells, 1, 1, 0)
        xw, sv, yw = e.decompose()
        xw, yw = hostmath.sign_convention(xw, yw)
        e.set_original(xw, sv, yw)
        engs.append(e)
    dev = engs[0].device
    L, Tp = engs[0].L, engs[0].Tp
    idx = engs[0].index_tensor(resampling.gen_bootsamp([S], 1, n, seed=99, verbose=False))
    usum = [torch.zeros((B, L), dtype=torch.float64, device=dev) for _ in range(2)]
    usq = [torch.zeros((B, L), dtype=torch.float64, device=dev) for _ in range(2)]
    dist = torch.zeros((n, Tp, L), dtype=torch.float64, device=dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]

    def run(plan):
        """plan: list of (engine number, lo, hi) in enqueue order."""
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k, lo, hi in plan:
            with torch.cuda.stream(streams[k]):
                engs[k].boot_into(idx[lo:hi], usum[k], usq[k], dist[lo:hi])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    h, q, e8 = n // 2, n // 4, n // 8
    plans = {
        'one stream, 2 x %d' % h: [(0, 0, h), (0, h, n)],
        'two streams, %d each' % h: [(0, 0, h), (1, h, n)],
        'one stream, 4 x %d' % q: [(0, i * q, (i + 1) * q) for i in range(4)],
        'two streams, 2 x %d each, interleaved' % q: [(0, 0, q), (1, q, 2 * q), (0, 2 * q, 3 * q), (1, 3 * q, n)],
        'two streams, staggered (%d first)' % e8: [(1, 0, e8), (0, e8, e8 + q), (1, e8 + q, e8 + 2 * q),
                                                   (0, e8 + 2 * q, e8 + 3 * q), (1, e8 + 3 * q, n)],
        'two streams, 4 x %d each' % e8: [(i % 2, i * e8, (i + 1) * e8) for i in range(8)],
    }
    out = {}
    for name, plan in plans.items():
        run(plan)
        ts = [run(plan) for _ in range(3)]
        out[name] = [round(t, 2) for t in ts]
        print(name, out[name], flush=True)
    # the sums of the two engines together equal one engine's (sanity of the experiment)
    json.dump({'shape': [S, B, T, n], 'ms': out}, open(os.path.join('gpurun_out', 'overlap_probe.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
