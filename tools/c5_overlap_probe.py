"""c5 (SIMPLS 1000 x 100 000, T = 20, k = 15; 5000 permutations + 5000 bootstraps): the permutation chain and the
bootstrap chain are independent, and their kernels want different resources (k_sd_* : VALU issue / latency, the K products
and the closing pass: the matrix pipe).  Time one step with both chains on ONE context and stream (what bench.py --config
c5 does) and with the permutations on a SECOND context of the same device on its own stream, enqueued from one host
thread.  usage: python tools/c5_overlap_probe.py [steps]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from pypyls_amd import resampling
    from pypyls_amd.engine import Engine
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    S, B, T, k, n = 1000, 100000, 20, 15, 5000
    rs = np.random.RandomState(0)
    X = rs.randn(S, B)
    Y = rs.randn(S, T) + 0.3 * X[:, :T]
    Xc, Yc = X - X.mean(0), Y - Y.mean(0)
    dev = torch.device('cuda', 0)
    engs = []
    for _ in range(2):
        e = Engine(scratch_gb=24.0)
        e.set_data_regression(Xc, Yc, k)
        W, pct, cvec, _ = e.simpls_decompose()
        sg = np.sign(W[np.argmax(np.abs(W), axis=0), np.arange(k)])
        e.simpls_set_original(W * sg)
        engs.append(e)
    perm = engs[0].index_tensor(resampling.gen_permsamp([S], 1, n, seed=1, verbose=False))
    boot = engs[0].index_tensor(resampling.gen_bootsamp([S], 1, n, seed=2, verbose=False))
    out = torch.zeros((n, k), dtype=torch.float64, device=dev)
    yl = torch.zeros((n, T, k), dtype=torch.float64, device=dev)
    usum = torch.zeros((B, k), dtype=torch.float64, device=dev)
    usq = torch.zeros((B, k), dtype=torch.float64, device=dev)
    s2 = torch.cuda.Stream(device=dev)

    def boots(e):
        usum.zero_(); usq.zero_()
        e.boot_begin(n)
        e.simpls_boot_into(boot, usum, usq, yl)
        e.boot_finish(usum, usq)

    def sequential():
        engs[0].simpls_perm_into(perm, out)
        boots(engs[0])

    def overlapped():
        s2.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s2):
            engs[1].simpls_perm_into(perm, out)
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
        snap = (out.cpu().numpy().copy(), usum.cpu().numpy().copy(), yl.cpu().numpy().copy())
        if ref is None:
            ref = snap
        same = all(np.array_equal(a, b) for a, b in zip(ref, snap))
        res[name] = {'ms_per_step': ms, 'resamples_per_s': 2 * n / (ms * 1e-3), 'bit_identical_to_first': same}
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
