#!/usr/bin/env python3
"""Where the time of ONE rank of an emulated world goes (strong mode, c4): wall clock of the step, kernel
time per class, and when each chunk of rows was launched.  usage: tools/rank_timeline.py N rank [perms boots]"""
import argparse
import json
import sys
import time

sys.path.insert(0, '.')
import numpy as np
import torch

import bench


def main():
    n, r = int(sys.argv[1]), int(sys.argv[2])
    perms = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
    boots = int(sys.argv[4]) if len(sys.argv) > 4 else 10000
    args = argparse.Namespace(gpus=1, steps=1, warmup=1, config='c4', mode='strong', S=500, B=200000, T=50,
                              perms=perms, boots=boots, cpu_sample=0, no_primal=True, emulate_world='')
    torch.cuda.set_device(0)
    wl = bench.make_workload(args)
    wl.setup(0, 2, torch.device('cuda', 0))
    eng = wl.eng
    log = []
    for name in ('perm_into', 'boot_into'):
        fn = getattr(eng, name)

        def wrap(rows, *a, _fn=fn, _name=name, **k):
            log.append((_name, 1e3 * (time.perf_counter() - t0), int(rows.shape[0])))
            return _fn(rows, *a, **k)
        setattr(eng, name, wrap)
    for rep in range(3):
        wl.usum.zero_(); wl.usq.zero_()
        torch.cuda.synchronize()
        del log[:]
        eng.set_timing(True)
        t0 = time.perf_counter()
        wl._strong_step(1000 + rep, rank=r, world=n)
        t_host = 1e3 * (time.perf_counter() - t0)
        torch.cuda.synchronize()
        wall = 1e3 * (time.perf_counter() - t0)
        kt = eng.kernel_timing()
        eng.set_timing(False)
    print(json.dumps({'world': n, 'rank': r, 'wall_ms': wall, 'host_returned_ms': t_host,
                      'kernel_ms': {k: round(v[0], 2) for k, v in kt.items() if v[0] > 0},
                      'kernel_ms_total': round(sum(v[0] for v in kt.values()), 2),
                      'launches': [(a, round(b, 1), c) for a, b, c in log]}))


if __name__ == '__main__':
    main()
