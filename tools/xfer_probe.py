#!/usr/bin/env python3
"""Host <-> device transfer and host-side post-processing costs that bound the fixed (non-resampling) part of a
front-end call at the c4 shape: pageable / pinned H2D of X (800 MB), pinned allocation, D2H of the results
(B x L = 80 MB, n_boot x T' x L = 200 MB), host transposes.  One JSON line (gpurun_out/ -> profiles/)."""
import json
import time

import numpy as np
import torch


def t(fn, reps=3):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        dt = 1e3 * (time.perf_counter() - t0)
        best = dt if best is None else min(best, dt)
    return best


def main():
    dev = torch.device('cuda', 0)
    out = {}
    X = np.random.rand(500, 200000)
    out['h2d_pageable_800MB_ms'] = t(lambda: torch.from_numpy(X).to(dev))
    t0 = time.perf_counter()
    pin = torch.empty((500, 200000), dtype=torch.float64, pin_memory=True)
    out['pin_alloc_800MB_ms'] = 1e3 * (time.perf_counter() - t0)
    t0 = time.perf_counter()
    pin.numpy()[:] = X
    out['host_copy_into_pinned_800MB_ms'] = 1e3 * (time.perf_counter() - t0)
    out['h2d_pinned_800MB_ms'] = t(lambda: pin.to(dev, non_blocking=True))
    t0 = time.perf_counter()
    torch.cuda.cudart().cudaHostRegister(X.ctypes.data, X.nbytes, 0)
    out['host_register_800MB_ms'] = 1e3 * (time.perf_counter() - t0)
    out['h2d_registered_800MB_ms'] = t(lambda: torch.from_numpy(X).to(dev, non_blocking=True))
    torch.cuda.cudart().cudaHostUnregister(X.ctypes.data)
    d80 = torch.rand((200000, 50), dtype=torch.float64, device=dev)
    d200 = torch.rand((10000, 50, 50), dtype=torch.float64, device=dev)
    out['d2h_pageable_80MB_ms'] = t(lambda: d80.cpu())
    out['d2h_pageable_200MB_ms'] = t(lambda: d200.cpu())
    t0 = time.perf_counter()
    p80 = torch.empty((200000, 50), dtype=torch.float64, pin_memory=True)
    out['pin_alloc_80MB_ms'] = 1e3 * (time.perf_counter() - t0)
    t0 = time.perf_counter()
    p200 = torch.empty((50, 50, 10000), dtype=torch.float64, pin_memory=True)
    out['pin_alloc_200MB_ms'] = 1e3 * (time.perf_counter() - t0)
    out['d2h_pinned_80MB_ms'] = t(lambda: p80.copy_(d80, non_blocking=True))
    dT = d200.permute(1, 2, 0).contiguous()
    out['device_transpose_200MB_ms'] = t(lambda: d200.permute(1, 2, 0).contiguous())
    out['d2h_pinned_200MB_ms'] = t(lambda: p200.copy_(dT, non_blocking=True))
    h200 = d200.cpu().numpy()
    out['host_moveaxis_200MB_ms'] = t(lambda: np.ascontiguousarray(np.moveaxis(h200, 0, -1)), reps=2)
    h80 = d80.cpu().numpy()
    out['host_sign_convention_80MB_ms'] = t(lambda: np.argmax(np.abs(h80), axis=0), reps=2)
    rows = np.random.randint(0, 500, size=(10000, 500)).astype(np.int32)
    out['host_samples_int64_T_40MB_ms'] = t(lambda: np.ascontiguousarray(rows.T, dtype=np.int64), reps=2)
    out['device_argmax_80MB_ms'] = t(lambda: torch.argmax(d80.abs(), dim=0))
    print(json.dumps(out))


if __name__ == '__main__':
    main()
