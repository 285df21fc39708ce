#!/usr/bin/env python3
"""
bench.py -- resamples/sec (perm + boot) of the PLS-C resampling hot path.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric): behavioral PLS,
X (500 x 200000), Y (500 x 50), fp64, synthetic
(RandomState(0): X = randn, Y = randn + 0.3 * X[:, :T]).  One "step" = one
pass of the hot path over one batch of resamples per GPU: PERMS permutations
+ BOOTS bootstraps (index arrays already in HBM), i.e. for every resample
gather/permute -> per-cell z-score -> R = Yn^T Xn -> Gram-side Jacobi SVD ->
Procrustes -> null value / running sum U, sum U^2 + distrib.  Weak scaling:
every rank processes its own PERMS + BOOTS per step with a full replica of X;
no collective inside the data path (the one all-gather of results happens
once per analysis, outside the steady-state step).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6     # MI355X fp64 matrix peak (vendor; = FP32 matrix 157.3 / 2)
PEAK_HBM_TBS = 8.0               # MI355X_MICROARCH.md


def synth(S, B, T):
    rs = np.random.RandomState(0)
    X = rs.randn(S, B)
    Y = rs.randn(S, T) + 0.3 * X[:, :T]
    return X, Y


def cpu_baseline(X, Y, x_weights, y_weights, n_each, seed=1234):
    """Oracle (numpy restatement of the reference path) timed on the host
    cores: n_each permutations + n_each bootstraps of the same workload."""
    from oracle import cpu_ref as ref
    from pypyls_amd import resampling
    S = X.shape[0]
    spec = ref.Spec('behavioral', [S], 1)
    perms = resampling.gen_permsamp([S], 1, n_each, seed=seed, verbose=False)
    boots = resampling.gen_bootsamp([S], 1, n_each, seed=seed + 1, verbose=False)
    t0 = time.perf_counter()
    for i in range(n_each):
        ref.single_perm(spec, X, Y, perms[:, i], y_weights)
    for i in range(n_each):
        ref.single_boot(spec, X, Y, boots[:, i], x_weights)
    dt = time.perf_counter() - t0
    return 2 * n_each / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--S', type=int, default=500)
    ap.add_argument('--B', type=int, default=200000)
    ap.add_argument('--T', type=int, default=50)
    ap.add_argument('--perms', type=int, default=1008, help='permutations per step per GPU')
    ap.add_argument('--boots', type=int, default=1008, help='bootstraps per step per GPU')
    ap.add_argument('--cpu-sample', type=int, default=8,
                    help='permutations and bootstraps (each) timed for the CPU baseline; 0 = skip')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    backend = os.environ.get('PLSX_BENCH_BACKEND', 'nccl')   # 'gloo' only for single-GPU dry runs
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=torch.device('cuda', torch.cuda.current_device()))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    dev = torch.device('cuda', torch.cuda.current_device())

    from pypyls_amd import resampling, hostmath
    from pypyls_amd.engine import Engine

    S, B, T = args.S, args.B, args.T
    X, Y = synth(S, B, T)
    # long-lived engine: fixed 48 GB super-batch scratch, mapped during warm-up
    eng = Engine(scratch_gb=float(os.environ.get('PLSX_SCRATCH_GB', 48)))
    eng.set_data(X, Y, resampling.cell_of_row([S], 1), 1, 1, 0)
    xw, sv, yw = eng.decompose()
    xw, yw = hostmath.sign_convention(xw, yw)
    eng.set_original(xw, sv, yw)
    L, Tp = eng.L, eng.Tp

    # distinct index arrays per rank and per step, resident in HBM before timing
    n_steps = args.steps + args.warmup
    perm_idx, boot_idx = [], []
    for s in range(n_steps):
        seed = 1234 + 1000 * rank + s
        perm_idx.append(eng.index_tensor(
            resampling.gen_permsamp([S], 1, args.perms, seed=seed, verbose=False)))
        boot_idx.append(eng.index_tensor(
            resampling.gen_bootsamp([S], 1, args.boots, seed=seed + 500, verbose=False)))
    out_sv = torch.empty((args.perms, L), dtype=torch.float64, device=dev)
    dist_out = torch.empty((args.boots, Tp, L), dtype=torch.float64, device=dev)
    usum = torch.zeros((B, L), dtype=torch.float64, device=dev)
    usq = torch.zeros((B, L), dtype=torch.float64, device=dev)

    leg_events = []

    def step(i, timed=False):
        if timed:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
        eng.perm_into(perm_idx[i], out_sv, rotate=True)
        if timed:
            ev[1].record()
        eng.boot_into(boot_idx[i], usum, usq, dist_out)
        if timed:
            ev[2].record()
            leg_events.append(ev)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    eng.set_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, n_steps):
        step(i, timed=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device=dev if backend == 'nccl' else torch.device('cpu'))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    timing = eng.last_timing()
    eng.set_timing(False)

    if rank == 0:
        per_step = args.perms + args.boots
        total = per_step * args.steps * world
        value = total / elapsed
        # dominant kernel: k_xprod.  Algorithmic flops per resample of this
        # kernel: 2*S*T'*B (first term of SURVEY section 8d W_F); a launch
        # processes `launch_units` resamples.
        launches = max(int(timing.get('xprod_launches', 0)), 1)
        avg_ms = timing.get('xprod_ms', 0.0) / launches
        dual = bool(timing.get('dual_perm', 0))
        # resamples the timed k_xprod launches covered (bootstraps only when the
        # permutations take the dual S x S path and launch no k_xprod at all)
        units_per_launch = timing.get('xprod_resamples', per_step * args.steps) / launches
        flops_launch = 2.0 * S * Tp * B * units_per_launch
        achieved = flops_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic_xprod.json')
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get('hbm_bytes_per_launch') * units_per_launch / tj.get('resamples_per_launch')
            except Exception:
                traffic = None
        # whole-pipeline fractions: algorithmic flops / bytes of THIS formulation
        # (DESIGN.md section 5).  Bootstrap: cross-product + Gram + R.U0 + U
        # rotation, all O(B).  Permutation: O(B) only without the dual path;
        # with it, the S x S kernel once per call plus O(T' S^2) per permutation.
        wf_boot = 2.0 * S * Tp * B + 2.0 * Tp * Tp * B + 4.0 * Tp * L * B
        wb_boot = 8.0 * S * (B + Tp)
        if dual:
            wf_perm = 2.0 * Tp * S * S + 2.0 * Tp * Tp * S + 2.0 * S * S * B / max(args.perms, 1)
            wb_perm = 8.0 * (S * B / max(args.perms, 1) + 3.0 * Tp * S)
        else:
            wf_perm = 2.0 * S * Tp * B + 2.0 * Tp * Tp * B
            wb_perm = wb_boot
        steps_per_s = args.steps / elapsed
        frac_mfma = steps_per_s * (args.perms * wf_perm + args.boots * wf_boot) / (PEAK_FP64_MFMA_TFLOPS * 1e12)
        frac_hbm = steps_per_s * (args.perms * wb_perm + args.boots * wb_boot) / (PEAK_HBM_TBS * 1e12)
        perm_ms = sum(e[0].elapsed_time(e[1]) for e in leg_events) / max(len(leg_events), 1)
        boot_ms = sum(e[1].elapsed_time(e[2]) for e in leg_events) / max(len(leg_events), 1)
        out = {
            'metric': 'resamples/sec (perm+boot), behavioral_pls X({}x{})/Y({}x{}) fp64'
                      .format(S, B, S, T),
            'value': value, 'unit': 'resamples/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'behavioral_pls X({}x{}) Y({}x{}) fp64 (BASELINE configs[3] '
                                   'shape on {} GPU(s)), n_split=0, test_split=0'
                                   .format(S, B, S, T, world),
                       'perms_per_step_per_gpu': args.perms, 'boots_per_step_per_gpu': args.boots,
                       'perm_path': 'dual (S x S kernel)' if dual else 'feature pass',
                       'perm_ms_per_step': perm_ms, 'boot_ms_per_step': boot_ms,
                       'parallelism': 'resample-sharded x{}'.format(world)},
            'roofline': {'bound': 'mfma', 'kernel': 'k_xprod<{}>'.format(int(timing.get('m_tiles', 0))),
                         'achieved': achieved, 'peak': PEAK_FP64_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / PEAK_FP64_MFMA_TFLOPS, 'traffic': traffic,
                         'avg_launch_ms': avg_ms, 'launches': launches,
                         'resamples_per_launch': units_per_launch,
                         'measured_mfma_f64_peak_tflops': eng.mfma_f64_peak(),
                         'pipeline_frac_mfma': frac_mfma, 'pipeline_frac_hbm': frac_hbm},
        }
        if world == 1 and args.cpu_sample > 0:
            cores = os.cpu_count() or 1
            v, dt = cpu_baseline(X, Y, xw, yw, args.cpu_sample)
            out['cpu_baseline'] = {
                'value': v, 'unit': 'resamples/s', 'cores': cores, 'kind': 'port',
                'sample': '{} permutations + {} bootstraps of the same workload through '
                          'oracle/cpu_ref.py (numpy, BLAS threads = host cores), {:.1f} s'
                          .format(args.cpu_sample, args.cpu_sample, dt)}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
