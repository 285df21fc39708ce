#!/usr/bin/env python3
"""
bench.py -- throughput of the PLS resampling hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c4] [--mode weak]

``--gpus N`` with N > 1 launches N ranks itself (one process per GPU,
``torch.distributed.run`` on 127.0.0.1) unless it is already running under a
launcher (WORLD_SIZE set); it fails loudly when fewer GPUs are visible.  The
reported ``n_gpus`` is the world size of the RCCL process group.

Default workload = BASELINE.json's metric: behavioral PLS, X (500 x 200000),
Y (500 x 50), fp64, synthetic (RandomState(0): X = randn, Y = randn +
0.3 X[:, :T]).  One "step" (weak mode) = one analysis-sized pass of the hot
path per GPU: PERMS permutations + BOOTS bootstraps with index arrays already
in HBM -- gather/permute -> per-cell z-score -> R = Yn^T Xn -> Gram-side Jacobi
SVD -> Procrustes -> null value / running sum U, sum U^2 + distrib -- followed
by THE one collective of the analysis (all-gather of [perm slice | distrib
slice | sum U | sum U^2], pypyls_amd/parallel.py) on RCCL.  Weak scaling:
every rank runs its own PERMS + BOOTS per step on a full replica of X.

``value`` is the north-star pipeline: every permutation AND every bootstrap
passes over X (R = A X on the matrix pipe).  The product's default permutation
route for PLS-C is the S x S dual-space shortcut (no pass over X per
permutation); SURVEY.md section 8d asks for it to be reported separately, so
the same step with that route is timed in a second region of the same run and
reported as ``value_dual`` / ``ms_per_step_dual``.

The driver's command (``--gpus 1``, default config) also embeds, under
``configs``, one compact sub-record per other BASELINE config (c2, c3,
c4split, c5: value, ms_per_step, roofline.frac, cpu_baseline) and the
end-to-end public call at the headline shape (``c4_analysis``: index
generation inside the clock); ``--no-configs`` skips them.

``--mode strong``: one step = ONE analysis of --perms + --boots resamples
(default 10000 + 10000) split over the ranks with parallel.shard_bounds, run
the way the front-end runs it: one RandomState drawn on a host thread in the
reference's order while the rank ships its shard to the device chunk by chunk
(resampling.IndexStream); index generation, the H2D copies of the shards and
the one all-gather are inside the clock.  ``--emulate-world 1,2,4,8`` (one
GPU): critical path of one analysis on rank 0 and rank N-1 of an emulated
world N, both permutation routes, with the implied efficiency -- an emulation,
not a hardware scaling curve.

``--config`` selects another BASELINE config (bench lines for the record; the
driver runs the default): c2, c3, c5, c4split.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6     # MI355X fp64 matrix peak (vendor; = FP32 matrix 157.3 / 2)
PEAK_HBM_TBS = 8.0               # MI355X_MICROARCH.md



def _engine_kwargs():
    """Engine arguments of a bench run: a fixed 48 GB super-batch scratch (PLSX_SCRATCH_GB overrides) and the
    PLSX_<KEY> route switches of an A/B command line -- translated HERE; the library reads no environment."""
    from pypyls_amd.engine import options_from_env
    kw = options_from_env()
    kw.setdefault('scratch_gb', 48.0)
    return kw


def synth(S, B, T, seed=0):
    rs = np.random.RandomState(seed)
    X = rs.randn(S, B)
    Y = rs.randn(S, T) + 0.3 * X[:, :T]
    return X, Y


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """--gpus N > 1 outside a launcher: spawn the N ranks."""
    import torch
    have = torch.cuda.device_count()
    if os.environ.get('PLSX_BENCH_SHARE_GPU'):
        have = max(have, args.gpus) if have else 0        # dry run: ranks share the visible GPU(s) (gloo only)
    if have < args.gpus:
        sys.stderr.write('bench.py: --gpus {} requested but only {} GPU(s) are visible\n'
                         .format(args.gpus, have))
        return 2
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node',
           str(args.gpus), '--master-addr', '127.0.0.1', '--master-port', str(free_port()),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------

def quad_closing_work(S, mt, gpl):
    """Closing pass of a bootstrap series on the quadratic-form route (plsx_boot_finish): per latent variable and
    feature column the kernel multiplies gpl row blocks of mt tiles of C_l (S x S, symmetric), each from its own first
    row on, a tile of 16 rows inside the diagonal block from its own first row on.  Returns (flop it NEEDS per LV and column = the upper triangle incl. the diagonal, 2 x S (S + 1) / 2;
    flop it ISSUES per LV and column)."""
    issued = 0.0
    for p in range(gpl):
        s0 = p * mt * 16
        kp = max(4 * ((S + 3) // 4) - s0, 0)             # contraction length of the block
        issued += 2.0 * (mt * 16) * kp
        # inside its diagonal block tile m does not issue the 4 m k-steps left of its first row (128 flop each)
        issued -= 128.0 * sum(min(4 * m, kp // 4) for m in range(mt))
    return float(S) * (S + 1), issued


class PLSC(object):
    """behavioral / mean-centred PLS: permutations + bootstraps."""

    def __init__(self, args, name, method, S, B, T, groups, n_cond, perms, boots, cpu_each):
        self.args, self.name, self.method = args, name, method
        self.S, self.B, self.T, self.groups, self.n_cond = S, B, T, groups, n_cond
        self.perms, self.boots, self.cpu_each = perms, boots, cpu_each
        self.unit = 'resamples/s'

    def describe(self, world):
        if self.method == 'behavioral':
            return 'behavioral_pls X({}x{}) Y({}x{}) fp64, n_split=0, test_split=0'.format(
                self.S, self.B, self.S, self.T)
        return 'meancentered_pls X({}x{}) groups={} n_cond={} mean_centering=0 fp64'.format(
            self.S, self.B, self.groups, self.n_cond)

    def setup(self, rank, n_steps, dev):
        import torch
        from pypyls_amd import resampling, hostmath
        from pypyls_amd.engine import Engine
        S, B, T = self.S, self.B, self.T
        rs = np.random.RandomState(0)
        if self.method == 'behavioral':
            self.X, self.Y = synth(S, B, T)
        else:
            self.X = rs.randn(S, B)
            self.X += 0.25 * rs.randn(len(self.groups) * self.n_cond, B)[
                resampling.cell_of_row(self.groups, self.n_cond)]
            self.Y = None
        # long-lived engine: fixed super-batch scratch, mapped during warm-up
        eng = self.eng = Engine(**_engine_kwargs())
        eng.set_data(self.X, self.Y, resampling.cell_of_row(self.groups, self.n_cond), len(self.groups),
                     self.n_cond, 0 if self.method == 'behavioral' else 1)
        xw, sv, yw = eng.decompose()
        self.xw, self.yw = hostmath.sign_convention(xw, yw)
        self.sv = sv
        eng.set_original(self.xw, sv, self.yw)
        L, Tp = eng.L, eng.Tp
        self.Tp, self.L = Tp, L
        self.dev = dev
        if self.args.mode == 'weak':
            # distinct index arrays per rank and per step, resident in HBM before timing
            self.perm_idx, self.boot_idx = [], []
            for s in range(n_steps):
                seed = 1234 + 1000 * rank + s
                self.perm_idx.append(eng.index_tensor(
                    resampling.gen_permsamp(self.groups, self.n_cond, self.perms, seed=seed, verbose=False)))
                self.boot_idx.append(eng.index_tensor(
                    resampling.gen_bootsamp(self.groups, self.n_cond, self.boots, seed=seed + 500,
                                            verbose=False)))
            pmax, rmax = self.perms, self.boots
        else:
            from pypyls_amd import parallel
            _, world = parallel.rank_world()
            pmax = parallel.shard_bounds(self.perms, 0, world)[1]
            rmax = max(sum(hi - lo for lo, hi in parallel.shard_chunks(self.boots, r, world)) for r in range(world))
        self.pmax, self.rmax = pmax, rmax
        self.out_sv = torch.zeros((pmax, L), dtype=torch.float64, device=dev)
        self.dist_out = torch.zeros((rmax, Tp, L), dtype=torch.float64, device=dev)
        self.usum = torch.zeros((B, L), dtype=torch.float64, device=dev)
        self.usq = torch.zeros((B, L), dtype=torch.float64, device=dev)
        self.legs = []

    def units_per_step(self, world):
        return (self.perms + self.boots) * (world if self.args.mode == 'weak' else 1)

    def step(self, i, timed=False):
        import torch
        from pypyls_amd import parallel
        eng = self.eng
        ev = None
        if timed:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
        self.usum.zero_()
        self.usq.zero_()
        if self.args.mode == 'weak':
            eng.set_perm_path(None)                     # one step = one analysis: it forms its own S x S kernel
            eng.perm_into(self.perm_idx[i], self.out_sv, rotate=True)
            if ev:
                ev[1].record()
            eng.boot_begin(self.boots)                  # the step's bootstraps are one series (plsx_boot_begin)
            eng.boot_into(self.boot_idx[i], self.usum, self.usq, self.dist_out)
            eng.boot_finish(self.usum, self.usq)
        else:
            self._strong_step(i)
            if ev:
                ev[1].record()
        if ev:
            ev[2].record()
        # the ONE collective of the analysis
        self.result = parallel.gather_device([(self.out_sv, self.pmax), (self.dist_out, self.rmax)],
                                             [self.usum, self.usq])
        if ev:
            ev[3].record()
            self.legs.append(ev)

    def _strong_step(self, i, rank=None, world=None):
        """One whole analysis, the way the front-end runs it (pypyls_amd/plsc.py): ONE
        RandomState drawn on a host thread in the reference's order -- permutation arrays,
        then bootstrap arrays (every rank draws the same full arrays from the same seed) --
        while this rank ships its shard (permutations: contiguous; bootstraps: chunk-cyclic,
        parallel.shard_chunks) to the device chunk by chunk as the rows become final
        (resampling.IndexStream), permutations first, then bootstraps; the
        S x S kernel of the dual permutation route is formed once per analysis."""
        from pypyls_amd import parallel, resampling
        eng = self.eng
        if rank is None:
            rank, world = parallel.rank_world()
        ps = resampling.IndexStream('perm', self.groups, self.n_cond, self.perms)
        bs = resampling.IndexStream('boot', self.groups, self.n_cond, self.boots)
        draws = resampling.DrawThread(np.random.RandomState(4321 + i), [ps.draw, bs.draw]).start()
        try:
            eng.set_perm_path(None)
            lo, hi = parallel.shard_bounds(self.perms, rank, world)
            for a, b in ps.chunks(lo, hi):
                eng.perm_into(eng.rows_tensor(ps.rows[a:b]), self.out_sv[a - lo:b - lo], rotate=True)
            off = 0
            bchunks = parallel.shard_chunks(self.boots, rank, world)           # chunk-cyclic share
            eng.boot_begin(sum(hi - lo for lo, hi in bchunks))
            for lo, hi in bchunks:
                for a, b in bs.chunks(lo, hi):
                    eng.boot_into(eng.rows_tensor(bs.rows[a:b]), self.usum, self.usq,
                                  self.dist_out[off + a - lo:off + b - lo])
                off += hi - lo
            eng.boot_finish(self.usum, self.usq)
        finally:
            draws.thread.join()
        if draws.error is not None:
            raise draws.error

    def emulate(self, worlds, reps=2):
        """Critical path of ONE analysis on rank r of an emulated world N, measured on this
        one GPU: the rank draws the full index arrays (as every rank of a real run does),
        runs only its shard, and the all-gather is skipped (no peers).  Both end ranks are
        timed -- rank N-1 owns the rows that are drawn last -- and the slower one counts.
        Returns {N: {'rank0_ms', 'last_rank_ms', 'critical_path_ms'}}."""
        import torch
        out = {}
        for n in worlds:
            row = {}
            for label, r in (('rank0_ms', 0), ('last_rank_ms', n - 1)):
                best = None
                for rep in range(reps + 1):                 # first pass warms the allocator
                    self.usum.zero_()
                    self.usq.zero_()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    self._strong_step(1000 + rep, rank=r, world=n)
                    torch.cuda.synchronize()
                    dt = 1e3 * (time.perf_counter() - t0)
                    if rep > 0:
                        best = dt if best is None else min(best, dt)
                row[label] = best
                if n == 1:
                    row['last_rank_ms'] = best
                    break
            row['critical_path_ms'] = max(row['rank0_ms'], row['last_rank_ms'])
            out[n] = row
        return out

    def roofline(self, kt, steps, world):
        """Dominant kernel k_xprod.  Algorithmic work per resample of THIS kernel:
        2 S T' B flop (first term of SURVEY 8d W_F) and 8 S (B + T') bytes (W_B)."""
        S, B, Tp = self.S, self.B, self.Tp
        tm = self.eng.last_timing()
        if tm.get('quad_series', 0) > 0:
            return self._roofline_quad(kt, steps, tm)
        launches = max(int(tm.get('xprod_launches', 0)), 1)
        avg_ms = tm.get('xprod_ms', 0.0) / launches
        units = tm.get('xprod_resamples', 0) / launches
        fl = 2.0 * S * Tp * B * units
        by = 8.0 * S * (B + Tp) * units
        tf = fl / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        tb = by / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        f_m, f_h = tf / PEAK_FP64_MFMA_TFLOPS, tb / PEAK_HBM_TBS
        # the resamples of a group share one pass over X, so the algorithmic-byte rate may
        # exceed the HBM peak; a kernel is not bound by a roof it exceeds -> MFMA then
        mf = f_m >= f_h or f_h > 1.0
        mom_ms = kt.get('k_xprod_moments', (0.0, 0))[0] / max(steps, 1)
        kname = 'k_xprod<{}>'.format(int(tm.get('m_tiles', 0)))
        if mom_ms > 0:
            kname += ' (data-only blocks; the feature moments of all (resample, cell) pairs come from moment-only ' \
                     'blocks, k_xprod_moments: {:.2f} ms per step, not in this kernel\'s time)'.format(mom_ms)
        out = {'bound': 'mfma' if mf else 'hbm', 'kernel': kname,
               'achieved': tf if mf else tb * 1e3, 'peak': PEAK_FP64_MFMA_TFLOPS if mf else PEAK_HBM_TBS * 1e3,
               'unit': 'TFLOP/s' if mf else 'GB/s', 'frac': f_m if mf else f_h,
               'frac_mfma': f_m, 'hbm_algorithmic_over_peak': f_h,
               'avg_launch_ms': avg_ms, 'launches': launches, 'resamples_per_launch': units,
               'work_per_resample': '2 S T\' B = {:.3e} flop (SURVEY 8d, first term of W_F); 8 S (B + T\') = {:.3e} B (W_B; '
                                    'the resamples of a block share one pass over X, so hbm_algorithmic_over_peak may pass 1 -- '
                                    'it is a ratio of the per-resample model to the peak, not a utilisation)'.format(
                                        2.0 * S * Tp * B, 8.0 * S * (B + Tp))}
        self.row_fraction = 1.0
        crow = tm.get('compact_row_fraction', 0.0)
        if crow > 0:
            out['kernel'] = out['kernel'].replace('k_xprod<', 'k_xprod_compact<', 1).replace(
                'data-only blocks', 'one bootstrap per block, contraction over the rows it draws')
            # compact blocks: one bootstrap per block, contraction over the DISTINCT rows it draws (k-steps of
            # 4 rows), T' rows on ceil(T'/16) tiles whose last one runs on the 4x4x4 shape when it holds <= 4.
            # The formulation needs 2 (crow S) T' B flop per bootstrap -- undrawn rows have weight zero, rows drawn
            # twice fold into one column of A -- and THAT is what frac prices; the dense 2 S T' B of SURVEY 8d is
            # kept as dense_equivalent_tflops with the ratio between the two as algorithmic_speedup_vs_dense.
            self.row_fraction = crow
            mt = -(-Tp // 16)
            rows = (mt - 1) * 16 + 4 if (mt >= 2 and Tp - (mt - 1) * 16 <= 4) else mt * 16
            issued = 2.0 * S * crow * rows * (B + self.L) * units
            t_iss = issued / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
            need = tf * crow
            out.update({'achieved': need, 'frac': need / PEAK_FP64_MFMA_TFLOPS, 'frac_mfma': need / PEAK_FP64_MFMA_TFLOPS,
                        'distinct_row_fraction': crow, 'algorithmic_speedup_vs_dense': 1.0 / crow,
                        'dense_equivalent_tflops': tf,
                        'issued_tflops': t_iss, 'frac_issued': t_iss / PEAK_FP64_MFMA_TFLOPS,
                        'work_per_resample': '2 (r S) T\' B = {:.3e} flop with r = {:.3f} the share of the S rows a bootstrap '
                                             'contracts (distinct rows, in k-steps of 4): the min-flop formulation; dense '
                                             'SURVEY 8d figure 2 S T\' B = {:.3e}'.format(2.0 * S * crow * Tp * B, crow,
                                                                                         2.0 * S * Tp * B),
                        'note': 'frac = min-flop work / time / peak (a true fraction); frac_issued = flop the kernel issues '
                                '(T\' padded to {} rows, score columns) / time / peak = matrix-pipe utilisation; '
                                'dense_equivalent_tflops = what a dense contraction over all S rows would have had to '
                                'sustain'.format(rows)})
        return out

    def _roofline_quad(self, kt, steps, tm):
        """Unscaled modes on the quadratic-form route: per bootstrap only dual-space products (S x S kernel,
        k_nt_gemm), per SERIES one pass over the features (k_xprod EPI 7: x_j^T C_l x_j).  The dominant kernel by
        time is reported with its own work; the closing pass beside it."""
        S, B, Tp, L = self.S, self.B, self.Tp, self.L
        self.row_fraction = 1.0
        self.quad_route = True
        mt, gpl = int(tm.get('quad_m_tiles', 24)), int(tm.get('quad_blocks_per_lv', 1))
        need, issued = quad_closing_work(S, mt, gpl)
        x_ms, x_n = kt.get('k_xprod', (0.0, 0))
        nt_ms, nt_n = kt.get('k_nt_gemm', (0.0, 0))
        series = max(int(tm.get('quad_series', 1)), 1)
        x_need = need * L * B * series / (x_ms * 1e-3) / 1e12 if x_ms > 0 else 0.0
        x_iss = issued * L * B * series / (x_ms * 1e-3) / 1e12 if x_ms > 0 else 0.0
        nt_tf = tm.get('nt_flops', 0.0) / (nt_ms * 1e-3) / 1e12 if nt_ms > 0 else 0.0
        dom = max(kt, key=lambda n: kt[n][0]) if kt else 'k_nt_gemm'
        closing = {'kernel': 'k_xprod<{},8,KT,0,7> (KT = 2 k-steps per stage when the padded row count is even, else 1; closing pass of a bootstrap series: x_j^T C_l x_j, {} row blocks per LV)'.format(mt, gpl),
                   'ms_per_series': x_ms / series, 'achieved': x_need, 'frac': x_need / PEAK_FP64_MFMA_TFLOPS,
                   'issued_tflops': x_iss, 'frac_issued': x_iss / PEAK_FP64_MFMA_TFLOPS,
                   'work': 'S (S + 1) L B = {:.3e} flop per series (upper triangle of L symmetric S x S forms per column), '
                           '{:.3e} issued (row blocks from their own first row on)'.format(need * L * B, issued * L * B)}
        direct = 2.0 * S * L * B * self.boots          # what the per-bootstrap feature pass of one step multiplies
        if dom == 'k_xprod':
            out = dict(closing)
            out.update({'bound': 'mfma', 'peak': PEAK_FP64_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'avg_launch_ms': x_ms / max(x_n, 1),
                        'launches': x_n})
        else:
            out = {'bound': 'mfma', 'kernel': 'k_nt_gemm (dual-space products of the batch: A K, W A^T, A Sc per resample, '
                                              'C_l += V_l^T V_l per batch; LDS-tiled 64 x 64 blocks, contraction 200 - 1000)',
                   'achieved': nt_tf, 'peak': PEAK_FP64_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': nt_tf / PEAK_FP64_MFMA_TFLOPS,
                   'avg_launch_ms': nt_ms / max(nt_n, 1), 'launches': nt_n,
                   'work': '{:.3e} flop in the timed launches (products issued; symmetric ones by half)'.format(tm.get('nt_flops', 0.0)),
                   'closing_pass': closing}
        out.update({'dominant_by_time': dom, 'route': 'quadratic form (plsx_boot_begin / plsx_boot_finish)',
                    'frac_mfma': out['frac'], 'hbm_algorithmic_over_peak': None,
                    'per_bootstrap_pass_flop_per_step': direct,
                    'algorithmic_speedup_vs_per_bootstrap_pass': direct / max(need * L * B, 1.0),
                    'note': 'the bootstrap weights are linear in the fixed features here (U_b = Xc^T V_b): sum U_b^2 = '
                            'x_j^T (sum_b v v^T) x_j, one feature pass per series instead of one per bootstrap'})
        return out

    def pipeline(self, ms_per_step, primal):
        """Whole-step fractions with the work the formulation NEEDS (min-flop): per bootstrap
        r 2 S T' B (cross-product over the r S rows it contracts) + 4 S J B (feature moments, all rows)
        + 2 T'^2 B (Gram) + 2 T' L B (R U0) + 2 T' L B (R^T M); per feature-pass permutation
        2 S T' B + 2 T'^2 B; per dual permutation 2 T' S^2 + 2 T'^2 S + its share of K = X X^T."""
        S, B, Tp, L = self.S, self.B, self.Tp, self.L
        if getattr(self, 'quad_route', False) and not primal:
            # quadratic-form route: S (S + 1) L B per series + per resample the dual-space products
            per = 2.0 * Tp * S * S + 2.0 * Tp * Tp * S + 2.0 * Tp * L * S
            wf = (self.perms + self.boots) * per + self.boots * 1.0 * S * S * L + float(S) * (S + 1) * L * B + 2.0 * S * S * B
            self.pipeline_formula = ('quadratic-form route: per resample 2 T\' S^2 + 2 T\'^2 S + 2 T\' L S (dual space), per '
                                     'bootstrap S^2 L (C_l, symmetric), per step S (S + 1) L B (closing pass) + 2 S^2 B '
                                     '(K = X X^T) = {:.3e} flop'.format(wf))
            return (1e3 / ms_per_step * wf / (PEAK_FP64_MFMA_TFLOPS * 1e12),
                    1e3 / ms_per_step * 8.0 * (2.0 * S * B) / (PEAK_HBM_TBS * 1e12))
        r = getattr(self, 'row_fraction', 1.0)
        mom = 4.0 * S * len(self.groups) * self.n_cond * B if self.method == 'behavioral' else 0.0
        wf_boot = r * 2.0 * S * Tp * B + mom + 2.0 * Tp * Tp * B + 4.0 * Tp * L * B
        wb = 8.0 * S * (B + Tp)
        if primal:
            wf_perm, wb_perm = 2.0 * S * Tp * B + 2.0 * Tp * Tp * B, wb
        else:
            wf_perm = 2.0 * Tp * S * S + 2.0 * Tp * Tp * S + 2.0 * S * S * B / max(self.perms, 1)
            wb_perm = 8.0 * (S * B / max(self.perms, 1) + 3.0 * Tp * S)
        per_s = 1e3 / ms_per_step
        self.pipeline_formula = ('per bootstrap {:.3f} x 2 S T\' B + 4 S J B + 2 T\'^2 B + 4 T\' L B = {:.3e} flop; per '
                                 'permutation {:.3e} flop ({})'.format(r, wf_boot, wf_perm,
                                                                      'feature pass' if primal else 'dual S x S route'))
        return (per_s * (self.perms * wf_perm + self.boots * wf_boot) / (PEAK_FP64_MFMA_TFLOPS * 1e12),
                per_s * (self.perms * wb_perm + self.boots * wb) / (PEAK_HBM_TBS * 1e12))

    def cpu_baseline(self):
        from oracle import cpu_ref as ref
        from pypyls_amd import resampling
        n = self.cpu_each
        spec = ref.Spec(self.method, self.groups, self.n_cond)
        Y = self.Y if self.Y is not None else spec.dummy.astype(float)
        perms = resampling.gen_permsamp(self.groups, self.n_cond, n, seed=1234, verbose=False)
        boots = resampling.gen_bootsamp(self.groups, self.n_cond, n, seed=1235, verbose=False)
        t0 = time.perf_counter()
        for i in range(n):
            ref.single_perm(spec, self.X, Y, perms[:, i], self.yw)
        for i in range(n):
            ref.single_boot(spec, self.X, Y, boots[:, i], self.xw, np.diag(self.sv))
        dt = time.perf_counter() - t0
        return 2 * n / dt, '{} permutations + {} bootstraps of the same workload through oracle/cpu_ref.py ' \
                           '(numpy, BLAS threads = host cores), {:.1f} s'.format(n, n, dt)


class Simpls(object):
    """c5: pls_regression (SIMPLS) X (1000 x 100000), Y (1000 x 20), k = 15."""

    def __init__(self, args):
        self.args, self.name = args, 'c5'
        self.S, self.B, self.T, self.k = 1000, 100000, 20, 15
        self.perms, self.boots = args.perms or 5000, args.boots or 5000          # BASELINE configs[4], literally
        self.unit = 'resamples/s'

    def describe(self, world):
        return 'pls_regression (SIMPLS) X({}x{}) Y({}x{}) n_components={} fp64 (T = 20 > 11: pinned on the ' \
               'oracle\'s exact SIMPLS, reference parity unpinned)'.format(self.S, self.B, self.S, self.T, self.k)

    def setup(self, rank, n_steps, dev):
        import torch
        from pypyls_amd import resampling
        from pypyls_amd.engine import Engine
        S, B, T, k = self.S, self.B, self.T, self.k
        X, Y = synth(S, B, T)
        self.Xc = X - X.mean(axis=0, keepdims=True)
        self.Yc = Y - Y.mean(axis=0, keepdims=True)
        eng = self.eng = Engine(**_engine_kwargs())
        eng.set_data_regression(self.Xc, self.Yc, k)
        W, pct, cvec, _ = eng.simpls_decompose()
        idx = np.argmax(np.abs(W), axis=0)
        sg = np.sign(W[idx, np.arange(k)])
        self.W = W * sg
        eng.simpls_set_original(self.W)
        self.perm_idx, self.boot_idx = [], []
        for s_ in range(n_steps):
            seed = 1234 + 1000 * rank + s_
            self.perm_idx.append(eng.index_tensor(resampling.gen_permsamp([S], 1, self.perms, seed=seed,
                                                                          verbose=False)))
            bs = resampling.gen_bootsamp([S], 1, self.boots, seed=seed + 500, verbose=False)
            if s_ == 0:          # share of the S rows a bootstrap draws at least once (weight > 0 in X0_r^T Wd)
                self.row_fraction = float(np.mean([np.unique(bs[:, j]).size for j in range(bs.shape[1])])) / S
            self.boot_idx.append(eng.index_tensor(bs))
        self.out = torch.zeros((self.perms, k), dtype=torch.float64, device=dev)
        self.yl = torch.zeros((self.boots, T, k), dtype=torch.float64, device=dev)
        self.usum = torch.zeros((B, k), dtype=torch.float64, device=dev)
        self.usq = torch.zeros((B, k), dtype=torch.float64, device=dev)
        self.legs = []

    def units_per_step(self, world):
        return (self.perms + self.boots) * world

    def step(self, i, timed=False):
        import torch
        from pypyls_amd import parallel
        ev = None
        if timed:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
        self.usum.zero_()
        self.usq.zero_()
        self.eng.simpls_perm_into(self.perm_idx[i], self.out)
        if ev:
            ev[1].record()
        self.eng.boot_begin(self.boots)                 # the step's bootstraps are one series (plsx_boot_begin)
        self.eng.simpls_boot_into(self.boot_idx[i], self.usum, self.usq, self.yl)
        self.eng.boot_finish(self.usum, self.usq)
        if ev:
            ev[2].record()
        self.result = parallel.gather_device([(self.out, self.perms), (self.yl, self.boots)],
                                             [self.usum, self.usq])
        if ev:
            ev[3].record()
            self.legs.append(ev)

    def roofline(self, kt, steps, world):
        """Dominant kernel by summed time.  With the K products of a batch of
        resamples done as GEMMs (plsx_simpls.h) the dual-space solver is no longer
        it: the bootstrap's B-long weights x_weights = X0_r^T Wd (k x B per
        bootstrap, 2 S k B flop, the only pass over the features) are -- k_xprod,
        MFMA bound.  The solver's own algorithmic work in this formulation,
        (T + 1 + k) products with K of 2 S^2 flop each per resample, is reported
        beside it, and SURVEY 8d's primal model (1 + 2k passes over X) for scale."""
        S, T, k, B = self.S, self.T, self.k, self.B
        dom = max(kt, key=lambda n: kt[n][0]) if kt else 'k_xprod'
        ms, n = kt.get('k_xprod', (0.0, 0))
        fl = 2.0 * S * k * B * steps * self.boots
        tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        sms = kt.get('k_simpls_dual', (0.0, 0))[0] + kt.get('k_nt_gemm', (0.0, 0))[0]
        sfl = (T + 1.0 + k) * 2.0 * S * S * steps * (self.perms + self.boots)
        # min-flop: a bootstrap's weights only need the r S rows it draws (the others have weight zero in Wd), so
        # frac prices r x the dense work; the kernel is the DENSE grouped product on purpose (25 bootstraps x k = 15
        # rows fill 24 tiles and share every X fragment; one bootstrap alone is ONE 16-row tile, so a compact block
        # would load an X fragment per two MFMAs, 12 x the cache traffic per flop of the dense blocks: 505 GB per
        # 1000 bootstraps through L2): frac_issued = what the matrix pipe does.
        r = getattr(self, 'row_fraction', 1.0)
        tm = self.eng.last_timing()
        if tm.get('quad_series', 0) > 0:
            # quadratic-form route: ONE feature pass per series (k_xprod EPI 7), the rest in dual space
            mt, gpl = int(tm.get('quad_m_tiles', 24)), int(tm.get('quad_blocks_per_lv', 1))
            need, issued = quad_closing_work(S, mt, gpl)
            series = max(int(tm['quad_series']), 1)
            x_need = need * k * B * series / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            x_iss = issued * k * B * series / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            out = {'bound': 'mfma', 'kernel': 'k_xprod<{},8,KT,0,7> (KT = 2 k-steps per stage when the padded row count is even, else 1; closing pass of the bootstrap series: sum_b w_b[j,c]^2 = '
                                              'x_j^T C_c x_j, {} row blocks per component)'.format(mt, gpl),
                   'dominant_by_time': dom, 'achieved': x_need, 'peak': PEAK_FP64_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                   'frac': x_need / PEAK_FP64_MFMA_TFLOPS, 'issued_tflops': x_iss, 'frac_issued': x_iss / PEAK_FP64_MFMA_TFLOPS,
                   'avg_launch_ms': ms / max(n, 1), 'launches': n, 'route': 'quadratic form (plsx_boot_begin / plsx_boot_finish)',
                   'work': 'S (S + 1) k B = {:.3e} flop per series (upper triangle of k symmetric S x S forms per column), '
                           '{:.3e} issued'.format(need * k * B, issued * k * B),
                   'per_bootstrap_pass_flop_per_step': fl / max(steps, 1),
                   'algorithmic_speedup_vs_per_bootstrap_pass': fl / max(steps, 1) / (need * k * B),
                   'dual_solver_ms_per_resample': sms / max(steps * (self.perms + self.boots), 1),
                   'dual_solver_tflops': sfl / (sms * 1e-3) / 1e12 if sms > 0 else 0.0,
                   'note': 'frac = needed flop of the closing pass / its time / peak; the per-bootstrap feature pass it '
                           'replaces multiplies 2 S k B per bootstrap'}
            return out
        return {'bound': 'mfma', 'kernel': 'k_xprod<24> (bootstrap x_weights = X0_r^T Wd)',
                'dominant_by_time': dom, 'achieved': tf * r, 'peak': PEAK_FP64_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': tf * r / PEAK_FP64_MFMA_TFLOPS, 'frac_issued': tf / PEAK_FP64_MFMA_TFLOPS,
                'issued_tflops': tf, 'distinct_row_fraction': r,
                'note': 'frac = min-flop work (2 r S k B per bootstrap, r = share of rows drawn) / time / peak; '
                        'frac_issued = the dense 2 S k B the grouped kernel issues / time / peak',
                'avg_launch_ms': ms / max(n, 1), 'launches': n,
                'dual_solver_ms_per_resample': sms / max(steps * (self.perms + self.boots), 1),
                'dual_solver_tflops': sfl / (sms * 1e-3) / 1e12 if sms > 0 else 0.0,
                'primal_model_hbm_resamples_per_s': PEAK_HBM_TBS * 1e12 / ((1 + 2 * k) * 8.0 * S * B)}

    def pipeline(self, ms_per_step, primal):
        return None, None

    def cpu_baseline(self):
        from oracle import cpu_ref as ref
        from pypyls_amd import resampling
        perms = resampling.gen_permsamp([self.S], 1, 2, seed=1234, verbose=False)
        boots = resampling.gen_bootsamp([self.S], 1, 2, seed=1235, verbose=False)
        t0 = time.perf_counter()
        for i in range(2):
            ref.regression_single_perm(self.Xc, self.Yc, perms[:, i], self.k)
        for i in range(2):
            ref.regression_single_boot(self.Xc, self.Yc, boots[:, i], self.k, self.W)
        dt = time.perf_counter() - t0
        return 4 / dt, '2 permutations + 2 bootstraps through oracle/cpu_ref.py (primal SIMPLS, numpy), ' \
                       '{:.1f} s'.format(dt)


class SplitHalf(object):
    """c4 split-half leg: permuted arrangements x n_split = 100 splits."""

    def __init__(self, args):
        self.args, self.name = args, 'c4split'
        self.S, self.B, self.T = args.S, args.B, args.T
        # weak: `arr` arrangements per rank and step; strong: ONE set of `arr` arrangements per step, sharded over
        # the ranks (contiguous slices, parallel.shard_bounds -- what the front-end does with the permutations of a
        # call with n_split, pypyls_amd/plsc.py)
        self.strong = args.mode == 'strong'
        self.arr, self.ns = args.perms or (64 if self.strong else 8), getattr(args, 'n_split', 0) or 100
        self.unit = 'splits/s'

    def describe(self, world):
        return 'behavioral_pls X({}x{}) Y({}x{}) fp64 split-half leg: {} permuted arrangements x n_split={} ' \
               'per step (each arrangement decomposed on the device first)'.format(
                   self.S, self.B, self.S, self.T, self.arr, self.ns)

    def setup(self, rank, n_steps, dev):
        import torch
        from pypyls_amd import resampling
        from pypyls_amd.engine import Engine
        S, B, T = self.S, self.B, self.T
        self.X, self.Y = synth(S, B, T)
        eng = self.eng = Engine(**_engine_kwargs())
        eng.set_data(self.X, self.Y, resampling.cell_of_row([S], 1), 1, 1, 0)
        self.L = eng.L
        self.perm_idx, self.masks = [], []
        from pypyls_amd import parallel
        self.rank, self.world = parallel.rank_world()
        self.lo, self.hi = parallel.shard_bounds(self.arr, self.rank, self.world) if self.strong else (0, self.arr)
        self.nmax = parallel.shard_bounds(self.arr, 0, self.world)[1] if self.strong else self.arr
        for s in range(n_steps):
            seed = 77 + (0 if self.strong else 1000 * rank) + s      # strong: every rank draws the SAME arrays
            self.perm_idx.append(eng.index_tensor(resampling.gen_permsamp([S], 1, self.arr, seed=seed,
                                                                          verbose=False)))
            m = np.stack([resampling.gen_splits([S], 1, self.ns, seed=seed * 131 + i) for i in range(self.arr)])
            self.masks.append(torch.from_numpy(np.ascontiguousarray(m.transpose(0, 2, 1), dtype=np.uint8)).to(dev))
        self.uc = torch.zeros((self.arr, self.ns, self.L), dtype=torch.float64, device=dev)
        self.vc = torch.zeros((self.arr, self.ns, self.L), dtype=torch.float64, device=dev)
        self.legs = []

    def units_per_step(self, world):
        return self.arr * self.ns * (1 if self.strong else world)

    def step(self, i, timed=False):
        import torch
        from pypyls_amd import parallel
        ev = None
        if timed:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
        lo, hi = self.lo, self.hi
        if hi > lo:
            self.eng.split_half_into(self.perm_idx[i][lo:hi], self.masks[i][lo:hi], self.uc[:hi - lo], self.vc[:hi - lo])
        if ev:
            ev[1].record()
            ev[2].record()
        self.result = parallel.gather_device([(self.uc[:hi - lo], self.nmax), (self.vc[:hi - lo], self.nmax)], [])
        if ev:
            ev[3].record()
            self.legs.append(ev)

    def roofline(self, kt, steps, world):
        """Whole split against SURVEY 8d's W_F(split) = 2 S T' B + 8 T' L B flop
        (two half-sample cross-products + four projections), MFMA bound; the
        dominant kernel (largest summed time) is named with its share."""
        S, B, Tp, L = self.S, self.B, self.T, self.L
        tot = sum(v[0] for v in kt.values()) or 1.0
        dom = max(kt, key=lambda k: kt[k][0]) if kt else 'k_xprod'
        units = steps * self.arr * self.ns
        wf_dense = (2.0 * S * Tp * B + 8.0 * Tp * L * B) * units
        # min-flop: the fused pass contracts over the FIRST half only (gen_splits: ceil / floor of S / 2 rows; the second
        # half follows from the arrangement's full cross-product), feature moments of that half 4 (S/2) J B, the four
        # projections 8 T' L B as in SURVEY's W_F(split)
        wf = (0.5 * 2.0 * S * Tp * B + 2.0 * S * B + 8.0 * Tp * L * B) * units
        tf = wf / (tot * 1e-3) / 1e12
        # the two large kernels on their own work (HIP events of the library, launch stream): the writer contracts the
        # first half (0.5 x 2 S T' B per split), the one-pass reader multiplies G_h = D_h R_p^T and E_h = vd^T D_h of both
        # halves (8 T' L B per split; on the two-reader route the same work is k_gram4 + k_ucorr_partial)
        one_pass = bool(self.eng.split_route())
        per_kernel = {}
        for key, label, w in (('k_xprod', 'k_xprod_compact<4,3,8,true> (raw first-half sums, one split per block)'
                               if one_pass else 'k_xprod_compact<4,3,5,true> (both z-scored halves)', 0.5 * 2.0 * S * Tp * B),
                              ('k_ucorr_partial', 'k_split_fused12<13> (one reader pass per pair of splits; 12-wave block with dedicated construction waves)' if one_pass
                               else 'k_ucorr_partial (projections; the cross-Gram is timed under k_gram)', 8.0 * Tp * L * B)):
            if key in kt and kt[key][0] > 0:
                ms, n = kt[key]
                per_kernel[key] = {'kernel': label, 'avg_launch_ms': ms / max(n, 1), 'launches': n,
                                   'achieved_tflops': w * units / (ms * 1e-3) / 1e12,
                                   'frac': w * units / (ms * 1e-3) / 1e12 / PEAK_FP64_MFMA_TFLOPS}
        if one_pass:
            dom = per_kernel.get(dom, {}).get('kernel', dom)
        return {'bound': 'mfma', 'kernel': dom, 'achieved': tf, 'peak': PEAK_FP64_MFMA_TFLOPS, 'kernels': per_kernel,
                'route': 'one reader pass over raw first-half sums' if one_pass else 'two readers over both halves',
                'unit': 'TFLOP/s', 'frac': tf / PEAK_FP64_MFMA_TFLOPS,
                'dense_equivalent_tflops': wf_dense / (tot * 1e-3) / 1e12,
                'algorithmic_speedup_vs_dense': wf_dense / wf,
                'dominant_kernel_share': kt.get(dom, (0, 0))[0] / tot,
                'work_per_split': '0.5 x 2 S T\' B + 2 S B + 8 T\' L B = {:.3e} flop (min-flop: one half through the matrix '
                                  'pipe); SURVEY 8d W_F(split) = 2 S T\' B + 8 T\' L B = {:.3e} (dense)'.format(
                                      wf / units, wf_dense / units),
                'note': 'achieved = min-flop work x splits / summed kernel time of the split-half launches'}

    def pipeline(self, ms_per_step, primal):
        return None, None

    def cpu_baseline(self):
        from oracle import cpu_ref as ref
        from pypyls_amd import resampling
        spec = ref.Spec('behavioral', [self.S], 1)
        masks = resampling.gen_splits([self.S], 1, 2, seed=5)
        U, d, V = ref.decompose(spec, self.X, self.Y)
        di = np.linalg.inv(d)
        t0 = time.perf_counter()
        ref.split_half(spec, self.X, self.Y, U @ di, V @ di, masks)
        dt = time.perf_counter() - t0
        return 2 / dt, '2 splits of the original arrangement through oracle/cpu_ref.py split_half (numpy), ' \
                       '{:.1f} s'.format(dt)


def run_analysis(args, world=1, rank=0, dev=None, backend='nccl'):
    """--mode analysis: END-TO-END wall time of the PUBLIC front-end call (the counterpart of BasePLS.run_pls,
    pyls/base.py:341-399 + behavioral.py:197-227) at --config c4 / c2 / c3 / c5 with --perms + --boots resamples
    (default: the literal 10 000 + 10 000; --splits S adds split-half with n_split = S): index generation,
    H2D of X, binding, original decomposition, resampling, THE collective, percentile intervals, bootstrap
    ratios and the D2H of what PLSResults holds -- everything the resampling-step modes leave out.  A fresh
    engine per call, as a user's call gets (after one small warm-up call: loaded code objects, a warm
    allocator).  --emulate-world 1,2,4,8 (one GPU): the same call as rank 0 and as rank N - 1 of an emulated
    world N -- every rank draws the full index arrays, runs its shard, and the all-gather is replaced by a
    device-side surrogate of the same volume followed by the real rank-ordered sums
    (parallel._surrogate_gather) -- an EMULATION of the end-to-end critical path, not a hardware curve.
    One profiled call per world (a device sync at every phase boundary) gives the per-phase split.
    Under N REAL ranks (torchrun / --gpus N) every rank makes the same call collectively -- the front-end shards the
    resamples and runs THE all-gather over the process group --, the clock is bracketed by barriers and the slowest
    rank counts; the emulation is skipped.  Returns the record (rank 0) or None."""
    import torch
    import torch.distributed as dist
    multi = world > 1 and dist.is_initialized()
    import pypyls_amd as pls
    cfg = args.config
    if cfg == 'c4':
        S, B, T, groups, n_cond = args.S, args.B, args.T, [args.S], 1
        X, Y = synth(S, B, T)
        call = lambda **kw: pls.behavioral_pls(X, Y, test_split=0, verbose=False, **kw)
        desc = 'behavioral_pls X({}x{}) Y({}x{}) fp64, test_split=0'.format(S, B, S, T)
    elif cfg == 'c2':
        S, B, T = 80, 10000, 10
        X, Y = synth(S, B, T)
        call = lambda **kw: pls.behavioral_pls(X, Y, test_split=0, verbose=False, **kw)
        desc = 'behavioral_pls X(80x10000) Y(80x10) fp64, test_split=0'
    elif cfg == 'c3':
        rs = np.random.RandomState(0)
        groups, n_cond = [25, 25, 25, 25], 2
        X = rs.randn(200, 50000)
        call = lambda **kw: pls.meancentered_pls(X, groups=groups, n_cond=n_cond, verbose=False, **kw)
        desc = 'meancentered_pls X(200x50000) groups=[25]*4 n_cond=2 fp64'
    elif cfg == 'c5':
        X, Y = synth(1000, 100000, 20)
        call = lambda **kw: pls.pls_regression(X, Y, n_components=15, verbose=False, **kw)
        desc = 'pls_regression (SIMPLS) X(1000x100000) Y(1000x20) n_components=15 fp64'
        if args.splits:
            raise SystemExit('--mode analysis --config c5: no --splits (pls_regression has no split-half)')
    else:
        raise SystemExit('--mode analysis supports --config c4 | c2 | c3 | c5')
    n_perm = args.perms or (5000 if cfg in ('c2', 'c5') else 10000)
    n_boot = args.boots or (5000 if cfg in ('c2', 'c5') else 10000)
    extra = {'n_split': args.splits} if args.splits else {}
    call(n_perm=64, n_boot=64, seed=1)                      # warm-up
    worlds = sorted({1} | {int(v) for v in args.emulate_world.split(',') if v.strip()}) if (args.emulate_world and not multi) else [1]
    reps = max(args.steps, 1)
    table = {}
    for n in worlds:
        row = {}
        for label, r in (('rank0', 0), ('last_rank', n - 1)):
            emu = {'_emulate': (r, n)} if n > 1 else {}
            best = None
            for rep in range(reps):
                torch.cuda.synchronize()
                if multi:
                    dist.barrier()
                t0 = time.perf_counter()
                res = call(n_perm=n_perm, n_boot=n_boot, seed=1234, **extra, **emu)
                torch.cuda.synchronize()
                if multi:
                    dist.barrier()
                dt = 1e3 * (time.perf_counter() - t0)
                if multi:                                   # the slowest rank counts
                    t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    dt = float(t.item())
                best = dt if best is None else min(best, dt)
            row[label + '_ms'] = best
            phases = {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            call(n_perm=n_perm, n_boot=n_boot, seed=1234, _phases=phases, **extra, **emu)
            row[label + '_profiled_ms'] = 1e3 * (time.perf_counter() - t0)
            row[label + '_phases_ms'] = {k: round(v, 2) for k, v in phases.items()}
            if multi:                                       # every rank's own phase split (the collective phase is where a fast rank waits)
                allph = [None] * world
                dist.all_gather_object(allph, {'rank': rank, **row[label + '_phases_ms']})
                row['per_rank_phases_ms'] = allph
            if n == 1:
                row['last_rank_ms'] = best
                break
        row['critical_path_ms'] = max(row['rank0_ms'], row['last_rank_ms'])
        table[n] = row
    base = table[1]['critical_path_ms']
    for n, row in table.items():
        row['ideal_ms'] = base / n
        row['over_ideal'] = row['critical_path_ms'] * n / base
        row['efficiency'] = base / (n * row['critical_path_ms'])
    ph1 = table[1]['rank0_phases_ms']
    resample_ms = sum(ph1.get(k, 0.0) for k in ('permutations', 'bootstraps', 'split_half'))
    out = {'metric': 'end-to-end resamples/sec of the public front-end call, {}'.format(desc),
           'value': (n_perm + n_boot) / (base * 1e-3), 'unit': 'resamples/s', 'n_gpus': world, 'steps': reps, 'warmup': 1,
           'ms_per_step': base, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64',
           'data': 'synthetic',
           'config': {'workload': '{} (BASELINE config {}): ONE analysis of {} permutations + {} bootstraps{} through '
                                  'the public call, seed-compatible index generation inside the clock'.format(
                                      desc, cfg, n_perm, n_boot, ' + n_split={}'.format(args.splits) if args.splits else ''),
                      'mode': 'analysis', 'per': 'analysis'},
           'fixed_cost_ms': {'profiled_total_ms': table[1]['rank0_profiled_ms'], 'resampling_phases_ms': resample_ms,
                             'everything_else_ms': table[1]['rank0_profiled_ms'] - resample_ms,
                             'unprofiled_total_ms': base,
                             'note': 'fixed cost = wall time of the call minus its permutation / bootstrap / split-half '
                                     'phases (one profiled call with a device sync at every phase boundary)'},
           'end_to_end_emulation': {
               'what': 'the public call as rank 0 and as rank N - 1 of a world emulated on ONE GPU: full index '
                       'generation, own shard, surrogate gather of the real volume, real rank-ordered sums, full '
                       'post-processing and D2H; critical path = slower of the two; efficiency = t(1) / (N t(N)).  '
                       'NOT a hardware scaling curve.',
               'worlds': {str(n): row for n, row in table.items()}}}
    if multi:
        out.pop('end_to_end_emulation')
        out['per_rank_phases_ms'] = table[1].get('per_rank_phases_ms')
        out['collective_ms'] = max(p.get('collective', 0.0) for p in out['per_rank_phases_ms']) if out['per_rank_phases_ms'] else None
        from pypyls_amd import parallel
        out['config']['collective'] = '{} over {} ranks, 1 per analysis (the front-end\'s own)'.format(
            parallel.collective_name(), world)
    pls.release_default_engine()
    return out if rank == 0 else None


def make_workload(args):
    c = args.config
    if c == 'c4':
        return PLSC(args, 'c4', 'behavioral', args.S, args.B, args.T, [args.S], 1,
                    args.perms or (10000 if args.mode == 'strong' else 1008),
                    args.boots or (10000 if args.mode == 'strong' else 1008), args.cpu_sample)
    if c == 'c2':
        return PLSC(args, 'c2', 'behavioral', 80, 10000, 10, [80], 1, args.perms or 5000, args.boots or 5000,
                    200)
    if c == 'c3':
        return PLSC(args, 'c3', 'meancentered', 200, 50000, 0, [25, 25, 25, 25], 2, args.perms or 10000,
                    args.boots or 10000, 100)
    if c == 'c5':
        return Simpls(args)
    if c == 'c4split':
        return SplitHalf(args)
    raise SystemExit('unknown --config ' + c)


def measure(args, wl, env):
    """Time ``args.steps`` steps of workload ``wl`` (after ``args.warmup`` untimed ones) between barriers and
    device syncs, slowest rank counts; returns the record (rank 0) or None.  For PLS-C a second timed region
    runs the same step with the other permutation route, so that BOTH are on record:
    ``value`` = permutations through the feature pass R_p = A_p X (the north-star pipeline),
    ``value_dual`` = the product's default, the S x S dual-space route (no pass over X per permutation)."""
    import torch
    import torch.distributed as dist
    world, rank, dev, backend, collective = env['world'], env['rank'], env['dev'], env['backend'], env['collective']
    n_steps = args.steps + args.warmup
    wl.setup(rank, n_steps, dev)
    eng = wl.eng

    def timed_region():
        for i in range(args.warmup):
            wl.step(i)
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        eng.set_timing(True)
        wl.legs = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.warmup, n_steps):
            wl.step(i, timed=True)
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if dist.is_initialized() and world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    elapsed = timed_region()
    timing = eng.last_timing()
    kt = eng.kernel_timing()
    roof = wl.roofline(kt, args.steps, world) if rank == 0 else None
    legs = [[e[j].elapsed_time(e[j + 1]) for j in range(3)] for e in wl.legs]
    eng.set_timing(False)
    dual = bool(timing.get('dual_perm', 0))

    # second region: permutations through the feature pass (the north-star pipeline)
    elapsed_primal = None
    if isinstance(wl, PLSC) and dual and not args.no_primal:
        eng.set_perm_path(False)
        elapsed_primal = timed_region()
        kt_p = eng.kernel_timing()
        legs_p = [[e[j].elapsed_time(e[j + 1]) for j in range(3)] for e in wl.legs]
        eng.set_timing(False)
        eng.set_perm_path(True)

    emu = None
    if args.emulate_world and isinstance(wl, PLSC) and args.mode == 'strong' and world == 1:
        worlds = sorted({1} | {int(v) for v in args.emulate_world.split(',') if v.strip()})
        emu = {}
        for route, is_dual in (('dual', True), ('feature_pass', False)):
            if is_dual and not dual:
                continue
            eng.set_perm_path(is_dual)
            tab = wl.emulate(worlds)
            base = tab[1]['critical_path_ms']
            for n, row in tab.items():
                row['ideal_ms'] = base / n
                row['efficiency'] = base / (n * row['critical_path_ms'])
                row['over_ideal'] = row['critical_path_ms'] * n / base
            emu[route] = {str(n): row for n, row in tab.items()}
        eng.set_perm_path(True)

    per_rank = None
    if world > 1 and dist.is_initialized() and legs:
        mine = {'rank': rank, 'resample_ms_per_step': float(np.mean([l[0] + l[1] for l in legs])),
                'collective_ms_per_step': float(np.mean([l[2] for l in legs]))}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    if rank != 0:
        return None
    units = wl.units_per_step(world)
    value = units * args.steps / elapsed
    ms_step = 1e3 * elapsed / args.steps
    cfgd = {'workload': '{} (BASELINE config {}) on {} GPU(s)'.format(wl.describe(world), wl.name, world),
            'mode': args.mode, 'collective': '{}, 1 per step, world {}'.format(collective, world),
            'parallelism': 'resample-sharded x{}'.format(world),
            'kernel_ms_per_step': {k: v[0] / args.steps for k, v in kt.items()}}
    if isinstance(wl, (PLSC, Simpls)):
        cfgd.update({'perms_per_step': wl.perms, 'boots_per_step': wl.boots,
                     'per': 'GPU' if args.mode == 'weak' else 'analysis (all GPUs)'})
    if per_rank is not None:
        cfgd['per_rank'] = per_rank                     # device time of each rank's own leg and of ITS wait in the collective
    if legs:
        names = ['perm_ms_per_step', 'boot_ms_per_step', 'collective_ms_per_step'] if (args.mode == 'weak' and isinstance(wl, PLSC)) \
            else ['split_half_ms_per_step', 'unused', 'collective_ms_per_step'] if isinstance(wl, SplitHalf) \
            else ['indexgen_h2d_resample_ms_per_step', 'unused', 'collective_ms_per_step']
        for j, nm in enumerate(names):
            if nm != 'unused':
                cfgd[nm] = float(np.mean([l[j] for l in legs]))
    out = {
        'metric': 'resamples/sec (perm+boot), behavioral_pls X({}x{})/Y({}x{}) fp64'.format(
            wl.S, wl.B, wl.S, wl.T) if wl.name == 'c4' else '{} ({})'.format(wl.unit, wl.name),
        'value': value, 'unit': wl.unit, 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': ms_step, 'higher_is_better': True,
        'scaling': args.mode, 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': cfgd, 'roofline': roof,
    }
    if emu is not None:
        out['strong_scaling_emulation'] = {
            'what': 'one analysis of {} + {} resamples; rank r of an emulated world N on ONE GPU: full index '
                    'generation on the host (as every rank of a real run), own contiguous shard on the device, '
                    'all-gather skipped; critical path = slower of rank 0 and rank N-1; efficiency = '
                    't(1) / (N t(N)).  NOT a hardware scaling curve.'.format(wl.perms, wl.boots),
            'routes': emu}
    if isinstance(wl, PLSC):
        pm, ph = wl.pipeline(ms_step, primal=not dual)
        if elapsed_primal is not None:
            # the dual-space shortcut is reported on its own line -- `value` is the north-star pipeline
            # (every permutation passes over X), `value_dual` the product's default route
            out['value_dual'], out['ms_per_step_dual'] = value, ms_step
            out['value'] = units * args.steps / elapsed_primal
            out['ms_per_step'] = 1e3 * elapsed_primal / args.steps
            cfgd['perm_path'] = 'value: feature pass R_p = A_p X per permutation (north-star pipeline); value_dual: ' \
                                'S x S dual-space route (the product default; no pass over X per permutation)'
            cfgd['perm_ms_per_step_dual'] = cfgd.pop('perm_ms_per_step', None)
            cfgd['perm_ms_per_step'] = float(np.mean([l[0] for l in legs_p]))
            cfgd['kernel_ms_per_step_dual'] = cfgd['kernel_ms_per_step']
            cfgd['kernel_ms_per_step'] = {k: v[0] / args.steps for k, v in kt_p.items()}
            roof['pipeline_frac_mfma_dual'], roof['pipeline_hbm_algorithmic_over_peak_dual'] = pm, ph
            roof['pipeline_work_dual'] = wl.pipeline_formula
            pm, ph = wl.pipeline(out['ms_per_step'], primal=True)
            roof['pipeline_frac_mfma'], roof['pipeline_hbm_algorithmic_over_peak'] = pm, ph
            roof['pipeline_work'] = wl.pipeline_formula
            # the check SURVEY 8d implies for `value`: its algorithmic bytes per second stay below the HBM peak
            out['hbm_algorithmic_TBps'] = out['value'] / world * 8.0 * wl.S * (wl.B + wl.Tp) / 1e12
        else:
            cfgd['perm_path'] = 'dual S x S route only (--no-primal: value is NOT the north-star pipeline)' if dual \
                else 'feature pass'
            roof['pipeline_frac_mfma'], roof['pipeline_hbm_algorithmic_over_peak'] = pm, ph
            roof['pipeline_work'] = wl.pipeline_formula
    roof['traffic'] = None
    tpath = os.path.join(ROOT, 'profiles', 'traffic_{}.json'.format(wl.name))
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            prefix = roof.get('kernel', 'k_xprod').split('<')[0].split(' ')[0]
            kname = tj.get('kernel', '')
            if not kname.startswith(prefix):           # the file's pick is the busiest k_xprod; ours may be another class
                cands = {k: v for k, v in tj.get('kernels', {}).items() if k.startswith(prefix)}
                kname = max(cands, key=lambda k: cands[k]['total_over_run']) if cands else kname
            roof['traffic'] = tj['kernels'][kname]['hbm_bytes_per_launch'] if kname in tj.get('kernels', {}) \
                else tj['hbm_bytes_per_launch']
            roof['traffic_source'] = 'profiles/traffic_{}.json ({})'.format(wl.name, kname)
            # c5: the dual-space solver (k_sd_*) is the largest class by time and is bound by HBM, not by the matrix
            # pipe: its counter traffic per step (2 k launches of k_sd_step, 2 of k_sd_init / k_sd_post0, 1 of
            # k_sd_final; bytes per full-size launch from the same file) over its measured time
            sd_ms = out['config'].get('kernel_ms_per_step', {}).get('k_simpls_dual', 0.0)
            if wl.name == 'c5' and sd_ms > 0 and getattr(wl, 'perms', 0) == 5000 and getattr(wl, 'boots', 0) == 5000:
                def per_launch(prefix):
                    c = [v for k, v in tj.get('kernels', {}).items() if k.startswith(prefix)]
                    return max(v['hbm_bytes_per_launch'] for v in c) if c else 0.0
                kk = int(getattr(wl, 'k', 15))
                tot = 2 * kk * per_launch('k_sd_step') + 2 * per_launch('k_sd_init') + 2 * per_launch('k_sd_post0') \
                    + per_launch('k_sd_final')
                if tot > 0:
                    # ALGORITHMIC bytes of the solver: what must cross HBM per resample given that
                    # nothing S x T-sized fits a wave's share of LDS (13 KB at three waves per SIMD; Z0 alone is 8 S T =
                    # 160 KB) while the shared Y (160 KB for ALL resamples) and K stay in L2:
                    #   k_sd_init   writes the S x (T + 1) operand of GEMM 0, reads the index row       8 S (T + 2)
                    #   k_sd_post0  reads GEMM 0's S x (T + 1), writes Z0 (T-major) and K cnt            8 S (2 T + 2)
                    #   k_sd_step c reads Z0 once, every earlier basis pair (t_j, K beta_j) once, K beta_{c-1}; writes
                    #               t_c, beta_c and the scattered GEMM operand                            8 S (T + 2 c + 4)
                    #   k_sd_final  (bootstraps) reads the k score vectors, writes the k scattered weights 8 S (2 k)
                    # one pass over each array that is needed -- DESIGN.md section 5 ("c5 solver: a denominator")
                    Ss, Tt = wl.S, wl.T
                    steps_b = sum(8.0 * Ss * (Tt + 2 * c + 4) for c in range(kk))
                    per_perm = 8.0 * Ss * (Tt + 2) + 8.0 * Ss * (2 * Tt + 2) + steps_b
                    per_boot = per_perm + 8.0 * Ss * 2 * kk
                    alg = wl.perms * per_perm + wl.boots * per_boot
                    roof['solver_hbm'] = {'kernels': 'k_sd_init / k_sd_post0 / k_sd_step / k_sd_final', 'bound': 'hbm',
                                          'traffic_per_step': tot, 'ms_per_step': sd_ms,
                                          'achieved': tot / (sd_ms * 1e-3) / 1e12, 'peak': PEAK_HBM_TBS, 'unit': 'TB/s',
                                          'frac': alg / (sd_ms * 1e-3) / 1e12 / PEAK_HBM_TBS,
                                          'frac_counter_traffic': tot / (sd_ms * 1e-3) / 1e12 / PEAK_HBM_TBS,
                                          'algorithmic_bytes_per_step': alg,
                                          'algorithmic_bytes_per_resample': {'permutation': per_perm, 'bootstrap': per_boot},
                                          'waste_ratio': tot / alg,
                                          'work': 'per resample: 8 S (T + 2) [init] + 8 S (2 T + 2) [post0] + sum_c 8 S (T + 2 c + 4) '
                                                  '[step c: Z0 once, every earlier basis pair once] (+ 8 S 2 k [final, bootstraps])',
                                          'note': 'frac = algorithmic bytes / solver time / 8 TB/s (a true roofline fraction); '
                                                  'frac_counter_traffic = PMC bytes of full-size launches (5000 resamples) from the '
                                                  'traffic file / time / peak = bandwidth utilisation; waste_ratio = counter / '
                                                  'algorithmic.  Valid for the literal 5000 + 5000 step only'}
        except Exception:
            pass
    roof['measured_mfma_f64_peak_tflops'] = eng.mfma_f64_peak()
    if world == 1 and args.cpu_sample > 0:
        v, sample = wl.cpu_baseline()
        out['cpu_baseline'] = {'value': v, 'unit': wl.unit, 'cores': os.cpu_count() or 1, 'kind': 'port',
                               'sample': sample}
    return out


SUB_CONFIGS = ('c2', 'c3', 'c4split', 'c5')


def sub_records(args, env):
    """The other BASELINE configs and the end-to-end call, measured in the SAME driver run as the headline line
    (VERDICT r4 item 2): compact records {value, unit, ms_per_step, roofline {kernel, frac, ...}, cpu_baseline}
    under ``configs``.  One GPU only (a driver scaling run times the headline step alone)."""
    import copy
    import gc
    import torch
    out = {}
    t_all = time.perf_counter()
    for cfg in SUB_CONFIGS:
        a = copy.copy(args)
        a.config, a.mode, a.perms, a.boots, a.emulate_world = cfg, 'weak', 0, 0, ''
        a.steps, a.warmup = (2, 1) if cfg == 'c4split' else (3, 1)
        a.cpu_sample = 8
        t0 = time.perf_counter()
        try:
            wl = make_workload(a)
            rec = measure(a, wl, env)
            keep = {k: rec[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'warmup', 'dtype',
                                        'value_dual', 'ms_per_step_dual', 'cpu_baseline') if k in rec}
            keep['workload'] = rec['config']['workload']
            keep['kernel_ms_per_step'] = rec['config']['kernel_ms_per_step']
            r = rec['roofline']
            keep['roofline'] = {k: r[k] for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'frac_issued',
                                                  'pipeline_frac_mfma', 'traffic', 'dominant_by_time', 'route', 'solver_hbm')
                                if k in r}
            if cfg == 'c5':
                keep['parity_note'] = 'SIMPLS with T = 20 > 11: pinned on the oracle\'s exact SIMPLS; REFERENCE parity ' \
                                      'unpinned (its rank-1 randomized_svd is seed-dependent at 1e-2, SURVEY 0.3)'
            keep['wall_s'] = time.perf_counter() - t0
            out[cfg] = keep
            del wl.eng
            del wl
        except Exception as exc:                           # the headline line must survive a failing sub-record
            out[cfg] = {'error': '{}: {}'.format(type(exc).__name__, str(exc)[:300])}
        gc.collect()
        torch.cuda.empty_cache()
    # the END-TO-END public call at the headline shape: index generation, H2D, decomposition, resampling, collective,
    # finishing and D2H inside the clock
    t0 = time.perf_counter()
    try:
        a = copy.copy(args)
        a.config, a.mode, a.perms, a.boots, a.emulate_world, a.splits, a.steps = 'c4', 'analysis', 0, 0, '', 0, 2
        rec = run_analysis(a, 1, 0, env['dev'], env['backend'])
        out['c4_analysis'] = {'metric': rec['metric'], 'value': rec['value'], 'unit': rec['unit'],
                              'ms_per_step': rec['ms_per_step'], 'workload': rec['config']['workload'],
                              'fixed_cost_ms': rec['fixed_cost_ms']['everything_else_ms'],
                              'phases_ms': rec['end_to_end_emulation']['worlds']['1']['rank0_phases_ms'],
                              'wall_s': time.perf_counter() - t0}
    except Exception as exc:
        out['c4_analysis'] = {'error': '{}: {}'.format(type(exc).__name__, str(exc)[:300])}
    out['wall_s'] = time.perf_counter() - t_all
    return out


def sharded_records(args, env):
    """Under --gpus N > 1 the headline is the weak step (every rank a full 1008 + 1008 on its replica: near-linear by
    construction).  What north_star's "1e4 resamples/s at 8 GPUs" is about -- ONE analysis sharded over the N ranks --
    rides in the same line, so that the driver's single command per N yields all three curves:
      strong   ONE public call of --strong-resamples permutations + as many bootstraps (default 10 000 + 10 000),
               end to end: seed-compatible index generation, H2D, decomposition, the rank's shards, THE all-gather,
               finishing, D2H; barriers around the call, slowest rank counts; every rank's phase split
      c4split  ONE set of --split-arrangements permuted arrangements x n_split = 100 sharded over the ranks
               (contiguous slices, like the permutations of a call with n_split), gathered once; per-rank leg times.
    Collective: every rank makes the same calls."""
    import copy
    import gc
    import torch
    out = {}
    rank = env['rank']
    t0 = time.perf_counter()
    try:
        a = copy.copy(args)
        a.config, a.mode, a.emulate_world, a.splits, a.steps = 'c4', 'analysis', '', 0, 2
        a.perms = a.boots = args.strong_resamples
        rec = run_analysis(a, env['world'], rank, env['dev'], env['backend'])
        if rank == 0:
            out['strong'] = {'metric': rec['metric'], 'value': rec['value'], 'unit': rec['unit'], 'scaling': 'strong',
                             'ms_per_step': rec['ms_per_step'], 'n_gpus': rec['n_gpus'],
                             'workload': rec['config']['workload'], 'collective': rec['config'].get('collective'),
                             'collective_ms': rec.get('collective_ms'),
                             'per_rank_phases_ms': rec.get('per_rank_phases_ms'),
                             'wall_s': time.perf_counter() - t0}
    except Exception as exc:                               # the headline line must survive a failing sub-record
        out['strong'] = {'error': '{}: {}'.format(type(exc).__name__, str(exc)[:300])}
    gc.collect()
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    try:
        a = copy.copy(args)
        a.config, a.mode, a.emulate_world = 'c4split', 'strong', ''
        a.perms, a.boots, a.steps, a.warmup, a.cpu_sample = args.split_arrangements, 0, 2, 1, 0
        wl = make_workload(a)
        rec = measure(a, wl, env)
        if rank == 0:
            out['c4split'] = {'metric': rec['metric'], 'value': rec['value'], 'unit': rec['unit'], 'scaling': 'strong',
                              'ms_per_step': rec['ms_per_step'], 'n_gpus': rec['n_gpus'],
                              'workload': rec['config']['workload'] + ': ONE set of {} arrangements sharded over the '
                                          'ranks'.format(wl.arr),
                              'split_half_ms_per_step': rec['config'].get('split_half_ms_per_step'),
                              'collective_ms_per_step': rec['config'].get('collective_ms_per_step'),
                              'per_rank': rec['config'].get('per_rank'),
                              'roofline': {k: rec['roofline'][k] for k in ('bound', 'kernel', 'frac', 'route')
                                           if k in rec['roofline']},
                              'wall_s': time.perf_counter() - t0}
        del wl.eng
        del wl
    except Exception as exc:
        out['c4split'] = {'error': '{}: {}'.format(type(exc).__name__, str(exc)[:300])}
    gc.collect()
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', default='c4', choices=['c4', 'c2', 'c3', 'c5', 'c4split'])
    ap.add_argument('--mode', default='weak', choices=['weak', 'strong', 'analysis'])
    ap.add_argument('--splits', type=int, default=0, help='--mode analysis: n_split of the call (0 = no split-half)')
    ap.add_argument('--S', type=int, default=500)
    ap.add_argument('--B', type=int, default=200000)
    ap.add_argument('--T', type=int, default=50)
    ap.add_argument('--perms', type=int, default=0,
                    help='permutations per step (per GPU in weak mode, total in strong mode)')
    ap.add_argument('--boots', type=int, default=0, help='bootstraps per step (same convention)')
    ap.add_argument('--cpu-sample', type=int, default=8,
                    help='permutations and bootstraps (each) timed for the CPU baseline; 0 = skip')
    ap.add_argument('--no-primal', action='store_true', help='skip the second (feature-pass) timed region')
    ap.add_argument('--no-configs', action='store_true',
                    help='default run only: skip the sub-records of the other BASELINE configs (configs: {...})')
    ap.add_argument('--strong-resamples', type=int, default=10000,
                    help='--gpus N > 1, default run: n_perm = n_boot of the ONE sharded end-to-end analysis embedded '
                         'under "sharded" (strong scaling)')
    ap.add_argument('--split-arrangements', type=int, default=64,
                    help='--gpus N > 1, default run: arrangements (x n_split = 100) of the ONE sharded split-half leg '
                         'embedded under "sharded"')
    ap.add_argument('--sharded-timeout', type=float, default=600.0,
                    help='--gpus N > 1, default run: seconds the sharded sub-records may take before the headline line is '
                         'printed without them')
    ap.add_argument('--n-split', type=int, default=0, help='--config c4split: splits per arrangement (default 100)')
    ap.add_argument('--emulate-world', default='',
                    help='strong mode, one GPU: comma-separated world sizes N; times the critical path of one '
                         'analysis on rank 0 and on rank N-1 of an emulated world N (full index generation, own '
                         'shard only, no gather) on both permutation routes and reports the implied efficiency')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))
    # The one JSON line must be the LAST thing on stdout.  RCCL prints a version
    # banner through C stdio (flushed at exit, i.e. after Python's own output), so
    # everything written to fd 1 from here on goes to stderr and the JSON line is
    # written straight to the real stdout at the very end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world and rank == 0:
        sys.stderr.write('bench.py: --gpus {} but the launcher started {} rank(s); using {}\n'
                         .format(args.gpus, world, world))
    if torch.cuda.device_count() < (world if world > 1 and not os.environ.get('PLSX_BENCH_SHARE_GPU') else 1):
        sys.stderr.write('bench.py: {} rank(s) but {} visible GPU(s)\n'.format(world, torch.cuda.device_count()))
        sys.exit(2)
    backend = os.environ.get('PLSX_BENCH_BACKEND', 'nccl')   # 'gloo' only for single-GPU dry runs
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dev = torch.device('cuda', torch.cuda.current_device())
    # The process group exists at N = 1 too, so the collective of the step runs on
    # RCCL in every configuration the driver measures.
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(free_port()))
    collective = backend
    try:
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    except Exception as exc:                               # pragma: no cover
        if world > 1:
            raise
        collective = 'none (process group init failed: {})'.format(str(exc)[:120])
    world = dist.get_world_size() if dist.is_initialized() else 1
    if dist.is_initialized():
        from pypyls_amd import parallel
        # EXPLICIT collective open of the communicator behind plsx_allgather (every rank).  With more than one rank only on
        # request (PLSX_BENCH_NATIVE_COLLECTIVE=1): that open -- a second RCCL communicator next to the process group's
        # -- has never run with N > 1 on hardware, and the first multi-GPU lease should yield a scaling curve rather than
        # test it; the process group's own all_gather_into_tensor is RCCL all the same.  `config.collective` says which ran.
        if world == 1 or os.environ.get('PLSX_BENCH_NATIVE_COLLECTIVE') == '1':
            parallel.open_native_comm()
        collective = parallel.collective_name()           # a pure query: which all-gather the steps will issue
    env = {'world': world, 'rank': rank, 'dev': dev, 'backend': backend, 'collective': collective}

    if args.mode == 'analysis':
        out = run_analysis(args, world, rank, dev, backend)
    else:
        wl = make_workload(args)
        out = measure(args, wl, env)
        if (out is not None and world == 1 and args.config == 'c4' and args.mode == 'weak' and not args.no_configs
                and (args.S, args.B, args.T) == (500, 200000, 50)):
            del wl.eng
            del wl
            torch.cuda.empty_cache()
            out['configs'] = sub_records(args, env)
        elif world > 1 and args.config == 'c4' and args.mode == 'weak' and not args.no_configs:
            del wl                                          # (its engine and scratch go with it)
            torch.cuda.empty_cache()
            # The sub-records are collective calls that have never run on several real GPUs: if they hang (a rank that
            # raised while its peers wait in a collective), the weak headline measured above must still reach the driver.
            # A watchdog on every rank prints it (rank 0) and leaves the process after `--sharded-timeout` seconds.
            def bail():
                if rank == 0 and out is not None:
                    out['sharded'] = {'error': 'sub-records did not finish within {} s; headline only'.format(args.sharded_timeout)}
                    os.write(real_stdout, (json.dumps(out) + '\n').encode())
                os._exit(0)
            dog = threading.Timer(args.sharded_timeout, bail)
            dog.daemon = True
            dog.start()
            sharded = sharded_records(args, env)            # collective: every rank
            dog.cancel()
            if rank == 0 and out is not None:
                out['sharded'] = sharded
    if rank == 0 and out is not None:
        os.write(real_stdout, (json.dumps(out) + '\n').encode())
    if dist.is_initialized():
        dist.barrier()
        from pypyls_amd import parallel
        parallel.release_native_comm()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
