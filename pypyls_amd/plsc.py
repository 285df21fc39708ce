"""
Front-ends ``behavioral_pls`` / ``meancentered_pls`` with the keyword surface
and ``PLSResults`` layout of the reference (pyls/types/behavioral.py:231-242,
pyls/types/meancentered.py:182-192), driving the MI355X engine.

Host code here mirrors the ORCHESTRATION of BasePLS.run_pls
(pyls/base.py:341-399) and the subclass ``run_pls`` bodies; every
per-resample computation (gather/permute, z-score, cross-product, SVD,
Procrustes, accumulation) runs on the device through the C ABI of
include/plsx.h.  There is no CPU fallback.

Differences from the reference at this commit, on purpose:
  * ``permindices`` defaults to True, the documented default
    (pyls/structures.py:115-120); the shipped ``None`` makes
    BasePLS._single_perm treat index vectors as data.
  * ``n_proc`` counts GPUs, not CPU workers: the joblib pool of the reference
    (pyls/utils.py:252-279) is what the devices replace.  ``n_proc=8`` (or
    'max' / -1) on an 8-GPU node shards the resamples of THIS call over the
    GPUs from this one process -- one context and one host thread per device,
    ONE all-gather (pypyls_amd/team.py); ``device_ids=[...]`` names them
    explicitly.  On a one-GPU box both reduce to the ordinary call.
  * When torch.distributed is initialised (a script under torchrun: one
    process per GPU) the resamples are sharded across the RANKS instead and
    collected with one all-gather (pypyls_amd/parallel.py); ``n_proc`` /
    ``device_ids`` are then ignored.
"""
import warnings

import numpy as np

from . import hostmath, parallel, resampling
from .structures import PLSInputs, PLSResults

_METHOD_CODE = {'behavioral': 0, 'meancentered': 1}

# The B- and n_boot-sized result arrays (x_weights, bootstrap ratios / standard errors, the (T', L, n_boot)
# distributions: 440 MB at the headline shape) LAND in page-locked memory (57 GB/s instead of 8 - 10 pageable) that is
# mapped while the device resamples.  PLSResults holds ordinary numpy arrays, like the reference's: they are copied out
# of the page-locked landing zone by a few host threads (a memcpy releases the GIL; 440 MB in ~10 ms, one thread needs
# 55) and the landing zone is released with the call -- a caller that keeps many results (sweeps, per-subject loops)
# does not accumulate unswappable memory.  ``COPY_RESULTS_OUT_OF_PINNED = False`` hands out views of the
# page-locked memory instead (no second pass; that memory then stays locked for as long as the results live).
COPY_RESULTS_OUT_OF_PINNED = True
_COPY_THREADS = 8
_COPY_POOL = None


def _host_array(t):
    a = t.numpy()
    if not COPY_RESULTS_OUT_OF_PINNED:
        return a
    out = np.empty_like(a)
    n = a.size
    if n * a.itemsize < (8 << 20) or not a.flags.c_contiguous:
        np.copyto(out, a)
        return out
    global _COPY_POOL
    if _COPY_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _COPY_POOL = ThreadPoolExecutor(max_workers=_COPY_THREADS, thread_name_prefix='plsx-copy')
    src, dst = a.reshape(-1), out.reshape(-1)
    step = -(-n // _COPY_THREADS)
    list(_COPY_POOL.map(lambda lo: np.copyto(dst[lo:lo + step], src[lo:lo + step]), range(0, n, step)))
    return out


def _as_float_array(A, name):
    A = np.asarray(A)
    if A.ndim != 2:
        raise ValueError('Expected 2D array for `{}`, got {}D array instead'.format(name, A.ndim))
    return A.astype(np.float64, copy=False)


def _check_finite(A, name):
    if not np.all(np.isfinite(A)):
        raise ValueError('Input `{}` contains NaN, infinity or a value too large'.format(name))


class _PLSCRun(object):
    """One analysis; the counterpart of BasePLS + subclass (pyls/base.py:232)."""

    def __init__(self, method, X, Y=None, groups=None, n_cond=1, **kwargs):
        self.method = method
        if groups is None:
            groups = [len(X) // n_cond]
        elif not isinstance(groups, (list, np.ndarray)):
            groups = [groups]
        groups = [int(g) for g in groups]
        # pyls/base.py:265-277
        n_samples = sum(g * n_cond for g in groups)
        if len(X) != n_samples:
            raise ValueError('Number of samples specified by `groups` and `n_cond` does not '
                             'match number of samples in input array(s).\n'
                             '    EXPECTED: {}\n    ACTUAL:   {} (groups: {} * n_cond: {})'
                             .format(len(X), n_samples, groups, n_cond))
        if Y is not None and len(X) != len(Y):
            raise ValueError('Provided `X` and `Y` matrices must have the same number of '
                             'samples. Provided matrices differed: X: {}, Y: {}'
                             .format(len(X), len(Y)))
        kwargs.setdefault('permindices', True)
        # measurement hooks of bench.py --mode analysis (not inputs of the analysis): a dict that receives
        # per-phase wall times; (rank, world) of a world emulated on one GPU
        self.phases = kwargs.pop('_phases', None)
        self.emulate = kwargs.pop('_emulate', None)
        self.device_ids = kwargs.pop('device_ids', None)
        self.transport = kwargs.pop('_transport', 'auto')
        self.inputs = PLSInputs(X=X, Y=Y, groups=groups, n_cond=n_cond, **kwargs)
        # under torch.distributed every rank must draw the same index arrays
        self.rs = resampling.check_random_state(parallel.shared_seed(self.inputs.get('seed')))
        self.cells = resampling.cell_of_row(groups, n_cond)
        self.n_cells = len(groups) * n_cond
        self.engine = kwargs.get('_engine')

    # ------------------------------------------------------------------
    def _plan_draws(self, L):
        """Everything the analysis draws from ``self.rs``, in the reference's order
        (BasePLS.run_pls, base.py:362-397; crossval after the bootstrap,
        behavioral.py:219-221), as jobs of ONE host thread that starts before the
        data goes to the device: the SVD's seed matrix, the permutation arrays, the
        split masks of the original data, the bootstrap arrays, the cross-validation
        splits.  Only the ORDER of consumption is pinned by a seed; the device takes
        finished rows while later ones are still drawn (resampling.IndexStream)."""
        inp = self.inputs
        S = self.cells.size
        # the reference's randomized_svd consumes normal((L, L + 10)) from self.rs for the
        # original decomposition; draw it so that later index arrays match for a seed
        jobs = [lambda rs: rs.normal(size=(L, L + 10))]
        self.perm_stream = self.boot_stream = self.ystack = None
        self.orig_splits = self.cv_splits = None
        n_perm = inp.get('n_perm') or 0
        n_boot = inp.get('n_boot') or 0
        from .engine import check_index_array
        if n_perm > 0:
            # BasePLS.permutation, base.py:601-652
            permsamp = inp.get('permsamples')
            if permsamp is None:
                self.perm_stream = resampling.IndexStream('perm', inp.groups, inp.n_cond, n_perm)
                jobs.append(self.perm_stream.draw)
            elif not inp.get('permindices'):
                # pre-permuted Y matrices, (n_perm, S, T) (base.py:636-639)
                permsamp = np.asarray(permsamp)
                if self.method != 'behavioral' or permsamp.ndim != 3:
                    raise ValueError('permindices=False expects `permsamples` of shape '
                                     '(n_perm, S, T) and behavioral PLS')
                self.ystack = permsamp.astype(np.float64, copy=False)
            else:
                permsamp = np.asarray(permsamp)
                if permsamp.ndim != 2 or permsamp.shape[0] != S:
                    raise ValueError('resampling array must have shape (S, n) with S = {}; got {}'
                                     .format(S, permsamp.shape))
                self.perm_stream = resampling.IndexStream.of_array(check_index_array(permsamp, S))
                self.perm_given = permsamp
        n_split = inp.get('n_split')
        if (self.perm_stream is not None or self.ystack is not None) and n_split is not None:
            # the reference draws the split masks of the ORIGINAL data from self.rs after the
            # permutation arrays and before the bootstrap arrays (base.py:373-380); permutation
            # i uses a fresh RandomState(i) (base.py:705-708, 738-742)
            self.orig_splits = inp.get('_splitsamples')
            if self.orig_splits is None:
                jobs.append(lambda rs: setattr(self, 'orig_splits', resampling.gen_splits(
                    inp.groups, inp.n_cond, n_split, seed=rs, test_size=0.5)))
        if n_boot > 0:
            # BasePLS.bootstrap, base.py:439-528 (index arrays drawn AFTER the permutation
            # arrays, as in the reference's run_pls order)
            bootsamp = inp.get('bootsamples')
            if bootsamp is None:
                self.boot_stream = resampling.IndexStream('boot', inp.groups, inp.n_cond, n_boot)
                jobs.append(self.boot_stream.draw)
            else:
                bootsamp = np.asarray(bootsamp)
                if bootsamp.ndim != 2 or bootsamp.shape[0] != S:
                    raise ValueError('resampling array must have shape (S, n) with S = {}; got {}'
                                     .format(S, bootsamp.shape))
                self.boot_stream = resampling.IndexStream.of_array(check_index_array(bootsamp, S))
                self.boot_given = bootsamp
        if (self.method == 'behavioral' and inp.get('test_split') is not None
                and (inp.get('test_size') or 0) > 0):
            # behavioral.py:219-221, 82-170
            self.cv_splits = inp.get('_cvsplits')
            if self.cv_splits is None:
                jobs.append(lambda rs: setattr(self, 'cv_splits', resampling.gen_splits(
                    inp.groups, inp.n_cond, inp.test_split, seed=rs, test_size=inp.test_size)))
        return resampling.DrawThread(self.rs, jobs)

    def run(self):
        import torch
        from .engine import Engine
        inp = self.inputs
        X = _as_float_array(inp.X, 'X')
        Y = None if self.method == 'meancentered' else _as_float_array(inp.Y, 'Y')
        Tp = self.n_cells * Y.shape[1] if Y is not None else self.n_cells
        self.perm_given = self.boot_given = None
        from .engine import default_engine, touch_idle_release
        from . import team as _team
        touch_idle_release()                           # (a pending idle release is pushed back before the engine is looked up)
        devices = None
        if self.engine is None and self.emulate is None and parallel._dist() is None:
            # n_proc workers of the reference (pyls/utils.py:252-279) = GPUs of this node, driven from this process
            devices = _team.resolve_devices(inp.get('n_proc'), self.device_ids)
        self._mstreams = []
        self.team = None
        if devices is not None and len(devices) > 1:
            self.team = _team.team_for(devices, self.transport)
        elif devices is not None:
            self.engine = default_engine(devices[0])
        draws = self._plan_draws(min(Tp, X.shape[1])).start()
        try:
            unrefined = 0
            if self.team is not None:
                # one host thread per device; the draws above are shared (drawn ONCE for all ranks)
                res = self.team.run(lambda rank, world, eng: self._run_device(
                    X, Y, draws, eng, team=(rank, self.team)))
                unrefined = self.team.unrefined
            else:
                eng = self.engine or default_engine()
                ok = False
                with eng.lock:                         # one analysis at a time per context (shared default engine)
                    try:
                        res = self._run_device(X, Y, draws, eng)
                        ok = True
                    finally:
                        # state that must not leak into the next analysis on this context, whatever happened:
                        # the announced shard size and the refined / unrefined counters (a stale count would raise
                        # a spurious GradedSpectrumWarning -- and stale scratch sizing -- in an unrelated call)
                        if getattr(eng, 'ctx', None):
                            unrefined = eng.end_analysis(warn=ok) or 0
            # outside every finally: under -W error the warning must not replace a computed result half-way, and it
            # is attributed to the user's call (run <- behavioral_pls / meancentered_pls <- USER)
            Engine.warn_unrefined(unrefined, stacklevel=4)
            return res
        finally:
            draws.thread.join()                        # never leave the generators running on an error
            from .engine import touch_idle_release
            touch_idle_release()                       # (the cached engines are released after IDLE_RELEASE_S idle seconds)
            for ms in self._mstreams:
                ms.close()

    def _run_device(self, X, Y, draws, eng, team=None):
        """The analysis after the draws started (``team`` = (rank, Team): this thread is one rank of a
        single-process team, team.py -- every rank runs its shard and the collective, rank 0 alone finishes).  Everything B- or n_boot-sized stays on the device from the
        H2D copy of X to the finished statistics: the sign convention, the original's scores, the resampling,
        THE one collective, percentile intervals, bootstrap ratios and the (T', L, n_boot) layout of the
        distributions all run there; what PLSResults holds comes back once, into page-locked memory that is
        mapped while the device resamples.  (Round 3 crossed PCIe with the (B, L) weights five times and
        transposed the 200 MB distributions on the host: 0.26 - 0.30 s of fixed cost per call at c4.)"""
        import time
        import torch
        inp = self.inputs
        lead = team is None or team[0] == 0            # the rank that finishes the analysis and speaks for it
        phases = self.phases if lead else None         # dict: per-phase wall times (bench.py --mode analysis)

        t_last = [time.perf_counter()]

        def tick(name):
            if phases is None:
                return
            torch.cuda.synchronize(eng.device)
            now = time.perf_counter()
            phases[name] = phases.get(name, 0.0) + 1e3 * (now - t_last[0])
            t_last[0] = now

        eng.set_data(X, Y, self.cells, len(inp.groups), inp.n_cond, _METHOD_CODE[self.method],
                     mean_centering=inp.get('mean_centering') or 0,
                     covariance=bool(inp.get('covariance')))
        tick('h2d_and_bind')
        L = eng.L
        res = PLSResults(inputs=inp)
        # finite-input check (sklearn check_X_y in compute.xcorr, compute.py:78):
        # a NaN / inf anywhere in a column of X makes the device column mean
        # non-finite, which avoids a host pass over the (S, B) matrix
        d_xmean = eng.colmean_dev()
        if not bool(torch.isfinite(d_xmean).all().item()):
            raise ValueError('Input `X` contains NaN, infinity or a value too large')
        if Y is not None:
            _check_finite(Y, 'Y')

        # ---- original decomposition (BasePLS.svd, base.py:362-364), signs and scores on the device -------
        d_xw, d_sv, d_yw = eng.decompose_dev()
        eng.svd_flip(d_xw, d_yw)
        eng.set_original(d_xw, d_sv, d_yw)
        d_scores = eng.project_dev(d_xw)               # (X - mean) @ x_weights; the mean's share is added on the host
        sv = d_sv.cpu().numpy()
        yw = d_yw.cpu().numpy()
        tick('decompose')
        emulate = self.emulate                         # (rank, world) of an emulated run on one GPU (bench.py)
        if team is not None:
            rank, world = team[0], team[1].world
        else:
            rank, world = emulate if emulate is not None else parallel.rank_world()
        rotate = bool(inp.get('rotate', True))

        # ---- resampling: this rank's contiguous shard of the permutations and chunk-cyclic share
        # ---- of the bootstraps, launched chunk by chunk as the index rows arrive; results stay
        # ---- on the device through THE one collective (parallel.collect_device) ---------
        pstream, bstream, ystack = self.perm_stream, self.boot_stream, self.ystack
        n_perm_tot = pstream.n if pstream is not None else (ystack.shape[0] if ystack is not None else 0)
        n_boot_tot = bstream.n if bstream is not None else 0
        d_perm = d_dist = usum = usq = None
        plo, phi = parallel.shard_bounds(n_perm_tot, rank, world)
        # Chunked shipping (IndexStream.chunks: 256 rows, then 4 x longer ranges) lets the device start while the
        # generator thread still draws.  It pays at both ends of the size range: c4's shard is device-seconds long
        # against 25 ms of drawing, and c2's 5000 + 5000 index rows take the host about as long to DRAW (4 - 9 ms)
        # as the device needs to process them -- shipping them as one range when the draw ends (tried: est. device
        # time < 50 ms -> one range) made the call slower, 9.9 -> 11.4 ms.  Caller-supplied arrays are one range.
        first = 256
        n_split = inp.get('n_split')
        mstream = None
        if n_split is not None and (pstream is not None or ystack is not None) and phi > plo:
            # split masks of this rank's permutations: produced block by block on their own host thread, from
            # now on, while the device runs the permutations and the bootstraps (permutation i uses
            # RandomState(i), base.py:705-708)
            mstream = resampling.MaskStream(inp.groups, inp.n_cond, n_split, plo, phi,
                                            given=inp.get('_perm_splitsamples'))
            self._mstreams.append(mstream)
        # the shard arrives in chunks: size the super-batch scratch once, for all of it
        eng.set_option('expect_resamples', max(phi - plo, sum(hi - lo for lo, hi in parallel.shard_chunks(
            n_boot_tot, rank, world)) if bstream is not None else 0))
        # verbose=True: the reference's tqdm bars (pyls/utils.py:128-152), counting resamples the DEVICE has finished;
        # they only appear when a leg is still running after 2 s (progress.py)
        from .progress import Bar
        show = lead and bool(inp.get('verbose')) and (emulate is None) and (parallel.rank_world()[0] == 0)
        bars = []
        if pstream is not None:
            d_perm = eng._zeros((phi - plo, L))
            bar = Bar('Running permutations', phi - plo, show, eng.device)
            bars.append(bar)
            for a, b in pstream.chunks(plo, phi, first=first):
                eng.perm_into(eng.rows_tensor(pstream.rows[a:b]), d_perm[a - plo:b - plo], rotate=rotate)
                bar.queued(b - a)
        elif ystack is not None:
            host = eng.perm_ystack(ystack[plo:phi], rotate=rotate) if phi > plo else np.zeros((L, 0))
            d_perm = torch.from_numpy(np.ascontiguousarray(host.T)).to(eng.device)
        tick('permutations')
        if bstream is not None:
            # chunk-cyclic share (parallel.shard_chunks): no rank waits for the END of the draw
            # before its device has anything to do
            bchunks = parallel.shard_chunks(n_boot_tot, rank, world)
            usum, usq = eng._zeros((eng.B, L)), eng._zeros((eng.B, L))
            d_dist = eng._zeros((sum(hi - lo for lo, hi in bchunks), eng.Tp, L))
            off = 0
            # the series of this rank's bootstraps: where the weights are linear in the (unscaled) features the
            # library accumulates S x S moments per batch and passes the features once, in boot_finish
            eng.boot_begin(sum(hi - lo for lo, hi in bchunks))
            bar = Bar('Running bootstraps', sum(hi - lo for lo, hi in bchunks), show, eng.device)
            bars.append(bar)
            for blo, bhi in bchunks:
                for a, b in bstream.chunks(blo, bhi, first=first):
                    eng.boot_into(eng.rows_tensor(bstream.rows[a:b]), usum, usq,
                                  d_dist[off + a - blo:off + b - blo])
                    for done in bars:
                        done.poll()
                    bar.queued(b - a)
                off += bhi - blo
            eng.boot_finish(usum, usq)
        # host work that needs no device result runs while the device is busy: page-locked landing zones of
        # the results (56 us per MB to map), the index arrays in the reference's layout and dtype ((S, n)
        # C-contiguous int64), the host-sized scores of the original data
        pin = {}
        permsamp = bootsamp = None
        if lead:
            pin = {'xw': eng.pinned_like(d_xw), 'scores': eng.pinned_like(d_scores)}
            if bstream is not None:
                pin['bsr'], pin['se'] = eng.pinned_like(d_xw), eng.pinned_like(d_xw)
                pin['dist'] = torch.empty((eng.Tp * L, n_boot_tot), dtype=torch.float64, pin_memory=True)
            if pstream is not None:
                permsamp = self.perm_given if self.perm_given is not None else pstream.samples
            if bstream is not None:
                bootsamp = self.boot_given if self.boot_given is not None else bstream.samples
        draws.join()
        for st in (pstream, bstream):
            if st is not None and lead:
                st.warn()
        for bar in bars:
            bar.watch()
        try:
            eng.sync()             # numerical status of the launches above is raised HERE, for the batch that set it
        finally:
            for bar in bars:
                bar.close()
        tick('bootstraps')
        orig_splits = self.orig_splits
        slices, totals = [], []
        if d_perm is not None:
            if orig_splits is not None:
                # split-half reliability of every permuted arrangement (base.py:705-708): block by block as the
                # masks arrive, the mean over splits (base.py:770) taken on the device; nothing visits the host
                d_uc, d_vc = eng._zeros((phi - plo, L)), eng._zeros((phi - plo, L))
                if mstream is not None:
                    sbar = Bar('Running split-half resampling of the permutations', phi - plo, show, eng.device)
                    try:
                        for a, b, masks in mstream:
                            dm = torch.from_numpy(masks).to(eng.device)
                            dp = eng.rows_tensor(pstream.rows[a:b]) if ystack is None else None
                            dy = eng._dev(ystack[a:b], np.float64) if ystack is not None else None
                            uc = eng._empty((b - a, masks.shape[1], L))
                            vc = eng._empty((b - a, masks.shape[1], L))
                            eng.split_half_into(dp, dm, uc, vc, ystack_dev=dy)
                            eng.mean_splits_into(uc, d_uc[a - plo:b - plo])
                            eng.mean_splits_into(vc, d_vc[a - plo:b - plo])
                            sbar.queued(b - a)
                        sbar.watch()
                        eng.sync()
                    finally:
                        sbar.close()
                        mstream.close()
                    if mstream.duplicates and lead:
                        warnings.warn('WARNING: Duplicate split halves used.')
                # ride along with the permutation block of the single collective
                d_perm = torch.cat([d_perm, d_uc, d_vc], dim=1)
                eng.sync()
                tick('split_half')
            slices.append(d_perm)
            totals.append(n_perm_tot)
        cyclic = []
        if d_dist is not None:
            cyclic.append(len(slices))
            slices.append(d_dist)
            totals.append(n_boot_tot)
        cv_splits = self.cv_splits
        if cv_splits is not None:
            # cross-validation (behavioral.py:219-221, 82-170): its shard rides in the same buffer
            clo, chi = parallel.shard_bounds(cv_splits.shape[1], rank, world)
            if chi > clo:
                r, r2 = eng.crossval(cv_splits[:, clo:chi])
                local_cv = np.vstack([r, r2]).T                                       # (m_loc, 2 T)
            else:
                local_cv = np.zeros((0, 2 * Y.shape[1]))
            slices.append(torch.from_numpy(np.ascontiguousarray(local_cv)).to(eng.device))
            totals.append(cv_splits.shape[1])
            tick('crossval')
        sums = [usum, usq] if usum is not None else []
        full, summed = parallel.collect_device(slices, totals, sums, cyclic=cyclic, emulate=emulate, team=team)
        if not lead:
            return None                                # rank 0 holds everything the ranks computed: it finishes
        if usum is not None:
            usum, usq = summed
        tick('collective')

        # ---- finish on the device; one trip home -----------------------------------------------------
        k = 0
        d_perm = distrib = None
        lo_hi = None
        if n_perm_tot > 0:
            blk = full[k].cpu().numpy().T                                             # (L or 3 L, n_perm)
            k += 1
            d_perm = np.ascontiguousarray(blk[:L])
            if orig_splits is not None:
                ucorrs, vcorrs = np.ascontiguousarray(blk[L:2 * L]), np.ascontiguousarray(blk[2 * L:])
        h_bsr = h_se = h_dist = None
        if bstream is not None:
            d_full = full[k]                                                           # (n_boot, T', L)
            k += 1
            d_series = eng.transpose_dev(d_full.reshape(n_boot_tot, eng.Tp * L))       # (T' L, n_boot) series
            lo_hi = eng.percentile_ci_dev(d_series, ci=inp.get('ci', 95))
            h_dist = eng.to_host_async(d_series, pin['dist'])
            d_bs = eng.scale_columns(d_xw, d_sv)                                       # x_weights @ singvals
            if self.method == 'behavioral':
                # add the original back, n_boot + 1 (behavioral.py:201-207)
                d_bsr, d_se = eng.boot_rel_dev(d_bs, usum, usq, n_boot_tot + 1, add_orig=True)
            else:
                # no add-back, n_boot (meancentered.py:162-164)
                d_bsr, d_se = eng.boot_rel_dev(d_bs, usum, usq, n_boot_tot, add_orig=False)
            h_bsr, h_se = eng.to_host_async(d_bsr, pin['bsr']), eng.to_host_async(d_se, pin['se'])
        h_xw = eng.to_host_async(d_xw, pin['xw'])
        h_scores = eng.to_host_async(d_scores, pin['scores'])
        xmean = d_xmean.cpu().numpy()
        if cv_splits is not None:
            Tn = Y.shape[1]
            cv = full[k].cpu().numpy().T
            res['cvres'].update(dict(pearson_r=np.ascontiguousarray(cv[:Tn]),
                                     r_squared=np.ascontiguousarray(cv[Tn:])))
        eng.sync()
        tick('finish_and_d2h')
        xw = _host_array(h_xw)
        res['x_weights'], res['y_weights'] = xw, yw
        res['x_scores'] = h_scores.numpy() + (xmean @ xw)[None, :]
        if orig_splits is not None and d_perm is not None:
            uc, vc = eng.split_half(orig_splits)
            orig_uc, orig_vc = uc[0].mean(axis=-1), vc[0].mean(axis=-1)
            ci = inp.get('ci', 95)
            ull, uul = hostmath.boot_ci(ucorrs, ci=ci)
            vll, vul = hostmath.boot_ci(vcorrs, ci=ci)
            res['splitres'].update(dict(
                ucorr=orig_uc, vcorr=orig_vc,
                ucorr_pvals=hostmath.perm_sig(orig_uc, ucorrs),
                vcorr_pvals=hostmath.perm_sig(orig_vc, vcorrs),
                ucorr_lolim=ull, vcorr_lolim=vll, ucorr_uplim=uul, vcorr_uplim=vul))
            self.split_null = (ucorrs, vcorrs)
        if d_perm is not None:
            res['permres']['pvals'] = hostmath.perm_sig(sv, d_perm)
            res['permres']['permsamples'] = permsamp if ystack is None else \
                np.transpose(ystack, (1, 2, 0))                   # base.py:638-639 layout
            res['permres']['perm_singval'] = d_perm

        # ---- scores / loadings (subclass run_pls) --------------------------
        if self.method == 'behavioral':
            y_scores = np.zeros((len(X), L))
            T = Y.shape[1]
            for c in range(self.n_cells):
                m = self.cells == c
                y_scores[m] = Y[m] @ yw[c * T:(c + 1) * T]
            res['y_scores'] = y_scores
            res['y_loadings'] = hostmath.cellwise_xcorr(res['x_scores'], Y, self.cells,
                                                        self.n_cells, bool(inp.get('covariance')))
        else:
            dummy = resampling.dummy_code(inp.groups, inp.n_cond)
            inp['Y'] = dummy
            res['y_scores'] = dummy @ yw
            # contrast = cell means of the mean-centred brain scores
            # (meancentered.py:151-155) = gen_covcorr(X) @ x_weights
            contrast = eng.crosscov(n=1)[0] @ xw

        # ---- bootstrap ratios / intervals ----------------------------------
        if bootsamp is not None:
            distrib = _host_array(h_dist).reshape(eng.Tp, L, n_boot_tot)                   # (T', L, n_boot)
            if lo_hi is not None:
                ci_arr = np.stack([lo_hi[0].cpu().numpy().reshape(eng.Tp, L),
                                   lo_hi[1].cpu().numpy().reshape(eng.Tp, L)], -1)
            else:                                       # more than 16384 bootstraps: numpy on the host copy
                ci_arr = np.stack(hostmath.boot_ci(distrib, ci=inp.get('ci', 95)), -1)
            bsr, se = _host_array(h_bsr), _host_array(h_se)
            if self.method == 'behavioral':
                res['bootres'].update(dict(
                    x_weights_normed=bsr, x_weights_stderr=se,
                    y_loadings=res['y_loadings'].copy(), y_loadings_boot=distrib,
                    y_loadings_ci=ci_arr, bootsamples=bootsamp))
            else:
                res['bootres'].update(dict(
                    x_weights_normed=bsr, x_weights_stderr=se, bootsamples=bootsamp,
                    contrast=contrast, contrast_boot=distrib, contrast_ci=ci_arr))

        res['varexp'] = hostmath.varexp(sv)
        res['singvals'] = sv
        tick('host_finish')
        self.engine_used = eng
        return res


def behavioral_pls(X, Y, *, groups=None, n_cond=1, n_perm=5000, n_boot=5000, n_split=0,
                   test_size=0.25, test_split=100, covariance=False, rotate=True, ci=95,
                   permsamples=None, bootsamples=None, seed=None, verbose=True, n_proc=None,
                   **kwargs):
    """Behavioral PLS of X (S, B) against Y (S, T); see pyls.behavioral_pls.  ``n_proc``: GPUs of this node to
    shard the resamples over (module docstring); ``device_ids=[...]`` names them."""
    run = _PLSCRun('behavioral', np.asarray(X), np.asarray(Y), groups=groups, n_cond=n_cond,
                   n_perm=n_perm, n_boot=n_boot, n_split=n_split, test_size=test_size,
                   test_split=test_split, covariance=covariance, rotate=rotate, ci=ci,
                   permsamples=permsamples, bootsamples=bootsamples, seed=seed, verbose=verbose,
                   n_proc=n_proc, **kwargs)
    return run.run()


def meancentered_pls(X, *, groups=None, n_cond=1, mean_centering=0, n_perm=5000, n_boot=5000,
                     n_split=0, rotate=True, ci=95, permsamples=None, bootsamples=None,
                     seed=None, verbose=True, n_proc=None, **kwargs):
    """Mean-centred PLS of X (S, B) sorted into groups x conditions; see
    pyls.meancentered_pls (argument checks of pyls/types/meancentered.py:16-38)."""
    X = np.asarray(X)
    if groups is None:
        if len(X) // n_cond != len(X) / n_cond:
            raise ValueError('Provided `X` matrix with {} samples is not evenly divisible into '
                             '{} conditions. Please confirm inputs are correct and try again. '
                             .format(len(X), n_cond))
        groups = [len(X) // n_cond]
    elif not isinstance(groups, (list, np.ndarray)):
        groups = [groups]
    if n_cond == 1 and len(groups) == 1:
        raise ValueError('Cannot perform PLS with only one group and one condition. Please '
                         'confirm inputs are correct.')
    if n_cond == 1 and mean_centering == 0:
        warnings.warn('Cannot set mean_centering to 0 when there is only one condition. '
                      'Resetting mean_centering to 1.')
        mean_centering = 1
    elif len(groups) == 1 and mean_centering == 1:
        warnings.warn('Cannot set mean_centering to 1 when there is only one group. '
                      'Resetting mean_centering to 0.')
        mean_centering = 0
    if mean_centering not in (0, 1, 2):
        raise ValueError('Mean centering type must be in [0, 1, 2].')
    run = _PLSCRun('meancentered', X, None, groups=groups, n_cond=n_cond,
                   mean_centering=mean_centering, n_perm=n_perm, n_boot=n_boot, n_split=n_split,
                   rotate=rotate, ci=ci, permsamples=permsamples, bootsamples=bootsamples,
                   seed=seed, verbose=verbose, n_proc=n_proc, **kwargs)
    return run.run()
