"""
``pls_regression`` front-end (SIMPLS) with the reference's keyword surface
(pyls/types/regression.py:432-440) on the MI355X engine.

The per-resample SIMPLS fits run on the device in the S-dimensional dual
space (csrc/plsx_simpls.h); host code mirrors PLSRegression.run_pls
(regression.py:375-428).

Differences from the reference at this commit, on purpose:
  * ``n_perm > 0`` works.  In the reference ``PLSRegression._single_perm`` is
    called with keywords it does not accept (regression.py:329 vs
    base.py:646-648) and raises TypeError; the null statistic implemented here
    is what that method computes when driven directly (variance of the
    permuted Y explained per component, ``original=None`` branch, :369).
  * the caller's X is not centred in place (regression.py:395 mutates it).
  * the leading singular triplet per component is exact, the reference's
    rank-1 randomized SVD is approximate when Y has more than 11 columns
    (SURVEY.md section 0.3).
  * 3-D Y: the reference builds its (2, n_boot) resampling array with
    ``np.array(list(zip(s.T, c.T))).T`` (regression.py:216), which numpy >= 1.24
    rejects; here the equivalent object array is built explicitly.  The
    Rows (or, for 3-D Y, subjects) that are only partly NaN raise
    NotImplementedError (they poison the reference's fit as well).
"""
import numpy as np

from . import hostmath, parallel, resampling
from .plsc import _host_array
from .structures import PLSInputs, PLSResults


def resid_yscores(x_scores, y_scores):
    """Residualise column c of y_scores against x_scores columns < c, two
    rounds of modified Gram-Schmidt (regression.py:9-45)."""
    x_scores = np.array(x_scores, dtype=float)
    y_scores = np.array(y_scores, dtype=float)
    for comp in range(x_scores.shape[1]):
        ui = y_scores[:, comp].copy()
        for _ in range(2):
            for j in range(comp):
                tj = x_scores[:, j]
                ui -= (tj @ ui) * tj
        y_scores[:, comp] = ui
    return y_scores


_AGGFUNCS = dict(mean=np.mean, median=np.median, sum=np.sum)


def _row_ok(A):
    """False for all-NaN rows (get_mask, regression.py:48-53); rows that are
    only partly NaN cannot be handled (they poison the reference as well)."""
    nan = np.isnan(A)
    allnan, anynan = nan.all(axis=1), nan.any(axis=1)
    if np.any(anynan & ~allnan):
        raise NotImplementedError(
            'rows with some (not all) NaN entries are not supported: the reference masks only rows that are NaN '
            'throughout (get_mask, pyls/types/regression.py:48-53) and lets a partly-NaN row poison the whole fit '
            '(NaN weights, :313, :324-325); impute or drop such rows before the call')
    return ~allnan


def pls_regression(X, Y, *, n_components=None, n_perm=5000, n_boot=5000, rotate=True, ci=95,
                   aggfunc='mean', permsamples=None, bootsamples=None, seed=None, verbose=True,
                   n_proc=None, **kwargs):
    """PLS regression of Y (S, T) or (S, T, C) on X (S, B) with SIMPLS; see
    pyls.pls_regression.  ``n_proc``: GPUs of this node to shard the resamples over (one process, team.py);
    ``device_ids=[...]`` names them.

    Missing data: rows of X or Y that are NaN THROUGHOUT are masked like the reference's ``get_mask`` masks them
    (pyls/types/regression.py:48-53).  A row that is only PARTLY NaN raises NotImplementedError here; in the
    reference it is not masked and turns the weights of the whole fit into NaN (regression.py:313, 324-325) --
    a behavioural difference of this drop-in, on purpose: impute or drop such rows first."""
    from .engine import Engine
    X, Y = np.asarray(X), np.asarray(Y)
    if X.ndim != 2:
        raise ValueError('Expected 2D array for `X`, got {}D array instead'.format(X.ndim))
    max_components = min(len(X) - 1, X.shape[1])
    if n_components is None:
        n_components = max_components
    else:
        n_components = int(n_components)
        if n_components > max_components:
            raise ValueError('Provided `n_components` cannot be greater than {}'
                             .format(max_components))
    if Y.ndim not in (2, 3) or len(X) != len(Y):
        raise ValueError('Provided `X` and `Y` matrices must have the same number of samples. '
                         'Provided matrices differed: X: {}, Y: {}'.format(len(X), len(Y)))
    S = len(X)
    agg = None
    third = None                                   # (C, n_boot) third-axis resamples for 3-D Y
    bootsamples_out = None
    seed = parallel.shared_seed(seed)              # all ranks draw the same index arrays
    device_ids = kwargs.pop('device_ids', None)
    transport = kwargs.pop('_transport', 'auto')
    if Y.ndim == 3:
        # regression.py:208-235
        if not callable(aggfunc) and aggfunc not in _AGGFUNCS:
            raise ValueError('Provided `aggfunc` must either be callable or one of {}'
                             .format(sorted(_AGGFUNCS)))
        agg = _AGGFUNCS.get(aggfunc, aggfunc)
        C = Y.shape[-1]
        if n_boot > 0:
            if bootsamples is None:
                # both draws restart from `seed`, as the reference's do (:212-215)
                subj = resampling.gen_bootsamp([S], 1, n_boot, seed=seed, verbose=verbose)
                third = resampling.gen_bootsamp([C], 1, n_boot, seed=seed, verbose=verbose)
            else:
                bs = np.asarray(bootsamples, dtype=object) if not isinstance(bootsamples, np.ndarray) \
                    else bootsamples
                ok = bs.shape[0] == 2 and bs.shape[-1] == n_boot
                if ok:
                    subj = np.stack([np.asarray(bs[0][i]) for i in range(n_boot)], axis=-1)
                    third = np.stack([np.asarray(bs[1][i]) for i in range(n_boot)], axis=-1)
                    ok = subj.shape[0] == S and third.shape[0] == C
                if not ok:
                    raise ValueError('Provided bootsamples arrays does not match size of provided '
                                     'input arrays or number of bootstraps requested via `nboot`.')
            packed = np.empty((2, n_boot), dtype=object)
            for i in range(n_boot):
                packed[0, i], packed[1, i] = subj[:, i], third[:, i]
            bootsamples_out, bootsamples = packed, subj
        try:
            Y_agg = agg(Y, axis=-1)
        except TypeError:
            raise TypeError('Provided callable `aggfun` must accept `axis` keyword argument to '
                            'condense an array along the specified axis.')
        if np.isnan(Y).any():
            # a subject is either complete or missing altogether: a value missing in SOME
            # slices would make the aggregated row depend on the resampled third axis
            sub_nan = np.isnan(Y).reshape(S, -1)
            if np.any(sub_nan.any(axis=1) & ~sub_nan.all(axis=1)):
                raise NotImplementedError('3-D Y with partly missing subjects is not supported')
    else:
        Y_agg = Y
        bootsamples_out = None
    kwargs.update(n_split=0, test_split=0)         # regression.py:238
    kwargs.setdefault('permindices', True)
    inputs = PLSInputs(X=X, Y=Y, groups=[S], n_cond=1, n_components=n_components, n_perm=n_perm,
                       n_boot=n_boot, rotate=rotate, ci=ci, aggfunc=aggfunc,
                       permsamples=permsamples, bootsamples=bootsamples_out if Y.ndim == 3 else bootsamples,
                       seed=seed, verbose=verbose, n_proc=n_proc, **kwargs)
    rs = resampling.check_random_state(seed)
    k = n_components
    B, T = X.shape[1], Y_agg.shape[1]

    # ---- everything drawn from `rs`, in the reference's order, on one host thread that
    # ---- starts before the data goes to the device (resampling.DrawThread): the rank-1
    # ---- randomized SVD's normal((min(B, T), 11)) per component (regression.py:103 ->
    # ---- compute.py:43-50), the permutation arrays, the bootstrap arrays
    from .engine import check_index_array

    def svd_seed_draws(r):
        for _ in range(k):
            r.normal(size=(min(B, T), 11))
    jobs = [svd_seed_draws]
    pstream = bstream = None
    if n_perm > 0:
        if permsamples is None:
            pstream = resampling.IndexStream('perm', [S], 1, n_perm)
            jobs.append(pstream.draw)
        else:
            ps = np.asarray(permsamples)
            if ps.ndim != 2 or ps.shape[0] != S:
                raise ValueError('resampling array must have shape (S, n) with S = {}; got {}'.format(S, ps.shape))
            pstream = resampling.IndexStream.of_array(check_index_array(ps, S))
    if n_boot > 0:
        if bootsamples is None:
            bstream = resampling.IndexStream('boot', [S], 1, n_boot)
            jobs.append(bstream.draw)
        else:
            bs = np.asarray(bootsamples)
            if bs.ndim != 2 or bs.shape[0] != S:
                raise ValueError('resampling array must have shape (S, n) with S = {}; got {}'.format(S, bs.shape))
            bstream = resampling.IndexStream.of_array(check_index_array(bs, S))
    from .engine import default_engine, touch_idle_release
    from . import team as _team
    touch_idle_release()                               # (a pending idle release is pushed back before the engine is looked up)
    eng = kwargs.get('_engine')
    team = None
    if eng is None and kwargs.get('_emulate') is None and parallel._dist() is None:
        # n_proc workers of the reference (pyls/utils.py:252-279) = GPUs of this node, driven from this process
        # (team.py: one context and one host thread per device, ONE all-gather)
        devices = _team.resolve_devices(inputs.get('n_proc'), device_ids)
        if devices is not None and len(devices) > 1:
            team = _team.team_for(devices, transport)
        elif devices is not None:
            eng = default_engine(devices[0])
    draws = resampling.DrawThread(rs, jobs).start()
    try:
        unrefined = 0
        if team is not None:
            res = team.run(lambda rank, world, e: _run_device(
                X, Y, Y_agg, agg, third, inputs, pstream, bstream, draws, permsamples, bootsamples, bootsamples_out,
                k, ci, e, kwargs.get('_phases') if rank == 0 else None, None, team=(rank, team)))
            unrefined = team.unrefined
        else:
            eng = eng or default_engine()
            ok = False
            with eng.lock:                             # one analysis at a time per context (shared default engine)
                try:
                    res = _run_device(X, Y, Y_agg, agg, third, inputs, pstream, bstream, draws, permsamples,
                                      bootsamples, bootsamples_out, k, ci, eng, kwargs.get('_phases'),
                                      kwargs.get('_emulate'))
                    ok = True
                finally:
                    if getattr(eng, 'ctx', None):      # nothing of this call leaks into the next one on the context
                        unrefined = eng.end_analysis(warn=ok) or 0
        Engine.warn_unrefined(unrefined, stacklevel=3)  # (outside the finally; attributed to the caller of pls_regression)
        return res
    finally:
        draws.thread.join()
        from .engine import touch_idle_release
        touch_idle_release()                       # (the cached engines are released after IDLE_RELEASE_S idle seconds)


def _run_device(X, Y, Y_agg, agg, third, inputs, pstream, bstream, draws, permsamples, bootsamples,
                bootsamples_out, k, ci, engine, phases=None, emulate=None, team=None):
    import time
    import torch
    S = len(X)
    t_last = [time.perf_counter()]
    lead = team is None or team[0] == 0                 # the rank that finishes the analysis and speaks for it

    def tick(name):                                     # per-phase wall times (bench.py --mode analysis)
        if phases is None:
            return
        torch.cuda.synchronize(engine.device)
        now = time.perf_counter()
        phases[name] = phases.get(name, 0.0) + 1e3 * (now - t_last[0])
        t_last[0] = now

    # regression.py:395-397 (on copies: the reference centres the caller's X in place)
    Yc = Y_agg.astype(np.float64) - np.nanmean(Y_agg, axis=0, keepdims=True)
    B, T = X.shape[1], Yc.shape[1]
    eng = engine
    clean = bool(np.isfinite(Yc).all())
    if clean:
        # no missing data in Y: bind X as it is -- the device centres it itself (plsx_set_data) -- and let the
        # device say whether X is clean too (a NaN / inf anywhere in a column makes its column mean non-finite):
        # the S x B matrix is not copied, centred or scanned on the host (a host pass over c5's X is 80 ms)
        import torch
        eng.set_data_regression(X, Yc, k)
        clean = bool(torch.isfinite(eng.colmean_dev()).all().item())
    if clean:
        okx = oky = mask = np.ones(S, dtype=bool)
        masked = False
    else:
        Xc = X.astype(np.float64) - np.nanmean(X, axis=0, keepdims=True)
        okx, oky = _row_ok(Xc), _row_ok(Yc)
        mask = okx & oky
        masked = not mask.all()
        eng.set_data_regression(np.nan_to_num(Xc), np.nan_to_num(Yc), k)
        if masked:
            eng.simpls_set_row_masks(okx, oky)
    res = PLSResults(inputs=inputs)
    tick('h2d_and_bind')

    # the original fit: x_weights (B, k) stay on the device -- sign rule of compute.svd (on r, proportional to the
    # x_weights column, when B > T, otherwise on c: plsx_svd_flip), centring for the sign alignment of the bootstraps,
    # scores -- and come back once, into page-locked memory, while the device resamples
    d_W, pctvar, _ = eng.simpls_decompose_dev()
    eng.simpls_set_original_dev(d_W)
    d_scores = eng.project_dev(d_W)                                # X already centred
    h_W = eng.to_host_async(d_W)
    x_scores = d_scores.cpu().numpy()
    x_scores[~okx] = np.nan                                        # NaN rows stay NaN (X @ W)
    res['x_scores'] = x_scores
    # emulate = (rank, world) of an emulated run on one GPU (bench.py --mode analysis --emulate-world): own shard, the
    # all-gather replaced by a surrogate of the same volume (parallel._surrogate_gather)
    if team is not None:
        rank, world = team[0], team[1].world
    else:
        rank, world = emulate if emulate is not None else parallel.rank_world()
    tick('decompose')

    # this rank's shards (permutations contiguous, bootstraps chunk-cyclic), launched chunk by chunk as the index rows arrive; the
    # results stay on the device until the one collective
    d_perm = d_yl = usum = usq = None
    n_perm_tot = pstream.n if pstream is not None else 0
    n_boot_tot = bstream.n if bstream is not None else 0
    from .progress import Bar                            # verbose=True: the reference's bars (pyls/utils.py:128-152)
    show = lead and bool(inputs.get('verbose')) and emulate is None and parallel.rank_world()[0] == 0
    bars = []
    if pstream is not None:
        lo, hi = parallel.shard_bounds(n_perm_tot, rank, world)
        d_perm = eng._zeros((hi - lo, k))
        bars.append(Bar('Running permutations', hi - lo, show, eng.device))
        # (a solver batch is a latency chain of ~3 k launches whatever its size: few, large chunks -- the first 2048 rows
        # are drawn in 5 ms)
        for a, b in pstream.chunks(lo, hi, first=2048):
            eng.simpls_perm_into(eng.rows_tensor(pstream.rows[a:b]), d_perm[a - lo:b - lo])
            bars[-1].queued(b - a)
    tick('permutations')
    if bstream is not None:
        bchunks = parallel.shard_chunks(n_boot_tot, rank, world)     # chunk-cyclic share of the bootstraps
        usum, usq = eng._zeros((B, k)), eng._zeros((B, k))
        d_yl = eng._zeros((sum(hi - lo for lo, hi in bchunks), T, k))
        off = 0
        eng.boot_begin(sum(hi - lo for lo, hi in bchunks))     # (plsx_boot_begin: the feature pass may move to boot_finish)
        bars.append(Bar('Running bootstraps', sum(hi - lo for lo, hi in bchunks), show, eng.device))
        for lo, hi in bchunks:
            for a, b in bstream.chunks(lo, hi, first=2048 if third is None else 256, grow=4 if third is None else 1,
                                       limit=None if third is None else 256):
                ystack = None
                if third is not None:
                    # Y aggregated over the resampled third axis, NOT centred
                    # (the reference bootstraps the original Y, regression.py:308-310, 408)
                    ystack = np.stack([agg(Y[..., third[:, i]], axis=-1) for i in range(a, b)])
                    if masked:
                        ystack = np.nan_to_num(ystack)         # all-NaN rows are dropped by the row masks
                    ystack = eng._dev(ystack, np.float64)
                eng.simpls_boot_into(eng.rows_tensor(bstream.rows[a:b]), usum, usq,
                                     d_yl[off + a - lo:off + b - lo], ystack=ystack)
                for done in bars:
                    done.poll()
                bars[-1].queued(b - a)
            off += hi - lo
        eng.boot_finish(usum, usq)
    tick('bootstraps')
    permsamp = bootsamp = None
    if pstream is not None and lead:
        permsamp = np.asarray(permsamples) if permsamples is not None else pstream.samples
    if bstream is not None and lead:
        bootsamp = np.asarray(bootsamples) if bootsamples is not None else bstream.samples
    draws.join()
    for st in (pstream, bstream):
        if st is not None and lead:
            st.warn()
    for bar in bars:
        bar.watch()
    try:
        eng.sync()                 # numerical status of the launches above is raised here
    finally:
        for bar in bars:
            bar.close()
    slices = [t for t in (d_perm, d_yl) if t is not None]
    totals = [n for t, n in ((d_perm, n_perm_tot), (d_yl, n_boot_tot)) if t is not None]
    full, summed = parallel.collect_device(slices, totals, [usum, usq] if usum is not None else [], emulate=emulate,
                                           cyclic=[len(slices) - 1] if d_yl is not None else [], team=team)
    if not lead:
        return None                                     # rank 0 holds everything the ranks computed: it finishes
    full = [t.detach().cpu().numpy() for t in full]
    if usum is not None:
        usum, usq = summed
    tick('collective')
    i = 0
    d_perm = distrib = None
    if pstream is not None:
        d_perm = np.ascontiguousarray(full[i].T)                    # (k, n_perm)
        i += 1
    if bstream is not None:
        distrib = np.ascontiguousarray(np.moveaxis(full[i], 0, -1))  # (T, k, n_boot)
    if permsamp is not None:
        res['permres']['pvals'] = hostmath.perm_sig(pctvar, d_perm)
        res['permres']['permsamples'] = permsamp
        res['permres']['perm_singval'] = d_perm

    res['y_loadings'] = Yc[mask].T @ x_scores[mask]                # regression.py:401
    y_scores = np.full((S, k), np.nan)
    y_scores[mask] = resid_yscores(x_scores[mask], Yc[mask] @ res['y_loadings'])
    res['y_scores'] = y_scores
    res['x_weights'] = _host_array(h_W)
    if bootsamp is not None:
        # add the original back, n_boot + 1 (regression.py:409-415)
        d_bsr, d_se = eng.boot_rel_dev(d_W, usum, usq, bootsamp.shape[1] + 1, add_orig=True)
        h_bsr, h_se = eng.to_host_async(d_bsr), eng.to_host_async(d_se)
        eng.sync()
        bsr, se = _host_array(h_bsr), _host_array(h_se)
        res['bootres'].update(dict(
            x_weights_normed=bsr, x_weights_stderr=se, y_loadings=res['y_loadings'],
            y_loadings_boot=distrib,
            y_loadings_ci=np.stack(eng.percentile_ci(distrib, ci=ci), -1),
            bootsamples=bootsamples_out if third is not None else bootsamp))
    res['varexp'] = pctvar                                          # regression.py:425-426
    tick('host_finish')
    return res
