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
  * ``n_proc`` is accepted and ignored: the joblib process pool
    (pyls/utils.py:252-279) is what the device replaces.
  * When torch.distributed is initialised the resamples are sharded across
    ranks and collected with one all-gather (pypyls_amd/parallel.py).
"""
import warnings

import numpy as np

from . import hostmath, parallel, resampling
from .structures import PLSInputs, PLSResults

_METHOD_CODE = {'behavioral': 0, 'meancentered': 1}


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
        self.inputs = PLSInputs(X=X, Y=Y, groups=groups, n_cond=n_cond, **kwargs)
        # under torch.distributed every rank must draw the same index arrays
        self.rs = resampling.check_random_state(parallel.shared_seed(self.inputs.get('seed')))
        self.cells = resampling.cell_of_row(groups, n_cond)
        self.n_cells = len(groups) * n_cond
        self.engine = kwargs.get('_engine')

    # ------------------------------------------------------------------
    def run(self):
        from .engine import Engine
        inp = self.inputs
        X = _as_float_array(inp.X, 'X')
        Y = None if self.method == 'meancentered' else _as_float_array(inp.Y, 'Y')
        eng = self.engine or Engine()
        eng.set_data(X, Y, self.cells, len(inp.groups), inp.n_cond, _METHOD_CODE[self.method],
                     mean_centering=inp.get('mean_centering') or 0,
                     covariance=bool(inp.get('covariance')))
        L = eng.L
        res = PLSResults(inputs=inp)
        # finite-input check (sklearn check_X_y in compute.xcorr, compute.py:78):
        # a NaN / inf anywhere in a column of X makes the device column mean
        # non-finite, which avoids a host pass over the (S, B) matrix
        xmean = eng.colmean()
        if not np.all(np.isfinite(xmean)):
            raise ValueError('Input `X` contains NaN, infinity or a value too large')
        if Y is not None:
            _check_finite(Y, 'Y')

        # ---- original decomposition (BasePLS.svd, base.py:362-364) -------
        # the reference's randomized_svd consumes normal((L, L + 10)) from
        # self.rs here; draw it so that later index arrays match for a seed
        self.rs.normal(size=(L, L + 10))
        xw, sv, yw = eng.decompose()
        xw, yw = hostmath.sign_convention(xw, yw)
        eng.set_original(xw, sv, yw)
        res['x_weights'], res['y_weights'] = xw, yw
        res['x_scores'] = eng.project(xw) + (xmean @ xw)[None, :]
        rank, world = parallel.rank_world()

        # ---- resampling: local shard of permutations and bootstraps, then
        # ---- ONE collective (parallel.collect) ----------------------------
        n_perm = inp.get('n_perm') or 0
        n_boot = inp.get('n_boot') or 0
        permsamp = bootsamp = local_perm = local_dist = usum = usq = ystack = None
        if n_perm > 0:
            # BasePLS.permutation, base.py:601-652
            permsamp = inp.get('permsamples')
            if permsamp is None:
                permsamp = resampling.gen_permsamp(inp.groups, inp.n_cond, n_perm, seed=self.rs,
                                                   verbose=inp.get('verbose'))
            permsamp = np.asarray(permsamp)
            if not inp.get('permindices'):
                # pre-permuted Y matrices, (n_perm, S, T) (base.py:636-639)
                if self.method != 'behavioral' or permsamp.ndim != 3:
                    raise ValueError('permindices=False expects `permsamples` of shape '
                                     '(n_perm, S, T) and behavioral PLS')
                ystack, permsamp = permsamp.astype(np.float64, copy=False), None
        n_split = inp.get('n_split')
        orig_splits = None
        if (permsamp is not None or ystack is not None) and n_split is not None:
            # the reference draws the split masks of the ORIGINAL data from
            # self.rs after the permutation arrays and before the bootstrap
            # arrays (base.py:373-380); permutation i uses a fresh
            # RandomState(i) (base.py:705-708, 738-742)
            orig_splits = inp.get('_splitsamples')
            if orig_splits is None:
                orig_splits = resampling.gen_splits(inp.groups, inp.n_cond, n_split, seed=self.rs,
                                                    test_size=0.5)
        if n_boot > 0:
            # BasePLS.bootstrap, base.py:439-528 (index arrays drawn AFTER the
            # permutation arrays, as in the reference's run_pls order)
            bootsamp = inp.get('bootsamples')
            if bootsamp is None:
                bootsamp = resampling.gen_bootsamp(inp.groups, inp.n_cond, n_boot, seed=self.rs,
                                                   verbose=inp.get('verbose'))
            bootsamp = np.asarray(bootsamp)
        n_perm_tot = 0 if permsamp is None else permsamp.shape[1]
        if ystack is not None:
            n_perm_tot = ystack.shape[0]
            lo, hi = parallel.shard_bounds(n_perm_tot, rank, world)
            local_perm = eng.perm_ystack(ystack[lo:hi], rotate=bool(inp.get('rotate', True))) \
                if hi > lo else np.zeros((L, 0))
        if permsamp is not None:
            lo, hi = parallel.shard_bounds(permsamp.shape[1], rank, world)
            local_perm = eng.perm(permsamp[:, lo:hi], rotate=bool(inp.get('rotate', True))) \
                if hi > lo else np.zeros((L, 0))
        if bootsamp is not None:
            lo, hi = parallel.shard_bounds(bootsamp.shape[1], rank, world)
            if hi > lo:
                usum, usq, local_dist = eng.boot(bootsamp[:, lo:hi])
            else:
                usum, usq = eng._zeros((eng.B, L)), eng._zeros((eng.B, L))
                local_dist = np.zeros((eng.Tp, L, 0))
        local_uc = local_vc = None
        if orig_splits is not None:
            lo, hi = parallel.shard_bounds(n_perm_tot, rank, world)
            pmasks = inp.get('_perm_splitsamples')
            if pmasks is None:
                pmasks = resampling.gen_splits_seeded(inp.groups, inp.n_cond, n_split, np.arange(lo, hi),
                                                      test_size=0.5, rows=True)
                rows = True
            else:
                pmasks, rows = np.asarray(pmasks)[lo:hi], False
            if hi > lo:
                if ystack is not None:
                    uc, vc = eng.split_half(pmasks, ystack=ystack[lo:hi], mask_rows=rows)
                else:
                    uc, vc = eng.split_half(pmasks, perms=permsamp[:, lo:hi], mask_rows=rows)
                local_uc, local_vc = uc.mean(axis=-1).T, vc.mean(axis=-1).T      # (L, p_loc)
            else:
                local_uc = local_vc = np.zeros((L, 0))
            # ride along with the permutation block of the single collective
            local_perm = np.vstack([local_perm, local_uc, local_vc])
        d_perm, distrib, usum, usq = parallel.collect(
            local_perm, n_perm_tot,
            local_dist, bootsamp.shape[1] if bootsamp is not None else 0, usum, usq)
        if orig_splits is not None:
            d_perm, ucorrs, vcorrs = d_perm[:L], d_perm[L:2 * L], d_perm[2 * L:]
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
            bs = xw * sv[None, :]
            if self.method == 'behavioral':
                # add the original back, n_boot + 1 (behavioral.py:201-207)
                bsr, se = eng.boot_rel(bs, usum, usq, bootsamp.shape[1] + 1, add_orig=True)
                res['bootres'].update(dict(
                    x_weights_normed=bsr, x_weights_stderr=se,
                    y_loadings=res['y_loadings'].copy(), y_loadings_boot=distrib,
                    y_loadings_ci=np.stack(eng.percentile_ci(distrib, ci=inp.get('ci', 95)), -1),
                    bootsamples=bootsamp))
            else:
                # no add-back, n_boot (meancentered.py:162-164)
                bsr, se = eng.boot_rel(bs, usum, usq, bootsamp.shape[1], add_orig=False)
                res['bootres'].update(dict(
                    x_weights_normed=bsr, x_weights_stderr=se, bootsamples=bootsamp,
                    contrast=contrast, contrast_boot=distrib,
                    contrast_ci=np.stack(eng.percentile_ci(distrib, ci=inp.get('ci', 95)), -1)))

        # ---- cross-validation (behavioral.py:219-221, 82-170) ----------------
        if (self.method == 'behavioral' and inp.get('test_split') is not None
                and (inp.get('test_size') or 0) > 0):
            splits = inp.get('_cvsplits')
            if splits is None:
                splits = resampling.gen_splits(inp.groups, inp.n_cond, inp.test_split, seed=self.rs,
                                               test_size=inp.test_size)
            lo, hi = parallel.shard_bounds(splits.shape[1], rank, world)
            if hi > lo:
                r, r2 = eng.crossval(splits[:, lo:hi])
                local_cv = np.vstack([r, r2])
            else:
                local_cv = np.zeros((2 * Y.shape[1], 0))
            cv, _, _, _ = parallel.collect(local_cv, splits.shape[1], None, 0, None, None)
            Tn = Y.shape[1]
            res['cvres'].update(dict(pearson_r=cv[:Tn], r_squared=cv[Tn:]))

        res['varexp'] = hostmath.varexp(sv)
        res['singvals'] = sv
        self.engine_used = eng
        return res


def behavioral_pls(X, Y, *, groups=None, n_cond=1, n_perm=5000, n_boot=5000, n_split=0,
                   test_size=0.25, test_split=100, covariance=False, rotate=True, ci=95,
                   permsamples=None, bootsamples=None, seed=None, verbose=True, n_proc=None,
                   **kwargs):
    """Behavioral PLS of X (S, B) against Y (S, T); see pyls.behavioral_pls."""
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
