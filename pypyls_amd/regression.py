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
  * 3-D Y (``aggfunc``) and NaN rows are not supported on the device path:
    NotImplementedError.
"""
import numpy as np

from . import hostmath, parallel, resampling
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


def pls_regression(X, Y, *, n_components=None, n_perm=5000, n_boot=5000, rotate=True, ci=95,
                   aggfunc='mean', permsamples=None, bootsamples=None, seed=None, verbose=True,
                   n_proc=None, **kwargs):
    """PLS regression of Y (S, T) on X (S, B) with SIMPLS; see pyls.pls_regression."""
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
    if Y.ndim == 3:
        if not callable(aggfunc) and aggfunc not in ('mean', 'median', 'sum'):
            raise ValueError("Provided `aggfunc` must either be callable or one of "
                             "['mean', 'median', 'sum']")
        raise NotImplementedError('3-D Y (aggfunc bootstrap, pyls/types/regression.py:208-235) '
                                  'is not supported by the device path yet')
    if Y.ndim != 2 or len(X) != len(Y):
        raise ValueError('Provided `X` and `Y` matrices must have the same number of samples. '
                         'Provided matrices differed: X: {}, Y: {}'.format(len(X), len(Y)))
    if np.isnan(X).any() or np.isnan(Y).any():
        raise NotImplementedError('NaN rows (get_mask, pyls/types/regression.py:48-53) are not '
                                  'supported by the device path yet')
    kwargs.update(n_split=0, test_split=0)         # regression.py:238
    kwargs.setdefault('permindices', True)
    S = len(X)
    inputs = PLSInputs(X=X, Y=Y, groups=[S], n_cond=1, n_components=n_components, n_perm=n_perm,
                       n_boot=n_boot, rotate=rotate, ci=ci, aggfunc=aggfunc,
                       permsamples=permsamples, bootsamples=bootsamples, seed=seed,
                       verbose=verbose, n_proc=n_proc, **kwargs)
    rs = resampling.check_random_state(seed)
    k = n_components

    Xc = X.astype(np.float64) - X.mean(axis=0, keepdims=True)     # regression.py:395-396
    Yc = Y.astype(np.float64) - Y.mean(axis=0, keepdims=True)
    B, T = Xc.shape[1], Yc.shape[1]
    eng = kwargs.get('_engine') or Engine()
    eng.set_data_regression(Xc, Yc, k)
    res = PLSResults(inputs=inputs)

    # the reference's rank-1 randomized SVD draws normal((min(B, T), 11)) per
    # component from self.rs (regression.py:103 -> compute.py:43-50)
    for _ in range(k):
        rs.normal(size=(min(B, T), 11))
    W, pctvar, cvec, _ = eng.simpls_decompose()
    # sign rule of compute.svd: on r (prop. to the x_weights column) when B > T,
    # otherwise on c
    lead = W if B > T else cvec
    idx = np.argmax(np.abs(lead), axis=0)
    signs = np.sign(lead[idx, np.arange(k)])
    signs[signs == 0] = 1.0
    W = W * signs
    eng.simpls_set_original(W)
    res['x_weights'] = W
    res['x_scores'] = eng.project(W)                               # X already centred
    rank, world = parallel.rank_world()

    permsamp = bootsamp = local_perm = local_dist = usum = usq = None
    if n_perm > 0:
        permsamp = permsamples
        if permsamp is None:
            permsamp = resampling.gen_permsamp([S], 1, n_perm, seed=rs, verbose=verbose)
        permsamp = np.asarray(permsamp)
    if n_boot > 0:
        bootsamp = bootsamples
        if bootsamp is None:
            bootsamp = resampling.gen_bootsamp([S], 1, n_boot, seed=rs, verbose=verbose)
        bootsamp = np.asarray(bootsamp)
    if permsamp is not None:
        lo, hi = parallel.shard_bounds(permsamp.shape[1], rank, world)
        local_perm = eng.simpls_perm(permsamp[:, lo:hi]) if hi > lo else np.zeros((k, 0))
    if bootsamp is not None:
        lo, hi = parallel.shard_bounds(bootsamp.shape[1], rank, world)
        if hi > lo:
            usum, usq, local_dist = eng.simpls_boot(bootsamp[:, lo:hi])
        else:
            usum, usq = eng._zeros((B, k)), eng._zeros((B, k))
            local_dist = np.zeros((T, k, 0))
    d_perm, distrib, usum, usq = parallel.collect(
        local_perm, permsamp.shape[1] if permsamp is not None else 0,
        local_dist, bootsamp.shape[1] if bootsamp is not None else 0, usum, usq)
    if permsamp is not None:
        res['permres']['pvals'] = hostmath.perm_sig(pctvar, d_perm)
        res['permres']['permsamples'] = permsamp
        res['permres']['perm_singval'] = d_perm

    res['y_loadings'] = Yc.T @ res['x_scores']                     # regression.py:401
    res['y_scores'] = resid_yscores(res['x_scores'], Yc @ res['y_loadings'])
    if bootsamp is not None:
        # add the original back, n_boot + 1 (regression.py:409-415)
        bsr, se = eng.boot_rel(W, usum, usq, bootsamp.shape[1] + 1, add_orig=True)
        res['bootres'].update(dict(
            x_weights_normed=bsr, x_weights_stderr=se, y_loadings=res['y_loadings'],
            y_loadings_boot=distrib,
            y_loadings_ci=np.stack(hostmath.boot_ci(distrib, ci=ci), -1), bootsamples=bootsamp))
    res['varexp'] = pctvar                                          # regression.py:425-426
    return res
