"""
oracle/cpu_ref.py -- CPU restatement of the pyls PLS-C / SIMPLS resampling path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module.  The product
(``pypyls_amd``) never imports it and has no CPU fallback.

Every function restates, from the maths, what the reference does at the cited
``file:line`` (paths relative to the reference checkout, ``pyls/...``).  The
only intentional deviation: the reference calls
``sklearn.utils.extmath.randomized_svd`` (third party, unpinned; sklearn 1.7.2
in the survey container).  For PLS-C it is called with
``n_components=min(shape)`` so the sketch spans the whole row space and the
result equals the exact thin SVD (measured 1e-15); this file therefore uses
``numpy.linalg.svd`` plus sklearn's ``svd_flip`` sign rule.  For SIMPLS the
reference's rank-1 call is approximate when T > 11;
here the exact leading singular triplet is used, so for T > 11 parity against
the reference itself is "unpinned" (the reference has no test that pins any
SIMPLS number) while T <= 11 matches to rounding.

Pinned against the reference: ``tests/golden/make_golden.py`` imports
``/root/reference/pyls`` in the build container, runs it on the cases in
``tests/golden/*.npz`` and stores inputs + outputs; ``tests/test_oracle.py``
checks this module against every one of those fixtures (and the Linnerud
doctest numbers of ``docs/user_guide/behavioral.rst:143-245``).

numpy only -- no scipy / sklearn / torch.
"""

import numpy as np


# --------------------------------------------------------------------------
# labels / dummy coding                                   pyls/utils.py:155-197
# --------------------------------------------------------------------------

def dummy_label(groups, n_cond=1):
    """Cell label (1-based) of every row; rows are ordered group-major, then
    condition, then subject.  pyls/utils.py:178-197."""
    groups = [int(g) for g in groups]
    counts = np.repeat(groups, n_cond)
    return np.repeat(np.arange(len(groups) * n_cond) + 1, counts)


def dummy_code(groups, n_cond=1):
    """One-hot (S, J) cell membership.  pyls/utils.py:155-175."""
    lab = dummy_label(groups, n_cond)
    return (lab[:, None] == np.unique(lab)[None, :]).astype(int)


# --------------------------------------------------------------------------
# numeric kernels                                            pyls/compute.py
# --------------------------------------------------------------------------

def xcorr(X, Y, covariance=False):
    """Cross-correlation (or cross-covariance) of the columns of Y with the
    columns of X -> (T, B).  pyls/compute.py:55-94 (z-score with ddof=1 at
    :84-85, centre-only at :87, product / (n - 1) at :92)."""
    X = np.asarray(X, dtype=float)
    Y = np.asarray(Y, dtype=float)
    if X.ndim != 2 or Y.ndim != 2 or len(X) != len(Y):
        raise ValueError('X and Y must be 2-D with the same number of rows')
    Xc = X - X.mean(axis=0)
    Yc = Y - Y.mean(axis=0)
    if not covariance:
        with np.errstate(divide='ignore', invalid='ignore'):
            Xc = Xc / X.std(axis=0, ddof=1)
            Yc = Yc / Y.std(axis=0, ddof=1)
    return (Yc.T @ Xc) / (len(X) - 1)


def normalize(X, axis=0):
    """Unit L2 norm along ``axis``; all-zero vectors stay zero.
    pyls/compute.py:97-126."""
    X = np.array(X, dtype=float)
    nrm = np.linalg.norm(X, axis=axis, keepdims=True)
    safe = np.where(nrm == 0, 1.0, nrm)
    out = X / safe
    out[np.broadcast_to(nrm == 0, out.shape)] = 0
    return out


def svd_flip_first(A, Bt):
    """sklearn ``svd_flip(u, v, u_based_decision=True)``: make the entry of
    largest magnitude in every column of ``A`` positive, flipping the matching
    row of ``Bt``."""
    idx = np.argmax(np.abs(A), axis=0)
    signs = np.sign(A[idx, np.arange(A.shape[1])])
    signs[signs == 0] = 1.0
    return A * signs[None, :], Bt * signs[:, None]


def svd(crosscov, n_components=None):
    """Thin SVD of ``crosscov`` (T', B) -> U (B, L), diag(d) (L, L), V (T', L).

    pyls/compute.py:10-52.  The reference decomposes ``crosscov.T`` when
    T' <= B (so the sign rule is applied to the columns of U, :43-46) and
    ``crosscov`` otherwise (sign rule on the columns of V, :47-50).
    """
    crosscov = np.asarray(crosscov, dtype=float)
    if n_components is None:
        n_components = min(crosscov.shape)
    if crosscov.shape[0] <= crosscov.shape[1]:
        U, d, Vt = np.linalg.svd(crosscov.T, full_matrices=False)
        U, Vt = svd_flip_first(U, Vt)
        U, d, V = U[:, :n_components], d[:n_components], Vt[:n_components].T
    else:
        V, d, Ut = np.linalg.svd(crosscov, full_matrices=False)
        V, Ut = svd_flip_first(V, Ut)
        V, d, U = V[:, :n_components], d[:n_components], Ut[:n_components].T
    return U, np.diag(d), V


def procrustes(original, permuted, singular):
    """Rotate ``permuted @ singular`` onto ``original``.
    pyls/compute.py:240-264: temp = original.T @ permuted = N S P ;
    result = permuted @ singular @ (P.T @ N.T)."""
    temp = original.T @ permuted
    N, _, P = np.linalg.svd(temp, full_matrices=False)
    return permuted @ singular @ (P.T @ N.T)


#: a latent variable is "live" when its singular value exceeds RANK_RTOL times
#: the largest one.  Mean-centred PLS always has exactly-zero singular values
#: (rank of the centring); the reference's own comparator masks them
#: (pyls/tests/matlab.py:160).
RANK_RTOL = 1e-6


def live_lvs(d):
    d = np.diag(d) if np.ndim(d) == 2 else np.asarray(d)
    return d > RANK_RTOL * d.max()


def procrustes_live(original, permuted, singular, live_o=None, live_p=None):
    """Procrustes rotation restricted to the live latent variables.

    Identical to :func:`procrustes` when every LV is live.  When the
    decomposition is rank deficient (mean-centred PLS) the reference feeds the
    ARBITRARY null-space singular vectors returned by randomized_svd into
    ``original.T @ permuted`` (compute.py:260); for the thin (B, L) bootstrap
    vectors the polar factor then depends on those noise-defined vectors and
    the reference's own output moves by ~1e-2 relative when only the SVD seed
    changes (measured).  The well-defined statistic -- and
    what the product computes -- aligns live bootstrap vectors with live
    original vectors only; null columns of the result are zero (they are
    multiplied by a zero singular value in the reference as well).
    """
    L = permuted.shape[1]
    live_o = np.ones(original.shape[1], bool) if live_o is None else live_o
    live_p = np.ones(L, bool) if live_p is None else live_p
    temp = original[:, live_o].T @ permuted[:, live_p]
    N, _, P = np.linalg.svd(temp, full_matrices=False)
    sing = np.asarray(singular)[np.ix_(live_p, live_p)]
    out = np.zeros((permuted.shape[0], original.shape[1]))
    out[:, live_o] = permuted[:, live_p] @ sing @ (P.T @ N.T)
    return out


def get_group_mean(X, Y, n_cond=1, mean_centering=0):
    """Reference mean removed from each cell mean.  pyls/compute.py:267-317."""
    X = np.asarray(X, dtype=float)
    Y = np.asarray(Y)
    J = Y.shape[1]
    if mean_centering == 0:
        # mean of each GROUP over all of its conditions, repeated per cond
        sizes = Y[:, ::n_cond].sum(axis=0).astype(int) * n_cond
        grp = dummy_code(sizes).T.astype(bool)
        gm = np.vstack([X[g].mean(axis=0) for g in grp])
        return np.repeat(gm, n_cond, axis=0)
    if mean_centering == 1:
        # mean of each CONDITION across groups (mean of the cell means)
        cm = np.vstack([X[c].mean(axis=0) for c in Y.T.astype(bool)])
        cm = cm.reshape(-1, n_cond, X.shape[1]).mean(axis=0)
        return np.tile(cm.T, J // n_cond).T
    if mean_centering == 2:
        return np.repeat(X.mean(axis=0)[None], J, axis=0)
    raise ValueError('Mean centering type must be in [0, 1, 2].')


def get_mean_center(X, Y, n_cond=1, mean_centering=0, means=True):
    """Cell means minus the reference mean (means=True, (J, B)) or the rows
    with the reference mean of their cell removed (means=False, (S, B)).
    pyls/compute.py:320-357."""
    X = np.asarray(X, dtype=float)
    mc = get_group_mean(X, Y, n_cond=n_cond, mean_centering=mean_centering)
    cells = np.asarray(Y).T.astype(bool)
    if means:
        return np.vstack([X[c].mean(axis=0) - mc[n] for n, c in enumerate(cells)])
    return np.vstack([X[c] - mc[n][None] for n, c in enumerate(cells)])


def efficient_corr(x, y):
    """Pearson r of matching columns.  pyls/compute.py:360-391."""
    x, y = np.vstack(x).astype(float), np.vstack(y).astype(float)
    if x.shape != y.shape and x.shape[-1] != 1 and y.shape[-1] != 1:
        raise ValueError('x and y must have matching shapes or one must be '
                         'a column vector')
    with np.errstate(divide='ignore', invalid='ignore'):
        zx = (x - x.mean(axis=0)) / x.std(axis=0, ddof=1)
        zy = (y - y.mean(axis=0)) / y.std(axis=0, ddof=1)
    return np.clip(np.sum(zx * zy, axis=0) / (len(x) - 1), -1, 1)


def perm_sig(orig, perm):
    """(count(perm > orig) + 1) / (P + 1), strict '>'.
    pyls/compute.py:154-181.  ``orig`` is the (L, L) diagonal matrix."""
    count = np.sum(perm > np.diag(orig)[:, None], axis=1) + 1
    return count / (perm.shape[-1] + 1)


def boot_ci(boot, ci=95):
    """Percentile interval along the last axis.  pyls/compute.py:184-209."""
    low = (100 - ci) / 2
    lo, hi = np.percentile(boot, [low, 100 - low], axis=-1)
    return lo, hi


def boot_rel(orig, u_sum, u_square, n_boot):
    """Bootstrap ratio and standard error.  pyls/compute.py:212-237."""
    u_se = np.sqrt(np.abs(u_square - (u_sum ** 2) / n_boot) / (n_boot - 1))
    with np.errstate(divide='ignore', invalid='ignore'):
        bsr = orig / u_se
    return bsr, u_se


def varexp(singular):
    """diag(s^2 / sum s^2).  pyls/compute.py:394-414."""
    s2 = np.diag(singular) ** 2
    return np.diag(s2 / s2.sum())


# --------------------------------------------------------------------------
# method hooks                      pyls/types/behavioral.py, meancentered.py
# --------------------------------------------------------------------------

class Spec(object):
    """What the reference keeps in ``self.inputs`` that the hooks consult."""

    def __init__(self, method, groups, n_cond=1, covariance=False,
                 mean_centering=0, rotate=True, n_split=None):
        self.method = method                      # 'behavioral' | 'meancentered'
        self.groups = [int(g) for g in groups]
        self.n_cond = int(n_cond)
        self.covariance = bool(covariance)
        self.mean_centering = int(mean_centering)
        self.rotate = bool(rotate)
        self.n_split = n_split
        self.dummy = dummy_code(self.groups, self.n_cond)


def gen_covcorr(spec, X, Y, dummy):
    """behavioral.py:27-52 (stack of per-cell xcorr) /
    meancentered.py:50-73 (cell means minus reference mean)."""
    if spec.method == 'behavioral':
        return np.vstack([xcorr(X[c], Y[c], covariance=spec.covariance)
                          for c in dummy.T.astype(bool)])
    return get_mean_center(X, Y, spec.n_cond, spec.mean_centering, means=True)


def gen_distrib(spec, X, Y, original, dummy):
    """behavioral.py:54-80 / meancentered.py:75-102."""
    if spec.method == 'behavioral':
        return gen_covcorr(spec, X @ normalize(original), Y, dummy)
    usc = get_mean_center(X, Y, spec.n_cond, spec.mean_centering, means=False)
    usc = usc @ normalize(original)
    return np.vstack([usc[c].mean(axis=0) for c in np.asarray(Y).T.astype(bool)])


def decompose(spec, X, Y, dummy=None):
    """BasePLS.svd, pyls/base.py:401-437."""
    if dummy is None:
        dummy = spec.dummy
    return svd(gen_covcorr(spec, X, Y, dummy))


def make_permutation(spec, X, Y, perminds):
    """base.py:578-599 permutes Y; meancentered.py:104-125 permutes X."""
    if spec.method == 'behavioral':
        return X, Y[perminds]
    return X[perminds], Y


def split_half(spec, X, Y, ud, vd, splitsamp, dummy=None):
    """BasePLS.split_half, pyls/base.py:714-770, with the split masks
    (S, n_split) supplied by the caller instead of drawn at :738-742."""
    if dummy is None:
        dummy = spec.dummy
    n_split = splitsamp.shape[1]
    ucorr = np.zeros((ud.shape[-1], n_split))
    vcorr = np.zeros((vd.shape[-1], n_split))
    for i in range(n_split):
        spl = splitsamp[:, i].astype(bool)
        D1 = gen_covcorr(spec, X[spl], Y[spl], dummy[spl])
        D2 = gen_covcorr(spec, X[~spl], Y[~spl], dummy[~spl])
        ucorr[:, i] = efficient_corr(D1.T @ vd, D2.T @ vd)
        vcorr[:, i] = efficient_corr(D1 @ ud, D2 @ ud)
    return ucorr.mean(axis=-1), vcorr.mean(axis=-1)


def single_perm(spec, X, Y, perminds, y_weights, splitsamp=None):
    """BasePLS._single_perm with use_permind=True, pyls/base.py:654-712."""
    Xp, Yp = make_permutation(spec, X, Y, perminds)
    U, d, V = decompose(spec, Xp, Yp)
    if spec.rotate:
        ssd = np.sqrt(np.sum(procrustes(y_weights, V, d) ** 2, axis=0))
    else:
        ssd = np.diag(d)
    if splitsamp is not None:
        di = np.linalg.inv(d)
        ucorr, vcorr = split_half(spec, Xp, Yp, U @ di, V @ di, splitsamp)
    else:
        ucorr = vcorr = None
    return ssd, ucorr, vcorr


def single_boot(spec, X, Y, inds, x_weights, d_orig=None):
    """BasePLS._single_boot, pyls/base.py:530-576; the rotation at :570 uses
    :func:`procrustes_live` (equal to the reference's when full rank)."""
    U, d, _ = decompose(spec, X[inds], Y[inds])
    live_o = None if d_orig is None else live_lvs(d_orig)
    U_boot = procrustes_live(x_weights, U, d, live_o, live_lvs(d))
    distrib = gen_distrib(spec, X[inds], Y[inds], x_weights, spec.dummy)
    return distrib, U_boot


def rescale_test(X_train, X_test, Y_train, U, V):
    """compute.rescale_test, pyls/compute.py:129-151: z-map the test rows with
    the training mean / std (ddof=1), project, add the training mean of Y."""
    mu = X_train.mean(axis=0)
    sd = X_train.std(axis=0, ddof=1)
    with np.errstate(divide='ignore', invalid='ignore'):
        X_resc = (X_test - mu) / sd
    return (X_resc @ U @ V.T) + Y_train.mean(axis=0, keepdims=True)


def r2_score_raw(y_true, y_pred):
    """sklearn.metrics.r2_score(..., multioutput='raw_values') for the
    non-degenerate case: 1 - SS_res / SS_tot per column."""
    ss_res = np.sum((y_true - y_pred) ** 2, axis=0)
    ss_tot = np.sum((y_true - y_true.mean(axis=0)) ** 2, axis=0)
    return 1.0 - ss_res / ss_tot


def single_crossval(spec, X, Y, inds):
    """BehavioralPLS._single_crossval, pyls/types/behavioral.py:126-170.
    ``inds`` True = training row."""
    inds = np.asarray(inds, dtype=bool)
    dummy = spec.dummy
    Xtr, Ytr, dtr = X[inds], Y[inds], dummy[inds]
    Xte, Yte, dte = X[~inds], Y[~inds], dummy[~inds]
    U, d, V = decompose(spec, Xtr, Ytr, dtr)
    pred = []
    for n, V_spl in enumerate(np.split(V, dummy.shape[-1])):
        tr, te = dtr[:, n].astype(bool), dte[:, n].astype(bool)
        pred.append(rescale_test(Xtr[tr], Xte[te], Ytr[tr], U, V_spl))
    pred = np.vstack(pred)
    return efficient_corr(Yte, pred), r2_score_raw(Yte, pred)


def crossval(spec, X, Y, splits):
    """BehavioralPLS.crossval (behavioral.py:82-124) with the (S, n) train
    masks supplied.  Returns pearson_r, r_squared of shape (T, n)."""
    out = [single_crossval(spec, X, Y, splits[:, i]) for i in range(splits.shape[1])]
    return np.stack([o[0] for o in out], -1), np.stack([o[1] for o in out], -1)


# --------------------------------------------------------------------------
# full drivers (what BasePLS.run_pls + the subclass run_pls assemble)
# --------------------------------------------------------------------------

def run_plsc(X, Y=None, *, method='behavioral', groups=None, n_cond=1,
             covariance=False, mean_centering=0, rotate=True, ci=95,
             permsamples=None, bootsamples=None, splitsamples=None,
             perm_splitsamples=None):
    """PLS-C analysis with caller-supplied resampling arrays.

    Follows BasePLS.run_pls (pyls/base.py:341-399) and the subclass
    ``run_pls`` (behavioral.py:172-227, meancentered.py:127-179) with
    ``permindices=True`` and ``test_split=0``.

    permsamples (S, P) / bootsamples (S, R) are index arrays; ``splitsamples``
    (S, n_split) bool are the split masks used for the ORIGINAL data and
    ``perm_splitsamples`` (P, S, n_split) those used inside permutation i
    (the reference draws them with gen_splits(seed=self.rs) resp. seed=i).
    Returns a plain nested dict with the PLSResults key layout.
    """
    X = np.asarray(X, dtype=float)
    if groups is None:
        groups = [len(X) // n_cond]
    spec = Spec(method, groups, n_cond, covariance, mean_centering, rotate)
    if method == 'meancentered':
        Y = spec.dummy
    Y = np.asarray(Y, dtype=float)
    res = dict(permres={}, bootres={}, splitres={})

    U, d, V = decompose(spec, X, Y)
    res['x_weights'], res['y_weights'] = U, V
    res['x_scores'] = X @ U

    if permsamples is not None:
        P = permsamples.shape[1]
        out = [single_perm(spec, X, Y, permsamples[:, i], V,
                           None if perm_splitsamples is None
                           else perm_splitsamples[i]) for i in range(P)]
        d_perm = np.stack([o[0] for o in out], axis=-1)
        res['permres'] = dict(pvals=perm_sig(d, d_perm), perm_singval=d_perm,
                              permsamples=permsamples)
        if splitsamples is not None:
            ucorrs = np.stack([o[1] for o in out], axis=-1)
            vcorrs = np.stack([o[2] for o in out], axis=-1)
            di = np.linalg.inv(d)
            ou, ov = split_half(spec, X, Y, U @ di, V @ di, splitsamples)
            ull, uul = boot_ci(ucorrs, ci=ci)
            vll, vul = boot_ci(vcorrs, ci=ci)
            res['splitres'] = dict(
                ucorr=ou, vcorr=ov,
                ucorr_pvals=perm_sig(np.diag(ou), ucorrs),
                vcorr_pvals=perm_sig(np.diag(ov), vcorrs),
                ucorr_lolim=ull, vcorr_lolim=vll,
                ucorr_uplim=uul, vcorr_uplim=vul,
                ucorr_perm=ucorrs, vcorr_perm=vcorrs)

    if method == 'behavioral':
        cells = np.repeat(spec.groups, spec.n_cond)
        edges = np.cumsum(cells)[:-1]
        res['y_scores'] = np.vstack([
            y @ v for y, v in zip(np.split(Y, edges), np.split(V, len(cells)))])
        res['y_loadings'] = gen_covcorr(spec, res['x_scores'], Y, spec.dummy)
    else:
        res['y_scores'] = Y @ V
        bs_dm = get_mean_center(X, Y, n_cond, mean_centering, False) @ U
        contrast = np.vstack([bs_dm[c].mean(axis=0) for c in Y.T.astype(bool)])

    if bootsamples is not None:
        R = bootsamples.shape[1]
        u_sum, u_square = np.zeros_like(U), np.zeros_like(U)
        distrib = []
        for i in range(R):
            dist, ub = single_boot(spec, X, Y, bootsamples[:, i], U, d)
            u_sum += ub
            u_square += ub ** 2
            distrib.append(dist)
        distrib = np.stack(distrib, axis=-1)
        bs = U @ d
        if method == 'behavioral':
            # behavioral.py:201-207 adds the original back and uses R + 1
            u_sum, u_square = u_sum + bs, u_square + bs ** 2
            bsr, se = boot_rel(bs, u_sum, u_square, R + 1)
            res['bootres'] = dict(
                x_weights_normed=bsr, x_weights_stderr=se,
                y_loadings=res['y_loadings'].copy(), y_loadings_boot=distrib,
                y_loadings_ci=np.stack(boot_ci(distrib, ci=ci), -1),
                bootsamples=bootsamples)
        else:
            # meancentered.py:162-164: no add-back, n_boot
            bsr, se = boot_rel(bs, u_sum, u_square, R)
            res['bootres'] = dict(
                x_weights_normed=bsr, x_weights_stderr=se,
                contrast=contrast, contrast_boot=distrib,
                contrast_ci=np.stack(boot_ci(distrib, ci=ci), -1),
                bootsamples=bootsamples)

    res['varexp'] = np.diag(varexp(d))
    res['singvals'] = np.diag(d)
    return res


# --------------------------------------------------------------------------
# SIMPLS regression                              pyls/types/regression.py
# --------------------------------------------------------------------------

def _top_triplet(Cov):
    """Leading singular triplet of Cov (B, T) with the sign rule the
    reference's ``compute.svd(Cov, n_components=1)`` applies
    (regression.py:103 -> compute.py:43-50).  ``compute.svd`` treats its
    argument as 'crosscov' of shape (rows, cols): rows <= cols decomposes the
    transpose and flips on the first factor of THAT decomposition."""
    if Cov.shape[0] <= Cov.shape[1]:
        A, s, Bt = np.linalg.svd(Cov.T, full_matrices=False)
        A, Bt = svd_flip_first(A[:, :1], Bt[:1])
        first, second = A, Bt.T          # first: (cols, 1), second: (rows, 1)
    else:
        A, s, Bt = np.linalg.svd(Cov, full_matrices=False)
        A, Bt = svd_flip_first(A[:, :1], Bt[:1])
        second, first = A, Bt.T          # A: (rows, 1), Bt.T: (cols, 1)
    # compute.svd returns (U, d, V) with U of length cols(crosscov) when
    # rows <= cols, i.e. ``ci, si, ri = svd(Cov)``: ci has len Cov.shape[1],
    # ri has len Cov.shape[0] -- in both branches ri spans the rows (B).
    return first, s[0], second


def resid_yscores(x_scores, y_scores):
    """regression.py:9-45: two rounds of MGS of column c of y_scores against
    x_scores columns < c."""
    x_scores = np.array(x_scores, dtype=float)
    y_scores = np.array(y_scores, dtype=float)
    for comp in range(x_scores.shape[1]):
        ui = y_scores[:, [comp]]
        for _ in range(2):
            for j in range(comp):
                tj = x_scores[:, [j]]
                ui = ui - (tj.T @ ui) * tj
        y_scores[:, [comp]] = ui
    return y_scores


def simpls(X, Y, n_components=None):
    """SIMPLS (de Jong 1993) as restated from regression.py:56-186.  Returns
    the quantities the resampling path uses: x_weights, x_loadings,
    y_loadings, x_scores, y_scores and pctvar (2, k)."""
    X = np.asarray(X, dtype=float)
    Y = np.asarray(Y, dtype=float)
    if n_components is None:
        n_components = min(len(X) - 1, X.shape[1])
    X0 = X - X.mean(axis=0, keepdims=True)
    Y0 = Y - Y.mean(axis=0, keepdims=True)
    Cov = X0.T @ Y0
    k = n_components
    B, T, S = X.shape[1], Y.shape[1], X.shape[0]
    x_load, y_load = np.zeros((B, k)), np.zeros((T, k))
    x_sc, y_sc = np.zeros((S, k)), np.zeros((S, k))
    x_w, basis = np.zeros((B, k)), np.zeros((B, k))
    for comp in range(k):
        _, _, ri = _top_triplet(Cov)
        ti = X0 @ ri
        nrm = np.linalg.norm(ti)
        x_w[:, [comp]] = ri / nrm
        ti = ti / nrm
        x_sc[:, [comp]] = ti
        x_load[:, [comp]] = X0.T @ ti
        qi = Y0.T @ ti
        y_load[:, [comp]] = qi
        y_sc[:, [comp]] = Y0 @ qi
        vi = x_load[:, [comp]]
        for _ in range(2):
            for j in range(comp):
                vj = basis[:, [j]]
                vi = vi - (vj.T @ vi) * vj
        vi = vi / np.linalg.norm(vi)
        basis[:, [comp]] = vi
        Cov = Cov - vi @ (vi.T @ Cov)
        Vi = basis[:, :comp]
        Cov = Cov - Vi @ (Vi.T @ Cov)
    y_sc = resid_yscores(x_sc, y_sc)
    pctvar = np.vstack([np.sum(x_load ** 2, axis=0) / np.sum(X0 ** 2),
                        np.sum(y_load ** 2, axis=0) / np.sum(Y0 ** 2)])
    return dict(x_weights=x_w, x_loadings=x_load, y_loadings=y_load,
                x_scores=x_sc, y_scores=y_sc, pctvar=pctvar)


def get_mask(X, Y):
    """Rows where neither X nor Y is entirely NaN (regression.py:48-53)."""
    return np.logical_not(np.logical_or(np.all(np.isnan(X), axis=1),
                                        np.all(np.isnan(Y), axis=1)))


_AGG = dict(mean=np.mean, median=np.median, sum=np.sum)


def regression_single_boot(X, Y, inds, k, original, aggfunc=None):
    """PLSRegression._single_boot, regression.py:279-327.  For 3-D Y ``inds``
    is the pair (subject resample, third-axis resample) and Y is aggregated
    over the resampled third axis (:308-310); NaN rows of the resample are
    masked (:313, :324-325)."""
    if Y.ndim == 3:
        sboot, cboot = inds
        Xi, Yi = X[sboot], aggfunc(Y[..., cboot], axis=-1)[sboot]
    else:
        Xi, Yi = X[inds], Y[inds]
    mask = get_mask(Xi, Yi)
    w = simpls(Xi[mask], Yi[mask], k)['x_weights']
    w = w * np.sign(efficient_corr(w, original))
    return Yi[mask].T @ (Xi @ w)[mask], w


def regression_single_perm(X, Y, inds, k):
    """PLSRegression._single_perm with original=None (the only reachable
    branch), regression.py:329-373."""
    Yp = Y[inds]
    mask = get_mask(X, Yp)
    return simpls(X[mask], Yp[mask], k)['pctvar'][1]


def run_regression(X, Y, n_components, permsamples=None, bootsamples=None,
                   ci=95, aggfunc='mean'):
    """PLSRegression.run_pls, regression.py:375-428, incl. 3-D Y (aggregated
    with ``aggfunc``) and all-NaN rows.  The reference mean-centres the
    caller's X in place (:395); here a copy is centred."""
    X = np.array(X, dtype=float)
    Y = np.array(Y, dtype=float)
    agg = _AGG.get(aggfunc, aggfunc)
    Y_agg = agg(Y, axis=-1) if Y.ndim == 3 else Y.copy()
    X -= np.nanmean(X, axis=0, keepdims=True)
    Y_agg = Y_agg - np.nanmean(Y_agg, axis=0, keepdims=True)
    mask = get_mask(X, Y_agg)
    k = int(n_components)
    out = simpls(X[mask], Y_agg[mask], k)
    res = dict(permres={}, bootres={})
    W = out['x_weights']
    res['x_weights'] = W
    res['x_scores'] = X @ W
    res['varexp'] = out['pctvar'][1]
    res['y_loadings'] = Y_agg[mask].T @ res['x_scores'][mask]
    res['y_scores'] = np.full((len(Y_agg), k), np.nan)
    res['y_scores'][mask] = resid_yscores(res['x_scores'][mask],
                                          Y_agg[mask] @ res['y_loadings'])
    if permsamples is not None:
        d_perm = np.stack([regression_single_perm(X, Y_agg, permsamples[:, i], k)
                           for i in range(permsamples.shape[1])], axis=-1)
        res['permres'] = dict(
            pvals=perm_sig(np.diag(res['varexp']), d_perm),
            perm_singval=d_perm, permsamples=permsamples)
    if bootsamples is not None:
        R = bootsamples.shape[-1]
        u_sum, u_square = np.zeros_like(W), np.zeros_like(W)
        distrib = []
        # the reference bootstraps the ORIGINAL (un-centred, un-aggregated) Y
        # for 3-D input and the centred Y for 2-D input (regression.py:408, 395-397)
        Yb = Y if Y.ndim == 3 else Y_agg
        for i in range(R):
            yl, w = regression_single_boot(X, Yb, bootsamples[..., i], k, W, agg)
            u_sum += w
            u_square += w ** 2
            distrib.append(yl)
        distrib = np.stack(distrib, axis=-1)
        u_sum, u_square = u_sum + W, u_square + W ** 2
        bsr, se = boot_rel(W, u_sum, u_square, R + 1)
        res['bootres'] = dict(
            x_weights_normed=bsr, x_weights_stderr=se,
            y_loadings=res['y_loadings'], y_loadings_boot=distrib,
            y_loadings_ci=np.stack(boot_ci(distrib, ci=ci), -1),
            bootsamples=bootsamples)
    return res
