"""The reference's integration matrix (pyls/tests/types/test_svd.py:13-143):
{1, 3 groups} x {1, 2, 4 conditions} x {n_split None / 5} x {rotate} x
{mean_centering}; attribute presence and shapes, plus a parity spot-check of
every combination against the oracle."""
import warnings

import numpy as np
import pytest

from conftest import assert_close
from oracle import cpu_ref as ref

pytestmark = pytest.mark.gpu

S, B, T = 120, 300, 8
rs = np.random.RandomState(1234)
X = rs.rand(S, B)
Y = rs.rand(S, T) + 0.5 * X[:, :T]


def _groups(n_groups, n_cond):
    return [S // n_cond // n_groups] * n_groups


@pytest.mark.parametrize('n_groups', [1, 3])
@pytest.mark.parametrize('n_cond', [1, 2, 4])
@pytest.mark.parametrize('n_split', [None, 5])
@pytest.mark.parametrize('rotate', [True, False])
def test_behavioral_matrix(n_groups, n_cond, n_split, rotate):
    import pypyls_amd as pls
    groups = _groups(n_groups, n_cond)
    res = pls.behavioral_pls(X, Y, groups=groups, n_cond=n_cond, n_perm=6, n_boot=5,
                             n_split=n_split or 0, test_split=3, test_size=0.25, rotate=rotate,
                             seed=1234, verbose=False)
    J = n_groups * n_cond
    L = min(J * T, B)
    for attr, shape in [('x_weights', (B, L)), ('y_weights', (J * T, L)), ('singvals', (L,)),
                        ('varexp', (L,)), ('x_scores', (S, L)), ('y_scores', (S, L)),
                        ('y_loadings', (J * T, L))]:
        assert res[attr].shape == shape, attr
    assert res.permres.pvals.shape == (L,) and res.permres.perm_singval.shape == (L, 6)
    assert res.bootres.x_weights_normed.shape == (B, L)
    assert res.bootres.y_loadings_boot.shape == (J * T, L, 5)
    assert res.cvres.pearson_r.shape == (T, 3) and res.cvres.r_squared.shape == (T, 3)
    if n_split:
        for k in ('ucorr', 'vcorr', 'ucorr_pvals', 'vcorr_pvals', 'ucorr_lolim', 'vcorr_uplim'):
            assert res.splitres[k].shape == (L,), k
    # parity spot-check (no split-half here: masks are drawn inside the front-end)
    want = ref.run_plsc(X, Y, method='behavioral', groups=groups, n_cond=n_cond, rotate=rotate,
                        permsamples=res.permres.permsamples, bootsamples=res.bootres.bootsamples)
    assert_close(res.singvals, want['singvals'], 1e-6, what='singvals')
    assert_close(res.permres.perm_singval, want['permres']['perm_singval'], 1e-6, what='perm')
    full_rank = all(len(np.unique(res.bootres.bootsamples[c, i])) - 1 >= T
                    for i in range(5) for c in ref.dummy_code(groups, n_cond).T.astype(bool))
    if full_rank:
        assert_close(res.bootres.x_weights_normed, want['bootres']['x_weights_normed'], 1e-5, what='bsr')


@pytest.mark.parametrize('n_groups,n_cond', [(1, 2), (1, 4), (3, 1), (3, 2), (3, 4)])
@pytest.mark.parametrize('mean_centering', [0, 1, 2])
@pytest.mark.parametrize('n_split', [None, 5])
def test_meancentered_matrix(n_groups, n_cond, mean_centering, n_split):
    import pypyls_amd as pls
    groups = _groups(n_groups, n_cond)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        res = pls.meancentered_pls(X, groups=groups, n_cond=n_cond, mean_centering=mean_centering,
                                   n_perm=6, n_boot=5, n_split=n_split or 0, seed=1234, verbose=False)
    J = n_groups * n_cond
    L = min(J, B)
    for attr, shape in [('x_weights', (B, L)), ('y_weights', (J, L)), ('singvals', (L,)),
                        ('x_scores', (S, L)), ('y_scores', (S, L))]:
        assert res[attr].shape == shape, attr
    assert res.bootres.contrast.shape == (J, L)
    assert res.bootres.contrast_boot.shape == (J, L, 5)
    assert res.bootres.contrast_ci.shape == (J, L, 2)
    mc = res.inputs.mean_centering
    want = ref.run_plsc(X, method='meancentered', groups=groups, n_cond=n_cond, mean_centering=mc,
                        permsamples=res.permres.permsamples)
    live = ref.live_lvs(want['singvals'])
    assert_close(res.singvals[live], want['singvals'][live], 1e-6, what='singvals')
    assert_close(res.permres.perm_singval[live], want['permres']['perm_singval'][live], 1e-6, what='perm')


def test_warnings_and_errors():
    """pyls/tests/types/test_svd.py:128-143."""
    import pypyls_amd as pls
    with pytest.warns(UserWarning):
        pls.meancentered_pls(X, groups=[40, 40, 40], mean_centering=0, n_perm=0, n_boot=0)
    with pytest.warns(UserWarning):
        pls.meancentered_pls(X, n_cond=2, mean_centering=1, n_perm=0, n_boot=0)
    with pytest.raises(ValueError):
        pls.meancentered_pls(X, groups=[60, 60], mean_centering=3, n_perm=0, n_boot=0)
    with pytest.raises(ValueError):
        pls.meancentered_pls(X, groups=[S], n_perm=0, n_boot=0)
    with pytest.raises(ValueError):
        pls.meancentered_pls(X, n_cond=7, n_perm=0, n_boot=0)        # 120 % 7 != 0
    with pytest.raises(ValueError):
        pls.behavioral_pls(X, Y[:-1], n_perm=0, n_boot=0, test_split=0)
    # scalar `groups` is accepted (BasePLS.__init__, pyls/base.py:258-259)
    res = pls.behavioral_pls(X, Y, groups=S, n_perm=0, n_boot=0, test_split=0)
    assert res.singvals.shape == (T,)


# ---- pyls/tests/types/test_regression.py -------------------------------------
RS, RB, RT = 50, 1000, 100
Xr = rs.rand(RS, RB)
Yr = rs.rand(RS, RT)


def _reg(n_components=None, X=Xr, Y=Yr, **kw):
    import pypyls_amd as pls
    kw.setdefault('n_perm', 20)
    kw.setdefault('n_boot', 10)
    res = pls.pls_regression(X, Y, n_components=n_components, ci=95, seed=1234, verbose=False, **kw)
    k = RS - 1 if n_components is None else n_components
    for attr, shape in [('x_weights', (RB, k)), ('x_scores', (RS, k)), ('y_scores', (RS, k)),
                        ('y_loadings', (RT, k)), ('varexp', (k,))]:
        assert res[attr].shape == shape, attr
    return res


@pytest.mark.parametrize('n_components', [None, 2, 5, 10, 15])
def test_regression_components(n_components):
    """100 Y columns, up to S-1 = 49 components (test_regression.py:59-63), with
    parity against the oracle on the leading components."""
    few = dict(n_perm=4, n_boot=3) if n_components is None else {}    # keeps the oracle leg short
    res = _reg(n_components, **few)
    k = RS - 1 if n_components is None else n_components
    want = ref.run_regression(Xr, Yr, k, permsamples=res.permres.permsamples,
                              bootsamples=res.bootres.bootsamples)
    lead = min(k, 10)               # trailing components of a 49-step deflation amplify rounding
    assert_close(res.varexp[:lead], want['varexp'][:lead], 1e-6, what='varexp')
    assert_close(res.x_weights[:, :lead], want['x_weights'][:, :lead], 1e-5, what='x_weights')
    assert_close(res.y_loadings[:, :lead], want['y_loadings'][:, :lead], 1e-5, what='y_loadings')
    assert_close(res.permres.perm_singval[:lead], want['permres']['perm_singval'][:lead], 1e-6, what='perm')
    if k <= 15:
        assert_close(res.varexp, want['varexp'], 1e-5, what='varexp all')
        assert_close(res.permres.pvals, want['permres']['pvals'], 0, what='pvals')
        assert_close(res.bootres.x_weights_normed, want['bootres']['x_weights_normed'], 1e-5, what='bsr')


@pytest.mark.parametrize('aggfunc', ['mean', 'median', 'sum'])
def test_regression_3d(aggfunc):
    Y3 = np.random.RandomState(5).rand(RS, RT, 20)
    res = _reg(2, Y=Y3, aggfunc=aggfunc)
    assert res.bootres.y_loadings_boot.shape == (RT, 2, 10)
    from pypyls_amd import resampling
    sboot = resampling.gen_bootsamp([RS], 1, n_boot=10, seed=1, verbose=False)
    nboot = resampling.gen_bootsamp([20], 1, n_boot=10, seed=2, verbose=False)
    packed = np.empty((2, 10), dtype=object)
    for i in range(10):
        packed[0, i], packed[1, i] = sboot[:, i], nboot[:, i]
    _reg(2, Y=Y3, aggfunc=aggfunc, bootsamples=packed)


def test_regression_missing():
    Xn = Xr.copy()
    Xn[10] = np.nan
    _reg(2, X=Xn)
    Xn[20] = np.nan
    _reg(2, X=Xn)
    Yn = Yr.copy()
    Yn[11] = np.nan
    res = _reg(2, X=Xn, Y=Yn)
    # x_scores = X @ x_weights (base.py:364): NaN only where X is; y_scores are NaN on every masked row
    assert np.isnan(res.x_scores[[10, 20]]).all() and np.isfinite(res.x_scores[11]).all()
    assert np.isnan(res.y_scores[[10, 11, 20]]).all()


def test_regression_errors():
    with pytest.raises(ValueError):
        _reg(1000)
    with pytest.raises(ValueError):
        _reg(Y=rs.rand(RS - 1, RT))
    with pytest.raises(ValueError):
        _reg(Y=rs.rand(RS, RT, 10), aggfunc='notafunc')
    with pytest.raises(TypeError):
        _reg(Y=rs.rand(RS, RT, 10), aggfunc=lambda x: x)
    with pytest.raises(ValueError):
        _reg(Y=rs.rand(RS, RT, 10), bootsamples=[[10], [10]])


# ---- the reference's integration shapes at their real size -------------------------
@pytest.mark.parametrize('groups,n_cond', [([100], 1), ([33, 34, 33], 1), ([25], 4), ([25, 25], 2)])
@pytest.mark.parametrize('n_split', [None, 5])
def test_behavioral_reference_shapes(groups, n_cond, n_split):
    """pyls/tests/types/test_svd.py:7-9,62-96 verbatim: 100 subjects, 1000 features,
    100 behaviours -> T' = 100 / 300 / 400 / 400 (the last two need the sliced
    cross-product layout), n_perm 20, n_boot 10; attribute shapes as the reference
    asserts them plus parity of the live LVs against the oracle."""
    import pypyls_amd as pls
    r2 = np.random.RandomState(1234)
    Xr, Yr = r2.rand(100, 1000), r2.rand(100, 100)
    res = pls.behavioral_pls(Xr, Yr, groups=groups, n_cond=n_cond, n_perm=20, n_boot=10,
                             n_split=n_split or 0, test_split=0, seed=1234, verbose=False)
    J = len(groups) * n_cond
    L = min(1000, 100 * J)
    for attr, shape in [('x_weights', (1000, L)), ('y_weights', (100 * J, L)), ('singvals', (L,)),
                        ('varexp', (L,)), ('x_scores', (100, L)), ('y_scores', (100, L))]:
        assert res[attr].shape == shape, attr
    want = ref.run_plsc(Xr, Yr, method='behavioral', groups=groups, n_cond=n_cond,
                        permsamples=res.permres.permsamples)
    live = want['singvals'] > 1e-6 * want['singvals'][0]
    assert_close(res.singvals[live], want['singvals'][live], 1e-6, what='singvals')
    assert_close(res.permres.perm_singval[live], want['permres']['perm_singval'][live], 1e-5, what='perm')
    assert np.all(np.isfinite(res.bootres.x_weights_normed[:, live]))


# ---- wide behaviour matrices (the reference's tests use 100 Y columns) ----------
@pytest.mark.parametrize('n_groups,n_cond,n_split', [(1, 1, 4), (1, 2, None), (2, 1, 3)])
def test_behavioral_wide_y(n_groups, n_cond, n_split):
    """T = 100 behaviours, T' = 100 / 200 stacked rows: Householder + QL small
    solver, chunked L tiles; parity against the oracle incl. split-half and
    cross-validation."""
    import pypyls_amd as pls
    Sw, Bw, Tw = 320, 420, 100          # training cells keep >= T rows: full-rank train decompositions
    r2 = np.random.RandomState(77)
    Xw = r2.randn(Sw, Bw)
    Yw = r2.randn(Sw, Tw) + 0.4 * Xw[:, :Tw]
    groups = [Sw // n_cond // n_groups] * n_groups
    from pypyls_amd import resampling as rsmp
    kw = {}
    n_perm = 5
    if n_split:
        kw['_splitsamples'] = rsmp.gen_splits(groups, n_cond, n_split, seed=1)
        kw['_perm_splitsamples'] = np.stack([rsmp.gen_splits(groups, n_cond, n_split, seed=10 + i)
                                             for i in range(n_perm)])
    cvs = rsmp.gen_splits(groups, n_cond, 2, seed=3, test_size=0.25)
    kw['_cvsplits'] = cvs
    res = pls.behavioral_pls(Xw, Yw, groups=groups, n_cond=n_cond, n_perm=n_perm, n_boot=4,
                             n_split=n_split or 0, test_split=2, test_size=0.25, seed=4321, verbose=False, **kw)
    J = n_groups * n_cond
    assert res.singvals.shape == (J * Tw,)
    assert res.bootres.x_weights_normed.shape == (Bw, J * Tw)
    assert res.cvres.pearson_r.shape == (Tw, 2)
    want = ref.run_plsc(Xw, Yw, method='behavioral', groups=groups, n_cond=n_cond,
                        permsamples=res.permres.permsamples, bootsamples=res.bootres.bootsamples,
                        splitsamples=kw.get('_splitsamples'), perm_splitsamples=kw.get('_perm_splitsamples'))
    spec = ref.Spec('behavioral', groups, n_cond)
    cv_r, cv_r2 = ref.crossval(spec, Xw, Yw, cvs)
    assert_close(res.singvals, want['singvals'], 1e-6, what='singvals')
    assert_close(res.permres.perm_singval, want['permres']['perm_singval'], 1e-6, what='perm')
    assert_close(res.permres.pvals, want['permres']['pvals'], 0, what='pvals')
    lead = slice(0, 20)
    assert_close(res.bootres.x_weights_normed[:, lead], want['bootres']['x_weights_normed'][:, lead], 1e-5, what='bsr')
    assert_close(res.cvres.pearson_r, cv_r, 1e-6, what='cv r')
    assert_close(res.cvres.r_squared, cv_r2, 1e-6, what='cv r2')
    if n_split:
        assert_close(res.splitres.ucorr, want['splitres']['ucorr'], 1e-6, what='ucorr')
        assert_close(res.splitres.vcorr, want['splitres']['vcorr'], 1e-6, what='vcorr')
        assert_close(res.splitres.ucorr_pvals, want['splitres']['ucorr_pvals'], 0, what='ucorr p')
