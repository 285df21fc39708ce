"""The quadratic-form route of the bootstrap sums (include/plsx.h, plsx_boot_begin / plsx_boot_finish):
where the rotated bootstrap weights are linear in the bound, unscaled feature matrix (U_b = Xc^T V_b) the
library accumulates C_l = sum_b v_bl v_bl^T per batch and passes the features once per series,
    sum_b U_b = Xc^T sum_b V_b,    sum_b U_b[j,l]^2 = x_j^T C_l x_j.
Same sums as the per-bootstrap pass, the two-pass route and the oracle (BasePLS.bootstrap,
pyls/base.py:490-511; compute.boot_rel)."""
import numpy as np
import pytest

from conftest import assert_close, assert_close_per_lv
from oracle import cpu_ref as ref

pytestmark = pytest.mark.gpu


def _engine(**opts):
    from pypyls_amd.engine import Engine, options_from_env
    kw = options_from_env()
    kw.setdefault('options', {}).update(opts)
    return Engine(**kw)


def _case(case, rs):
    from pypyls_amd import resampling as rsmp
    if case.startswith('mc'):
        groups, n_cond, mc = [14, 11, 12], 3, int(case[2])
        S, B = sum(groups) * n_cond, 1700
        cells = rsmp.cell_of_row(groups, n_cond)
        X = rs.randn(S, B) + 0.5 * rs.randn(len(groups) * n_cond, B)[cells]
        return X, None, groups, n_cond, mc, 1, ref.Spec('meancentered', groups, n_cond, False, mc)
    groups, n_cond = ([60], 1) if case == 'cov' else ([21, 24], 2)
    S, B, T = sum(groups) * n_cond, 1500, 6
    X = rs.randn(S, B) * (1.0 + rs.rand(1, B))
    Y = rs.randn(S, T) * np.array([1, 2, .5, 3, 1, .7]) + 0.5 * X[:, :T]
    return X, Y, groups, n_cond, 0, 0, ref.Spec('behavioral', groups, n_cond, True, 0)


@pytest.mark.parametrize('case', ['mc0', 'mc1', 'mc2', 'cov', 'cov_2g2c'])
def test_quadratic_form_sums_equal_per_bootstrap_pass_and_oracle(case):
    from pypyls_amd import resampling as rsmp
    rs = np.random.RandomState(23)
    X, Y, groups, n_cond, mc, method, spec = _case(case, rs)
    cells = rsmp.cell_of_row(groups, n_cond)
    Yo = Y if Y is not None else spec.dummy.astype(float)
    n = 150
    boots = rsmp.gen_bootsamp(groups, n_cond, n, seed=5)
    U, d, V = ref.decompose(spec, X, Yo)
    live = ref.live_lvs(d)
    out = {}
    for route, opt in (('quad', 1), ('direct', -1)):
        eng = _engine(quad_sums=opt)
        eng.set_data(X, Y, cells, len(groups), n_cond, method, mean_centering=mc, covariance=(method == 0))
        eng.set_original(U, np.diag(d), V)
        assert eng.boot_begin(n) == (1 if route == 'quad' else 0)
        eng.boot_finish(eng._zeros((eng.B, eng.L)), eng._zeros((eng.B, eng.L)))      # (an empty series is legal)
        usum, usq, dist = eng.boot(boots)
        out[route] = (usum.cpu().numpy(), usq.cpu().numpy(), dist)
        if route == 'quad':
            # the same series in uneven chunks, accumulated on top of a non-zero start
            u2, q2 = eng._zeros((eng.B, eng.L)) + 1.5, eng._zeros((eng.B, eng.L)) + 2.5
            dd = eng._zeros((n, eng.Tp, eng.L))
            assert eng.boot_begin(n) == 1
            for a, b in ((0, 7), (7, 64), (64, 65), (65, n)):
                eng.boot_into(eng.index_tensor(boots[:, a:b]), u2, q2, dd[a:b])
            assert float(u2.max()) == 1.5 and float(q2.min()) == 2.5          # untouched until the series is closed
            eng.boot_finish(u2, q2)
            eng.sync()
            assert_close(u2.cpu().numpy()[:, live] - 1.5, out['quad'][0][:, live], 1e-11, what='chunked series, sum U')
            assert_close(q2.cpu().numpy()[:, live] - 2.5, out['quad'][1][:, live], 1e-11, what='chunked series, sum U^2')
            assert_close(dd.cpu().numpy().transpose(1, 2, 0)[:, live], dist[:, live], 1e-12, what='chunked series, distrib')
    for a, b, what in zip(out['quad'], out['direct'], ('sum U', 'sum U^2', 'distrib')):
        assert_close_per_lv(a, b, 1, 1e-10, what=what + ': quadratic form vs per-bootstrap pass', keep=live)
    ws, wq = np.zeros_like(U), np.zeros_like(U)
    for i in range(n):
        _, ub = ref.single_boot(spec, X, Yo, boots[:, i], U, d)
        ws += ub
        wq += ub ** 2
    assert_close_per_lv(out['quad'][0], ws, 1, 1e-8, what='quadratic form sum U vs oracle', keep=live)
    assert_close_per_lv(out['quad'][1], wq, 1, 1e-8, what='quadratic form sum U^2 vs oracle', keep=live)
    # what the sums are for: bootstrap ratios / standard errors (compute.boot_rel), element by element
    u0 = U @ d
    for add in (0, 1):
        se_o = np.sqrt(np.abs(wq + add * u0 ** 2 - (ws + add * u0) ** 2 / (n + add)) / (n + add - 1))
        se_q = np.sqrt(np.abs(out['quad'][1] + add * u0 ** 2 - (out['quad'][0] + add * u0) ** 2 / (n + add)) / (n + add - 1))
        np.testing.assert_allclose(se_q[:, live], se_o[:, live], rtol=1e-7)


def test_route_choice():
    """Auto: the closing pass multiplies about (1 + 1 / blocks) / 2 of the S (padded) rows per LV, a bootstrap L rows -- the series has to be
    longer than that to win; correlation-mode behavioral PLS re-scales the features per bootstrap and never
    qualifies; an engine without plsx_boot_begin never leaves the in-place route."""
    from pypyls_amd import resampling as rsmp
    rs = np.random.RandomState(3)
    X, Y, groups, n_cond, mc, method, spec = _case('mc0', rs)
    cells = rsmp.cell_of_row(groups, n_cond)
    U, d, V = ref.decompose(spec, X, spec.dummy.astype(float))
    eng = _engine(quad_sums=0)                                            # (the library's own choice, whatever the environment says)
    eng.set_data(X, None, cells, len(groups), n_cond, 1, mean_centering=0)
    eng.set_original(U, np.diag(d), V)
    assert eng.boot_begin(100) == 0 and eng.boot_begin(222) == 0 and eng.boot_begin(223) == 1          # S = 111 (7 tiles, one block), B = 1700
    eng.set_original(U, np.diag(d), V)                                    # ends the open series
    assert eng.lib.plsx_boot_route(eng.ctx) == 0
    few = _engine(quad_sums=0)                                            # few features: the S x S moments cost more than they save
    few.set_data(X[:, :120], None, cells, len(groups), n_cond, 1, mean_centering=0)
    Uf, df, Vf = ref.decompose(spec, X[:, :120], spec.dummy.astype(float))
    few.set_original(Uf, np.diag(df), Vf)
    assert few.boot_begin(100000) == 0
    Xb = rs.randn(40, 500)
    Yb = rs.randn(40, 4)
    specb = ref.Spec('behavioral', [40], 1, False, 0)
    Ub, db, Vb = ref.decompose(specb, Xb, Yb)
    engb = _engine(quad_sums=1)
    engb.set_data(Xb, Yb, rsmp.cell_of_row([40], 1), 1, 1, 0)
    engb.set_original(Ub, np.diag(db), Vb)
    assert engb.boot_begin(100000) == 0


@pytest.mark.parametrize('masked', [False, True])
def test_regression_front_end_quadratic_form_equals_per_bootstrap_pass(masked):
    import pypyls_amd as pls
    rs = np.random.RandomState(11)
    S, B, T, k = 70, 2100, 7, 5
    X = rs.randn(S, B) + rs.rand(1, B)
    Y = rs.randn(S, T) + 0.5 * X[:, :T]
    if masked:
        X[5] = np.nan
        Y[9] = np.nan
    out = {}
    for route, opt in (('quad', 1), ('direct', -1)):
        out[route] = pls.pls_regression(X, Y, n_components=k, n_perm=6, n_boot=600, seed=99, verbose=False,
                                        _engine=_engine(quad_sums=opt))
    a, b = out['quad'], out['direct']
    np.testing.assert_array_equal(a.bootres.bootsamples, b.bootres.bootsamples)
    for key in ('x_weights_normed', 'x_weights_stderr', 'y_loadings_boot', 'y_loadings_ci'):
        assert_close_per_lv(a['bootres'][key], b['bootres'][key], 1, 1e-9, what='quadratic form vs direct: ' + key)


def test_plsc_front_end_takes_the_route_by_itself():
    """meancentered_pls with n_boot above the threshold announces its series (plsc.py) and the library takes the
    route; same PLSResults as with the route switched off."""
    import pypyls_amd as pls
    from pypyls_amd import engine
    from pypyls_amd import resampling as rsmp
    rs = np.random.RandomState(31)
    groups, n_cond = [12, 13], 2
    S, B = sum(groups) * n_cond, 900
    X = rs.randn(S, B) + 0.7 * rs.randn(len(groups) * n_cond, B)[rsmp.cell_of_row(groups, n_cond)]
    kw = dict(groups=groups, n_cond=n_cond, n_perm=20, n_boot=1200, seed=4, verbose=False)
    eng = engine.default_engine()
    res = {}
    try:
        for route, opt in (('auto', 0), ('direct', -1)):
            eng.set_option('quad_sums', opt)
            res[route] = pls.meancentered_pls(X, **kw)
    finally:
        eng.set_option('quad_sums', 0)
    live = ref.live_lvs(res['direct'].singvals)
    a, b = res['auto'].bootres, res['direct'].bootres
    assert_close_per_lv(a.x_weights_normed, b.x_weights_normed, 1, 1e-8, what='bootstrap ratios', keep=live)
    assert_close_per_lv(a.x_weights_stderr, b.x_weights_stderr, 1, 1e-8, what='standard errors', keep=live)
    assert_close(a.contrast_ci, b.contrast_ci, 1e-10, what='contrast CIs')
    assert np.max(np.abs(a.x_weights_normed - b.x_weights_normed)[:, live]) > 0.0       # (two routes did run)


@pytest.fixture
def forced_route():
    """The front-ends' cached engine with the route forced on: the goldens' bootstrap series (20 - 100) are shorter
    than what the library would pick it for."""
    from pypyls_amd import engine
    eng = engine.default_engine()
    eng.set_option('quad_sums', 1)
    try:
        yield eng
    finally:
        eng.set_option('quad_sums', 0)


_UNSCALED = ['mpls_1g3c_norot', 'mpls_2g2c_split', 'mpls_3g1c', 'mpls_3g2c_mc0', 'mpls_3g2c_mc1', 'mpls_3g2c_mc2',
             'mat_mpls_multigroup_onecond_nosplit', 'mat_mpls_multigroup_onecond_split',
             'bpls_1g1c_cov', 'bpls_2g2c_cov', 'bpls_cv_cov']


@pytest.mark.parametrize('name', _UNSCALED)
def test_reference_goldens_on_the_route(name, forced_route):
    """Every golden of an unscaled mode (the reference's own outputs and the oracle, key by key) with the bootstrap
    sums taken as quadratic forms."""
    import test_gpu_frontend as tf
    if name in tf.MEANC:
        tf.test_meancentered_vs_reference_and_oracle(name)
    else:
        tf.test_behavioral_vs_reference_and_oracle(name)
    assert forced_route.boot_begin(10) == 1           # (the route is what ran: it applies to this binding)
    forced_route.boot_finish(forced_route._zeros((forced_route.B, forced_route.L)),
                             forced_route._zeros((forced_route.B, forced_route.L)))


@pytest.mark.parametrize('which', ['t4', 't8', 't16', '3d_mean', '3d_median', '3d_nan', 'nan'])
def test_regression_goldens_on_the_route(which, forced_route):
    import test_gpu_regression as tr
    if which.startswith('t'):
        tr.test_pls_regression(which)
    elif which == '3d_nan':
        tr.test_pls_regression_3d_y_with_nan_rows()
    elif which == 'nan':
        tr.test_pls_regression_nan_rows()
    else:
        tr.test_pls_regression_3d_y(which[3:])
    assert forced_route.boot_begin(10) == 1
    forced_route.boot_finish(forced_route._zeros((forced_route.B, forced_route.k)),
                             forced_route._zeros((forced_route.B, forced_route.k)))


def test_full_size_c3_route_vs_oracle_and_per_bootstrap_pass():
    """c3 shape (200 x 50 000, 8 cells): 10 bootstraps on the forced route against the oracle, then the series the
    front-end would run (2000 here) on the route the library picks against the per-bootstrap pass."""
    from pypyls_amd import resampling as rsmp
    groups, n_cond = [25, 25, 25, 25], 2
    S, B = 200, 50000
    rs = np.random.RandomState(0)
    cells = rsmp.cell_of_row(groups, n_cond)
    X = rs.randn(S, B) + 0.25 * rs.randn(8, B)[cells]
    spec = ref.Spec('meancentered', groups, n_cond, False, 0)
    Y = spec.dummy.astype(float)
    U, d, V = ref.decompose(spec, X, Y)
    live = ref.live_lvs(d)
    eng = _engine(quad_sums=1)
    eng.set_data(X, None, cells, len(groups), n_cond, 1, mean_centering=0)
    eng.set_original(U, np.diag(d), V)
    n = 10
    boots = rsmp.gen_bootsamp(groups, n_cond, n, seed=1235)
    usum, usq, dist = eng.boot(boots)
    ws, wq = np.zeros_like(U), np.zeros_like(U)
    for i in range(n):
        _, ub = ref.single_boot(spec, X, Y, boots[:, i], U, d)
        ws += ub
        wq += ub ** 2
    assert_close_per_lv(usum.cpu().numpy(), ws, 1, 1e-7, what='c3 sum U on the route vs oracle', keep=live)
    assert_close_per_lv(usq.cpu().numpy(), wq, 1, 1e-7, what='c3 sum U^2 on the route vs oracle', keep=live)
    many = rsmp.gen_bootsamp(groups, n_cond, 2000, seed=99)
    out = {}
    for route, opt in (('auto', 0), ('direct', -1)):
        eng.set_option('quad_sums', opt)
        assert eng.boot_begin(2000) == (1 if route == 'auto' else 0)
        eng.boot_finish(usum, usq)
        u, q, _ = eng.boot(many)
        out[route] = (u.cpu().numpy(), q.cpu().numpy())
    for a, b, what in zip(out['auto'], out['direct'], ('sum U', 'sum U^2')):
        assert_close_per_lv(a, b, 1, 1e-10, what='c3, 2000 bootstraps, ' + what, keep=live)


def test_full_size_c5_route_vs_per_bootstrap_pass():
    """c5 shape (1000 x 100 000, k = 15): 1200 bootstraps, the route the library picks for that series against the
    per-bootstrap pass (which test_config_c5_regression_1000x100000 pins on the oracle): sums and standard errors."""
    from pypyls_amd import resampling as rsmp
    S, B, T, k = 1000, 100000, 20, 15
    rs = np.random.RandomState(2)
    X = rs.randn(S, B)
    Y = rs.randn(S, T) + 0.3 * X[:, :T]
    Xc, Yc = X - X.mean(0), Y - Y.mean(0)
    boots = rsmp.gen_bootsamp([S], 1, 1200, seed=7, verbose=False)
    out = {}
    for route, opt in (('auto', 0), ('direct', -1)):
        eng = _engine(quad_sums=opt)
        eng.set_data_regression(Xc, Yc, k)
        W, pct, cvec, _ = eng.simpls_decompose()
        W = W * np.sign(W[np.argmax(np.abs(W), axis=0), np.arange(k)])
        eng.simpls_set_original(W)
        assert eng.boot_begin(1200) == (1 if route == 'auto' else 0)
        u, q, yl = eng.simpls_boot(boots)
        out[route] = (u.cpu().numpy(), q.cpu().numpy(), yl)
        del eng
    assert_close_per_lv(out['auto'][0], out['direct'][0], 1, 1e-10, what='c5 sum of weights')
    assert_close_per_lv(out['auto'][1], out['direct'][1], 1, 1e-10, what='c5 sum of squared weights')
    assert_close(out['auto'][2], out['direct'][2], 1e-12, what='c5 y_loadings')
    n = 1200
    se = [np.sqrt(np.abs(q - u ** 2 / n) / (n - 1)) for u, q, _ in (out['auto'], out['direct'])]
    np.testing.assert_allclose(se[0], se[1], rtol=1e-8)
