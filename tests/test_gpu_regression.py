"""SIMPLS regression on the GPU vs the reference goldens (T <= 11: exact) and
the oracle."""
import numpy as np
import pytest

from conftest import load_golden, assert_close
from oracle import cpu_ref as ref

pytestmark = pytest.mark.gpu
RTOL = 1e-5


@pytest.mark.parametrize('tag', ['t4', 't8', 't16'])
def test_pls_regression(tag):
    import pypyls_amd as pls
    g = load_golden('simpls_' + tag)
    k = int(g['n_components'])
    X0 = g['X'].copy()
    res = pls.pls_regression(g['X'], g['Y'], n_components=k, n_perm=g['permsamples'].shape[1],
                             n_boot=g['ref_bootres__bootsamples'].shape[1],
                             permsamples=g['permsamples'],
                             bootsamples=g['ref_bootres__bootsamples'], seed=1234, verbose=False)
    np.testing.assert_array_equal(g['X'], X0)                 # caller's X untouched
    want = ref.run_regression(g['X'], g['Y'], k, permsamples=g['permsamples'],
                              bootsamples=g['ref_bootres__bootsamples'])
    # oracle (exact leading triplet, same convention): tight
    for key in ('x_weights', 'x_scores', 'y_scores', 'y_loadings', 'varexp'):
        assert_close(res[key], want[key], RTOL, what='oracle ' + key)
    assert_close(res['permres']['perm_singval'], want['permres']['perm_singval'], RTOL, what='perm')
    np.testing.assert_array_equal(res['permres']['pvals'], want['permres']['pvals'])
    for key in ('x_weights_normed', 'x_weights_stderr', 'y_loadings_boot', 'y_loadings_ci'):
        assert_close(res['bootres'][key], want['bootres'][key], RTOL, what='oracle ' + key)
    if tag != 't16':
        # T <= 11: the reference's rank-1 randomized SVD is exact
        for key in ('x_weights', 'x_scores', 'y_scores', 'y_loadings', 'varexp'):
            assert_close(res[key], g['ref_' + key], RTOL, what='reference ' + key)
        assert_close(res['permres']['perm_singval'], g['ref_perm_varexp'], RTOL, what='reference perm')
        for key in ('x_weights_normed', 'x_weights_stderr', 'y_loadings_boot', 'y_loadings_ci'):
            assert_close(res['bootres'][key], g['ref_bootres__' + key], RTOL, what='reference ' + key)
    else:
        # T = 16 > 11: the reference is approximate and seed dependent (parity unpinned)
        assert_close(res['varexp'], g['ref_varexp'], 5e-2, what='reference varexp (approximate)')


def test_seed_gives_reference_bootsamples():
    import pypyls_amd as pls
    g = load_golden('simpls_t4')
    res = pls.pls_regression(g['X'], g['Y'], n_components=int(g['n_components']), n_perm=0,
                             n_boot=g['ref_bootres__bootsamples'].shape[1], seed=1234, verbose=False)
    np.testing.assert_array_equal(res['bootres']['bootsamples'], g['ref_bootres__bootsamples'])


@pytest.mark.parametrize('agg', ['mean', 'median'])
def test_pls_regression_3d_y(agg):
    """3-D Y bootstrap (regression.py:208-235, 308-310) vs the reference."""
    import pypyls_amd as pls
    g = load_golden('simpls_3d_' + agg)
    n = g['boot_subjects'].shape[1]
    bs = np.empty((2, n), dtype=object)
    for i in range(n):
        bs[0, i], bs[1, i] = g['boot_subjects'][:, i], g['boot_third'][:, i]
    k = int(g['n_components'])
    res = pls.pls_regression(g['X'], g['Y'], n_components=k, n_perm=0, n_boot=n, aggfunc=agg,
                             bootsamples=bs, seed=1234, verbose=False)
    for key in ('x_weights', 'x_scores', 'y_scores', 'y_loadings', 'varexp'):
        assert_close(res[key], g['ref_' + key], RTOL, what=key)
    for key in ('x_weights_normed', 'x_weights_stderr', 'y_loadings_boot', 'y_loadings_ci'):
        assert_close(res['bootres'][key], g['ref_bootres__' + key], RTOL, what=key)
    # generated resampling arrays: both draws restart from the seed (regression.py:212-215)
    res2 = pls.pls_regression(g['X'], g['Y'], n_components=k, n_perm=0, n_boot=n, aggfunc=agg,
                              seed=1234, verbose=False)
    got = res2['bootres']['bootsamples']
    np.testing.assert_array_equal(np.stack(list(got[0]), -1), g['boot_subjects'])
    np.testing.assert_array_equal(np.stack(list(got[1]), -1), g['boot_third'])
    assert_close(res2['bootres']['x_weights_normed'], g['ref_bootres__x_weights_normed'], RTOL,
                 what='bsr (generated bootsamples)')
    with pytest.raises(ValueError):
        pls.pls_regression(g['X'], g['Y'], n_components=k, n_perm=0, n_boot=n, aggfunc='notafunc')
    with pytest.raises(TypeError):
        pls.pls_regression(g['X'], g['Y'], n_components=k, n_perm=0, n_boot=n, aggfunc=lambda x: x)
    with pytest.raises(ValueError):
        pls.pls_regression(g['X'], g['Y'], n_components=k, n_perm=0, n_boot=n, bootsamples=[[10], [10]])


def test_pls_regression_3d_y_with_nan_rows():
    """3-D Y plus all-NaN rows in X and a missing subject in Y vs the reference
    (regression.py:48-53, 308-313)."""
    import pypyls_amd as pls
    g = load_golden('simpls_3d_nan')
    n = g['boot_subjects'].shape[1]
    bs = np.empty((2, n), dtype=object)
    for i in range(n):
        bs[0, i], bs[1, i] = g['boot_subjects'][:, i], g['boot_third'][:, i]
    res = pls.pls_regression(g['X'], g['Y'], n_components=int(g['n_components']), n_perm=0, n_boot=n,
                             aggfunc='mean', bootsamples=bs, seed=1234, verbose=False)
    for key in ('x_weights', 'x_scores', 'y_scores', 'y_loadings', 'varexp'):
        np.testing.assert_array_equal(np.isnan(res[key]), np.isnan(g['ref_' + key]))
        assert_close(np.nan_to_num(res[key]), np.nan_to_num(g['ref_' + key]), RTOL, what=key)
    for key in ('x_weights_normed', 'x_weights_stderr', 'y_loadings_boot', 'y_loadings_ci'):
        assert_close(res['bootres'][key], g['ref_bootres__' + key], RTOL, what=key)
    Yp = g['Y'].copy()
    Yp[3, 1, 2] = np.nan                                   # a subject missing in SOME slices only
    with pytest.raises(NotImplementedError):
        pls.pls_regression(g['X'], Yp, n_components=2, n_perm=0, n_boot=2, seed=1, verbose=False)
    # n_boot = 0 with a 3-D Y (used to die on an unbound local)
    r0 = pls.pls_regression(g['X'], g['Y'], n_components=2, n_perm=0, n_boot=0, verbose=False)
    assert r0.x_weights.shape == (g['X'].shape[1], 2)


def test_pls_regression_nan_rows():
    """All-NaN rows are masked per resample (get_mask, regression.py:48-53)."""
    import pypyls_amd as pls
    g = load_golden('simpls_nan')
    res = pls.pls_regression(g['X'], g['Y'], n_components=3, n_perm=g['permsamples'].shape[1],
                             n_boot=g['ref_bootres__bootsamples'].shape[1],
                             permsamples=g['permsamples'],
                             bootsamples=g['ref_bootres__bootsamples'], seed=1234, verbose=False)
    for key in ('x_weights', 'x_scores', 'y_scores', 'y_loadings', 'varexp'):
        a, b = res[key], g['ref_' + key]
        np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
        assert_close(np.nan_to_num(a), np.nan_to_num(b), RTOL, what=key)
    assert_close(res['permres']['perm_singval'], g['ref_perm_varexp'], RTOL, what='perm')
    for key in ('x_weights_normed', 'x_weights_stderr', 'y_loadings_boot', 'y_loadings_ci'):
        assert_close(res['bootres'][key], g['ref_bootres__' + key], RTOL, what=key)
    X = g['X'].copy()
    X[3, 2] = np.nan                                  # partly-NaN row
    with pytest.raises(NotImplementedError):
        pls.pls_regression(X, g['Y'], n_components=2, n_perm=0, n_boot=0)


def test_regression_errors():
    import pypyls_amd as pls
    rs = np.random.RandomState(0)
    X, Y = rs.rand(20, 30), rs.rand(20, 3)
    with pytest.raises(ValueError):
        pls.pls_regression(X, Y, n_components=25, n_perm=0, n_boot=0)
    with pytest.raises(ValueError):
        pls.pls_regression(X, Y[:-1], n_components=2, n_perm=0, n_boot=0)
    res = pls.pls_regression(X, Y, n_components=2, n_perm=4, n_boot=4, seed=1, verbose=False)
    assert res.x_weights.shape == (30, 2) and res.varexp.shape == (2,)
    assert 'singvals' not in res or res.get('singvals') is None
    assert res.bootres.y_loadings_boot.shape == (3, 2, 4)
    assert res.permres.perm_singval.shape == (2, 4)


def test_single_pass_bootstrap_equals_two_pass(monkeypatch):
    """The SIMPLS bootstrap aligns its signs in dual space and takes one feature pass whose
    epilogue accumulates the aligned weights (plsx_simpls_boot_batch, k_xprod EPI = 2); the
    two-pass route (weights written, cross-Gram with the original, rotation pass) gives the
    same bootstrap ratios and y_loadings -- also with NaN rows and more bootstraps than one
    solver batch."""
    import pypyls_amd as pls
    rs = np.random.RandomState(11)
    S, B, T, k = 70, 2600, 9, 6
    X = rs.randn(S, B) + rs.rand(1, B)
    Y = rs.randn(S, T) + 0.5 * X[:, :T]
    X[5] = np.nan
    Y[9] = np.nan
    out = {}
    for route in ('single', 'two'):
        if route == 'two':
            monkeypatch.setenv('PLSX_TWO_PASS_BOOT', '1')
        from pypyls_amd.engine import Engine, options_from_env
        out[route] = pls.pls_regression(X, Y, n_components=k, n_perm=10, n_boot=700, seed=99, verbose=False,
                                        _engine=Engine(**options_from_env()))
    monkeypatch.delenv('PLSX_TWO_PASS_BOOT')
    a, b = out['single'], out['two']
    np.testing.assert_array_equal(a.bootres.bootsamples, b.bootres.bootsamples)
    for key in ('x_weights_normed', 'x_weights_stderr', 'y_loadings_boot', 'y_loadings_ci'):
        assert_close(a['bootres'][key], b['bootres'][key], 1e-9, what='single vs two pass ' + key)


@pytest.mark.parametrize('S,B,T,k', [(70, 900, 1, 1), (70, 900, 2, 2), (90, 1500, 5, 4), (120, 2000, 11, 6), (200, 2500, 20, 9)])
def test_leading_eigenpair_solver_equals_jacobi_and_oracle(S, B, T, k, monkeypatch):
    """SIMPLS takes only the leading eigenpair of the T x T matrix H per component: Householder + multisection +
    inverse iteration on one wavefront (round 4, wave_top_eig; needs 8 T + 8 <= S of the wave's scatter buffer)
    against the full one-sided Jacobi solve it replaces (PLSX_SIMPLS_JACOBI -> plsx_set_option) and the oracle's
    exact SIMPLS -- including behaviours that are near copies of each other (clustered eigenvalues of H)."""
    import pypyls_amd as pls
    from pypyls_amd.engine import Engine, options_from_env
    rs = np.random.RandomState(S + T)
    X = rs.randn(S, B)
    Y = rs.randn(S, T) + 0.4 * X[:, :T]
    if T >= 5:
        Y[:, 1] = Y[:, 0] + 1e-7 * rs.randn(S)            # two nearly identical behaviours
    out = {}
    for route in ('top', 'jacobi'):
        monkeypatch.delenv('PLSX_SIMPLS_JACOBI', raising=False)
        if route == 'jacobi':
            monkeypatch.setenv('PLSX_SIMPLS_JACOBI', '1')
        out[route] = pls.pls_regression(X, Y, n_components=k, n_perm=40, n_boot=40, seed=9, verbose=False,
                                        _engine=Engine(**options_from_env()))
    monkeypatch.delenv('PLSX_SIMPLS_JACOBI', raising=False)
    a, b = out['top'], out['jacobi']
    assert_close(a.x_weights, b.x_weights, 1e-9, what='x_weights top vs jacobi')
    assert_close(a.varexp, b.varexp, 1e-10, what='pctvar top vs jacobi')
    assert_close(a.permres.perm_singval, b.permres.perm_singval, 1e-9, what='perm pctvar top vs jacobi')
    assert_close(a.bootres.x_weights_stderr, b.bootres.x_weights_stderr, 1e-7, what='stderr top vs jacobi')
    Xc, Yc = X - X.mean(axis=0), Y - Y.mean(axis=0)
    fit = ref.simpls(Xc, Yc, k)
    sgn = np.sign(np.sum(fit['x_weights'] * a.x_weights, axis=0))
    assert_close(a.x_weights * sgn, fit['x_weights'], 1e-6, what='x_weights vs oracle')
    assert_close(a.varexp, fit['pctvar'][1], 1e-8, what='pctvar vs oracle')


@pytest.mark.parametrize('S,B,T,k', [(60, 1200, 7, 5), (50, 6, 9, 4), (40, 9, 9, 3)])
def test_original_fit_on_the_device_equals_host_rule(S, B, T, k):
    """The regression front-end keeps the original fit on the device (round 5): plsx_svd_flip applies the sign rule of
    compute.svd to data bound for regression (on the x_weights when B > T, else on the right vectors c;
    pyls/types/regression.py:103, compute.py:43-50) and plsx_center_rows forms the column-centred weights of the sign
    alignment.  Same x_weights as the host rule on the arrays plsx_simpls_decompose returns, for B > T, B < T and
    B == T, and the same bootstrap sums whichever way the original was set."""
    from pypyls_amd.engine import Engine
    from pypyls_amd import resampling as rsmp
    rs = np.random.RandomState(S + B)
    X = rs.randn(S, B) + rs.rand(1, B)
    Y = rs.randn(S, T) + 0.4 * X[:, :1]
    Yc = Y - Y.mean(0)
    boots = rsmp.gen_bootsamp([S], 1, 9, seed=3)
    out = []
    for dev_side in (False, True):
        eng = Engine()
        eng.set_data_regression(X, Yc, k)
        if dev_side:
            d_W, pct, yl = eng.simpls_decompose_dev()
            eng.simpls_set_original_dev(d_W)
            W = d_W.cpu().numpy()
        else:
            W, pct, cvec, yl = eng.simpls_decompose()
            lead = W if B > T else cvec
            idx = np.argmax(np.abs(lead), axis=0)
            signs = np.sign(lead[idx, np.arange(k)])
            signs[signs == 0] = 1.0
            W = W * signs
            eng.simpls_set_original(W)
        usum, usq, ylb = eng.simpls_boot(boots)
        out.append((W, usum.cpu().numpy(), usq.cpu().numpy(), np.asarray(ylb)))
    assert np.array_equal(out[0][0], out[1][0])                   # signs only: bit for bit
    for a, b in zip(out[0][1:], out[1][1:]):
        assert_close(a, b, 1e-12, what='original set from the host vs on the device')


@pytest.mark.parametrize('S,B,T,k', [(48, 70, 6, 4), (40, 33, 20, 5), (70, 40, 40, 3)])
def test_large_batches_take_the_three_wave_solver(S, B, T, k):
    """A solver batch of more than 2048 resamples runs the SD_RC = 8 instantiations of k_sd_post0 / k_sd_step /
    k_sd_final (three waves per SIMD, csrc/plsx_simpls.h), a smaller one the SD_RC = 16 ones: ONE call of 2304
    permutations / bootstraps against the same rows in two calls of 1152 (different summation order inside a wave:
    1e-10, not bit-identical), and a sample of both against the oracle (regression.py:279-373)."""
    from pypyls_amd.engine import Engine
    from pypyls_amd import resampling as rsmp
    rs = np.random.RandomState(S * T)
    X = rs.randn(S, B) + rs.rand(1, B)
    Y = rs.randn(S, T) + 0.4 * X[:, :1] + 0.2 * X[:, 1:2]
    Xc, Yc = X - X.mean(0), Y - Y.mean(0)
    n = 2304
    perms = rsmp.gen_permsamp([S], 1, n, seed=5, verbose=False)
    boots = rsmp.gen_bootsamp([S], 1, n, seed=6, verbose=False)
    eng = Engine()
    eng.set_data_regression(Xc, Yc, k)
    W, pct, cvec, yl = eng.simpls_decompose()
    lead = W if B > T else cvec
    signs = np.sign(lead[np.argmax(np.abs(lead), axis=0), np.arange(k)])
    signs[signs == 0] = 1.0
    W = W * signs
    eng.simpls_set_original(W)
    big_p = np.asarray(eng.simpls_perm(perms))
    small_p = np.concatenate([np.asarray(eng.simpls_perm(perms[:, :n // 2])),
                              np.asarray(eng.simpls_perm(perms[:, n // 2:]))], axis=-1)
    assert_close(big_p, small_p, 1e-10, what='perm pctvar: one batch of 2304 vs two of 1152')
    us, uq, ylb = eng.simpls_boot(boots)
    us1, uq1, ylb1 = eng.simpls_boot(boots[:, :n // 2])
    us2, uq2, ylb2 = eng.simpls_boot(boots[:, n // 2:])
    assert_close(us.cpu().numpy(), (us1 + us2).cpu().numpy(), 1e-9, what='bootstrap sum of weights')
    assert_close(uq.cpu().numpy(), (uq1 + uq2).cpu().numpy(), 1e-9, what='bootstrap sum of squared weights')
    assert_close(np.asarray(ylb), np.concatenate([np.asarray(ylb1), np.asarray(ylb2)], axis=-1), 1e-9,
                 what='bootstrap y-loadings')
    pick = [0, 1, 1151, 1152, 2047, 2048, 2303]
    for i in pick:
        want = ref.regression_single_perm(Xc, Yc, perms[:, i], k)
        assert_close(big_p[:, i], want, 1e-8, what='permutation {} vs oracle'.format(i))
        ylw, _ = ref.regression_single_boot(Xc, Yc, boots[:, i], k, W)
        assert_close(np.asarray(ylb)[..., i], ylw, 1e-7, what='bootstrap {} y-loadings vs oracle'.format(i))


@pytest.mark.parametrize('S,B,T,k', [(90, 400, 40, 5), (120, 300, 64, 4), (110, 250, 70, 4), (50, 200, 56, 3), (24, 150, 30, 2),
                                     (40, 120, 44, 3)])
def test_wide_y_takes_the_solver_instantiations_of_its_class(S, B, T, k):
    """The solver kernels are instantiated per class of T (csrc/plsx_simpls.h: T <= 32, T <= 64, larger; the
    leading-eigenpair solver where T <= 64 and T <= S, the one-sided Jacobi solve otherwise): 32 < T <= 64, T > 64 and
    T > S against the oracle's exact SIMPLS (regression.py:56-186, 279-373)."""
    import pypyls_amd as pls
    rs = np.random.RandomState(S + T)
    X = rs.randn(S, B) + rs.rand(1, B)
    Y = rs.randn(S, T)
    Y[:, :8] += 0.5 * X[:, :8]
    res = pls.pls_regression(X, Y, n_components=k, n_perm=12, n_boot=10, seed=21, verbose=False)
    want = ref.run_regression(X, Y, k, permsamples=res.permres.permsamples, bootsamples=res.bootres.bootsamples)
    for key in ('x_weights', 'x_scores', 'y_scores', 'y_loadings', 'varexp'):
        assert_close(res[key], want[key], 1e-6, what='oracle ' + key)
    assert_close(res['permres']['perm_singval'], want['permres']['perm_singval'], 1e-6, what='perm')
    for key in ('x_weights_normed', 'x_weights_stderr', 'y_loadings_boot'):
        assert_close(res['bootres'][key], want['bootres'][key], 1e-5, what='oracle ' + key)


@pytest.mark.parametrize('S,B,T,k', [(90, 400, 40, 5), (120, 300, 64, 4), (40, 120, 44, 3), (50, 200, 56, 3), (70, 260, 33, 3)])
def test_jacobi_eigen_solve_of_the_middle_class_of_t(S, B, T, k):
    """32 < T <= 64 with the one-sided Jacobi eigen-solve (k_sd_step<1, true, 16>: wave_jacobi_cols<16>, 16 rows per
    lane) -- taken when T > S (rank-deficient H: the shapes on which round 5 saw a wrong leading vector, T = 44 / 56)
    and, forced with the ``simpls_jacobi`` option, on full-rank problems -- against the oracle's exact SIMPLS
    (regression.py:56-186) and against the leading-eigenpair solver where that one applies.  tools/jacobi16_probe.py
    is the wider version (seven code-generation variants, bit-identical to the 36-row instantiation)."""
    from pypyls_amd.engine import Engine
    rs = np.random.RandomState(S + T)
    X = rs.randn(S, B) + rs.rand(1, B)
    Y = rs.randn(S, T)
    Y[:, :8] += 0.5 * X[:, :8]
    Xc, Yc = X - X.mean(0), Y - Y.mean(0)
    fit = ref.simpls(Xc, Yc, k)
    got = {}
    for forced in ((1,) if T > S else (1, 0)):
        eng = Engine(options={'simpls_jacobi': forced})
        try:
            eng.set_data_regression(Xc, Yc, k)
            W, pct, cvec, yl = eng.simpls_decompose()
        finally:
            eng.close()
        sg = np.sign(np.sum(W * fit['x_weights'], axis=0))
        assert_close(W * sg, fit['x_weights'], 1e-9, what='x_weights (jacobi forced = {})'.format(forced))
        assert_close(pct, fit['pctvar'][1], 1e-10, what='pctvar (jacobi forced = {})'.format(forced))
        got[forced] = W * sg
    if 0 in got:
        assert_close(got[1], got[0], 1e-10, what='Jacobi vs leading-eigenpair solver')
