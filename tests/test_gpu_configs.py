"""Parity at the LITERAL sizes of BASELINE.json's configs (c2, c3, c4 split-half,
c5): the HIP path through the C ABI against the oracle on a sample of
resamples.  The oracle needs seconds per resample at these sizes, hence the
small samples; tolerances are the north-star's 1e-5 relative or tighter.

    c2  behavioral   X(80 x 10000)    Y(80 x 10)
    c3  meancentered X(200 x 50000)   groups [25, 25, 25, 25], n_cond 2, mc 0
    c4  behavioral   X(500 x 200000)  Y(500 x 50), split-half leg
    c5  regression   X(1000 x 100000) Y(1000 x 20), k = 15 (SIMPLS; T = 20 > 11:
        the reference's own rank-1 randomized SVD is approximate there, so this
        config is pinned on the oracle's exact SIMPLS, not on the reference --
        "parity unpinned" per SURVEY.md section 0.3)
"""
import numpy as np
import pytest

from conftest import assert_close
from oracle import cpu_ref as ref

pytestmark = pytest.mark.gpu


def _engine():
    from pypyls_amd.engine import Engine, options_from_env
    return Engine(**options_from_env())          # PLSX_<KEY>=1 (monkeypatched per test) -> plsx_set_option


def _synth(S, B, T, seed=0):
    rs = np.random.RandomState(seed)
    X = rs.randn(S, B)
    Y = rs.randn(S, T) + 0.3 * X[:, :T]
    return X, Y


def test_config_c2_behavioral_80x10000():
    """20 permutations + 20 bootstraps vs the oracle, then the literal
    n_perm = n_boot = 5000 front-end call (p-value counts vs the oracle's null
    on the shared sample; layout)."""
    import pypyls_amd as pls
    from pypyls_amd import resampling as rsmp
    S, B, T = 80, 10000, 10
    X, Y = _synth(S, B, T)
    eng = _engine()
    eng.set_data(X, Y, rsmp.cell_of_row([S], 1), 1, 1, 0)
    spec = ref.Spec('behavioral', [S], 1)
    U, d, V = ref.decompose(spec, X, Y)
    xw, sv, yw = eng.decompose()
    assert_close(sv, np.diag(d), 1e-9, what='c2 singvals')
    sgn = np.sign(np.sum(xw * U, axis=0))
    assert_close(xw * sgn, U, 1e-7, what='c2 x_weights')
    eng.set_original(U, np.diag(d), V)
    n = 20
    perms = rsmp.gen_permsamp([S], 1, n, seed=1234)
    boots = rsmp.gen_bootsamp([S], 1, n, seed=1235)
    got = eng.perm(perms)
    want = np.stack([ref.single_perm(spec, X, Y, perms[:, i], V)[0] for i in range(n)], -1)
    assert_close(got, want, 1e-8, what='c2 rotated perm singvals')
    usum, usq, dist = eng.boot(boots)
    ws, wq, wd = np.zeros_like(U), np.zeros_like(U), []
    for i in range(n):
        dd, ub = ref.single_boot(spec, X, Y, boots[:, i], U, d)
        ws += ub
        wq += ub ** 2
        wd.append(dd)
    assert_close(usum.cpu().numpy(), ws, 1e-7, what='c2 u_sum')
    assert_close(usq.cpu().numpy(), wq, 1e-7, what='c2 u_square')
    assert_close(dist, np.stack(wd, -1), 1e-7, what='c2 distrib')
    # the literal config through the front-end
    res = pls.behavioral_pls(X, Y, n_perm=5000, n_boot=5000, test_split=0, seed=1234, verbose=False)
    assert res.permres.perm_singval.shape == (T, 5000)
    assert res.bootres.y_loadings_boot.shape == (T, T, 5000)
    assert_close(res.singvals, np.diag(d), 1e-9, what='c2 front-end singvals')
    # the first 20 permutations of the front-end run against the oracle
    ps = res.permres.permsamples
    want = np.stack([ref.single_perm(spec, X, Y, ps[:, i], res.y_weights)[0] for i in range(n)], -1)
    assert_close(res.permres.perm_singval[:, :n], want, 1e-8, what='c2 front-end perm')
    assert np.array_equal(res.permres.pvals, ref.perm_sig(np.diag(res.singvals), res.permres.perm_singval))
    bsr = res.bootres.x_weights_normed
    assert np.all(np.isfinite(bsr)) and bsr.shape == (B, T)


def test_config_c3_meancentered_200x50000():
    """10 permutations + 10 bootstraps vs the oracle at the c3 shape."""
    from pypyls_amd import resampling as rsmp
    groups, n_cond = [25, 25, 25, 25], 2
    S, B = 200, 50000
    rs = np.random.RandomState(0)
    X = rs.randn(S, B)
    cells = rsmp.cell_of_row(groups, n_cond)
    X += 0.25 * rs.randn(8, B)[cells]                    # cell effects so the LVs separate
    eng = _engine()
    eng.set_data(X, None, cells, len(groups), n_cond, 1, mean_centering=0)
    spec = ref.Spec('meancentered', groups, n_cond, False, 0)
    Y = spec.dummy.astype(float)
    assert_close(eng.crosscov(n=1)[0], ref.gen_covcorr(spec, X, Y, spec.dummy), 1e-10, what='c3 R')
    U, d, V = ref.decompose(spec, X, Y)
    live = ref.live_lvs(d)
    assert live.sum() == 4                              # mc 0: (n_cond - 1) x n_groups live LVs
    xw, sv, yw = eng.decompose()
    assert_close(sv[live], np.diag(d)[live], 1e-9, what='c3 singvals')
    eng.set_original(U, np.diag(d), V)
    n = 10
    perms = rsmp.gen_permsamp(groups, n_cond, n, seed=1234)
    boots = rsmp.gen_bootsamp(groups, n_cond, n, seed=1235)
    got = eng.perm(perms)
    want = np.stack([ref.single_perm(spec, X, Y, perms[:, i], V)[0] for i in range(n)], -1)
    assert_close(got[live], want[live], 1e-8, what='c3 perm')
    usum, usq, dist = eng.boot(boots)
    ws, wq, wd = np.zeros_like(U), np.zeros_like(U), []
    for i in range(n):
        dd, ub = ref.single_boot(spec, X, Y, boots[:, i], U, d)
        ws += ub
        wq += ub ** 2
        wd.append(dd)
    assert_close(usum.cpu().numpy()[:, live], ws[:, live], 1e-7, what='c3 u_sum')
    assert_close(usq.cpu().numpy()[:, live], wq[:, live], 1e-7, what='c3 u_square')
    assert_close(dist[:, live], np.stack(wd, -1)[:, live], 1e-7, what='c3 distrib')


@pytest.mark.parametrize('fused', [True, False])
def test_config_c4_split_half_500x200000(fused, monkeypatch):
    """c4 split-half leg at full size: the original arrangement and one permuted
    arrangement x 3 splits against ref.split_half, with the fused one-pass
    epilogue and with the generic two-pass path (PLSX_NO_SPLIT_FUSE=1)."""
    from pypyls_amd import resampling as rsmp
    if not fused:
        monkeypatch.setenv('PLSX_NO_SPLIT_FUSE', '1')
    S, B, T = 500, 200000, 50
    X, Y = _synth(S, B, T, seed=4)
    eng = _engine()
    eng.set_data(X, Y, rsmp.cell_of_row([S], 1), 1, 1, 0)
    spec = ref.Spec('behavioral', [S], 1)
    ns, npm = 3, 3                                       # the original + three permuted arrangements x three splits
    masks = np.stack([rsmp.gen_splits([S], 1, ns, seed=30 + i) for i in range(1 + npm)])
    perm = rsmp.gen_permsamp([S], 1, npm, seed=7)
    uc0, vc0 = eng.split_half(masks[:1])
    uc1, vc1 = eng.split_half(masks[1:], perms=perm)
    cases = [(uc0[0], vc0[0], Y, masks[0])] + [(uc1[p], vc1[p], Y[perm[:, p]], masks[1 + p]) for p in range(npm)]
    for a, (uc, vc, Yp, mk) in enumerate(cases):
        U, d, V = ref.decompose(spec, X, Yp)
        di = np.linalg.inv(d)
        for i in range(ns):
            u, v = ref.split_half(spec, X, Yp, U @ di, V @ di, mk[:, [i]])
            assert_close(uc[:, i], u, 1e-6, what='c4 ucorr arrangement {} split {}'.format(a, i))
            assert_close(vc[:, i], v, 1e-6, what='c4 vcorr arrangement {} split {}'.format(a, i))


def test_config_c5_regression_1000x100000():
    """SIMPLS at the c5 shape: original fit, 2 permutations and 2 bootstraps vs the
    oracle's exact SIMPLS (regression_single_perm / regression_single_boot)."""
    import pypyls_amd as pls
    from pypyls_amd import resampling as rsmp
    S, B, T, k = 1000, 100000, 20, 15
    X, Y = _synth(S, B, T, seed=2)
    perms = rsmp.gen_permsamp([S], 1, 2, seed=1234)
    boots = rsmp.gen_bootsamp([S], 1, 2, seed=1235)
    res = pls.pls_regression(X, Y, n_components=k, n_perm=2, n_boot=2, permsamples=perms,
                             bootsamples=boots, verbose=False)
    Xc = X - X.mean(axis=0, keepdims=True)
    Yc = Y - Y.mean(axis=0, keepdims=True)
    fit = ref.simpls(Xc, Yc, k)
    W = fit['x_weights']
    sgn = np.sign(np.sum(W * res.x_weights, axis=0))
    assert_close(res.x_weights * sgn, W, 1e-6, what='c5 x_weights')
    assert_close(res.varexp, fit['pctvar'][1], 1e-7, what='c5 pctvar')
    want_p = np.stack([ref.regression_single_perm(Xc, Yc, perms[:, i], k) for i in range(2)], -1)
    assert_close(res.permres.perm_singval, want_p, 1e-6, what='c5 perm pctvar')
    W0 = res.x_weights
    us, uq, yl = W0.copy(), W0 ** 2, []
    for i in range(2):
        y, w = ref.regression_single_boot(Xc, Yc, boots[:, i], k, W0)
        us += w
        uq += w ** 2
        yl.append(y)
    bsr, se = ref.boot_rel(W0, us, uq, 3)
    assert_close(res.bootres.y_loadings_boot, np.stack(yl, -1), 1e-6, what='c5 y_loadings_boot')
    assert_close(res.bootres.x_weights_stderr, se, 1e-5, what='c5 x_weights_stderr')


def test_config_c5_one_batch_of_2304_takes_the_timed_solver_instantiations():
    """What bench.py times at c5 is ONE solver batch of 5000 resamples: above 2048 resamples per batch the solver runs
    its three-waves-per-SIMD instantiations (k_sd_post0 / k_sd_step<0, false, 8> / k_sd_final with SD_RC = 8,
    csrc/plsx_simpls.h), which the 2 + 2 test above never launches.  Here 8 distinct permutations and 8 distinct
    bootstraps at the LITERAL c5 shape (1000 x 100 000, T = 20, k = 15) are submitted as ONE batch of 2304 each
    (index columns replicated, the distinct ones spread over the batch incl. its first and last slot), and every
    distinct resample is compared with the oracle's exact SIMPLS (regression.py:279-373): permutation statistic,
    bootstrap y-loadings, and the accumulated weight sums (replication-weighted) through the quadratic-form closing
    pass that such a series takes."""
    from pypyls_amd.engine import Engine
    from pypyls_amd import resampling as rsmp
    S, B, T, k = 1000, 100000, 20, 15
    X, Y = _synth(S, B, T, seed=2)
    Xc = X - X.mean(axis=0, keepdims=True)
    Yc = Y - Y.mean(axis=0, keepdims=True)
    n, nd = 2304, 8
    perms = rsmp.gen_permsamp([S], 1, nd, seed=4321, verbose=False)
    boots = rsmp.gen_bootsamp([S], 1, nd, seed=4322, verbose=False)
    which = np.arange(n) % nd                                   # slot -> distinct resample
    which[[0, 1, n - 2, n - 1]] = [5, 2, 7, 0]                  # (the ends of the batch hold other ones than the cycle)
    eng = Engine()
    try:
        eng.set_data_regression(Xc, Yc, k)
        W, pct, cvec, yl = eng.simpls_decompose()
        signs = np.sign(W[np.argmax(np.abs(W), axis=0), np.arange(k)])
        signs[signs == 0] = 1.0
        W = W * signs
        eng.simpls_set_original(W)
        big_p = np.asarray(eng.simpls_perm(perms[:, which]))              # (k, n)
        us, uq, ylb = eng.simpls_boot(boots[:, which])
        assert eng.boot_begin(n) == 1                                     # such a series takes the quadratic-form route
        eng.boot_finish(eng._zeros((B, k)), eng._zeros((B, k)))
        us, uq, ylb = us.cpu().numpy(), uq.cpu().numpy(), np.asarray(ylb)
    finally:
        eng.close()
    # replicated slots agree among themselves (same rows -> same result, whatever wave and slot ran them) ...
    for d in range(nd):
        cols = np.flatnonzero(which == d)
        assert np.ptp(big_p[:, cols], axis=1).max() <= 1e-12 * np.abs(big_p[:, cols]).max(), d
        assert np.ptp(ylb[..., cols], axis=-1).max() <= 1e-11 * np.abs(ylb[..., cols]).max(), d
    # ... and with the oracle
    want_us, want_uq = np.zeros((B, k)), np.zeros((B, k))
    for d in range(nd):
        first = int(np.flatnonzero(which == d)[0])
        want = ref.regression_single_perm(Xc, Yc, perms[:, d], k)
        assert_close(big_p[:, first], want, 1e-6, what='c5 permutation {} (slot {}) vs oracle'.format(d, first))
        ylw, w = ref.regression_single_boot(Xc, Yc, boots[:, d], k, W)
        assert_close(ylb[..., first], ylw, 1e-6, what='c5 bootstrap {} y-loadings vs oracle'.format(d))
        mult = int(np.sum(which == d))
        want_us += mult * w
        want_uq += mult * w ** 2
    for c in range(k):
        assert_close(us[:, c], want_us[:, c], 1e-6, what='c5 sum of weights, component {}'.format(c))
        assert_close(uq[:, c], want_uq[:, c], 1e-6, what='c5 sum of squared weights, component {}'.format(c))
