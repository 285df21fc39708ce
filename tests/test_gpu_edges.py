"""Edge cases and full-size properties of the device path (GPU only)."""
import os
import numpy as np
import pytest

from conftest import assert_close
from oracle import cpu_ref as ref

pytestmark = pytest.mark.gpu


def _engine():
    from pypyls_amd.engine import Engine, options_from_env
    return Engine(**options_from_env())          # PLSX_<KEY>=1 (monkeypatched per test) -> plsx_set_option


def _bind(eng, X, Y, groups, n_cond, **kw):
    from pypyls_amd import resampling as rsmp
    eng.set_data(X, Y, rsmp.cell_of_row(groups, n_cond), len(groups), n_cond, 0, **kw)
    return ref.Spec('behavioral', groups, n_cond, kw.get('covariance', False))


@pytest.mark.parametrize('S,B,T,groups,n_cond', [
    (90, 333, 80, [90], 1),          # T' = 80 > 64: generic tiled Gram path, LT = 5, Householder + QL solver
    (96, 211, 24, [24, 24], 2),      # T' = J*T = 4*24 = 96, ragged B
    (100, 300, 64, [100], 1),        # T' = 64: the last size of the LDS Jacobi solver
    (100, 300, 65, [100], 1),        # T' = 65: the first size of the QL solver
    (140, 600, 100, [140], 1),       # T' = 100 (the reference's own test width), LT = 7
    (200, 400, 137, [200], 1),       # T' = 137: the largest matrix the QL solver keeps entirely in LDS
    (200, 400, 138, [200], 1),       # T' = 138: leading 137 x 137 block in LDS, the rest in the global workspace
    (260, 500, 192, [260], 1),       # T' = 192 / 193: one / two rows of the eigenvector matrix per rotating thread
    (260, 500, 193, [260], 1),
    (200, 500, 30, [25, 25], 4),     # T' = 8 cells x 30 = 240: two chunks of L tiles, rotation operand staged in pieces
    (400, 900, 352, [400], 1),       # T' = 352 = the block limit (22 data tiles + moments)
    (420, 700, 353, [420], 1),       # T' = 353: first sliced layout (one cell cut into two row slices)
    (100, 1000, 100, [25], 4),       # T' = 400, the reference's own test shape (pyls/tests/types/test_svd.py:87)
    (100, 1000, 100, [25, 25], 2),   # T' = 400, two groups x two conditions (test_svd.py:95)
    (120, 1300, 100, [10, 10, 10], 4),  # T' = 1200 = 3 groups x 4 conditions x 100 behaviours: 4 slices
                                        # (B >= T': with L < T' the Procrustes result depends on WHICH
                                        # null vectors the SVD returns, in the reference too)
    (160, 520, 30, [10] * 8, 2),     # T' = 480 from 16 cells: slices share a cell, 12 + 5 moment rows
    (450, 600, 385, [450], 1),       # T' = 385 / 577: three / seven rows per rotating thread
    (640, 800, 577, [640], 1),
    (33, 17, 2, [33], 1),            # tiny, S not a multiple of 8, B < 128
    (603, 140, 3, [603], 1),         # S > 512
])
def test_shapes_at_the_limits(S, B, T, groups, n_cond):
    from pypyls_amd import resampling as rsmp
    rs = np.random.RandomState(S + B)
    X = rs.randn(S, B)
    Y = rs.randn(S, T) + (0.5 * X[:, :T] if B >= T else 0)
    eng = _engine()
    spec = _bind(eng, X, Y, groups, n_cond)
    U, d, V = ref.decompose(spec, X, Y)
    xw, sv, yw = eng.decompose()
    live = sv > 1e-4 * sv[0]
    assert_close(sv[live], np.diag(d)[live], 1e-7, what='singvals')
    eng.set_original(U, np.diag(d), V)
    perms = rsmp.gen_permsamp(groups, n_cond, 3, seed=1)
    got = eng.perm(perms)
    want = np.stack([ref.single_perm(spec, X, Y, perms[:, i], V)[0] for i in range(3)], -1)
    assert_close(got, want, 1e-7, what='perm')
    boots = rsmp.gen_bootsamp(groups, n_cond, 3, seed=2)
    usum, usq, dist = eng.boot(boots)
    ws, wd = np.zeros_like(U), []
    for i in range(3):
        dd, ub = ref.single_boot(spec, X, Y, boots[:, i], U, d)
        ws += ub
        wd.append(dd)
    assert_close(usum.cpu().numpy()[:, live], ws[:, live], 1e-6, what='u_sum')
    assert_close(dist[:, live], np.stack(wd, -1)[:, live], 1e-6, what='distrib')


def test_unsupported_and_bad_arguments_fail_loudly():
    from pypyls_amd.engine import PlsxError
    from pypyls_amd import resampling as rsmp
    rs = np.random.RandomState(0)
    eng = _engine()
    with pytest.raises(PlsxError):                       # T' = 1281 > 1280
        eng.set_data(rs.randn(400, 50), rs.randn(400, 1281), rsmp.cell_of_row([400], 1), 1, 1, 0)
    with pytest.raises(PlsxError):                       # perm before set_data / set_original
        _engine().perm(np.zeros((0, 1), int))
    X, Y = rs.randn(30, 40), rs.randn(30, 3)
    _bind(eng, X, Y, [30], 1)
    with pytest.raises(PlsxError):                       # original not set
        eng.perm(rsmp.gen_permsamp([30], 1, 2, seed=0))
    with pytest.raises(ValueError):                      # wrong number of rows
        eng.crosscov(ysrc=np.zeros((29, 2), int))
    cells = np.array([0] * 10 + [1] * 10 + [0] * 10, np.int32)   # non-contiguous cells
    with pytest.raises(PlsxError):
        eng.set_data(X, Y, cells, 2, 1, 0)


def test_single_resample_and_batch_boundaries():
    """n = 1 and n straddling the group / super-batch sizes give the same
    per-resample numbers as one big call."""
    from pypyls_amd import resampling as rsmp
    rs = np.random.RandomState(3)
    X, Y = rs.randn(50, 500), rs.randn(50, 7)
    eng = _engine()
    spec = _bind(eng, X, Y, [50], 1)
    U, d, V = ref.decompose(spec, X, Y)
    eng.set_original(U, np.diag(d), V)
    perms = rsmp.gen_permsamp([50], 1, 77, seed=1)
    full = eng.perm(perms)
    np.testing.assert_allclose(eng.perm(perms[:, :1]), full[:, :1], rtol=1e-12)
    np.testing.assert_allclose(eng.perm(perms[:, 5:36]), full[:, 5:36], rtol=1e-12)
    boots = rsmp.gen_bootsamp([50], 1, 40, seed=2)
    u1, q1, d1 = eng.boot(boots)
    u2, q2, d2a = eng.boot(boots[:, :13])
    u2, q2, d2b = eng.boot(boots[:, 13:], usum=u2, usq=q2)
    np.testing.assert_allclose(u2.cpu().numpy(), u1.cpu().numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(np.concatenate([d2a, d2b], -1), d1, rtol=1e-12)


def test_full_size_properties():
    """BASELINE size (X 500 x 200000, Y 500 x 50): size-independent properties
    instead of an oracle run (one oracle resample takes ~5 s on the host)."""
    from pypyls_amd import hostmath, resampling as rsmp
    S, B, T = 500, 200000, 50
    rs = np.random.RandomState(0)
    X = rs.randn(S, B)
    Y = rs.randn(S, T) + 0.3 * X[:, :T]
    eng = _engine()
    eng.set_data(X, Y, rsmp.cell_of_row([S], 1), 1, 1, 0)
    xw, sv, yw = eng.decompose()
    xw, yw = hostmath.sign_convention(xw, yw)
    # decomposition: orthonormal factors, descending singular values, R = V d U^T on a sample
    assert np.all(np.diff(sv) <= 0)
    np.testing.assert_allclose(xw.T @ xw, np.eye(T), atol=1e-9)
    np.testing.assert_allclose(yw.T @ yw, np.eye(T), atol=1e-9)
    R = eng.crosscov(n=1)[0]
    cols = rs.choice(B, 64, replace=False)
    np.testing.assert_allclose((yw * sv) @ xw[cols].T, R[:, cols], atol=1e-9)
    np.testing.assert_allclose(R[:, cols], ref.xcorr(X[:, cols], Y), atol=1e-11)
    eng.set_original(xw, sv, yw)
    ident = np.arange(S)[:, None]
    # identity permutation: rotated singular values reproduce the originals
    perms = np.hstack([ident, rsmp.gen_permsamp([S], 1, 6, seed=1)])
    ssd = eng.perm(perms)
    np.testing.assert_allclose(ssd[:, 0], sv, rtol=1e-9)
    # Procrustes rotation preserves the total sum of squares
    raw = eng.perm(perms, rotate=False)
    np.testing.assert_allclose((ssd ** 2).sum(0), (raw ** 2).sum(0), rtol=1e-9)
    # identity bootstrap: U_rot = U0 d0, distrib = y_loadings
    boots = np.hstack([ident, rsmp.gen_bootsamp([S], 1, 3, seed=2)])
    usum, usq, dist = eng.boot(boots[:, :1])
    np.testing.assert_allclose(usum.cpu().numpy(), xw * sv, atol=1e-9)
    np.testing.assert_allclose(usq.cpu().numpy(), (xw * sv) ** 2, atol=1e-9)
    xs = eng.project(xw) + (eng.colmean() @ xw)[None]
    np.testing.assert_allclose(dist[:, :, 0], ref.xcorr(xs, Y), atol=1e-9)
    # accumulation is additive and order independent
    u_all, q_all, _ = eng.boot(boots)
    u_b, q_b, _ = eng.boot(boots[:, ::-1])
    np.testing.assert_allclose(u_all.cpu().numpy(), u_b.cpu().numpy(), rtol=1e-9, atol=1e-11)
    # relabelling the rows of X and Y together changes nothing
    order = rs.permutation(S)
    eng2 = _engine()
    eng2.set_data(X[order], Y[order], rsmp.cell_of_row([S], 1), 1, 1, 0)
    np.testing.assert_allclose(eng2.decompose()[1], sv, rtol=1e-10)


@pytest.mark.parametrize('n', [1, 2, 7, 100, 1000, 5000, 16384])
def test_percentile_ci_matches_numpy(n):
    """Device bitonic-sort percentile == numpy.percentile (linear), bit for bit."""
    rs = np.random.RandomState(n)
    boot = rs.randn(3, 5, n)
    if n >= 7:
        boot[1, 2, :3] = boot[1, 2, 3]                      # ties
    eng = _engine()
    for ci in (95, 90, 50):
        lo, hi = eng.percentile_ci(boot, ci=ci)
        wlo, whi = ref.boot_ci(boot, ci=ci)
        np.testing.assert_array_equal(lo, wlo)
        np.testing.assert_array_equal(hi, whi)
    boot[0, 0, 0] = np.nan
    lo, hi = eng.percentile_ci(boot)
    assert np.isnan(lo[0, 0]) and np.isnan(hi[0, 0]) and np.isfinite(lo[1, 1])


def test_percentile_selection_equals_full_sort_and_numpy(monkeypatch):
    """Long series with both ranks in the tails are settled by selection (k_percentile_sel: pivots from a sorted
    sample, one counting pass, a sort of the values beyond the pivots) instead of a full sort -- the same order
    statistics bit for bit: random, heavily tied, constant, sorted, periodic, heavy-tailed, with infinities and
    NaN, at the lengths and intervals the front-ends use and at the edge of the selection's range."""
    rs = np.random.RandomState(7)
    for n in (4096, 5000, 10000, 16384):
        rows = [rs.randn(n), np.round(rs.randn(n), 1), np.full(n, 3.25), np.sort(rs.randn(n)),
                np.sort(rs.randn(n))[::-1].copy(), np.tile(rs.randn(8), n // 8 + 1)[:n], rs.standard_cauchy(n),
                np.where(rs.rand(n) < 0.97, 0.0, rs.randn(n)), np.arange(n, dtype=float) % 3,
                np.concatenate([np.full(n - 5, 1.0), [np.inf, -np.inf, 2.0, -2.0, 0.5]])]
        boot = np.stack(rows)
        for ci in (95, 99, 90, 80):
            monkeypatch.delenv('PLSX_PERCENTILE_SORT', raising=False)
            lo, hi = _engine().percentile_ci(boot, ci=ci)
            monkeypatch.setenv('PLSX_PERCENTILE_SORT', '1')
            lo2, hi2 = _engine().percentile_ci(boot, ci=ci)
            low = (100 - ci) / 2
            wlo, whi = np.percentile(boot, [low, 100 - low], axis=-1)
            for a, b, c, what in ((lo, lo2, wlo, 'lower'), (hi, hi2, whi, 'upper')):
                np.testing.assert_array_equal(a, b, err_msg='selection vs sort, {} n={} ci={}'.format(what, n, ci))
                np.testing.assert_array_equal(a, c, err_msg='selection vs numpy, {} n={} ci={}'.format(what, n, ci))
    monkeypatch.delenv('PLSX_PERCENTILE_SORT', raising=False)
    boot = rs.randn(4, 10000)
    boot[2, 17] = np.nan
    lo, hi = _engine().percentile_ci(boot)
    assert np.isnan(lo[2]) and np.isnan(hi[2]) and np.isfinite(lo[[0, 1, 3]]).all()


def test_nonfinite_input_rejected():
    import pypyls_amd as pls
    rs = np.random.RandomState(0)
    X, Y = rs.rand(30, 40), rs.rand(30, 3)
    Xb = X.copy()
    Xb[4, 7] = np.nan
    with pytest.raises(ValueError):
        pls.behavioral_pls(Xb, Y, n_perm=0, n_boot=0, test_split=0)
    Xb[4, 7] = np.inf
    with pytest.raises(ValueError):
        pls.behavioral_pls(Xb, Y, n_perm=0, n_boot=0, test_split=0)
    Yb = Y.copy()
    Yb[2, 1] = np.nan
    with pytest.raises(ValueError):
        pls.behavioral_pls(X, Yb, n_perm=0, n_boot=0, test_split=0)
    with pytest.raises(ValueError):
        pls.meancentered_pls(Xb, groups=[15, 15], n_perm=0, n_boot=0)


def test_context_rebinding_matches_fresh_context():
    """plsx_set_data on a context that already ran another problem (larger, then
    smaller, different method): stale scratch must not leak into the results."""
    from pypyls_amd import resampling as rsmp
    rs = np.random.RandomState(9)

    def run(eng, S, B, T, groups, n_cond):
        X = np.random.RandomState(S + B).randn(S, B)
        Y = np.random.RandomState(T).randn(S, T) + 0.4 * X[:, :T]
        spec = _bind(eng, X, Y, groups, n_cond)
        U, d, V = ref.decompose(spec, X, Y)
        eng.set_original(U, np.diag(d), V)
        perms = rsmp.gen_permsamp(groups, n_cond, 6, seed=1)
        boots = rsmp.gen_bootsamp(groups, n_cond, 6, seed=2)
        masks = rsmp.gen_splits(groups, n_cond, 5, seed=3)
        usum, usq, dist = eng.boot(boots)
        uc, vc = eng.split_half(masks)
        return eng.perm(perms), usum.cpu().numpy(), usq.cpu().numpy(), dist, uc, vc

    shapes = [(90, 700, 12, [45, 45], 1), (40, 130, 3, [40], 1), (64, 300, 20, [16, 16], 2)]
    shared = _engine()
    for shp in shapes:
        got = run(shared, *shp)
        want = run(_engine(), *shp)
        for a, b in zip(got, want):
            assert np.array_equal(a, b, equal_nan=True)


def test_full_size_resamples_against_oracle():
    """Eight permutations (both routes) and eight bootstraps at the BASELINE size, directly against the
    oracle, per resample (~25 s of host time on the box's 256 cores)."""
    from pypyls_amd import hostmath, resampling as rsmp
    S, B, T, n = 500, 200000, 50, 8
    rs = np.random.RandomState(3)
    X = rs.randn(S, B)
    Y = rs.randn(S, T) + 0.3 * X[:, :T]
    eng = _engine()
    spec = _bind(eng, X, Y, [S], 1)
    xw, sv, yw = eng.decompose()
    xw, yw = hostmath.sign_convention(xw, yw)
    eng.set_original(xw, sv, yw)
    perms = rsmp.gen_permsamp([S], 1, n, seed=11)
    boots = rsmp.gen_bootsamp([S], 1, n, seed=12)
    want_p = np.stack([ref.single_perm(spec, X, Y, perms[:, i], yw)[0] for i in range(n)], -1)
    for dual in (True, False):
        eng.set_perm_path(dual)
        got_p = eng.perm(perms)
        for i in range(n):
            assert_close(got_p[:, i], want_p[:, i], 1e-9, what='perm singvals at full size, dual={} #{}'.format(dual, i))
    eng.set_perm_path(True)
    for i in range(n):                                   # one at a time: sum U of ONE bootstrap is its rotated weights
        usum, usq, dist = eng.boot(boots[:, [i]])
        wd, wu = ref.single_boot(spec, X, Y, boots[:, i], xw, np.diag(sv))
        assert_close(usum.cpu().numpy(), wu, 1e-8, what='rotated bootstrap weights at full size #{}'.format(i))
        assert_close(usq.cpu().numpy(), wu ** 2, 1e-8, what='squared bootstrap weights at full size #{}'.format(i))
        assert_close(dist[:, :, 0], wd, 1e-9, what='bootstrap distrib at full size #{}'.format(i))
    usum, usq, dist = eng.boot(boots)                    # ... and the batch of eight together
    ws = sum(ref.single_boot(spec, X, Y, boots[:, i], xw, np.diag(sv))[1] for i in range(n))
    assert_close(usum.cpu().numpy(), ws, 1e-8, what='sum U of eight bootstraps at full size')


def test_randomised_parity_sweep():
    """tools/fuzz_parity.py: random designs / shapes / options (behavioral with and
    without covariance, mean-centred with every centring, 1-3 groups x 1-3
    conditions, B from 3 to 1500, rotate on / off, split-half) against the oracle."""
    import gc
    import subprocess
    import sys
    import torch
    from conftest import ROOT
    gc.collect()                                        # engines of earlier tests hold tens of GB of device scratch;
    torch.cuda.empty_cache()                            # the sweep runs in its own process on the same GPU
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'fuzz_parity.py'), '40', '123'],
                          capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-2000:]
    assert 'failures: 0 of 40' in proc.stdout


@pytest.mark.parametrize('case', ['mc0', 'mc1', 'mc2', 'cov', 'cov_2g2c'])
def test_single_pass_bootstrap_equals_two_pass(case, monkeypatch):
    """Unscaled modes (mean-centred PLS, covariance-mode behavioral PLS): the bootstrap takes ONE
    pass over the features per resample (G, P from the S x S kernel, U = X^T (A^T M) accumulated in the
    cross-product epilogue; plsx_core.hip boot_single_pass).  Same sum U, sum U^2, distrib as the
    two-pass route (R written, Gram pass, rotation pass) and as the oracle."""
    from pypyls_amd import resampling as rsmp
    rs = np.random.RandomState(17)
    if case.startswith('mc'):
        groups, n_cond, mc = [14, 11, 12], 3, int(case[2])
        S, B = sum(groups) * n_cond, 3000
        cells = rsmp.cell_of_row(groups, n_cond)
        X = rs.randn(S, B) + 0.5 * rs.randn(len(groups) * n_cond, B)[cells]
        Y, method, spec = None, 1, ref.Spec('meancentered', groups, n_cond, False, mc)
    else:
        groups, n_cond, mc = ([60], 1, 0) if case == 'cov' else ([21, 24], 2, 0)
        S, B, T = sum(groups) * n_cond, 2500, 6
        cells = rsmp.cell_of_row(groups, n_cond)
        X = rs.randn(S, B) * (1.0 + rs.rand(1, B))
        Y = rs.randn(S, T) * np.array([1, 2, .5, 3, 1, .7]) + 0.5 * X[:, :T]
        method, spec = 0, ref.Spec('behavioral', groups, n_cond, True, 0)
    Yo = Y if Y is not None else spec.dummy.astype(float)
    boots = rsmp.gen_bootsamp(groups, n_cond, 130, seed=5)
    U, d, V = ref.decompose(spec, X, Yo)
    out = {}
    for route in ('single', 'two'):
        if route == 'two':
            monkeypatch.setenv('PLSX_TWO_PASS_BOOT', '1')
        eng = _engine()
        eng.set_data(X, Y, cells, len(groups), n_cond, method, mean_centering=mc, covariance=(method == 0))
        eng.set_original(U, np.diag(d), V)
        usum, usq, dist = eng.boot(boots)
        out[route] = (usum.cpu().numpy(), usq.cpu().numpy(), dist)
    monkeypatch.delenv('PLSX_TWO_PASS_BOOT')
    live = ref.live_lvs(d)
    for a, b, what in zip(out['single'], out['two'], ('sum U', 'sum U^2', 'distrib')):
        assert_close(a[:, live] if a.ndim == 2 else a[:, live], b[:, live] if b.ndim == 2 else b[:, live], 1e-9,
                     what=what + ' single pass vs two pass')
    n = 12                                                  # and against the oracle on a sample
    usum, usq, dist = eng.boot(boots[:, :n])                # (eng = the two-pass engine; run the single-pass one too)
    ws, wq, wd = np.zeros_like(U), np.zeros_like(U), []
    for i in range(n):
        dd, ub = ref.single_boot(spec, X, Yo, boots[:, i], U, d)
        ws += ub
        wq += ub ** 2
        wd.append(dd)
    eng1 = _engine()
    eng1.set_data(X, Y, cells, len(groups), n_cond, method, mean_centering=mc, covariance=(method == 0))
    eng1.set_original(U, np.diag(d), V)
    usum, usq, dist = eng1.boot(boots[:, :n])
    assert_close(usum.cpu().numpy()[:, live], ws[:, live], 1e-8, what='single-pass sum U vs oracle')
    assert_close(usq.cpu().numpy()[:, live], wq[:, live], 1e-8, what='single-pass sum U^2 vs oracle')
    assert_close(dist[:, live], np.stack(wd, -1)[:, live], 1e-8, what='single-pass distrib vs oracle')


def test_rebinding_a_context_equals_a_fresh_one():
    """a re-bound context skips the zero fill of its R scratch when the slot geometry is unchanged.
    The geometry key holds (T', T'pp, Bpad, B, L, method); bindings that differ only in what the key must tell
    apart -- another B inside the same Bpad, another method with the same T' -- give the results of a fresh
    engine."""
    from pypyls_amd import resampling as rsmp
    rs = np.random.RandomState(3)
    S, T = 60, 6
    X1, X2 = rs.randn(S, 1010), rs.randn(S, 1000)          # B + L = 1016 / 1006: both pad to 1024
    Y = rs.randn(S, T) + 0.4 * X1[:, :T]
    boots = rsmp.gen_bootsamp([S], 1, 12, seed=5)
    perms = rsmp.gen_permsamp([S], 1, 12, seed=6)

    def behavioral(eng, X):
        _bind(eng, X, Y, [S], 1)
        xw, sv, yw = eng.decompose()
        eng.set_original(xw, sv, yw)
        return (sv, eng.perm(perms)) + tuple(eng.boot(boots))

    def regression(eng, X):
        Xc, Yc = X - X.mean(0), Y - Y.mean(0)
        eng.set_data_regression(Xc, Yc, T)                  # n_components = T: the same T' = 6 rows per resample
        W, pct, cvec, yl = eng.simpls_decompose()
        eng.simpls_set_original(W)
        return (pct, eng.simpls_perm(perms)) + tuple(eng.simpls_boot(boots))

    shared = _engine()
    seq = [behavioral(shared, X1), behavioral(shared, X2), regression(shared, X2), behavioral(shared, X1)]
    fresh = [behavioral(_engine(), X1), behavioral(_engine(), X2), regression(_engine(), X2), behavioral(_engine(), X1)]
    for k, (a, b) in enumerate(zip(seq, fresh)):
        for x, y in zip(a, b):
            x, y = [t.cpu().numpy() if hasattr(t, 'cpu') else np.asarray(t) for t in (x, y)]
            assert_close(x, y, 1e-12, what='binding {}'.format(k))


def test_a_failed_analysis_leaves_no_state_on_the_shared_engine(monkeypatch):
    """the front-ends share one cached engine.  An analysis that dies half way must not leave its
    announced shard size or its graded-spectrum counters behind (they would size the scratch of, and raise a
    spurious GradedSpectrumWarning in, the next unrelated call), and concurrent callers are serialised."""
    import threading
    import warnings
    import pypyls_amd as pls
    from pypyls_amd import hostmath
    from pypyls_amd.engine import default_engine, GradedSpectrumWarning
    from test_gpu_graded import graded_behaviours
    rs = np.random.RandomState(11)
    S, B, T = 80, 2000, 8
    X = rs.randn(S, B)
    Yg = graded_behaviours(rs, S, T, 3e5, 'mix')
    eng = default_engine()
    eng.set_option('no_refine', 1)                         # graded decompositions are counted, not repaired
    try:
        def boom(*a, **k):
            raise RuntimeError('injected failure after the resampling')
        monkeypatch.setattr(hostmath, 'varexp', boom)
        with pytest.raises(RuntimeError, match='injected'):
            pls.behavioral_pls(X, Yg, n_perm=16, n_boot=16, test_split=0, seed=1, verbose=False)
        monkeypatch.undo()
        assert eng.numeric_report(warn=False) == (0, 0)    # drained on the error path
    finally:
        eng.set_option('no_refine', 0)
    Y = rs.randn(S, T) + 0.4 * X[:, :T]
    with warnings.catch_warnings():
        warnings.simplefilter('error', GradedSpectrumWarning)
        a = pls.behavioral_pls(X, Y, n_perm=16, n_boot=16, test_split=0, seed=2, verbose=False)
    # two threads on the shared engine: serialised by Engine.lock, both get the single-threaded answer
    out = {}

    def work(name):
        out[name] = pls.behavioral_pls(X, Y, n_perm=16, n_boot=16, test_split=0, seed=2, verbose=False)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for i in range(2):
        assert np.array_equal(out[i].singvals, a.singvals)
        assert np.array_equal(out[i].bootres.x_weights_normed, a.bootres.x_weights_normed)
        assert np.array_equal(out[i].permres.perm_singval, a.permres.perm_singval)


def test_more_than_two_million_features():
    """the flat "B <= 2,000,000" rule is gone.  What bounds the shape is one cross-covariance
    matrix R_r (T'pp x Bpad doubles) below 2 GB -- the kernels address it through 31-bit buffer offsets --, i.e. 5.1
    million features at the headline T' = 50 (voxel-wise data at 1 mm has ~1.8 M), 16 M for mean-centred designs.
    The reference takes any shape (pyls/base.py:254-283).  At B = 2.6 M an oracle pass costs 15 s per decomposition,
    so the check is in two steps: the device's cross-covariance matrices against the oracle's xcorr on column windows
    (first, middle, LAST columns), then everything downstream of R -- singular values, weights, permutation null on
    both routes, rotated bootstrap sums -- against numpy working on the device's own R of the same resample.  SIMPLS:
    the original fit against the oracle.  And the shape the old rule let through to kernels that would have read zeros
    (T' = 200 x 1.4 M features: 2.2 GB per R_r) is refused loudly."""
    from pypyls_amd.engine import Engine, PlsxError
    from pypyls_amd import resampling as rsmp
    rs = np.random.RandomState(5)
    S, B, T = 30, 2600000, 3
    X = rs.randn(S, B)
    Y = rs.randn(S, T) + 0.5 * X[:, :T]
    perms = rsmp.gen_permsamp([S], 1, 2, seed=1, verbose=False)
    boots = rsmp.gen_bootsamp([S], 1, 2, seed=2, verbose=False)
    eng = Engine()
    try:
        eng.set_data(X, Y, rsmp.cell_of_row([S], 1), 1, 1, 0)
        ident = np.arange(S)[:, None]
        R0 = eng.crosscov(n=1)[0]
        Rp = eng.crosscov(ysrc=perms)
        Rb = eng.crosscov(xsrc=boots, ysrc=boots)
        for lo in (0, B // 2 - 500, B - 1000):
            w = slice(lo, lo + 1000)
            assert_close(R0[:, w], ref.xcorr(X[:, w], Y), 1e-10, what='R columns %d..' % lo)
            assert_close(Rp[1][:, w], ref.xcorr(X[:, w], Y[perms[:, 1]]), 1e-10, what='permuted R columns %d..' % lo)
            assert_close(Rb[0][:, w], ref.xcorr(X[boots[:, 0]][:, w], Y[boots[:, 0]]), 1e-10, what='bootstrap R columns %d..' % lo)
        xw, sv, yw = eng.decompose()
        U0, d0, V0t = np.linalg.svd(R0.T, full_matrices=False)           # R^T = U d V^T, as compute.svd (compute.py:10-52)
        assert_close(sv, d0, 1e-10, what='singvals')
        sg = np.sign(np.sum(xw * U0, axis=0))
        assert_close(xw * sg, U0, 1e-8, what='x_weights')
        assert np.abs(xw[-5:]).max() > 0                                 # the last columns are real columns
        eng.set_original(xw, sv, yw)
        pd_ = eng.perm(perms, rotate=True)
        eng.set_perm_path(False)
        pf = eng.perm(perms, rotate=True)
        eng.set_perm_path(True)
        assert_close(pd_, pf, 1e-9, what='permutation null: S x S route vs feature pass')
        for i in range(2):
            Up, dp, Vpt = np.linalg.svd(Rp[i].T, full_matrices=False)
            rot = ref.procrustes(yw, Vpt.T, np.diag(dp))
            assert_close(pf[:, i], np.sqrt(np.sum(rot ** 2, axis=0)), 1e-9, what='permutation %d' % i)
        usum, usq, dist = eng.boot(boots)
        want = np.zeros((B, T))
        for i in range(2):
            Ub, db, _ = np.linalg.svd(Rb[i].T, full_matrices=False)
            want += ref.procrustes(xw, Ub, np.diag(db))
        assert_close(usum.cpu().numpy(), want, 1e-8, what='sum of rotated bootstrap weights')
    finally:
        eng.close()
    import pypyls_amd as pls
    rr = pls.pls_regression(X, Y, n_components=2, n_perm=0, n_boot=0, verbose=False)
    fit = ref.simpls(X - X.mean(0), Y - Y.mean(0), 2)
    sg = np.sign(np.sum(rr.x_weights * fit['x_weights'], axis=0))
    assert_close(rr.x_weights * sg, fit['x_weights'], 1e-7, what='simpls x_weights')
    assert_close(rr.varexp, fit['pctvar'][1], 1e-8, what='simpls pctvar')
    pls.release_default_engine()
    eng = Engine()
    try:
        with pytest.raises(PlsxError, match='below 2 GB'):
            eng.set_data(np.zeros((10, 1400000)), np.zeros((10, 200)), np.zeros(10, np.int32), 1, 1, 0)
    finally:
        eng.close()
