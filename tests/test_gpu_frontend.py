"""
Front-end parity on the GPU: pypyls_amd.behavioral_pls / meancentered_pls
against (a) the golden outputs of the REFERENCE itself and (b) the oracle, on
the same inputs and resampling arrays.  Tolerance: 1e-5 relative (north_star),
null LVs masked as the reference's comparator does (pyls/tests/matlab.py:160).
"""
import numpy as np
import pytest

from conftest import load_golden, golden_names, live_lvs, assert_close
from oracle import cpu_ref as ref

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def _run(g, method, with_samples=True, **extra):
    import pypyls_amd as pls
    kw = dict(groups=list(g['groups']), n_cond=int(g['n_cond']), verbose=False,
              rotate=bool(g.get('rotate', True)), seed=int(g['seed']) if 'seed' in g else None)
    n_perm = g['ref_permres__perm_singval'].shape[1] if 'ref_permres__perm_singval' in g else 0
    n_boot = g['ref_bootres__bootsamples'].shape[1] if 'ref_bootres__bootsamples' in g else 0
    kw.update(n_perm=n_perm, n_boot=n_boot)
    if 'n_split' in g:
        kw['n_split'] = int(g['n_split'])
    if with_samples:
        kw['permsamples'] = g.get('ref_permres__permsamples')
        kw['bootsamples'] = g.get('ref_bootres__bootsamples')
        if 'splitsamples' in g:          # the very masks the reference drew
            kw['_splitsamples'] = g['splitsamples']
            kw['_perm_splitsamples'] = g['perm_splitsamples']
    kw.update(extra)
    if method == 'behavioral':
        if 'cv_splits' in g:
            kw.update(test_split=g['cv_splits'].shape[1], test_size=float(g['test_size']))
            if with_samples:
                kw['_cvsplits'] = g['cv_splits']
        else:
            kw['test_split'] = 0
        return pls.behavioral_pls(g['X'], g['Y'], covariance=bool(g.get('covariance', False)), **kw)
    return pls.meancentered_pls(g['X'], mean_centering=int(g.get('mean_centering', 0)), **kw)


def _oracle(g, method):
    return ref.run_plsc(g['X'], g.get('Y'), method=method, groups=list(g['groups']),
                        n_cond=int(g['n_cond']), covariance=bool(g.get('covariance', False)),
                        rotate=bool(g.get('rotate', True)),
                        mean_centering=int(g.get('mean_centering', 0)),
                        permsamples=g.get('ref_permres__permsamples'),
                        bootsamples=g.get('ref_bootres__bootsamples'),
                        splitsamples=g.get('splitsamples'),
                        perm_splitsamples=g.get('perm_splitsamples'))


def _boots_full_rank(g):
    if 'Y' not in g or 'ref_bootres__bootsamples' not in g:
        return False
    T = g['Y'].shape[1]
    cells = ref.dummy_code(list(g['groups']), int(g['n_cond'])).T.astype(bool)
    boots = g['ref_bootres__bootsamples']
    return all(len(np.unique(boots[c, i])) - 1 >= T
               for i in range(boots.shape[1]) for c in cells)


def _compare(res, g, want, keep, boot_tight, per_lv=False):
    if per_lv:
        # against the ORACLE (same conventions): every live latent variable on its own scale
        from conftest import assert_close_per_lv
        assert_close_per_lv(res['singvals'], want['singvals'], 0, RTOL, 'singvals (per LV)', keep)
        for k in ('x_weights', 'y_weights', 'x_scores', 'y_scores', 'y_loadings'):
            if want.get(k) is not None and res.get(k) is not None:
                assert_close_per_lv(res[k], want[k], 1, RTOL, k + ' (per LV)', keep)
        if want.get('permres') and bool(g.get('rotate', True)) is False:
            # unrotated: row k of the null is the k-th singular value of the permuted data (its own scale);
            # rotated rows mix all of them
            assert_close_per_lv(res['permres']['perm_singval'], want['permres']['perm_singval'], 0, RTOL,
                                'perm_singval (per LV)', keep)
        if want.get('bootres'):
            for k in ('y_loadings_boot', 'contrast_boot', 'x_weights_stderr'):
                if want['bootres'].get(k) is not None:
                    assert_close_per_lv(res['bootres'][k], want['bootres'][k], 1, RTOL, k + ' (per LV)', keep)
    assert_close(res['singvals'][keep], want['singvals'][keep], RTOL, what='singvals')
    assert_close(res['varexp'][keep], want['varexp'][keep], RTOL, what='varexp')
    for k in ('x_weights', 'y_weights', 'x_scores', 'y_scores', 'y_loadings'):
        if want.get(k) is not None and res.get(k) is not None:
            assert_close(res[k][:, keep], want[k][:, keep], RTOL, what=k)
    if want.get('permres'):
        assert_close(res['permres']['perm_singval'][keep], want['permres']['perm_singval'][keep],
                     RTOL, what='perm_singval')
        P = want['permres']['perm_singval'].shape[1]
        got = np.rint(res['permres']['pvals'][keep] * (P + 1))
        exp = np.rint(want['permres']['pvals'][keep] * (P + 1))
        np.testing.assert_array_equal(got, exp)          # integer counts
    if want.get('bootres'):
        for k in ('x_weights_normed', 'x_weights_stderr', 'y_loadings_boot', 'y_loadings_ci',
                  'contrast', 'contrast_boot', 'contrast_ci'):
            if k in want['bootres'] and want['bootres'][k] is not None:
                a, b = res['bootres'][k][:, keep], want['bootres'][k][:, keep]
                if boot_tight or k.startswith('contrast') or k.startswith('y_loadings'):
                    assert_close(a, b, RTOL, what=k)
                else:
                    r = ref.efficient_corr(a, b)
                    assert np.all(r >= 0.9), (k, r)


def _compare_split(res, want, keep):
    if not want.get('splitres'):
        return
    for k in ('ucorr', 'vcorr', 'ucorr_lolim', 'ucorr_uplim', 'vcorr_lolim', 'vcorr_uplim'):
        assert_close(res['splitres'][k][keep], want['splitres'][k][keep], RTOL, what='splitres.' + k)
    for k in ('ucorr_pvals', 'vcorr_pvals'):
        np.testing.assert_allclose(res['splitres'][k][keep], want['splitres'][k][keep], atol=1e-12)


def _ref_as_dict(g):
    out = dict(permres={}, bootres={}, splitres={})
    for k, v in g.items():
        if k.startswith('ref_permres__'):
            out['permres'][k[13:]] = v
        elif k.startswith('ref_bootres__'):
            out['bootres'][k[13:]] = v
        elif k.startswith('ref_splitres__'):
            out['splitres'][k[14:]] = v
        elif k.startswith('ref_') and '__' not in k:
            out[k[4:]] = v
    return out


BEHAV = golden_names('bpls_') + ['linnerud'] + golden_names('mat_bpls')
MEANC = golden_names('mpls_') + golden_names('mat_mpls')


@pytest.mark.parametrize('name', BEHAV)
def test_behavioral_vs_reference_and_oracle(name):
    g = load_golden(name)
    res = _run(g, 'behavioral')
    keep = live_lvs(g['ref_singvals'])
    # (b) oracle: same algorithmic conventions -> tight everywhere
    _compare(res, g, _oracle(g, 'behavioral'), keep, boot_tight=True, per_lv=True)
    _compare_split(res, _oracle(g, 'behavioral'), keep)
    _compare_split(res, _ref_as_dict(g), keep)
    if 'cv_splits' in g:
        spec = ref.Spec('behavioral', list(g['groups']), int(g['n_cond']), bool(g.get('covariance', False)))
        r, r2 = ref.crossval(spec, g['X'], g['Y'], g['cv_splits'])
        for got, want_o, want_r, what in ((res['cvres']['pearson_r'], r, g['ref_cvres__pearson_r'], 'pearson_r'),
                                          (res['cvres']['r_squared'], r2, g['ref_cvres__r_squared'], 'r_squared')):
            assert_close(got, want_o, RTOL, what='oracle ' + what)
            assert_close(got, want_r, RTOL, what='reference ' + what)
    # (a) the reference's own numbers; rank-deficient bootstrap rotations of the
    # reference are noise-defined (oracle.procrustes_live) -> functional check
    _compare(res, g, _ref_as_dict(g), keep, boot_tight=_boots_full_rank(g) and bool(np.all(keep)))


@pytest.mark.parametrize('name', MEANC)
def test_meancentered_vs_reference_and_oracle(name):
    g = load_golden(name)
    res = _run(g, 'meancentered')
    keep = live_lvs(g['ref_singvals'])
    _compare(res, g, _oracle(g, 'meancentered'), keep, boot_tight=True, per_lv=True)
    _compare_split(res, _oracle(g, 'meancentered'), keep)
    _compare_split(res, _ref_as_dict(g), keep)
    _compare(res, g, _ref_as_dict(g), keep, boot_tight=False)


@pytest.mark.parametrize('name', ['linnerud', 'bpls_2g2c', 'mpls_3g2c_mc0'])
def test_seed_reproduces_reference_index_arrays(name):
    """With only ``seed`` given the front-end must draw the same permutation /
    bootstrap arrays as the reference (RNG stream compatibility)."""
    g = load_golden(name)
    method = 'behavioral' if 'Y' in g else 'meancentered'
    res = _run(g, method, with_samples=False)
    np.testing.assert_array_equal(res['permres']['permsamples'], g['ref_permres__permsamples'])
    np.testing.assert_array_equal(res['bootres']['bootsamples'], g['ref_bootres__bootsamples'])
    keep = live_lvs(g['ref_singvals'])
    assert_close(res['permres']['perm_singval'][keep], g['ref_permres__perm_singval'][keep],
                 RTOL, what='perm_singval')


def test_cv_seed_reproduces_reference_splits():
    """seed only: the cross-validation masks are drawn from the same stream
    position as the reference's (after the bootstrap arrays)."""
    g = load_golden('bpls_2g2c_cv')
    res = _run(g, 'behavioral', with_samples=False)
    np.testing.assert_array_equal(res['bootres']['bootsamples'], g['ref_bootres__bootsamples'])
    assert_close(res['cvres']['pearson_r'], g['ref_cvres__pearson_r'], RTOL, what='pearson_r')
    assert_close(res['cvres']['r_squared'], g['ref_cvres__r_squared'], RTOL, what='r_squared')


def test_prepermuted_y_stack_equals_index_permutations():
    """permindices=False (pyls/base.py:636-639): a (n_perm, S, T) stack of
    already-permuted Y matrices gives the same null as the index arrays."""
    import pypyls_amd as pls
    g = load_golden('bpls_2g2c')
    perms = g['ref_permres__permsamples']
    ystack = np.stack([g['Y'][perms[:, i]] for i in range(perms.shape[1])])
    kw = dict(groups=list(g['groups']), n_cond=int(g['n_cond']), n_perm=perms.shape[1], n_boot=0,
              test_split=0, verbose=False)
    a = pls.behavioral_pls(g['X'], g['Y'], permsamples=ystack, permindices=False, **kw)
    b = pls.behavioral_pls(g['X'], g['Y'], permsamples=perms, **kw)
    np.testing.assert_allclose(a.permres.perm_singval, b.permres.perm_singval, rtol=1e-12)
    np.testing.assert_array_equal(a.permres.pvals, b.permres.pvals)
    assert a.permres.permsamples.shape == (g['Y'].shape[0], g['Y'].shape[1], perms.shape[1])
    keep = live_lvs(g['ref_singvals'])
    assert_close(a.permres.perm_singval[keep], g['ref_permres__perm_singval'][keep], RTOL, what='vs reference')


def test_prepermuted_y_stack_with_split_half():
    """permindices=False together with n_split (pyls/base.py:691-692, 705-708): the
    split-half null of pre-permuted Y matrices equals that of the index arrays, and
    both equal the reference's golden."""
    import pypyls_amd as pls
    g = load_golden('bpls_2g2c_split')
    perms = g['ref_permres__permsamples']
    ystack = np.stack([g['Y'][perms[:, i]] for i in range(perms.shape[1])])
    kw = dict(groups=list(g['groups']), n_cond=int(g['n_cond']), n_perm=perms.shape[1], n_boot=0,
              n_split=int(g['n_split']), test_split=0, verbose=False,
              _splitsamples=g['splitsamples'], _perm_splitsamples=g['perm_splitsamples'])
    a = pls.behavioral_pls(g['X'], g['Y'], permsamples=ystack, permindices=False, **kw)
    b = pls.behavioral_pls(g['X'], g['Y'], permsamples=perms, **kw)
    for key in ('ucorr', 'vcorr', 'ucorr_pvals', 'vcorr_pvals', 'ucorr_uplim', 'vcorr_lolim'):
        np.testing.assert_allclose(a.splitres[key], b.splitres[key], rtol=1e-9, err_msg=key)
    keep = live_lvs(g['ref_singvals'])
    assert_close(a.splitres['ucorr_pvals'][keep], g['ref_splitres__ucorr_pvals'][keep], RTOL, what='ucorr p')
    assert_close(a.splitres['vcorr_pvals'][keep], g['ref_splitres__vcorr_pvals'][keep], RTOL, what='vcorr p')


def test_linnerud_known_answers():
    g = load_golden('linnerud')
    res = _run(g, 'behavioral')
    np.testing.assert_allclose(res['x_weights'][:, 0], [0.61330742, 0.7469717, 0.25668519], atol=1e-7)
    np.testing.assert_allclose(res['y_weights'][:, 0], [-0.58989118, -0.77134059, 0.23887675], atol=1e-7)
    np.testing.assert_allclose(res['singvals'], [1.1280186599, 0.0752124667, 0.0332524411], rtol=1e-6)
    np.testing.assert_allclose(res['permres']['pvals'], [0.0495049505, 0.9306930693, 1.0], atol=1e-9)
    np.testing.assert_allclose(res['bootres']['x_weights_normed'][:, 0],
                               [2.9107550162, 4.6882819063, 1.5249567446], rtol=1e-5)


def test_errors_and_layout():
    import pypyls_amd as pls
    rs = np.random.RandomState(0)
    X, Y = rs.rand(30, 40), rs.rand(30, 5)
    with pytest.raises(ValueError):
        pls.behavioral_pls(X, Y[:-1], n_perm=0, n_boot=0, test_split=0)
    with pytest.raises(ValueError):
        pls.behavioral_pls(X, Y, groups=[15, 14], n_perm=0, n_boot=0, test_split=0)
    with pytest.raises(ValueError):
        pls.meancentered_pls(X, n_perm=0, n_boot=0)                 # 1 group, 1 cond
    with pytest.raises(pls.engine.PlsxError):
        pls.behavioral_pls(X, rs.rand(30, 1281), n_perm=0, n_boot=0, test_split=0)   # T' > 1280
    res = pls.behavioral_pls(X, Y, n_perm=8, n_boot=8, test_split=0, seed=3, verbose=False)
    assert res.x_weights.shape == (40, 5) and res.y_weights.shape == (5, 5)
    assert res.x_scores.shape == (30, 5) and res.y_scores.shape == (30, 5)
    assert res.singvals.shape == (5,) and res.varexp.shape == (5,)
    assert res.permres.perm_singval.shape == (5, 8) and res.permres.permsamples.shape == (30, 8)
    assert res.bootres.x_weights_normed.shape == (40, 5)
    assert res.bootres.y_loadings_boot.shape == (5, 5, 8)
    assert res.bootres.y_loadings_ci.shape == (5, 5, 2)
    assert 'PLSResults' in repr(res)


def test_default_engine_is_reused_and_rebinding_changes_nothing():
    """The public calls share ONE cached engine per device (engine.default_engine): a call after calls of other
    shapes and other methods returns bit-identical results to the first call of its kind, and to a call on a
    fresh engine; release_default_engine() frees it."""
    import pypyls_amd as pls
    from pypyls_amd import engine
    rs = np.random.RandomState(5)
    X1, Y1 = rs.randn(40, 600), rs.randn(40, 4)
    X2 = rs.randn(48, 900)
    kw = dict(n_perm=20, n_boot=20, test_split=0, seed=3, verbose=False)
    a = pls.behavioral_pls(X1, Y1, **kw)
    eng = engine.default_engine()
    pls.meancentered_pls(X2, groups=[8, 8], n_cond=3, n_perm=15, n_boot=15, seed=4, verbose=False)
    pls.pls_regression(X1, Y1, n_components=3, n_perm=10, n_boot=10, seed=5, verbose=False)
    assert engine.default_engine() is eng
    b = pls.behavioral_pls(X1, Y1, **kw)
    c = pls.behavioral_pls(X1, Y1, _engine=engine.Engine(), **kw)
    for key in ('x_weights', 'singvals', 'x_scores', 'y_loadings'):
        assert np.array_equal(a[key], b[key]) and np.array_equal(a[key], c[key]), key
    for key in ('x_weights_normed', 'y_loadings_boot', 'y_loadings_ci'):
        assert np.array_equal(a.bootres[key], b.bootres[key]) and np.array_equal(a.bootres[key], c.bootres[key]), key
    assert np.array_equal(a.permres.perm_singval, b.permres.perm_singval)
    pls.release_default_engine()
    assert engine.default_engine() is not eng


def test_results_are_ordinary_arrays_by_default(monkeypatch):
    """PLSResults holds ordinary numpy arrays (copied out of the page-locked landing zone by a few host
    threads), so that a caller who keeps many results does not accumulate unswappable memory;
    plsc.COPY_RESULTS_OUT_OF_PINNED = False hands out views of the page-locked buffers instead (no second pass).  Same
    values either way, also for arrays large enough to take the threaded copy (> 8 MB)."""
    import pypyls_amd as pls
    from pypyls_amd import plsc
    rs = np.random.RandomState(8)
    X, Y = rs.randn(40, 400000), rs.randn(40, 3)
    kw = dict(n_perm=4, n_boot=4, test_split=0, seed=2, verbose=False)
    assert plsc.COPY_RESULTS_OUT_OF_PINNED is True
    b = pls.behavioral_pls(X, Y, **kw)
    monkeypatch.setattr(plsc, 'COPY_RESULTS_OUT_OF_PINNED', False)
    a = pls.behavioral_pls(X, Y, **kw)
    assert b.x_weights.nbytes > (8 << 20)
    for key in ('x_weights',):
        assert np.array_equal(a[key], b[key]) and b[key].flags.owndata and not a[key].flags.owndata
    for key in ('x_weights_normed', 'x_weights_stderr'):
        assert np.array_equal(a.bootres[key], b.bootres[key]) and b.bootres[key].flags.owndata
    assert np.array_equal(a.bootres.y_loadings_boot, b.bootres.y_loadings_boot)


@pytest.mark.parametrize('n_perm,n_boot', [(0, 0), (7, 0), (0, 7)])
def test_calls_without_permutations_or_bootstraps(n_perm, n_boot):
    """Only the legs that were asked for run and only their results exist (BasePLS.run_pls, base.py:366-397):
    the device-resident finish must cope with an absent null / absent bootstrap block."""
    import pypyls_amd as pls
    rs = np.random.RandomState(2)
    X, Y = rs.randn(36, 400), rs.randn(36, 3)
    for res, method in ((pls.behavioral_pls(X, Y, n_perm=n_perm, n_boot=n_boot, test_split=0, seed=1, verbose=False),
                         'behavioral'),
                        (pls.meancentered_pls(X, groups=[9, 9], n_cond=2, n_perm=n_perm, n_boot=n_boot, seed=1,
                                              verbose=False), 'meancentered')):
        want = ref.run_plsc(X, Y if method == 'behavioral' else None, method=method,
                            groups=[36] if method == 'behavioral' else [9, 9], n_cond=1 if method == 'behavioral' else 2)
        keep = live_lvs(want['singvals'])
        assert_close(res.singvals[keep], want['singvals'][keep], RTOL, what='singvals')
        assert_close(np.abs(res.x_weights[:, keep]), np.abs(want['x_weights'][:, keep]), RTOL, what='x_weights')
        assert ('perm_singval' in res.permres) == (n_perm > 0)
        assert ('x_weights_normed' in res.bootres) == (n_boot > 0)
        if n_perm:
            assert res.permres.perm_singval.shape == (len(res.singvals), n_perm)
        if n_boot:
            assert res.bootres.x_weights_normed.shape == res.x_weights.shape


def _flatten(res, prefix=''):
    out = {}
    for k, v in dict(res).items():
        if k == 'inputs':
            continue
        if hasattr(v, 'keys'):
            out.update(_flatten(v, prefix + k + '.'))
        elif isinstance(v, np.ndarray) and v.dtype != object:
            out[prefix + k] = v
    return out


@pytest.mark.parametrize('method', ['behavioral', 'behavioral_cov', 'meancentered', 'regression'])
def test_results_are_bit_reproducible(method, monkeypatch):
    """the same analysis twice gives the SAME BITS in every result array -- nothing in the
    resampling path depends on the order in which concurrent threads' floating-point adds land (accumulating
    cross-product epilogue: fixed-order reduce; A-operand build: duplicates of a source row carry bit-identical
    addends; super-batch sizes: a function of the shard only).  The per-bootstrap feature pass of the unscaled modes
    (the route short series and 8-rank shards take) is forced as well as the default route."""
    import pypyls_amd as pls
    rs = np.random.RandomState(17)
    S, B, T = 72, 1500, 6
    X = rs.randn(S, B)
    Y = rs.randn(S, T) + 0.4 * X[:, :T]
    kw = dict(n_perm=64, n_boot=64, seed=99, verbose=False)

    def call():
        if method == 'behavioral':
            return pls.behavioral_pls(X, Y, groups=[36, 36], n_split=5, test_split=10, **kw)
        if method == 'behavioral_cov':
            return pls.behavioral_pls(X, Y, groups=[36, 36], covariance=True, test_split=0, **kw)
        if method == 'meancentered':
            return pls.meancentered_pls(X, groups=[18, 18], n_cond=2, n_split=5, **kw)
        return pls.pls_regression(X, Y, n_components=4, **kw)
    for quad in ('0', '-1'):
        monkeypatch.setenv('PLSX_QUAD_SUMS', quad)
        from pypyls_amd import engine
        engine.release_default_engine()
        eng = engine.default_engine()
        eng.set_option('quad_sums', int(quad))
        a, b = _flatten(call()), _flatten(call())
        assert set(a) == set(b) and len(a) >= 8
        for k in sorted(a):
            assert np.array_equal(a[k], b[k], equal_nan=True), (method, quad, k, np.nanmax(np.abs(a[k] - b[k])))
    engine.release_default_engine()


@pytest.mark.parametrize('tag', ['a', 'b'])
@pytest.mark.parametrize('two_pass', [False, True])
def test_mpls_bootstrap_sums_against_the_reference_seed_envelope(tag, two_pass):
    """The device's rotated bootstrap vectors of mean-centred PLS (rank-deficient Procrustes) against
    tests/golden/seeds_mpls.npz: the reference's own BasePLS._single_boot for analysis seeds 0 .. 5 on six bootstraps
    (tests/test_oracle.py::test_mpls_oracle_against_the_reference_seed_envelope has the numbers).  The kernels return
    the SUM of the rotated vectors: it equals the oracle's sum tightly on the live LVs (both routes of the unscaled
    bootstrap) and lies as close to every reference run as the reference runs lie to each other."""
    from test_oracle import _mpls_envelope
    from pypyls_amd.engine import Engine
    g, spec, live, runs, mine, dist, _ = _mpls_envelope(tag)
    X = g[tag + '_X']
    eng = Engine(options={'two_pass_boot': 1} if two_pass else None)
    try:
        from pypyls_amd import resampling as rsmp
        eng.set_data(X, None, rsmp.cell_of_row(spec.groups, spec.n_cond), len(spec.groups), spec.n_cond, 1,
                     mean_centering=spec.mean_centering)
        sv = g[tag + '_singvals']
        d0 = np.diag(sv) if sv.ndim == 2 else sv
        yw = ref.decompose(spec, X, spec.dummy)[2]
        eng.set_original(g[tag + '_x_weights'], d0, yw)
        usum, usq, _ = eng.boot(g[tag + '_bootsamples'])
        usum = usum.cpu().numpy()
    finally:
        eng.close()
    want = mine.sum(axis=-1)                                                       # oracle: sum over the six bootstraps
    for k in np.flatnonzero(live):
        assert np.abs(usum[:, k] - want[:, k]).max() <= 1e-9 * np.abs(want[:, k]).max(), k
    sums = runs.sum(axis=-1)                                                       # (seed, B, L)
    n = len(sums)
    spread = np.max([dist(sums[a], sums[b]) for a in range(n) for b in range(a + 1, n)], axis=0)
    od = np.max([dist(s, usum) for s in sums], axis=0)
    # measured with the oracle's sums: 1.43 / 1.51 / 0.98 (a), 1.17 / 1.70 (b) x the runs' own largest distance -- the
    # seed noise of single vectors partly averages out of a sum, the part that comes from the null columns of the
    # ORIGINAL (fixed within an analysis, different between seeds) does not
    assert np.all(od[live] <= 2.0 * spread[live]), (od[live], spread[live])
    assert spread[live].max() > 5e-3                       # percents: there is no 1e-5 reference answer to pin here


def test_verbose_prints_the_references_progress_bars(monkeypatch, capfd):
    """``verbose=True`` (the reference's default) drives tqdm bars named like the reference's (pyls/utils.py:128-152,
    pyls/base.py:484, 641): here they count resamples the DEVICE has finished (an event per asynchronous chunk) and
    appear only when a leg runs longer than progress.DELAY_S -- forced to 0 for the test.  verbose=False: silence."""
    import pypyls_amd as pls
    from pypyls_amd import progress
    rs = np.random.RandomState(3)
    X, Y = rs.randn(60, 4000), rs.randn(60, 4)
    monkeypatch.setattr(progress, 'DELAY_S', 0.0)
    quiet = pls.behavioral_pls(X, Y, n_perm=600, n_boot=600, n_split=2, test_split=0, seed=5, verbose=False)
    err = capfd.readouterr().err
    assert 'Running' not in err
    loud = pls.behavioral_pls(X, Y, n_perm=600, n_boot=600, n_split=2, test_split=0, seed=5, verbose=True)
    err = capfd.readouterr().err
    for name in ('Running permutations', 'Running bootstraps', 'Running split-half'):
        assert name in err, (name, err[-400:])
    assert '/600' in err
    assert np.array_equal(quiet.permres.perm_singval, loud.permres.perm_singval)
    assert np.array_equal(quiet.bootres.x_weights_normed, loud.bootres.x_weights_normed)
    rr = pls.pls_regression(X, Y, n_components=2, n_perm=300, n_boot=300, seed=5, verbose=True)
    err = capfd.readouterr().err
    assert 'Running permutations' in err and 'Running bootstraps' in err and rr.varexp.shape == (2,)


def test_cached_engine_is_released_after_an_idle_period(monkeypatch):
    """the cached default engine keeps X and the scratch mapped between calls (re-mapping costs
    25 ms per GB) -- but not for ever: engine.IDLE_RELEASE_S seconds after the last call ended it is released, and
    the next call builds a new one with the same results."""
    import time
    import pypyls_amd as pls
    from pypyls_amd import engine
    rs = np.random.RandomState(4)
    X, Y = rs.randn(30, 500), rs.randn(30, 3)
    kw = dict(n_perm=6, n_boot=6, test_split=0, seed=9, verbose=False)
    monkeypatch.setattr(engine, 'IDLE_RELEASE_S', 0.4)
    a = pls.behavioral_pls(X, Y, **kw)
    eng = engine.default_engine()
    assert eng.ctx
    engine.touch_idle_release()
    for _ in range(50):
        if not eng.ctx:
            break
        time.sleep(0.1)
    assert not eng.ctx and not engine._DEFAULT
    b = pls.behavioral_pls(X, Y, **kw)
    assert engine.default_engine() is not eng
    assert np.array_equal(a.permres.perm_singval, b.permres.perm_singval)
    assert np.array_equal(a.bootres.x_weights_normed, b.bootres.x_weights_normed)
    monkeypatch.setattr(engine, 'IDLE_RELEASE_S', 0)       # (no timer left behind for the tests that follow)
    engine.touch_idle_release()
