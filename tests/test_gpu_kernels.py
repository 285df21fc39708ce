"""Kernel-level parity (through the C ABI) against the oracle. GPU only."""
import numpy as np
import pytest

from conftest import assert_close
from oracle import cpu_ref as ref

pytestmark = pytest.mark.gpu


def _engine():
    from pypyls_amd.engine import Engine, options_from_env
    return Engine(**options_from_env())          # PLSX_<KEY>=1 (monkeypatched per test) -> plsx_set_option


def _data(S, B, T, seed=0, signal=0.5):
    rs = np.random.RandomState(seed)
    X = rs.randn(S, B) + 3.0 * rs.rand(1, B)
    Y = rs.randn(S, T) + 1.5
    k = min(T, B)
    Y[:, :k] += signal * X[:, :k]
    return X, Y, rs


def _setup(eng, X, Y, groups, n_cond, method='behavioral', covariance=False, mc=0):
    from pypyls_amd import resampling as rsmp
    eng.set_data(X, Y if method == 'behavioral' else None, rsmp.cell_of_row(groups, n_cond),
                 len(groups), n_cond, 0 if method == 'behavioral' else 1,
                 mean_centering=mc, covariance=covariance)
    return ref.Spec(method, groups, n_cond, covariance, mc)


CASES = [
    dict(S=40, B=300, T=6, groups=[40], n_cond=1),
    dict(S=46, B=130, T=4, groups=[11, 12], n_cond=2),
    dict(S=36, B=70, T=5, groups=[12], n_cond=3, covariance=True),
    dict(S=20, B=3, T=3, groups=[20], n_cond=1),
    dict(S=30, B=7, T=5, groups=[15, 15], n_cond=1),          # T' = 10 > B = 7
    dict(S=81, B=1000, T=10, groups=[81], n_cond=1),
]


@pytest.mark.parametrize('case', CASES)
def test_crosscov_and_decompose_behavioral(case):
    case = dict(case)
    cov = case.pop('covariance', False)
    X, Y, rs = _data(case['S'], case['B'], case['T'])
    eng = _engine()
    spec = _setup(eng, X, Y, case['groups'], case['n_cond'], covariance=cov)
    from pypyls_amd import resampling as rsmp
    R0 = eng.crosscov(n=1)[0]
    assert_close(R0, ref.gen_covcorr(spec, X, Y, spec.dummy), 1e-10, what='R original')
    perms = rsmp.gen_permsamp(case['groups'], case['n_cond'], 5, seed=1)
    boots = rsmp.gen_bootsamp(case['groups'], case['n_cond'], 5, seed=2)
    Rp = eng.crosscov(ysrc=perms)
    Rb = eng.crosscov(xsrc=boots, ysrc=boots)
    for i in range(5):
        assert_close(Rp[i], ref.gen_covcorr(spec, X, Y[perms[:, i]], spec.dummy), 1e-10, what='R perm')
        assert_close(Rb[i], ref.gen_covcorr(spec, X[boots[:, i]], Y[boots[:, i]], spec.dummy),
                     1e-10, what='R boot')
    xw, sv, yw = eng.decompose()
    U, d, V = ref.decompose(spec, X, Y)
    assert_close(sv, np.diag(d), 1e-9, what='singvals')
    sgn = np.sign(np.sum(xw * U, axis=0))
    assert_close(xw * sgn, U, 1e-7, what='x_weights')
    assert_close(yw * sgn, V, 1e-7, what='y_weights')


@pytest.mark.parametrize('case', CASES)
def test_perm_and_boot_behavioral(case):
    case = dict(case)
    cov = case.pop('covariance', False)
    X, Y, rs = _data(case['S'], case['B'], case['T'], seed=3)
    eng = _engine()
    spec = _setup(eng, X, Y, case['groups'], case['n_cond'], covariance=cov)
    from pypyls_amd import resampling as rsmp
    U, d, V = ref.decompose(spec, X, Y)
    eng.set_original(U, np.diag(d), V)
    perms = rsmp.gen_permsamp(case['groups'], case['n_cond'], 9, seed=1)
    got = eng.perm(perms, rotate=True)
    want = np.stack([ref.single_perm(spec, X, Y, perms[:, i], V)[0] for i in range(9)], -1)
    assert_close(got, want, 1e-8, what='rotated perm singvals')
    spec.rotate = False
    got = eng.perm(perms, rotate=False)
    want = np.stack([ref.single_perm(spec, X, Y, perms[:, i], V)[0] for i in range(9)], -1)
    assert_close(got, want, 1e-8, what='raw perm singvals')
    boots = rsmp.gen_bootsamp(case['groups'], case['n_cond'], 9, seed=2)
    usum, usq, dist = eng.boot(boots)
    ws, wq, wd = np.zeros_like(U), np.zeros_like(U), []
    for i in range(9):
        dd, ub = ref.single_boot(spec, X, Y, boots[:, i], U, d)
        ws += ub
        wq += ub ** 2
        wd.append(dd)
    assert_close(usum.cpu().numpy(), ws, 1e-7, what='u_sum')
    assert_close(usq.cpu().numpy(), wq, 1e-7, what='u_square')
    assert_close(dist, np.stack(wd, -1), 1e-7, what='distrib')


@pytest.mark.parametrize('mc', [0, 1, 2])
def test_meancentered(mc):
    rs = np.random.RandomState(5)
    groups, n_cond = [8, 9, 10], 2
    X = rs.randn(54, 400)
    X[:18] += 0.8
    X[27:] -= 0.5
    eng = _engine()
    spec = _setup(eng, X, None, groups, n_cond, method='meancentered', mc=mc)
    from pypyls_amd import resampling as rsmp
    Y = spec.dummy.astype(float)
    assert_close(eng.crosscov(n=1)[0], ref.gen_covcorr(spec, X, Y, spec.dummy), 1e-10, what='R')
    U, d, V = ref.decompose(spec, X, Y)
    live = ref.live_lvs(d)
    xw, sv, yw = eng.decompose()
    assert_close(sv[live], np.diag(d)[live], 1e-9, what='singvals')
    eng.set_original(U, np.diag(d), V)
    perms = rsmp.gen_permsamp(groups, n_cond, 7, seed=1)
    got = eng.perm(perms)
    want = np.stack([ref.single_perm(spec, X, Y, perms[:, i], V)[0] for i in range(7)], -1)
    assert_close(got[live], want[live], 1e-8, what='perm')
    boots = rsmp.gen_bootsamp(groups, n_cond, 7, seed=2)
    usum, usq, dist = eng.boot(boots)
    ws, wd = np.zeros_like(U), []
    for i in range(7):
        dd, ub = ref.single_boot(spec, X, Y, boots[:, i], U, d)
        ws += ub
        wd.append(dd)
    assert_close(usum.cpu().numpy()[:, live], ws[:, live], 1e-7, what='u_sum')
    assert_close(dist[:, live], np.stack(wd, -1)[:, live], 1e-7, what='distrib')


@pytest.mark.parametrize('case', [
    dict(S=40, B=300, T=6, groups=[40], n_cond=1),
    dict(S=46, B=130, T=4, groups=[11, 12], n_cond=2),
    dict(S=36, B=70, T=5, groups=[12], n_cond=3, covariance=True),
])
def test_split_half_behavioral(case):
    case = dict(case)
    cov = case.pop('covariance', False)
    X, Y, rs = _data(case['S'], case['B'], case['T'], seed=7)
    eng = _engine()
    spec = _setup(eng, X, Y, case['groups'], case['n_cond'], covariance=cov)
    from pypyls_amd import resampling as rsmp
    ns = 5
    # original arrangement
    masks = rsmp.gen_splits(case['groups'], case['n_cond'], ns, seed=3)
    U, d, V = ref.decompose(spec, X, Y)
    di = np.linalg.inv(d)
    want_u = np.zeros((U.shape[1], ns))
    want_v = np.zeros((U.shape[1], ns))
    for i in range(ns):
        u, v = ref.split_half(spec, X, Y, U @ di, V @ di, masks[:, [i]])
        want_u[:, i], want_v[:, i] = u, v
    uc, vc = eng.split_half(masks)
    assert_close(uc[0], want_u, 1e-7, what='ucorr original')
    assert_close(vc[0], want_v, 1e-7, what='vcorr original')
    # permuted arrangements, each with its own masks
    perms = rsmp.gen_permsamp(case['groups'], case['n_cond'], 3, seed=1)
    pm = np.stack([rsmp.gen_splits(case['groups'], case['n_cond'], ns, seed=10 + i) for i in range(3)])
    uc, vc = eng.split_half(pm, perms=perms)
    for p in range(3):
        Xp, Yp = ref.make_permutation(spec, X, Y, perms[:, p])
        U, d, V = ref.decompose(spec, Xp, Yp)
        di = np.linalg.inv(d)
        for i in range(ns):
            u, v = ref.split_half(spec, Xp, Yp, U @ di, V @ di, pm[p][:, [i]])
            assert_close(uc[p][:, i], u, 1e-7, what='ucorr perm')
            assert_close(vc[p][:, i], v, 1e-7, what='vcorr perm')


def test_split_half_meancentered():
    rs = np.random.RandomState(5)
    groups, n_cond = [11, 13], 2
    X = rs.randn(48, 300)
    X[:20] += 0.7
    eng = _engine()
    spec = _setup(eng, X, None, groups, n_cond, method='meancentered', mc=0)
    from pypyls_amd import resampling as rsmp
    Y = spec.dummy.astype(float)
    ns = 4
    perms = rsmp.gen_permsamp(groups, n_cond, 2, seed=1)
    pm = np.stack([rsmp.gen_splits(groups, n_cond, ns, seed=20 + i) for i in range(2)])
    uc, vc = eng.split_half(pm, perms=perms)
    for p in range(2):
        Xp, Yp = ref.make_permutation(spec, X, Y, perms[:, p])
        U, d, V = ref.decompose(spec, Xp, Yp)
        live = ref.live_lvs(d)
        dl = np.diag(d)[live]
        ud, vd = U[:, live] / dl, V[:, live] / dl
        for i in range(ns):
            u, v = ref.split_half(spec, Xp, Yp, ud, vd, pm[p][:, [i]])
            assert_close(uc[p][live, i], u, 1e-7, what='ucorr')
            assert_close(vc[p][live, i], v, 1e-7, what='vcorr')


@pytest.mark.parametrize('case', [
    dict(S=40, B=300, T=6, groups=[40], n_cond=1),
    dict(S=46, B=130, T=4, groups=[11, 12], n_cond=2),
    dict(S=36, B=70, T=5, groups=[12], n_cond=3, covariance=True),
    dict(S=30, B=7, T=5, groups=[15, 15], n_cond=1),
    dict(S=48, B=500, groups=[8, 8], n_cond=3, method='meancentered', mc=0),
    dict(S=48, B=500, groups=[8, 8], n_cond=3, method='meancentered', mc=1),
    dict(S=48, B=500, groups=[8, 8], n_cond=3, method='meancentered', mc=2),
])
@pytest.mark.parametrize('rotate', [True, False])
def test_dual_perm_path_equals_feature_pass(case, rotate, monkeypatch):
    """The S x S kernel formulation of the permutation test (G_p = A_p K A_p^T)
    against the O(B) pass over the features (PLSX_NO_DUAL_PERM=1), and against
    the oracle's single_perm."""
    from pypyls_amd import resampling as rsmp, hostmath
    case = dict(case)
    method = case.pop('method', 'behavioral')
    cov, mc = case.pop('covariance', False), case.pop('mc', 0)
    X, Y, rs = _data(case['S'], case['B'], case.get('T', 3), seed=5)
    perms = rsmp.gen_permsamp(case['groups'], case['n_cond'], 9, seed=3)
    got = {}
    Yo = Y if method == 'behavioral' else None
    for flag in ('0', '1'):
        monkeypatch.setenv('PLSX_NO_DUAL_PERM', flag)
        eng = _engine()
        spec = _setup(eng, X, Yo, case['groups'], case['n_cond'], method=method, covariance=cov, mc=mc)
        U, d, V = ref.decompose(spec, X, Yo if Yo is not None else spec.dummy)
        eng.set_original(U, np.diag(d), V)
        assert bool(eng.last_timing()['dual_perm']) == (flag == '0')
        got[flag] = eng.perm(perms, rotate=rotate)
    live = ref.live_lvs(np.diag(d))
    assert_close(got['0'][live], got['1'][live], 1e-9, what='dual vs feature pass')
    spec.rotate = rotate
    want = np.stack([ref.single_perm(spec, X, Yo if Yo is not None else spec.dummy, perms[:, i], V)[0]
                     for i in range(perms.shape[1])], -1)
    assert_close(got['0'][live], want[live], 1e-7, what='dual vs oracle')


@pytest.mark.parametrize('shape', [(60, 4000, 13, [60], 1), (60, 2500, 10, [15, 15], 2), (64, 3001, 50, [64], 1),
                                   (40, 300, 6, [40], 1)])
def test_bootstrap_kernel_variants_agree(shape, monkeypatch):
    """The 4x4x4-MFMA Gram kernel vs the 16x16x4 one (PLSX_NO_GRAM4), the
    software-pipelined rotation kernel vs the generic one (PLSX_UROT_GENERIC, bit for bit when
    both multiply the last tile of L on 16x16x4: PLSX_UROT_NO_TAIL4), and the rotation kernel
    with that tile on the 4x4x4 shape (L mod 16 in 1..4: T' = 50, 20) vs without:
    same bootstraps, same sums."""
    from pypyls_amd import resampling as rsmp
    S, B, T, groups, n_cond = shape
    X, Y, rs = _data(S, B, T, seed=11)
    boots = rsmp.gen_bootsamp(groups, n_cond, 12, seed=4)
    got = {}
    for key, env in (('default', {}), ('gram16', {'PLSX_NO_GRAM4': '1'}), ('urot_generic', {'PLSX_UROT_GENERIC': '1'}),
                     ('no_tail4', {'PLSX_UROT_NO_TAIL4': '1'})):
        for k in ('PLSX_NO_GRAM4', 'PLSX_UROT_GENERIC', 'PLSX_UROT_NO_TAIL4'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = _engine()
        spec = _setup(eng, X, Y, groups, n_cond)
        U, d, V = ref.decompose(spec, X, Y)
        eng.set_original(U, np.diag(d), V)
        usum, usq, dist = eng.boot(boots)
        got[key] = (usum.cpu().numpy(), usq.cpu().numpy(), dist)
    for a, b in zip(got['no_tail4'], got['urot_generic']):
        assert np.array_equal(a, b)                       # same arithmetic, same order
    for a, b in zip(got['default'], got['no_tail4']):
        assert_close(a, b, 1e-12, what='4x4x4 tail tile vs 16x16x4')
    for a, b in zip(got['default'], got['gram16']):
        assert_close(a, b, 1e-9, what='gram4 vs gram16')


@pytest.mark.parametrize('shape', [(60, 900, 5, [60], 1), (72, 500, 4, [18, 18], 2)])
def test_split_half_fused_equals_two_pass(shape, monkeypatch):
    """Fused split-half (second half = full sample - first half, one MFMA pass
    per split) against the two-pass path (PLSX_NO_SPLIT_FUSE), original and
    permuted arrangements."""
    from pypyls_amd import resampling as rsmp
    S, B, T, groups, n_cond = shape
    X, Y, rs = _data(S, B, T, seed=21)
    n_split = 9
    perms = rsmp.gen_permsamp(groups, n_cond, 3, seed=5)
    masks = np.stack([rsmp.gen_splits(groups, n_cond, n_split, seed=30 + i) for i in range(3)])   # (P, S, ns)
    got = {}
    for key in ('fused', 'two_pass'):
        monkeypatch.delenv('PLSX_NO_SPLIT_FUSE', raising=False)
        if key == 'two_pass':
            monkeypatch.setenv('PLSX_NO_SPLIT_FUSE', '1')
        eng = _engine()
        _setup(eng, X, Y, groups, n_cond)
        got[key] = (eng.split_half(masks[0]), eng.split_half(masks, perms=perms))
    for a, b in zip(got['fused'], got['two_pass']):
        for x, y in zip(a, b):
            assert_close(x, y, 1e-9, what='fused vs two-pass split-half')


@pytest.mark.parametrize('shape', [(64, 3001, 50, [64], 1), (60, 2500, 10, [15, 15], 2), (90, 1300, 7, [10, 12, 8], 3)])
def test_separate_moments_layout_equals_in_block(shape):
    """Correlation mode: data-only cross-product blocks scaled from the table that moment-only blocks
    write (k_xprod EPI 3 / 4; chosen per launch by tile passes, here forced: PLSX_SEPMOM_ALWAYS) against
    in-block moment rows (PLSX_INBLOCK_MOMENTS) and the oracle -- R itself, bootstraps, and the two-pass
    split-half route (masked resamples: rows with xsrc = -1).  Own processes (historical: the switches used to be read once per process)."""
    import subprocess
    import sys
    import json
    from conftest import ROOT
    code = r"""
import sys, json, numpy as np
sys.path.insert(0, %r)
from pypyls_amd import resampling as rsmp
from pypyls_amd.engine import Engine, options_from_env
S, B, T, groups, n_cond = %r
rs = np.random.RandomState(3)
X = rs.randn(S, B) * (0.5 + rs.rand(1, B)); Y = rs.randn(S, T) + 0.4 * X[:, :T]
eng = Engine(**options_from_env())
eng.set_data(X, Y, rsmp.cell_of_row(groups, n_cond), len(groups), n_cond, 0)
boots = rsmp.gen_bootsamp(groups, n_cond, 45, seed=4)
R = eng.crosscov(xsrc=boots, ysrc=boots)
xw, sv, yw = eng.decompose()
eng.set_original(xw, sv, yw)
usum, usq, dist = eng.boot(boots)
masks = rsmp.gen_splits(groups, n_cond, 6, seed=9)
uc, vc = eng.split_half(masks)
np.savez(sys.argv[1], R=R, usum=usum.cpu().numpy(), usq=usq.cpu().numpy(), dist=dist, uc=uc, vc=vc, sv=sv)
""" % (ROOT, shape)
    import os
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for key, env in (('sep', {'PLSX_SEPMOM_ALWAYS': '1', 'PLSX_NO_SPLIT_FUSE': '1', 'PLSX_NO_COMPACT_BOOT': '1'}),
                         ('inblock', {'PLSX_INBLOCK_MOMENTS': '1', 'PLSX_NO_SPLIT_FUSE': '1',
                                      'PLSX_NO_COMPACT_BOOT': '1'})):
            path = os.path.join(tmp, key + '.npz')
            proc = subprocess.run([sys.executable, '-c', code, path], env=dict(os.environ, **env),
                                  capture_output=True, text=True, timeout=600)
            assert proc.returncode == 0, proc.stderr[-2000:]
            out[key] = dict(np.load(path))
    for k in ('R', 'sv', 'usum', 'usq', 'dist', 'uc', 'vc'):
        assert_close(out['sep'][k], out['inblock'][k], 1e-11, what='separate vs in-block moments: ' + k)
    # and R against the oracle
    from pypyls_amd import resampling as rsmp
    S, B, T, groups, n_cond = shape
    rs = np.random.RandomState(3)
    X = rs.randn(S, B) * (0.5 + rs.rand(1, B))
    Y = rs.randn(S, T) + 0.4 * X[:, :T]
    boots = rsmp.gen_bootsamp(groups, n_cond, 45, seed=4)
    spec = ref.Spec('behavioral', groups, n_cond)
    for i in (0, 17, 44):
        want = ref.gen_covcorr(spec, X[boots[:, i]], Y[boots[:, i]], spec.dummy)
        assert_close(out['sep']['R'][i], want, 1e-10, what='separate-moments R vs oracle')


@pytest.mark.parametrize('shape', [(64, 3001, 50, [64], 1), (60, 2500, 10, [15, 15], 2), (90, 1300, 7, [10, 12, 8], 3),
                                   (60, 800, 20, [60], 1), (40, 700, 8, [40], 1), (160, 900, 33, [80], 2), (96, 700, 3, [6, 6, 6, 6], 4),
                                   (130, 900, 100, [130], 1), (150, 800, 60, [75], 2), (200, 600, 180, [200], 1), (240, 500, 200, [240], 1)])
def test_compact_blocks_equal_dense_blocks(shape):
    """Compact cross-product blocks -- one bootstrap per block contracting over the DISTINCT rows it draws
    (k_xprod IDX row table, multiplicities folded into A; last tile on the 4x4x4 shape when it holds <= 4
    rows: T' = 50, 20, 100; bootstraps up to T' = 208 = 13 tiles, split-half up to 64) -- and one split per block over its first half
    (PLSX_SPLIT_INBLOCK = the 7-per-block fused layout) against the dense layouts: bootstrap sums, distrib,
    split-half correlations.  Every tile count 1..4, with and without the tail, J = 1..16 cells (16 cells:
    compact bootstraps, but the split-half epilogue's column tables no longer fit: dense fused layout).
    Own processes (historical: the switches used to be read once per process)."""
    import json
    import os
    import subprocess
    import sys
    import tempfile
    from conftest import ROOT
    code = r"""
import sys, json, numpy as np
sys.path.insert(0, %r)
from pypyls_amd import resampling as rsmp
from pypyls_amd.engine import Engine, options_from_env
S, B, T, groups, n_cond = %r
rs = np.random.RandomState(3)
X = rs.randn(S, B) * (0.5 + rs.rand(1, B)); Y = rs.randn(S, T) + 0.4 * X[:, :T]
eng = Engine(**options_from_env())
eng.set_data(X, Y, rsmp.cell_of_row(groups, n_cond), len(groups), n_cond, 0)
boots = rsmp.gen_bootsamp(groups, n_cond, 45, seed=4)
xw, sv, yw = eng.decompose()
eng.set_original(xw, sv, yw)
usum, usq, dist = eng.boot(boots)
cf = eng.last_timing().get('compact_row_fraction', 0.0)
masks = rsmp.gen_splits(groups, n_cond, 11, seed=9)
perms = rsmp.gen_permsamp(groups, n_cond, 2, seed=6)
uc, vc = eng.split_half(masks)
m3 = np.stack([rsmp.gen_splits(groups, n_cond, 5, seed=40 + i) for i in range(2)])
ucp, vcp = eng.split_half(m3, perms=perms)
# lopsided masks (any mask is legal through the C ABI): a first half of nearly all rows, one of three rows
# per cell, one that misses a cell entirely (NaN, as the dense layout gives)
cells = rsmp.cell_of_row(groups, n_cond)
odd = np.array(masks[:, :3], dtype=bool)
odd[:, 0] = True
odd[:, 1] = False
for j in np.unique(cells):
    rows = np.flatnonzero(cells == j)
    odd[rows[:2], 0] = False
    odd[rows[:3], 1] = True
    if j == 0:
        odd[rows, 2] = False
uco, vco = eng.split_half(odd)
# bootstraps that draw very few distinct rows (every cell: two subjects, over and over)
few = np.array(boots[:, :6])
for j in np.unique(cells):
    rows = np.flatnonzero(cells == j)
    few[rows, :] = rows[:2][np.arange(len(rows))[:, None] %% 2 * np.ones((1, 6), dtype=int)]
usum2, usq2, dist2 = eng.boot(few)
np.savez(sys.argv[1], usum=usum.cpu().numpy(), usq=usq.cpu().numpy(), dist=dist, uc=uc, vc=vc, ucp=ucp, vcp=vcp,
         compact=cf, uco=uco, vco=vco, usum2=usum2.cpu().numpy(), usq2=usq2.cpu().numpy(), dist2=dist2)
""" % (ROOT, shape)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for key, env in (('compact', {'PLSX_COMPACT_BOOT_ALWAYS': '1'}),
                         ('dense', {'PLSX_NO_COMPACT_BOOT': '1', 'PLSX_SPLIT_INBLOCK': '1'})):
            path = os.path.join(tmp, key + '.npz')
            proc = subprocess.run([sys.executable, '-c', code, path], env=dict(os.environ, **env),
                                  capture_output=True, text=True, timeout=600)
            assert proc.returncode == 0, proc.stderr[-2000:]
            out[key] = dict(np.load(path))
    S, B, T, groups, n_cond = shape
    if T * len(groups) * n_cond <= 208:
        assert 0.5 < float(out['compact']['compact']) < 0.8, 'the compact layout did not run'
    assert float(out['dense']['compact']) == 0.0
    for k in ('usum', 'usq', 'dist', 'uc', 'vc', 'ucp', 'vcp'):
        assert_close(out['compact'][k], out['dense'][k], 1e-10, what='compact vs dense blocks: ' + k)
    for k in ('uco', 'vco', 'usum2', 'usq2', 'dist2'):
        a, b = out['compact'][k], out['dense'][k]
        assert np.array_equal(np.isnan(a), np.isnan(b)), 'NaN pattern of ' + k
        ok = ~np.isnan(b)
        if ok.any():
            scale = np.max(np.abs(b[ok]))
            assert np.max(np.abs(a[ok] - b[ok])) <= 1e-8 * scale, 'lopsided resamples, compact vs dense: ' + k


def test_device_side_finishing_steps():
    """The pieces the device-resident front-end finishes with (round 4), each against numpy: the sign convention
    of compute.svd (plsx_svd_flip vs hostmath.sign_convention, both branches T' <= B and T' > B), column scaling,
    the (n, T' L) -> (T' L, n) transpose, the mean over splits, percentile intervals of device series."""
    import torch
    from pypyls_amd import resampling as rsmp, hostmath
    for S, B, T in ((40, 700, 6), (30, 5, 9)):              # T' <= B: flip on x_weights; T' > B: on y_weights
        X, Y, rs = _data(S, B, T, seed=3)
        eng = _engine()
        _setup(eng, X, Y, [S], 1)
        xw, sv, yw = eng.decompose_dev()
        want_x, want_y = hostmath.sign_convention(xw.cpu().numpy(), yw.cpu().numpy())
        eng.svd_flip(xw, yw)
        eng.sync()
        assert np.array_equal(xw.cpu().numpy(), want_x) and np.array_equal(yw.cpu().numpy(), want_y)
        lead = want_x if T <= B else want_y
        assert (lead[np.argmax(np.abs(lead), axis=0), np.arange(lead.shape[1])] > 0).all()
        sc = eng.scale_columns(xw, sv)
        assert np.array_equal(sc.cpu().numpy(), want_x * sv.cpu().numpy()[None, :])
    A = torch.rand((37, 91), dtype=torch.float64, device=eng.device)
    assert np.array_equal(eng.transpose_dev(A).cpu().numpy(), A.cpu().numpy().T)
    C = torch.rand((5, 7, 3), dtype=torch.float64, device=eng.device)
    C[2, 4, 1] = float('nan')
    out = torch.zeros((5, 3), dtype=torch.float64, device=eng.device)
    eng.mean_splits_into(C, out)
    want = C.cpu().numpy().mean(axis=1)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-15, equal_nan=True)
    series = torch.randn((11, 1000), dtype=torch.float64, device=eng.device)
    lo, hi = eng.percentile_ci_dev(series, ci=95)
    wl, wh = np.percentile(series.cpu().numpy(), [2.5, 97.5], axis=-1)
    assert np.array_equal(lo.cpu().numpy(), wl) and np.array_equal(hi.cpu().numpy(), wh)


@pytest.mark.parametrize('shape', [
    (120, 1235, 50, [120], 1),          # the headline T' = 50 (13 row blocks), B not a multiple of 16
    (64, 1000, 25, [32, 32], 1),        # T' = 2 cells x 25 = 50: per-cell column constants
    (90, 700, 18, [90], 1),             # T' = 18 (5 row blocks)
    (96, 520, 12, [16, 16], 3),         # T' = 6 cells x 12 = 72: outside the reader's shapes -> the two-kernel route (asserted)
    (108, 900, 12, [36], 3),            # T' = 3 cells x 12 = 36 (9 row blocks)
    (130, 40000, 52, [130], 1),         # T' = 52: no padding row in the last block; several column chunks
    (70, 640, 11, [35], 2),             # T' = 22 (6 row blocks): the last tile of L holds 6 LVs -> 16x16x4, no 4x4x4 tail
    (80, 800, 7, [20], 4),              # T' = 28 (7 row blocks), four cells
    (96, 500, 32, [96], 1),             # T' = 32 (8 row blocks): two full tiles of L
    (100, 1100, 13, [25, 25], 2),       # T' = 4 cells x 13 = 52 -> 13 blocks with cells; (below) 10, 11, 12 row blocks
    (90, 600, 39, [90], 1),             # T' = 39 (10 row blocks)
    (120, 750, 14, [40], 3),            # T' = 42 (11 row blocks)
    (110, 900, 47, [110], 1),           # T' = 47 (12 row blocks)
])
def test_split_half_one_pass_reader(shape, monkeypatch):
    """Split-half with ONE reader pass over the raw first-half sums (k_xprod_compact epilogue 8 + k_split_fused) against
    the two-kernel route over both z-scored halves (option split_two_readers) and against the oracle: original and
    permuted arrangements, pre-permuted Y stacks, an odd number of splits (the last pair of a block is one split)."""
    from pypyls_amd import resampling as rsmp
    S, B, T, groups, n_cond = shape
    X, Y, rs = _data(S, B, T, seed=31)
    n_split, n_arr = 7, 2
    perms = rsmp.gen_permsamp(groups, n_cond, n_arr, seed=5)
    masks = np.stack([rsmp.gen_splits(groups, n_cond, n_split, seed=40 + i) for i in range(1 + n_arr)])
    Tp = len(groups) * n_cond * T
    expect_route = 1 if 5 <= -(-Tp // 4) <= 13 and len(groups) * n_cond <= 7 else 0
    got = {}
    for key in ('one_pass', 'two_readers'):
        monkeypatch.delenv('PLSX_SPLIT_TWO_READERS', raising=False)
        if key == 'two_readers':
            monkeypatch.setenv('PLSX_SPLIT_TWO_READERS', '1')
        eng = _engine()
        spec = _setup(eng, X, Y, groups, n_cond)
        a = eng.split_half(masks[0])
        assert eng.split_route() == (expect_route if key == 'one_pass' else 0)
        b = eng.split_half(masks[1:], perms=perms)
        c = eng.split_half(masks[1:], ystack=np.stack([Y[perms[:, p]] for p in range(n_arr)]))
        got[key] = (a, b, c)
    for a, b in zip(got['one_pass'], got['two_readers']):
        for x, y in zip(a, b):
            assert_close(x, y, 1e-9, what='one reader pass vs two readers')
    for x, y in zip(got['one_pass'][1], got['one_pass'][2]):
        assert_close(x, y, 1e-12, what='index permutation vs pre-permuted Y')
    uc, vc = got['one_pass'][1]
    for p in range(n_arr):
        Xp, Yp = ref.make_permutation(spec, X, Y, perms[:, p])
        U, d, V = ref.decompose(spec, Xp, Yp)
        di = np.linalg.inv(d)
        for i in (0, n_split - 1):
            u, v = ref.split_half(spec, Xp, Yp, U @ di, V @ di, masks[1 + p][:, [i]])
            assert_close(uc[p][:, i], u, 1e-7, what='ucorr vs oracle')
            assert_close(vc[p][:, i], v, 1e-7, what='vcorr vs oracle')
