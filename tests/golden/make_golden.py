#!/usr/bin/env python3
"""
Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (``/root/reference`` does not exist on the GPU
box):      python tests/golden/make_golden.py

What it does
  * puts an empty stub module named ``h5py`` on sys.path (pyls/io.py:6 imports
    it; HDF5 persistence is not on the hot path) and imports
    ``/root/reference/pyls``;
  * always calls the reference with ``permindices=True`` (documented default,
    pyls/structures.py:115-120; the shipped default ``None`` breaks
    BasePLS._single_perm), ``verbose=False`` and
    ``test_split=0``;
  * records every split mask the reference draws inside
    ``BasePLS.split_half`` (pyls/base.py:738-742) by wrapping
    ``pyls.base.gen_splits``, so the oracle and the product can be fed the
    very same masks;
  * stores inputs + outputs as small ``.npz`` files (data only -- no reference
    source text).

Fixtures hold DATA: the inputs, the resampling arrays and the reference's
outputs (plus, for the Matlab cases, what the Matlab toolbox produced, as
held by the reference's own ``pyls/tests/data/*.mat``).
"""

import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'

sys.modules.setdefault('h5py', types.ModuleType('h5py'))
sys.path.insert(0, REF)
warnings.simplefilter('ignore')

import pyls                                                    # noqa: E402
import pyls.base as pbase                                      # noqa: E402
from pyls.types.regression import PLSRegression               # noqa: E402


_SPLIT_LOG = []
_orig_gen_splits = pbase.gen_splits


def _logging_gen_splits(*args, **kwargs):
    out = _orig_gen_splits(*args, **kwargs)
    _SPLIT_LOG.append(np.array(out, dtype=bool))
    return out


pbase.gen_splits = _logging_gen_splits
import pyls.types.behavioral as _pbehav                        # noqa: E402
_pbehav.gen_splits = _logging_gen_splits      # crossval imports the name directly


def flat(res, prefix='ref_'):
    """PLSResults -> flat dict of arrays (inputs dropped)."""
    out = {}
    for key in ('x_weights', 'y_weights', 'x_scores', 'y_scores',
                'y_loadings', 'singvals', 'varexp'):
        if res.get(key) is not None:
            out[prefix + key] = np.asarray(res[key])
    for sub in ('permres', 'bootres', 'splitres', 'cvres'):
        for key, val in res[sub].items():
            if val is not None:
                out['{}{}__{}'.format(prefix, sub, key)] = np.asarray(val)
    return out


def run_plsc(name, fcn, X, Y=None, **kw):
    """Run behavioral/meancentered PLS through the reference, capture."""
    del _SPLIT_LOG[:]
    kw = dict(kw)
    kw.setdefault('verbose', False)
    kw['permindices'] = True
    if fcn is pyls.behavioral_pls:
        kw.setdefault('test_split', 0)
        res = fcn(X.copy(), Y.copy(), **kw)
    else:
        res = fcn(X.copy(), **kw)
    out = flat(res)
    out['X'] = X
    if Y is not None:
        out['Y'] = Y
    out['groups'] = np.asarray(res.inputs.groups)
    out['n_cond'] = np.asarray(res.inputs.n_cond)
    for k in ('covariance', 'rotate', 'mean_centering', 'ci', 'seed',
              'n_split'):
        v = kw.get(k)
        if v is not None:
            out[k] = np.asarray(v)
    if kw.get('test_split'):
        # the cross-validation masks are the LAST gen_splits call (behavioral.py:220)
        out['cv_splits'] = _SPLIT_LOG[-1]
        out['test_size'] = np.asarray(kw.get('test_size', 0.25))
    n_perm = kw.get('n_perm', 0)
    if kw.get('n_split') and n_perm:
        # order of gen_splits calls: permutation i = 0..P-1 (seed=i,
        # base.py:705-708), then the original data (seed=self.rs, :377-380)
        assert len(_SPLIT_LOG) == n_perm + 1, len(_SPLIT_LOG)
        out['perm_splitsamples'] = np.stack(_SPLIT_LOG[:n_perm])
        out['splitsamples'] = _SPLIT_LOG[n_perm]
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('wrote', name, sorted(out)[:4], '...')


def main():
    rs = np.random.RandomState(20240917)

    # ---- c1: Linnerud (docs/user_guide/behavioral.rst:143-245) ------------
    X = np.loadtxt(os.path.join(REF, 'data', 'linnerud_exercise.csv'),
                   delimiter=',', skiprows=1, usecols=(1, 2, 3))
    Y = np.loadtxt(os.path.join(REF, 'data', 'linnerud_physio.csv'),
                   delimiter=',', skiprows=1, usecols=(1, 2, 3))
    run_plsc('linnerud', pyls.behavioral_pls, X, Y, n_perm=100, n_boot=100,
             seed=1234)

    # ---- synthetic behavioral --------------------------------------------
    def synth(S, B, T, signal=0.5):
        Xs = rs.randn(S, B)
        Ys = rs.randn(S, T)
        Ys[:, :min(T, B)] += signal * Xs[:, :min(T, B)]
        return Xs, Ys

    Xs, Ys = synth(40, 120, 6)
    run_plsc('bpls_1g1c', pyls.behavioral_pls, Xs, Ys, n_perm=30, n_boot=30,
             seed=1234)
    run_plsc('bpls_1g1c_norot', pyls.behavioral_pls, Xs, Ys, n_perm=20,
             n_boot=0, rotate=False, seed=7)
    run_plsc('bpls_1g1c_cov', pyls.behavioral_pls, Xs, Ys, n_perm=20,
             n_boot=20, covariance=True, seed=11)
    Xs, Ys = synth(46, 90, 4)
    run_plsc('bpls_2g2c', pyls.behavioral_pls, Xs, Ys, groups=[11, 12],
             n_cond=2, n_perm=25, n_boot=25, seed=1234)
    run_plsc('bpls_2g2c_cov', pyls.behavioral_pls, Xs, Ys, groups=[11, 12],
             n_cond=2, n_perm=15, n_boot=15, covariance=True, seed=5)
    Xs, Ys = synth(36, 70, 5)
    run_plsc('bpls_1g3c', pyls.behavioral_pls, Xs, Ys, groups=[12], n_cond=3,
             n_perm=15, n_boot=15, seed=3)
    # wide Y: J*T > B  (sign rule switches to y_weights, compute.py:47-50)
    Xs, Ys = synth(30, 7, 5)
    run_plsc('bpls_wideY', pyls.behavioral_pls, Xs, Ys, groups=[15, 15],
             n_perm=15, n_boot=15, seed=9)
    # split-half
    Xs, Ys = synth(40, 80, 5)
    run_plsc('bpls_split', pyls.behavioral_pls, Xs, Ys, n_perm=12, n_boot=10,
             n_split=6, seed=1234)
    Xs, Ys = synth(44, 60, 3)
    run_plsc('bpls_2g2c_split', pyls.behavioral_pls, Xs, Ys, groups=[10, 12],
             n_cond=2, n_perm=8, n_boot=0, n_split=4, seed=21)

    # ---- synthetic mean-centred ------------------------------------------
    Xm = rs.randn(54, 100)
    Xm[:18] += 0.8
    Xm[27:] -= 0.5
    for mc in (0, 1, 2):
        run_plsc('mpls_3g2c_mc{}'.format(mc), pyls.meancentered_pls, Xm,
                 groups=[8, 9, 10], n_cond=2, mean_centering=mc, n_perm=20,
                 n_boot=20, seed=1234)
    run_plsc('mpls_3g1c', pyls.meancentered_pls, Xm, groups=[17, 18, 19],
             n_cond=1, mean_centering=1, n_perm=20, n_boot=20, seed=8)
    run_plsc('mpls_1g3c_norot', pyls.meancentered_pls, Xm, groups=[18],
             n_cond=3, mean_centering=0, n_perm=15, n_boot=0, rotate=False,
             seed=4)
    run_plsc('mpls_2g2c_split', pyls.meancentered_pls, Xm[:48],
             groups=[11, 13], n_cond=2, mean_centering=0, n_perm=8, n_boot=8,
             n_split=4, seed=2)

    # ---- the reference's own Matlab fixtures (pyls/tests/data/*.mat) ------
    for fn in ('bpls_onegroup_onecond_nosplit', 'bpls_onegroup_onecond_split',
               'mpls_multigroup_onecond_nosplit',
               'mpls_multigroup_onecond_split'):
        mat = pyls.matlab.import_matlab_result(
            os.path.join(REF, 'pyls', 'tests', 'data', fn + '.mat'))
        inp = dict(mat['inputs'])
        method = inp.pop('method')
        keep = {}
        # trim resampling arrays so the fixture stays small and quick
        n_perm, n_boot = 40, 40
        keep['permsamples'] = np.asarray(inp['permsamples'])[:, :n_perm]
        keep['bootsamples'] = np.asarray(inp['bootsamples'])[:, :n_boot]
        n_split = 5 if inp.get('n_split') else 0
        kw = dict(groups=[int(g) for g in np.atleast_1d(inp['groups'])],
                  n_cond=int(inp['n_cond']), n_perm=n_perm, n_boot=n_boot,
                  n_split=n_split, seed=1234,
                  permsamples=keep['permsamples'],
                  bootsamples=keep['bootsamples'])
        # what Matlab itself produced (full-size run) -- compared at the
        # reference's own tolerances (pyls/tests/matlab.py:108-199)
        extra = {'matlab_' + k: np.asarray(mat[k]) for k in
                 ('x_weights', 'y_weights', 'singvals', 'x_scores',
                  'y_scores', 'y_loadings') if mat.get(k) is not None}
        if method == 3:
            run_plsc('mat_' + fn, pyls.behavioral_pls, np.asarray(inp['X']),
                     np.asarray(inp['Y']), **kw)
        else:
            kw['mean_centering'] = int(inp.get('mean_centering', 0))
            run_plsc('mat_' + fn, pyls.meancentered_pls, np.asarray(inp['X']),
                     **kw)
        path = os.path.join(HERE, 'mat_' + fn + '.npz')
        data = dict(np.load(path))
        data.update(extra)
        np.savez_compressed(path, **data)

    # ---- seeded index generators (pyls/base.py:10-229, utils.py:200-224) --
    gen = {}
    cases = [([6], 1), ([10], 2), ([5, 7], 1), ([4, 6, 5], 2), ([20], 1),
             ([9, 9], 3)]
    for n, (groups, n_cond) in enumerate(cases):
        tag = 'case{}'.format(n)
        gen[tag + '_groups'] = np.asarray(groups)
        gen[tag + '_n_cond'] = np.asarray(n_cond)
        gen[tag + '_perm'] = pbase.gen_permsamp(groups, n_cond, 12, seed=1234,
                                                verbose=False)
        gen[tag + '_boot'] = pbase.gen_bootsamp(groups, n_cond, 12, seed=1234,
                                                verbose=False)
        gen[tag + '_split'] = _orig_gen_splits(groups, n_cond, 6, seed=1234,
                                               test_size=0.5)
        gen[tag + '_split25'] = _orig_gen_splits(groups, n_cond, 6, seed=99,
                                                 test_size=0.25)
    gen['permute_cols_kat'] = pyls.utils.permute_cols(
        np.arange(9).reshape(3, 3), seed=np.random.RandomState(1234))
    # tiny-group case that exhausts the 500 tries and warns (base.py:73-75)
    gen['dup_perm'] = pbase.gen_permsamp([3], 1, 10, seed=1234, verbose=False)
    gen['dup_boot'] = pbase.gen_bootsamp([3], 1, 12, seed=1234, verbose=False)
    np.savez_compressed(os.path.join(HERE, 'indexgen.npz'), **gen)
    print('wrote indexgen')

    # ---- SIMPLS regression (pyls/types/regression.py) ---------------------
    # pls_regression(n_perm > 0) raises TypeError at this commit (SURVEY
    # section 0.2): drive the bootstrap through the public entry and the
    # permutation through PLSRegression._single_perm directly.
    for tag, (S, B, T, k) in dict(t4=(30, 50, 4, 3), t8=(36, 80, 8, 5),
                                  t16=(60, 200, 16, 4)).items():
        Xs, Ys = synth(S, B, T, signal=0.7)
        res = pyls.pls_regression(Xs.copy(), Ys.copy(), n_components=k,
                                  n_perm=0, n_boot=12, seed=1234,
                                  verbose=False)
        out = flat(res)
        obj = PLSRegression(Xs.copy(), Ys.copy(), n_components=k, n_perm=0,
                            n_boot=0, seed=1234, verbose=False)
        Xc = Xs - Xs.mean(axis=0, keepdims=True)
        Yc = Ys - Ys.mean(axis=0, keepdims=True)
        permsamp = pbase.gen_permsamp([S], 1, 10, seed=77, verbose=False)
        out['permsamples'] = permsamp
        out['ref_perm_varexp'] = np.stack([
            obj._single_perm(Xc, Yc, inds=permsamp[:, i], original=None,
                             seed=i)[0] for i in range(permsamp.shape[1])], -1)
        out['X'], out['Y'], out['n_components'] = Xs, Ys, np.asarray(k)
        np.savez_compressed(os.path.join(HERE, 'simpls_' + tag + '.npz'),
                            **out)
        print('wrote simpls_' + tag)


def main_cv():
    """Cross-validation cases (own RandomState so the fixtures above stay
    byte-stable when cases are added)."""
    rs = np.random.RandomState(424242)

    def synth(S, B, T, signal=0.8):
        Xs = rs.randn(S, B)
        Ys = rs.randn(S, T)
        Ys[:, :T] += signal * Xs[:, :T]
        return Xs, Ys

    # default ON in behavioral_pls: test_split=100, test_size=.25 (behavioral.py:231-235)
    Xs, Ys = synth(60, 150, 4)
    run_plsc('bpls_cv', pyls.behavioral_pls, Xs, Ys, n_perm=0, n_boot=0, test_split=8,
             test_size=0.25, seed=1234)
    Xs, Ys = synth(64, 90, 3)
    run_plsc('bpls_2g2c_cv', pyls.behavioral_pls, Xs, Ys, groups=[14, 18], n_cond=2,
             n_perm=5, n_boot=5, test_split=6, test_size=0.3, seed=77)


def main_cv_cov():
    """Cross-validation in covariance mode: the decomposition uses centred data
    only (compute.py:86-87) while rescale_test still z-maps the test rows with the
    training mean / std (compute.py:148), behavioral.py:126-170."""
    rs = np.random.RandomState(515151)
    Xs = rs.randn(56, 120) * (1.0 + rs.rand(1, 120))
    Ys = rs.randn(56, 3) * np.array([1.0, 2.5, 0.4]) + 0.8 * Xs[:, :3]
    run_plsc('bpls_cv_cov', pyls.behavioral_pls, Xs, Ys, groups=[12, 16], n_cond=2, n_perm=5,
             n_boot=5, covariance=True, test_split=6, test_size=0.25, seed=31)


def main_reg_extra():
    """pls_regression with 3-D Y (aggfunc) and with all-NaN rows
    (pyls/tests/types/test_regression.py:69-90)."""
    rs = np.random.RandomState(777)
    S, B, T, C, k = 34, 60, 5, 7, 3
    X = rs.randn(S, B)
    Y3 = rs.randn(S, T, C) + 0.8 * X[:, :T, None]
    # the reference builds its own (2, n_boot) resampling array with
    # np.array(list(zip(s.T, c.T))).T (regression.py:216), which numpy >= 1.24
    # rejects (inhomogeneous shape); supply the equivalent object array.
    sb = pbase.gen_bootsamp([S], 1, 9, seed=1234, verbose=False)
    cb = pbase.gen_bootsamp([C], 1, 9, seed=1234, verbose=False)
    boots = np.empty((2, 9), dtype=object)
    for i in range(9):
        boots[0, i], boots[1, i] = sb[:, i], cb[:, i]
    for agg in ('mean', 'median'):
        res = pyls.pls_regression(X.copy(), Y3.copy(), n_components=k, n_perm=0, n_boot=9,
                                  aggfunc=agg, bootsamples=boots, seed=1234, verbose=False)
        out = flat(res)
        bs = res['bootres']['bootsamples']
        out['boot_subjects'] = np.stack([np.asarray(b) for b in bs[0]], axis=-1)
        out['boot_third'] = np.stack([np.asarray(b) for b in bs[1]], axis=-1)
        del out['ref_bootres__bootsamples']
        out['X'], out['Y'], out['n_components'] = X, Y3, np.asarray(k)
        np.savez_compressed(os.path.join(HERE, 'simpls_3d_{}.npz'.format(agg)), **out)
        print('wrote simpls_3d_' + agg)
    Xn = rs.randn(40, 70)
    Yn = rs.randn(40, 4) + 0.8 * Xn[:, :4]
    Xn[[5, 17]] = np.nan
    Yn[11] = np.nan
    res = pyls.pls_regression(Xn.copy(), Yn.copy(), n_components=3, n_perm=0, n_boot=10,
                              seed=1234, verbose=False)
    out = flat(res)
    obj = PLSRegression(Xn.copy(), Yn.copy(), n_components=3, n_perm=0, n_boot=0, seed=1234,
                        verbose=False)
    Xc = Xn - np.nanmean(Xn, axis=0, keepdims=True)
    Yc = Yn - np.nanmean(Yn, axis=0, keepdims=True)
    permsamp = pbase.gen_permsamp([40], 1, 8, seed=77, verbose=False)
    out['permsamples'] = permsamp
    out['ref_perm_varexp'] = np.stack([
        obj._single_perm(Xc, Yc, inds=permsamp[:, i], original=None, seed=i)[0]
        for i in range(permsamp.shape[1])], -1)
    out['X'], out['Y'], out['n_components'] = Xn, Yn, np.asarray(3)
    np.savez_compressed(os.path.join(HERE, 'simpls_nan.npz'), **out)
    print('wrote simpls_nan')


def main_reg_3d_nan():
    """pls_regression with 3-D Y AND all-NaN rows in X and in Y (regression.py:48-53,
    308-313): the rows are masked per bootstrap after the third axis is aggregated."""
    rs = np.random.RandomState(999)
    S, B, T, C, k = 30, 50, 4, 6, 3
    X = rs.randn(S, B)
    Y3 = rs.randn(S, T, C) + 0.8 * X[:, :T, None]
    X[[4, 19]] = np.nan
    Y3[9] = np.nan
    sb = pbase.gen_bootsamp([S], 1, 7, seed=1234, verbose=False)
    cb = pbase.gen_bootsamp([C], 1, 7, seed=1234, verbose=False)
    boots = np.empty((2, 7), dtype=object)
    for i in range(7):
        boots[0, i], boots[1, i] = sb[:, i], cb[:, i]
    res = pyls.pls_regression(X.copy(), Y3.copy(), n_components=k, n_perm=0, n_boot=7, aggfunc='mean',
                              bootsamples=boots, seed=1234, verbose=False)
    out = flat(res)
    out['boot_subjects'], out['boot_third'] = sb, cb
    del out['ref_bootres__bootsamples']
    out['X'], out['Y'], out['n_components'] = X, Y3, np.asarray(k)
    np.savez_compressed(os.path.join(HERE, 'simpls_3d_nan.npz'), **out)
    print('wrote simpls_3d_nan')


def main_reg_seed_envelope():
    """SIMPLS with T = 20 > 11 behaviours (the regime of BASELINE config c5): the
    reference's rank-1 randomized SVD (regression.py:103 -> compute.py:43-50: 1 + 10
    oversampled directions, 4 power iterations) is APPROXIMATE there and depends on its
    seed.  Records what the reference returns for three seeds on one design, so that the
    distance of the exact restatement (the oracle) can be stated against the reference's
    own spread -- the survey's ruling for "parity unpinned" (SURVEY.md section 0.3)."""
    from pyls.types.regression import simpls
    rs = np.random.RandomState(20200)
    S, B, T, k = 80, 400, 20, 8
    Xs = rs.randn(S, B)
    Ys = rs.randn(S, T) + 0.5 * Xs[:, :T]
    out = dict(X=Xs, Y=Ys, n_components=np.asarray(k), seeds=np.array([0, 1, 2]))
    for sd in (0, 1, 2):
        r = simpls(Xs.copy(), Ys.copy(), n_components=k, seed=sd)
        out['ref_x_weights_seed{}'.format(sd)] = r['x_weights']
        out['ref_pctvar_y_seed{}'.format(sd)] = np.asarray(r['pctvar'][1])
        out['ref_y_loadings_seed{}'.format(sd)] = r['y_loadings']
    np.savez_compressed(os.path.join(HERE, 'simpls_t20_seeds.npz'), **out)
    print('wrote simpls_t20_seeds')


def main_mpls_seed_envelope():
    """Rank-deficient Procrustes (mean-centred PLS): the reference feeds the ARBITRARY null-space singular vectors
    that randomized_svd returns into ``original.T @ permuted`` (compute.py:260), so the rotated bootstrap vectors
    of ``BasePLS._single_boot`` (base.py:530-574) depend on the SVD seed.  Records what the reference returns for
    analysis seeds 0 ... 5 on the first six bootstraps of two committed designs -- fixture ``mpls_3g2c_mc0`` (3 groups
    x 2 conditions, 6 cells: 5 live LVs, one null) and ``mpls_2g2c_split`` (2 groups x 2 conditions) -- so that
    the distance of the oracle / the device (which align live LVs with live LVs only) can be stated against the
    reference's OWN seed-to-seed spread, per live LV (the ``simpls_t20_seeds`` pattern)."""
    from pyls.types.meancentered import MeanCenteredPLS
    out = {}
    for tag, name in (('a', 'mpls_3g2c_mc0'), ('b', 'mpls_2g2c_split')):
        g = np.load(os.path.join(HERE, name + '.npz'), allow_pickle=True)
        X, groups, n_cond = g['X'], [int(v) for v in g['groups']], int(g['n_cond'])
        mc = int(g['mean_centering']) if 'mean_centering' in g else 0
        boots = g['ref_bootres__bootsamples'][:, :6]
        pls = MeanCenteredPLS(X=X.copy(), groups=groups, n_cond=n_cond, mean_centering=mc, n_perm=0, n_boot=0,
                              seed=1234, verbose=False)
        res = pls.results
        out[tag + '_fixture'] = np.asarray(name)
        out[tag + '_X'], out[tag + '_groups'], out[tag + '_n_cond'] = X, np.asarray(groups), np.asarray(n_cond)
        out[tag + '_mean_centering'] = np.asarray(mc)
        out[tag + '_bootsamples'] = boots
        out[tag + '_x_weights'], out[tag + '_singvals'] = res['x_weights'], res['singvals']
        # the seed of an analysis reaches BOTH decompositions: the original's (run_pls: self.svd(X, Y, seed=self.rs),
        # base.py:362-364 -- its null columns are part of ``original``) and every bootstrap's (base.py:503-506)
        out[tag + '_seeds'] = np.arange(6)
        for sd in range(6):
            U0 = pls.svd(pls.inputs.X, pls.inputs.Y, groups=pls.dummy, seed=sd)[0]
            ub = []
            for i in range(boots.shape[1]):
                _, U_boot = pls._single_boot(pls.inputs.X, pls.inputs.Y, boots[:, i], groups=pls.dummy,
                                             original=U0, seed=sd)
                ub.append(U_boot)
            out['{}_ref_uboot_seed{}'.format(tag, sd)] = np.stack(ub, -1)          # (B, L, 6)
            out['{}_ref_original_seed{}'.format(tag, sd)] = U0
    np.savez_compressed(os.path.join(HERE, 'seeds_mpls.npz'), **out)
    print('wrote seeds_mpls')


def main_matimport():
    """pyls.matlab.import_matlab_result on the reference's own .mat fixtures
    (pyls/tests/data/*.mat, mirrored as data files under tests/golden/mat/):
    every key of the returned PLSResults, flattened; arrays above 16 KB are
    recorded as shape + dtype + sha256 of their bytes."""
    import glob
    import hashlib

    def flatten(d, pre=''):
        out = {}
        for k, v in d.items():
            if hasattr(v, 'items'):
                out.update(flatten(v, pre + k + '__'))
            elif v is not None:
                out[pre + k] = np.asarray(v)
        return out
    for f in sorted(glob.glob(os.path.join(REF, 'pyls', 'tests', 'data', '*.mat'))):
        name = os.path.basename(f)[:-4]
        if name == 'empty':
            continue
        flat_res = flatten(pyls.matlab.import_matlab_result(f))
        out = {}
        for k, v in flat_res.items():
            if v.nbytes <= 16384:
                out[k] = v
            else:
                out['sha256:' + k] = np.array([str(v.shape), str(v.dtype),
                                               hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()])
        np.savez_compressed(os.path.join(HERE, 'matimport_' + name + '.npz'), **out)
        print('wrote matimport_' + name, len(out), 'keys')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'matimport':
        main_matimport()
    elif len(sys.argv) > 1 and sys.argv[1] == 'reg3dnan':
        main_reg_3d_nan()
    elif len(sys.argv) > 1 and sys.argv[1] == 'cv':
        main_cv()
    elif len(sys.argv) > 1 and sys.argv[1] == 'reg':
        main_reg_extra()
    elif len(sys.argv) > 1 and sys.argv[1] == 'cvcov':
        main_cv_cov()
    elif len(sys.argv) > 1 and sys.argv[1] == 'regseeds':
        main_reg_seed_envelope()
    elif len(sys.argv) > 1 and sys.argv[1] == 'mplsseeds':
        main_mpls_seed_envelope()
    else:
        main()
        main_cv()
        main_reg_extra()
        main_cv_cov()
        main_matimport()
        main_reg_3d_nan()
        main_reg_seed_envelope()
        main_mpls_seed_envelope()
