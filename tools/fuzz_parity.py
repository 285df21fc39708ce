#!/usr/bin/env python3
"""Randomised parity sweep (GPU): random shapes / designs / options through the
engine against the oracle.  usage: python tools/fuzz_parity.py [n_cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cpu_ref as ref                      # noqa: E402  (checker only)
from pypyls_amd import resampling as rsmp              # noqa: E402
from pypyls_amd.engine import Engine                   # noqa: E402


LAST = None


def close(a, b, rtol, what):
    a, b = np.asarray(a, float), np.asarray(b, float)
    scale = np.max(np.abs(b)) if b.size else 1.0
    err = np.max(np.abs(a - b)) if b.size else 0.0
    if not err <= rtol * scale + 1e-300:
        raise AssertionError('%s: err %.3e scale %.3e' % (what, err, scale))


def one_case(rs, idx):
    method = rs.choice(['behavioral', 'behavioral', 'meancentered'])
    n_groups = int(rs.choice([1, 1, 2, 3]))
    n_cond = int(rs.choice([1, 1, 2, 3]))
    if method == 'meancentered' and n_groups * n_cond == 1:
        n_cond = 2
    per = [int(rs.randint(6, 30)) for _ in range(n_groups)]
    groups = per
    S = sum(per) * n_cond
    B = int(rs.choice([3, 17, 64, 129, 500, 1500]))
    # (occasionally wide: T' = J T beyond one cross-product block -> sliced layout, global-memory solvers)
    T = int(rs.choice([70, 85, 110, 140, 33] if os.environ.get('FUZZ_WIDE') else [1, 2, 5, 11, 24, 24, 60, 110])) if method == 'behavioral' else 0
    cov = bool(method == 'behavioral' and rs.rand() < 0.25)
    mc = int(rs.randint(0, 3))
    if method == 'meancentered':
        if n_cond == 1 and mc == 0:
            mc = 1
        if n_groups == 1 and mc == 1:
            mc = 0
    rotate = bool(rs.rand() < 0.7)
    X = rs.randn(S, B) + 2.0 * rs.rand(1, B)
    if method == 'meancentered':
        X[: per[0]] += 0.8
    Y = rs.randn(S, max(T, 1)) + 0.7
    k = min(T, B)
    if k:
        Y[:, :k] += 0.5 * X[:, :k]
    global LAST
    LAST = desc = dict(i=idx, method=method, groups=groups, n_cond=n_cond, S=S, B=B, T=T, cov=cov, mc=mc, rotate=rotate)
    if method == 'behavioral' and n_groups * n_cond * T > 1280:
        return desc, 'skipped'
    from pypyls_amd.engine import options_from_env
    eng = Engine(**options_from_env())
    # bootstrap sums: quadratic-form route forced on / off (it only applies to the unscaled modes; plsx_boot_begin)
    quad = bool(rs.rand() < 0.5)
    eng.set_option('quad_sums', 1 if quad else -1)
    desc['quad_sums'] = quad
    eng.set_data(X, Y if method == 'behavioral' else None, rsmp.cell_of_row(groups, n_cond), n_groups, n_cond,
                 0 if method == 'behavioral' else 1, mean_centering=mc, covariance=cov)
    spec = ref.Spec(method, groups, n_cond, cov, mc, rotate)
    Yo = Y if method == 'behavioral' else spec.dummy
    U, d, V = ref.decompose(spec, X, Yo)
    dv = np.diag(d)
    live = ref.live_lvs(dv)
    xw, sv, yw = eng.decompose()
    close(sv[live], dv[live], 1e-7, 'singvals')
    eng.set_original(U, dv, V)
    nres = 5
    perms = rsmp.gen_permsamp(groups, n_cond, nres, seed=int(rs.randint(1 << 30)))
    boots = rsmp.gen_bootsamp(groups, n_cond, nres, seed=int(rs.randint(1 << 30)))
    # rank-deficient AND rectangular V (B < T'): the rotated permutation statistic depends on the
    # arbitrary null-space vectors of the reference's SVD -> raw values only
    if live.sum() < len(dv) and len(dv) < V.shape[0]:
        rotate = False
        spec.rotate = False
    got = eng.perm(perms, rotate=rotate)
    want = np.stack([ref.single_perm(spec, X, Yo, perms[:, i], V)[0] for i in range(nres)], -1)
    close(got[live], want[live], 1e-6, 'perm')
    usum, usq, dist = eng.boot(boots)
    ws, wq, wd = np.zeros_like(U), np.zeros_like(U), []
    for i in range(nres):
        dd, ub = ref.single_boot(spec, X, Yo, boots[:, i], U, d)
        ws += ub
        wq += ub ** 2
        wd.append(dd)
    close(usum.cpu().numpy()[:, live], ws[:, live], 1e-5, 'u_sum')
    close(usq.cpu().numpy()[:, live], wq[:, live], 1e-5, 'u_square')
    close(dist[:, live], np.stack(wd, -1)[:, live], 1e-6, 'distrib')
    if method == 'behavioral' and not cov and min(per) >= 8:
        masks = rsmp.gen_splits(groups, n_cond, 3, seed=int(rs.randint(1 << 30)))
        uc, vc = eng.split_half(masks)
        wu, wv = ref.split_half(spec, X, Yo, U @ np.linalg.pinv(d), V @ np.linalg.pinv(d), masks)
        good = live & np.isfinite(wu) & np.isfinite(wv)
        close(uc[0].mean(-1)[good], wu[good], 1e-5, 'ucorr')
        close(vc[0].mean(-1)[good], wv[good], 1e-5, 'vcorr')
    return desc, 'ok'


def regression_case(rs, idx):
    import pypyls_amd as pls
    global LAST
    S = int(rs.randint(12, 90))
    B = int(rs.choice([5, 40, 300, 1200]))
    T = int(rs.choice([1, 3, 8, 20, 40, 70] if os.environ.get('FUZZ_REGRESSION') else [1, 3, 8, 20]))     # (the wide ones: the solver's T classes)
    # beyond rank(Y) = T the components are noise-defined; a bootstrap of S rows keeps ~0.63 S distinct ones,
    # components beyond the centred rank of a resample are arbitrary (in the reference too)
    k = int(rs.randint(1, max(1, min(S // 2 - 2, B, T, 12)) + 1))
    X = rs.randn(S, B) + rs.rand(1, B)
    Y = rs.randn(S, T)
    m = min(T, B)
    Y[:, :m] += 0.6 * X[:, :m]
    nan_rows = rs.rand() < 0.3
    if nan_rows:
        X[int(rs.randint(S))] = np.nan
        Y[int(rs.randint(S))] = np.nan
        k = min(k, S - 5)
        if not os.environ.get('FUZZ_OLD_GUARD'):
            k = max(1, min(k, (S - 2) // 2 - 2))     # (the rank guard above, on the S - 2 rows that survive)
    LAST = desc = dict(i=idx, method='regression', S=S, B=B, T=T, k=k, nan_rows=bool(nan_rows))
    from pypyls_amd.engine import Engine, options_from_env
    eng = Engine(**options_from_env())
    quad = bool(rs.rand() < 0.5)
    eng.set_option('quad_sums', 1 if quad else -1)
    desc['quad_sums'] = quad
    res = pls.pls_regression(X, Y, n_components=k, n_perm=6, n_boot=5, seed=int(rs.randint(1 << 30)), verbose=False,
                             _engine=eng)
    want = ref.run_regression(X, Y, k, permsamples=res.permres.permsamples, bootsamples=res.bootres.bootsamples)
    bs = np.asarray(res.bootres.bootsamples)
    desc['min_distinct_rows'] = int(min(len(np.unique(bs[:, i])) for i in range(bs.shape[1])))
    if os.environ.get('FUZZ_DUMP'):
        np.savez(os.path.join(ROOT, 'gpurun_out', 'fuzz_reg_%d.npz' % idx), X=X, Y=Y, k=k, perms=res.permres.permsamples,
                 boots=res.bootres.bootsamples)
    for key in ('x_weights', 'y_loadings', 'varexp'):
        close(res[key], want[key], 1e-5, key)
    close(res.permres.perm_singval, want['permres']['perm_singval'], 1e-5, 'perm varexp')
    close(res.bootres.x_weights_normed, want['bootres']['x_weights_normed'], 1e-5, 'bsr')
    close(res.bootres.x_weights_stderr, want['bootres']['x_weights_stderr'], 1e-5, 'stderr')
    close(res.bootres.y_loadings_boot, want['bootres']['y_loadings_boot'], 1e-5, 'y_loadings_boot')
    return desc, 'ok'


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rs = np.random.RandomState(seed)
    bad = 0
    for i in range(n):
        sub = np.random.RandomState(rs.randint(1 << 30))
        try:
            desc, status = regression_case(sub, i) if (i % 5 == 4 or os.environ.get('FUZZ_REGRESSION')) else one_case(sub, i)
        except Exception as e:                          # report and go on
            bad += 1
            print('FAIL', i, type(e).__name__, str(e)[:200], LAST, flush=True)
            continue
        print(status, desc, flush=True)
    print('failures:', bad, 'of', n)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
