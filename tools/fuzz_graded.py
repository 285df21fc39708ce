#!/usr/bin/env python3
"""Randomised GRADED-spectrum parity sweep (GPU): behaviours made nearly collinear / cell effects of graded size with
d_1/d_L drawn log-uniformly from 1e2 to 8e5, random designs (1-3 groups x 1-3 conditions, T = 3..12, both PLS-C
methods, every mean-centring), every comparison PER LATENT VARIABLE at 1e-5 against the oracle -- the checks of
tests/test_gpu_graded.py (decomposition, permutations on both routes rotated and not, bootstraps, bootstrap
ratios), which assert that nothing graded was left unrefined.  FUZZ_WIDE=1: T' = 66 .. 300 instead (the QL solver's
refinement).  usage: python tools/fuzz_graded.py [n_cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_graded as tg                          # noqa: E402  (checker: imports the oracle)
from oracle import cpu_ref as ref                     # noqa: E402
from pypyls_amd import resampling as rsmp             # noqa: E402


def one_wide(rs, idx):
    """T' = 65 .. 300 (one group, one condition: T' = T): the Householder + QL solver and its refinement (round 5)."""
    T = int(rs.choice([66, 80, 100, 130, 190, 200, 260, 300]))
    S = T + int(rs.randint(20, 120))
    B = int(rs.choice([900, 2000, 3500]))
    ratio = float(10 ** rs.uniform(2.0, 5.3))
    X = rs.randn(S, B)
    Y = tg.graded_behaviours(rs, S, T, ratio, 'mix')
    desc = dict(i=idx, method='behavioral', groups=[S], n_cond=1, S=S, B=B, T=T, kind='mix', target=ratio)
    got, worst = tg.run_case(X, Y, [S], 1, 'behavioral', n=3, null_lvs=None)
    desc.update(ratio=float(got), worst=float(worst))
    return desc, 'ok'


def one(rs, idx):
    if os.environ.get('FUZZ_WIDE'):
        return one_wide(rs, idx)
    method = rs.choice(['behavioral', 'behavioral', 'meancentered'])
    n_groups, n_cond = int(rs.choice([1, 1, 2, 3])), int(rs.choice([1, 2, 3]))
    ratio = float(10 ** rs.uniform(2.0, 5.9))
    if method == 'behavioral':
        per = [int(rs.randint(20, 60)) for _ in range(n_groups)]
        S = sum(per) * n_cond
        T = int(rs.randint(3, 13))
        if n_groups * n_cond * T > 64:
            T = max(2, 64 // (n_groups * n_cond))
        B = int(rs.choice([800, 2500, 6000]))
        kind = str(rs.choice(['mix', 'dup']))
        X = rs.randn(S, B)
        Y = tg.graded_behaviours(rs, S, T, ratio, kind)
        desc = dict(i=idx, method=method, groups=per, n_cond=n_cond, S=S, B=B, T=T, kind=kind, target=ratio)
        # the realised ratio decides what is live: every LV of these designs is (the assertion inside run_case)
        spec = ref.Spec('behavioral', per, n_cond, False, 0)
        d = np.diag(ref.decompose(spec, X, Y)[1])
        if d.min() <= 1.2e-6 * d.max():                # too close to the rank threshold: resamples would straddle it
            return desc, 'skipped'
        got, worst = tg.run_case(X, Y, per, n_cond, 'behavioral', null_lvs=0)
    else:
        if n_groups * n_cond == 1:
            n_cond = 2
        mc = int(rs.randint(0, 3))
        if n_cond == 1 and mc == 0:
            mc = 1
        if n_groups == 1 and mc == 1:
            mc = 0
        per = [int(rs.randint(15, 40))] * n_groups
        S, B = sum(per) * n_cond, int(rs.choice([1500, 4000]))
        cells = rsmp.cell_of_row(per, n_cond)
        J = n_groups * n_cond
        spec = ref.Spec('meancentered', per, n_cond, False, mc)
        M = ref.gen_covcorr(spec, np.eye(J)[cells], spec.dummy.astype(float), spec.dummy)
        Um, sm, _ = np.linalg.svd(M)
        r = int((sm > 1e-8).sum())
        if r < 2:
            return dict(i=idx, method=method), 'skipped'
        P = rs.randn(r, B) / np.sqrt(B)
        eff = np.linalg.pinv(M) @ ((Um[:, :r] * np.logspace(0, -np.log10(ratio), r)) @ P)
        X = eff[cells] + (1.5 / ratio / np.sqrt(B)) * rs.randn(S, B)
        desc = dict(i=idx, method=method, groups=per, n_cond=n_cond, S=S, B=B, mc=mc, target=ratio)
        d = np.diag(ref.decompose(spec, X, spec.dummy.astype(float))[1])
        live = ref.live_lvs(d)
        if d[live].min() <= 1.2e-6 * d.max():
            return desc, 'skipped'
        got, worst = tg.run_case(X, None, per, n_cond, 'meancentered', mean_centering=mc, null_lvs=int((~live).sum()))
    desc.update(ratio=float(got), worst=float(worst))
    return desc, 'ok'


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    fails, worst, done = 0, 0.0, 0
    for i in range(n):
        try:
            desc, status = one(rs, i)
        except AssertionError as exc:
            fails += 1
            print('FAIL', i, str(exc)[:300])
            continue
        if status == 'ok':
            done += 1
            worst = max(worst, desc['worst'])
        print(status, {k: (round(v, 3) if isinstance(v, float) and k != 'worst' else v) for k, v in desc.items()})
    print('failures: {} of {} run ({} skipped); worst per-LV rel err {:.2e}'.format(fails, done + fails, n - done - fails, worst))
    return 1 if fails else 0


if __name__ == '__main__':
    sys.exit(main())
