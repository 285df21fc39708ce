#!/usr/bin/env python3
"""
ONE literal BASELINE configs[3] analysis through the public call:

    behavioral_pls(X, Y, n_perm=10000, n_boot=10000, n_split=100, test_split=0, seed=1234)

at X (500 x 200000), Y (500 x 50), fp64 -- wall time in total and per phase -- and a sampled check of what
it returned against the oracle (oracle/cpu_ref.py) on the SAME index arrays and split masks:

  * 20 sampled permutations: their row of perm_singval vs ref.single_perm; the first 3 of their 100 split
    masks (RandomState(i), pyls/base.py:705-708) through the device's split-half vs ref.split_half per split;
    their entry of the split-half null (mean over 100 splits) vs the device's own 100 per-split values;
  * 8 sampled bootstraps: their y_loadings_boot slice vs ref.single_boot's distrib;
  * singular values, p-value COUNTS recomputed from the returned null, split-half of the original data.

    python tools/literal_c4.py [--perms 10000 --boots 10000 --splits 100] > profiles/r04_literal_c4.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--S', type=int, default=500)
    ap.add_argument('--B', type=int, default=200000)
    ap.add_argument('--T', type=int, default=50)
    ap.add_argument('--perms', type=int, default=10000)
    ap.add_argument('--boots', type=int, default=10000)
    ap.add_argument('--splits', type=int, default=100)
    ap.add_argument('--sample-perms', type=int, default=20)
    ap.add_argument('--sample-boots', type=int, default=8)
    ap.add_argument('--sample-splits', type=int, default=3)
    args = ap.parse_args()
    import torch
    import pypyls_amd as pls
    from pypyls_amd import plsc, resampling
    from oracle import cpu_ref as ref
    S, B, T = args.S, args.B, args.T
    rs = np.random.RandomState(0)
    X = rs.randn(S, B)
    Y = rs.randn(S, T) + 0.3 * X[:, :T]
    pls.behavioral_pls(X, Y, n_perm=32, n_boot=32, n_split=2, test_split=0, seed=1, verbose=False)   # warm-up
    phases = {}
    run = plsc._PLSCRun('behavioral', X, Y, groups=None, n_cond=1, n_perm=args.perms, n_boot=args.boots,
                        n_split=args.splits, test_size=0.25, test_split=0, covariance=False, rotate=True, ci=95,
                        permsamples=None, bootsamples=None, seed=1234, verbose=False, n_proc=None, _phases=phases)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run.run()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    out = {'call': 'behavioral_pls(X, Y, n_perm={}, n_boot={}, n_split={}, test_split=0, seed=1234), X ({} x {}), '
                   'Y ({} x {}), fp64, one MI355X'.format(args.perms, args.boots, args.splits, S, B, S, T),
           'wall_s': wall, 'phases_ms': {k: round(v, 1) for k, v in phases.items()},
           'splits_total': args.perms * args.splits,
           'splits_per_s_through_the_front_end': args.perms * args.splits / (phases.get('split_half', 0.0) * 1e-3)
           if phases.get('split_half') else None,
           'resamples_per_s_perm_boot_phases': (args.perms + args.boots) /
           ((phases.get('permutations', 0) + phases.get('bootstraps', 0)) * 1e-3),
           'note': 'phases are wall times of ONE profiled call (a device sync at every phase boundary)'}
    # ---- sampled parity against the oracle on the same arrays -------------------------------------------
    spec = ref.Spec('behavioral', [S], 1)
    t0 = time.perf_counter()
    U, d, V = ref.decompose(spec, X, Y)
    dv = np.diag(d)
    chk = {'singvals_rel': rel(res.singvals, dv)}
    sgn = np.sign(np.sum(res.x_weights * U, axis=0))
    chk['x_weights_rel'] = rel(res.x_weights * sgn, U)
    perms = res.permres.permsamples
    boots = res.bootres.bootsamples
    prs = np.random.RandomState(99)
    pick_p = np.sort(prs.choice(args.perms, size=min(args.sample_perms, args.perms), replace=False))
    pick_b = np.sort(prs.choice(args.boots, size=min(args.sample_boots, args.boots), replace=False))
    ucn, vcn = run.split_null                             # (L, n_perm) means over the splits
    eng = run.engine_used
    worst = {'perm_singval_rel': 0.0, 'split_ucorr_abs': 0.0, 'split_vcorr_abs': 0.0,
             'null_mean_vs_own_splits_abs': 0.0, 'distrib_rel': 0.0}
    di = np.linalg.inv(d)
    for i in pick_p:
        want = ref.single_perm(spec, X, Y, perms[:, i], V)[0]
        worst['perm_singval_rel'] = max(worst['perm_singval_rel'], rel(res.permres.perm_singval[:, i], want))
        masks = resampling.gen_splits([S], 1, args.splits, seed=int(i))        # (S, n_split): what permutation i drew
        uc, vc = eng.split_half(masks[None], perms=perms[:, [i]])              # (1, L, n_split)
        worst['null_mean_vs_own_splits_abs'] = max(
            worst['null_mean_vs_own_splits_abs'],
            float(np.max(np.abs(uc[0].mean(axis=-1) - ucn[:, i]))), float(np.max(np.abs(vc[0].mean(axis=-1) - vcn[:, i]))))
        ns = args.sample_splits
        Yp = Y[perms[:, i]]
        Up, dp, Vp = ref.decompose(spec, X, Yp)
        dpi = np.linalg.inv(dp)
        for j in range(min(ns, args.splits)):
            wu, wv = ref.split_half(spec, X, Yp, Up @ dpi, Vp @ dpi, masks[:, [j]])     # one split: its mean is itself
            worst['split_ucorr_abs'] = max(worst['split_ucorr_abs'], float(np.max(np.abs(uc[0][:, j] - wu))))
            worst['split_vcorr_abs'] = max(worst['split_vcorr_abs'], float(np.max(np.abs(vc[0][:, j] - wv))))
    for i in pick_b:
        dist, _ = ref.single_boot(spec, X, Y, boots[:, i], U, d)
        worst['distrib_rel'] = max(worst['distrib_rel'], rel(res.bootres.y_loadings_boot[:, :, i], dist))
    chk.update(worst)
    # p-values: counts from the returned null (strict >, compute.py:178), integer identical by construction of
    # hostmath.perm_sig; recorded with the counts themselves
    counts = np.sum(res.permres.perm_singval > res.singvals[:, None], axis=1)
    chk['pval_counts_first5'] = [int(c) for c in counts[:5]]
    chk['pvals_equal_counts_formula'] = bool(np.array_equal(res.permres.pvals, (counts + 1) / (args.perms + 1)))
    chk['splitres'] = {k: [float(v) for v in np.asarray(res.splitres[k])[:3]] for k in
                       ('ucorr', 'vcorr', 'ucorr_pvals', 'vcorr_pvals')}
    chk['oracle_seconds'] = time.perf_counter() - t0
    chk['sampled'] = {'permutations': [int(i) for i in pick_p], 'bootstraps': [int(i) for i in pick_b],
                      'splits_per_sampled_permutation_vs_oracle': args.sample_splits}
    out['parity_vs_oracle'] = chk
    ok = (chk['singvals_rel'] < 1e-9 and chk['perm_singval_rel'] < 1e-9 and chk['split_ucorr_abs'] < 1e-6 and
          chk['split_vcorr_abs'] < 1e-6 and chk['null_mean_vs_own_splits_abs'] < 1e-9 and chk['distrib_rel'] < 1e-9)
    out['parity_ok'] = bool(ok)
    print(json.dumps(out))
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
