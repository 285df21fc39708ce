#!/usr/bin/env python3
"""Parity at the full BASELINE size beyond what the test-suite affords: N permutations
(dual AND feature-pass route) + N bootstraps of c4 (X 500 x 200000, Y 500 x 50) against
the oracle, maximal relative errors as JSON.   python tools/fullsize_parity.py [N]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(b)))


def main():
    from oracle import cpu_ref as ref
    from pypyls_amd import resampling, hostmath
    from pypyls_amd.engine import Engine
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    S, B, T = 500, 200000, 50
    rs = np.random.RandomState(0)
    X = rs.randn(S, B)
    Y = rs.randn(S, T) + 0.3 * X[:, :T]
    from pypyls_amd.engine import options_from_env
    eng = Engine(**options_from_env())
    eng.set_data(X, Y, resampling.cell_of_row([S], 1), 1, 1, 0)
    spec = ref.Spec('behavioral', [S], 1)
    xw, sv, yw = eng.decompose()
    xw, yw = hostmath.sign_convention(xw, yw)
    U, d, V = ref.decompose(spec, X, Y)
    sg = np.sign(np.sum(xw * U, axis=0))
    out = dict(shape=[S, B, T], n=n, singvals=rel(sv, np.diag(d)), x_weights=rel(xw * sg, U), y_weights=rel(yw * sg, V))
    eng.set_original(U, np.diag(d), V)
    perms = resampling.gen_permsamp([S], 1, n, seed=1234)
    boots = resampling.gen_bootsamp([S], 1, n, seed=1235)
    t0 = time.time()
    want_p = np.stack([ref.single_perm(spec, X, Y, perms[:, i], V)[0] for i in range(n)], -1)
    ws, wq, wd = np.zeros_like(U), np.zeros_like(U), []
    for i in range(n):
        dd, ub = ref.single_boot(spec, X, Y, boots[:, i], U, d)
        ws += ub
        wq += ub ** 2
        wd.append(dd)
    out['oracle_seconds'] = time.time() - t0
    out['perm_dual'] = rel(eng.perm(perms), want_p)
    eng.set_perm_path(False)
    out['perm_feature_pass'] = rel(eng.perm(perms), want_p)
    eng.set_perm_path(True)
    usum, usq, dist = eng.boot(boots)
    out['boot_usum'] = rel(usum.cpu().numpy(), ws)
    out['boot_usq'] = rel(usq.cpu().numpy(), wq)
    out['boot_distrib'] = rel(dist, np.stack(wd, -1))
    bsr_g, se_g = eng.boot_rel(U @ d, usum, usq, n + 1, add_orig=True)
    bs = U @ d
    bsr_w, se_w = ref.boot_rel(bs, ws + bs, wq + bs ** 2, n + 1)
    out['bootstrap_ratio'] = rel(bsr_g, bsr_w)
    out['tolerance_north_star'] = 1e-5
    print(json.dumps(out))


if __name__ == '__main__':
    main()
