"""Runs the shapes on which wave_jacobi_cols<16> went wrong (T = 44 / 56 > S: rank-deficient H, Jacobi eigen-solve) plus
full-rank controls with the Jacobi solve forced, under each library variant of tools/jacobi16_probe.sh; prints the error
of every SIMPLS component against the oracle."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(40, 120, 44, 3, 0), (50, 200, 56, 3, 0), (90, 400, 40, 5, 1), (120, 300, 64, 4, 1), (24, 150, 30, 2, 0)]

WORKER = r'''
import sys, json
import numpy as np
sys.path.insert(0, {root!r})
from pypyls_amd import _build
if {lib!r}:
    _build.LIB = {lib!r}
    _build.is_stale = lambda: False
from pypyls_amd.engine import Engine
from oracle import cpu_ref as ref
out = []
for S, B, T, k, force in {cases!r}:
    rs = np.random.RandomState(S + T)
    X = rs.randn(S, B) + rs.rand(1, B)
    Y = rs.randn(S, T)
    Y[:, :8] += 0.5 * X[:, :8]
    Xc, Yc = X - X.mean(0), Y - Y.mean(0)
    eng = Engine(options={{'simpls_jacobi': 1}} if force else None)
    eng.set_data_regression(Xc, Yc, k)
    W, pct, cvec, yl = eng.simpls_decompose()
    eng.close()
    fit = ref.simpls(Xc, Yc, k)
    Wr = fit['x_weights']
    sg = np.sign(np.sum(W * Wr, axis=0))
    err = (np.abs(W * sg - Wr).max(axis=0) / np.abs(Wr).max(axis=0)).tolist()
    perr = (np.abs(pct - fit['pctvar'][1]) / np.abs(fit['pctvar'][1])).tolist()
    rec = dict(S=S, T=T, k=k, forced=force, w_err=err, pct_err=perr)
    if not force:
        # the whole front-end call of tests/test_gpu_regression.py::test_wide_y_takes_the_solver_instantiations_of_its_class:
        # 12 permutations + 10 bootstraps (several waves per block) against the oracle
        import pypyls_amd as pls
        res = pls.pls_regression(X, Y, n_components=k, n_perm=12, n_boot=10, seed=21, verbose=False)
        want = ref.run_regression(X, Y, k, permsamples=res.permres.permsamples, bootsamples=res.bootres.bootsamples)
        def rel(a, b):
            return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(np.asarray(b))))
        rec['frontend'] = dict(x_weights=rel(res['x_weights'], want['x_weights']),
                               perm=rel(res['permres']['perm_singval'], want['permres']['perm_singval']),
                               bsr=rel(res['bootres']['x_weights_normed'], want['bootres']['x_weights_normed']),
                               ylb=rel(res['bootres']['y_loadings_boot'], want['bootres']['y_loadings_boot']))
        pls.release_default_engine()
    out.append(rec)
print('RESULT ' + json.dumps(out))
'''


def main():
    d = os.path.join(ROOT, 'tools', 'bin', 'jac16')
    variants = [('base', '')] + [(f[8:-3], os.path.join(d, f)) for f in sorted(os.listdir(d)) if f.endswith('.so')]
    for name, lib in variants:
        p = subprocess.run([sys.executable, '-c', WORKER.format(root=ROOT, lib=lib, cases=CASES)], capture_output=True,
                           text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith('RESULT ')]
        if not line:
            print(name, 'FAILED', p.stderr[-500:])
            continue
        for r in json.loads(line[0][7:]):
            print('{:5s} S={:3d} T={:2d} forced={} w_err={} pct_err={} frontend={}'.format(
                name, r['S'], r['T'], r['forced'], ['%.1e' % e for e in r['w_err']], ['%.1e' % e for e in r['pct_err']],
                {k: '%.1e' % v for k, v in r.get('frontend', {}).items()}))


if __name__ == '__main__':
    main()
