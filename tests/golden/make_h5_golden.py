#!/opt/conda/bin/python3.9
"""
HDF5 interchange fixtures: files written by THE REFERENCE's ``pyls.save_results`` (real h5py).

Run in the build container only, with the interpreter that has h5py (this image: /opt/conda/bin/python3.9, h5py 3.3.0
on HDF5 1.10.6; the main interpreter has none):

    /opt/conda/bin/python3.9 tests/golden/make_h5_golden.py

For each small analysis the reference runs (behavioral with split-half, mean-centred, regression) it stores
  tests/golden/h5/ref_<name>.hdf5   the file pyls.save_results wrote                       (data, ~50 KB each)
  tests/golden/h5/ref_<name>.npz    every leaf of the same PLSResults, flattened to 'a/b/c' keys, read back through
                                    h5py itself (what pyls.load_results returns), so the test can compare key by key
Data only: inputs and outputs, no reference source text.
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, '/root/reference')
warnings.simplefilter('ignore')
import pyls                                                                    # noqa: E402


def flatten(rec, prefix=''):
    out = {}
    for key, val in rec.items():
        if isinstance(val, dict):
            out.update(flatten(val, prefix + key + '/'))
        elif val is None:
            out[prefix + key] = np.array('None')
        else:
            out[prefix + key] = np.asarray(val)
    return out


def main():
    out = os.path.join(HERE, 'h5')
    os.makedirs(out, exist_ok=True)
    rs = np.random.RandomState(7)
    X = rs.randn(24, 40)
    Y = rs.randn(24, 3) + 0.5 * X[:, :3]
    cases = {
        'bpls': lambda: pyls.behavioral_pls(X, Y, groups=[12, 12], n_perm=6, n_boot=5, n_split=3, test_split=0,
                                            permindices=True, seed=11, verbose=False, n_proc=1),
        'mpls': lambda: pyls.meancentered_pls(X, groups=[12, 12], n_cond=1, n_perm=6, n_boot=5, n_split=0,
                                              test_split=0, permindices=True, seed=12, verbose=False, n_proc=1),
        'simpls': lambda: pyls.pls_regression(X, Y, n_components=2, n_perm=0, n_boot=4, seed=13,   # n_perm > 0 raises in the reference
                                              verbose=False, n_proc=1),
    }
    for name, run in cases.items():
        res = run()
        fname = pyls.save_results(os.path.join(out, 'ref_' + name), res)
        back = pyls.load_results(fname)
        same = back == res          # (the reference's own round trip of the regression record is not '==': NaN leaves)
        np.savez_compressed(os.path.join(out, 'ref_' + name + '.npz'), **flatten(back))
        print(name, os.path.getsize(fname), 'bytes,', len(flatten(back)), 'leaves, reference round trip ==:', same)


if __name__ == '__main__':
    main()
