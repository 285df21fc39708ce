import os
import sys
import glob

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4]
                  for p in glob.glob(os.path.join(GOLDEN, prefix + '*.npz')))


def live_lvs(singvals, rtol=1e-8):
    """Mask of latent variables with non-negligible singular value -- the
    reference's own comparator masks isclose(singvals, 0) LVs
    (pyls/tests/matlab.py:160)."""
    s = np.asarray(singvals)
    return s > rtol * s.max()


def assert_close(a, b, rtol=1e-5, atol=0.0, what=''):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    assert a.shape == b.shape, '{}: shape {} vs {}'.format(what, a.shape, b.shape)
    scale = np.max(np.abs(b)) if b.size else 1.0
    err = np.max(np.abs(a - b)) if b.size else 0.0
    assert err <= rtol * scale + atol, \
        '{}: max abs err {:.3e} (scale {:.3e}, rtol {:g})'.format(what, err, scale, rtol)


def assert_close_per_lv(a, b, axis, rtol=1e-5, what='', keep=None):
    """Every slice along ``axis`` (one latent variable each) within rtol of ITS OWN scale -- assert_close
    is norm-wise and would not see a 1e-3 relative error in an LV 100 x smaller than the first."""
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    assert a.shape == b.shape, '{}: shape {} vs {}'.format(what, a.shape, b.shape)
    am, bm = np.moveaxis(a, axis, 0), np.moveaxis(b, axis, 0)
    for k in range(bm.shape[0]):
        if keep is not None and not keep[k]:
            continue
        scale = np.max(np.abs(bm[k])) if bm[k].size else 1.0
        err = np.max(np.abs(am[k] - bm[k])) if bm[k].size else 0.0
        assert err <= rtol * scale, '{}: LV {}: max abs err {:.3e} (scale {:.3e}, rtol {:g})'.format(
            what, k, err, scale, rtol)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
