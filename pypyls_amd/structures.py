"""
Result / input containers with the reference's key layout
(pyls/structures.py:146-351, pyls/utils.py:16-125): dictionary objects with
attribute access that silently drop keys they do not know.
"""
import os

import numpy as np


class KeyedRecord(dict):
    """dict with attribute access restricted to ``allowed`` keys."""

    allowed = ()

    def __init__(self, **kwargs):
        super().__init__()
        for key, val in kwargs.items():
            self[key] = val

    def __setitem__(self, key, val):
        if key in type(self).allowed:
            super().__setitem__(key, val)

    def update(self, *args, **kwargs):
        for key, val in dict(*args, **kwargs).items():
            self[key] = val

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)

    def __setattr__(self, key, val):
        self[key] = val

    def __delattr__(self, key):
        try:
            del self[key]
        except KeyError:
            raise AttributeError(key)

    def __dir__(self):
        return list(self.keys())

    def _filled(self):
        out = set()
        for key, val in self.items():
            if val is None:
                continue
            if isinstance(val, dict) and len(val) == 0:
                continue
            if isinstance(val, KeyedRecord) and not val._filled():
                continue
            out.add(key)
        return out

    def __repr__(self):
        keys = [k for k in type(self).allowed if k in self._filled()]
        return '{}({})'.format(type(self).__name__, ', '.join(keys))

    __str__ = __repr__

    def __eq__(self, other):
        if not isinstance(other, type(self)):
            return False
        if self._filled() != other._filled():
            return False
        for key in self._filled():
            a, b = self[key], other[key]
            if isinstance(a, dict) and isinstance(b, dict):
                if a != b:
                    return False
                continue
            try:
                np.testing.assert_array_almost_equal(a, b)
            except (TypeError, AssertionError):
                return False
        return True

    def __ne__(self, other):
        return not self == other

    __hash__ = None


class PLSInputs(KeyedRecord):
    allowed = (
        'X', 'Y', 'groups', 'n_cond', 'n_perm', 'n_boot', 'n_split',
        'test_split', 'test_size', 'mean_centering', 'covariance', 'rotate',
        'ci', 'seed', 'verbose', 'n_proc', 'bootsamples', 'permsamples',
        'method', 'n_components', 'aggfunc', 'permindices',
        # build-only knobs (filtered like any other key): pre-drawn split masks, engine
        '_splitsamples', '_perm_splitsamples', '_cvsplits', '_engine',
    )

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        # pyls/structures.py:156-172
        if self.get('n_split') == 0:
            self['n_split'] = None
        if self.get('test_split') == 0:
            self['test_split'] = None
        n_proc = self.get('n_proc')
        if n_proc is not None:
            ncpu = os.cpu_count() or 1
            if n_proc == 'max' or n_proc == -1:
                self['n_proc'] = ncpu
            elif n_proc < 0:
                self['n_proc'] = ncpu + 1 + n_proc
        ts = self.get('test_size')
        if ts is not None and (ts < 0 or ts >= 1):
            raise ValueError('test_size must be in [0, 1). Provided value: {}'.format(ts))


class PLSBootResults(KeyedRecord):
    allowed = ('x_weights_normed', 'x_weights_stderr', 'bootsamples',
               'y_loadings', 'y_loadings_boot', 'y_loadings_ci',
               'contrast', 'contrast_boot', 'contrast_ci')


class PLSPermResults(KeyedRecord):
    allowed = ('pvals', 'permsamples', 'perm_singval')


class PLSSplitHalfResults(KeyedRecord):
    allowed = ('ucorr', 'vcorr', 'ucorr_pvals', 'vcorr_pvals',
               'ucorr_uplim', 'vcorr_uplim', 'ucorr_lolim', 'vcorr_lolim')


class PLSCrossValidationResults(KeyedRecord):
    allowed = ('pearson_r', 'r_squared')


class PLSResults(KeyedRecord):
    """Top-level result object; layout of pyls/structures.py:198-246."""
    allowed = ('x_weights', 'y_weights', 'x_scores', 'y_scores', 'y_loadings',
               'singvals', 'varexp', 'permres', 'bootres', 'splitres', 'cvres',
               'inputs')

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self['inputs'] = PLSInputs(**kwargs.get('inputs', kwargs))
        self['bootres'] = PLSBootResults(**kwargs.get('bootres', kwargs))
        self['permres'] = PLSPermResults(**kwargs.get('permres', kwargs))
        self['splitres'] = PLSSplitHalfResults(**kwargs.get('splitres', kwargs))
        self['cvres'] = PLSCrossValidationResults(**kwargs.get('cvres', kwargs))
