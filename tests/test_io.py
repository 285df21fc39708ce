"""save_results / load_results (pyls/io.py:12-122) against an in-memory stand-in for
h5py: the traversal -- layout, None handling, nested records, suffix rule.  The ON-DISK
format (HDF5 through libhdf5, interchange with the reference's own files) is
tests/test_io_disk.py."""
import sys
import types

import numpy as np
import pytest

from pypyls_amd.structures import PLSResults


class _Dataset(object):
    def __init__(self, data):
        self.data = np.array(data)

    def __getitem__(self, key):
        assert key == ()
        return self.data


class _Group(dict):
    def __init__(self):
        super().__init__()
        self.attrs = {}

    def _walk(self, path, create=False):
        node = self
        for part in [p for p in path.split('/') if p]:
            if part not in node:
                if not create:
                    raise KeyError(path)
                dict.__setitem__(node, part, _Group())
            node = dict.__getitem__(node, part)
        return node

    def create_group(self, path):
        assert path.strip('/') and path.strip('/').split('/')[-1] not in self._walk('/'.join(path.split('/')[:-1]), True)
        return self._walk(path, create=True)

    def create_dataset(self, key, shape=None, dtype=None, data=None):
        ds = _Dataset(np.zeros(shape, dtype) if data is None else data)
        dict.__setitem__(self, key, ds)
        return ds

    def __getitem__(self, path):
        return self._walk(path)


_FILES = {}


class _File(_Group):
    def __init__(self, fname, mode):
        super().__init__()
        if mode == 'r':
            self.update(_FILES[fname])
            self.attrs = _FILES[fname].attrs
        else:
            _FILES[fname] = self

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


@pytest.fixture
def fake_h5py(monkeypatch):
    mod = types.ModuleType('h5py')
    mod.File, mod.Group, mod.Dataset = _File, _Group, _Dataset
    mod.is_hdf5 = lambda fname: fname in _FILES
    monkeypatch.setitem(sys.modules, 'h5py', mod)
    _FILES.clear()
    return mod


def _results():
    rs = np.random.RandomState(0)
    return PLSResults(
        x_weights=rs.rand(7, 3), y_weights=rs.rand(3, 3), singvals=rs.rand(3), varexp=rs.rand(3),
        x_scores=rs.rand(5, 3), y_scores=rs.rand(5, 3),
        permres=dict(pvals=rs.rand(3), perm_singval=rs.rand(3, 4), permsamples=rs.randint(0, 5, (5, 4))),
        bootres=dict(x_weights_normed=rs.rand(7, 3), bootsamples=rs.randint(0, 5, (5, 4))),
        inputs=dict(X=rs.rand(5, 7), Y=rs.rand(5, 3), groups=[5], n_cond=1, n_perm=4, n_boot=4,
                    n_split=None, test_split=None, rotate=True, ci=95, seed=1234, verbose=False,
                    method=3, covariance=False))


def test_round_trip_and_layout(fake_h5py):
    from pypyls_amd import io
    res = _results()
    path = io.save_results('mem_results', res)
    assert path == 'mem_results.hdf5'                       # suffix appended (io.py:58-59)
    h5 = _FILES[path]
    top = h5['/results']                                    # group /results (io.py:61-62)
    assert isinstance(top['x_weights'], _Dataset)           # ndarray -> dataset (io.py:48-49)
    assert isinstance(top['permres'], _Group) and isinstance(top['inputs'], _Group)   # nested record -> sub-group
    assert top['inputs'].attrs['n_split'] == 'None'         # None -> 'None' attribute (io.py:51-54)
    assert top['inputs'].attrs['seed'] == 1234 and top['inputs'].attrs['rotate'] is True
    np.testing.assert_array_equal(top['permres']['perm_singval'][()], res.permres.perm_singval)
    back = io.load_results('mem_results')                   # suffix appended on load too (io.py:113-114)
    assert isinstance(back, PLSResults)
    assert back == res
    assert back.inputs.n_split is None and back.inputs.seed == 1234
    np.testing.assert_array_equal(back.bootres.bootsamples, res.bootres.bootsamples)


def test_not_hdf5_rejected(fake_h5py):
    from pypyls_amd import io
    with pytest.raises(TypeError):
        io.load_results('never_written')                    # io.py:116-118


def test_without_h5py_and_without_libhdf5_both_raise(monkeypatch):
    from pypyls_amd import io, _h5lite
    monkeypatch.setitem(sys.modules, 'h5py', None)

    def no_lib():
        raise ImportError('no HDF5 C library found (test)')
    monkeypatch.setattr(_h5lite, '_LIB', None)
    monkeypatch.setattr(_h5lite, '_find', no_lib)
    with pytest.raises(ImportError):
        io.save_results('x', _results())
    with pytest.raises(ImportError):
        io.load_results('x')
