"""On-disk HDF5 interchange of save_results / load_results (pyls/io.py:12-122).

The main interpreter of this image has no h5py, but the image ships libhdf5: pypyls_amd/_h5lite.py speaks to it through
ctypes with h5py's on-disk conventions.  Pinned BOTH ways against the reference:
  * tests/golden/h5/ref_*.hdf5 were written by the reference's own pyls.save_results under real h5py
    (tests/golden/make_h5_golden.py); load_results here must return, leaf by leaf, what pyls.load_results returned
    (ref_*.npz);
  * files written here are read back by real h5py -- and, where /root/reference exists, by pyls.load_results itself --
    in the image's second interpreter (/opt/conda/bin/python3.9), when it is there.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H5DIR = os.path.join(ROOT, 'tests', 'golden', 'h5')
PY_H5 = '/opt/conda/bin/python3.9'
CASES = ['bpls', 'mpls', 'simpls']


@pytest.fixture
def h5lite(monkeypatch):
    """Force the ctypes backend (and skip where the image has no libhdf5)."""
    from pypyls_amd import _h5lite
    try:
        _h5lite.lib()
    except ImportError as exc:
        pytest.skip('no HDF5 C library: {}'.format(exc))
    monkeypatch.setitem(sys.modules, 'h5py', None)
    return _h5lite


def _flatten(rec, prefix=''):
    out = {}
    for key, val in rec.items():
        if isinstance(val, dict):
            out.update(_flatten(val, prefix + key + '/'))
        elif val is None:
            out[prefix + key] = np.array('None')
        else:
            out[prefix + key] = np.asarray(val)
    return out


def _assert_same_leaves(got, want, what):
    assert sorted(got) == sorted(want), (what, sorted(set(got) ^ set(want)))
    for key in want:
        a, b = got[key], want[key]
        assert a.shape == b.shape, (what, key, a.shape, b.shape)
        if b.dtype.kind in 'US':
            assert str(a) == str(b), (what, key)
        else:
            assert a.dtype == b.dtype, (what, key, a.dtype, b.dtype)
            np.testing.assert_array_equal(a, b, err_msg='{} {}'.format(what, key))     # NaN == NaN here


@pytest.mark.parametrize('case', CASES)
def test_reads_the_reference_files(h5lite, case):
    from pypyls_amd import io
    from pypyls_amd.structures import PLSResults
    res = io.load_results(os.path.join(H5DIR, 'ref_' + case))          # suffix rule (io.py:113-114)
    assert isinstance(res, PLSResults)
    want = dict(np.load(os.path.join(H5DIR, 'ref_' + case + '.npz')))
    _assert_same_leaves(_flatten(res), want, case)
    assert res.inputs.n_split is None or int(res.inputs.n_split) == 3   # 'None' attribute -> None (io.py:104-106)


@pytest.mark.parametrize('case', CASES)
def test_written_files_round_trip_and_are_read_by_h5py(h5lite, case, tmp_path):
    from pypyls_amd import io
    res = io.load_results(os.path.join(H5DIR, 'ref_' + case + '.hdf5'))
    mine = io.save_results(str(tmp_path / ('mine_' + case)), res)
    assert mine.endswith('.hdf5') and h5lite.is_hdf5(mine)
    back = io.load_results(mine)
    _assert_same_leaves(_flatten(back), _flatten(res), case + ' (own round trip)')
    if not os.path.exists(PY_H5):
        pytest.skip('no interpreter with h5py in this image')
    # real h5py reads it: every leaf as pyls.load_results would return it ...
    reader = r'''
import sys, numpy as np, h5py
def read(g):
    out = {}
    for k, v in g.items():
        out[k] = v[()] if isinstance(v, h5py.Dataset) else read(v)
    for k, v in g.attrs.items():
        out[k] = None if (isinstance(v, str) and v == 'None') else v
    return out
def flatten(rec, prefix=''):
    out = {}
    for key, val in rec.items():
        if isinstance(val, dict): out.update(flatten(val, prefix + key + '/'))
        elif val is None: out[prefix + key] = np.array('None')
        else: out[prefix + key] = np.asarray(val)
    return out
assert h5py.is_hdf5(sys.argv[1])
with h5py.File(sys.argv[1], 'r') as f:
    rec = read(f['/results'])
np.savez(sys.argv[2], **flatten(rec))
same = 'n/a'
try:
    sys.path.insert(0, '/root/reference')
    import pyls                                          # ... and, in the build container, the reference itself
    same = 'same' if pyls.load_results(sys.argv[1]) == pyls.load_results(sys.argv[3]) else 'DIFFERENT'
except ImportError:
    pass
print('reference:', same)
'''
    out = str(tmp_path / 'h5py_view.npz')
    proc = subprocess.run([PY_H5, '-W', 'ignore', '-c', reader, mine, out,
                           os.path.join(H5DIR, 'ref_' + case + '.hdf5')], capture_output=True, text=True, timeout=300)
    if proc.returncode != 0 and 'No module named' in proc.stderr:
        pytest.skip('that interpreter has no h5py: ' + proc.stderr[-200:])
    assert proc.returncode == 0, proc.stderr[-2000:]
    want = dict(np.load(os.path.join(H5DIR, 'ref_' + case + '.npz')))
    _assert_same_leaves(dict(np.load(out)), want, case + ' (h5py view of our file)')
    if case != 'simpls':                                 # (the reference's own round trip of that record is not '==': NaN leaves)
        assert 'DIFFERENT' not in proc.stdout, proc.stdout


def test_layout_on_disk(h5lite, tmp_path):
    """Group /results, nested records as sub-groups, ndarrays as datasets, the rest as attributes with h5py's types
    (io.py:40-63): str -> variable-length UTF-8, bool -> enum over int8, int -> int64, None -> 'None'."""
    from pypyls_amd import io
    from pypyls_amd.structures import PLSResults
    rs = np.random.RandomState(0)
    res = PLSResults(x_weights=rs.rand(7, 3), singvals=rs.rand(3),
                     permres=dict(pvals=rs.rand(3), permsamples=rs.randint(0, 5, (5, 4))),
                     inputs=dict(X=rs.rand(5, 7), groups=[5], n_cond=1, n_split=None, rotate=True, ci=95, seed=1234))
    path = io.save_results(tmp_path / 'lay', res)         # a Path is accepted (io.py:55-56)
    with h5lite.File(path, 'r') as f:
        top = f['/results']
        assert isinstance(top['x_weights'], h5lite.Dataset) and top['x_weights'].shape == (7, 3)
        assert isinstance(top['permres'], h5lite.Group) and 'permsamples' in top['permres']
        attrs = dict(top['inputs'].attrs.items())
        assert attrs['n_split'] == 'None' and attrs['seed'] == 1234 and attrs['seed'].dtype == np.int64
        assert attrs['rotate'] is np.True_ or attrs['rotate'] == True          # noqa: E712
        assert attrs['rotate'].dtype == np.bool_
        np.testing.assert_array_equal(attrs['groups'], [5])
    back = io.load_results(path)
    assert back == res and back.inputs.n_split is None
    with h5lite.File(str(tmp_path / 'b.hdf5'), 'w') as f:            # boolean / integer / empty / 0-d datasets
        g = f.create_group('/g')
        g.create_dataset('m', data=np.array([True, False, True]))
        g.create_dataset('i4', (2, 2), np.int32)[...] = np.array([[1, 2], [3, 4]], dtype=np.int32)
        g.create_dataset('e', data=np.zeros((0, 3)))
        g.create_dataset('z', data=np.float64(2.5))
        with pytest.raises(ValueError):
            f.create_group('/g')                          # exists
    with h5lite.File(str(tmp_path / 'b.hdf5'), 'r') as f:
        g = f['/g']
        assert g['m'][()].dtype == np.bool_ and g['m'][()].tolist() == [True, False, True]
        assert g['i4'][()].dtype == np.int32 and g['i4'][()].tolist() == [[1, 2], [3, 4]]
        assert g['e'][()].shape == (0, 3) and g['z'][()] == 2.5 and g['z'].shape == ()
        assert sorted(g.keys()) == ['e', 'i4', 'm', 'z']
    h5dump = '/opt/conda/bin/h5dump'
    if os.path.exists(h5dump):                            # the HDF5 project's own tool agrees about the types
        txt = subprocess.run([h5dump, '-H', path], capture_output=True, text=True, timeout=120).stdout
        assert 'GROUP "results"' in txt and 'H5T_IEEE_F64LE' in txt and 'H5T_STD_I64LE' in txt
        assert 'H5T_CSET_UTF8' in txt and 'STRSIZE H5T_VARIABLE' in txt and '"TRUE"             1' in txt


def test_not_hdf5_rejected_on_disk(h5lite, tmp_path):
    from pypyls_amd import io
    bad = tmp_path / 'junk.hdf5'
    bad.write_bytes(b'not an hdf5 file')
    with pytest.raises(TypeError):
        io.load_results(str(bad))                         # io.py:116-118
    with pytest.raises(TypeError):
        io.load_results(str(tmp_path))                    # a directory: the reference's own test (tests/test_io.py:17)
