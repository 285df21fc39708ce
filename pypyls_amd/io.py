"""
HDF5 persistence of :class:`~pypyls_amd.structures.PLSResults` in the layout of
``pyls.save_results`` / ``pyls.load_results`` (pyls/io.py:12-122): group
``/results``, one sub-group per nested record, ndarrays as datasets, everything
else as group attributes (``None`` stored as the string 'None'), so files are
interchangeable with the reference's tooling.

Backend: ``h5py`` when it is installed; otherwise the HDF5 C library itself through
ctypes (:mod:`pypyls_amd._h5lite`: the same calls with h5py's on-disk conventions --
this build's image ships libhdf5 but no h5py for its interpreter).  With neither, both
functions raise ImportError.  tests/test_io.py pins the on-disk format both ways against
files written / read by the reference's own ``pyls.save_results`` / ``load_results``.
"""
import numpy as np

from .structures import PLSResults


def _h5py():
    try:
        import h5py
        return h5py
    except ImportError:
        pass
    try:
        from . import _h5lite
        _h5lite.lib()
        return _h5lite
    except ImportError as exc:
        raise ImportError('save_results / load_results need h5py or an HDF5 C library (libhdf5.so); '
                          'neither was found: {}'.format(exc)) from exc


def _with_suffix(fname):
    fname = str(fname)
    return fname if fname.endswith('.hdf5') else fname + '.hdf5'


def save_results(fname, results):
    """Write ``results`` to ``fname`` ('.hdf5' appended when missing); returns the path."""
    h5py = _h5py()
    fname = _with_suffix(fname)
    with h5py.File(fname, 'w') as h5:
        todo = [('/results', results)]
        while todo:
            path, record = todo.pop()
            grp = h5.create_group(path)
            for key, item in record.items():
                if isinstance(item, dict):
                    todo.append((path + '/' + key, item))
                elif isinstance(item, np.ndarray):
                    grp.create_dataset(key, data=item)
                else:
                    grp.attrs[key] = 'None' if item is None else item
    return fname


def load_results(fname):
    """Read a file written by :func:`save_results` (or by ``pyls.save_results``)."""
    h5py = _h5py()
    fname = _with_suffix(fname)
    if not h5py.is_hdf5(fname):
        raise TypeError('Provided file {} is not valid HDF5 format.'.format(fname))

    def read(grp):
        out = {}
        for key, item in grp.items():
            out[key] = item[()] if isinstance(item, h5py.Dataset) else read(item)
        for key, value in grp.attrs.items():
            out[key] = None if (isinstance(value, str) and value == 'None') else value
        return out

    with h5py.File(fname, 'r') as h5:
        return PLSResults(**read(h5['/results']))
