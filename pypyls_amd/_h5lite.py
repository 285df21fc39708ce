"""
The slice of the h5py interface that :mod:`pypyls_amd.io` uses, on the HDF5 C library through ctypes.

``pyls.save_results`` / ``pyls.load_results`` (pyls/io.py:12-122) are written against h5py.  Where h5py is not
installed but ``libhdf5.so`` is (this build's image: /opt/conda/lib), this module provides the same calls --
``File``, ``Group.create_group / create_dataset / items / attrs``, ``Dataset[()]``, ``is_hdf5`` -- with h5py's
on-disk conventions, so that files written here are read by the reference (and its files here):

* ndarray -> dataset of the native type of its dtype (little-endian IEEE / two's complement), simple dataspace;
  0-d array -> scalar dataspace;
* ``bool`` -> HDF5 enum {FALSE = 0, TRUE = 1} over int8 (h5py's mapping of numpy.bool_);
* ``str`` attribute -> variable-length UTF-8 string, scalar dataspace; Python int / float -> int64 / float64 scalar;
  lists and arrays -> 1-D (n-D) simple dataspace.

Nothing here is on the resampling path; host persistence only.
"""
import ctypes
import ctypes.util
import os

import numpy as np

hid_t = ctypes.c_int64
hsize_t = ctypes.c_uint64
_SEARCH = ('/opt/conda/lib/libhdf5.so', '/opt/conda/lib/libhdf5.so.103', '/usr/lib/x86_64-linux-gnu/libhdf5_serial.so',
           '/usr/lib/x86_64-linux-gnu/libhdf5.so', '/usr/local/lib/libhdf5.so')
_LIB = None

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT, H5S_ALL, H5S_SCALAR = 0, 0, 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_ENUM = 0, 1, 3, 8
H5T_CSET_UTF8 = 1
H5T_VARIABLE = ctypes.c_size_t(-1).value
H5G_GROUP, H5G_DATASET = 0, 1
H5_INDEX_NAME, H5_ITER_INC = 0, 0


class H5Error(OSError):
    pass


def _find():
    names = []
    found = ctypes.util.find_library('hdf5') or ctypes.util.find_library('hdf5_serial')
    if found:
        names.append(found)
    names += ['libhdf5.so', 'libhdf5_serial.so'] + list(_SEARCH)
    last = None
    for n in names:
        try:
            return ctypes.CDLL(n)
        except OSError as exc:
            last = exc
    raise ImportError('no HDF5 C library found (tried {}): {}'.format(', '.join(names), last))


def lib():
    """The loaded libhdf5 with argument / result types set; ImportError when there is none."""
    global _LIB
    if _LIB is not None:
        return _LIB
    L = _find()
    c_int, c_char_p, c_void_p, c_size_t = ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t
    P = ctypes.POINTER
    # hid_t is 64-bit from HDF5 1.10 on; a 1.8.x library returns 32-bit ids whose upper halves would be read as
    # garbage: H5get_libversion needs no hid_t, so ask it first and refuse anything older
    try:
        ver = [ctypes.c_uint(0) for _ in range(3)]
        L.H5get_libversion.argtypes, L.H5get_libversion.restype = [P(ctypes.c_uint)] * 3, c_int
        ok = L.H5get_libversion(*[ctypes.byref(v) for v in ver]) >= 0
    except AttributeError:
        ok = False
    version = tuple(v.value for v in ver) if ok else (0, 0, 0)
    if version < (1, 10, 0):
        raise ImportError('the HDF5 C library found is {} -- this binding needs >= 1.10 (64-bit hid_t)'.format(
            '.'.join(str(v) for v in version) if ok else 'of unknown version'))
    sig = {
        'H5open': ([], c_int), 'H5Eset_auto2': ([hid_t, c_void_p, c_void_p], c_int),
        'H5get_libversion': ([P(ctypes.c_uint)] * 3, c_int),
        'H5Fcreate': ([c_char_p, ctypes.c_uint, hid_t, hid_t], hid_t), 'H5Fopen': ([c_char_p, ctypes.c_uint, hid_t], hid_t),
        'H5Fclose': ([hid_t], c_int), 'H5Fis_hdf5': ([c_char_p], c_int),
        'H5Gcreate2': ([hid_t, c_char_p, hid_t, hid_t, hid_t], hid_t), 'H5Gopen2': ([hid_t, c_char_p, hid_t], hid_t),
        'H5Gclose': ([hid_t], c_int), 'H5Gget_num_objs': ([hid_t, P(hsize_t)], c_int),
        'H5Gget_objname_by_idx': ([hid_t, hsize_t, c_char_p, c_size_t], ctypes.c_ssize_t),
        'H5Gget_objtype_by_idx': ([hid_t, hsize_t], c_int),
        'H5Screate': ([c_int], hid_t), 'H5Screate_simple': ([c_int, P(hsize_t), P(hsize_t)], hid_t),
        'H5Sget_simple_extent_ndims': ([hid_t], c_int),
        'H5Sget_simple_extent_dims': ([hid_t, P(hsize_t), P(hsize_t)], c_int), 'H5Sclose': ([hid_t], c_int),
        'H5Dcreate2': ([hid_t, c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t], hid_t),
        'H5Dopen2': ([hid_t, c_char_p, hid_t], hid_t), 'H5Dclose': ([hid_t], c_int),
        'H5Dwrite': ([hid_t, hid_t, hid_t, hid_t, hid_t, c_void_p], c_int),
        'H5Dread': ([hid_t, hid_t, hid_t, hid_t, hid_t, c_void_p], c_int),
        'H5Dget_type': ([hid_t], hid_t), 'H5Dget_space': ([hid_t], hid_t),
        'H5Dvlen_reclaim': ([hid_t, hid_t, hid_t, c_void_p], c_int),
        'H5Tcopy': ([hid_t], hid_t), 'H5Tclose': ([hid_t], c_int), 'H5Tget_class': ([hid_t], c_int),
        'H5Tget_size': ([hid_t], c_size_t), 'H5Tget_sign': ([hid_t], c_int), 'H5Tset_size': ([hid_t, c_size_t], c_int),
        'H5Tset_cset': ([hid_t, c_int], c_int), 'H5Tis_variable_str': ([hid_t], c_int),
        'H5Tenum_create': ([hid_t], hid_t), 'H5Tenum_insert': ([hid_t, c_char_p, c_void_p], c_int),
        'H5Tget_super': ([hid_t], hid_t),
        'H5Acreate2': ([hid_t, c_char_p, hid_t, hid_t, hid_t, hid_t], hid_t), 'H5Awrite': ([hid_t, hid_t, c_void_p], c_int),
        'H5Aread': ([hid_t, hid_t, c_void_p], c_int), 'H5Aclose': ([hid_t], c_int),
        'H5Aget_type': ([hid_t], hid_t), 'H5Aget_space': ([hid_t], hid_t), 'H5Aget_num_attrs': ([hid_t], c_int),
        'H5Aopen_by_idx': ([hid_t, c_char_p, c_int, c_int, hsize_t, hid_t, hid_t], hid_t),
        'H5Aget_name': ([hid_t, c_size_t, c_char_p], ctypes.c_ssize_t), 'H5Aexists': ([hid_t, c_char_p], c_int),
        'H5Adelete': ([hid_t, c_char_p], c_int),
    }
    for name, (args, res) in sig.items():
        try:
            fn = getattr(L, name)
        except AttributeError:                             # e.g. a build without the deprecated H5Gget_*_by_idx /
            raise ImportError('the HDF5 C library {} lacks {} (built without deprecated symbols?)'.format(   # H5Dvlen_reclaim
                '.'.join(str(v) for v in version), name))
        fn.argtypes, fn.restype = args, res
    L._version = version
    if L.H5open() < 0:
        raise ImportError('H5open failed')
    L.H5Eset_auto2(0, None, None)                          # errors are reported through return values -> H5Error
    L._types = {}
    for key, sym in (('f8', 'H5T_NATIVE_DOUBLE_g'), ('f4', 'H5T_NATIVE_FLOAT_g'), ('i1', 'H5T_NATIVE_INT8_g'),
                     ('u1', 'H5T_NATIVE_UINT8_g'), ('i2', 'H5T_NATIVE_INT16_g'), ('u2', 'H5T_NATIVE_UINT16_g'),
                     ('i4', 'H5T_NATIVE_INT32_g'), ('u4', 'H5T_NATIVE_UINT32_g'), ('i8', 'H5T_NATIVE_INT64_g'),
                     ('u8', 'H5T_NATIVE_UINT64_g'), ('S', 'H5T_C_S1_g')):
        L._types[key] = hid_t.in_dll(L, sym).value
    _LIB = L
    return L


def _chk(rc, what):
    if rc < 0:
        raise H5Error('HDF5: {} failed'.format(what))
    return rc


def version():
    a, b, c = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
    lib().H5get_libversion(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    return (a.value, b.value, c.value)


def is_hdf5(fname):
    fname = str(fname)
    return os.path.isfile(fname) and lib().H5Fis_hdf5(fname.encode()) > 0


# ---- type mapping ---------------------------------------------------------------------------------------------
def _bool_type():
    L = lib()
    t = _chk(L.H5Tenum_create(L._types['i1']), 'H5Tenum_create')
    for name, val in ((b'FALSE', 0), (b'TRUE', 1)):
        v = ctypes.c_int8(val)
        _chk(L.H5Tenum_insert(t, name, ctypes.byref(v)), 'H5Tenum_insert')
    return t


def _vlen_str_type():
    L = lib()
    t = _chk(L.H5Tcopy(L._types['S']), 'H5Tcopy')
    _chk(L.H5Tset_size(t, H5T_VARIABLE), 'H5Tset_size')
    _chk(L.H5Tset_cset(t, H5T_CSET_UTF8), 'H5Tset_cset')
    return t


def _type_of(dtype):
    """(HDF5 type id, must be closed) for a numpy dtype."""
    L = lib()
    dtype = np.dtype(dtype)
    if dtype == np.bool_:
        return _bool_type(), True
    key = dtype.kind + str(dtype.itemsize)
    if dtype.kind in 'fiu' and key in L._types:
        return L._types[key], False
    raise TypeError('no HDF5 mapping for dtype {} (ndarrays of numbers and booleans only)'.format(dtype))


def _space_of(shape):
    L = lib()
    if len(shape) == 0:
        return _chk(L.H5Screate(H5S_SCALAR), 'H5Screate')
    dims = (hsize_t * len(shape))(*shape)
    return _chk(L.H5Screate_simple(len(shape), dims, None), 'H5Screate_simple')


def _shape_of(space):
    L = lib()
    nd = _chk(L.H5Sget_simple_extent_ndims(space), 'H5Sget_simple_extent_ndims')
    if nd == 0:
        return ()
    dims = (hsize_t * nd)()
    _chk(L.H5Sget_simple_extent_dims(space, dims, None), 'H5Sget_simple_extent_dims')
    return tuple(int(d) for d in dims)


def _read(obj, tid, space, reader):
    """Value of a dataset / attribute whose file type is ``tid``: ndarray (numpy scalar for a scalar attribute is
    the caller's business), str for strings.  ``reader(memtype, buffer)`` performs the H5Dread / H5Aread."""
    L = lib()
    shape = _shape_of(space)
    cls = L.H5Tget_class(tid)
    n = int(np.prod(shape)) if shape else 1
    if cls == H5T_STRING:
        if L.H5Tis_variable_str(tid) > 0:
            mt = _vlen_str_type()
            try:
                buf = (ctypes.c_char_p * n)()
                _chk(reader(mt, buf), 'read (string)')
                vals = [(b.decode('utf-8') if b is not None else '') for b in buf]
                L.H5Dvlen_reclaim(mt, space, H5P_DEFAULT, buf)
            finally:
                L.H5Tclose(mt)
        else:
            size = int(L.H5Tget_size(tid))
            raw = ctypes.create_string_buffer(size * n)
            mt = _chk(L.H5Tcopy(tid), 'H5Tcopy')
            try:
                _chk(reader(mt, raw), 'read (string)')
            finally:
                L.H5Tclose(mt)
            vals = [raw.raw[i * size:(i + 1) * size].split(b'\0')[0].decode('utf-8') for i in range(n)]
        return vals[0] if shape == () else np.array(vals, dtype=object).reshape(shape)
    if cls == H5T_ENUM:                                    # h5py's bool
        base = L.H5Tget_super(tid)
        size = int(L.H5Tget_size(base))
        L.H5Tclose(base)
        if size != 1:
            raise TypeError('enum over a {}-byte integer: only h5py booleans are supported'.format(size))
        mt = _bool_type()
        try:
            out = np.empty(shape, dtype=np.int8)
            _chk(reader(mt, out.ctypes.data_as(ctypes.c_void_p)), 'read (bool)')
        finally:
            L.H5Tclose(mt)
        return out.astype(np.bool_)
    if cls in (H5T_INTEGER, H5T_FLOAT):
        size = int(L.H5Tget_size(tid))
        kind = 'f' if cls == H5T_FLOAT else ('i' if L.H5Tget_sign(tid) else 'u')
        key = kind + str(size)
        if key not in L._types:
            raise TypeError('unsupported HDF5 number type ({} bytes)'.format(size))
        out = np.empty(shape, dtype=np.dtype(key))
        if out.size:
            _chk(reader(L._types[key], out.ctypes.data_as(ctypes.c_void_p)), 'read')
        return out
    raise TypeError('unsupported HDF5 type class {}'.format(cls))


# ---- objects --------------------------------------------------------------------------------------------------
class Dataset(object):
    def __init__(self, parent, name):
        self._parent, self._name = parent, name.encode()

    def _open(self):
        return _chk(lib().H5Dopen2(self._parent._id, self._name, H5P_DEFAULT), 'H5Dopen2')

    @property
    def shape(self):
        L, d = lib(), self._open()
        try:
            sp = L.H5Dget_space(d)
            try:
                return _shape_of(sp)
            finally:
                L.H5Sclose(sp)
        finally:
            L.H5Dclose(d)

    def __getitem__(self, key):
        if key != () and key is not Ellipsis:
            raise NotImplementedError('whole-dataset reads only: ds[()] or ds[...]')
        L, d = lib(), self._open()
        try:
            tid, sp = L.H5Dget_type(d), L.H5Dget_space(d)
            try:
                return _read(d, tid, sp, lambda mt, buf: L.H5Dread(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf))
            finally:
                L.H5Tclose(tid)
                L.H5Sclose(sp)
        finally:
            L.H5Dclose(d)

    def __setitem__(self, key, value):
        if key != () and key is not Ellipsis:
            raise NotImplementedError('whole-dataset writes only: ds[...] = array')
        arr = np.asarray(value, order='C')
        if arr.shape != self.shape:
            raise ValueError('shape {} does not match the dataset {}'.format(arr.shape, self.shape))
        L, d = lib(), self._open()
        try:
            mt, own = _type_of(arr.dtype)
            try:
                if arr.dtype == np.bool_:
                    arr = arr.astype(np.int8)
                if arr.size:
                    _chk(L.H5Dwrite(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data_as(ctypes.c_void_p)),
                         'H5Dwrite')
            finally:
                if own:
                    L.H5Tclose(mt)
        finally:
            L.H5Dclose(d)


class Attributes(object):
    def __init__(self, group):
        self._g = group

    def __setitem__(self, key, value):
        L, gid = lib(), self._g._id
        name = key.encode()
        if L.H5Aexists(gid, name) > 0:
            _chk(L.H5Adelete(gid, name), 'H5Adelete')
        if isinstance(value, (str, bytes)):
            text = value.encode('utf-8') if isinstance(value, str) else value
            tid, sp = _vlen_str_type(), _space_of(())
            try:
                a = _chk(L.H5Acreate2(gid, name, tid, sp, H5P_DEFAULT, H5P_DEFAULT), 'H5Acreate2')
                try:
                    buf = (ctypes.c_char_p * 1)(text)
                    _chk(L.H5Awrite(a, tid, buf), 'H5Awrite')
                finally:
                    L.H5Aclose(a)
            finally:
                L.H5Tclose(tid)
                L.H5Sclose(sp)
            return
        arr = np.asarray(value)
        if arr.dtype == object or arr.dtype.kind in 'US':
            raise TypeError('attribute {!r}: numbers, booleans, strings and arrays of numbers only'.format(key))
        arr = np.asarray(arr, order='C')
        tid, own = _type_of(arr.dtype)
        sp = _space_of(arr.shape)
        try:
            a = _chk(L.H5Acreate2(gid, name, tid, sp, H5P_DEFAULT, H5P_DEFAULT), 'H5Acreate2')
            try:
                raw = arr.astype(np.int8) if arr.dtype == np.bool_ else arr
                if raw.size:
                    _chk(L.H5Awrite(a, tid, raw.ctypes.data_as(ctypes.c_void_p)), 'H5Awrite')
            finally:
                L.H5Aclose(a)
        finally:
            if own:
                L.H5Tclose(tid)
            L.H5Sclose(sp)

    def _each(self):
        L, gid = lib(), self._g._id
        n = _chk(L.H5Aget_num_attrs(gid), 'H5Aget_num_attrs')
        for i in range(n):
            a = _chk(L.H5Aopen_by_idx(gid, b'.', H5_INDEX_NAME, H5_ITER_INC, i, H5P_DEFAULT, H5P_DEFAULT),
                     'H5Aopen_by_idx')
            try:
                size = _chk(L.H5Aget_name(a, 0, None), 'H5Aget_name')
                buf = ctypes.create_string_buffer(size + 1)
                L.H5Aget_name(a, size + 1, buf)
                tid, sp = L.H5Aget_type(a), L.H5Aget_space(a)
                try:
                    val = _read(a, tid, sp, lambda mt, b: L.H5Aread(a, mt, b))
                finally:
                    L.H5Tclose(tid)
                    L.H5Sclose(sp)
                if isinstance(val, np.ndarray) and val.shape == ():
                    val = val[()]                          # numpy scalar, as h5py returns it
                yield buf.value.decode('utf-8'), val
            finally:
                L.H5Aclose(a)

    def items(self):
        return list(self._each())

    def keys(self):
        return [k for k, _ in self._each()]

    def __getitem__(self, key):
        for k, v in self._each():
            if k == key:
                return v
        raise KeyError(key)

    def __contains__(self, key):
        return lib().H5Aexists(self._g._id, key.encode()) > 0

    def __len__(self):
        return _chk(lib().H5Aget_num_attrs(self._g._id), 'H5Aget_num_attrs')


class Group(object):
    def __init__(self, gid, keep=None):
        self._id = gid
        self._keep = keep                                   # the file object, so that it outlives its groups
        if keep is not None:
            keep._children.append(self)                     # ... and closes them with itself
        self.attrs = Attributes(self)

    def _close(self):
        if self._id is not None and not isinstance(self, File):
            lib().H5Gclose(self._id)
            self._id = None

    def __del__(self):
        try:
            self._close()
        except Exception:                                  # noqa: BLE001 -- interpreter shutdown
            pass

    def create_group(self, path):
        gid = lib().H5Gcreate2(self._id, path.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        if gid < 0:
            raise ValueError('unable to create group {!r} (exists already, or its parent does not)'.format(path))
        return Group(gid, keep=self._keep or self)

    def create_dataset(self, name, shape=None, dtype=None, data=None):
        L = lib()
        if data is not None:
            data = np.asarray(data, order='C')
            shape = data.shape if shape is None else tuple(np.atleast_1d(shape))
            dtype = data.dtype if dtype is None else np.dtype(dtype)
        shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        tid, own = _type_of(np.dtype(dtype if dtype is not None else 'f4'))
        sp = _space_of(shape)
        try:
            d = L.H5Dcreate2(self._id, name.encode(), tid, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
            if d < 0:
                raise ValueError('unable to create dataset {!r} (name already exists?)'.format(name))
            L.H5Dclose(d)
        finally:
            if own:
                L.H5Tclose(tid)
            L.H5Sclose(sp)
        ds = Dataset(self, name)
        if data is not None:
            ds[...] = data.astype(dtype, copy=False).reshape(shape)
        return ds

    def __getitem__(self, path):
        L = lib()
        parts = [p for p in path.split('/') if p]
        if not parts:
            return self
        base = self
        for i, part in enumerate(parts):
            kind = None
            n = hsize_t()
            _chk(L.H5Gget_num_objs(base._id, ctypes.byref(n)), 'H5Gget_num_objs')
            for j in range(n.value):
                if base._name_at(j) == part:
                    kind = L.H5Gget_objtype_by_idx(base._id, j)
                    break
            if kind is None:
                raise KeyError(path)
            if kind == H5G_DATASET:
                if i != len(parts) - 1:
                    raise KeyError(path)
                return Dataset(base, part)
            gid = _chk(L.H5Gopen2(base._id, part.encode(), H5P_DEFAULT), 'H5Gopen2')
            base = Group(gid, keep=self._keep or self)
        return base

    def _name_at(self, j):
        L = lib()
        size = _chk(L.H5Gget_objname_by_idx(self._id, j, None, 0), 'H5Gget_objname_by_idx')
        buf = ctypes.create_string_buffer(size + 1)
        L.H5Gget_objname_by_idx(self._id, j, buf, size + 1)
        return buf.value.decode('utf-8')

    def keys(self):
        n = hsize_t()
        _chk(lib().H5Gget_num_objs(self._id, ctypes.byref(n)), 'H5Gget_num_objs')
        return [self._name_at(j) for j in range(n.value)]

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def __contains__(self, path):
        try:
            self[path]
            return True
        except KeyError:
            return False


class File(Group):
    def __init__(self, fname, mode='r'):
        L = lib()
        fname = str(fname)
        if mode == 'w':
            fid = L.H5Fcreate(fname.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        elif mode == 'r':
            fid = L.H5Fopen(fname.encode(), H5F_ACC_RDONLY, H5P_DEFAULT)
        else:
            raise ValueError("mode 'r' or 'w'")
        if fid < 0:
            raise OSError('unable to open {!r} (mode {})'.format(fname, mode))
        self._children = []
        Group.__init__(self, fid)
        self.filename = fname

    def close(self):
        if self._id is not None:
            for g in self._children:
                g._close()
            self._children = []
            lib().H5Fclose(self._id)
            self._id = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:                                  # noqa: BLE001
            pass
