"""Minimal read-only HDF5 access through ctypes on the system's libhdf5 (no h5py / netCDF4 in this image).

The reference writes its stores with netCDF4 (multistatereporter.py:280-460), whose NETCDF4 format is an HDF5 file:
variables are datasets, groups are groups, global attributes are attributes of '/'.  This module reads what a reader of
such a store needs: numeric datasets of any rank (converted by the library to native types), character arrays
(`S1` variables of the fixed-dimension dictionaries, multistatereporter.py:1817-1880), variable-length strings
(`str` variables: options, mcmc_moves, timestamp) and string / numeric attributes.  Nothing is written.

The library is looked for in REMD_HDF5_LIB, the loader path, then /opt/conda/lib.  `available()` says whether one
was found; every other entry point raises ImportError without it.
"""
import ctypes
import ctypes.util
import os

import numpy as np

_lib = None
_hid = ctypes.c_int64          # hid_t is 64 bits since HDF5 1.10
_H5F_ACC_RDONLY = 0
_H5P_DEFAULT = 0
_H5S_ALL = 0
# H5T_class_t
_INTEGER, _FLOAT, _STRING = 0, 1, 3


def _find():
    names = [os.environ.get('REMD_HDF5_LIB'), ctypes.util.find_library('hdf5'), 'libhdf5.so',
             '/opt/conda/lib/libhdf5.so', '/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so']
    for n in names:
        if not n:
            continue
        try:
            return ctypes.CDLL(n)
        except OSError:
            continue
    return None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    lib = _find()
    if lib is None:
        raise ImportError('no libhdf5 found (set REMD_HDF5_LIB): stores written by the reference cannot be read')
    lib.H5open()
    maj, mnr, rel = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
    lib.H5get_libversion(ctypes.byref(maj), ctypes.byref(mnr), ctypes.byref(rel))
    if (maj.value, mnr.value) < (1, 10):
        raise ImportError('libhdf5 %d.%d is older than 1.10 (32-bit identifiers)' % (maj.value, mnr.value))
    sig = {
        'H5Fopen': (_hid, [ctypes.c_char_p, ctypes.c_uint, _hid]),
        'H5Fclose': (ctypes.c_int, [_hid]),
        'H5Oopen': (_hid, [_hid, ctypes.c_char_p, _hid]),
        'H5Oclose': (ctypes.c_int, [_hid]),
        'H5Iget_type': (ctypes.c_int, [_hid]),
        'H5Lexists': (ctypes.c_int, [_hid, ctypes.c_char_p, _hid]),
        'H5Dopen2': (_hid, [_hid, ctypes.c_char_p, _hid]),
        'H5Dclose': (ctypes.c_int, [_hid]),
        'H5Dget_space': (_hid, [_hid]),
        'H5Dget_type': (_hid, [_hid]),
        'H5Dread': (ctypes.c_int, [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]),
        'H5Dvlen_reclaim': (ctypes.c_int, [_hid, _hid, _hid, ctypes.c_void_p]),
        'H5Sget_simple_extent_ndims': (ctypes.c_int, [_hid]),
        'H5Sget_simple_extent_dims': (ctypes.c_int, [_hid, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
        'H5Sget_simple_extent_npoints': (ctypes.c_int64, [_hid]),
        'H5Sclose': (ctypes.c_int, [_hid]),
        'H5Tget_class': (ctypes.c_int, [_hid]),
        'H5Tget_size': (ctypes.c_size_t, [_hid]),
        'H5Tget_sign': (ctypes.c_int, [_hid]),
        'H5Tis_variable_str': (ctypes.c_int, [_hid]),
        'H5Tcopy': (_hid, [_hid]),
        'H5Tset_size': (ctypes.c_int, [_hid, ctypes.c_size_t]),
        'H5Tset_cset': (ctypes.c_int, [_hid, ctypes.c_int]),
        'H5Tget_cset': (ctypes.c_int, [_hid]),
        'H5Tclose': (ctypes.c_int, [_hid]),
        'H5Aexists': (ctypes.c_int, [_hid, ctypes.c_char_p]),
        'H5Aopen': (_hid, [_hid, ctypes.c_char_p, _hid]),
        'H5Aclose': (ctypes.c_int, [_hid]),
        'H5Aget_type': (_hid, [_hid]),
        'H5Aget_space': (_hid, [_hid]),
        'H5Aread': (ctypes.c_int, [_hid, _hid, ctypes.c_void_p]),
        'H5Gget_num_objs': (ctypes.c_int, [_hid, ctypes.POINTER(ctypes.c_uint64)]),
        'H5Gget_objname_by_idx': (ctypes.c_ssize_t, [_hid, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_size_t]),
        'H5Gget_objtype_by_idx': (ctypes.c_int, [_hid, ctypes.c_uint64]),
        'H5Eset_auto2': (ctypes.c_int, [_hid, ctypes.c_void_p, ctypes.c_void_p]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    lib.H5Eset_auto2(0, None, None)           # errors come back as return codes; no stack dumps on stderr
    _lib = lib
    return lib


def available():
    try:
        _load()
        return True
    except ImportError:
        return False


def _native(name):
    return _hid.in_dll(_load(), name).value


def _numpy_type(lib, tid):
    """(numpy dtype, native HDF5 memory type) for an integer / float file type."""
    cls, size = lib.H5Tget_class(tid), lib.H5Tget_size(tid)
    if cls == _FLOAT:
        return {4: (np.float32, 'H5T_NATIVE_FLOAT_g'), 8: (np.float64, 'H5T_NATIVE_DOUBLE_g')}[size]
    if cls == _INTEGER:
        signed = lib.H5Tget_sign(tid) != 0
        table = {(1, True): (np.int8, 'H5T_NATIVE_SCHAR_g'), (1, False): (np.uint8, 'H5T_NATIVE_UCHAR_g'),
                 (2, True): (np.int16, 'H5T_NATIVE_SHORT_g'), (2, False): (np.uint16, 'H5T_NATIVE_USHORT_g'),
                 (4, True): (np.int32, 'H5T_NATIVE_INT_g'), (4, False): (np.uint32, 'H5T_NATIVE_UINT_g'),
                 (8, True): (np.int64, 'H5T_NATIVE_LLONG_g'), (8, False): (np.uint64, 'H5T_NATIVE_ULLONG_g')}
        return table[(size, signed)]
    raise TypeError('unsupported HDF5 type class %d' % cls)


def _shape(lib, sid):
    nd = lib.H5Sget_simple_extent_ndims(sid)
    if nd <= 0:
        return ()
    dims = (ctypes.c_uint64 * nd)()
    lib.H5Sget_simple_extent_dims(sid, dims, None)
    return tuple(int(d) for d in dims)


def _read(lib, tid, sid, reader, reclaim):
    """Shared body of dataset and attribute reads: `reader(memory_type, buffer)` fills the buffer."""
    shape = _shape(lib, sid)
    n = int(lib.H5Sget_simple_extent_npoints(sid))
    cls = lib.H5Tget_class(tid)
    if cls == _STRING:
        mt = lib.H5Tcopy(_native('H5T_C_S1_g'))
        lib.H5Tset_cset(mt, lib.H5Tget_cset(tid))
        try:
            if lib.H5Tis_variable_str(tid) > 0:
                lib.H5Tset_size(mt, ctypes.c_size_t(-1).value)        # H5T_VARIABLE
                buf = (ctypes.c_char_p * max(n, 1))()
                if n and reader(mt, buf) < 0:
                    raise IOError('HDF5 read failed')
                out = [(buf[i] or b'').decode('utf-8') for i in range(n)]
                if n:
                    reclaim(mt, buf)
                return out[0] if shape == () else np.array(out, dtype=object).reshape(shape)
            size = lib.H5Tget_size(tid)
            lib.H5Tset_size(mt, size)
            raw = ctypes.create_string_buffer(max(n * size, 1))
            if n and reader(mt, raw) < 0:
                raise IOError('HDF5 read failed')
            if size == 1:                                             # netCDF 'S1' character array
                return np.frombuffer(raw.raw[:n], dtype='S1').reshape(shape).copy()
            items = [raw.raw[i * size:(i + 1) * size].split(b'\0', 1)[0].decode('utf-8') for i in range(n)]
            return items[0] if shape == () else np.array(items, dtype=object).reshape(shape)
        finally:
            lib.H5Tclose(mt)
    dtype, native = _numpy_type(lib, tid)
    out = np.empty(shape, dtype=dtype)
    if n and reader(_native(native), out.ctypes.data_as(ctypes.c_void_p)) < 0:
        raise IOError('HDF5 read failed')
    return out


class File:
    """A store opened read-only.  Paths are '/'-separated, like netCDF4's group / variable nesting."""

    def __init__(self, path):
        self._lib = _load()
        self.path = str(path)
        self._fid = self._lib.H5Fopen(self.path.encode(), _H5F_ACC_RDONLY, _H5P_DEFAULT)
        if self._fid < 0:
            raise IOError('cannot open %s as HDF5 / netCDF4' % path)

    def close(self):
        if self._fid >= 0:
            self._lib.H5Fclose(self._fid)
            self._fid = -1

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __contains__(self, path):
        """True when every link along the path exists (H5Lexists needs each prefix checked in turn)."""
        parts = [p for p in path.strip('/').split('/') if p]
        cur = ''
        for p in parts:
            cur += '/' + p
            if self._lib.H5Lexists(self._fid, cur.encode(), _H5P_DEFAULT) <= 0:
                return False
        return True

    def _kind(self, path):
        oid = self._lib.H5Oopen(self._fid, path.encode(), _H5P_DEFAULT)
        if oid < 0:
            raise KeyError(path)
        t = self._lib.H5Iget_type(oid)
        self._lib.H5Oclose(oid)
        return {2: 'group', 5: 'dataset'}.get(t, 'other')

    def is_group(self, path):
        return path in ('', '/') or (path in self and self._kind(path) == 'group')

    def keys(self, path='/'):
        """(groups, datasets) directly under a group, in the file's index order."""
        lib = self._lib
        gid = lib.H5Oopen(self._fid, (path or '/').encode(), _H5P_DEFAULT)
        if gid < 0:
            raise KeyError(path)
        try:
            n = ctypes.c_uint64()
            lib.H5Gget_num_objs(gid, ctypes.byref(n))
            groups, datasets = [], []
            for i in range(n.value):
                ln = lib.H5Gget_objname_by_idx(gid, i, None, 0)
                buf = ctypes.create_string_buffer(ln + 1)
                lib.H5Gget_objname_by_idx(gid, i, buf, ln + 1)
                kind = lib.H5Gget_objtype_by_idx(gid, i)      # H5G_GROUP = 0, H5G_DATASET = 1
                (groups if kind == 0 else datasets if kind == 1 else []).append(buf.value.decode())
            return groups, datasets
        finally:
            lib.H5Oclose(gid)

    def shape(self, path):
        lib = self._lib
        did = lib.H5Dopen2(self._fid, path.encode(), _H5P_DEFAULT)
        if did < 0:
            raise KeyError(path)
        sid = lib.H5Dget_space(did)
        try:
            return _shape(lib, sid)
        finally:
            lib.H5Sclose(sid)
            lib.H5Dclose(did)

    def read(self, path):
        """Whole dataset as a numpy array (numbers, 'S1' characters) or an object array / str of strings."""
        lib = self._lib
        did = lib.H5Dopen2(self._fid, path.encode(), _H5P_DEFAULT)
        if did < 0:
            raise KeyError(path)
        tid, sid = lib.H5Dget_type(did), lib.H5Dget_space(did)
        try:
            return _read(lib, tid, sid,
                         lambda mt, buf: lib.H5Dread(did, mt, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT, buf),
                         lambda mt, buf: lib.H5Dvlen_reclaim(mt, sid, _H5P_DEFAULT, buf))
        finally:
            lib.H5Tclose(tid)
            lib.H5Sclose(sid)
            lib.H5Dclose(did)

    def attr(self, name, path='/', default=None):
        lib = self._lib
        oid = lib.H5Oopen(self._fid, (path or '/').encode(), _H5P_DEFAULT)
        if oid < 0:
            raise KeyError(path)
        try:
            if lib.H5Aexists(oid, name.encode()) <= 0:
                return default
            aid = lib.H5Aopen(oid, name.encode(), _H5P_DEFAULT)
            tid, sid = lib.H5Aget_type(aid), lib.H5Aget_space(aid)
            try:
                return _read(lib, tid, sid, lambda mt, buf: lib.H5Aread(aid, mt, buf),
                             lambda mt, buf: lib.H5Dvlen_reclaim(mt, sid, _H5P_DEFAULT, buf))
            finally:
                lib.H5Tclose(tid)
                lib.H5Sclose(sid)
                lib.H5Aclose(aid)
        finally:
            lib.H5Oclose(oid)
