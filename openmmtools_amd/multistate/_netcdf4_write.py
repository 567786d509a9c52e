"""Writing the netCDF-4 flavour of HDF5 through ctypes on libhdf5 + libhdf5_hl (no netCDF4 / h5py in this image).

A netCDF-4 file is an HDF5 file with conventions (NetCDF-4 format specification, "HDF5 dimension scales"):
    dimension   a dataset that is an HDF5 dimension scale; one without a coordinate variable has type IEEE_F32BE and the NAME
                "This is a netCDF dimension but not a netCDF variable.%10d" (its length), unlimited ones are chunked with an
                unlimited maximum extent; `_Netcdf4Dimid` numbers the dimensions of the file
    variable    a dataset with the scales of its dimensions attached (DIMENSION_LIST / REFERENCE_LIST, written by
                H5DSattach_scale) and `_Netcdf4Coordinates` = its dimension ids; NC_CHAR = fixed strings of length 1, NC_STRING =
                variable-length UTF-8 strings; record variables are chunked with one record per chunk
    group       a group; link and attribute creation order are tracked so that netCDF lists objects in creation order
    attribute   text attributes are fixed-length null-terminated ASCII strings on a scalar space
This is the object structure `h5dump -H` shows for the stores the reference writes (multistatereporter.py:280-460 through
netCDF4-python); `tests/test_reference_store.py` compares the two.  Reading goes through `_hdf5.File` (the same handle).
"""
import ctypes
import ctypes.util
import os

import numpy as np

from . import _hdf5

_hid = ctypes.c_int64
_hsize = ctypes.c_uint64
_UNLIMITED = 0xFFFFFFFFFFFFFFFF
_H5F_ACC_RDWR, _H5F_ACC_TRUNC = 1, 2
_CRT_ORDER = 3                      # H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED
_lib = None
_hl = None

_FILE_TYPES = {'f8': 'H5T_IEEE_F64LE_g', 'f4': 'H5T_IEEE_F32LE_g', 'i8': 'H5T_STD_I64LE_g', 'i4': 'H5T_STD_I32LE_g',
               'i1': 'H5T_STD_I8LE_g'}
_MEM_TYPES = {'f8': ('H5T_NATIVE_DOUBLE_g', np.float64), 'f4': ('H5T_NATIVE_FLOAT_g', np.float32),
              'i8': ('H5T_NATIVE_LLONG_g', np.int64), 'i4': ('H5T_NATIVE_INT_g', np.int32), 'i1': ('H5T_NATIVE_SCHAR_g', np.int8)}


def _load():
    global _lib, _hl
    if _lib is not None:
        return _lib, _hl
    lib = _hdf5._load()
    base = getattr(lib, '_name', '') or ''
    names = [os.environ.get('REMD_HDF5_HL_LIB'), base.replace('libhdf5', 'libhdf5_hl') if 'libhdf5' in base else None,
             ctypes.util.find_library('hdf5_hl'), 'libhdf5_hl.so', '/opt/conda/lib/libhdf5_hl.so']
    hl = None
    for n in names:
        if not n:
            continue
        try:
            hl = ctypes.CDLL(n)
            break
        except OSError:
            continue
    if hl is None:
        raise ImportError('no libhdf5_hl found (set REMD_HDF5_HL_LIB): the reference\'s netCDF4 layout cannot be written')
    P = ctypes.POINTER
    sig = {
        'H5Fcreate': (_hid, [ctypes.c_char_p, ctypes.c_uint, _hid, _hid]),
        'H5Fflush': (ctypes.c_int, [_hid, ctypes.c_int]),
        'H5Pcreate': (_hid, [_hid]),
        'H5Pclose': (ctypes.c_int, [_hid]),
        'H5Pset_chunk': (ctypes.c_int, [_hid, ctypes.c_int, P(_hsize)]),
        'H5Pset_link_creation_order': (ctypes.c_int, [_hid, ctypes.c_uint]),
        'H5Pset_attr_creation_order': (ctypes.c_int, [_hid, ctypes.c_uint]),
        'H5Gcreate2': (_hid, [_hid, ctypes.c_char_p, _hid, _hid, _hid]),
        'H5Gclose': (ctypes.c_int, [_hid]),
        'H5Screate': (_hid, [ctypes.c_int]),
        'H5Screate_simple': (_hid, [ctypes.c_int, P(_hsize), P(_hsize)]),
        'H5Sselect_hyperslab': (ctypes.c_int, [_hid, ctypes.c_int, P(_hsize), P(_hsize), P(_hsize), P(_hsize)]),
        'H5Dcreate2': (_hid, [_hid, ctypes.c_char_p, _hid, _hid, _hid, _hid, _hid]),
        'H5Dset_extent': (ctypes.c_int, [_hid, P(_hsize)]),
        'H5Dwrite': (ctypes.c_int, [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]),
        'H5Acreate2': (_hid, [_hid, ctypes.c_char_p, _hid, _hid, _hid, _hid]),
        'H5Awrite': (ctypes.c_int, [_hid, _hid, ctypes.c_void_p]),
        'H5Adelete': (ctypes.c_int, [_hid, ctypes.c_char_p]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    hl.H5DSset_scale.restype, hl.H5DSset_scale.argtypes = ctypes.c_int, [_hid, ctypes.c_char_p]
    hl.H5DSattach_scale.restype, hl.H5DSattach_scale.argtypes = ctypes.c_int, [_hid, _hid, ctypes.c_uint]
    _lib, _hl = lib, hl
    return lib, hl


def available():
    try:
        _load()
        return True
    except ImportError:
        return False


def _g(name):
    return _hid.in_dll(_lib, name).value


def _ck(rc, what):
    if rc < 0:
        raise IOError('HDF5: %s failed' % what)
    return rc


class NetCDF4File(_hdf5.File):
    """A netCDF-4 file opened for writing ('w': create / truncate, 'a': extend an existing one).  Dimensions and variables are
    addressed by '/'-separated paths; a variable's dimensions are looked up in its own group, then in the root."""

    def __init__(self, path, mode='w'):
        lib, hl = _load()
        self._lib, self._hl = lib, hl
        self.path = str(path)
        self._dims = {}                 # '/group/name' -> (length or None, dimension id)
        self._vars = {}                 # '/group/name' -> (kind, dimension paths)
        self._open = {}                 # open dataset identifiers
        if mode == 'w':
            fcpl = lib.H5Pcreate(_g('H5P_CLS_FILE_CREATE_ID_g'))
            lib.H5Pset_link_creation_order(fcpl, _CRT_ORDER)
            lib.H5Pset_attr_creation_order(fcpl, _CRT_ORDER)
            self._fid = lib.H5Fcreate(self.path.encode(), _H5F_ACC_TRUNC, fcpl, 0)
            lib.H5Pclose(fcpl)
            if self._fid < 0:
                raise IOError('cannot create %s' % path)
            self.set_attr('/', '_NCProperties', 'version=2,netcdf=4.8.1,hdf5=1.10.4')
        elif mode == 'a':
            self._fid = lib.H5Fopen(self.path.encode(), _H5F_ACC_RDWR, 0)
            if self._fid < 0:
                raise IOError('cannot open %s for writing' % path)
            self._rediscover('/')
        else:
            raise ValueError("mode must be 'w' or 'a'")

    # ---- bookkeeping -------------------------------------------------------------------------------------------
    def _rediscover(self, group):
        """Dimensions and variables of an existing file (mode 'a'): datasets with CLASS = DIMENSION_SCALE are dimensions."""
        groups, datasets = self.keys(group)
        for name in datasets:
            path = (group.rstrip('/') + '/' + name)
            cls = self.attr('CLASS', path)
            if cls is not None and 'DIMENSION_SCALE' in str(np.asarray(cls).reshape(-1)[0]):
                shape, maxshape = self._extents(path)
                dimid = int(np.asarray(self.attr('_Netcdf4Dimid', path, default=len(self._dims))).reshape(-1)[0])
                self._dims[path] = (None if maxshape[0] == _UNLIMITED else int(shape[0]), dimid)
        for name in datasets:
            path = (group.rstrip('/') + '/' + name)
            if path not in self._dims:
                self._vars[path] = None
        for g in groups:
            self._rediscover(group.rstrip('/') + '/' + g)

    def _extents(self, path):
        lib = self._lib
        did = self._dataset(path)
        sid = lib.H5Dget_space(did)
        n = lib.H5Sget_simple_extent_ndims(sid)
        dims, maxd = (_hsize * max(1, n))(), (_hsize * max(1, n))()
        lib.H5Sget_simple_extent_dims(sid, dims, maxd)
        lib.H5Sclose(sid)
        return [int(dims[k]) for k in range(n)], [int(maxd[k]) for k in range(n)]

    def _dataset(self, path):
        if path not in self._open:
            did = self._lib.H5Dopen2(self._fid, path.encode(), 0)
            if did < 0:
                raise KeyError(path)
            self._open[path] = did
        return self._open[path]

    def flush(self):
        _ck(self._lib.H5Fflush(self._fid, 1), 'H5Fflush')       # H5F_SCOPE_GLOBAL

    def close(self):
        for did in self._open.values():
            self._lib.H5Dclose(did)
        self._open = {}
        super().close()

    # ---- groups, dimensions ------------------------------------------------------------------------------------
    def create_group(self, path):
        if path in self:
            return
        parent = path.rsplit('/', 1)[0]
        if parent and parent not in self:
            self.create_group(parent)
        lib = self._lib
        gcpl = lib.H5Pcreate(_g('H5P_CLS_GROUP_CREATE_ID_g'))
        lib.H5Pset_link_creation_order(gcpl, _CRT_ORDER)
        lib.H5Pset_attr_creation_order(gcpl, _CRT_ORDER)
        gid = lib.H5Gcreate2(self._fid, path.encode(), 0, gcpl, 0)
        lib.H5Pclose(gcpl)
        if gid < 0:
            raise IOError('cannot create group %s' % path)
        lib.H5Gclose(gid)

    def has_dimension(self, path):
        return path in self._dims

    def create_dimension(self, path, size):
        """size None: unlimited (the record dimension).  Returns the dimension's path."""
        if path in self._dims:
            return path
        lib = self._lib
        n = 0 if size is None else int(size)
        dims, maxd = (_hsize * 1)(n), (_hsize * 1)(_UNLIMITED if size is None else n)
        sid = lib.H5Screate_simple(1, dims, maxd)
        dcpl = lib.H5Pcreate(_g('H5P_CLS_DATASET_CREATE_ID_g'))
        lib.H5Pset_attr_creation_order(dcpl, _CRT_ORDER)
        if size is None:
            lib.H5Pset_chunk(dcpl, 1, (_hsize * 1)(1024))
        did = lib.H5Dcreate2(self._fid, path.encode(), _g('H5T_IEEE_F32BE_g'), sid, 0, dcpl, 0)
        lib.H5Pclose(dcpl)
        lib.H5Sclose(sid)
        if did < 0:
            raise IOError('cannot create dimension %s' % path)
        _ck(self._hl.H5DSset_scale(did, ('This is a netCDF dimension but not a netCDF variable.%10d' % n).encode()), 'H5DSset_scale(%s)' % path)
        self._open[path] = did
        dimid = len(self._dims)
        self._dims[path] = (None if size is None else n, dimid)
        self._set_numeric_attr(did, '_Netcdf4Dimid', np.array(dimid, dtype=np.int32), scalar=True)
        return path

    def _resolve_dim(self, group, name):
        for cand in ((group.rstrip('/') + '/' + name), '/' + name):
            if cand in self._dims:
                return cand
        raise KeyError('dimension %s not defined in %s or /' % (name, group or '/'))

    # ---- variables ---------------------------------------------------------------------------------------------
    def create_variable(self, path, kind, dimensions):
        """kind: 'f8', 'f4', 'i8', 'i4', 'i1', 'S1' (NC_CHAR) or 'str' (NC_STRING); dimensions: names, resolved in the
        variable's group and then in the root."""
        if path in self._vars:
            return
        lib = self._lib
        group = path.rsplit('/', 1)[0]
        dpaths = [self._resolve_dim(group, d) for d in dimensions]
        lens = [self._dims[d][0] for d in dpaths]
        dims = (_hsize * len(lens))(*[0 if l is None else l for l in lens])
        maxd = (_hsize * len(lens))(*[_UNLIMITED if l is None else l for l in lens])
        sid = lib.H5Screate_simple(len(lens), dims, maxd)
        dcpl = lib.H5Pcreate(_g('H5P_CLS_DATASET_CREATE_ID_g'))
        lib.H5Pset_attr_creation_order(dcpl, _CRT_ORDER)
        if any(l is None for l in lens):
            lib.H5Pset_chunk(dcpl, len(lens), (_hsize * len(lens))(*[1 if l is None else max(1, l) for l in lens]))
        tid, own = self._file_type(kind)
        did = lib.H5Dcreate2(self._fid, path.encode(), tid, sid, 0, dcpl, 0)
        if own:
            lib.H5Tclose(tid)
        lib.H5Pclose(dcpl)
        lib.H5Sclose(sid)
        if did < 0:
            raise IOError('cannot create variable %s' % path)
        for k, d in enumerate(dpaths):
            _ck(self._hl.H5DSattach_scale(did, self._dataset(d), k), 'H5DSattach_scale(%s, %s)' % (path, d))
        self._set_numeric_attr(did, '_Netcdf4Coordinates', np.array([self._dims[d][1] for d in dpaths], dtype=np.int32))
        self._open[path] = did
        self._vars[path] = (kind, dpaths)

    def _file_type(self, kind):
        lib = self._lib
        if kind in _FILE_TYPES:
            return _g(_FILE_TYPES[kind]), False
        t = lib.H5Tcopy(_g('H5T_C_S1_g'))
        if kind == 'S1':
            lib.H5Tset_size(t, 1)
        elif kind == 'str':
            lib.H5Tset_size(t, ctypes.c_size_t(-1).value)          # H5T_VARIABLE
            lib.H5Tset_cset(t, 1)                                   # UTF-8
        else:
            raise ValueError(kind)
        return t, True

    def _kind_of(self, path):
        info = self._vars.get(path)
        if info is not None:
            return info[0]
        lib = self._lib
        did = self._dataset(path)
        tid = lib.H5Dget_type(did)
        try:
            cls, size = lib.H5Tget_class(tid), lib.H5Tget_size(tid)
            if cls == 3:
                return 'str' if lib.H5Tis_variable_str(tid) > 0 else 'S1'
            if cls == 1:
                return 'f8' if size == 8 else 'f4'
            return {1: 'i1', 4: 'i4', 8: 'i8'}[size]
        finally:
            lib.H5Tclose(tid)

    def write(self, path, data, record=None):
        """record None: the whole variable (fixed shape); record = i: the i-th slab along the first (unlimited) dimension,
        extending the variable when needed."""
        lib = self._lib
        did = self._dataset(path)
        kind = self._kind_of(path)
        cur = list(self.shape(path))
        if record is not None:
            if record >= cur[0]:
                cur[0] = record + 1
                _ck(lib.H5Dset_extent(did, (_hsize * len(cur))(*cur)), 'H5Dset_extent(%s)' % path)
            start = [int(record)] + [0] * (len(cur) - 1)
            count = [1] + cur[1:]
        else:
            start, count = [0] * len(cur), cur
        fsp = lib.H5Dget_space(did)
        n = len(cur)
        _ck(lib.H5Sselect_hyperslab(fsp, 0, (_hsize * n)(*start), None, (_hsize * n)(*count), None), 'H5Sselect_hyperslab(%s)' % path)
        msp = lib.H5Screate_simple(n, (_hsize * n)(*count), None)
        try:
            if kind == 'str':
                items = [data] if isinstance(data, str) else list(np.asarray(data, dtype=object).reshape(-1))
                if len(items) != int(np.prod(count)):
                    raise ValueError('%s: %d strings for a slab of %s' % (path, len(items), count))
                enc = [str(s).encode('utf-8') for s in items]
                buf = (ctypes.c_char_p * len(enc))(*enc)
                mt, own = self._file_type('str')
                rc = lib.H5Dwrite(did, mt, msp, fsp, 0, buf)
                lib.H5Tclose(mt)
            elif kind == 'S1':
                raw = data.encode('ascii', 'replace') if isinstance(data, str) else bytes(data)
                arr = np.frombuffer(raw.ljust(int(np.prod(count)), b'\x00'), dtype='S1')
                if arr.size != int(np.prod(count)):
                    raise ValueError('%s: %d characters for a slab of %s' % (path, arr.size, count))
                mt, own = self._file_type('S1')
                rc = lib.H5Dwrite(did, mt, msp, fsp, 0, arr.ctypes.data_as(ctypes.c_void_p))
                lib.H5Tclose(mt)
            else:
                name, dt = _MEM_TYPES[kind]
                arr = np.ascontiguousarray(data, dtype=dt)
                if arr.size != int(np.prod(count)):
                    raise ValueError('%s: array of %d values for a slab of %s' % (path, arr.size, count))
                rc = lib.H5Dwrite(did, _g(name), msp, fsp, 0, arr.ctypes.data_as(ctypes.c_void_p))
            if rc < 0:
                raise IOError('write to %s failed' % path)
        finally:
            lib.H5Sclose(msp)
            lib.H5Sclose(fsp)

    # ---- attributes --------------------------------------------------------------------------------------------
    def _set_numeric_attr(self, oid, name, value, scalar=False):
        lib = self._lib
        arr = np.ascontiguousarray(value)
        kind = {np.dtype('int8'): 'i1', np.dtype('int32'): 'i4', np.dtype('int64'): 'i8', np.dtype('float64'): 'f8', np.dtype('float32'): 'f4'}[arr.dtype]
        if lib.H5Aexists(oid, name.encode()) > 0:
            lib.H5Adelete(oid, name.encode())
        sid = lib.H5Screate(0) if scalar else lib.H5Screate_simple(1, (_hsize * 1)(max(1, arr.size)), None)
        aid = _ck(lib.H5Acreate2(oid, name.encode(), _g(_FILE_TYPES[kind]), sid, 0, 0), 'H5Acreate2(%s)' % name)
        _ck(lib.H5Awrite(aid, _g(_MEM_TYPES[kind][0]), arr.ctypes.data_as(ctypes.c_void_p)), 'H5Awrite(%s)' % name)
        lib.H5Aclose(aid)
        lib.H5Sclose(sid)

    def set_attr(self, path, name, value):
        """Text (fixed-length ASCII, as netCDF writes NC_CHAR attributes) or a numeric scalar / 1-D array."""
        lib = self._lib
        oid = lib.H5Oopen(self._fid, (path or '/').encode(), 0)
        if oid < 0:
            raise KeyError(path)
        try:
            if isinstance(value, str):
                raw = value.encode('utf-8') or b' '
                if lib.H5Aexists(oid, name.encode()) > 0:
                    lib.H5Adelete(oid, name.encode())
                t = lib.H5Tcopy(_g('H5T_C_S1_g'))
                lib.H5Tset_size(t, len(raw))
                sid = lib.H5Screate(0)
                aid = _ck(lib.H5Acreate2(oid, name.encode(), t, sid, 0, 0), 'H5Acreate2(%s)' % name)
                _ck(lib.H5Awrite(aid, t, ctypes.c_char_p(raw)), 'H5Awrite(%s)' % name)
                lib.H5Aclose(aid)
                lib.H5Sclose(sid)
                lib.H5Tclose(t)
            else:
                arr = np.asarray(value)
                if arr.dtype == np.int8:
                    pass                                    # (a _FillValue carries its variable's type)
                elif arr.dtype.kind == 'i':
                    arr = arr.astype(np.int64)
                elif arr.dtype.kind == 'f':
                    arr = arr.astype(np.float64)
                self._set_numeric_attr(oid, name, arr.reshape(-1))
        finally:
            lib.H5Oclose(oid)
