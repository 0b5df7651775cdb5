# -*- coding: utf-8 -*-
'''
The few calls of the HDF5 C library that the heavy data of an XDMF time series
needs (flow_amd/fem/io.py: XDMFFile -- what dolfin's XDMFFile writes behind
tests/test_karman_vortex_street.py:214-227 of the reference), bound with
ctypes: there is no h5py in the image, libhdf5 itself is.  Datasets of
contiguous float64 / int64 / int32 arrays under paths whose groups are created
on the way; reading them back for the tests.

The library is looked for under FLOW_AMD_HDF5_LIB, the loader's search path and
the usual prefixes; `available()` says whether one was found -- XDMFFile then
falls back to inline XML data.
'''
import ctypes
import ctypes.util
import glob
import os

import numpy

_H5F_ACC_RDONLY, _H5F_ACC_RDWR, _H5F_ACC_TRUNC = 0, 1, 2
_H5P_DEFAULT, _H5S_ALL = 0, 0
_hid = ctypes.c_int64          # hid_t of HDF5 >= 1.10

_LIB = [None, False]           # (library or None, looked for it)


def _candidates():
    env = os.environ.get('FLOW_AMD_HDF5_LIB')
    if env:
        yield env
    found = ctypes.util.find_library('hdf5')
    if found:
        yield found
    for pat in ('/usr/lib/x86_64-linux-gnu/libhdf5_serial.so*',
                '/usr/lib/x86_64-linux-gnu/libhdf5.so*',
                '/usr/lib64/libhdf5.so*', '/usr/local/lib/libhdf5.so*',
                '/opt/conda/lib/libhdf5.so*'):
        for path in sorted(glob.glob(pat)):
            yield path


def _lib():
    if _LIB[1]:
        return _LIB[0]
    _LIB[1] = True
    for path in _candidates():
        try:
            lib = ctypes.CDLL(path)
            major, minor, rel = (ctypes.c_uint(), ctypes.c_uint(),
                                 ctypes.c_uint())
            lib.H5open()
            lib.H5get_libversion(ctypes.byref(major), ctypes.byref(minor),
                                 ctypes.byref(rel))
            if (major.value, minor.value) < (1, 10):
                continue            # (hid_t was 32 bits wide before)
        except (OSError, AttributeError):
            continue
        for name, res, args in (
                ('H5Fcreate', _hid, [ctypes.c_char_p, ctypes.c_uint, _hid, _hid]),
                ('H5Fopen', _hid, [ctypes.c_char_p, ctypes.c_uint, _hid]),
                ('H5Fflush', ctypes.c_int, [_hid, ctypes.c_int]),
                ('H5Fclose', ctypes.c_int, [_hid]),
                ('H5Screate_simple', _hid,
                 [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
                ('H5Sclose', ctypes.c_int, [_hid]),
                ('H5Pcreate', _hid, [_hid]),
                ('H5Pset_create_intermediate_group', ctypes.c_int,
                 [_hid, ctypes.c_uint]),
                ('H5Pclose', ctypes.c_int, [_hid]),
                ('H5Dcreate2', _hid,
                 [_hid, ctypes.c_char_p, _hid, _hid, _hid, _hid, _hid]),
                ('H5Dopen2', _hid, [_hid, ctypes.c_char_p, _hid]),
                ('H5Dwrite', ctypes.c_int,
                 [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]),
                ('H5Dread', ctypes.c_int,
                 [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]),
                ('H5Dget_space', _hid, [_hid]),
                ('H5Dget_type', _hid, [_hid]),
                ('H5Tget_class', ctypes.c_int, [_hid]),
                ('H5Tget_size', ctypes.c_size_t, [_hid]),
                ('H5Tclose', ctypes.c_int, [_hid]),
                ('H5Sget_simple_extent_ndims', ctypes.c_int, [_hid]),
                ('H5Sget_simple_extent_dims', ctypes.c_int,
                 [_hid, ctypes.c_void_p, ctypes.c_void_p]),
                ('H5Dclose', ctypes.c_int, [_hid]),
                ('H5Lexists', ctypes.c_int, [_hid, ctypes.c_char_p, _hid]),
                ):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _LIB[0] = lib
        break
    return _LIB[0]


def available():
    return _lib() is not None


def _global(name):
    return _hid.in_dll(_lib(), name).value


def _native(dtype):
    dtype = numpy.dtype(dtype)
    names = {numpy.dtype('float64'): 'H5T_NATIVE_DOUBLE_g',
             numpy.dtype('int64'): 'H5T_NATIVE_INT64_g',
             numpy.dtype('int32'): 'H5T_NATIVE_INT32_g'}
    if dtype not in names:
        raise TypeError('HDF5 datasets of float64 / int64 / int32, not %s'
                        % dtype)
    return _global(names[dtype])


def _check(rc, what):
    if rc < 0:
        raise IOError('HDF5: %s failed' % what)
    return rc


class File(object):
    '''`with File(path, 'w') as f: f.write('/Mesh/0/mesh/geometry', x)`; mode
    'r' for `read`, 'a' to add datasets to an existing file.'''

    def __init__(self, path, mode='r'):
        lib = _lib()
        if lib is None:
            raise IOError('no HDF5 library found (FLOW_AMD_HDF5_LIB)')
        self._lib = lib
        bpath = os.fsencode(path)
        if mode == 'w':
            self._id = lib.H5Fcreate(bpath, _H5F_ACC_TRUNC, _H5P_DEFAULT,
                                     _H5P_DEFAULT)
        elif mode == 'r':
            self._id = lib.H5Fopen(bpath, _H5F_ACC_RDONLY, _H5P_DEFAULT)
        elif mode == 'a':
            self._id = lib.H5Fopen(bpath, _H5F_ACC_RDWR, _H5P_DEFAULT)
        else:
            raise ValueError(mode)
        _check(self._id, 'opening %s' % path)
        self._lcpl = None
        if mode in ('w', 'a'):
            self._lcpl = _check(
                lib.H5Pcreate(_global('H5P_CLS_LINK_CREATE_ID_g')), 'H5Pcreate')
            _check(lib.H5Pset_create_intermediate_group(self._lcpl, 1),
                   'H5Pset_create_intermediate_group')

    def __enter__(self):
        return self

    def __exit__(self, tpe, value, traceback):
        self.close()
        return False

    def write(self, name, array):
        '''A contiguous dataset `name` (absolute path; missing groups are
        created) holding `array`.'''
        lib = self._lib
        a = numpy.ascontiguousarray(array)
        tid = _native(a.dtype)
        dims = (ctypes.c_uint64 * a.ndim)(*a.shape)
        space = _check(lib.H5Screate_simple(a.ndim, dims, None),
                       'H5Screate_simple')
        dset = lib.H5Dcreate2(self._id, name.encode(), tid, space, self._lcpl,
                              _H5P_DEFAULT, _H5P_DEFAULT)
        try:
            _check(dset, 'creating dataset %s' % name)
            _check(lib.H5Dwrite(dset, tid, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT,
                                a.ctypes.data_as(ctypes.c_void_p)),
                   'writing %s' % name)
        finally:
            if dset >= 0:
                lib.H5Dclose(dset)
            lib.H5Sclose(space)

    def exists(self, name):
        '''Every link on the path exists.'''
        parts = [p for p in name.split('/') if p]
        for k in range(1, len(parts) + 1):
            if self._lib.H5Lexists(
                    self._id, ('/' + '/'.join(parts[:k])).encode(),
                    _H5P_DEFAULT) <= 0:
                return False
        return True

    def read(self, name):
        lib = self._lib
        dset = _check(lib.H5Dopen2(self._id, name.encode(), _H5P_DEFAULT),
                      'opening dataset %s' % name)
        try:
            space = _check(lib.H5Dget_space(dset), 'H5Dget_space')
            nd = _check(lib.H5Sget_simple_extent_ndims(space), 'rank')
            dims = (ctypes.c_uint64 * max(nd, 1))()
            lib.H5Sget_simple_extent_dims(space, dims, None)
            lib.H5Sclose(space)
            ftype = _check(lib.H5Dget_type(dset), 'H5Dget_type')
            cls, size = lib.H5Tget_class(ftype), lib.H5Tget_size(ftype)
            lib.H5Tclose(ftype)
            # H5T_INTEGER = 0, H5T_FLOAT = 1
            dtype = {(1, 8): 'float64', (0, 8): 'int64', (0, 4): 'int32'}.get(
                (cls, size))
            if dtype is None:
                raise TypeError('dataset %s: class %d, %d bytes' % (name, cls, size))
            out = numpy.empty([int(d) for d in dims[:nd]], dtype=dtype)
            _check(lib.H5Dread(dset, _native(dtype), _H5S_ALL, _H5S_ALL,
                               _H5P_DEFAULT, out.ctypes.data_as(ctypes.c_void_p)),
                   'reading %s' % name)
            return out
        finally:
            lib.H5Dclose(dset)

    def flush(self):
        _check(self._lib.H5Fflush(self._id, 1), 'H5Fflush')     # H5F_SCOPE_GLOBAL

    def close(self):
        if self._id is None:
            return
        if self._lcpl is not None:
            self._lib.H5Pclose(self._lcpl)
        self._lib.H5Fclose(self._id)
        self._id = None
