"""Test infrastructure: writes small HDF5 files laid out like NetCDF-4 files (one dataset per variable, chunked +
shuffle + deflate like `createVariable(zlib=True)`, an unlimited leading dimension, dimension datasets carrying the
CLASS attribute, string and numeric attributes) through the system's libhdf5 -- inputs for the readers of
sitrack_amd/h5lite.py, since neither netCDF4 nor h5py exists here to produce them.  Not a NetCDF-4 writer: no dimension
scales are attached (the readers do not need them)."""
import ctypes as C
import os

import numpy as np

from sitrack_amd import h5lite

hid_t, hsize_t = h5lite.hid_t, h5lite.hsize_t
UNLIMITED = 2 ** 64 - 1


def _lib():
    L = h5lite._load()
    if getattr(L, "_w_ready", False):
        return L
    sig = {
        "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
        "H5Pcreate": (hid_t, [hid_t]), "H5Pclose": (C.c_int, [hid_t]),
        "H5Pset_chunk": (C.c_int, [hid_t, C.c_int, C.POINTER(hsize_t)]),
        "H5Pset_deflate": (C.c_int, [hid_t, C.c_uint]), "H5Pset_shuffle": (C.c_int, [hid_t]),
        "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        "H5Dwrite": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Screate": (hid_t, [C.c_int]),
        "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]),
        "H5Awrite": (C.c_int, [hid_t, hid_t, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    L._dcpl_class = hid_t.in_dll(L, "H5P_CLS_DATASET_CREATE_ID_g").value
    L._w_ready = True
    return L


def _attr(L, obj, name, val):
    if isinstance(val, str):
        b = val.encode() + b"\0"
        t = L.H5Tcopy(L._c_s1)
        L.H5Tset_size(t, len(b))
        sp = L.H5Screate(0)                                          # H5S_SCALAR
        a = L.H5Acreate2(obj, name.encode(), t, sp, 0, 0)
        assert a >= 0 and L.H5Awrite(a, t, C.create_string_buffer(b)) >= 0
        L.H5Aclose(a); L.H5Sclose(sp); L.H5Tclose(t)
    else:
        v = np.atleast_1d(np.asarray(val))
        key = v.dtype.kind + str(v.dtype.itemsize)
        sp = L.H5Screate(0)
        a = L.H5Acreate2(obj, name.encode(), L._native[key], sp, 0, 0)
        assert a >= 0 and L.H5Awrite(a, L._native[key], v.ctypes.data_as(C.c_void_p)) >= 0
        L.H5Aclose(a); L.H5Sclose(sp)


def write_h5(fname, variables, dims=(), deflate=4):
    """variables: {name: (numpy array, {attr: value}, unlimited_first_dim?)}; dims: names of dimension-only datasets
    or of coordinate variables (they get CLASS = DIMENSION_SCALE)."""
    L = _lib()
    fid = L.H5Fcreate(os.fsencode(fname), 2, 0, 0)                   # H5F_ACC_TRUNC
    assert fid >= 0
    for name, (arr, attrs, unlimited) in variables.items():
        arr = np.ascontiguousarray(arr)
        key = arr.dtype.newbyteorder('=').kind + str(arr.dtype.itemsize)
        arr = arr.astype(np.dtype(key))
        nd = arr.ndim
        dims_c = (hsize_t * max(nd, 1))(*arr.shape)
        maxd = (hsize_t * max(nd, 1))(*([UNLIMITED] + list(arr.shape[1:]) if unlimited else arr.shape))
        sp = L.H5Screate_simple(nd, dims_c, maxd) if nd else L.H5Screate(0)
        dcpl = L.H5Pcreate(L._dcpl_class)
        if nd and (unlimited or arr.size > 64):
            chunk = (hsize_t * nd)(*([1] + list(arr.shape[1:]) if nd > 1 else [max(1, min(arr.shape[0], 4096))]))
            L.H5Pset_chunk(dcpl, nd, chunk)
            L.H5Pset_shuffle(dcpl)
            L.H5Pset_deflate(dcpl, deflate)
        d = L.H5Dcreate2(fid, name.encode(), L._native[key], sp, 0, dcpl, 0)
        assert d >= 0, name
        assert L.H5Dwrite(d, L._native[key], 0, 0, 0, arr.ctypes.data_as(C.c_void_p)) >= 0
        for k, v in (attrs or {}).items():
            _attr(L, d, k, v)
        if name in dims:
            _attr(L, d, "CLASS", "DIMENSION_SCALE")
            _attr(L, d, "NAME", name)
        L.H5Pclose(dcpl); L.H5Sclose(sp); L.H5Dclose(d)
    for name, n in (dims.items() if isinstance(dims, dict) else ()):
        if name in variables:
            continue
        z = np.zeros(n, dtype=np.float32)                            # a dimension without coordinate variable
        sp = L.H5Screate_simple(1, (hsize_t * 1)(n), None)
        d = L.H5Dcreate2(fid, name.encode(), L._native["f4"], sp, 0, 0, 0)
        L.H5Dwrite(d, L._native["f4"], 0, 0, 0, z.ctypes.data_as(C.c_void_p))
        _attr(L, d, "CLASS", "DIMENSION_SCALE")
        _attr(L, d, "NAME", "This is a netCDF dimension but not a netCDF variable.%10d" % n)
        L.H5Sclose(sp); L.H5Dclose(d)
    L.H5Fclose(fid)
