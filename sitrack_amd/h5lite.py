"""Read-only access to NetCDF-4 (= HDF5) files through the system's libhdf5, for installations without the `netCDF4`
Python package (this image has none; it has HDF5 1.10 under /opt/conda/lib).

NEMO's mesh_mask and SI3 output files and the reference's seeding files (`sitrack/ncio.py`, netCDF4.Dataset) are HDF5
files whose variables are plain datasets and whose dimensions are datasets of the same name; reading them needs ~20
calls of the HDF5 C API, bound here with ctypes.  Only what `ncio._Reader` asks for: existence, shapes, numeric
datasets (whole, or a leading-index / slice selection read as a hyperslab so that one record of a multi-GB file is one
read), numeric and string attributes.  Nothing is written: outputs go through netCDF4 when present, NetCDF-3 otherwise.
"""
import ctypes as C
import ctypes.util
import os

import numpy as np

hid_t = C.c_int64
hsize_t = C.c_uint64
_H5 = None
HDF5_MAGIC = b"\x89HDF\r\n\x1a\n"


class H5Unavailable(RuntimeError):
    pass


def is_hdf5(path):
    with open(path, "rb") as f:
        return f.read(8) == HDF5_MAGIC


def _load():
    global _H5
    if _H5 is not None:
        return _H5
    cands = [os.environ.get("SITRK_LIBHDF5"), ctypes.util.find_library("hdf5"), "/opt/conda/lib/libhdf5.so",
             "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so", "/usr/lib/x86_64-linux-gnu/libhdf5.so", "libhdf5.so"]
    err = None
    for c in cands:
        if not c:
            continue
        try:
            L = C.CDLL(c)
            maj, mnr, rel = C.c_uint(), C.c_uint(), C.c_uint()
            if L.H5open() < 0 or L.H5get_libversion(C.byref(maj), C.byref(mnr), C.byref(rel)) < 0:
                raise OSError("H5open failed")
            if (maj.value, mnr.value) < (1, 10):
                raise OSError("HDF5 %d.%d: 64-bit identifiers (>= 1.10) needed" % (maj.value, mnr.value))
            break
        except OSError as e:            # not there / not loadable / too old: next candidate
            err = e
            L = None
    if L is None:
        raise H5Unavailable("no usable libhdf5 (set SITRK_LIBHDF5=/path/to/libhdf5.so, or install netCDF4): %s" % err)
    sig = {
        "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]), "H5Fclose": (C.c_int, [hid_t]),
        "H5Lexists": (C.c_int, [hid_t, C.c_char_p, hid_t]),
        "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Dclose": (C.c_int, [hid_t]),
        "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]),
        "H5Dread": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Sget_simple_extent_npoints": (C.c_int64, [hid_t]),
        "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Sselect_hyperslab": (C.c_int, [hid_t, C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t), C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Sclose": (C.c_int, [hid_t]),
        "H5Tget_class": (C.c_int, [hid_t]), "H5Tget_size": (C.c_size_t, [hid_t]), "H5Tget_sign": (C.c_int, [hid_t]),
        "H5Tis_variable_str": (C.c_int, [hid_t]), "H5Tclose": (C.c_int, [hid_t]), "H5Tcopy": (hid_t, [hid_t]),
        "H5Tset_size": (C.c_int, [hid_t, C.c_size_t]),
        "H5Aexists": (C.c_int, [hid_t, C.c_char_p]), "H5Aopen": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Aget_type": (hid_t, [hid_t]), "H5Aget_space": (hid_t, [hid_t]), "H5Aread": (C.c_int, [hid_t, hid_t, C.c_void_p]),
        "H5Aclose": (C.c_int, [hid_t]), "H5free_memory": (C.c_int, [C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    try:                                     # keep the library quiet: failures are reported through return codes
        L.H5Eset_auto2.restype, L.H5Eset_auto2.argtypes = C.c_int, [hid_t, C.c_void_p, C.c_void_p]
        L.H5Eset_auto2(0, None, None)
    except AttributeError:
        pass
    L._native = {k: hid_t.in_dll(L, "H5T_NATIVE_%s_g" % v).value for k, v in
                 (("i1", "SCHAR"), ("u1", "UCHAR"), ("i2", "SHORT"), ("u2", "USHORT"), ("i4", "INT"), ("u4", "UINT"),
                  ("i8", "LLONG"), ("u8", "ULLONG"), ("f4", "FLOAT"), ("f8", "DOUBLE"))}
    L._c_s1 = hid_t.in_dll(L, "H5T_C_S1_g").value
    _H5 = L
    return L


def available():
    try:
        _load()
        return True
    except H5Unavailable:
        return False


class H5File:
    def __init__(self, path):
        self.L = _load()
        self.path = path
        self.fid = self.L.H5Fopen(os.fsencode(path), 0, 0)             # H5F_ACC_RDONLY, H5P_DEFAULT
        if self.fid < 0:
            raise OSError("H5Fopen(%s) failed" % path)

    def close(self):
        if self.fid >= 0:
            self.L.H5Fclose(self.fid)
            self.fid = -1

    def __del__(self):
        try:
            self.close()
        except Exception:               # noqa: BLE001
            pass

    def has(self, name):
        return self.L.H5Lexists(self.fid, name.encode(), 0) > 0

    def _open(self, name):
        if not self.has(name):
            raise KeyError(name)
        d = self.L.H5Dopen2(self.fid, name.encode(), 0)
        if d < 0:
            raise KeyError(name)
        return d

    def _npkind(self, tid, what):
        cls, size = self.L.H5Tget_class(tid), self.L.H5Tget_size(tid)
        if cls == 0:                                                   # H5T_INTEGER
            key = ("i" if self.L.H5Tget_sign(tid) == 1 else "u") + str(size)
        elif cls == 1:                                                 # H5T_FLOAT
            key = "f" + str(size)
        else:
            raise TypeError("%s: HDF5 type class %d is not numeric" % (what, cls))
        if key not in self.L._native:
            raise TypeError("%s: unsupported numeric type %s" % (what, key))
        return key

    def dtype(self, name):
        """numpy dtype the dataset `name` is stored with"""
        d = self._open(name)
        try:
            tid = self.L.H5Dget_type(d)
            key = self._npkind(tid, name)
            self.L.H5Tclose(tid)
            return np.dtype(key)
        finally:
            self.L.H5Dclose(d)

    def shape(self, name):
        d = self._open(name)
        try:
            sp = self.L.H5Dget_space(d)
            nd = self.L.H5Sget_simple_extent_ndims(sp)
            dims = (hsize_t * max(nd, 1))()
            if nd > 0:
                self.L.H5Sget_simple_extent_dims(sp, dims, None)
            self.L.H5Sclose(sp)
            return tuple(int(dims[k]) for k in range(nd))
        finally:
            self.L.H5Dclose(d)

    def read(self, name, index=Ellipsis, out=None):
        """numpy array of dataset `name`.  `index`: Ellipsis, an int, a slice (step 1) or a tuple of those for the
        leading dimensions -> one hyperslab read; anything else is applied with numpy after reading everything.
        `out` (optional): a C-contiguous numeric array of the selection's shape to read INTO (HDF5 converts to its
        dtype) -- e.g. pinned staging memory of libsitrk, so that a record goes from the file to DMA-able memory in one
        pass; only for hyperslab selections."""
        shp = self.shape(name)
        idx = index if isinstance(index, tuple) else (index,)
        simple = all(isinstance(k, (int, np.integer)) or k is Ellipsis or (isinstance(k, slice) and k.step in (None, 1)) for k in idx)
        if not simple or sum(k is Ellipsis for k in idx) > 1 or (Ellipsis in idx and idx[-1] is not Ellipsis):
            if out is not None:
                out[...] = self.read(name)[index]
                return out
            return self.read(name)[index]
        idx = tuple(k for k in idx if k is not Ellipsis)
        if len(idx) > len(shp):
            raise IndexError("too many indices for %s%s" % (name, shp))
        start, count, keep = [], [], []
        for ax, n in enumerate(shp):
            if ax < len(idx) and not isinstance(idx[ax], slice):
                k = int(idx[ax])
                k = k + n if k < 0 else k
                if not 0 <= k < n:
                    raise IndexError("index %d out of range for axis %d of %s%s" % (int(idx[ax]), ax, name, shp))
                start.append(k); count.append(1)
            else:
                lo, hi, _ = (idx[ax] if ax < len(idx) else slice(None)).indices(n)
                start.append(lo); count.append(max(0, hi - lo)); keep.append(len(count) - 1)
        out_shape = tuple(count[k] for k in keep)
        d = self._open(name)
        try:
            tid = self.L.H5Dget_type(d)
            key = self._npkind(tid, name)
            self.L.H5Tclose(tid)
            if out is None:
                out = np.empty(out_shape, dtype=np.dtype(key))
            else:
                if tuple(out.shape) != out_shape or not out.flags.c_contiguous:
                    raise ValueError("%s: `out` must be C-contiguous of shape %s, got %s" % (name, out_shape, tuple(out.shape)))
                key = out.dtype.kind + str(out.dtype.itemsize)           # memory type: HDF5 converts on the fly
                if key not in self.L._native or not out.dtype.isnative:
                    raise TypeError("%s: cannot read into dtype %s" % (name, out.dtype))
            if out.size == 0:
                return out
            nd = len(shp)
            if nd == 0:
                rc = self.L.H5Dread(d, self.L._native[key], 0, 0, 0, out.ctypes.data_as(C.c_void_p))
            else:
                fsp = self.L.H5Dget_space(d)
                st, ct = (hsize_t * nd)(*start), (hsize_t * nd)(*count)
                self.L.H5Sselect_hyperslab(fsp, 0, st, None, ct, None)        # H5S_SELECT_SET
                msp = self.L.H5Screate_simple(nd, ct, None)
                rc = self.L.H5Dread(d, self.L._native[key], msp, fsp, 0, out.ctypes.data_as(C.c_void_p))
                self.L.H5Sclose(msp); self.L.H5Sclose(fsp)
            if rc < 0:
                raise OSError("H5Dread(%s) failed" % name)
            return out
        finally:
            self.L.H5Dclose(d)

    def has_attr(self, name, att):
        d = self._open(name)
        try:
            return self.L.H5Aexists(d, att.encode()) > 0
        finally:
            self.L.H5Dclose(d)

    def attr(self, name, att):
        d = self._open(name)
        try:
            if self.L.H5Aexists(d, att.encode()) <= 0:
                raise AttributeError("%s has no attribute %s" % (name, att))
            a = self.L.H5Aopen(d, att.encode(), 0)
            tid = self.L.H5Aget_type(a)
            sp = self.L.H5Aget_space(a)
            n = max(1, int(self.L.H5Sget_simple_extent_npoints(sp)))
            self.L.H5Sclose(sp)
            try:
                if self.L.H5Tget_class(tid) == 3:                      # H5T_STRING
                    if self.L.H5Tis_variable_str(tid) > 0:
                        ptrs = (C.c_void_p * n)()
                        mt = self.L.H5Tcopy(self.L._c_s1)
                        self.L.H5Tset_size(mt, C.c_size_t(-1).value)    # H5T_VARIABLE
                        rc = self.L.H5Aread(a, mt, ptrs)
                        self.L.H5Tclose(mt)
                        if rc < 0:
                            raise OSError("H5Aread(%s.%s) failed" % (name, att))
                        vals = [C.string_at(p).decode() if p else "" for p in ptrs]
                        for p in ptrs:
                            if p:
                                self.L.H5free_memory(p)
                    else:
                        sz = self.L.H5Tget_size(tid)
                        buf = C.create_string_buffer(sz * n + 1)
                        if self.L.H5Aread(a, tid, buf) < 0:
                            raise OSError("H5Aread(%s.%s) failed" % (name, att))
                        vals = [buf.raw[k * sz:(k + 1) * sz].split(b"\x00")[0].decode() for k in range(n)]
                    return vals[0] if n == 1 else vals
                key = self._npkind(tid, name + "." + att)
                out = np.empty(n, dtype=np.dtype(key))
                if self.L.H5Aread(a, self.L._native[key], out.ctypes.data_as(C.c_void_p)) < 0:
                    raise OSError("H5Aread(%s.%s) failed" % (name, att))
                return out[0] if n == 1 else out
            finally:
                self.L.H5Tclose(tid)
                self.L.H5Aclose(a)
        finally:
            self.L.H5Dclose(d)
