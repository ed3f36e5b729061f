"""NetCDF-4 (= HDF5) files through the system's libhdf5, for installations without the `netCDF4` Python package (this
image has none; it has HDF5 1.10 under /opt/conda/lib): a reader for NEMO / seeding inputs and a writer (`NC4Writer`)
for the trajectory / seeding files of reference `sitrack/ncio.py:131-197`.

NEMO's mesh_mask and SI3 output files and the reference's seeding files (`sitrack/ncio.py`, netCDF4.Dataset) are HDF5
files whose variables are plain datasets and whose dimensions are datasets of the same name; reading them needs ~20
calls of the HDF5 C API, bound here with ctypes.  Only what `ncio._Reader` asks for: existence, shapes, numeric
datasets (whole, or a leading-index / slice selection read as a hyperslab so that one record of a multi-GB file is one
read), numeric and string attributes.  Writing: `NC4Writer` lays a file out the way netCDF-C does -- every dimension a dimension-scale dataset (H5DS, from
libhdf5_hl), every variable a chunked dataset attached to its scales, unlimited leading dimension, shuffle + deflate,
`_FillValue` as a one-element attribute plus the dataset's own fill value, fixed-length string attributes -- so that
`netCDF4` / `ncdump` read it as a NetCDF-4 file (`h5dump -H` of an output equals the header of the reference's own
`tools/nc/...HSS5.nc__KEEP`, tests/test_driver.py).
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


# --------------------------------------------------------------------------- writer
_HL = None
UNLIMITED = 2 ** 64 - 1


def _load_hl():
    """libhdf5_hl (dimension scales), from the directory libhdf5 itself was loaded from"""
    global _HL
    if _HL is not None:
        return _HL
    L = _load()
    cands = [os.environ.get("SITRK_LIBHDF5_HL")]
    base = getattr(L, "_name", "") or ""
    if base and os.path.sep in base:
        cands.append(os.path.join(os.path.dirname(base), "libhdf5_hl.so"))
    cands += [ctypes.util.find_library("hdf5_hl"), "/opt/conda/lib/libhdf5_hl.so", "/usr/lib/x86_64-linux-gnu/libhdf5_serial_hl.so",
              "libhdf5_hl.so"]
    err = None
    for c in cands:
        if not c:
            continue
        try:
            H = C.CDLL(c)
            H.H5DSset_scale.restype, H.H5DSset_scale.argtypes = C.c_int, [hid_t, C.c_char_p]
            H.H5DSattach_scale.restype, H.H5DSattach_scale.argtypes = C.c_int, [hid_t, hid_t, C.c_uint]
            _HL = H
            return H
        except (OSError, AttributeError) as e:
            err = e
    raise H5Unavailable("no usable libhdf5_hl (dimension scales): %s" % err)


def writer_available():
    try:
        _load()
        _load_hl()
        return True
    except H5Unavailable:
        return False


def _wlib():
    L = _load()
    if getattr(L, "_w_ready", False):
        return L
    sig = {
        "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
        "H5Pcreate": (hid_t, [hid_t]), "H5Pclose": (C.c_int, [hid_t]),
        "H5Pset_chunk": (C.c_int, [hid_t, C.c_int, C.POINTER(hsize_t)]),
        "H5Pset_deflate": (C.c_int, [hid_t, C.c_uint]), "H5Pset_shuffle": (C.c_int, [hid_t]),
        "H5Pset_fill_value": (C.c_int, [hid_t, hid_t, C.c_void_p]),
        "H5Pset_attr_creation_order": (C.c_int, [hid_t, C.c_uint]),
        "H5Pset_link_creation_order": (C.c_int, [hid_t, C.c_uint]),
        "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        "H5Dwrite": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Screate": (hid_t, [C.c_int]),
        "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]),
        "H5Awrite": (C.c_int, [hid_t, hid_t, C.c_void_p]),
        "H5Gopen2": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Gclose": (C.c_int, [hid_t]),
    }
    try:                                 # direct chunk write (HDF5 >= 1.10.3): chunks deflated by our own threads
        L.H5Dwrite_chunk.restype = C.c_int
        L.H5Dwrite_chunk.argtypes = [hid_t, hid_t, C.c_uint32, C.POINTER(hsize_t), C.c_size_t, C.c_void_p]
        L._has_chunk_write = True
    except AttributeError:
        L._has_chunk_write = False
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    L._dcpl_class = hid_t.in_dll(L, "H5P_CLS_DATASET_CREATE_ID_g").value
    L._fcpl_class = hid_t.in_dll(L, "H5P_CLS_FILE_CREATE_ID_g").value
    L._std = {k: hid_t.in_dll(L, "H5T_%s_g" % v).value for k, v in
              (("i1", "STD_I8LE"), ("u1", "STD_U8LE"), ("i2", "STD_I16LE"), ("i4", "STD_I32LE"), ("i8", "STD_I64LE"),
               ("f4", "IEEE_F32LE"), ("f8", "IEEE_F64LE"))}
    L._w_ready = True
    return L


class NC4Writer:
    """One NetCDF-4 file: createDimension / createVariable / attributes, in the vocabulary of netCDF4.Dataset (only what
    ncSaveCloudBuoys needs).  Variables are written whole with write()."""

    def __init__(self, path):
        self.L, self.H = _wlib(), _load_hl()
        L = self.L
        # netCDF-C tracks creation order of links and attributes (ncdump lists variables in that order)
        fcpl = L.H5Pcreate(L._fcpl_class)
        L.H5Pset_link_creation_order(fcpl, 3)                     # H5P_CRT_ORDER_TRACKED | INDEXED
        L.H5Pset_attr_creation_order(fcpl, 3)
        self.fid = L.H5Fcreate(os.fsencode(path), 2, fcpl, 0)      # H5F_ACC_TRUNC
        L.H5Pclose(fcpl)
        if self.fid < 0:
            raise OSError("H5Fcreate(%s) failed" % path)
        self.path = path
        self.dims = {}                   # name -> (size, unlimited, dimid)
        self.vars = {}                   # name -> (hid, dtype key, dim names)

    # -- attributes
    def _attr(self, obj, name, val):
        L = self.L
        if isinstance(val, (str, bytes)):
            b = val.encode() if isinstance(val, str) else val
            t = L.H5Tcopy(L._c_s1)
            L.H5Tset_size(t, max(len(b), 1))
            sp = L.H5Screate(0)                                     # H5S_SCALAR
            a = L.H5Acreate2(obj, name.encode(), t, sp, 0, 0)
            buf = C.create_string_buffer(b, max(len(b), 1))
            ok = a >= 0 and L.H5Awrite(a, t, buf) >= 0
            L.H5Aclose(a); L.H5Sclose(sp); L.H5Tclose(t)
        else:
            v = np.atleast_1d(np.asarray(val))
            key = v.dtype.kind + str(v.dtype.itemsize)
            sp = L.H5Screate_simple(1, (hsize_t * 1)(v.size), None) if name != "_Netcdf4Dimid" else L.H5Screate(0)
            a = L.H5Acreate2(obj, name.encode(), L._std[key], sp, 0, 0)
            ok = a >= 0 and L.H5Awrite(a, L._native[key], v.ctypes.data_as(C.c_void_p)) >= 0
            L.H5Aclose(a); L.H5Sclose(sp)
        if not ok:
            raise OSError("writing attribute %s failed" % name)

    def set_global(self, name, val):
        g = self.L.H5Gopen2(self.fid, b"/", 0)
        try:
            self._attr(g, name, val)
        finally:
            self.L.H5Gclose(g)

    def set_attr(self, var, name, val):
        self._attr(self.vars[var][0], name, val)

    # -- dimensions and variables
    def createDimension(self, name, size):
        self.dims[name] = (0 if size is None else int(size), size is None, len(self.dims))

    def createVariable(self, name, dtype, dims, fill_value=None, zlib=False, complevel=4, nrec=0):
        """`nrec`: current length of the unlimited dimension (records about to be written)."""
        L = self.L
        key = np.dtype(dtype).kind + str(np.dtype(dtype).itemsize)
        shape, maxs = [], []
        for d in dims:
            size, unl, _ = self.dims[d]
            shape.append(nrec if unl else size)
            maxs.append(UNLIMITED if unl else size)
        nd = len(dims)
        sp = L.H5Screate_simple(nd, (hsize_t * nd)(*shape), (hsize_t * nd)(*maxs))
        dcpl = L.H5Pcreate(L._dcpl_class)
        L.H5Pset_attr_creation_order(dcpl, 3)
        unlimited = any(self.dims[d][1] for d in dims)
        if unlimited or zlib:
            # netCDF's default layout for (record, n): one record per chunk; long rows are cut (a chunk is one deflate unit)
            chunk = [1 if self.dims[d][1] else max(1, min(self.dims[d][0], 1 << 20)) for d in dims]
            if nd == 1 and unlimited:
                chunk = [1024]
            L.H5Pset_chunk(dcpl, nd, (hsize_t * nd)(*chunk))
            if zlib:
                L.H5Pset_shuffle(dcpl)
                L.H5Pset_deflate(dcpl, int(complevel))
        if fill_value is not None:
            fv = np.asarray([fill_value], dtype=np.dtype(key))
            L.H5Pset_fill_value(dcpl, L._native[key], fv.ctypes.data_as(C.c_void_p))
        d = L.H5Dcreate2(self.fid, name.encode(), L._std[key], sp, 0, dcpl, 0)
        L.H5Pclose(dcpl); L.H5Sclose(sp)
        if d < 0:
            raise OSError("H5Dcreate2(%s) failed" % name)
        self.vars[name] = (d, key, tuple(dims))
        self.layout = getattr(self, "layout", {})
        self.layout[name] = (tuple(chunk), int(complevel), fill_value) if ((unlimited or zlib) and zlib) else None
        if fill_value is not None:
            self._attr(d, "_FillValue", np.asarray([fill_value], dtype=np.dtype(key)))
        return name

    def write(self, name, data):
        d, key, dims = self.vars[name]
        a = np.ascontiguousarray(data, dtype=np.dtype(key))
        lay = self.layout.get(name)
        if lay is not None and a.ndim == 2 and a.size >= (1 << 20) and self.L._has_chunk_write and a.dtype.isnative:
            return self._write_chunks_parallel(d, a, *lay)
        if self.L.H5Dwrite(d, self.L._native[key], 0, 0, 0, a.ctypes.data_as(C.c_void_p)) < 0:
            raise OSError("H5Dwrite(%s) failed" % name)

    def write_rows(self, name, r0, data):
        """Rows [r0, r0 + len(data)) of a (record, n) variable that was created at its final length (createVariable(nrec=...)):
        what a streaming writer appends record by record.  Large compressed rows go through the same thread pool + direct chunk
        writes as write(); small ones through a hyperslab H5Dwrite.  Memory: the rows handed over, until flush()."""
        d, key, dims = self.vars[name]
        a = np.ascontiguousarray(data, dtype=np.dtype(key))
        if a.ndim == len(dims) - 1:
            a = a[None]
        lay = self.layout.get(name)
        if lay is not None and a.ndim == 2 and a.shape[1] >= (1 << 16) and self.L._has_chunk_write and a.dtype.isnative:
            return self._write_chunks_parallel(d, a, *lay, r0=int(r0))
        L = self.L
        nd = a.ndim
        fsp = L.H5Dget_space(d)
        st = (hsize_t * nd)(*([int(r0)] + [0] * (nd - 1)))
        ct = (hsize_t * nd)(*a.shape)
        msp = L.H5Screate_simple(nd, ct, None)
        try:
            if fsp < 0 or msp < 0 or L.H5Sselect_hyperslab(fsp, 0, st, None, ct, None) < 0 or \
                    L.H5Dwrite(d, L._native[key], msp, fsp, 0, a.ctypes.data_as(C.c_void_p)) < 0:
                raise OSError("H5Dwrite(%s, rows %d..) failed" % (name, r0))
        finally:
            if msp >= 0:
                L.H5Sclose(msp)
            if fsp >= 0:
                L.H5Sclose(fsp)

    def _write_chunks_parallel(self, d, a, chunk, level, fill, r0=0):
        """Shuffle + deflate of every (1 x C) chunk on a thread pool (zlib releases the GIL), raw chunks handed to
        H5Dwrite_chunk: the file is what H5Dwrite's filter pipeline would have produced, in a fraction of the time --
        libhdf5 deflates one chunk after the other on the calling thread, 45 s for the 10^7-buoy files at level 9.
        The chunks of ALL variables share one pool and are only queued here (a variable has 2 x 10 chunks at 10^7 buoys:
        on 16 threads its second round would keep 4 of them busy); close() -- or flush() -- hands them to the library."""
        import zlib
        from concurrent.futures import ThreadPoolExecutor
        nrec, n = a.shape
        cw = chunk[1]
        isz = a.dtype.itemsize

        def pack(r, c0):
            seg = a[r, c0:c0 + cw]
            if seg.shape[0] < cw:                                  # an edge chunk is stored whole
                pad = np.full(cw, 0 if fill is None else fill, dtype=a.dtype)
                pad[:seg.shape[0]] = seg
                seg = pad
            shuffled = np.ascontiguousarray(seg.view(np.uint8).reshape(cw, isz).T)      # the SHUFFLE filter: byte planes
            return zlib.compress(shuffled.tobytes(), level)
        if getattr(self, "_pool", None) is None:
            self._pool = ThreadPoolExecutor(min(16, os.cpu_count() or 1))
            self._pending = []
        for r in range(nrec):
            for c0 in range(0, n, cw):
                self._pending.append((d, r0 + r, c0, self._pool.submit(pack, r, c0)))
        # every queued job keeps its variable's array alive: bound what waits (10^8 buoys x 2 records x 6 variables = 4.8 GB)
        self._pending_bytes = getattr(self, "_pending_bytes", 0) + a.nbytes
        if self._pending_bytes > int(os.environ.get("SITRK_NC_PENDING_BYTES", str(4 << 30))):
            self.flush()

    def flush(self):
        """hand the queued chunks to libhdf5 (in the order they were queued; the calls themselves are serial)"""
        pending, self._pending = getattr(self, "_pending", []), []
        self._pending_bytes = 0
        try:
            for d, r, c0, fut in pending:
                blob = fut.result()
                off = (hsize_t * 2)(r, c0)
                if self.L.H5Dwrite_chunk(d, 0, 0, off, len(blob), blob) < 0:
                    raise OSError("H5Dwrite_chunk failed at (%d,%d)" % (r, c0))
        finally:
            for _, _, _, fut in pending:
                fut.cancel()

    def _release(self):
        """close every id this writer holds, whatever state the file is in"""
        L = self.L
        if getattr(self, "_pool", None) is not None:
            for _, _, _, fut in getattr(self, "_pending", []):
                fut.cancel()
            self._pending = []
            self._pool.shutdown(wait=True)
            self._pool = None
        for vid, _, _ in self.vars.values():
            L.H5Dclose(vid)
        self.vars = {}
        if self.fid >= 0:
            L.H5Fclose(self.fid)
        self.fid = -1

    def abort(self):
        """Give up on the file: ids closed, the partial file removed.  For `except` blocks -- never raises, so the
        exception that led here stays the one the caller sees."""
        try:
            self._release()
        except Exception:               # noqa: BLE001
            self.fid = -1
        try:
            os.remove(self.path)
        except OSError:
            pass

    def close(self):
        """Turn the coordinate variables into dimension scales, attach every variable to its scales, close.  If any of
        that fails the ids are closed all the same, the half-written file is removed and the error is raised."""
        if self.fid < 0:
            return
        L, H = self.L, self.H
        try:
            self.flush()
            for dname, (size, unl, dimid) in self.dims.items():
                if dname not in self.vars:
                    raise ValueError("dimension %s needs its coordinate variable" % dname)
                did = self.vars[dname][0]
                if H.H5DSset_scale(did, dname.encode()) < 0:
                    raise OSError("H5DSset_scale(%s) failed" % dname)
                self._attr(did, "_Netcdf4Dimid", np.int32(dimid))
            for vname, (vid, key, dims) in self.vars.items():
                if vname in self.dims:
                    continue
                for k, dname in enumerate(dims):
                    if H.H5DSattach_scale(vid, self.vars[dname][0], k) < 0:
                        raise OSError("H5DSattach_scale(%s, %s) failed" % (vname, dname))
        except BaseException:
            self.abort()
            raise
        self._release()
