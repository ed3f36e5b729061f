"""NetCDF input/output for the driver -- host-side mirror of reference `sitrack/ncio.py`.

Same function names, arguments and return values as the reference, same on-disk schema
(reference ncio.py:131-197 for trajectory/seeding files, :22-92 for the NEMO `mesh_mask`,
si3_part_tracker.py:365-374 for the `icemod` records).  Two differences, both forced by
the environment: (1) the geographic -> polar-stereographic conversion goes through
libsitrk's projection kernel instead of cartopy; (2) the reference's hard dependency on the
`netCDF4` package becomes optional: when it is absent, NetCDF-4 files are read AND written
through the system's libhdf5 (h5lite.py: same variable types, `id_buoy` int64, shuffle +
deflate 9, unlimited `time`, dimension scales -- a file `netCDF4`/`ncdump` read as NetCDF-4),
classic NetCDF-3 inputs are read with `scipy.io.netcdf_file`.  Only where libhdf5 cannot be
loaded either, outputs fall back to NetCDF-3 (no int64: `id_buoy` as float64; no compression).
"""
import os
from os import path

import numpy as np

tunits_default = 'seconds since 1970-01-01 00:00:00'      # reference ncio.py:15
FillValue = -9999.                                         # reference ncio.py:19

try:                                                       # pragma: no cover - depends on the environment
    import netCDF4 as _nc4
except Exception:                                          # noqa: BLE001
    _nc4 = None


def backend():
    """What reads and writes: netCDF4 when importable; otherwise NetCDF-4/HDF5 inputs are read through the system's
    libhdf5 (h5lite.py), classic inputs through scipy, and outputs are written as NetCDF-3."""
    if _nc4 is not None:
        return "netCDF4"
    from . import h5lite
    if h5lite.writer_available():
        return "libhdf5 (NetCDF-4 read + write) + scipy (NetCDF-3 read)"
    return "scipy-netcdf3" + (" + libhdf5 reader" if h5lite.available() else "")


def chck4f(cf):
    if not path.exists(cf):
        raise FileNotFoundError(' ERROR [chck4f()]: file ' + cf + ' does not exist!')


class _Reader:
    """Minimal uniform read access: var(name)[index] -> ndarray, attr(name, att), dim(name)."""

    def __init__(self, cfile):
        chck4f(cfile)
        self.nc4 = None
        self.sp = None
        self.h5 = None
        if _nc4 is not None:
            try:
                self.nc4 = _nc4.Dataset(cfile)
            except Exception:                              # noqa: BLE001  (e.g. classic file and odd build)
                self.nc4 = None
        if self.nc4 is None:
            from . import h5lite
            if h5lite.is_hdf5(cfile):                      # a NetCDF-4 file and no netCDF4 package: the system's libhdf5
                self.h5 = h5lite.H5File(cfile)
            else:
                from scipy.io import netcdf_file
                self.sp = netcdf_file(cfile, 'r', mmap=False, maskandscale=False)

    def has_var(self, name):
        if self.h5 is not None:                            # a dimension without coordinate variable is a dataset too
            return self.h5.has(name) and not (self.h5.has_attr(name, 'NAME') and
                                              str(self.h5.attr(name, 'NAME')).startswith('This is a netCDF dimension but not'))
        return name in (self.nc4.variables if self.nc4 is not None else self.sp.variables)

    def has_dim(self, name):
        if self.h5 is not None:                            # NetCDF-4 stores every dimension as a dataset of its name
            return self.h5.has(name) and self.h5.has_attr(name, 'CLASS')
        return name in (self.nc4.dimensions if self.nc4 is not None else self.sp.dimensions)

    def dim(self, name):
        if self.h5 is not None:
            return int(self.h5.shape(name)[0])
        if self.nc4 is not None:
            return self.nc4.dimensions[name].size
        n = self.sp.dimensions[name]
        if n is None:                                      # record dimension
            n = self.sp._recs
        return int(n)

    def var(self, name, index=Ellipsis):
        if self.nc4 is not None:
            v = self.nc4.variables[name]
            v.set_auto_mask(False)                         # raw values: the reference assigns masked slabs into plain arrays
            return np.array(v[index])                      # (netCDF4 still applies scale_factor / add_offset)
        if self.h5 is not None:
            a = self.h5.read(name, index)
            has = lambda att: self.h5.has_attr(name, att)                          # noqa: E731
        else:
            v = self.sp.variables[name]
            a = np.array(v[index])
            has = lambda att: hasattr(v, att)                                      # noqa: E731
        if has('scale_factor') or has('add_offset'):       # packed variable: what netCDF4 would hand out
            sf = self.attr(name, 'scale_factor') if has('scale_factor') else 1.0
            ao = self.attr(name, 'add_offset') if has('add_offset') else 0.0
            a = a * sf + ao
        return a

    def has_attr(self, name, att):
        if self.h5 is not None:
            return self.h5.has_attr(name, att)
        return hasattr((self.nc4 if self.nc4 is not None else self.sp).variables[name], att)

    def fill_of(self, name):
        """the variable's _FillValue, or None"""
        return self.attr(name, '_FillValue') if self.has_attr(name, '_FillValue') else None

    def var_masked(self, name, index=Ellipsis):
        """What netCDF4's default auto-masking hands out for seeding / trajectory files (the reference reads them with
        it, ncio.py:199-326): entries equal to `_FillValue` come back masked, so that `np.min`/`np.max` skip them.
        Model fields (u_ice, v_ice, siconc, mesh) are read raw with var(): the reference assigns those slabs into plain
        arrays, fill data included (si3_part_tracker.py:372-374)."""
        a = self.var(name, index)
        fv = self.fill_of(name)
        if fv is None:
            return a
        hit = (a == np.asarray(fv).astype(a.dtype))
        return np.ma.masked_array(a, mask=hit) if np.any(hit) else a

    def attr(self, name, att):
        if self.h5 is not None:
            return self.h5.attr(name, att)
        v = (self.nc4 if self.nc4 is not None else self.sp).variables[name]
        a = getattr(v, att)
        return a.decode() if isinstance(a, bytes) else a

    def close(self):
        (self.h5 if self.h5 is not None else self.nc4 if self.nc4 is not None else self.sp).close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _project(ctx, lat, lon):
    from .tracking import ConvertGeo2CartesianNPSkm
    return ConvertGeo2CartesianNPSkm(lat, lon, 70., -45., ctx=ctx)


def _lon360(zlon, fill=None):
    """`np.mod(zlon, 360.)` evaluated in the variable's STORED dtype like the reference (ncio.py:50-51,86-87,303: the
    file-dtype array is passed to np.mod and, for seeding files, assigned back into it), and only then promoted to
    float64 -- for a negative float32 longitude the two orders differ by up to ~1.5e-5 degrees.  Entries equal to
    `fill` (netCDF4 would hand them out masked, and np.mod leaves masked entries alone) keep their raw value."""
    z = np.asarray(zlon)
    m = np.mod(z, z.dtype.type(360.)) if z.dtype.kind == 'f' else np.mod(z, 360.)
    if fill is not None:
        m = np.where(z == z.dtype.type(fill), z, m)
    return np.asarray(m, dtype=np.float64)


def _ctx_or_default(ctx):
    if ctx is not None:
        return ctx
    from .tracking import default_context
    return default_context()


def GetModelGrid(fNCmeshmask, alsoF=False, ctx=None):
    """Reference ncio.py:22-63 -> kmaskt, zlatT, zlonT, zYt, zXt, zYf, zXf, zResKM (+ kmaskf, zlatF, zlonF with alsoF).
    `ctx` (extra, optional): the context whose GPU projects the coordinates; default = the process-wide one."""
    ctx = _ctx_or_default(ctx)
    with _Reader(fNCmeshmask) as f:
        kmaskf = f.var('fmask', (0, 0)) if alsoF else None
        kmaskt = f.var('tmask', (0, 0))
        zlonF = f.var('glamf', 0); zlatF = f.var('gphif', 0)
        zlonT = f.var('glamt', 0); zlatT = f.var('gphit', 0)
        ze1T = f.var('e1t', 0) / 1000.
        ze2T = f.var('e2t', 0) / 1000.
    kmaskt = np.array(kmaskt, dtype='i1')
    zlatT = np.asarray(zlatT, dtype=np.float64); zlatF = np.asarray(zlatF, dtype=np.float64)
    zlonT, zlonF = _lon360(zlonT), _lon360(zlonF)
    zYt, zXt = _project(ctx, zlatT, zlonT)
    zYf, zXf = _project(ctx, zlatF, zlonF)
    zResKM = np.sqrt(ze1T * ze1T + ze2T * ze2T).astype(np.float64)
    if alsoF:
        return kmaskt, zlatT, zlonT, zYt, zXt, zYf, zXf, zResKM, kmaskf, zlatF, zlonF
    return kmaskt, zlatT, zlonT, zYt, zXt, zYf, zXf, zResKM


def GetModelUVGrid(fNCmeshmask, ctx=None):
    """Reference ncio.py:66-92 -> zYv, zXv, zYu, zXu."""
    ctx = _ctx_or_default(ctx)
    with _Reader(fNCmeshmask) as f:
        zlonV = f.var('glamv', 0); zlatV = f.var('gphiv', 0)
        zlonU = f.var('glamu', 0); zlatU = f.var('gphiu', 0)
    zYv, zXv = _project(ctx, np.asarray(zlatV, dtype=np.float64), _lon360(zlonV))
    zYu, zXu = _project(ctx, np.asarray(zlatU, dtype=np.float64), _lon360(zlonU))
    return zYv, zXv, zYu, zXu


def LoadNCtime(cfile, ltime2d=False):
    """Reference ncio.py:199-239."""
    with _Reader(cfile) as f:
        if not f.has_dim('time') or not f.has_var('time'):
            raise ValueError(' ERROR [LoadNCtime()]: no `time` dimension/variable found into input file!')
        Nt = f.dim('time')
        if f.attr('time', 'units') != tunits_default:
            raise ValueError(' ERROR [LoadNCtime()]: we expect "' + tunits_default + '" as units for the time record vector')
        ztime = f.var_masked('time')
        if ltime2d:
            if not f.has_var('time_pos'):
                raise ValueError(' ERROR [LoadNCtime()]: no variable `time_pos` found into input file!')
            if f.attr('time_pos', 'units') != tunits_default:
                raise ValueError(' ERROR [LoadNCtime()]: wrong units for the 2D time record array')
            ztime2d = f.var_masked('time_pos')
            if ztime2d.shape[0] != Nt:
                raise ValueError(' ERROR [LoadNCtime()]: array `time_pos` has not the same number of records as `time`!')
            return Nt, ztime, ztime2d
    return Nt, ztime


def LoadNCdata(cfile, krec=-1, lmask=False, lGetTimePos=False):
    """Reference ncio.py:243-326: seeding / trajectory file -> ztime, kBIDs, zLatLon, zYX[, zmsk][, ztpos].
    Longitudes come back in [0,360) (:303); float32 variables are promoted to float64."""
    need = ['id_buoy', 'latitude', 'longitude', 'y_pos', 'x_pos'] + (['time_pos'] if lGetTimePos else [])
    with _Reader(cfile) as f:
        for cd in ('time', 'buoy'):
            if not f.has_dim(cd):
                raise ValueError(' ERROR [LoadNCdata()]: no dimensions `' + cd + '` found into input file!')
        for cv in need:
            if not f.has_var(cv):
                raise ValueError(' ERROR [LoadNCdata()]: no variable `' + cv + '` found into input file!')
        Nt, nP = f.dim('time'), f.dim('buoy')
        if f.attr('time', 'units') != tunits_default:
            raise ValueError(' ERROR [LoadNCdata()]: wrong units for the time record vector')
        idxR = krec if krec >= 0 else slice(None)
        ztime = f.var('time', idxR)
        kBIDs = np.zeros(nP, dtype=np.int64)
        kBIDs[:] = f.var('id_buoy')
        zlat = np.asarray(f.var('latitude', idxR), dtype=np.float64)
        zlon = _lon360(f.var('longitude', idxR), fill=f.fill_of('longitude'))     # (:303) masked entries are left alone
        zy = np.asarray(f.var('y_pos', idxR), dtype=np.float64)
        zx = np.asarray(f.var('x_pos', idxR), dtype=np.float64)
        zmsk = f.var('mask', idxR) if lmask else None
        ztpos = f.var('time_pos', idxR) if lGetTimePos else None
    zLatLon = np.stack([zlat, zlon], axis=-1)
    zYX = np.stack([zy, zx], axis=-1)
    out = [ztime, kBIDs, zLatLon, zYX]
    if lmask:
        out.append(zmsk)
    if lGetTimePos:
        out.append(ztpos)
    return tuple(out)


def SeedFileTimeInfo(fSeedNc, ltime2d=False):
    """Reference ncio.py:329-352 -> idate0, idateN (rounded to the hour), SeedName, SeedBatch, time_pos or []."""
    from math import ceil, floor
    cSeed = path.basename(fSeedNc).replace('SELECTION_', '').replace('.nc', '')
    cBtch = path.basename(fSeedNc).split('_')[2]
    if ltime2d:
        ntr, zt, zt2d = LoadNCtime(fSeedNc, ltime2d=True)
        idate0, idateN = np.min(zt2d), np.max(zt2d)
    else:
        ntr, zt = LoadNCtime(fSeedNc)
        idate0, idateN = zt[0], zt[ntr - 1]
        zt2d = []
    idate0, idateN = int(floor(idate0 / 3600.) * 3600.), int(ceil(idateN / 3600.) * 3600.)
    return idate0, idateN, cSeed, cBtch, zt2d


def ModelFileTimeInfo(fModelNc):
    """Reference ncio.py:356-384 -> Nt, ztime (i4), idate0, idateN, nconf, nexpr (from the file NAME)."""
    with _Reader(fModelNc) as f:
        Nt = f.dim('time_counter')
        if f.attr('time_counter', 'units') != tunits_default:
            raise ValueError('ERROR: wrong units for time calendar in file: ' + fModelNc)
        ztime = np.array(f.var('time_counter'), dtype='i4')
    idate0, idateN = np.min(ztime), np.max(ztime)
    vn = path.basename(fModelNc).split('_')
    zz = vn[1].split('-')
    nconf = vn[0]
    if len(zz) == 1:
        zz = vn[0].split('-')
        nconf = zz[0]
    nexpr = zz[1]
    return Nt, ztime, idate0, idateN, nconf, nexpr


class ModelRecords:
    """Keeps the SI3 file open and hands out the raw (unmasked) slabs of one record
    (reference si3_part_tracker.py:359,365-374)."""

    def __init__(self, cf_uv):
        self.f = _Reader(cf_uv)

    def time(self, jrec):
        return int(self.f.var('time_counter', jrec))

    def fields(self, jrec, names=('u_ice', 'v_ice', 'siconc')):
        return tuple(self.f.var(n, jrec) for n in names)

    def fields_rows(self, jrec, j0, j1, names=('u_ice', 'v_ice', 'siconc')):
        """rows [j0,j1) only (row-band ingest: a rank reads just the rows its buoys can touch)"""
        return tuple(self.f.var(n, (jrec, slice(j0, j1))) for n in names)

    def fields_rows_into(self, jrec, j0, j1, outs, names=('u_ice', 'v_ice', 'siconc')):
        """Rows [j0,j1) of record `jrec` read INTO the arrays `outs` (see fields_box_into)"""
        return self.fields_box_into(jrec, j0, j1, None, None, outs, names)

    def fields_box_into(self, jrec, j0, j1, i0, i1, outs, names=('u_ice', 'v_ice', 'siconc')):
        """The box rows [j0,j1) x columns [i0,i1) of record `jrec` (what this rank's buoys can touch: the hyperslab of the
        reference's whole-record reads, si3_part_tracker.py:372-374) read INTO the arrays `outs` (libsitrk's pinned staging:
        file -> DMA-able memory in one pass through libhdf5; other backends assign).  i0 = i1 = None: whole rows.  Values
        must survive the cast to the outs' dtype exactly."""
        index = (jrec, slice(j0, j1)) if i0 is None else (jrec, slice(j0, j1), slice(i0, i1))
        for n, out in zip(names, outs):
            packed = self.f.has_attr(n, 'scale_factor') or self.f.has_attr(n, 'add_offset')
            if self.f.h5 is not None and not packed and self.f.h5.dtype(n).itemsize <= out.dtype.itemsize:
                self.f.h5.read(n, index, out=out)
                continue
            a = np.asarray(self.f.var(n, index))
            if a.dtype.newbyteorder('=') != out.dtype and not np.array_equal(a.astype(out.dtype).astype(a.dtype), a, equal_nan=True):
                raise ValueError("%s is not exactly representable as %s; allocate float64 records" % (n, out.dtype))
            out[...] = a

    def close(self):
        self.f.close()


def ncSaveCloudBuoys(cf_out, ptime, pIDs, pY, pX, pLat, pLon, mask=[], xtime=[], tunits=tunits_default, fillVal=FillValue,
                     corigin=None, cauthor='si3_part_tracker.py'):
    """Reference ncio.py:131-197: dims time (unlimited), buoy; time i4, buoy i4, id_buoy i8, latitude/longitude/
    y_pos/x_pos f4 (_FillValue -9999, zlib 9), optional mask i1 and time_pos i4; global Origin/About/Author."""
    (Nt,) = np.shape(ptime)
    (Nb,) = np.shape(pIDs)
    for a in (pY, pX, pLat, pLon):
        if np.shape(a) != (Nt, Nb):
            raise ValueError('ERROR [ncSaveCloudBuoys]: one of the 2D arrays has a wrong shape!!!')
    lSaveMask = (np.shape(mask) == (Nt, Nb))
    lSaveTime = (np.shape(xtime) == (Nt, Nb))
    os.makedirs(path.dirname(cf_out) or '.', exist_ok=True)
    about = 'Lagrangian sea-ice drift'
    author = 'Generated with `' + cauthor + '` of `sitrack` (L. Brodeau, 2023)'
    from . import h5lite
    if _nc4 is None and h5lite.writer_available():
        return _save_nc4_hdf5(cf_out, ptime, pIDs, pY, pX, pLat, pLon, mask if lSaveMask else None, xtime if lSaveTime else None,
                              tunits, fillVal, corigin, about, author)
    if _nc4 is not None:
        f = _nc4.Dataset(cf_out, 'w', format='NETCDF4')
        f.createDimension('time', None)
        f.createDimension('buoy', Nb)
        v_time = f.createVariable('time', 'i4', ('time',))
        v_buoy = f.createVariable('buoy', 'i4', ('buoy',))
        v_bid = f.createVariable('id_buoy', 'i8', ('buoy',))
        kw = dict(fill_value=fillVal, zlib=True, complevel=9)
        x_lat = f.createVariable('latitude', 'f4', ('time', 'buoy'), **kw)
        x_lon = f.createVariable('longitude', 'f4', ('time', 'buoy'), **kw)
        x_ykm = f.createVariable('y_pos', 'f4', ('time', 'buoy'), **kw)
        x_xkm = f.createVariable('x_pos', 'f4', ('time', 'buoy'), **kw)
        if lSaveMask:
            v_mask = f.createVariable('mask', 'i1', ('time', 'buoy'), zlib=True, complevel=9)
        if lSaveTime:
            x_tim = f.createVariable('time_pos', 'i4', ('time', 'buoy'), **kw)
    else:
        from scipy.io import netcdf_file
        f = netcdf_file(cf_out, 'w', version=2)
        f.createDimension('time', None)
        f.createDimension('buoy', Nb)
        v_time = f.createVariable('time', 'i4', ('time',))
        v_buoy = f.createVariable('buoy', 'i4', ('buoy',))
        v_bid = f.createVariable('id_buoy', 'f8', ('buoy',))          # NetCDF-3 has no int64
        v_bid.note = 'int64 IDs stored as float64 (NetCDF-3 fall-back writer)'
        if np.any(np.abs(np.asarray(pIDs, dtype=np.int64)) > 2 ** 53):
            raise ValueError('ncSaveCloudBuoys: a buoy ID beyond 2^53 cannot be stored by the NetCDF-3 fall-back writer '
                             '(float64 id_buoy); install netCDF4 or make libhdf5 + libhdf5_hl loadable')
        x_lat = f.createVariable('latitude', 'f4', ('time', 'buoy'))
        x_lon = f.createVariable('longitude', 'f4', ('time', 'buoy'))
        x_ykm = f.createVariable('y_pos', 'f4', ('time', 'buoy'))
        x_xkm = f.createVariable('x_pos', 'f4', ('time', 'buoy'))
        for v in (x_lat, x_lon, x_ykm, x_xkm):
            v._FillValue = np.float32(fillVal)
        if lSaveMask:
            v_mask = f.createVariable('mask', 'i1', ('time', 'buoy'))
        if lSaveTime:
            x_tim = f.createVariable('time_pos', 'i4', ('time', 'buoy'))
            x_tim._FillValue = np.int32(fillVal)
    v_time.units = tunits
    v_bid.units = 'ID of buoy'
    x_lat.units = 'degrees north'
    x_lon.units = 'degrees south'
    x_ykm.units = 'km'
    x_xkm.units = 'km'
    if lSaveTime:
        x_tim.units = tunits
    v_buoy[:] = np.arange(Nb, dtype='i4')
    v_bid[:] = np.asarray(pIDs)[:]
    # whole arrays at once (the reference writes record by record; the NetCDF-3 writer would re-allocate per record)
    v_time[:] = np.asarray(ptime).astype('i4')
    x_lat[:, :] = np.asarray(pLat, dtype=np.float32)
    x_lon[:, :] = np.asarray(pLon, dtype=np.float32)
    x_ykm[:, :] = np.asarray(pY, dtype=np.float32)
    x_xkm[:, :] = np.asarray(pX, dtype=np.float32)
    if lSaveMask:
        v_mask[:, :] = np.asarray(mask, dtype='i1')
    if lSaveTime:
        x_tim[:, :] = np.asarray(xtime).astype('i4')
    if corigin:
        f.Origin = corigin
    f.About = about
    f.Author = author
    f.close()
    return 0


def _nc4_hdf5_create(cf_out, Nt, pIDs, with_mask, with_xtime, tunits, fillVal, corigin, about, author):
    """The NetCDF-4 file of ncSaveCloudBuoys without the netCDF4 package (reference ncio.py:143-195 -- dimensions, variable types,
    `_FillValue`, shuffle + deflate at level 9, units, global attributes), created at its final length of Nt records through
    libhdf5; the record variables are still to be written.  Returns the open h5lite.NC4Writer."""
    from . import h5lite
    Nb = len(pIDs)
    lvl = int(os.environ.get('SITRK_NC_COMPLEVEL', '9'))          # the reference's complevel; lower it for 10^7-buoy files
    w = h5lite.NC4Writer(cf_out)
    try:
        w.createDimension('time', None)
        w.createDimension('buoy', Nb)
        w.createVariable('time', 'i4', ('time',), nrec=Nt)
        w.createVariable('buoy', 'i4', ('buoy',))
        w.createVariable('id_buoy', 'i8', ('buoy',))
        kw = dict(fill_value=fillVal, zlib=True, complevel=lvl, nrec=Nt)
        for name in ('latitude', 'longitude', 'y_pos', 'x_pos'):
            w.createVariable(name, 'f4', ('time', 'buoy'), **kw)
        if with_mask:
            w.createVariable('mask', 'i1', ('time', 'buoy'), zlib=True, complevel=lvl, nrec=Nt)
        if with_xtime:
            w.createVariable('time_pos', 'i4', ('time', 'buoy'), **kw)
        w.set_attr('time', 'units', tunits)
        w.set_attr('id_buoy', 'units', 'ID of buoy')
        w.set_attr('latitude', 'units', 'degrees north')
        w.set_attr('longitude', 'units', 'degrees south')
        w.set_attr('y_pos', 'units', 'km')
        w.set_attr('x_pos', 'units', 'km')
        if with_xtime:
            w.set_attr('time_pos', 'units', tunits)
        w.write('buoy', np.arange(Nb, dtype='i4'))
        w.write('id_buoy', np.asarray(pIDs, dtype=np.int64))
        if corigin:
            w.set_global('Origin', corigin)
        w.set_global('About', about)
        w.set_global('Author', author)
    except BaseException:
        w.abort()                       # ids closed, partial file removed; never raises: the original error is the one seen
        raise
    return w


def _save_nc4_hdf5(cf_out, ptime, pIDs, pY, pX, pLat, pLon, mask, xtime, tunits, fillVal, corigin, about, author):
    """ncSaveCloudBuoys without the netCDF4 package: the same NetCDF-4 file written through libhdf5, whole arrays at once."""
    w = _nc4_hdf5_create(cf_out, len(ptime), pIDs, mask is not None, xtime is not None, tunits, fillVal, corigin, about, author)
    try:
        w.write('time', np.asarray(ptime).astype('i4'))
        w.write('latitude', np.asarray(pLat, dtype=np.float32))
        w.write('longitude', np.asarray(pLon, dtype=np.float32))
        w.write('y_pos', np.asarray(pY, dtype=np.float32))
        w.write('x_pos', np.asarray(pX, dtype=np.float32))
        if mask is not None:
            w.write('mask', np.asarray(mask, dtype='i1'))
        if xtime is not None:
            w.write('time_pos', np.asarray(xtime).astype('i4'))
    except BaseException:
        w.abort()
        raise
    w.close()
    return 0


class CloudBuoysStream:
    """ncSaveCloudBuoys RECORD BY RECORD: the same file (reference ncio.py:131-197, which itself writes record by record inside its
    loop :176-190 -- after having held the whole (Nt+1, nP, 2) series in memory twice, si3_part_tracker.py:324-330), fed while the
    records are produced, so that the writer's memory is O(nP) whatever the number of records: the `-F` series of 1e7 buoys x
    thousands of records is 0.3-1 TB as arrays.  The time axis is known in advance (`ptime`, all Nt output records); `put(k, ...)`
    casts record k to the file's types and hands it to the backend (libhdf5 through h5lite: rows deflated on a thread pool and
    written as raw chunks, flushed every `flush_bytes`; netCDF4 or the NetCDF-3 fall-back: one record slice per variable)."""

    def __init__(self, cf_out, ptime, pIDs, with_mask=True, tunits=tunits_default, fillVal=FillValue, corigin=None,
                 cauthor='si3_part_tracker.py', flush_bytes=None):
        from . import h5lite
        self.path, self.Nt, self.Nb = cf_out, len(ptime), len(pIDs)
        self.with_mask, self.nput = with_mask, 0
        os.makedirs(path.dirname(cf_out) or '.', exist_ok=True)
        about = 'Lagrangian sea-ice drift'
        author = 'Generated with `' + cauthor + '` of `sitrack` (L. Brodeau, 2023)'
        self.w = self.f = None
        if _nc4 is None and h5lite.writer_available():
            self.w = _nc4_hdf5_create(cf_out, self.Nt, pIDs, with_mask, False, tunits, fillVal, corigin, about, author)
            # rows wait (as arrays + deflated blobs) until this many bytes are queued: the writer's share of the peak memory
            self.flush_bytes = int(flush_bytes if flush_bytes is not None else os.environ.get('SITRK_NC_FLUSH_BYTES', str(96 << 20)))
            self.queued = 0
            self.w.write('time', np.asarray(ptime).astype('i4'))
            return
        if _nc4 is not None:
            f = _nc4.Dataset(cf_out, 'w', format='NETCDF4')
            kw = dict(fill_value=fillVal, zlib=True, complevel=9)
        else:
            from scipy.io import netcdf_file
            f = netcdf_file(cf_out, 'w', version=2)
            kw = {}
            if np.any(np.abs(np.asarray(pIDs, dtype=np.int64)) > 2 ** 53):
                raise ValueError('CloudBuoysStream: a buoy ID beyond 2^53 cannot be stored by the NetCDF-3 fall-back writer')
        f.createDimension('time', None)
        f.createDimension('buoy', self.Nb)
        v_time = f.createVariable('time', 'i4', ('time',))
        v_buoy = f.createVariable('buoy', 'i4', ('buoy',))
        v_bid = f.createVariable('id_buoy', 'i8' if _nc4 is not None else 'f8', ('buoy',))
        self.v = {n: f.createVariable(n, 'f4', ('time', 'buoy'), **kw) for n in ('latitude', 'longitude', 'y_pos', 'x_pos')}
        if with_mask:
            self.v['mask'] = f.createVariable('mask', 'i1', ('time', 'buoy'), **({'zlib': True, 'complevel': 9} if _nc4 is not None else {}))
        if _nc4 is None:
            for n in ('latitude', 'longitude', 'y_pos', 'x_pos'):
                self.v[n]._FillValue = np.float32(fillVal)
        v_time.units = tunits
        v_bid.units = 'ID of buoy'
        for n, un in (('latitude', 'degrees north'), ('longitude', 'degrees south'), ('y_pos', 'km'), ('x_pos', 'km')):
            self.v[n].units = un
        v_buoy[:] = np.arange(self.Nb, dtype='i4')
        v_bid[:] = np.asarray(pIDs)[:]
        v_time[:] = np.asarray(ptime).astype('i4')
        if corigin:
            f.Origin = corigin
        f.About = about
        f.Author = author
        self.f = f

    def put(self, k, pY, pX, pLat, pLon, mask=None):
        """record k of the file (0 <= k < Nt): (nP,) arrays"""
        if not (0 <= k < self.Nt):
            raise IndexError("record %d outside the %d records of %s" % (k, self.Nt, self.path))
        rows = {'latitude': np.asarray(pLat, dtype=np.float32), 'longitude': np.asarray(pLon, dtype=np.float32),
                'y_pos': np.asarray(pY, dtype=np.float32), 'x_pos': np.asarray(pX, dtype=np.float32)}
        if self.with_mask:
            rows['mask'] = np.asarray(mask, dtype='i1')
        for n, r in rows.items():
            if r.shape != (self.Nb,):
                raise ValueError('CloudBuoysStream.put: %s has shape %s, expected (%d,)' % (n, r.shape, self.Nb))
            if self.w is not None:
                self.w.write_rows(n, k, r[None, :])
                self.queued += r.nbytes
            else:
                self.v[n][k, :] = r
        if self.w is not None and self.queued >= self.flush_bytes:
            self.w.flush()
            self.queued = 0
        self.nput += 1

    def close(self):
        if self.nput != self.Nt:
            self.abort()
            raise ValueError('%s: %d of %d records were written' % (self.path, self.nput, self.Nt))
        if self.w is not None:
            self.w.close()
        else:
            self.f.close()
        self.w = self.f = None

    def abort(self):
        """give up on the file (an error on the way): ids closed, the partial file removed; never raises"""
        try:
            if self.w is not None:
                self.w.abort()
            elif self.f is not None:
                self.f.close()
                os.remove(self.path)
        except Exception:               # noqa: BLE001
            pass
        self.w = self.f = None
