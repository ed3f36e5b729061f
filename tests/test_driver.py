"""Driver plumbing: file-name logic, record windows, date helpers on CPU;
the whole command line end to end on the GPU against an independent oracle-driven run."""
import os

import numpy as np
import pytest

import sitrack_amd as sit
from sitrack_amd import driver as drv
from sitrack_amd import ncio
from sitrack_amd import synthetic as syn


def test_output_name_of_the_reference_readme_demo():
    # reference README.md:83-97: seeding file, model file, `-e 1997-04-20 -F` -> documented output name
    cdtbin, csfkm = drv.seed_name_tokens('sitrack_seeding_nemoTsi3_19961215_00_HSS5.nc')
    assert (cdtbin, csfkm) == ('_idlSeed', '')
    seed_batch = 'sitrack_seeding_nemoTsi3_19961215_00_HSS5.nc'.split('_')[2]
    t0 = drv.clock2epoch('1996-12-15_00:00:00')
    t1 = drv.clock2epoch('1997-04-20')
    name = ('./nc/NEMO-SI3_NANUK4_BBM23U06_tracking_' + seed_batch + cdtbin + '_' + drv.date_tag(t0) + '_' + drv.date_tag(t1)
            + csfkm + '.nc')
    assert name == './nc/NEMO-SI3_NANUK4_BBM23U06_tracking_nemoTsi3_idlSeed_19961215h00_19970420h00.nc'


def test_seed_name_tokens_variants():
    assert drv.seed_name_tokens('sitrack_seeding_sidfex_19961215_00_HSS5.nc') == ('_idlSeed', '')
    assert drv.seed_name_tokens('sitrack_seeding_nemoTsi3_19961215_00_HSS5_10km.nc') == ('_idlSeed', '_10km')
    # RGPS-style selection files: ..._dt72_10km.nc  /  ..._dt72.nc
    assert drv.seed_name_tokens('SELECTION_RGPS_S008_dt72_19970104h00_19970107h00_10km.nc')[1] == '_10km'
    with pytest.raises((ValueError, IndexError)):
        drv.seed_name_tokens('whatever_file_name_without_tokens.nc')


def test_dates_roundtrip():
    t = 850608000                                   # the reference fixture's time: 1996-12-15 00:00 UTC
    assert drv.epoch2clock(t) == '1996-12-15_00:00:00'
    assert drv.clock2epoch('1996-12-15_00:00:00') == t and drv.clock2epoch('1996-12-15') == t and drv.clock2epoch('19961215') == t
    assert drv.date_tag(t + 3 * 3600) == '19961215h03'


def test_record_windows():
    base = 850608000
    vt = (base + 1800 + 3600 * np.arange(24)).astype('i4')
    kstrt, kstop = 0, 23
    iTmA, iTmB = vt[kstrt], vt[kstop]
    zT = np.array([[base, base + 5 * 3600, base + 2 * 3600 + 1800], [vt[-1] + 1800, base + 12 * 3600, vt[-1]]])
    z1, zL = drv.record_windows(zT, vt, kstrt, kstop, iTmA, iTmB, 3)
    # hand-evaluated from reference si3_part_tracker.py:289-312 (strict comparisons against record centres +- rdt/2)
    assert list(z1) == [0, 4, 2]
    assert list(zL) == [23, 12, 23]
    # the all-buoys-at-once form on an increasing time axis == the reference's per-buoy np.where loop, incl. times
    # exactly on the +-rdt/2 boundaries
    rng = np.random.default_rng(3)
    n = 4000
    zT = np.stack([rng.integers(vt[0] - 7200, vt[-1] + 1800, n), rng.integers(vt[0] + 1800, vt[-1] + 7200, n)])
    zT[0, ::9] = vt[rng.integers(0, 24, len(zT[0, ::9]))] + rng.choice([-1800, 1800, 0, 1799, 1801], len(zT[0, ::9]))
    zT[1, ::7] = vt[rng.integers(0, 24, len(zT[1, ::7]))] + rng.choice([-1800, 1800, 0, -1799, -1801], len(zT[1, ::7]))
    zT[0][zT[0] == iTmA + 1800] += 1                       # (exactly there the reference's own search comes back empty)
    fast = drv.record_windows(zT, vt, kstrt, kstop, iTmA, iTmB, n)
    late = np.where(zT[0] >= iTmA + 1800)[0]; early = np.where(zT[1] < iTmB - 1800)[0]
    slow = drv._record_windows_loop(zT, vt, np.zeros(n, dtype=int) + kstrt, np.zeros(n, dtype=int) + kstop, late, early, 1800)
    assert np.array_equal(fast[0], slow[0]) and np.array_equal(fast[1], slow[1])
    with pytest.raises(IndexError):
        drv.record_windows(np.array([[iTmA + 1800], [iTmB]]), vt, kstrt, kstop, iTmA, iTmB, 1)
    assert len(np.unique(fast[0])) > 10 and len(np.unique(fast[1])) > 10


def test_longitudes_wrap_in_the_stored_dtype():
    """np.mod(lon, 360.) on the float32 the file holds, THEN float64 (reference ncio.py:50-51,86-87,303); the other order
    differs by up to ~1.5e-5 degrees for negative longitudes."""
    rng = np.random.default_rng(11)
    lon4 = rng.uniform(-180, 0, 20000).astype('f4')
    got = ncio._lon360(lon4)
    assert got.dtype == np.float64 and np.array_equal(got, np.mod(lon4, np.float32(360.)).astype('f8'))
    other = np.mod(lon4.astype('f8'), 360.)
    assert (got != other).any() and np.abs(got - other).max() < 2e-5
    lon8 = rng.uniform(-180, 360, 1000)
    assert np.array_equal(ncio._lon360(lon8), np.mod(lon8, 360.))
    assert np.array_equal(ncio._lon360(np.array([-9999., -10.], dtype='f4'), fill=-9999.), [-9999., 350.])


def test_seed_cache_writer_is_a_plain_npz(tmp_path):
    """The seed cache (reference si3_part_tracker.py:255: np.savez_compressed) is written by an own ZIP64 writer that deflates
    4-MB pieces on a thread pool: `np.load` and zipfile's CRC check must take it like any .npz -- scalars stay 0-d,
    empty and non-contiguous arrays, members longer than one piece."""
    import zipfile
    rng = np.random.default_rng(2)
    n = 700_001                                                   # xPosG0: 11 MB = three pieces
    arrs = dict(nP=n, xPosG0=rng.random((n, 2)), IDs=np.arange(n) * 7 + 300534062025510, VRTCS=rng.integers(0, 500, (1000, 2, 4)),
                empty=np.zeros((0, 2)), strided=np.arange(20).reshape(4, 5)[:, ::2], one=np.float32(1.5))
    f = str(tmp_path / "Initialized_buoys_x.npz")
    drv._savez_deflate(f, **arrs)
    with np.load(f) as z:
        assert sorted(z.files) == sorted(arrs)
        for k, v in arrs.items():
            assert np.array_equal(z[k], v) and z[k].dtype == np.asarray(v).dtype and z[k].shape == np.shape(v), k
    assert zipfile.ZipFile(f).testzip() is None


def test_record_windows_skip_masked_time_positions():
    base = 850608000
    vt = (base + 1800 + 3600 * np.arange(24)).astype('i4')
    zT = np.ma.masked_equal(np.array([[base, -9999, base + 5 * 3600], [vt[-1] + 1800, base + 12 * 3600, -9999]]), -9999)
    z1, zL = drv.record_windows(zT, vt, 0, 23, vt[0], vt[23], 3)
    assert list(z1) == [0, 0, 4] and list(zL) == [23, 12, 23]


def test_ncio_schema_and_roundtrip_both_backends(tmp_path, monkeypatch):
    """ncSaveCloudBuoys / LoadNCdata / LoadNCtime / *TimeInfo with the NetCDF-3 fall-back and with the netCDF4 code path
    (driven through a minimal stand-in for the absent package: tests/fakes/netCDF4.py)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "fakes"))
    import netCDF4 as fake
    sys.path.pop(0)
    rng = np.random.default_rng(0)
    Nt, Nb = 3, 7
    t = np.array([850608000, 850611600, 850615200])
    ids = (300534062025510 + np.arange(Nb)).astype(np.int64)
    Y, X = rng.uniform(-3000, 3000, (Nt, Nb)), rng.uniform(-3000, 3000, (Nt, Nb))
    La, Lo = rng.uniform(60, 90, (Nt, Nb)), rng.uniform(-180, 180, (Nt, Nb))
    Y[2, 3] = X[2, 3] = La[2, 3] = Lo[2, 3] = -9999.
    msk = np.ones((Nt, Nb), dtype='i1'); msk[2, 3] = 0
    tp = np.tile(t[:, None], (1, Nb)); tp[2, 3] = -9999
    for backend in (None, fake):
        monkeypatch.setattr(ncio, "_nc4", backend)
        f = str(tmp_path / ("NEMO-SI3_X_Y_tracking_nemoTsi3_idlSeed_%s.nc" % ("nc4" if backend else "nc3")))
        ncio.ncSaveCloudBuoys(f, t, ids, Y, X, La, Lo, mask=msk, xtime=tp, corigin="unit-test")
        tt, bid, ll, yx, mk, tpos = ncio.LoadNCdata(f, krec=-1, lmask=True, lGetTimePos=True)
        assert np.array_equal(tt, t) and np.array_equal(bid, ids) and np.array_equal(mk, msk) and np.array_equal(tpos, tp)
        assert np.array_equal(yx[..., 0], Y.astype('f4').astype('f8')) and np.array_equal(yx[..., 1], X.astype('f4').astype('f8'))
        assert np.array_equal(ll[..., 0], La.astype('f4').astype('f8'))
        # reference ncio.py:303: np.mod on the file-dtype (f4) array, assigned back into it; masked (_FillValue) entries untouched
        lo4 = Lo.astype('f4')
        want = np.where(lo4 == np.float32(-9999.), lo4, np.mod(lo4, np.float32(360.))).astype('f8')
        assert np.array_equal(ll[..., 1], want) and want[2, 3] == -9999.
        t1, b1, ll1, yx1 = ncio.LoadNCdata(f, krec=1)
        assert int(t1) == t[1] and yx1.shape == (Nb, 2) and np.array_equal(yx1, yx[1])
        assert ncio.LoadNCtime(f)[0] == Nt
        i0, iN, name, batch, t2d = ncio.SeedFileTimeInfo(f, ltime2d=True)
        assert (i0, iN) == (850608000, 850615200)      # the _FillValue entry of time_pos is masked (netCDF4 auto-mask), not a date
        assert np.ma.isMaskedArray(t2d) and t2d.mask[2, 3] and t2d.mask.sum() == 1
        assert batch == "Y"
        with ncio._Reader(f) as r:
            assert r.attr("time", "units") == ncio.tunits_default and r.attr("longitude", "units") == "degrees south"
            assert r.attr("id_buoy", "units") == "ID of buoy" and r.dim("buoy") == Nb
    # the netCDF4 branch asks for the reference's variable types, fill values and compression (ncio.py:153-171)
    monkeypatch.setattr(ncio, "_nc4", fake)
    made = {}
    orig = fake.Dataset.createVariable

    def spy(self, name, typ, dims, **kw):
        made[name] = (typ, dims, kw)
        return orig(self, name, typ, dims, **kw)
    monkeypatch.setattr(fake.Dataset, "createVariable", spy)
    ncio.ncSaveCloudBuoys(str(tmp_path / "a_b_c_d.nc"), t, ids, Y, X, La, Lo, mask=msk, xtime=tp)
    assert made["id_buoy"][0] == 'i8' and made["time"][0] == 'i4' and made["mask"][0] == 'i1' and made["time_pos"][0] == 'i4'
    for v in ("latitude", "longitude", "y_pos", "x_pos"):
        assert made[v] == ('f4', ('time', 'buoy'), dict(fill_value=-9999., zlib=True, complevel=9))


def _h5_header(fname):
    """`h5dump -H -p` parsed into {object: {"type", "space", "filters", "fill", "attrs": {name: (type, space)}}}"""
    import re
    import shutil
    import subprocess
    exe = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    if not os.path.exists(exe):
        pytest.skip("no h5dump here")
    txt = subprocess.run([exe, "-H", "-p", fname], capture_output=True, text=True, check=True).stdout
    objs, cur, att, stack = {"/": {"attrs": {}}}, "/", None, []
    lines = txt.splitlines()
    k = 0

    def block(k0):
        """text of the {...} block opening on line k0 (one line or several), index of its last line"""
        depth, out, k1 = 0, [], k0
        while True:
            out.append(lines[k1].strip())
            depth += lines[k1].count("{") - lines[k1].count("}")
            if depth <= 0:
                return " ".join(out), k1
            k1 += 1
    while k < len(lines):
        ln = lines[k].strip()
        m = re.match(r'DATASET "(.*)" \{', ln)
        if m:
            cur, att = m.group(1), None
            objs[cur] = {"attrs": {}}
        m = re.match(r'ATTRIBUTE "(.*)" \{', ln)
        if m:
            att = m.group(1)
            objs[cur]["attrs"][att] = [None, None]
        if ln.startswith("DATATYPE"):
            t, k = block(k)
            t = re.sub(r"\s+", " ", t)
            if att:
                objs[cur]["attrs"][att][0] = t
            else:
                objs[cur]["type"] = t
        elif ln.startswith("DATASPACE"):
            if att:
                objs[cur]["attrs"][att][1] = ln
                att = None
            else:
                objs[cur]["space"] = ln
        elif ln.startswith("FILTERS"):
            t, k = block(k)
            objs[cur]["filters"] = re.sub(r"\s+", " ", t)
        elif ln.startswith("FILLVALUE"):
            t, k = block(k)
            objs[cur]["fill"] = re.sub(r"\s+", " ", t)
        elif ln.startswith("STORAGE_LAYOUT"):
            t, k = block(k)
            objs[cur]["layout"] = "CHUNKED" if "CHUNKED" in t else "CONTIGUOUS"
        k += 1
    return objs


def test_netcdf4_output_header_equals_the_reference_files(tmp_path, golden):
    """The writer used when the `netCDF4` package is absent (libhdf5 + dimension scales) against the reference's own
    committed NetCDF-4 file, tools/nc/sitrack_seeding_sidfex_19961215_00_HSS5.nc__KEEP (written by netCDF4-python through
    `ncSaveCloudBuoys`, reference ncio.py:131-197): same content in, then `h5dump -H -p` variable for variable -- HDF5
    type, dataspace incl. the unlimited `time`, chunked layout, shuffle + deflate 9, fill value, and every attribute's
    type and shape (`id_buoy` is int64).  Allowed differences: netCDF-C's bookkeeping attributes `_NCProperties` /
    `_Netcdf4Coordinates`, which readers do not need."""
    from sitrack_amd import h5lite
    if not h5lite.writer_available():
        pytest.skip("libhdf5 / libhdf5_hl not loadable here")
    ref = os.path.join(os.path.dirname(__file__), "golden", "sitrack_seeding_sidfex_19961215_00_HSS5.nc")
    g = golden("g7_projection.npz")
    f = str(tmp_path / "sitrack_seeding_sidfex_19961215_00_HSS5.nc")
    import sitrack_amd.ncio as nio
    saved = nio._nc4
    nio._nc4 = None
    try:
        ncio.ncSaveCloudBuoys(f, g["time"].astype('i4'), g["id_buoy"], g["y_pos"][None], g["x_pos"][None], g["latitude"][None],
                              g["longitude"][None], corigin='idealized_seeding', cauthor='generate_sidfex_seeding.py')
    finally:
        nio._nc4 = saved
    assert h5lite.is_hdf5(f)
    a, b = _h5_header(ref), _h5_header(f)
    skip = {"_NCProperties", "_Netcdf4Coordinates"}
    assert set(a) == set(b) == {"/", "buoy", "id_buoy", "latitude", "longitude", "time", "x_pos", "y_pos"}
    for name in a:
        for key in ("type", "space", "filters", "fill", "layout"):
            if name == "time" and key == "fill":
                continue                 # netCDF-C gives an unfilled int variable its default fill (-2147483647); every record is written here
            if name in ("buoy", "id_buoy") and key in ("layout", "fill"):
                continue                 # small fixed-size variables: contiguous either way; fill-time bookkeeping only
            assert a[name].get(key) == b[name].get(key), (name, key, a[name].get(key), b[name].get(key))
        ra = {k: v for k, v in a[name]["attrs"].items() if k not in skip}
        rb = {k: v for k, v in b[name]["attrs"].items() if k not in skip}
        assert ra == rb, (name, ra, rb)
    assert "H5T_STD_I64LE" in b["id_buoy"]["type"] and "H5S_UNLIMITED" in b["latitude"]["space"]
    assert "DEFLATE { LEVEL 9 }" in b["x_pos"]["filters"] and "SHUFFLE" in b["x_pos"]["filters"]
    # and the data round-trip through the reader, next to the reference file's own
    for src in (ref, f):
        zt, ids, zg, zc = ncio.LoadNCdata(src, krec=0)
        assert np.array_equal(ids, g["id_buoy"]) and ids.dtype == np.int64 and int(zt) == int(g["time"][0])
        assert np.array_equal(zc[:, 0].astype('f4'), g["y_pos"]) and np.array_equal(zg[:, 0].astype('f4'), g["latitude"])
    # IDs beyond 2^53 survive (they could not as the float64 of the NetCDF-3 fall-back)
    big = np.array([2 ** 53 + 1, 2 ** 62 + 12345, 300534062025510], dtype=np.int64)
    f2 = str(tmp_path / "a_b_c.nc")
    nio._nc4 = None
    try:
        ncio.ncSaveCloudBuoys(f2, np.array([1, 2]), big, np.zeros((2, 3)), np.zeros((2, 3)), np.zeros((2, 3)), np.zeros((2, 3)),
                              mask=np.ones((2, 3), dtype='i1'), xtime=np.ones((2, 3), dtype=int))
    finally:
        nio._nc4 = saved
    assert np.array_equal(ncio.LoadNCdata(f2, krec=-1)[1], big)
    hdr = _h5_header(f2)
    assert "DEFLATE { LEVEL 9 }" in hdr["mask"]["filters"] and "H5T_STD_I8LE" in hdr["mask"]["type"] and "_FillValue" not in hdr["mask"]["attrs"]
    assert "H5T_STD_I32LE" in hdr["time_pos"]["type"] and "_FillValue" in hdr["time_pos"]["attrs"]


def test_netcdf4_writer_parallel_chunks_round_trip(tmp_path, monkeypatch):
    """Large variables are shuffled + deflated chunk by chunk on a thread pool and written with H5Dwrite_chunk; what comes
    back through libhdf5's own filter pipeline must be the data (row length not a multiple of the chunk: padded edge chunk)."""
    from sitrack_amd import h5lite
    if not h5lite.writer_available():
        pytest.skip("libhdf5 / libhdf5_hl not loadable here")
    monkeypatch.setattr(ncio, "_nc4", None)
    monkeypatch.setenv("SITRK_NC_COMPLEVEL", "2")
    rng = np.random.default_rng(1)
    Nb = (1 << 20) + 4321
    ids = np.arange(Nb, dtype=np.int64) * 3 + 1
    Y = np.round(rng.uniform(-3000, 3000, (2, Nb)), 1); X = np.round(rng.uniform(-3000, 3000, (2, Nb)), 1)
    Y[1, ::5] = -9999.
    msk = (rng.random((2, Nb)) < 0.9).astype('i1')
    f = str(tmp_path / "NEMO-SI3_A_B_tracking12_x.nc")
    ncio.ncSaveCloudBuoys(f, np.array([10, 20]), ids, Y, X, Y * 0 + 80., X * 0 + 10., mask=msk)
    tt, bid, ll, yx, mk = ncio.LoadNCdata(f, krec=-1, lmask=True)
    assert np.array_equal(bid, ids) and np.array_equal(mk, msk) and list(tt) == [10, 20]
    assert np.array_equal(yx[..., 0], Y.astype('f4').astype('f8')) and np.array_equal(yx[..., 1], X.astype('f4').astype('f8'))
    hdr = _h5_header(f)
    assert "CHUNKED" == hdr["y_pos"]["layout"] and "DEFLATE { LEVEL 2 }" in hdr["y_pos"]["filters"] and "SHUFFLE" in hdr["mask"]["filters"]


def test_netcdf4_writer_failure_closes_and_removes_the_partial_file(tmp_path, monkeypatch):
    """A failure while the queued chunks are handed to libhdf5 (or while the scales are attached): every id is closed, the
    half-written file is removed, the ORIGINAL error is what the caller sees, and the bounded queue flushes early."""
    from sitrack_amd import h5lite
    if not h5lite.writer_available():
        pytest.skip("libhdf5 / libhdf5_hl not loadable here")
    monkeypatch.setattr(ncio, "_nc4", None)
    monkeypatch.setenv("SITRK_NC_COMPLEVEL", "1")
    Nb = (1 << 20) + 7
    ids = np.arange(Nb, dtype=np.int64)
    Y = np.zeros((2, Nb)); X = np.ones((2, Nb))
    f = str(tmp_path / "NEMO-SI3_A_B_tracking12_fail.nc")
    real_flush = h5lite.NC4Writer.flush
    calls = []

    def bad_flush(self):
        calls.append(len(getattr(self, "_pending", [])))
        real_flush(self)
        raise OSError("disk full (injected)")
    monkeypatch.setattr(h5lite.NC4Writer, "flush", bad_flush)
    with pytest.raises(OSError, match="disk full"):
        ncio.ncSaveCloudBuoys(f, np.array([10, 20]), ids, Y, X, Y, X)
    assert not os.path.exists(f) and calls and calls[0] > 0
    # the same writer class then writes a good file (no id leaked that would keep the name busy)
    monkeypatch.setattr(h5lite.NC4Writer, "flush", real_flush)
    monkeypatch.setenv("SITRK_NC_PENDING_BYTES", str(5 << 20))          # every large variable flushes by itself
    nfl = []
    monkeypatch.setattr(h5lite.NC4Writer, "flush", lambda self: (nfl.append(1), real_flush(self))[1])
    ncio.ncSaveCloudBuoys(f, np.array([10, 20]), ids, Y, X, Y, X)
    assert len(nfl) >= 4
    tt, bid, ll, yx = ncio.LoadNCdata(f, krec=-1)[:4]
    assert np.array_equal(bid, ids) and np.all(yx[..., 1] == 1.0)
    # an error raised by the caller's own data before close(): abort() path
    w = h5lite.NC4Writer(f)
    w.createDimension('time', None)
    w.abort(); w.abort()
    assert not os.path.exists(f) and w.fid == -1


def _write_nc3(fname, dims, variables, attrs=None):
    from scipy.io import netcdf_file
    f = netcdf_file(fname, 'w', version=2)
    for d, n in dims.items():
        f.createDimension(d, n)
    for name, (typ, dd, data, att) in variables.items():
        v = f.createVariable(name, typ, dd)
        v[:] = data
        for k, val in (att or {}).items():
            setattr(v, k, val)
    for k, val in (attrs or {}).items():
        setattr(f, k, val)
    f.close()


def _write_h5_like_nc3(fname, dims, variables, attrs=None):
    """same arguments as _write_nc3, written as an HDF5 file laid out like NetCDF-4 (tests/h5write.py)"""
    import h5write
    unl = [d for d, n in dims.items() if n is None]
    vv = {}
    for name, (typ, dd, data, att) in variables.items():
        vv[name] = (np.asarray(data).astype(np.dtype(typ).newbyteorder('=')), att, bool(dd) and dd[0] in unl)
    dl = {d: (n if n is not None else max([np.shape(v[2])[0] for v in variables.values() if v[1] and v[1][0] == d] + [0]))
          for d, n in dims.items()}
    h5write.write_h5(fname, vv, dims=dl)


def make_case(tmp, nrec=14, nP=300, two_d_time=False, Nj=60, Ni=70, dkm=10.0, fmt="nc3", ice_mask_seeding=False):
    """Synthetic NANUK-like inputs: mesh_mask, icemod (hourly), seeding file -- as NetCDF-3 files, or (fmt="hdf5") as
    HDF5 files laid out like the NetCDF-4 files NEMO and the reference's seeding tools write.
    ice_mask_seeding: keep only the random seeds whose nearest T-point `nemoSeed` (reference tracking.py:365-442) would
    seed: tmask = 1, latitude >= 55, siconc >= 0.9 at the first record."""
    from oracle import oracle as orc
    _write_nc3 = globals()["_write_nc3"] if fmt == "nc3" else _write_h5_like_nc3
    g = syn.make_grid(Nj, Ni, dkm=dkm, warp=1.0)
    for k in ("Yt", "Yu", "Yv", "Yf"):
        g[k] = g[k] - 250.
    for k in ("Xt", "Xu", "Xv", "Xf"):
        g[k] = g[k] + 150.
    ll = {p: orc.CartNPSkm2Geo1D(np.stack([g["Y" + p].ravel(), g["X" + p].ravel()], axis=1)) for p in "tufv"}
    tmask = g["tmask"].copy(); tmask[25:30, 40:46] = 0
    if Nj > 200:                                       # larger meshes: an island of proportionate size
        tmask[Nj // 3:Nj // 3 + Nj // 12, Ni // 2:Ni // 2 + Ni // 10] = 0
    mm = os.path.join(tmp, "mesh_mask_TEST4.nc")
    var = {"tmask": ('i1', ('t', 'z', 'y', 'x'), tmask[None, None], None),
           "e1t": ('f8', ('t', 'y', 'x'), np.full((1, Nj, Ni), dkm * 1000.), None),
           "e2t": ('f8', ('t', 'y', 'x'), np.full((1, Nj, Ni), dkm * 1000.), None)}
    for p in "tufv":
        var["glam" + p] = ('f8', ('t', 'y', 'x'), ll[p][:, 1].reshape(1, Nj, Ni), None)
        var["gphi" + p] = ('f8', ('t', 'y', 'x'), ll[p][:, 0].reshape(1, Nj, Ni), None)
    _write_nc3(mm, {"t": 1, "z": 1, "y": Nj, "x": Ni}, var)
    base = 850608000
    u, v, sic = syn.make_fields(g, K=nrec, seed=77, umax=0.9, drift=0.3, ripple=0.1)
    sic[:, 10:16, 12:30] = 0.03
    if Nj > 200:                                       # ... and a polynya
        sic[:, Nj // 2:Nj // 2 + Nj // 10, Ni // 5:Ni // 5 + Ni // 6] = 0.03
    tc = (base + 1800 + 3600 * np.arange(nrec)).astype('i4')
    si3 = os.path.join(tmp, "TEST4-EXP01_1h_19961215_19961216_icemod.nc")
    _write_nc3(si3, {"time_counter": None, "y": Nj, "x": Ni},
               {"time_counter": ('i4', ('time_counter',), tc, {"units": ncio.tunits_default}),
                "siconc": ('f4', ('time_counter', 'y', 'x'), sic, None),
                "u_ice": ('f4', ('time_counter', 'y', 'x'), u, None),
                "v_ice": ('f4', ('time_counter', 'y', 'x'), v, None)})
    rng = np.random.default_rng(5)
    yx = np.stack([rng.uniform(g["Yt"].min() + 30, g["Yt"].max() - 30, nP), rng.uniform(g["Xt"].min() + 30, g["Xt"].max() - 30, nP)], axis=1)
    if ice_mask_seeding:
        jiN = syn.nearest_t_plane(g, yx)
        latT = ll["t"][:, 0].reshape(Nj, Ni)
        ok = (tmask[jiN[:, 0], jiN[:, 1]] == 1) & (latT[jiN[:, 0], jiN[:, 1]] >= 55.) & (sic[0][jiN[:, 0], jiN[:, 1]] >= 0.9)
        yx = yx[ok]
        nP = len(yx)
    sll = orc.CartNPSkm2Geo1D(yx)
    ids = (300534062025510 + 7 * np.arange(nP)).astype(np.int64)
    seed = os.path.join(tmp, "sitrack_seeding_nemoTsi3_19961215_00_HSS5.nc")
    sv = {"time": ('i4', ('time',), np.array([base], dtype='i4'), {"units": ncio.tunits_default}),
          "buoy": ('i4', ('buoy',), np.arange(nP, dtype='i4'), None),
          "id_buoy": (('f8', ('buoy',), ids.astype(np.float64), {"units": "ID of buoy"}) if fmt == "nc3" else
                      ('i8', ('buoy',), ids, {"units": "ID of buoy"})),
          "latitude": ('f4', ('time', 'buoy'), sll[None, :, 0].astype('f4'), None),
          "longitude": ('f4', ('time', 'buoy'), sll[None, :, 1].astype('f4'), None),
          "y_pos": ('f4', ('time', 'buoy'), yx[None, :, 0].astype('f4'), None),
          "x_pos": ('f4', ('time', 'buoy'), yx[None, :, 1].astype('f4'), None)}
    if two_d_time:
        tp = np.stack([np.full(nP, base), np.full(nP, tc[-1] + 1800)]).astype('i4')
        tp[0, ::7] = base + 4 * 3600                  # some buoys start later
        tp[1, ::5] = base + 9 * 3600                  # some stop earlier
        sv["time"] = ('i4', ('time',), np.array([base, tc[-1] + 1800], dtype='i4'), {"units": ncio.tunits_default})
        for k in ("latitude", "longitude", "y_pos", "x_pos"):
            sv[k] = (sv[k][0], sv[k][1], np.repeat(sv[k][2], 2, axis=0), None)
        sv["time_pos"] = ('i4', ('time', 'buoy'), tp, {"units": ncio.tunits_default})
    _write_nc3(seed, {"time": None, "buoy": nP}, sv)
    return dict(g=g, ll=ll, tmask=tmask, u=u, v=v, sic=sic, tc=tc, yx=yx, sll=sll, ids=ids, mm=mm, si3=si3, seed=seed, base=base, dkm=dkm)


def oracle_run(c, two_d_time, rdt=3600., nthreads=1):
    """Independent restatement of the whole driver with the CPU oracle (same file contents)."""
    from oracle import oracle as orc
    g = c["g"]
    Nj, Ni = g["Nj"], g["Ni"]
    grid = {}
    for p, k in (("f", "f"), ("u", "u"), ("v", "v"), ("t", "t")):
        lat = c["ll"][p][:, 0]; lon = np.mod(c["ll"][p][:, 1], 360.)
        yx = orc.Geo2CartNPSkm1D(np.stack([lat, lon], axis=1))
        grid["Y" + k] = np.ascontiguousarray(yx[:, 0].reshape(Nj, Ni)); grid["X" + k] = np.ascontiguousarray(yx[:, 1].reshape(Nj, Ni))
    grid["tmask"] = c["tmask"]
    latT = c["ll"]["t"][:, 0].reshape(Nj, Ni); lonT = np.mod(c["ll"]["t"][:, 1], 360.).reshape(Nj, Ni)
    pSG = np.stack([c["sll"][:, 0].astype('f4').astype('f8'),
                    np.mod(c["sll"][:, 1].astype('f4'), np.float32(360.)).astype('f8')], axis=1)      # reference ncio.py:303: mod on the f4
    pSC = c["yx"].astype('f4').astype('f8')
    res = np.full((Nj, Ni), np.sqrt(2.) * c.get("dkm", 10.0))
    tc = c["tc"]
    kstrt, Nt = 0, len(tc)
    nP, oSG, oSC, oIDs, ojiT, overt, keep = orc.SeedInit(c["ids"], pSG, pSC, np.ascontiguousarray(latT), np.ascontiguousarray(lonT),
                                                          grid["Yf"], grid["Xf"], res, c["tmask"], c["sic"][kstrt].astype('f8'),
                                                          nthreads=nthreads)
    z1 = np.zeros(nP, dtype=int) + kstrt; zL = np.zeros(nP, dtype=int) + (kstrt + Nt - 1)
    if two_d_time:
        base = c["base"]
        tp0 = np.full(len(c["ids"]), base); tp0[::7] = base + 4 * 3600
        tp1 = np.full(len(c["ids"]), tc[-1] + 1800); tp1[::5] = base + 9 * 3600
        z1, zL = drv.record_windows(np.stack([tp0, tp1]), tc, kstrt, kstrt + Nt - 1, tc[0], tc[-1], len(c["ids"]))
        z1, zL = z1[keep], zL[keep]
    trk = orc.Tracker(grid, oSC, ojiT, rec_first=z1, rec_last=zL, nthreads=nthreads)
    pos = np.zeros((Nt + 1, nP, 2)) + -9999.; msk = np.zeros((Nt + 1, nP), dtype='i1')
    pos[z1 - kstrt, np.arange(nP)] = oSC; msk[z1 - kstrt, np.arange(nP)] = 1
    for jt in range(Nt):
        pn, mn = trk.step(jt + kstrt, c["u"][jt].astype('f8'), c["v"][jt].astype('f8'), c["sic"][jt].astype('f8'))
        pos[jt + 1, mn == 1] = pn[mn == 1]; msk[jt + 1, mn == 1] = 1
    return dict(nP=nP, ids=oIDs, pos=pos, msk=msk, jiT=trk.jiT, alive=trk.alive, z1=z1, zL=zL, seedG=oSG)


@pytest.mark.gpu
@pytest.mark.parametrize("two_d_time", [False, True])
def test_cli_end_to_end_vs_oracle(tmp_path, monkeypatch, two_d_time):
    from oracle import oracle as orc
    monkeypatch.chdir(tmp_path)
    c = make_case(str(tmp_path), two_d_time=two_d_time)
    argv = ["-i", c["si3"], "-m", c["mm"], "-s", c["seed"], "-N", "TEST4"] + ([] if two_d_time else ["-F"])
    out = drv.main(argv)
    ref = oracle_run(c, two_d_time)
    assert out["nP"] == ref["nP"] and np.array_equal(out["IDs"], ref["ids"])
    assert np.array_equal(out["vJIt"], ref["jiT"]) and np.array_equal(out["iAlive"], ref["alive"])
    assert os.path.exists("./seed/Initialized_buoys_sitrack_seeding_nemoTsi3_19961215_00_HSS5_TEST4.npz")
    with np.load("./seed/Initialized_buoys_sitrack_seeding_nemoTsi3_19961215_00_HSS5_TEST4.npz") as z:
        assert sorted(z.files) == sorted(['nP', 'xPosG0', 'xPosC0', 'IDs', 'vJIt', 'VRTCS', 'idxKeep'])
    Nt = len(c["tc"])
    if not two_d_time:
        f_full, f_12 = out["files"]
        assert f_full == './nc/NEMO-SI3_TEST4_EXP01_tracking_nemoTsi3_idlSeed_19961215h00_19961215h14.nc'
        assert f_12 == './nc/NEMO-SI3_TEST4_EXP01_tracking12_nemoTsi3_idlSeed_19961215h00_19961215h14.nc'
        t, ids, llo, yxo, mko = ncio.LoadNCdata(f_full, krec=-1, lmask=True)
        assert np.array_equal(ids, ref["ids"]) and t.shape == (Nt + 1,) and t[0] == c["base"] and t[-1] == c["base"] + Nt * 3600
        assert np.array_equal(mko, ref["msk"])
        assert np.array_equal(yxo.astype('f4'), ref["pos"].astype('f4'))         # y_pos/x_pos are stored as f4
        # lat/lon of records >= 1: inverse projection (1e-5 relative is the bar; device libm vs glibc ~1e-12)
        want = orc.CartNPSkm2Geo1D(ref["pos"][1:].reshape(-1, 2)).reshape(Nt, -1, 2)
        got = llo[1:].copy(); want[..., 1] = np.mod(want[..., 1], 360.)           # LoadNCdata returns lon in [0,360)
        assert np.allclose(got, want.astype('f4').astype('f8'), rtol=1e-6, atol=1e-4)
        # record 0 keeps the seeding file's own coordinates
        assert np.allclose(llo[0], ref["seedG"], rtol=0, atol=1e-4)
        t2, _, _, yx2, mk2 = ncio.LoadNCdata(f_12, krec=-1, lmask=True)
        assert np.array_equal(yx2[0].astype('f4'), ref["pos"][0].astype('f4')) and np.array_equal(yx2[1].astype('f4'), ref["pos"][Nt].astype('f4'))
        assert np.array_equal(mk2[1], ref["msk"][Nt])
    else:
        (f_12,) = out["files"]
        assert '_tracking12_nemoTsi3_idlSeed_' in f_12
        t2, _, _, yx2, mk2, tp2 = ncio.LoadNCdata(f_12, krec=-1, lmask=True, lGetTimePos=True)
        nP = ref["nP"]
        kN = ref["zL"] + 1
        k0 = ref["z1"]
        assert np.array_equal(yx2[0].astype('f4'), ref["pos"][k0, np.arange(nP)].astype('f4'))
        assert np.array_equal(yx2[1].astype('f4'), ref["pos"][kN, np.arange(nP)].astype('f4'))
        assert np.array_equal(mk2[1], ref["msk"][kN, np.arange(nP)])
        want_t1 = np.where(ref["msk"][kN, np.arange(nP)] == 1, c["tc"][ref["zL"]] - 1800 + 3600, -9999)
        assert np.array_equal(tp2[1], want_t1) and np.array_equal(tp2[0], c["tc"][k0] - 1800)
    # second run hits the seed cache and reproduces the same result
    out2 = drv.main(argv)
    assert np.array_equal(out2["vJIt"], out["vJIt"]) and np.array_equal(out2["iAlive"], out["iAlive"])


@pytest.mark.parametrize("backend", ["hdf5", "netcdf3"])
def test_stream_writer_equals_the_whole_array_writer(tmp_path, monkeypatch, backend):
    """`ncio.CloudBuoysStream` (records appended one at a time: what the driver's `-F` does) against `ncSaveCloudBuoys` (whole arrays:
    what the reference does, ncio.py:131-197): the same file, variable for variable -- through libhdf5 (NetCDF-4, incl. rows wide enough
    for the thread-pool chunk path) and through the NetCDF-3 fall-back writer."""
    from sitrack_amd import h5lite
    if backend == "netcdf3":
        monkeypatch.setattr(h5lite, "writer_available", lambda: False)
    elif not h5lite.writer_available():
        pytest.skip("no libhdf5 here")
    Nt, Nb = 6, 70_000 if backend == "hdf5" else 3_000
    rng = np.random.default_rng(1)
    t = np.arange(Nt) * 3600 + 850608000
    ids = np.arange(Nb) + 300534062025510 if backend == "hdf5" else np.arange(Nb) + 900120
    Y, X = rng.normal(size=(Nt, Nb)) * 1000, rng.normal(size=(Nt, Nb)) * 1000
    La, Lo = rng.uniform(60, 90, (Nt, Nb)), rng.uniform(-180, 180, (Nt, Nb))
    M = (rng.random((Nt, Nb)) > 0.1).astype('i1')
    fa, fb = str(tmp_path / "whole.nc"), str(tmp_path / "stream.nc")
    ncio.ncSaveCloudBuoys(fa, t, ids, Y, X, La, Lo, mask=M, corigin='X')
    st = ncio.CloudBuoysStream(fb, t, ids, corigin='X', flush_bytes=1 << 20)
    with pytest.raises(IndexError):
        st.put(Nt, Y[0], X[0], La[0], Lo[0], M[0])
    for k in range(Nt):
        st.put(k, Y[k], X[k], La[k], Lo[k], M[k])
    st.close()
    assert open(fa, 'rb').read(4) == open(fb, 'rb').read(4) == (b'\x89HDF' if backend == "hdf5" else b'CDF\x02')
    ra, rb = ncio.LoadNCdata(fa, krec=-1, lmask=True), ncio.LoadNCdata(fb, krec=-1, lmask=True)
    assert all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(ra, rb))
    if backend == "hdf5":
        assert os.path.getsize(fa) == os.path.getsize(fb)
    # a stream that is closed before every record was written does not leave a half-written file behind
    st = ncio.CloudBuoysStream(str(tmp_path / "short.nc"), t, ids)
    st.put(0, Y[0], X[0], La[0], Lo[0], M[0])
    with pytest.raises(ValueError):
        st.close()
    assert not os.path.exists(str(tmp_path / "short.nc"))


@pytest.mark.gpu
def test_cli_F_streams_the_series_in_bounded_memory_and_out_stride(tmp_path, monkeypatch):
    """`-F` at a size where the reference's way would not fit a laptop: >= 1e6 buoys x 200 hourly records.  The reference holds the
    whole series twice as (Nt+1, nP, 2) f8 arrays (si3_part_tracker.py:324-330: 6.7 GB here, 0.3-1 TB at 1e7 buoys x thousands of
    records) and writes it at the end (:515-519); the driver appends every record to the file as it is fetched.  Checked:
    (1) the command line's peak resident memory stays below 2 GB (own process, measured by itself); (2) the file equals, variable for
    variable, the one the in-memory writer (ncSaveCloudBuoys on whole arrays) makes of an independent run of the same tracking;
    (3) `--out-stride 8`: the records in between go through fused launches and the file holds every 8th record of the `-F` file."""
    import json
    import subprocess
    import sys
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("SITRK_NC_COMPLEVEL", "1")       # (three 5-GB series are deflated here: the reference's level 9 is tested elsewhere)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    Nrec = 200
    c = make_case(str(tmp_path), nrec=Nrec, nP=1_150_000)
    launcher = ("import json, sys; sys.path.insert(0, %r); from sitrack_amd import driver as drv; "
                "out = drv.main(sys.argv[1:]); "
                "print('RESULT ' + json.dumps({'files': out['files'], 'nP': int(out['nP']), 'launches': out['launches'], "
                "'maxrss_kb': int([l.split()[1] for l in open('/proc/self/status') if l.startswith('VmHWM')][0])}))" % root)
    # (VmHWM, not ru_maxrss: the latter survives fork + exec, i.e. it starts at whatever the pytest process had resident)
    res = {}
    for tag, extra in (("full", []), ("stride", ["--out-stride", "8"])):
        r = subprocess.run([sys.executable, "-c", launcher, "-i", c["si3"], "-m", c["mm"], "-s", c["seed"], "-N", "TEST4", "-F"] + extra,
                           capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        res[tag] = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    nP = res["full"]["nP"]
    assert nP >= 1_000_000
    series_bytes = 2 * (Nrec + 1) * nP * 2 * 8
    assert series_bytes > 6e9
    for tag in res:
        assert res[tag]["maxrss_kb"] < 2 * 1024 * 1024, (tag, res[tag]["maxrss_kb"])          # (1)
    assert res["full"]["launches"]["step_launches"] == Nrec                                 # every record fetched: batches of one
    assert res["stride"]["launches"]["fused_launches"] == Nrec // 8 and res["stride"]["launches"]["fused_records"] == Nrec
    f_full, f_12 = res["full"]["files"]
    assert res["stride"]["files"][0] == f_full.replace(".nc", "_stride8.nc") and res["stride"]["files"][1] == f_12
    # (2) an independent run of the same tracking, the series held in memory like the reference does, written at once
    with np.load("./seed/Initialized_buoys_sitrack_seeding_nemoTsi3_19961215_00_HSS5_TEST4.npz") as z:
        xPosG0, xPosC0, IDs, vJIt = z["xPosG0"], z["xPosC0"], z["IDs"], z["vJIt"]
    imaskt, _, _, _, _, xYf, xXf, _ = ncio.GetModelGrid(c["mm"])
    xYv, xXv, xYu, xXu = ncio.GetModelUVGrid(c["mm"])
    trk = sit.IceTracker(xYf, xXf, xYu, xXu, xYv, xXv, imaskt, nslots=2)
    trk.set_buoys(xPosC0, vJIt)
    xPosC = np.zeros((Nrec + 1, nP, 2)) + drv.FILL; xPosG = np.zeros((Nrec + 1, nP, 2)) + drv.FILL
    xmask = np.zeros((Nrec + 1, nP), dtype='i1')
    xPosC[0], xPosG[0], xmask[0] = xPosC0, xPosG0, 1
    for jt in range(Nrec):
        trk.load_record(jt % 2, c["u"][jt], c["v"][jt], c["sic"][jt])
        trk.step(jt, jt % 2)
        pos, msk = trk.record(jt)
        xPosC[jt + 1, msk == 1] = pos[msk == 1]; xmask[jt + 1, msk == 1] = 1
        xPosG[jt + 1] = trk.ctx.cart2geo(xPosC[jt + 1])                               # :493
    trk.close()
    assert 0 < (xmask[Nrec] == 0).sum() < nP
    vTime = c["base"] + 3600 * np.arange(Nrec + 1)
    ncio.ncSaveCloudBuoys("./nc/in_memory.nc", vTime, IDs, xPosC[:, :, 0], xPosC[:, :, 1], xPosG[:, :, 0], xPosG[:, :, 1], mask=xmask,
                          corigin='NEMO-SI3_TEST4_EXP01')
    from sitrack_amd import h5lite
    for f, rows in ((f_full, slice(None)), (res["stride"]["files"][0], slice(None, None, 8))):
        a, b = h5lite.H5File("./nc/in_memory.nc"), h5lite.H5File(f)
        for name in ("time", "buoy", "id_buoy", "latitude", "longitude", "y_pos", "x_pos", "mask"):
            va, vb = a.read(name), b.read(name)
            want = va if name in ("buoy", "id_buoy") else va[rows]
            assert want.dtype == vb.dtype and np.array_equal(want, vb), (f, name)
            if name in ("time", "latitude"):
                assert a.attr(name, "units") == b.attr(name, "units")
        a.close(); b.close()
    assert (Nrec // 8 + 1) == h5lite.H5File(res["stride"]["files"][0]).shape("time")[0]


@pytest.mark.gpu
def test_seeding_tool_feeds_the_tracker(tmp_path, monkeypatch):
    """tools/generate_idealized_seeding.py -> si3_part_tracker.py, reference README.md:36-97 workflow."""
    import importlib.util
    monkeypatch.chdir(tmp_path)
    c = make_case(str(tmp_path))
    spec = importlib.util.spec_from_file_location("gis", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools",
                                                                      "generate_idealized_seeding.py"))
    gis = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gis)
    f = gis.main(["-d", "1996-12-15_00:00:00", "-m", c["mm"], "-i", c["si3"], "-k", "0", "-S", "3", "-N", "TEST4"])
    assert f == './nc/sitrack_seeding_nemoTsi3_19961215_00_HSS3.nc'
    t, ids, ll, yx = ncio.LoadNCdata(f, krec=0)
    assert len(ids) > 50 and np.array_equal(ids, np.arange(1, len(ids) + 1)) and int(t) == c["base"]
    assert np.all(ll[:, 0] >= 55.)
    out = drv.main(["-i", c["si3"], "-m", c["mm"], "-s", f, "-N", "TEST4", "-F", "-e", "1996-12-15_10:00:00"])
    assert out["Nt"] == 10 and out["nP"] <= len(ids)
    assert out["files"][0] == './nc/NEMO-SI3_TEST4_EXP01_tracking_nemoTsi3_idlSeed_19961215h00_19961215h10.nc'


def _dist_worker(rank, world, port, tmp, argv, q):
    os.environ.update({"WORLD_SIZE": str(world), "RANK": str(rank), "LOCAL_RANK": str(rank), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port), "SITRK_DIST_BACKEND": "gloo", "SITRK_DEVICE": "0"})
    os.chdir(tmp)
    out = drv.main(argv)
    if rank == 0:
        q.put({k: out[k] for k in ("files", "nP", "IDs", "vJIt", "iAlive", "rebalances", "migrated")})


@pytest.mark.gpu
@pytest.mark.parametrize("two_d_time,extra", [(False, []), (True, []), (False, ["--full-records"]),
                                              (False, ["--rebalance", "3"]), (True, ["--rebalance", "4"])])
def test_cli_two_ranks_equals_one(tmp_path, monkeypatch, two_d_time, extra):
    """N>1 driver path rehearsed with 2 ranks on the one GPU of the box (gloo moves the record slabs; RCCL refuses
    two ranks per device): buoy-range partition, per-range SeedInit, record delivery, gathers -> same files.
    `--rebalance R`: every R records the buoys are re-partitioned by their current host row and their states migrate
    between the ranks (the fast field of the test case carries buoys across the band boundary, and some die) -- the
    files must still be identical to the single-rank run's, byte for byte in every variable."""
    import socket
    import torch.multiprocessing as mp
    d1, d2 = tmp_path / "one", tmp_path / "two"
    d1.mkdir(); d2.mkdir()
    c = make_case(str(tmp_path), two_d_time=two_d_time)
    argv = ["-i", c["si3"], "-m", c["mm"], "-s", c["seed"], "-N", "TEST4"] + ([] if two_d_time else ["-F"])
    monkeypatch.chdir(d1)
    one = drv.main(argv + ["--full-records"])          # the single-rank run ingests whole records
    argv = argv + extra                                # the 2-rank run: row bands by default
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, str(d2), argv, q)) for r in range(2)]
    for p in procs:
        p.start()
    two = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert two["nP"] == one["nP"] and np.array_equal(two["IDs"], one["IDs"])
    assert np.array_equal(two["vJIt"], one["vJIt"]) and np.array_equal(two["iAlive"], one["iAlive"])
    assert two["files"] == one["files"]
    if "--rebalance" in extra:
        assert two["rebalances"] >= 1 and two["migrated"] > 0, (two["rebalances"], two["migrated"])      # buoys did change owner
    for f in one["files"]:
        a = ncio.LoadNCdata(str(d1 / f), krec=-1, lmask=True)
        b = ncio.LoadNCdata(str(d2 / f), krec=-1, lmask=True)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("two_d_time,extra", [(False, ["--full-records"]), (False, ["--rebalance", "3"]), (True, ["--rebalance", "4"]), (True, [])])
def test_cli_one_rccl_rank_through_the_multi_rank_paths(tmp_path, monkeypatch, two_d_time, extra):
    """The `nccl` branches of the driver -- RCCL tensor all-gathers of the per-range SeedInit results, the seed cache by tensor
    broadcast, `RecordBroadcaster` (in-place slab broadcast on a communication stream), tensor gathers at output records, and the
    re-balancing all-to-all (`all_to_all_single`) with `sitrk_restore_state` -- rehearsed with ONE RCCL rank (SITRK_FORCE_DIST=1:
    a process group of one; RCCL refuses two ranks per device and the box has one GPU): the files must equal the plain
    single-process run's in every variable.  With more than one RCCL rank these paths are unrun."""
    import subprocess
    import sys
    d1, d2 = tmp_path / "one", tmp_path / "forced"
    d1.mkdir(); d2.mkdir()
    c = make_case(str(tmp_path), two_d_time=two_d_time)
    argv = ["-i", c["si3"], "-m", c["mm"], "-s", c["seed"], "-N", "TEST4"] + ([] if two_d_time else ["-F"])
    monkeypatch.chdir(d1)
    one = drv.main(argv + ["--full-records"])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = __import__("socket").socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SITRK_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONPATH=root)
    r = subprocess.run([sys.executable, os.path.join(root, "si3_part_tracker.py")] + argv + extra, cwd=str(d2), env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "1 ranks (nccl)" in r.stdout or "ranks (nccl)" in r.stdout, r.stdout[:600]
    for f in one["files"]:
        a = ncio.LoadNCdata(str(d1 / f), krec=-1, lmask=True)
        b = ncio.LoadNCdata(str(d2 / f), krec=-1, lmask=True)
        for x, y in zip(a, b):
            assert np.array_equal(x, y), f


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--rebalance", "3"], ["--full-records"]])
def test_cli_rank_without_buoys(tmp_path, monkeypatch, extra):
    """Three ranks, two seeds: the last rank owns no buoy from the start (and with --rebalance ranks lose and gain all
    their buoys as the two drift) -- the empty ranges go through SeedInit, stepping, gathers and the writers, and the
    files equal the single-rank run's."""
    import socket
    import torch.multiprocessing as mp
    d1, d2 = tmp_path / "one", tmp_path / "three"
    d1.mkdir(); d2.mkdir()
    c = make_case(str(tmp_path), nP=2)
    argv = ["-i", c["si3"], "-m", c["mm"], "-s", c["seed"], "-N", "TEST4", "-F"]
    monkeypatch.chdir(d1)
    one = drv.main(argv + ["--full-records"])
    assert one["nP"] == 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dist_worker, args=(r, 3, port, str(d2), argv + extra, q)) for r in range(3)]
    for p in procs:
        p.start()
    three = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert three["nP"] == 2 and np.array_equal(three["IDs"], one["IDs"]) and np.array_equal(three["vJIt"], one["vJIt"])
    assert three["files"] == one["files"]
    for f in one["files"]:
        a = ncio.LoadNCdata(str(d1 / f), krec=-1, lmask=True)
        b = ncio.LoadNCdata(str(d2 / f), krec=-1, lmask=True)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


@pytest.mark.gpu
def test_sidfex_seeding_reproduces_the_reference_fixture(tmp_path, monkeypatch, golden):
    """The reference's committed seeding file tools/nc/sitrack_seeding_sidfex_19961215_00_HSS5.nc__KEEP was made by
    `generate_sidfex_seeding.py -d 1996-12-15_00:00:00 --lsidfex 1 -k 0 -S 5` (tools/cmd.sh) from tools/sidfexloc.dat.
    Same command through this build (projection on the GPU, own NetCDF writer) -> same name, same numbers."""
    import importlib.util
    g = golden("g7_projection.npz")
    monkeypatch.chdir(tmp_path)
    with open("sidfexloc.dat", "w") as f:
        for i, (lon, lat) in zip(g["dat_id"], g["dat_lonlat"]):
            f.write("%d %r %r\n" % (i, float(lon), float(lat)))
    spec = importlib.util.spec_from_file_location("gis2", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools",
                                                                       "generate_idealized_seeding.py"))
    gis = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gis)
    fout = gis.main(["-d", "1996-12-15_00:00:00", "--lsidfex", "1", "-k", "0", "-S", "5"])
    assert fout == "./nc/sitrack_seeding_sidfex_19961215_00_HSS5.nc"
    with ncio._Reader(fout) as f:
        assert np.array_equal(np.asarray(f.var("id_buoy")).astype(np.int64), g["id_buoy"])
        assert np.array_equal(np.asarray(f.var("time")).astype(np.int64), g["time"])
        for name, key in (("latitude", "latitude"), ("longitude", "longitude"), ("y_pos", "y_pos"), ("x_pos", "x_pos")):
            got = np.asarray(f.var(name)).astype(np.float32)[0]
            assert np.array_equal(got, g[key]), name          # float32, bit for bit
        assert f.attr("time", "units") == ncio.tunits_default and f.attr("y_pos", "units") == "km"


def test_reference_seeding_file_is_read_without_netcdf4(golden):
    """The reference's committed seeding file (tools/nc/sitrack_seeding_sidfex_19961215_00_HSS5.nc__KEEP, a NetCDF-4 =
    HDF5 file with deflate + shuffle; copied here as a data fixture) through this build's readers: netCDF4 is absent,
    the system's libhdf5 does it (sitrack_amd/h5lite.py).  Values = golden set G7 (decoded independently with h5dump)."""
    from sitrack_amd import h5lite
    if ncio.backend() != "netCDF4" and not h5lite.available():
        pytest.skip("neither netCDF4 nor a loadable libhdf5 here")
    f = os.path.join(os.path.dirname(__file__), "golden", "sitrack_seeding_sidfex_19961215_00_HSS5.nc")
    g = golden("g7_projection.npz")
    idate0, idateN, seed_name, seed_type, _ = ncio.SeedFileTimeInfo(f)
    assert (idate0, idateN, seed_type) == (850608000, 850608000, 'sidfex') and seed_name == 'sitrack_seeding_sidfex_19961215_00_HSS5'
    zt, ids, zg, zc = ncio.LoadNCdata(f, krec=0)
    assert int(zt) == int(g["time"][0]) and np.array_equal(ids, g["id_buoy"]) and ids.dtype == np.int64
    assert np.array_equal(zc[:, 0].astype('f4'), g["y_pos"]) and np.array_equal(zc[:, 1].astype('f4'), g["x_pos"])
    assert np.array_equal(zg[:, 0].astype('f4'), g["latitude"])
    assert np.array_equal(zg[:, 1], np.mod(g["longitude"], np.float32(360.)).astype('f8'))      # ncio.py:303, in the file's f4
    vt = ncio.LoadNCtime(f)
    assert list(np.atleast_1d(vt[-1] if isinstance(vt, tuple) else vt)) == [850608000]
    if ncio.backend() != "netCDF4":
        h = h5lite.H5File(f)
        assert h.shape("latitude") == (1, 10) and h.read("latitude", (0, slice(2, 5))).tolist() == g["latitude"][2:5].tolist()
        assert h.attr("time", "units") == ncio.tunits_default and float(h.attr("x_pos", "_FillValue")) == -9999.
        assert np.array_equal(h.read("x_pos", (Ellipsis,)), h.read("x_pos")) and h.read("id_buoy", -1) == g["id_buoy"][-1]
        with pytest.raises(KeyError):
            h.read("no_such_variable")
        with pytest.raises(IndexError):
            h.read("latitude", 3)
        h.close()


@pytest.mark.gpu
def test_cli_with_the_reference_seeding_file(tmp_path, monkeypatch):
    """`-s` = the reference's own seeding file (NetCDF-4), mesh and SI3 records synthetic (NetCDF-3): whole command
    against the oracle-driven restatement of the driver."""
    from sitrack_amd import h5lite
    if ncio.backend() != "netCDF4" and not h5lite.available():
        pytest.skip("neither netCDF4 nor a loadable libhdf5 here")
    monkeypatch.chdir(tmp_path)
    c = make_case(str(tmp_path), nrec=10, Nj=96, Ni=96, dkm=50.0)
    f = os.path.join(os.path.dirname(__file__), "golden", "sitrack_seeding_sidfex_19961215_00_HSS5.nc")
    _, ids, zg, zc = ncio.LoadNCdata(f, krec=0)
    c["ids"], c["sll"], c["yx"] = ids, zg, zc              # what the driver will read from the file (f4 values)
    out = drv.main(["-i", c["si3"], "-m", c["mm"], "-s", f, "-N", "TEST4", "-F"])
    ref = oracle_run(c, False)
    assert out["nP"] == ref["nP"] >= 5 and np.array_equal(out["IDs"], ref["ids"])
    assert np.array_equal(out["vJIt"], ref["jiT"]) and np.array_equal(out["iAlive"], ref["alive"])
    _, ids2, _, yxo, mko = ncio.LoadNCdata(out["files"][0], krec=-1, lmask=True)
    assert np.array_equal(ids2, ref["ids"]) and np.array_equal(mko, ref["msk"])
    assert np.array_equal(yxo.astype('f4'), ref["pos"].astype('f4'))
    assert '_tracking_sidfex_' in out["files"][0] or 'sidfex' in out["files"][0]


def test_packed_variables_are_unpacked_like_netcdf4_would(tmp_path):
    """scale_factor / add_offset: netCDF4 (what the reference reads with) applies them on access; the fall-back readers
    do the same."""
    fn = str(tmp_path / "packed.nc")
    raw = np.arange(12, dtype='i2').reshape(3, 4)
    _write_nc3(fn, {"y": 3, "x": 4}, {"packed": ('i2', ('y', 'x'), raw, {"scale_factor": 0.5, "add_offset": 10.0}),
                                        "plain": ('i2', ('y', 'x'), raw, None)})
    with ncio._Reader(fn) as f:
        assert np.array_equal(f.var("packed"), raw * 0.5 + 10.0) and np.array_equal(f.var("packed", 1), raw[1] * 0.5 + 10.0)
        assert np.array_equal(f.var("plain"), raw) and f.var("plain").dtype.kind == 'i'


@pytest.mark.gpu
@pytest.mark.parametrize("two_d_time", [False, True])
def test_cli_on_hdf5_inputs(tmp_path, monkeypatch, two_d_time):
    """mesh_mask, SI3 records and seeding file as HDF5 files laid out like NetCDF-4 (chunked, shuffle + deflate,
    unlimited time axis, int64 ids), no netCDF4 package: read through libhdf5 (whole variables, single records and the
    row bands of records as hyperslabs).  Same results as the oracle-driven driver, and as the same case from NetCDF-3."""
    from sitrack_amd import h5lite
    if ncio.backend() == "netCDF4" or not h5lite.available():
        pytest.skip("exercises the libhdf5 reader (needs libhdf5 and no netCDF4)")
    (tmp_path / "h5").mkdir(); (tmp_path / "n3").mkdir()
    c = make_case(str(tmp_path / "h5"), two_d_time=two_d_time, fmt="hdf5")
    assert h5lite.is_hdf5(c["si3"]) and h5lite.is_hdf5(c["mm"]) and h5lite.is_hdf5(c["seed"])
    argv = lambda cc: ["-i", cc["si3"], "-m", cc["mm"], "-s", cc["seed"], "-N", "TEST4"] + ([] if two_d_time else ["-F"])   # noqa: E731
    monkeypatch.chdir(tmp_path / "h5")
    out = drv.main(argv(c))
    ref = oracle_run(c, two_d_time)
    assert out["nP"] == ref["nP"] and np.array_equal(out["IDs"], ref["ids"])
    assert np.array_equal(out["vJIt"], ref["jiT"]) and np.array_equal(out["iAlive"], ref["alive"])
    c3 = make_case(str(tmp_path / "n3"), two_d_time=two_d_time)
    monkeypatch.chdir(tmp_path / "n3")
    out3 = drv.main(argv(c3))
    assert out3["files"] == out["files"]
    for f in out["files"]:
        a = ncio.LoadNCdata(str(tmp_path / "h5" / f), krec=-1, lmask=True)
        b = ncio.LoadNCdata(str(tmp_path / "n3" / f), krec=-1, lmask=True)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_hdf5_reader_on_netcdf4_like_layout(tmp_path):
    """chunked + shuffle + deflate datasets with an unlimited leading dimension, NUL-terminated fixed-length string
    attributes, dimension-only datasets: whole reads, single records, row bands (hyperslabs), negative indices."""
    from sitrack_amd import h5lite
    if not h5lite.available():
        pytest.skip("no loadable libhdf5 here")
    import h5write
    rng = np.random.default_rng(0)
    u = rng.standard_normal((6, 33, 41)).astype('f4')
    tc = (850608000 + 1800 + 3600 * np.arange(6)).astype('f8')
    fn = str(tmp_path / "TEST4-EXP01_1h_19961215_19961216_icemod.nc")      # ModelFileTimeInfo parses the NAME
    h5write.write_h5(fn, {"time_counter": (tc, {"units": ncio.tunits_default}, True),
                          "u_ice": (u, {"units": "m/s", "_FillValue": np.float32(1e20)}, True),
                          "tmask": (np.ones((1, 1, 33, 41), 'i1'), None, False)}, dims={"time_counter": 6, "y": 33, "x": 41})
    assert h5lite.is_hdf5(fn)
    if ncio.backend() == "netCDF4":
        pytest.skip("netCDF4 is installed: the libhdf5 path is not the one ncio takes")
    with ncio._Reader(fn) as r:
        assert r.has_dim("time_counter") and r.has_dim("y") and not r.has_var("y") and r.has_var("u_ice") and not r.has_var("nope")
        assert r.dim("time_counter") == 6 and r.dim("x") == 41
        assert r.attr("time_counter", "units") == ncio.tunits_default and r.attr("u_ice", "units") == "m/s"
        assert np.float32(r.attr("u_ice", "_FillValue")) == np.float32(1e20)
        assert np.array_equal(r.var("u_ice"), u) and np.array_equal(r.var("u_ice", 3), u[3]) and np.array_equal(r.var("u_ice", -1), u[-1])
        assert np.array_equal(r.var("u_ice", (2, slice(5, 9))), u[2, 5:9]) and np.array_equal(r.var("u_ice", (slice(1, 4),)), u[1:4])
        assert np.array_equal(r.var("u_ice", (4, slice(0, 0))), u[4, 0:0]) and r.var("tmask", (0, 0)).shape == (33, 41)
        assert np.array_equal(r.var("u_ice", np.array([0, 5])), u[[0, 5]])            # fancy index: read all, then numpy
    nrec, vt = ncio.ModelFileTimeInfo(fn)[:2]
    assert nrec == 6 and np.array_equal(vt, tc.astype('i4'))


@pytest.mark.parametrize("fmt", ["nc3", "h5"])
def test_model_records_box_reads(tmp_path, fmt):
    """`ModelRecords.fields_box_into` -- the hyperslab of the reference's whole-record reads (si3_part_tracker.py:372-374) that the
    driver reads straight into the library's staging: equal to slicing the whole record, for both file layouts, into buffers of the
    file's own precision and of double precision; values that do not survive the cast to a narrower buffer are refused."""
    from sitrack_amd import h5lite
    if fmt == "h5" and (not h5lite.available() or ncio.backend() == "netCDF4"):
        pytest.skip("the libhdf5 path is not the one ncio takes here")
    rng = np.random.default_rng(3)
    nrec, Nj, Ni = 4, 23, 37
    flds = {n: rng.standard_normal((nrec, Nj, Ni)).astype('f4') for n in ("u_ice", "v_ice", "siconc")}
    wide = rng.standard_normal((nrec, Nj, Ni))                                   # float64 values with no float32 twin
    tc = (850608000 + 1800 + 3600 * np.arange(nrec)).astype('f8')
    fn = str(tmp_path / "TEST4-EXP01_1h_19961215_19961216_icemod.nc")
    variables = {"time_counter": ('f8', ('time_counter',), tc, {"units": ncio.tunits_default})}
    for n, a in flds.items():
        variables[n] = ('f4', ('time_counter', 'y', 'x'), a, None)
    variables["wide"] = ('f8', ('time_counter', 'y', 'x'), wide, None)
    variables["narrowable"] = ('f8', ('time_counter', 'y', 'x'), flds["u_ice"].astype('f8'), None)
    (_write_nc3 if fmt == "nc3" else _write_h5_like_nc3)(fn, {"time_counter": None, "y": Nj, "x": Ni}, variables)
    rec = ncio.ModelRecords(fn)
    try:
        assert rec.time(2) == int(tc[2])
        for (j0, j1, i0, i1) in ((0, Nj, 0, Ni), (5, 17, 8, 31), (22, 23, 36, 37), (3, 4, 0, Ni)):
            for dt in ('f4', 'f8'):
                outs = [np.full((j1 - j0, i1 - i0), np.nan, dtype=dt) for _ in range(3)]
                rec.fields_box_into(2, j0, j1, i0, i1, outs)
                for o, n in zip(outs, ("u_ice", "v_ice", "siconc")):
                    assert np.array_equal(o, flds[n][2, j0:j1, i0:i1])
        outs = [np.empty((6, Ni), dtype='f4') for _ in range(3)]
        rec.fields_rows_into(1, 9, 15, outs)                                      # whole rows
        assert all(np.array_equal(o, flds[n][1, 9:15]) for o, n in zip(outs, ("u_ice", "v_ice", "siconc")))
        o4 = [np.empty((4, 5), dtype='f4')]
        rec.fields_box_into(0, 1, 5, 2, 7, o4, names=("narrowable",))             # float64 file values that ARE float32 values
        assert np.array_equal(o4[0], flds["u_ice"][0, 1:5, 2:7])
        with pytest.raises(ValueError, match="not exactly representable"):
            rec.fields_box_into(0, 1, 5, 2, 7, o4, names=("wide",))
    finally:
        rec.close()
