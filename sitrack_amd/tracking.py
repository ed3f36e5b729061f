"""Host-side mirror of the reference's `sit.*` functions on the hot path.

Same names, argument order and return shapes/dtypes as the reference
(`sitrack/tracking.py`, `sitrack/util.py`), so a driver written against the
reference cannot tell; the arithmetic runs in libsitrk.so on the GPU through
the C ABI of include/sitrk.h.  There is no CPU fallback.
"""
import numpy as np

from . import _lib

FillValue = _lib.FillValue     # sitrack/ncio.py:19
rmin_conc = 0.1                # sitrack/tracking.py:4
rFoundKM = 2.5                 # sitrack/tracking.py:5

_default_ctx = None


def default_context(device=0):
    """Process-wide context (one process = one GPU)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = _lib.Context(device)
    return _default_ctx


def vertices_of(jiT):
    """VRTCS from vJIt: [[jT-1,jT-1,jT,jT],[iT-1,iT,iT,iT-1]] (reference locate.py:320-321;
    tracking.py:257-300 shifts both identically, so VRTCS is a pure function of vJIt)."""
    jiT = np.asarray(jiT, dtype=np.int64)
    j, i = jiT[:, 0], jiT[:, 1]
    v = np.empty((jiT.shape[0], 2, 4), dtype=np.int64)
    v[:, 0, 0] = j - 1; v[:, 0, 1] = j - 1; v[:, 0, 2] = j; v[:, 0, 3] = j
    v[:, 1, 0] = i - 1; v[:, 1, 1] = i; v[:, 1, 2] = i; v[:, 1, 3] = i - 1
    return v


def GetTimeSpan(dt, vtime_mod, iSdA, iMdA, iMdB, iStop=None, iverbose=0):
    """First and last model record of a run (reference sitrack/tracking.py:8-37, same arguments and 5-tuple; raises where
    the reference prints and exits).  `vtime_mod` holds the record centres, the seeding date `iSdA` must fall inside the
    model span shifted back by half a step; the run starts at the first record centred strictly after the seeding date
    and ends at the record nearest to `iStop`, or at the last one."""
    centres = np.asarray(vtime_mod)
    half = dt / 2
    if not (iMdA - half <= iSdA <= iMdB - half):
        raise ValueError("PROBLEM: time in the seeding file (%d) is outside of what model spans!" % iSdA)
    first = int(np.abs(centres - iSdA).argmin())
    if centres[first] <= iSdA:
        first += 1
    last = int(np.abs(centres - iStop).argmin()) if iStop else len(centres) - 1
    if iverbose > 0:
        print('    * [GetTimeSpan]: records %d..%d => %d model records' % (first, last, last - first + 1))
    return last - first + 1, first, last, centres[first], centres[last]


def SeedInit(pIDs, pSG, pSC, platT, plonT, pYf, pXf, pResolKM, maskT, xIceConc=[], iverbose=0, ctx=None):
    """Reference sitrack/tracking.py:98-178, same 7-tuple:
    (nP, pSG[iKeep], pSC[iKeep], pIDs[iKeep], zjiT[iKeep] (nP,2) int, zJIvrt[iKeep] (nP,2,4) int, iKeep)."""
    pSG = np.asarray(pSG)
    pSC = np.asarray(pSC)
    (nP, n2) = np.shape(pSG)
    if np.shape(pSC) != (nP, n2):
        raise ValueError('ERROR [SeedInit]: shape disagreement for `pSG` and `pSC`!')
    if n2 != 2:
        raise ValueError('ERROR [SeedInit]: wrong shape for `pSG` and `pSC`!')
    if len(np.shape(xIceConc)) != 2:
        # the reference raises UnboundLocalError in Survive() here (tracking.py:86-89)
        raise ValueError('SeedInit: `xIceConc` must be the 2-D ice concentration at the seeding record')
    own = ctx is None
    if own:
        ctx = _lib.Context(default_context().device)
    try:
        if (ctx.Nj, ctx.Ni) != np.shape(pYf) or own:
            # only F-points and the mask are read by the locate kernels
            ctx.set_grid(pYf, pXf, pYf, pXf, pYf, pXf, maskT)
        jiT, keep, why = ctx.seed_init(pSG, pSC, platT, plonT, pResolKM, xIceConc)
    finally:
        if own:
            ctx.close()
    pIDs = np.asarray(pIDs)
    iKeep = np.arange(nP, dtype=int)
    nPn = int(np.sum(keep))
    if nPn < nP:
        (iKeep,) = np.where(keep == 1)
        if iverbose > 0:
            print(' * [SeedInit()]: ' + str(nP - nPn) + ' "to-be-seeded" buoys have to be canceled.')
        nP = nPn
    zjiT = jiT.astype(np.int64)
    zJIvrt = vertices_of(zjiT)
    return nP, pSG[iKeep, :], pSC[iKeep, :], pIDs[iKeep], zjiT[iKeep, :], zJIvrt[iKeep, :, :], iKeep


def FindContainingCell(pyx, kjiT, pYf=None, pXf=None, ctx=None):
    """Vectorised reference sitrack/locate.py:280-330: pyx (n,2), kjiT (n,2) ->
    (lPin (n) bool, jiT (n,2) int64, vertices (n,2,4) int64).  Needs a context whose grid is set."""
    ctx = ctx or default_context()
    found, ji = ctx.find_cells(pyx, kjiT)
    ji = ji.astype(np.int64)
    return found, ji, vertices_of(ji)


def CartNPSkm2Geo1D(pcoorC, lat0=70., lon0=-45., ctx=None):
    """Reference sitrack/util.py:413-429: (n,2) [y,x] km -> (n,2) [lat,lon] degrees."""
    (_, n2) = np.shape(pcoorC)
    if n2 != 2:
        raise ValueError(' ERROR [CartNPSkm2Geo1D()]: input array `pcoorC` has a wrong a shape!')
    return (ctx or default_context()).cart2geo(pcoorC, lat0, lon0)


def Geo2CartNPSkm1D(pcoorG, lat0=70., lon0=-45., ctx=None):
    """Reference sitrack/util.py:394-410: (n,2) [lat,lon] degrees -> (n,2) [y,x] km."""
    (_, n2) = np.shape(pcoorG)
    if n2 != 2:
        raise ValueError(' ERROR [Geo2CartNPSkm1D()]: input array `pcoorG` has a wrong a shape!')
    return (ctx or default_context()).geo2cart(pcoorG, lat0, lon0)


def ConvertGeo2CartesianNPSkm(plat, plon, lat0=70., lon0=-45., ctx=None):
    """reference util.py:434-451: geographic (lat, lon) [deg], any shape -> (Y, X) [km] of the WGS84 north polar
    stereographic plane (true-scale latitude lat0, central longitude lon0), same shape.  Projection on the device."""
    ctx = ctx or default_context()
    shp = np.shape(plat)
    ll = np.stack([np.asarray(plat, dtype=np.float64).ravel(), np.asarray(plon, dtype=np.float64).ravel()], axis=1)
    yx = ctx.geo2cart(ll, lat0, lon0)
    return np.ascontiguousarray(yx[:, 0].reshape(shp)), np.ascontiguousarray(yx[:, 1].reshape(shp))


def ConvertCartesianNPSkm2Geo(pY, pX, lat0=70., lon0=-45., ctx=None):
    """reference util.py:455-472: the inverse, (Y, X) [km] -> (lat, lon) [deg], same shape."""
    ctx = ctx or default_context()
    shp = np.shape(pX)
    yx = np.stack([np.asarray(pY, dtype=np.float64).ravel(), np.asarray(pX, dtype=np.float64).ravel()], axis=1)
    ll = ctx.cart2geo(yx, lat0, lon0)
    return np.ascontiguousarray(ll[:, 0].reshape(shp)), np.ascontiguousarray(ll[:, 1].reshape(shp))


class IceTracker:
    """The record loop body of the reference driver (si3_part_tracker.py:361-496) as an object.

    Holds the loop's state on the GPU.  Typical use, mirroring the driver:

        trk = IceTracker(xYf, xXf, xYu, xXu, xYv, xXv, imaskt, rdt=3600., iUVstrategy=1)
        trk.set_buoys(xPosC0, vJIt, z1stModelRec, zLstModelRec)
        for jt in range(Nt):
            jrec = jt + kstrt
            trk.load_record(0, xUu, xVv, xIC)          # :372-374
            trk.step(jrec, 0)                           # :378-490
            xPosC[jt+1], xmask[jt+1,:,0], xPosG[jt+1] = trk.record(jrec, latlon=True)   # :459-460,493
    """

    def __init__(self, xYf, xXf, xYu, xXu, xYv, xXv, imaskt, rdt=3600., iUVstrategy=1, rmin=rmin_conc,
                 nslots=1, field_dtype=np.float32, device=0, ctx=None):
        self.ctx = ctx or _lib.Context(device)
        self.ctx.set_grid(xYf, xXf, xYu, xXu, xYv, xXv, imaskt)
        self.ctx.set_params(rdt, iUVstrategy, rmin)
        self.ctx.alloc_records(nslots, field_dtype)

    def set_buoys(self, xPosC0, vJIt, z1stModelRec=None, zLstModelRec=None, sort=True):
        self.ctx.set_buoys(xPosC0, vJIt, z1stModelRec, zLstModelRec, sort=sort)

    def _check_exact(self, xUu, xVv, xIC):
        dt = self.ctx.field_dtype
        for nm, a in (("u_ice", xUu), ("v_ice", xVv), ("siconc", xIC)):
            a = np.asarray(a)
            if a.dtype.newbyteorder('=') == dt:              # same type up to byte order (NetCDF-3 data are big-endian)
                continue
            if not np.array_equal(a.astype(dt).astype(a.dtype), a, equal_nan=True):
                raise ValueError("%s is not exactly representable as %s; allocate float64 records" % (nm, dt))

    def load_record(self, slot, xUu, xVv, xIC):
        """Fields must be exactly representable in the record dtype (NEMO output is f4)."""
        self._check_exact(xUu, xVv, xIC)
        self.ctx.push_record(slot, xUu, xVv, xIC)

    def band(self, age=0):
        """Rows [j0,j1) of the next record(s) that this tracker's buoys can touch (see sitrk_buoy_rows)."""
        return self.ctx.band(age)

    def load_record_rows(self, slot, j0, j1, xUu_rows, xVv_rows, xIC_rows):
        """Row-band ingest: only rows [j0,j1) of the record, as returned by band()."""
        self._check_exact(xUu_rows, xVv_rows, xIC_rows)
        self.ctx.push_record_rows(slot, j0, j1, xUu_rows, xVv_rows, xIC_rows)

    def step(self, jrec, slot=0):
        self.ctx.step(slot, jrec)

    def run(self, jrec0, slot0, nrec):
        """records jrec0 .. jrec0+nrec-1 from slots (slot0+k) % nslots: one fused launch where the library can
        (sitrk_run), same results as nrec calls of step()"""
        if nrec == 1:
            self.ctx.step(slot0, jrec0)
        else:
            self.ctx.run(slot0, jrec0, nrec)

    def record(self, jrec, latlon=False):
        return self.ctx.fetch_record(jrec, latlon=latlon)

    def state(self):
        s = self.ctx.fetch()
        s["vJIt"] = s.pop("jiT").astype(np.int64)
        s["VRTCS"] = vertices_of(s["vJIt"])
        s["iAlive"] = s.pop("alive")
        return s

    def alive_count(self):
        return self.ctx.count_alive()

    def close(self):
        self.ctx.close()
