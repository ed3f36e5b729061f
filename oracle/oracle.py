"""ctypes binding of the CPU oracle (oracle/sitrk_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
Function names mirror the reference's (`sit.*`) so the parity tests read like
calls into the reference.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

FillValue = -9999.0
rmin_conc = 0.1      # sitrack/tracking.py:4
rFoundKM = 2.5       # sitrack/tracking.py:5

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_i8p = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")
_i64 = C.c_int64
_dbl = C.c_double
_int = C.c_int


def build(force=False):
    """Compile liboracle.so with the committed Makefile (gcc, -ffp-contract=off)."""
    src = os.path.join(_HERE, "sitrk_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_ccw.argtypes = [_f64p] * 3
        L.orc_intersect2seg.argtypes = [_f64p] * 4
        L.orc_is_inside_quadrangle.argtypes = [_dbl, _dbl, _f64p]
        L.orc_crossed_edge.argtypes = [_f64p, _f64p, _i64p, _f64p, _f64p, _i64, _i64]
        L.orc_new_host_cell.argtypes = [_int, _f64p, _f64p, _i64p, _f64p, _f64p, _i64, _i64]
        L.orc_updt_ind4newcell.argtypes = [_int, _i64p, _i64p]
        L.orc_survive.argtypes = [_i64, _i64, _i8p, _f64p, _i64, _i64, _dbl, C.POINTER(_int)]
        L.orc_haversine.argtypes = [_dbl] * 4
        L.orc_haversine.restype = _dbl
        L.orc_haversine_field.argtypes = [_dbl, _dbl, _f64p, _f64p, _i64, _f64p]
        L.orc_haversine_field.restype = None
        L.orc_nearest_point.argtypes = [_dbl, _dbl, _f64p, _f64p, C.c_void_p, _i64, _i64, _dbl, _int,
                                        C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_dbl)]
        L.orc_nearest_point.restype = None
        L.orc_find_containing_cell.argtypes = [_dbl, _dbl, _i64, _i64, _f64p, _f64p, _i64, _i64, _i64p, _i64p]
        L.orc_seed_init.argtypes = [_i64, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _i8p, _f64p,
                                    _i64, _i64, _dbl, _dbl, _int, _i64p, _i64p, _i8p, _i8p, _int]
        L.orc_advect_record.argtypes = [_i64, _i64, _dbl, _int, _dbl, _i64, _i64,
                                        _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _i8p,
                                        _f64p, _f64p, _f64p, _i64p, _i64p,
                                        _f64p, _i64p, _i64p, _i8p,
                                        C.c_void_p, C.c_void_p, C.POINTER(_i64), _int]
        L.orc_geo2cart.argtypes = [_i64, _f64p, _dbl, _dbl, _f64p]
        L.orc_geo2cart.restype = None
        L.orc_cart2geo.argtypes = [_i64, _f64p, _dbl, _dbl, _f64p]
        L.orc_cart2geo.restype = None
        L.orc_get_time_span.argtypes = [_dbl, _i64p, _i64, _i64, _i64, _i64, _int, _i64] + [C.POINTER(_i64)] * 5
        _lib = L
    return _lib


def _pt(p):
    return np.ascontiguousarray(p, dtype=np.float64)


def _chk(st, what):
    if st < 0:
        raise IndexError("oracle %s: reference would fail here (status %d)" % (what, st))
    return st


# ---- predicates -----------------------------------------------------------
def ccw(A, B, Cc):
    return bool(lib().orc_ccw(_pt(A), _pt(B), _pt(Cc)))


def intersect2Seg(A, B, Cc, D):
    return bool(lib().orc_intersect2seg(_pt(A), _pt(B), _pt(Cc), _pt(D)))


def IsInsideQuadrangle(y, x, quad):
    return bool(lib().orc_is_inside_quadrangle(float(y), float(x), _pt(quad)))


def CrossedEdge(P1, P2, ji4vert, Yf, Xf):
    Nj, Ni = Yf.shape
    v = np.ascontiguousarray(ji4vert, dtype=np.int64)
    return _chk(lib().orc_crossed_edge(_pt(P1), _pt(P2), v, Yf, Xf, Nj, Ni), "CrossedEdge")


def NewHostCell(kcross, P1, P2, ji4vert, Yf, Xf):
    Nj, Ni = Yf.shape
    v = np.ascontiguousarray(ji4vert, dtype=np.int64)
    return _chk(lib().orc_new_host_cell(int(kcross), _pt(P1), _pt(P2), v, Yf, Xf, Nj, Ni), "NewHostCell")


def UpdtInd4NewCell(knhc, ji4vert, jiT):
    v = np.ascontiguousarray(ji4vert, dtype=np.int64).copy()
    t = np.ascontiguousarray(jiT, dtype=np.int64).copy()
    _chk(lib().orc_updt_ind4newcell(int(knhc), v, t), "UpdtInd4NewCell")
    return v, t


def Survive(jiT, tmask, sic, rmin=rmin_conc, return_which=False):
    Nj, Ni = tmask.shape
    w = _int(0)
    st = _chk(lib().orc_survive(int(jiT[0]), int(jiT[1]), tmask, np.ascontiguousarray(sic, dtype=np.float64),
                                Nj, Ni, rmin, C.byref(w)), "Survive")
    return (st, w.value) if return_which else st


# ---- locate ----------------------------------------------------------------
def Haversine(plat, plon, xlat, xlon):
    xlat = np.ascontiguousarray(xlat, dtype=np.float64)
    xlon = np.ascontiguousarray(xlon, dtype=np.float64)
    out = np.empty_like(xlat)
    lib().orc_haversine_field(float(plat), float(plon), xlat.ravel(), xlon.ravel(), xlat.size, out.ravel())
    return out


def NearestPoint(pnt, latT, lonT, rd_found_km=10., resolkm=None, max_itr=5, return_dist=False):
    Nj, Ni = latT.shape
    jy, jx, d = _i64(0), _i64(0), _dbl(0)
    rp = None
    if resolkm is not None and np.shape(resolkm) == (Nj, Ni):
        resolkm = np.ascontiguousarray(resolkm, dtype=np.float64)
        rp = resolkm.ctypes.data_as(C.c_void_p)
    lib().orc_nearest_point(float(pnt[0]), float(pnt[1]), latT, lonT, rp, Nj, Ni, float(rd_found_km),
                            int(max_itr), C.byref(jy), C.byref(jx), C.byref(d))
    return (jy.value, jx.value, d.value) if return_dist else (jy.value, jx.value)


def FindContainingCell(pyx, kjiT, Yf, Xf):
    Nj, Ni = Yf.shape
    jiT = np.zeros(2, dtype=np.int64)
    vert = np.zeros((2, 4), dtype=np.int64)
    st = _chk(lib().orc_find_containing_cell(float(pyx[0]), float(pyx[1]), int(kjiT[0]), int(kjiT[1]),
                                             Yf, Xf, Nj, Ni, jiT, vert.reshape(-1)), "FindContainingCell")
    return bool(st), jiT, vert


def SeedInit(pIDs, pSG, pSC, latT, lonT, Yf, Xf, resolkm, tmask, sic, return_why=False, nthreads=1):
    """Same return tuple as the reference (tracking.py:178)."""
    nP = pSG.shape[0]
    Nj, Ni = latT.shape
    jiT = np.zeros((nP, 2), dtype=np.int64)
    vert = np.zeros((nP, 2, 4), dtype=np.int64)
    keep = np.zeros(nP, dtype=np.int8)
    why = np.zeros(nP, dtype=np.int8)
    pSG = np.ascontiguousarray(pSG, dtype=np.float64)
    pSC = np.ascontiguousarray(pSC, dtype=np.float64)
    _chk(lib().orc_seed_init(nP, pSG.reshape(-1), pSC.reshape(-1), latT, lonT, Yf, Xf,
                             np.ascontiguousarray(resolkm, dtype=np.float64), tmask,
                             np.ascontiguousarray(sic, dtype=np.float64), Nj, Ni,
                             rmin_conc, rFoundKM, 10, jiT.reshape(-1), vert.reshape(-1), keep, why, int(nthreads)), "SeedInit")
    iKeep = np.where(keep == 1)[0]
    out = (len(iKeep), pSG[iKeep, :], pSC[iKeep, :], np.asarray(pIDs)[iKeep], jiT[iKeep, :], vert[iKeep, :, :], iKeep)
    return out + (why,) if return_why else out


# ---- the hot loop ------------------------------------------------------------
def vertices_of(jiT):
    """VRTCS from vJIt (locate.py:320-321): [[jT-1,jT-1,jT,jT],[iT-1,iT,iT,iT-1]]."""
    jiT = np.asarray(jiT, dtype=np.int64)
    j, i = jiT[:, 0], jiT[:, 1]
    v = np.empty((jiT.shape[0], 2, 4), dtype=np.int64)
    v[:, 0, 0] = j - 1; v[:, 0, 1] = j - 1; v[:, 0, 2] = j; v[:, 0, 3] = j
    v[:, 1, 0] = i - 1; v[:, 1, 1] = i; v[:, 1, 2] = i; v[:, 1, 3] = i - 1
    return v


class Tracker:
    """State of the reference hot loop (si3_part_tracker.py:324-330) driven by the oracle."""

    def __init__(self, grid, yx0, jiT0, vert0=None, rec_first=None, rec_last=None,
                 rdt=3600., uv_strategy=1, rmin=rmin_conc, nthreads=1):
        self.g = {k: np.ascontiguousarray(grid[k], dtype=np.float64) for k in ("Yf", "Xf", "Yu", "Xu", "Yv", "Xv")}
        self.tmask = np.ascontiguousarray(grid["tmask"], dtype=np.int8)
        self.Nj, self.Ni = self.tmask.shape
        self.nP = yx0.shape[0]
        self.pos = np.ascontiguousarray(yx0, dtype=np.float64).copy()
        self.jiT = np.ascontiguousarray(jiT0, dtype=np.int64).copy()
        self.vert = (vertices_of(self.jiT) if vert0 is None else np.ascontiguousarray(vert0, dtype=np.int64).copy())
        self.alive = np.ones(self.nP, dtype=np.int8)
        big = np.iinfo(np.int64).max
        self.rec_first = (np.zeros(self.nP, dtype=np.int64) if rec_first is None
                          else np.ascontiguousarray(rec_first, dtype=np.int64))
        self.rec_last = (np.full(self.nP, big, dtype=np.int64) if rec_last is None
                         else np.ascontiguousarray(rec_last, dtype=np.int64))
        self.rdt, self.uv_strategy, self.rmin, self.nthreads = float(rdt), int(uv_strategy), float(rmin), int(nthreads)
        self.ncross = 0

    def step(self, jrec, u, v, sic, want_out=True):
        """One record; returns (pos_next, mask_next) = (xPosC[jt+1], xmask[jt+1,:,0])."""
        u = np.ascontiguousarray(u, dtype=np.float64)
        v = np.ascontiguousarray(v, dtype=np.float64)
        sic = np.ascontiguousarray(sic, dtype=np.float64)
        if want_out:
            pn = np.empty((self.nP, 2), dtype=np.float64)
            mn = np.empty(self.nP, dtype=np.int8)
            pnp, mnp = pn.ctypes.data_as(C.c_void_p), mn.ctypes.data_as(C.c_void_p)
        else:
            pn = mn = pnp = mnp = None
        nc = _i64(0)
        g = self.g
        _chk(lib().orc_advect_record(self.nP, int(jrec), self.rdt, self.uv_strategy, self.rmin, self.Nj, self.Ni,
                                     g["Yf"], g["Xf"], g["Yu"], g["Xu"], g["Yv"], g["Xv"], self.tmask,
                                     u, v, sic, self.rec_first, self.rec_last,
                                     self.pos.reshape(-1), self.jiT.reshape(-1), self.vert.reshape(-1), self.alive,
                                     pnp, mnp, C.byref(nc), self.nthreads), "advect_record")
        self.ncross += nc.value
        return pn, mn


# ---- projection --------------------------------------------------------------
def Geo2CartNPSkm1D(pcoorG, lat0=70., lon0=-45.):
    g = np.ascontiguousarray(pcoorG, dtype=np.float64)
    out = np.empty_like(g)
    lib().orc_geo2cart(g.shape[0], g.reshape(-1), lat0, lon0, out.reshape(-1))
    return out


def CartNPSkm2Geo1D(pcoorC, lat0=70., lon0=-45.):
    c = np.ascontiguousarray(pcoorC, dtype=np.float64)
    out = np.empty_like(c)
    lib().orc_cart2geo(c.shape[0], c.reshape(-1), lat0, lon0, out.reshape(-1))
    return out


def GetTimeSpan(dt, vtime_mod, iSdA, iMdA, iMdB, iStop=None):
    vt = np.ascontiguousarray(vtime_mod, dtype=np.int64)
    o = [_i64(0) for _ in range(5)]
    st = lib().orc_get_time_span(float(dt), vt, vt.size, int(iSdA), int(iMdA), int(iMdB),
                                 0 if iStop is None else 1, 0 if iStop is None else int(iStop),
                                 *[C.byref(x) for x in o])
    if st == -1:
        raise SystemExit("PROBLEM: time in the seeding file is outside of what model spans!")
    _chk(st, "GetTimeSpan")
    return tuple(x.value for x in o)
