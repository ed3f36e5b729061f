"""The reference's per-buoy predicates under their own names and argument orders, evaluated on the GPU.

`si3_part_tracker.py:378-490` calls these one buoy at a time (`sit.intersect2Seg` :432-433, `sit.IsInsideQuadrangle`
:466, `sit.CrossedEdge` :474, `sit.NewHostCell` :477, `sit.UpdtInd4NewCell` :480, `sit.Survive` :483).  `IceTracker.step`
replaces that whole loop; the functions below exist so that code (and tests) written against the reference's scalar
interface keep working: each call evaluates the device-side predicate of the hot path through the C ABI's probes
(`sitrk_eval_inside`, `sitrk_eval_intersect`, `sitrk_survive_mask`) -- a GPU round trip per call, so they are for checking
and porting, not for speed.  Index arithmetic on the caller's arrays (which vertex, numpy's negative-index wrap, the
in-place update of `UpdtInd4NewCell`) is done here exactly as numpy does it for the reference.
"""
import numpy as np

from . import _lib

rmin_conc = 0.1                          # reference sitrack/tracking.py:4
_scratch_ctx = None


def _default(ctx):
    if ctx is not None:
        return ctx
    from .tracking import default_context
    return default_context()


def _scratch():
    """A context of its own for predicates that need a grid (Survive): never disturbs a tracker's context."""
    global _scratch_ctx
    if _scratch_ctx is None:
        from .tracking import default_context
        _scratch_ctx = _lib.Context(default_context().device)
    return _scratch_ctx


def _pt(p):
    return [float(p[0]), float(p[1])]


def _ccw_(pcA, pcB, pcC, ctx=None):
    """reference tracking.py:44-49"""
    segs = np.array([[_pt(pcA), _pt(pcB), _pt(pcC), _pt(pcC)]], dtype=np.float64)
    return bool(_default(ctx).eval_intersect(segs)[1][0])


def intersect2Seg(pcA, pcB, pcC, pcD, ctx=None):
    """reference tracking.py:51-58: True if segments AB and CD intersect."""
    segs = np.array([[_pt(pcA), _pt(pcB), _pt(pcC), _pt(pcD)]], dtype=np.float64)
    return bool(_default(ctx).eval_intersect(segs)[0][0])


def IsInsideQuadrangle(y, x, quad, ctx=None):
    """reference locate.py:49-78: `quad` = 4 vertices [y,x]; same answers as the reference's ray cast
    (tools/tests/test_pnt_inside_quad.py of the reference: True False False True on its four points)."""
    q = np.asarray(quad, dtype=np.float64).reshape(1, 4, 2)
    return bool(_default(ctx).eval_inside(np.array([[float(y), float(x)]]), q)[0])


def CrossedEdge(pP1, pP2, ji4vert, pY, pX, iverbose=0, ctx=None):
    """reference tracking.py:182-200: 1 bottom, 2 right-hand, 3 upper, 4 left-hand edge crossed by P1->P2
    (the first that intersects; 4 if none does)."""
    ji = np.asarray(ji4vert)
    P1, P2 = _pt(pP1), _pt(pP2)
    segs = np.empty((4, 4, 2), dtype=np.float64)
    for kk in range(4):
        j1, i1 = ji[:, kk]
        j2, i2 = ji[:, (kk + 1) % 4]
        segs[kk] = [P1, P2, [pY[j1, i1], pX[j1, i1]], [pY[j2, i2], pX[j2, i2]]]
    hit = _default(ctx).eval_intersect(segs)[0]
    kk = int(np.argmax(hit)) if hit.any() else 3
    if iverbose > 0:
        vdir = ['bottom', 'right-hand', 'upper', 'left-hand']
        print('    [CrossedEdge()]: particle is crossing the ' + vdir[kk] + ' edge of the mesh!')
    return kk + 1


# NewHostCell: for the crossed edge, the two (vertex, step to the next F-point along the prolonged grid line, code) tests
# in the reference's order (tracking.py:217-243); vertex numbering 0 bl, 1 br, 2 ur, 3 ul
_NHC = {1: ((0, (-1, 0), 5), (1, (-1, 0), 6)),
        2: ((1, (0, 1), 6), (2, (0, 1), 7)),
        3: ((3, (1, 0), 8), (2, (1, 0), 7)),
        4: ((3, (0, -1), 8), (0, (0, -1), 5))}


def NewHostCell(kcross, pP1, pP2, ji4vert, pY, pX, iverbose=0, ctx=None):
    """reference tracking.py:203-249: 1..4 = the cell across the crossed edge, 5..8 = bottom-left, bottom-right,
    upper-right, upper-left diagonal neighbour."""
    knhc = int(kcross)
    if knhc in _NHC:
        ji = np.asarray(ji4vert)
        P1, P2 = _pt(pP1), _pt(pP2)
        segs = np.empty((2, 4, 2), dtype=np.float64)
        for t, (kv, (dj, di), _) in enumerate(_NHC[knhc]):
            j, i = ji[:, kv]
            segs[t] = [P1, P2, [pY[j, i], pX[j, i]], [pY[j + dj, i + di], pX[j + dj, i + di]]]      # numpy wraps j-1 = -1
        hit = _default(ctx).eval_intersect(segs)[0]
        if hit[0]:
            knhc = _NHC[int(kcross)][0][2]
        elif hit[1]:
            knhc = _NHC[int(kcross)][1][2]
    if iverbose > 0:
        vdir = ['bottom', 'RHS', 'upper', 'LHS', 'bottom-LHS', 'bottom-RHS', 'upper-RHS', 'upper-LHS']
        print('    *** Particle is moving into the ' + vdir[knhc - 1] + ' mesh !')
    return knhc


_SHIFT = {1: (-1, 0), 2: (0, 1), 3: (1, 0), 4: (0, -1), 5: (-1, -1), 6: (-1, 1), 7: (1, 1), 8: (1, -1)}


def UpdtInd4NewCell(knhc, ji4vert, kjiT, iverbose=0):
    """reference tracking.py:253-305: shifts the 4 vertex indices and the T-point indices IN PLACE (and returns them);
    an unknown code ends the program like the reference does."""
    if knhc not in _SHIFT:
        print('ERROR: unknown direction, knhc=', knhc)
        raise SystemExit(0)
    dj, di = _SHIFT[knhc]
    if iverbose > 0 and knhc >= 5:
        print(' * [UpdtInd4NewCell()]: WE HAVE A %d !!!!' % knhc)
    if dj:
        ji4vert[0, :] = ji4vert[0, :] + dj
        kjiT[0] = kjiT[0] + dj
    if di:
        ji4vert[1, :] = ji4vert[1, :] + di
        kjiT[1] = kjiT[1] + di
    return ji4vert, kjiT


def Haversine(plat, plon, xlat, xlon, ctx=None):
    """reference util.py:85-103: distance [km] between the point (plat,plon) and every point of (xlat,xlon), R = 6360 km."""
    xlat = np.asarray(xlat, dtype=np.float64)
    return _default(ctx).eval_haversine(float(plat), float(plon), xlat, np.asarray(xlon, dtype=np.float64)).reshape(xlat.shape)


def nearest_point_with_previous(haversine, pntGcoor, lat, lon, rd_found_km, res, ji_prv, np_box_r, max_itr):
    """`NearestPoint` with a previous position (reference locate.py:241-271): the first pass searches the box of `np_box_r` points
    around `ji_prv`, every later pass the whole domain; the acceptance radius starts at 0.5 * resolkm[jy, jx] -- indexed, like the
    reference, by the minimum's position INSIDE THE BOX -- or at `rd_found_km`, and grows by 20 % per failed pass after the first.
    `haversine(plat, plon, xlat, xlon)` evaluates util.Haversine on an array (the device probe here; the CPU tests pass their own)."""
    Ny, Nx = lat.shape
    j_prv, i_prv = int(ji_prv[0]), int(ji_prv[1])
    j1, j2 = max(j_prv - np_box_r, 0), min(j_prv + np_box_r + 1, Ny)
    i1, i2 = max(i_prv - np_box_r, 0), min(i_prv + np_box_r + 1, Nx)
    jy = jx = -1
    found, rfnd, igo = False, rd_found_km, 0
    while not found and igo < max_itr:
        igo += 1
        if igo > 1:
            j1, i1, j2, i2 = 0, 0, Ny, Nx                 # the whole domain from the second pass on
        xd = np.asarray(haversine(pntGcoor[0], pntGcoor[1], np.ascontiguousarray(lat[j1:j2, i1:i2]), np.ascontiguousarray(lon[j1:j2, i1:i2])))
        jy, jx = (int(k) for k in np.unravel_index(np.argmin(xd), xd.shape))      # first minimum in C order (find_ji_of_min, :13-20)
        if igo == 1 and res is not None:
            rfnd = 0.5 * res[jy, jx]
        found = bool(xd[jy, jx] < rfnd)
        if igo > 1 and not found:
            rfnd = 1.2 * rfnd
    jy, jx = jy + j1, jx + i1
    if jy < 0 or jx < 0 or jy >= Ny or jx >= Nx or igo == max_itr:
        return (-1, -1)
    return (jy, jx)


def NearestPoint(pntGcoor, pLat, pLon, rd_found_km=10., resolkm=[], ji_prv=(), np_box_r=10, max_itr=5, ctx=None):
    """reference locate.py:222-276: (j,i) of the grid point nearest to `pntGcoor` = (lat,lon), (-1,-1) if the acceptance
    loop gives up.  Whole-domain search (as SeedInit uses it) through the device's bounding-sphere search; with a previous
    position `ji_prv` (round 4) the reference's box-then-domain passes, the distances evaluated on the device."""
    lat = np.ascontiguousarray(pLat, dtype=np.float64)
    lon = np.ascontiguousarray(pLon, dtype=np.float64)
    if lon.shape != lat.shape:
        print('ERROR [NearestPoint]: `pLat` & `pLon` do not have the same shape!')
        raise SystemExit(0)
    res = np.asarray(resolkm, dtype=np.float64) if np.shape(resolkm) == lat.shape else None       # `l2Dresol` of the reference
    c = ctx if ctx is not None else _scratch()
    if len(ji_prv) == 2:
        return nearest_point_with_previous(lambda a, b, x, y: c.eval_haversine(float(a), float(b), x, y), (float(pntGcoor[0]), float(pntGcoor[1])),
                                           lat, lon, rd_found_km, res, ji_prv, int(np_box_r), int(max_itr))
    if ctx is None:
        z = np.zeros(lat.shape)
        c.set_grid(z, z, z, z, z, z, np.ones(lat.shape, dtype=np.int8))
    ji, _ = c.nearest_point(np.array([[float(pntGcoor[0]), float(pntGcoor[1])]]), lat, lon, res, rd_found_km, max_itr)
    return (int(ji[0, 0]), int(ji[0, 1]))


def Survive(kID, kjiT, pmskT, pIceC=[], iverbose=0, ctx=None):
    """reference tracking.py:62-93: 1 = kill (domain rim, land-sea mask stencil, 5-point mean ice concentration
    < rmin_conc), 0 = survive.  Like the reference, fails with UnboundLocalError when a buoy passes the first two
    tests and `pIceC` is not a 2-D field."""
    m = np.ascontiguousarray(pmskT)
    Nj, Ni = m.shape
    jT, iT = int(kjiT[0]), int(kjiT[1])
    has_ice = (len(np.shape(pIceC)) == 2)
    c = ctx if ctx is not None else _scratch()
    if ctx is None:
        z = np.zeros((Nj, Ni))
        c.set_grid(z, z, z, z, z, z, m.astype(np.int8))
        c.set_params(3600., 1, rmin_conc)
    sic = np.ascontiguousarray(pIceC, dtype=np.float64) if has_ice else np.ones((Nj, Ni))
    if not (0 <= jT < Nj and 0 <= iT < Ni):
        raise IndexError("Survive: cell (%d,%d) outside the %dx%d domain" % (jT, iT, Nj, Ni))
    ikill = int(c.survive_mask(sic)[jT, iT])
    if ikill == 0 and not has_ice:
        raise UnboundLocalError("local variable 'zic' referenced before assignment")
    if iverbose > 0 and ikill:
        print('        ===> I CANCEL buoy ' + str(kID) + '!!!')
    return ikill
