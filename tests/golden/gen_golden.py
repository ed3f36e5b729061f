#!/usr/bin/env python3
"""Generate golden vectors G1..G9 from the REFERENCE's own Python functions.

Runs only in the build container (needs /root/reference mounted read-only):

    python tests/golden/gen_golden.py

It imports sitrack/{util,locate,tracking}.py through tests/golden/refload.py,
calls the reference functions on seeded inputs and commits ONLY the numeric
inputs/outputs as small .npz files next to this script.  The hot loop itself
(si3_part_tracker.py:361-496) lives under `__main__` and cannot be imported, so
`reference_loop()` below drives the imported reference predicates in that
loop's order.  cartopy/netCDF4 are absent, so the
projection fixture G7 is read from the reference's committed NetCDF data file
with the image's `h5dump`.
"""
import contextlib
import io
import os
import re
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from refload import load_reference  # noqa: E402
from sitrack_amd import synthetic as syn  # noqa: E402

util, locate, tracking = load_reference()
FILL = -9999.0


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def save(name, **kw):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **kw)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024.))


# --------------------------------------------------------------------------- G1
def g1_inside():
    rng = np.random.default_rng(1234)
    quads, pts = [], []
    # the reference's own manual test (tools/tests/test_pnt_inside_quad.py:16-24)
    q0 = np.array([[0., 0.], [3., 0.], [4., 4.], [1., 3.5]])
    for p in ([2., 2.], [6., 6.], [-1., 2.], [3.1, 3.6]):
        quads.append(q0); pts.append(p)
    # random convex-ish and arbitrary quads, points in the bounding box
    for _ in range(1500):
        c = rng.uniform(-50, 50, 2)
        ang = np.sort(rng.uniform(0, 2 * np.pi, 4))
        rad = rng.uniform(1, 6, 4)
        q = np.stack([c[0] + rad * np.sin(ang), c[1] + rad * np.cos(ang)], axis=1)
        if rng.random() < 0.3:
            q = q[rng.permutation(4)]             # self-intersecting / clockwise
        quads.append(q); pts.append(c + rng.uniform(-7, 7, 2))
    # axis-aligned cells: points on edges, vertices, horizontal edges (stale xints)
    for _ in range(600):
        y0, x0 = rng.integers(-5, 5, 2) * 4.0
        q = np.array([[y0, x0], [y0, x0 + 4], [y0 + 4, x0 + 4], [y0 + 4, x0]])
        if rng.random() < 0.3:
            q = np.roll(q, rng.integers(0, 4), axis=0)
        ch = rng.integers(0, 6)
        yy = [y0, y0 + 4, y0 + 2, y0 + rng.uniform(0, 4), y0 - 1, y0 + 5][ch]
        ch = rng.integers(0, 6)
        xx = [x0, x0 + 4, x0 + 2, x0 + rng.uniform(0, 4), x0 - 1, x0 + 5][ch]
        quads.append(q); pts.append([yy, xx])
    # trapezoids with one horizontal edge and a point level with it
    for _ in range(300):
        y0 = rng.uniform(-5, 5); x0 = rng.uniform(-5, 5)
        q = np.array([[y0, x0], [y0, x0 + 3], [y0 + rng.uniform(1, 3), x0 + 3.5], [y0 + rng.uniform(1, 3), x0 - 0.5]])
        k = rng.integers(0, 4)
        p = [q[k, 0], q[k, 1] + rng.choice([-1., 0., 1.])]
        quads.append(q); pts.append(p)
    quads = np.array(quads); pts = np.array(pts, dtype=np.float64)
    out = np.array([locate.IsInsideQuadrangle(p[0], p[1], q) for p, q in zip(pts, quads)], dtype=bool)
    assert list(out[:4]) == [True, False, False, True]
    save("g1_inside.npz", quads=quads, pts=pts, inside=out)


# --------------------------------------------------------------------------- G2
def g2_intersect():
    rng = np.random.default_rng(1235)
    n = 3000
    P = rng.uniform(-10, 10, (n, 4, 2))
    # integer lattice cases -> exact collinear / touching configurations
    m = 1200
    L = rng.integers(-3, 4, (m, 4, 2)).astype(np.float64)
    P = np.concatenate([P, L])
    out = np.array([tracking.intersect2Seg(list(p[0]), list(p[1]), list(p[2]), list(p[3])) for p in P], dtype=bool)
    ccw = np.array([tracking._ccw_(list(p[0]), list(p[1]), list(p[2])) for p in P], dtype=bool)
    save("g2_intersect.npz", P=P, intersect=out, ccw=ccw)


# --------------------------------------------------------------------------- G3
def g3_crossing():
    rng = np.random.default_rng(1236)
    g = syn.make_grid(40, 44, dkm=4.0, warp=1.0)
    Yf, Xf = g["Yf"], g["Xf"]
    n = 2500
    jiT = np.stack([rng.integers(3, 37, n), rng.integers(3, 41, n)], axis=1).astype(np.int64)
    P1 = np.empty((n, 2)); P2 = np.empty((n, 2))
    icross = np.empty(n, dtype=np.int64); inhc = np.empty(n, dtype=np.int64)
    vert_in = np.empty((n, 2, 4), dtype=np.int64)
    vert_out = np.empty((n, 2, 4), dtype=np.int64)
    jiT_out = np.empty((n, 2), dtype=np.int64)
    for k in range(n):
        j, i = jiT[k]
        vert = np.array([[j - 1, j - 1, j, j], [i - 1, i, i, i - 1]], dtype=np.int64)
        quad = np.array([[Yf[vert[0, c], vert[1, c]], Xf[vert[0, c], vert[1, c]]] for c in range(4)])
        w = rng.dirichlet(np.ones(4))
        p1 = w @ quad                                     # inside the cell
        mode = k % 5
        if mode == 0:      # short move, may or may not leave: fall-through cases included
            p2 = p1 + rng.normal(0, 1.5, 2)
        elif mode == 1:    # towards a vertex -> diagonal moves
            c = rng.integers(0, 4)
            p2 = p1 + (quad[c] - p1) * rng.uniform(1.02, 1.6)
        elif mode == 2:    # long move
            p2 = p1 + rng.normal(0, 5.0, 2)
        elif mode == 3:    # P1 outside the cell, P2 further (no edge intersected -> 4)
            p1 = quad.mean(axis=0) + np.array([9., 9.])
            p2 = p1 + rng.normal(0, 1.0, 2)
        else:              # exactly through a vertex
            c = rng.integers(0, 4)
            p2 = p1 + (quad[c] - p1) * 2.0
        P1[k], P2[k] = p1, p2
        vert_in[k] = vert
        ic = tracking.CrossedEdge(list(p1), list(p2), vert, Yf, Xf)
        nh = tracking.NewHostCell(ic, list(p1), list(p2), vert, Yf, Xf)
        v2, t2 = tracking.UpdtInd4NewCell(nh, vert.copy(), jiT[k].copy())
        icross[k], inhc[k] = ic, nh
        vert_out[k], jiT_out[k] = v2, t2
    assert set(np.unique(inhc)) == set(range(1, 9)), np.unique(inhc)
    save("g3_crossing.npz", Yf=Yf, Xf=Xf, jiT=jiT, vert=vert_in, P1=P1, P2=P2,
         icross=icross, inhc=inhc, vert_out=vert_out, jiT_out=jiT_out)


# --------------------------------------------------------------------------- G4
def g4_survive():
    rng = np.random.default_rng(1237)
    Nj, Ni = 24, 30
    tmask = (rng.random((Nj, Ni)) > 0.12).astype(np.int8)
    sic = rng.choice([0.0, 0.05, 0.09, 0.1, 0.11, 0.3, 1.0], size=(Nj, Ni)).astype(np.float64)
    sic32 = rng.uniform(0, 0.3, (Nj, Ni)).astype(np.float32)     # fp32 field promoted like the driver does
    jj, ii = np.meshgrid(np.arange(Nj), np.arange(Ni), indexing="ij")
    jiT = np.stack([jj.ravel(), ii.ravel()], axis=1).astype(np.int64)
    with quiet():
        k1 = np.array([tracking.Survive(0, list(t), tmask, pIceC=sic) for t in jiT], dtype=np.int64)
        k2 = np.array([tracking.Survive(0, list(t), tmask, pIceC=sic32.astype(np.float64)) for t in jiT], dtype=np.int64)
        ones = np.ones((Nj, Ni), dtype=np.int8)
        k3 = np.array([tracking.Survive(0, list(t), ones, pIceC=sic32.astype(np.float64)) for t in jiT], dtype=np.int64)
    save("g4_survive.npz", tmask=tmask, sic=sic, sic32=sic32, jiT=jiT, kill_a=k1, kill_b=k2, kill_c=k3)


def g4b_survive_wide():
    """`Survive` for every cell of a mesh whose rows are 16-byte aligned and wider than one strip of the device's register-rolling
    kernel (44 x 252: strip boundary at column 248), and of a 40 x 48 one: mask values other than 0/1 (a 5-point sum can reach 5
    with a land point in it), ice exactly at / around the threshold, the asymmetric [j-1,i-1] stencil point."""
    rng = np.random.default_rng(1243)
    out = {}
    for tag, (Nj, Ni) in (("a", (44, 252)), ("b", (40, 48))):
        tmask = (rng.random((Nj, Ni)) > 0.08).astype(np.int8)
        tmask[rng.integers(2, Nj - 2, 8), rng.integers(2, Ni - 2, 8)] = 2
        tmask[rng.integers(2, Nj - 2, 8), rng.integers(2, Ni - 2, 8)] = -1
        sic = rng.choice([0.0, 0.05, 0.0999, 0.1, 0.1001, 0.12, 0.5, 1.0], size=(Nj, Ni)).astype(np.float64)
        sic32 = rng.uniform(0.05, 0.16, (Nj, Ni)).astype(np.float32)
        with quiet():
            k1 = np.array([[tracking.Survive(0, [j, i], tmask, pIceC=sic) for i in range(Ni)] for j in range(Nj)], dtype=np.int8)
            k2 = np.array([[tracking.Survive(0, [j, i], tmask, pIceC=sic32.astype(np.float64)) for i in range(Ni)] for j in range(Nj)], dtype=np.int8)
        print("   G4b %s: killed %.2f / %.2f of the interior" % (tag, k1[2:-2, 2:-2].mean(), k2[2:-2, 2:-2].mean()))
        out.update({tag + "_tmask": tmask, tag + "_sic": sic, tag + "_sic32": sic32, tag + "_kill": k1, tag + "_kill32": k2})
    save("g4b_survive_wide.npz", **out)


# --------------------------------------------------------------------------- G5
def polar_grid(Nj, Ni, dkm, warp, yc=-300., xc=200.):
    """Synthetic grid placed near the pole; lat/lon of T-points from the build's own
    inverse projection (inputs only -- any consistent lat/lon would do)."""
    from oracle import oracle as orc
    g = syn.make_grid(Nj, Ni, dkm=dkm, warp=warp)
    for k in ("Yt", "Yu", "Yv", "Yf"):
        g[k] = g[k] + yc
    for k in ("Xt", "Xu", "Xv", "Xf"):
        g[k] = g[k] + xc
    ll = orc.CartNPSkm2Geo1D(np.stack([g["Yt"].ravel(), g["Xt"].ravel()], axis=1))
    g["latT"] = np.ascontiguousarray(ll[:, 0].reshape(Nj, Ni))
    g["lonT"] = np.ascontiguousarray(np.mod(ll[:, 1], 360.).reshape(Nj, Ni))   # ncio.py:50
    return g


def g5_seedinit():
    from oracle import oracle as orc
    rng = np.random.default_rng(1238)
    Nj, Ni, dkm = 40, 44, 12.0
    g = polar_grid(Nj, Ni, dkm, warp=1.0)
    tmask = g["tmask"].copy()
    tmask[18:22, 10:14] = 0
    sic = np.ones((Nj, Ni)); sic[5:12, 30:40] = 0.02; sic[25:30, 5:9] = 0.12
    # seeds: random inside, exactly on T-points, near F-points (slightly off the tie), off-grid
    yx = []
    n_rand = 260
    ylo, yhi = g["Yt"].min(), g["Yt"].max(); xlo, xhi = g["Xt"].min(), g["Xt"].max()
    for _ in range(n_rand):
        yx.append([rng.uniform(ylo - 30, yhi + 30), rng.uniform(xlo - 30, xhi + 30)])
    for _ in range(60):
        j, i = rng.integers(0, Nj), rng.integers(0, Ni)
        yx.append([g["Yt"][j, i], g["Xt"][j, i]])
    for _ in range(60):
        j, i = rng.integers(2, Nj - 2), rng.integers(2, Ni - 2)
        yx.append([g["Yf"][j, i] + rng.normal(0, 0.3), g["Xf"][j, i] + rng.normal(0, 0.3)])
    for _ in range(40):   # ring just around the acceptance radius 0.5*1.2^7*resol of an edge point
        i = rng.integers(3, Ni - 3)
        r = rng.uniform(1.5, 2.1) * g["resol"][0, 0]
        yx.append([g["Yt"][0, i] - r, g["Xt"][0, i]])
    yx = np.array(yx)
    nP = yx.shape[0]
    ll = orc.CartNPSkm2Geo1D(yx)
    # seeds reach SeedInit as float32 promoted to float64 with lon mod 360 (ncio.py:294-309)
    pSG = np.stack([ll[:, 0].astype(np.float32).astype(np.float64),
                    np.mod(ll[:, 1].astype(np.float32).astype(np.float64), 360.)], axis=1)
    pSC = yx.astype(np.float32).astype(np.float64)
    ids = (np.arange(nP) + 1).astype(np.int64) * 7
    # NearestPoint alone (whole-domain form, as called from tracking.py:134)
    npj = np.empty((nP, 2), dtype=np.int64)
    dmin = np.empty(nP)
    with quiet():
        for k in range(nP):
            jy, jx = locate.NearestPoint((pSG[k, 0], pSG[k, 1]), g["latT"], g["lonT"], rd_found_km=tracking.rFoundKM,
                                         resolkm=g["resol"], max_itr=10)
            npj[k] = (jy, jx)
            xd = util.Haversine(pSG[k, 0], pSG[k, 1], g["latT"], g["lonT"])
            dmin[k] = xd.min()
        # plain-radius variants (no 2-D resolution)
        np_plain = np.array([locate.NearestPoint((pSG[k, 0], pSG[k, 1]), g["latT"], g["lonT"], rd_found_km=8.,
                                                 max_itr=5) for k in range(0, nP, 7)], dtype=np.int64)
        # FindContainingCell from the found nearest point
        fcc_ok = np.zeros(nP, dtype=bool); fcc_ji = np.zeros((nP, 2), dtype=np.int64); fcc_v = np.zeros((nP, 2, 4), dtype=np.int64)
        for k in range(nP):
            if npj[k, 0] >= 2 and npj[k, 0] < Nj - 2 and npj[k, 1] >= 2 and npj[k, 1] < Ni - 2:
                ok, ji, vv = locate.FindContainingCell((pSC[k, 0], pSC[k, 1]), (npj[k, 0], npj[k, 1]), g["Yf"], g["Xf"])
                fcc_ok[k] = ok; fcc_ji[k] = ji; fcc_v[k] = np.array(vv)
        out = tracking.SeedInit(ids.copy(), pSG.copy(), pSC.copy(), g["latT"], g["lonT"], g["Yf"], g["Xf"],
                                g["resol"], tmask, xIceConc=sic, iverbose=0)
    nPn, oSG, oSC, oIDs, ojiT, overt, okeep = out
    print("   G5: %d seeds -> %d kept; nearest-not-found %d" % (nP, nPn, (npj[:, 0] < 0).sum()))
    save("g5_seedinit.npz", latT=g["latT"], lonT=g["lonT"], Yf=g["Yf"], Xf=g["Xf"], resol=g["resol"],
         tmask=tmask, sic=sic, ids=ids, pSG=pSG, pSC=pSC, nearest=npj, dmin=dmin, nearest_plain=np_plain,
         fcc_ok=fcc_ok, fcc_ji=fcc_ji, fcc_vert=fcc_v,
         nPn=np.int64(nPn), oSG=oSG, oSC=oSC, oIDs=oIDs, ojiT=np.asarray(ojiT), overt=np.asarray(overt), okeep=okeep)


# --------------------------------------------------------------------------- G5b
def g5b_seedinit_on_points():
    """Seeds EXACTLY on T- and F-points of a curvilinear polar mesh, as the reference's own seeding emits them
    (`tracking.nemoSeed(..., platF, plonF)`, tracking.py:365-442: the grid's lat/lon values themselves), through the
    reference's `SeedInit` / `NearestPoint`.  An F-point is (nearly) equidistant from four T-points: the argmin is decided
    by the last bits of the Haversine distances, `Survive` is evaluated on whichever T-point wins and
    `FindContainingCell` only tries that point's five candidate cells, so kept/cancelled depends on the tie.
    Two meshes (warped: near ties; regular: rounding-level ties) x two seed precisions: as the seeding file stores them
    (float32 promoted to float64, lon mod 360, ncio.py:294-309) and at full precision.
    The seeds' plane coordinates come from the build's forward projection (inputs only: cartopy is absent)."""
    from oracle import oracle as orc
    out = {}
    for tag, warp, (yc, xc) in (("w", 1.0, (-300., 200.)), ("r", 0.0, (-700., -450.)), ("l", None, (0., 0.))):
        Nj, Ni, dkm = 30, 34, 12.0
        if warp is not None:
            g = polar_grid(Nj, Ni, dkm, warp=warp, yc=yc, xc=xc)
            llF = orc.CartNPSkm2Geo1D(np.stack([g["Yf"].ravel(), g["Xf"].ravel()], axis=1))
            latF = np.ascontiguousarray(llF[:, 0].reshape(Nj, Ni))
            lonF = np.ascontiguousarray(np.mod(llF[:, 1], 360.).reshape(Nj, Ni))
        else:
            # a mesh that is REGULAR IN LATITUDE / LONGITUDE with dyadic steps: an F-point sits exactly half a step east of
            # T[j,i] and T[j,i+1] at a common latitude, so the two Haversine distances are EQUAL to the last bit and the
            # reference's argmin (first minimum in C order, locate.py:13-20) decides -- exact two-way ties
            jj, ii = np.meshgrid(np.arange(Nj), np.arange(Ni), indexing="ij")
            latT = 72. + 0.125 * jj; lonT = 10. + 0.5 * ii
            latF = latT + 0.0625; lonF = lonT + 0.25
            def plane(la, lo):
                yx_ = orc.Geo2CartNPSkm1D(np.stack([la.ravel(), lo.ravel()], axis=1))
                return np.ascontiguousarray(yx_[:, 0].reshape(Nj, Ni)), np.ascontiguousarray(yx_[:, 1].reshape(Nj, Ni))
            Yf, Xf = plane(latF, lonF)
            tm = np.ones((Nj, Ni), dtype="i1"); tm[:2] = 0; tm[-2:] = 0; tm[:, :2] = 0; tm[:, -2:] = 0
            g = {"latT": latT, "lonT": lonT, "Yf": Yf, "Xf": Xf, "resol": np.full((Nj, Ni), 19.0), "tmask": tm}
        tmask = g["tmask"].copy()
        tmask[12:15, 8:11] = 0
        sic = np.ones((Nj, Ni)); sic[4:9, 20:28] = 0.02; sic[18:22, 5:9] = 0.93; sic[20, 14] = 0.1
        with quiet():
            seeds = tracking.nemoSeed(tmask, g["latT"], g["lonT"], sic, khss=1, platF=latF, plonF=lonF)
        nP = seeds.shape[0]
        yx = orc.Geo2CartNPSkm1D(seeds)
        ids = (np.arange(nP) + 1).astype(np.int64) * 3
        out.update({tag + "_latT": g["latT"], tag + "_lonT": g["lonT"], tag + "_Yf": g["Yf"], tag + "_Xf": g["Xf"],
                    tag + "_resol": g["resol"], tag + "_tmask": tmask, tag + "_sic": sic, tag + "_latF": latF, tag + "_lonF": lonF,
                    tag + "_ids": ids})
        for prec in ("f4", "f8"):
            if prec == "f4":
                pSG = np.stack([seeds[:, 0].astype(np.float32).astype(np.float64),
                                np.mod(seeds[:, 1].astype(np.float32).astype(np.float64), 360.)], axis=1)
                pSC = yx.astype(np.float32).astype(np.float64)
            else:
                pSG = np.stack([seeds[:, 0], np.mod(seeds[:, 1], 360.)], axis=1)
                pSC = yx.copy()
            npj = np.empty((nP, 2), dtype=np.int64)
            dmin = np.empty(nP); gap = np.empty(nP)
            with quiet():
                for k in range(nP):
                    npj[k] = locate.NearestPoint((pSG[k, 0], pSG[k, 1]), g["latT"], g["lonT"], rd_found_km=tracking.rFoundKM,
                                                 resolkm=g["resol"], max_itr=10)
                    xd = np.sort(util.Haversine(pSG[k, 0], pSG[k, 1], g["latT"], g["lonT"]).ravel())
                    dmin[k] = xd[0]; gap[k] = xd[1] - xd[0]
                res = tracking.SeedInit(ids.copy(), pSG.copy(), pSC.copy(), g["latT"], g["lonT"], g["Yf"], g["Xf"],
                                        g["resol"], tmask, xIceConc=sic, iverbose=0)
            nPn, oSG, oSC, oIDs, ojiT, overt, okeep = res
            k0 = tag + "_" + prec + "_"
            out.update({k0 + "pSG": pSG, k0 + "pSC": pSC, k0 + "nearest": npj, k0 + "dmin": dmin, k0 + "gap": gap,
                        k0 + "nPn": np.int64(nPn), k0 + "oSG": oSG, k0 + "oSC": oSC, k0 + "oIDs": oIDs,
                        k0 + "ojiT": np.asarray(ojiT), k0 + "overt": np.asarray(overt), k0 + "okeep": okeep})
            ntie = int((gap < 1e-9).sum())
            print("   G5b %s/%s: %d seeds (T then F) -> %d kept; %d with the two smallest distances within 1e-9 km, %d exactly equal"
                  % (tag, prec, nP, nPn, ntie, int((gap == 0).sum())))
    save("g5b_seeds_on_points.npz", **out)


# --------------------------------------------------------------------------- G5c
def g5d_nearest_point_local_box():
    """G5d: `NearestPoint` with a previous position (`ji_prv`, `np_box_r`; locate.py:241-245,253-268) -- the local-box variant the
    tracker itself never uses (SeedInit passes no `ji_prv`, tracking.py:134) but the function offers: first pass over the box around
    `ji_prv` (with `resolkm` indexed by the BOX-relative minimum, as the reference does), whole domain from the second pass on.
    On G5's mesh: the true neighbourhood, a previous cell far away (falls back to the whole domain), boxes cut by the domain's
    edges, with and without the 2-D resolution, several box radii and iteration limits, points outside the mesh."""
    g5 = dict(np.load(os.path.join(HERE, "g5_seedinit.npz")))
    latT, lonT, resol, pSG = g5["latT"], g5["lonT"], g5["resol"], g5["pSG"]
    Nj, Ni = latT.shape
    rng = np.random.default_rng(1242)
    cases, outs = [], []
    for k in range(0, len(pSG), 2):
        near = g5["nearest"][k]
        jj, ii = (int(near[0]), int(near[1])) if near[0] >= 0 else (int(rng.integers(0, Nj)), int(rng.integers(0, Ni)))
        mode = int(rng.integers(0, 4))
        if mode == 0:                                   # previous cell a few cells away: found in the box
            prv = (int(np.clip(jj + rng.integers(-4, 5), 0, Nj - 1)), int(np.clip(ii + rng.integers(-4, 5), 0, Ni - 1)))
        elif mode == 1:                                 # far away: the box misses, whole domain from pass 2
            prv = (int(rng.integers(0, Nj)), int(rng.integers(0, Ni)))
        elif mode == 2:                                 # a corner of the domain: the box is cut
            prv = (int(rng.choice([0, 1, Nj - 2, Nj - 1])), int(rng.choice([0, 1, Ni - 2, Ni - 1])))
        else:
            prv = (jj, ii)
        box_r = int(rng.choice([1, 3, 10]))
        max_itr = int(rng.choice([2, 3, 5, 10]))
        use_res = bool(rng.random() < 0.6)
        rd = float(rng.choice([2.5, 8.0, 30.0]))
        with quiet():
            jy, jx = locate.NearestPoint((pSG[k, 0], pSG[k, 1]), latT, lonT, rd_found_km=rd, resolkm=(resol if use_res else []),
                                         ji_prv=prv, np_box_r=box_r, max_itr=max_itr)
        cases.append([k, prv[0], prv[1], box_r, max_itr, int(use_res), rd])
        outs.append([jy, jx])
    outs = np.array(outs, dtype=np.int64)
    print("   G5d: %d cases, %d not found, %d found" % (len(outs), int((outs[:, 0] < 0).sum()), int((outs[:, 0] >= 0).sum())))
    save("g5d_nearest_local.npz", cases=np.array(cases, dtype=np.float64), ji=outs)


def g5c_seedinit_larger_mesh():
    """`SeedInit` / `NearestPoint` of the reference on a mesh large enough for the device search's bounding-sphere hierarchy
    (16 x 16-point blocks, 16 x 16-block superblocks): 300 x 330 T-points around the pole (the pole is INSIDE the mesh: longitudes
    wrap through 0/360 and every meridian is present), 2 600 seeds -- random inside and around the mesh, exactly on T-points, near
    F-points, far outside.  Only the seeds and the reference's outputs are stored; the mesh is rebuilt by the test from its
    parameters with the same code (`polar_grid`, deterministic) and guarded by checksums of its lat/lon arrays."""
    from oracle import oracle as orc
    rng = np.random.default_rng(1242)
    Nj, Ni, dkm, warp, yc, xc = 300, 330, 6.0, 1.0, 40., -25.
    g = polar_grid(Nj, Ni, dkm, warp=warp, yc=yc, xc=xc)
    tmask = g["tmask"].copy(); tmask[100:130, 200:260] = 0; tmask[10:14, 10:60] = 0
    sic = np.ones((Nj, Ni)); sic[200:240, 40:120] = 0.02; sic[150:155, 150:300] = 0.0999
    ylo, yhi = g["Yt"].min(), g["Yt"].max(); xlo, xhi = g["Xt"].min(), g["Xt"].max()
    yx = [np.stack([rng.uniform(ylo - 40, yhi + 40, 1800), rng.uniform(xlo - 40, xhi + 40, 1800)], axis=1)]
    jj, ii = rng.integers(0, Nj, 350), rng.integers(0, Ni, 350)
    yx.append(np.stack([g["Yt"][jj, ii], g["Xt"][jj, ii]], axis=1))
    jj, ii = rng.integers(2, Nj - 2, 350), rng.integers(2, Ni - 2, 350)
    yx.append(np.stack([g["Yf"][jj, ii] + rng.normal(0, 0.2, 350), g["Xf"][jj, ii] + rng.normal(0, 0.2, 350)], axis=1))
    yx.append(np.array([[yhi + 600., 0.], [0., xlo - 900.], [ylo - 3000., xhi + 3000.]] + [[0.001 * k, -0.002 * k] for k in range(97)]))   # far away; around the pole
    yx = np.concatenate(yx)
    nP = yx.shape[0]
    ll = orc.CartNPSkm2Geo1D(yx)
    pSG = np.stack([ll[:, 0].astype(np.float32).astype(np.float64), np.mod(ll[:, 1].astype(np.float32).astype(np.float64), 360.)], axis=1)
    pSC = yx.astype(np.float32).astype(np.float64)
    ids = (np.arange(nP) + 1).astype(np.int64) * 11
    npj = np.empty((nP, 2), dtype=np.int64)
    with quiet():
        for k in range(nP):
            npj[k] = locate.NearestPoint((pSG[k, 0], pSG[k, 1]), g["latT"], g["lonT"], rd_found_km=tracking.rFoundKM,
                                         resolkm=g["resol"], max_itr=10)
        out = tracking.SeedInit(ids.copy(), pSG.copy(), pSC.copy(), g["latT"], g["lonT"], g["Yf"], g["Xf"], g["resol"], tmask,
                                xIceConc=sic, iverbose=0)
    nPn, oSG, oSC, oIDs, ojiT, overt, okeep = out
    print("   G5c: %d seeds -> %d kept; nearest-not-found %d" % (nP, nPn, (npj[:, 0] < 0).sum()))
    save("g5c_seedinit_300x330.npz", mesh=np.array([Nj, Ni, dkm, warp, yc, xc]), lat_sum=np.float64(g["latT"].sum()),
         lon_sum=np.float64(g["lonT"].sum()), lat_probe=g["latT"][::37, ::41].copy(), lon_probe=g["lonT"][::37, ::41].copy(),
         tmask_boxes=np.array([[100, 130, 200, 260], [10, 14, 10, 60]]), sic_boxes=np.array([[200, 240, 40, 120], [150, 155, 150, 300]]),
         sic_vals=np.array([0.02, 0.0999]), ids=ids, pSG=pSG, pSC=pSC, nearest=npj,
         nPn=np.int64(nPn), oIDs=oIDs, ojiT=np.asarray(ojiT), okeep=okeep)


# --------------------------------------------------------------------------- G6
def reference_loop(g, tmask, u, v, sic, yx0, jiT0, vert0, rec_first, rec_last, kstrt, Nt, rdt, strategy):
    """Drives the reference predicates in the order of si3_part_tracker.py:361-496.

    Field record jrec uses slab jrec % K.  Returns per-record positions (FillValue
    where a buoy did not step), masks, and the final index/alive state.
    """
    Yf, Xf, Yu, Xu, Yv, Xv = (g[k] for k in ("Yf", "Xf", "Yu", "Xu", "Yv", "Xv"))
    K = u.shape[0]
    nP = yx0.shape[0]
    alive = np.ones(nP, dtype="i1")
    msk = np.zeros((Nt + 1, nP), dtype="i1")
    pos = np.zeros((Nt + 1, nP, 2)) + FILL
    jit_rec = np.zeros((Nt + 1, nP, 2), dtype=np.int64)
    alive_rec = np.zeros((Nt + 1, nP), dtype="i1")
    mesh = np.zeros((nP, 4, 2))
    still = np.zeros(nP, dtype=bool)
    jiT = jiT0.copy(); vert = vert0.copy()
    for b in range(nP):
        k0 = rec_first[b] - kstrt
        pos[k0, b, :] = yx0[b]
        msk[k0, b] = 1
    jit_rec[0] = jiT; alive_rec[0] = alive
    codes = np.zeros(9, dtype=np.int64)
    xIC = np.zeros_like(Yf); xU = np.zeros_like(Yf); xV = np.zeros_like(Yf)
    for jt in range(Nt):
        jrec = jt + kstrt
        xIC[:, :] = sic[jrec % K]; xU[:, :] = u[jrec % K]; xV[:, :] = v[jrec % K]
        for b in range(nP):
            if alive[b] == 1 and jrec >= rec_first[b] and jrec <= rec_last[b]:
                ry, rx = pos[jt, b, :]
                if not still[b]:
                    vj, vi = vert[b]
                    mesh[b] = [[Yf[vj[c], vi[c]], Xf[vj[c], vi[c]]] for c in range(4)]
                jT, iT = jiT[b]
                if strategy == 0:
                    zU = 0.5 * (xU[jT, iT] + xU[jT, iT - 1])
                    zV = 0.5 * (xV[jT, iT] + xV[jT - 1, iT])
                else:
                    F = [Yf[jT, iT], Xf[jT, iT]]
                    um1 = tracking.intersect2Seg([ry, rx], F, [Yv[jT - 1, iT], Xv[jT - 1, iT]], [Yv[jT, iT], Xv[jT, iT]])
                    vm1 = tracking.intersect2Seg([ry, rx], F, [Yu[jT, iT - 1], Xu[jT, iT - 1]], [Yu[jT, iT], Xu[jT, iT]])
                    zU = xU[jT, iT - 1] if um1 else xU[jT, iT]
                    zV = xV[jT - 1, iT] if vm1 else xV[jT, iT]
                dx = zU * rdt
                dy = zV * rdt
                rxn = rx + dx / 1000.
                ryn = ry + dy / 1000.
                pos[jt + 1, b, :] = [ryn, rxn]
                msk[jt + 1, b] = 1
                lin = locate.IsInsideQuadrangle(ryn, rxn, mesh[b])
                still[b] = lin
                if not lin:
                    ic = tracking.CrossedEdge([ry, rx], [ryn, rxn], vert[b], Yf, Xf)
                    nh = tracking.NewHostCell(ic, [ry, rx], [ryn, rxn], vert[b], Yf, Xf)
                    codes[nh] += 1
                    vert[b], jiT[b] = tracking.UpdtInd4NewCell(nh, vert[b], jiT[b])
                    if tracking.Survive(0, jiT[b], tmask, pIceC=xIC) > 0:
                        alive[b] = 0
        jit_rec[jt + 1] = jiT; alive_rec[jt + 1] = alive
    return pos, msk, jit_rec, alive_rec, vert, codes


def g6_trajectories():
    rng = np.random.default_rng(1239)
    Nj, Ni, dkm = 48, 48, 4.0
    nP, Nt, K, kstrt, rdt = 128, 64, 8, 3, 3600.
    for tag, warp in (("curvi", 1.0), ("regular", 0.0)):
        g = syn.make_grid(Nj, Ni, dkm=dkm, warp=warp)
        u, v, sic = syn.make_fields(g, K=K, seed=2024, umax=0.75, drift=0.25, ripple=0.12)
        tmask = g["tmask"].copy()
        tmask[30:34, 28:33] = 0                    # island
        sic = sic.copy()
        for k in range(K):
            sic[k, 8:14, 6 + k:14 + k] = 0.04      # drifting open-water patch
        _, cand = syn.make_buoys(g, 2 * nP, seed=1234, frac=0.72)
        if warp == 0.0:
            # put some buoys exactly on cell edges / vertices of the regular grid
            cand[:8, 0] = g["Yf"][20:28, 0]
            cand[4:12, 1] = g["Xf"][0, 18:26]
        # host cells with the reference's own FindContainingCell, from the nearest T-point in the plane
        yx0 = np.zeros((nP, 2)); jiT0 = np.zeros((nP, 2), dtype=np.int64); vert0 = np.zeros((nP, 2, 4), dtype=np.int64)
        b = 0
        for c in cand:
            d2 = (g["Yt"] - c[0]) ** 2 + (g["Xt"] - c[1]) ** 2
            gj, gi = np.unravel_index(np.argmin(d2), d2.shape)
            ok, ji, vv = locate.FindContainingCell((c[0], c[1]), (gj, gi), g["Yf"], g["Xf"])
            if ok and b < nP:
                yx0[b] = c; jiT0[b] = ji; vert0[b] = np.array(vv); b += 1
        assert b == nP, (tag, b)
        ids = np.arange(1, nP + 1, dtype=np.int64) * 3 + 300534062025510   # IDs reach 3e14 (tools/sidfexloc.dat:1)
        rec_first = np.full(nP, kstrt, dtype=np.int64)
        rec_last = np.full(nP, kstrt + Nt - 1, dtype=np.int64)
        late = rng.choice(nP, 12, replace=False)
        rec_first[late] = kstrt + rng.integers(1, 20, 12)
        early = rng.choice(nP, 12, replace=False)
        rec_last[early] = kstrt + Nt - 1 - rng.integers(1, 20, 12)
        out = {}
        for strat in (1, 0):
            with quiet():
                pos, msk, jit_rec, alive_rec, vert, codes = reference_loop(
                    g, tmask, u.astype(np.float64), v.astype(np.float64), sic.astype(np.float64),
                    yx0, jiT0, vert0, rec_first, rec_last, kstrt, Nt, rdt, strat)
            print("   G6 %-8s strat %d: crossings by code %s  dead %d/%d" %
                  (tag, strat, codes[1:].tolist(), int((alive_rec[-1] == 0).sum()), nP))
            out["pos_s%d" % strat] = pos
            out["msk_s%d" % strat] = msk
            out["jiT_s%d" % strat] = jit_rec.astype(np.int32)
            out["alive_s%d" % strat] = alive_rec
            out["vert_s%d" % strat] = vert
            out["codes_s%d" % strat] = codes
        save("g6_traj_%s.npz" % tag, warp=np.float64(warp), dkm=np.float64(dkm), Nj=np.int64(Nj), Ni=np.int64(Ni),
             tmask=tmask, u=u, v=v, sic=sic, ids=ids, yx0=yx0, jiT0=jiT0, vert0=vert0,
             rec_first=rec_first, rec_last=rec_last, kstrt=np.int64(kstrt), Nt=np.int64(Nt), rdt=np.float64(rdt), **out)


def traj_digest(pos, msk, jit, alive):
    """per-record digests of a trajectory set: XOR of the bit patterns of the positions of the buoys that stepped, their
    number, a checksum of the host cells, the number alive -- bit-exact comparison without storing (Nt, nP, 2) arrays"""
    Nt1 = pos.shape[0]
    dg = np.zeros((Nt1, 4), dtype=np.uint64)
    for k in range(Nt1):
        m = msk[k] == 1
        bits = pos[k][m].reshape(-1).view(np.uint64)
        dg[k, 0] = np.bitwise_xor.reduce(bits) if bits.size else 0
        dg[k, 1] = int(m.sum())
        dg[k, 2] = int((jit[k].astype(np.int64) * np.array([100003, 1])).sum() % (1 << 62))
        dg[k, 3] = int((alive[k] == 1).sum())
    return dg


def g6b_fast_flow_trajectories():
    """A larger trajectory set from the restated reference loop: 120 x 140 warped grid, 1 500 buoys, 90 records of a FAST flow
    (up to two cells per record: `CrossedEdge` falls through to 4 and `NewHostCell` puts buoys into cells that do not contain
    them -- the reference's own behaviour, which the build must reproduce), islands, drifting open water, both velocity rules.
    Stored: the initial state, the final state and per-record digests (traj_digest); grid and fields are rebuilt by the test
    from their parameters with sitrack_amd.synthetic and guarded by checksums."""
    Nj, Ni, dkm, warp = 120, 140, 4.0, 1.0
    nP, Nt, K, kstrt, rdt = 1500, 90, 6, 2, 3600.
    g = syn.make_grid(Nj, Ni, dkm=dkm, warp=warp)
    u, v, sic = syn.make_fields(g, K=K, seed=777, umax=2.2, drift=0.6, ripple=0.2)
    tmask = g["tmask"].copy()
    tmask[40:46, 60:70] = 0; tmask[80:83, 20:50] = 0
    sic = sic.copy()
    for k in range(K):
        sic[k, 20:30, 90 + 2 * k:105 + 2 * k] = 0.03
    _, cand = syn.make_buoys(g, 2 * nP, seed=4321, frac=0.8)
    yx0 = np.zeros((nP, 2)); jiT0 = np.zeros((nP, 2), dtype=np.int64); vert0 = np.zeros((nP, 2, 4), dtype=np.int64)
    b = 0
    for c in cand:
        d2 = (g["Yt"] - c[0]) ** 2 + (g["Xt"] - c[1]) ** 2
        gj, gi = np.unravel_index(np.argmin(d2), d2.shape)
        if not (2 <= gj < Nj - 2 and 2 <= gi < Ni - 2):
            continue
        ok, ji, vv = locate.FindContainingCell((c[0], c[1]), (gj, gi), g["Yf"], g["Xf"])
        if ok and b < nP and tmask[ji[0], ji[1]] == 1:
            yx0[b] = c; jiT0[b] = ji; vert0[b] = np.array(vv); b += 1
    assert b == nP, b
    rec_first = np.full(nP, kstrt, dtype=np.int64); rec_last = np.full(nP, kstrt + Nt - 1, dtype=np.int64)
    out = {}
    for strat in (1, 0):
        with quiet():
            pos, msk, jit_rec, alive_rec, vert, codes = reference_loop(
                g, tmask, u.astype(np.float64), v.astype(np.float64), sic.astype(np.float64),
                yx0, jiT0, vert0, rec_first, rec_last, kstrt, Nt, rdt, strat)
        print("   G6b strat %d: crossings by code %s  dead %d/%d" % (strat, codes[1:].tolist(), int((alive_rec[-1] == 0).sum()), nP))
        last = np.full((nP, 2), FILL)
        for k in range(Nt + 1):
            m = msk[k] == 1
            last[m] = pos[k][m]
        out["digest_s%d" % strat] = traj_digest(pos, msk, jit_rec, alive_rec)
        out["last_pos_s%d" % strat] = last
        out["jiT_end_s%d" % strat] = jit_rec[-1].astype(np.int32)
        out["alive_end_s%d" % strat] = alive_rec[-1]
        out["codes_s%d" % strat] = codes
    save("g6b_traj_fast.npz", mesh=np.array([Nj, Ni, dkm, warp]), fields=np.array([K, 777, 2.2, 0.6, 0.2]),
         tmask_boxes=np.array([[40, 46, 60, 70], [80, 83, 20, 50]]), u_sum=np.float64(u.astype(np.float64).sum()),
         sic_sum=np.float64(sic.astype(np.float64).sum()), yx0=yx0, jiT0=jiT0.astype(np.int32),
         kstrt=np.int64(kstrt), Nt=np.int64(Nt), rdt=np.float64(rdt), **out)


PROBE_RECORDS = [0, 5, 10, 15, 20, 25, 30, 31]       # records whose exact fp64 sums guard the rebuilt BASELINE fields (G6c / G6d)


def g6cd_baseline_cuts(configs=("c2", "c3"), nP=1000, Nt=100):
    """G6c / G6d: REFERENCE trajectories on the BASELINE workloads themselves.  The first 10^3 buoys x 100
    records of the real C2 (512^2, 1e5 buoys) and C3 (4096^2, 1e7 buoys) inputs of bench.py -- same grid, same buoy seed
    default_rng(1234), same 32 resident records of default_rng(2024) cycled -- through the restated reference loop, both
    velocity rules, once with every buoy stepping every record and once with 12 late starters / 12 early stoppers like G6
    (per-buoy record windows, si3_part_tracker.py:264-318,380).  Stored: host cells at the start, windows, final state and
    per-record digests (traj_digest); positions, grid and fields are rebuilt by the tests from the seeds and guarded by
    checksums.  Until round 3 the BASELINE grids were held against the oracle only (tools/time_reference_loop.py threw these
    trajectories away)."""
    from sitrack_amd.tracking import vertices_of
    shapes = {"c2": (512, 512, 100_000, "g6c_c2cut.npz"), "c3": (4096, 4096, 10_000_000, "g6d_c3cut.npz")}
    K, kstrt, rdt = 32, 0, 3600.
    for cfg in configs:
        Nj, Ni, nAll, fname = shapes[cfg]
        g = syn.make_grid(Nj, Ni, dkm=4.0, warp=0.0)
        _, yx = syn.make_buoys(g, nAll, seed=1234, frac=0.6)
        yx0 = np.ascontiguousarray(yx[:nP])
        del yx
        jiT0 = syn.regular_host_cell(g, yx0).astype(np.int64)
        # the reference's own FindContainingCell agrees with the analytic host cell on every buoy of the cut
        for b in range(0, nP, 7):
            ok, ji, _ = locate.FindContainingCell((yx0[b, 0], yx0[b, 1]), (int(jiT0[b, 0]), int(jiT0[b, 1])), g["Yf"], g["Xf"])
            assert ok and tuple(ji) == tuple(jiT0[b]), (cfg, b)
        vert0 = vertices_of(jiT0)
        u, v, sic = syn.make_fields(g, K=K, seed=2024, umax=0.3, drift=0.05)
        rng = np.random.default_rng(1239)
        wf = np.full(nP, kstrt, dtype=np.int64); wl = np.full(nP, kstrt + Nt - 1, dtype=np.int64)
        late = rng.choice(nP, 12, replace=False); wf[late] = kstrt + rng.integers(1, 20, 12)
        early = rng.choice(nP, 12, replace=False); wl[early] = kstrt + Nt - 1 - rng.integers(1, 20, 12)
        all_f = np.full(nP, kstrt, dtype=np.int64); all_l = np.full(nP, 10**9, dtype=np.int64)
        out = {}
        for strat in (1, 0):
            for tag, (rf, rl) in (("", (all_f, all_l)), ("w", (wf, wl))):
                with quiet():
                    pos, msk, jit_rec, alive_rec, vert, codes = reference_loop(g, g["tmask"], u, v, sic, yx0, jiT0, vert0, rf, rl, kstrt, Nt, rdt, strat)
                last = np.full((nP, 2), FILL)
                for k in range(Nt + 1):
                    m = msk[k] == 1
                    last[m] = pos[k][m]
                key = "s%d%s" % (strat, tag)
                out["digest_" + key] = traj_digest(pos, msk, jit_rec, alive_rec)
                out["last_pos_" + key] = last
                out["jiT_end_" + key] = jit_rec[-1].astype(np.int32)
                out["alive_end_" + key] = alive_rec[-1]
                out["codes_" + key] = codes
                print("   %s strat %d %-8s: %d particle-steps, crossings by code %s, dead %d/%d" %
                      (fname, strat, "windows" if tag else "all", int(msk[1:].sum()), codes[1:].tolist(), int((alive_rec[-1] == 0).sum()), nP))
        save(fname, mesh=np.array([Nj, Ni, 4.0, 0.0]), buoys=np.array([nAll, 1234, nP]), fields=np.array([K, 2024, 0.3, 0.05]),
             probe_records=np.array(PROBE_RECORDS), u_sum=np.array([u[k].astype(np.float64).sum() for k in PROBE_RECORDS]),
             v_sum=np.array([v[k].astype(np.float64).sum() for k in PROBE_RECORDS]), sic_sum=np.float64(sic.sum(dtype=np.float64)),
             yx0_sum=np.float64(yx0.sum()),
             jiT0=jiT0.astype(np.int32), rec_first=wf, rec_last=wl, kstrt=np.int64(kstrt), Nt=np.int64(Nt), rdt=np.float64(rdt), **out)
        del g, u, v, sic


def g6ef_curvilinear_cuts(configs=("c3warp", "c5shape"), nP=1000, Nt=100):
    """G6e / G6f: REFERENCE trajectories on the two CURVILINEAR workloads of bench.py (round 4: `--warp 1.0`, `--config c5shape`),
    cut like G6c / G6d to the first 10^3 buoys x 100 records, both velocity rules, with and without record windows.
      G6e  C3 on the sheared / stretched mesh (warp 1): the buoys of default_rng(1234) in bench.py's order; their host cells from the
           reference's own FindContainingCell (guess: synthetic.nearest_t_index), the first 10^3 it finds.
      G6f  the NANUK4-shaped 566 x 492 mesh at 12.5 km with an island and a polynya, flow up to 0.9 m/s: the first candidate seeds of
           default_rng(1234) (float32 lat/lon + km like a seeding file) through the reference's own SeedInit (NearestPoint by Haversine,
           Survive, FindContainingCell: tracking.py:98-178), the first 10^3 it keeps; buoys that drift into the open water die.
    Stored: which candidates the buoys are, their host cells, windows, final state, per-record digests; everything else is rebuilt
    by the tests from the seeds (conftest.baseline_cut_case) and guarded by checksums."""
    from oracle import oracle as orc
    from sitrack_amd.tracking import vertices_of
    K, kstrt, rdt = 32, 0, 3600.
    for cfg in configs:
        extra = {}
        if cfg == "c3warp":
            Nj, Ni, nAll, fname = 4096, 4096, 10_000_000, "g6e_c3warp_cut.npz"
            g = syn.make_grid(Nj, Ni, dkm=4.0, warp=1.0)
            tmask = g["tmask"]
            _, yx = syn.make_buoys(g, nAll, seed=1234, frac=0.6)
            ncand = nP + 50
            cand = np.ascontiguousarray(yx[:ncand]); del yx
            guess = syn.nearest_t_index(g, cand)
            idx, jis = [], []
            with quiet():
                for b in range(ncand):
                    ok, ji, _ = locate.FindContainingCell((cand[b, 0], cand[b, 1]), (int(guess[b, 0]), int(guess[b, 1])), g["Yf"], g["Xf"])
                    if ok:
                        idx.append(b); jis.append(ji)
                    if len(idx) == nP:
                        break
            assert len(idx) == nP
            cand_idx, jiT0 = np.array(idx, dtype=np.int32), np.array(jis, dtype=np.int64)
            yx0 = np.ascontiguousarray(cand[cand_idx])
            fpar = dict(seed=2024, umax=0.3, drift=0.05)
            u, v, sic = syn.make_fields(g, K=K, **fpar)
            mesh = np.array([Nj, Ni, 4.0, 1.0, 0.0, 0.0])
            fields = np.array([K, 2024, 0.3, 0.05, 0.0])
        else:
            Nj, Ni, nAll, fname = 566, 492, 11_300_000, "g6f_c5shape_cut.npz"
            g = syn.shift_grid(syn.make_grid(Nj, Ni, dkm=12.5, warp=1.0), -250., 150.)
            tmask = g["tmask"]
            isl = (Nj // 3, Nj // 3 + Nj // 12, Ni // 2, Ni // 2 + Ni // 10)
            pol = (Nj // 2, Nj // 2 + Nj // 10, Ni // 5, Ni // 5 + Ni // 6)
            tmask[isl[0]:isl[1], isl[2]:isl[3]] = 0
            rng = np.random.default_rng(1234)
            yx = np.stack([rng.uniform(g["Yt"].min() + 30, g["Yt"].max() - 30, nAll), rng.uniform(g["Xt"].min() + 30, g["Xt"].max() - 30, nAll)], axis=1)
            ncand = nP + 200
            cand = yx[:ncand].astype(np.float32).astype(np.float64); del yx
            ll = orc.CartNPSkm2Geo1D(np.stack([g["Yt"].ravel(), g["Xt"].ravel()], axis=1))
            latT, lonT = np.ascontiguousarray(ll[:, 0].reshape(Nj, Ni)), np.ascontiguousarray(np.mod(ll[:, 1], 360.).reshape(Nj, Ni))
            sll = orc.CartNPSkm2Geo1D(cand); sll[:, 1] = np.mod(sll[:, 1], 360.)
            sic0 = np.ones((Nj, Ni)); sic0[pol[0]:pol[1], pol[2]:pol[3]] = 0.03
            ids = np.arange(1, ncand + 1, dtype=np.int64)
            with quiet():
                nPn, oSG, oSC, oIDs, ojiT, overt, okeep = tracking.SeedInit(ids.copy(), sll.copy(), cand.copy(), latT, lonT, g["Yf"], g["Xf"],
                                                                           g["resol"], tmask, xIceConc=sic0, iverbose=0)
            assert nPn >= nP, nPn
            cand_idx = np.asarray(okeep, dtype=np.int32)[:nP]
            jiT0 = np.asarray(ojiT, dtype=np.int64)[:nP]
            assert np.array_equal(np.asarray(oIDs)[:nP], ids[cand_idx])
            yx0 = np.ascontiguousarray(cand[cand_idx])
            fpar = dict(seed=77, umax=0.9, drift=0.3, ripple=0.1)
            u, v, sic = syn.make_fields(g, K=K, **fpar)
            sic[:, pol[0]:pol[1], pol[2]:pol[3]] = 0.03
            mesh = np.array([Nj, Ni, 12.5, 1.0, -250., 150.])
            fields = np.array([K, 77, 0.9, 0.3, 0.1])
            extra = dict(island=np.array(isl), polynya=np.array(pol), seed_cancelled=np.int64(int(cand_idx[-1]) + 1 - nP),
                         lat_sum=np.float64(latT.sum()), lon_sum=np.float64(lonT.sum()))
            print("   %s: SeedInit kept %d of the first %d candidates (%d cancelled before the %d-th kept)"
                  % (fname, nPn, ncand, int(extra["seed_cancelled"]), nP))
        vert0 = vertices_of(jiT0)
        rng = np.random.default_rng(1239)
        wf = np.full(nP, kstrt, dtype=np.int64); wl = np.full(nP, kstrt + Nt - 1, dtype=np.int64)
        late = rng.choice(nP, 12, replace=False); wf[late] = kstrt + rng.integers(1, 20, 12)
        early = rng.choice(nP, 12, replace=False); wl[early] = kstrt + Nt - 1 - rng.integers(1, 20, 12)
        all_f = np.full(nP, kstrt, dtype=np.int64); all_l = np.full(nP, 10**9, dtype=np.int64)
        out = {}
        for strat in (1, 0):
            for tag, (rf, rl) in (("", (all_f, all_l)), ("w", (wf, wl))):
                with quiet():
                    pos, msk, jit_rec, alive_rec, vert, codes = reference_loop(g, tmask, u, v, sic, yx0, jiT0, vert0, rf, rl, kstrt, Nt, rdt, strat)
                last = np.full((nP, 2), FILL)
                for k in range(Nt + 1):
                    m = msk[k] == 1
                    last[m] = pos[k][m]
                key = "s%d%s" % (strat, tag)
                out["digest_" + key] = traj_digest(pos, msk, jit_rec, alive_rec)
                out["last_pos_" + key] = last
                out["jiT_end_" + key] = jit_rec[-1].astype(np.int32)
                out["alive_end_" + key] = alive_rec[-1]
                out["codes_" + key] = codes
                print("   %s strat %d %-8s: %d particle-steps, crossings by code %s, dead %d/%d" %
                      (fname, strat, "windows" if tag else "all", int(msk[1:].sum()), codes[1:].tolist(), int((alive_rec[-1] == 0).sum()), nP))
        save(fname, kind=np.array(cfg), mesh=mesh, buoys=np.array([nAll, 1234, nP]), fields=fields,
             probe_records=np.array(PROBE_RECORDS), u_sum=np.array([u[k].astype(np.float64).sum() for k in PROBE_RECORDS]),
             v_sum=np.array([v[k].astype(np.float64).sum() for k in PROBE_RECORDS]), sic_sum=np.float64(sic.sum(dtype=np.float64)),
             yx0_sum=np.float64(yx0.sum()), cand_idx=cand_idx,
             jiT0=jiT0.astype(np.int32), rec_first=wf, rec_last=wl, kstrt=np.int64(kstrt), Nt=np.int64(Nt), rdt=np.float64(rdt), **extra, **out)
        del g, u, v, sic


# --------------------------------------------------------------------------- G7
def h5_values(path, name, fmt):
    txt = subprocess.run(["/opt/conda/bin/h5dump", "-m", fmt, "-d", name, path], check=True,
                         capture_output=True, text=True).stdout
    body = txt.split("DATA {", 1)[1].split("}", 1)[0]
    body = re.sub(r"\(\d+(,\d+)*\):", " ", body)
    return [t for t in re.split(r"[,\s]+", body) if t]


def g7_projection():
    f = "/root/reference/tools/nc/sitrack_seeding_sidfex_19961215_00_HSS5.nc__KEEP"
    lat = np.array(h5_values(f, "latitude", "%.9g"), dtype=np.float32)
    lon = np.array(h5_values(f, "longitude", "%.9g"), dtype=np.float32)
    ypos = np.array(h5_values(f, "y_pos", "%.9g"), dtype=np.float32)
    xpos = np.array(h5_values(f, "x_pos", "%.9g"), dtype=np.float32)
    ids = np.array(h5_values(f, "id_buoy", "%ld"), dtype=np.int64)
    tim = np.array(h5_values(f, "time", "%d"), dtype=np.int64)
    # the text file the seeding tool read (tools/sidfexloc.dat: id lon lat)
    dat = np.genfromtxt("/root/reference/tools/sidfexloc.dat")
    print("   G7 ids", ids.tolist())
    save("g7_projection.npz", id_buoy=ids, time=tim, latitude=lat, longitude=lon, y_pos=ypos, x_pos=xpos,
         dat_lonlat=dat[:, 1:3], dat_id=dat[:, 0].astype(np.int64))


# --------------------------------------------------------------------------- G8
def g8_timespan():
    cases = []
    base = 850608000           # 1996-12-15 00:00 UTC (the fixture's time)
    vt = base + 1800 + 3600 * np.arange(48)
    for sd, stop in ((base, None), (base + 3600, None), (base + 5 * 3600 + 1800, None), (base, base + 10 * 3600),
                     (base + 7200, base + 30 * 3600 + 1799), (base + 1800, vt[-1]), (base + 3 * 3600, base + 3 * 3600 + 1800)):
        with quiet():
            r = tracking.GetTimeSpan(3600., vt, sd, vt[0], vt[-1], iStop=stop)
        cases.append([sd, -1 if stop is None else stop] + [int(x) for x in r])
    save("g8_timespan.npz", vtime=vt.astype(np.int64), cases=np.array(cases, dtype=np.int64))


# --------------------------------------------------------------------------- G9
def g9_haversine():
    rng = np.random.default_rng(1240)
    n = 4000
    plat = rng.uniform(55, 90, n); plon = rng.uniform(0, 360, n)
    xlat = np.clip(plat + rng.normal(0, 2, n), -90, 90); xlon = np.mod(plon + rng.normal(0, 5, n), 360.)
    xlat[:50] = plat[:50]; xlon[:50] = plon[:50]
    d = np.array([util.Haversine(plat[k], plon[k], xlat[k:k + 1], xlon[k:k + 1])[0] for k in range(n)])
    save("g9_haversine.npz", plat=plat, plon=plon, xlat=xlat, xlon=xlon, dist=d)


# --------------------------------------------------------------------------- G10
def g10_nemoseed():
    rng = np.random.default_rng(1241)
    Nj, Ni = 37, 41
    lat = 50. + 40. * rng.random((Nj, Ni))
    lon = 360. * rng.random((Nj, Ni))
    latF = lat + 0.05; lonF = lon + 0.07
    tmask = (rng.random((Nj, Ni)) > 0.1).astype('i1')
    sic = rng.choice([0.0, 0.5, 0.89, 0.9, 0.95, 1.0], size=(Nj, Ni))
    rmask = (rng.random((Nj, Ni)) > 0.3).astype('i1')
    out = {}
    for tag, kw in (("a", dict(khss=1)), ("b", dict(khss=3)), ("c", dict(khss=2, fmsk_rstrct=rmask)),
                    ("d", dict(khss=1, platF=latF, plonF=lonF)), ("e", dict(khss=4, fmsk_rstrct=rmask, platF=latF, plonF=lonF))):
        with quiet():
            out["seed_" + tag] = tracking.nemoSeed(tmask, lat, lon, sic, **kw)
    save("g10_nemoseed.npz", lat=lat, lon=lon, latF=latF, lonF=lonF, tmask=tmask, sic=sic, rmask=rmask, **out)


if __name__ == "__main__":
    only = sys.argv[1:]
    for name, fn in (("g1", g1_inside), ("g2", g2_intersect), ("g3", g3_crossing), ("g4", g4_survive), ("g4b", g4b_survive_wide),
                     ("g5", g5_seedinit), ("g5b", g5b_seedinit_on_points), ("g5c", g5c_seedinit_larger_mesh), ("g5d", g5d_nearest_point_local_box), ("g6", g6_trajectories), ("g6b", g6b_fast_flow_trajectories), ("g6cd", g6cd_baseline_cuts), ("g6ef", g6ef_curvilinear_cuts),
                     ("g7", g7_projection), ("g8", g8_timespan),
                     ("g9", g9_haversine), ("g10", g10_nemoseed)):
        if not only or name in only:
            fn()
