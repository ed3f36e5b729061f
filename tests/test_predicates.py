"""The reference's scalar predicates under their own names (sitrack_amd/predicates.py), called the way the reference's
loop body and its own test call them, against the golden vectors produced by the reference functions themselves."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sit():
    import sitrack_amd
    return sitrack_amd


def test_pnt_inside_quad_like_the_reference_test(sit):
    # reference tools/tests/test_pnt_inside_quad.py:16-24
    quad = np.array([[0., 0.], [3., 0.], [4., 4.], [1., 3.5]])
    got = [sit.IsInsideQuadrangle(y, x, quad) for (y, x) in ((2., 2.), (6., 6.), (-1., 2.), (3.1, 3.6))]
    assert got == [True, False, False, True]


def test_inside_and_intersect_scalars(sit, golden):
    g1, g2 = golden("g1_inside.npz"), golden("g2_intersect.npz")
    for k in range(0, len(g1["pts"]), 40):
        assert sit.IsInsideQuadrangle(g1["pts"][k, 0], g1["pts"][k, 1], g1["quads"][k]) == bool(g1["inside"][k])
    for k in range(0, len(g2["P"]), 70):
        A, B, Cc, D = g2["P"][k]
        assert sit.intersect2Seg(A, B, Cc, D) == bool(g2["intersect"][k])
        assert sit._ccw_(A, B, Cc) == bool(g2["ccw"][k])


def test_crossing_chain_like_the_loop_body(sit, golden):
    """si3_part_tracker.py:474-480: icross = CrossedEdge(...); inhc = NewHostCell(icross, ...); UpdtInd4NewCell(inhc, ...)"""
    g = golden("g3_crossing.npz")
    Yf, Xf = g["Yf"], g["Xf"]
    seen = set()
    for k in range(0, len(g["P1"]), 6):
        vert = g["vert"][k].copy()
        jiT = g["jiT"][k].copy()
        icross = sit.CrossedEdge(g["P1"][k], g["P2"][k], vert, Yf, Xf)
        inhc = sit.NewHostCell(icross, g["P1"][k], g["P2"][k], vert, Yf, Xf)
        assert (icross, inhc) == (int(g["icross"][k]), int(g["inhc"][k])), k
        v2, j2 = sit.UpdtInd4NewCell(inhc, vert, jiT)
        assert v2 is vert and j2 is jiT                                  # in place, like the reference
        assert np.array_equal(vert, g["vert_out"][k]) and np.array_equal(jiT, g["jiT_out"][k])
        seen.add(inhc)
    assert seen == set(range(1, 9))
    with pytest.raises(SystemExit):
        sit.UpdtInd4NewCell(9, g["vert"][0].copy(), g["jiT"][0].copy())


def test_survive_scalar(sit, golden, capsys):
    g = golden("g4_survive.npz")
    for k in range(0, len(g["jiT"]), 3):
        assert sit.Survive(k, g["jiT"][k], g["tmask"], g["sic"]) == int(g["kill_a"][k]), g["jiT"][k]
    assert set(np.unique(g["kill_a"])) == {0, 1}
    # float32 ice field, as read from an SI3 file
    for k in range(1, len(g["jiT"]), 17):
        assert sit.Survive(k, g["jiT"][k], g["tmask"], g["sic32"]) == int(g["kill_b"][k])
    # no 2-D ice field: the rim and the mask still kill; a buoy that passes them trips the reference's own failure
    alive = np.flatnonzero(g["kill_a"] == 0)[0]
    with pytest.raises(UnboundLocalError):
        sit.Survive(7, g["jiT"][alive], g["tmask"])
    assert sit.Survive(7, [0, 5], g["tmask"]) == 1
    assert sit.Survive(7, [0, 5], g["tmask"], iverbose=1) == 1 and "CANCEL buoy 7" in capsys.readouterr().out
    with pytest.raises(IndexError):
        sit.Survive(7, [g["tmask"].shape[0], 5], g["tmask"], g["sic"])


def test_nearest_point_and_haversine_scalars(sit, golden):
    """locate.NearestPoint as SeedInit calls it (tracking.py:134) and with a plain radius; util.Haversine."""
    g = golden("g5_seedinit.npz")
    for k in range(0, len(g["ids"]), 5):
        jy, jx = sit.NearestPoint((g["pSG"][k, 0], g["pSG"][k, 1]), g["latT"], g["lonT"], rd_found_km=sit.rFoundKM,
                                  resolkm=g["resol"], max_itr=10)
        assert (jy, jx) == tuple(g["nearest"][k]), k
    plain = np.array([sit.NearestPoint((g["pSG"][k, 0], g["pSG"][k, 1]), g["latT"], g["lonT"], rd_found_km=8., max_itr=5)
                      for k in range(0, len(g["ids"]), 7)])
    assert np.array_equal(plain, g["nearest_plain"])
    assert (g["nearest"] < 0).any() and (g["nearest_plain"] < 0).any()            # both outcomes are in the vectors
    # round 4: with a previous position (`ji_prv`, `np_box_r`): the reference's box-then-domain passes, G5d from the reference
    d = golden("g5d_nearest_local.npz")
    for (k, pj, pi, box_r, max_itr, use_res, rd), want in zip(d["cases"], d["ji"]):
        k = int(k)
        got = sit.NearestPoint((g["pSG"][k, 0], g["pSG"][k, 1]), g["latT"], g["lonT"], rd_found_km=float(rd),
                               resolkm=(g["resol"] if use_res else []), ji_prv=(int(pj), int(pi)), np_box_r=int(box_r), max_itr=int(max_itr))
        assert got == tuple(want), (k, got, tuple(want))
    # the whole batch through the context method: indices and distances
    ctx = sit.Context(0)
    z = np.zeros_like(g["latT"])
    ctx.set_grid(z, z, z, z, z, z, np.ones(z.shape, dtype=np.int8))
    ji, dmin = ctx.nearest_point(g["pSG"], g["latT"], g["lonT"], g["resol"], sit.rFoundKM, 10)
    assert np.array_equal(ji, g["nearest"])
    near = np.isfinite(dmin)                                                     # far seeds are rejected without a search
    assert np.all(ji[~near] == -1) and np.allclose(dmin[near], g["dmin"][near], rtol=1e-12, atol=1e-9)
    ctx.close()
    h = golden("g9_haversine.npz")
    for k in range(0, 4000, 400):
        d = sit.Haversine(h["plat"][k], h["plon"][k], h["xlat"][k:k + 50], h["xlon"][k:k + 50])
        assert d.shape == (50,) and abs(d[0] - h["dist"][k]) <= 1e-9 + 1e-12 * h["dist"][k]
    d2 = sit.Haversine(75., 20., g["latT"], g["lonT"])
    assert d2.shape == g["latT"].shape and d2.min() >= 0


def test_projection_2d_helpers(sit, golden):
    """util.ConvertGeo2CartesianNPSkm / ConvertCartesianNPSkm2Geo: the reference's committed seeding file pins the
    forward map at float32 (G7); the inverse is held to the forward by round trip."""
    g = golden("g7_projection.npz")
    lon = g["dat_lonlat"][:, 0].reshape(2, 5); lat = g["dat_lonlat"][:, 1].reshape(2, 5)      # tools/sidfexloc.dat values
    Y, X = sit.ConvertGeo2CartesianNPSkm(lat, lon)
    assert Y.shape == (2, 5) and X.shape == (2, 5)
    assert np.array_equal(Y.astype(np.float32).ravel(), g["y_pos"]) and np.array_equal(X.astype(np.float32).ravel(), g["x_pos"])
    la2, lo2 = sit.ConvertCartesianNPSkm2Geo(Y, X)
    dlon = (lo2 - lon + 180.) % 360. - 180.
    assert np.allclose(la2, lat, rtol=0, atol=1e-9) and np.allclose(dlon, 0., rtol=0, atol=1e-9)


def test_the_loop_body_written_against_sit_reproduces_the_reference_trajectories(sit, golden):
    """A per-buoy loop in the order of si3_part_tracker.py:378-490 whose `sit.*` calls land in THIS package, on the
    inputs of golden set G6 (trajectories the reference's own functions produced in that loop): same positions, masks,
    cells and deaths, bit for bit.  32 buoys x 24 records; every predicate call is one round trip to the GPU."""
    from sitrack_amd import synthetic as syn
    g = golden("g6_traj_curvi.npz")
    grid = syn.make_grid(int(g["Nj"]), int(g["Ni"]), dkm=float(g["dkm"]), warp=float(g["warp"]))
    xYf, xXf, xYu, xXu, xYv, xXv = (grid[k] for k in ("Yf", "Xf", "Yu", "Xu", "Yv", "Xv"))
    K, kstrt, rdt, Nt = g["u"].shape[0], int(g["kstrt"]), float(g["rdt"]), 24
    sel = np.arange(0, 128, 4)
    nP = len(sel)
    z1st, zLst = g["rec_first"][sel], g["rec_last"][sel]
    vJIt = g["jiT0"][sel].copy(); VRTCS = g["vert0"][sel].copy()
    iAlive = np.ones(nP, dtype='i1')
    xPosC = np.zeros((Nt + 1, nP, 2)) + sit.FillValue
    xmask = np.zeros((Nt + 1, nP), dtype='i1')
    for jP in range(nP):
        xPosC[z1st[jP] - kstrt, jP] = g["yx0"][sel[jP]]; xmask[z1st[jP] - kstrt, jP] = 1
    vMesh = np.zeros((nP, 4, 2)); lStillIn = np.zeros(nP, dtype=bool)
    ncross = 0
    for jt in range(Nt):
        jrec = jt + kstrt
        xUu = g["u"][jrec % K].astype(np.float64); xVv = g["v"][jrec % K].astype(np.float64); xIC = g["sic"][jrec % K].astype(np.float64)
        for jP in range(nP):
            if iAlive[jP] == 1 and z1st[jP] <= jrec <= zLst[jP]:
                ry, rx = xPosC[jt, jP]
                if not lStillIn[jP]:
                    vj, vi = VRTCS[jP]
                    vMesh[jP] = np.stack([xYf[vj, vi], xXf[vj, vi]], axis=1)
                jT, iT = vJIt[jP]
                zF = [xYf[jT, iT], xXf[jT, iT]]
                llum1 = sit.intersect2Seg([ry, rx], zF, [xYv[jT - 1, iT], xXv[jT - 1, iT]], [xYv[jT, iT], xXv[jT, iT]])
                llvm1 = sit.intersect2Seg([ry, rx], zF, [xYu[jT, iT - 1], xXu[jT, iT - 1]], [xYu[jT, iT], xXu[jT, iT]])
                zU = xUu[jT, iT - 1] if llum1 else xUu[jT, iT]
                zV = xVv[jT - 1, iT] if llvm1 else xVv[jT, iT]
                dx = zU * rdt; dy = zV * rdt
                rx_nxt = rx + dx / 1000.; ry_nxt = ry + dy / 1000.
                xPosC[jt + 1, jP] = [ry_nxt, rx_nxt]; xmask[jt + 1, jP] = 1
                lStillIn[jP] = sit.IsInsideQuadrangle(ry_nxt, rx_nxt, vMesh[jP])
                if not lStillIn[jP]:
                    icross = sit.CrossedEdge([ry, rx], [ry_nxt, rx_nxt], VRTCS[jP], xYf, xXf)
                    inhc = sit.NewHostCell(icross, [ry, rx], [ry_nxt, rx_nxt], VRTCS[jP], xYf, xXf)
                    VRTCS[jP], vJIt[jP] = sit.UpdtInd4NewCell(inhc, VRTCS[jP], vJIt[jP])
                    ncross += 1
                    if sit.Survive(jP, vJIt[jP], g["tmask"], pIceC=xIC) > 0:
                        iAlive[jP] = 0
        assert np.array_equal(vJIt, g["jiT_s1"][jt + 1][sel]) and np.array_equal(iAlive, g["alive_s1"][jt + 1][sel]), jt
    assert np.array_equal(xPosC, g["pos_s1"][:Nt + 1, sel]) and np.array_equal(xmask, g["msk_s1"][:Nt + 1, sel])
    assert ncross > 40 and (iAlive == 0).any()
