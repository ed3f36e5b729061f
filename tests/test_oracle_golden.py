"""Pins the CPU oracle (oracle/sitrk_oracle.c) against golden vectors produced by
the reference's own Python functions (tests/golden/gen_golden.py) and against the
reference's own known answers.  CPU only."""
import numpy as np
import pytest

from oracle import oracle as orc
from sitrack_amd import synthetic as syn


def test_g1_is_inside_quadrangle(golden):
    g = golden("g1_inside.npz")
    got = np.array([orc.IsInsideQuadrangle(p[0], p[1], q) for p, q in zip(g["pts"], g["quads"])])
    # reference's own manual test, tools/tests/test_pnt_inside_quad.py:16-24
    assert list(got[:4]) == [True, False, False, True]
    assert np.array_equal(got, g["inside"])


def test_g2_intersect2seg_and_ccw(golden):
    g = golden("g2_intersect.npz")
    P = g["P"]
    got = np.array([orc.intersect2Seg(p[0], p[1], p[2], p[3]) for p in P])
    ccw = np.array([orc.ccw(p[0], p[1], p[2]) for p in P])
    assert np.array_equal(got, g["intersect"])
    assert np.array_equal(ccw, g["ccw"])


def test_g3_crossing_chain(golden):
    g = golden("g3_crossing.npz")
    Yf, Xf = g["Yf"], g["Xf"]
    for k in range(len(g["jiT"])):
        ic = orc.CrossedEdge(g["P1"][k], g["P2"][k], g["vert"][k], Yf, Xf)
        nh = orc.NewHostCell(ic, g["P1"][k], g["P2"][k], g["vert"][k], Yf, Xf)
        v2, t2 = orc.UpdtInd4NewCell(nh, g["vert"][k], g["jiT"][k])
        assert ic == g["icross"][k] and nh == g["inhc"][k], k
        assert np.array_equal(v2, g["vert_out"][k]) and np.array_equal(t2, g["jiT_out"][k])
        # VRTCS stays a pure function of vJIt
        assert np.array_equal(orc.vertices_of(t2[None, :])[0], v2)
    assert set(np.unique(g["inhc"])) == set(range(1, 9))


def test_updt_unknown_direction_is_an_error():
    with pytest.raises(IndexError):
        orc.UpdtInd4NewCell(9, np.zeros((2, 4), dtype=np.int64), np.zeros(2, dtype=np.int64))


def test_g4_survive(golden):
    g = golden("g4_survive.npz")
    ones = np.ones_like(g["tmask"])
    a = np.array([orc.Survive(t, g["tmask"], g["sic"]) for t in g["jiT"]])
    b = np.array([orc.Survive(t, g["tmask"], g["sic32"].astype(np.float64)) for t in g["jiT"]])
    c = np.array([orc.Survive(t, ones, g["sic32"].astype(np.float64)) for t in g["jiT"]])
    assert np.array_equal(a, g["kill_a"])
    assert np.array_equal(b, g["kill_b"])
    assert np.array_equal(c, g["kill_c"])


def test_survive_uses_asymmetric_stencil():
    # tracking.py:79: the 5th point is [j-1,i-1], not [j-1,i]
    tm = np.ones((9, 9), dtype=np.int8)
    sic = np.ones((9, 9))
    tm[3, 4] = 0          # [j-1,i] of (4,4): must NOT kill
    assert orc.Survive((4, 4), tm, sic) == 0
    tm[3, 4] = 1
    tm[3, 3] = 0          # [j-1,i-1]: kills
    assert orc.Survive((4, 4), tm, sic, return_which=True) == (1, 2)


def test_g9_haversine(golden):
    g = golden("g9_haversine.npz")
    d = np.array([orc.Haversine(g["plat"][k], g["plon"][k], g["xlat"][k:k + 1], g["xlon"][k:k + 1])[0]
                  for k in range(len(g["plat"]))])
    # numpy's vector sin/cos/arcsin and libm may differ in the last bits
    assert np.allclose(d, g["dist"], rtol=1e-12, atol=1e-9)
    assert np.all(d[:50] == 0.0)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_g4b_survive_every_cell_of_wide_meshes(golden, tag):
    g = golden("g4b_survive_wide.npz")
    tm = g[tag + "_tmask"]
    Nj, Ni = tm.shape
    for sic, key in ((g[tag + "_sic"], "_kill"), (g[tag + "_sic32"].astype(np.float64), "_kill32")):
        got = np.array([[orc.Survive((j, i), tm, sic) for i in range(Ni)] for j in range(Nj)], dtype=np.int8)
        assert np.array_equal(got, g[tag + key])


def test_g5_nearest_point_and_cells(golden):
    g = golden("g5_seedinit.npz")
    nP = len(g["ids"])
    for k in range(nP):
        jy, jx, dmin = orc.NearestPoint(g["pSG"][k], g["latT"], g["lonT"], rd_found_km=orc.rFoundKM,
                                        resolkm=g["resol"], max_itr=10, return_dist=True)
        assert (jy, jx) == tuple(g["nearest"][k]), k
        assert abs(dmin - g["dmin"][k]) <= 1e-9 + 1e-12 * g["dmin"][k]
    plain = np.array([orc.NearestPoint(g["pSG"][k], g["latT"], g["lonT"], rd_found_km=8., max_itr=5)
                      for k in range(0, nP, 7)])
    assert np.array_equal(plain, g["nearest_plain"])
    Nj, Ni = g["latT"].shape
    for k in range(nP):
        jy, jx = g["nearest"][k]
        if 2 <= jy < Nj - 2 and 2 <= jx < Ni - 2:
            ok, ji, vv = orc.FindContainingCell(g["pSC"][k], (jy, jx), g["Yf"], g["Xf"])
            assert ok == bool(g["fcc_ok"][k]), k
            assert np.array_equal(ji, g["fcc_ji"][k]) and np.array_equal(vv, g["fcc_vert"][k]), k


def test_nearest_point_acceptance_threshold(golden):
    # with max_itr=10 the loop reduces to  d_min < 0.5*1.2**7*resol
    g = golden("g5_seedinit.npz")
    fac = 0.5 * 1.2 ** 7
    for k in range(len(g["ids"])):
        jy, jx, dmin = orc.NearestPoint(g["pSG"][k], g["latT"], g["lonT"], rd_found_km=orc.rFoundKM,
                                        resolkm=g["resol"], max_itr=10, return_dist=True)
        if abs(dmin / g["resol"][0, 0] - fac) > 1e-6:
            assert (jy >= 0) == (dmin < fac * g["resol"][0, 0])


def test_g5_seedinit(golden):
    g = golden("g5_seedinit.npz")
    nPn, oSG, oSC, oIDs, ojiT, overt, okeep = orc.SeedInit(g["ids"], g["pSG"], g["pSC"], g["latT"], g["lonT"],
                                                           g["Yf"], g["Xf"], g["resol"], g["tmask"], g["sic"])
    assert nPn == int(g["nPn"])
    assert np.array_equal(okeep, g["okeep"])
    assert np.array_equal(oIDs, g["oIDs"])
    assert np.array_equal(ojiT, g["ojiT"])
    assert np.array_equal(overt, g["overt"])
    assert np.array_equal(oSG, g["oSG"]) and np.array_equal(oSC, g["oSC"])


@pytest.mark.parametrize("prec", ["f4", "f8"])
@pytest.mark.parametrize("mesh", ["w", "r", "l"])
def test_g5b_seeds_exactly_on_t_and_f_points(golden, mesh, prec):
    """G5b: the reference's SeedInit / NearestPoint on seeds that sit exactly on T- and F-points as `nemoSeed(..., platF,
    plonF)` emits them -- near ties (warped and regular polar meshes) and EXACT two-way ties (a mesh regular in lat/lon:
    the first minimum in C order wins, locate.py:13-20), at the seeding file's float32 and at full precision."""
    g = golden("g5b_seeds_on_points.npz")
    G = lambda k: g[mesh + "_" + k]                                       # noqa: E731
    P = lambda k: g[mesh + "_" + prec + "_" + k]                          # noqa: E731
    nP = len(G("ids"))
    for k in range(0, nP, 3):
        jy, jx = orc.NearestPoint(P("pSG")[k], G("latT"), G("lonT"), rd_found_km=orc.rFoundKM, resolkm=G("resol"), max_itr=10)
        assert (jy, jx) == tuple(P("nearest")[k]), k
    nPn, oSG, oSC, oIDs, ojiT, overt, okeep = orc.SeedInit(G("ids"), P("pSG"), P("pSC"), G("latT"), G("lonT"), G("Yf"), G("Xf"),
                                                           G("resol"), G("tmask"), G("sic"))
    assert nPn == int(P("nPn")) and np.array_equal(okeep, P("okeep")) and np.array_equal(oIDs, P("oIDs"))
    assert np.array_equal(ojiT, P("ojiT")) and np.array_equal(overt, P("overt"))
    assert np.array_equal(oSG, P("oSG")) and np.array_equal(oSC, P("oSC"))
    if mesh == "l":
        assert (P("gap") == 0).sum() > 500                                # the exact ties are really in the set


def test_g5c_seedinit_on_a_300x330_mesh_around_the_pole(golden):
    """G5c: the reference's SeedInit on a mesh of 99 000 T-points with the pole inside it (2 600 seeds).  The oracle's scalar
    whole-grid scan must give the reference's nearest points (every 9th seed here: 1e5 Haversines each), kept set and cells."""
    from conftest import g5c_mesh
    g = golden("g5c_seedinit_300x330.npz")
    m, tmask, sic = g5c_mesh(g)
    for k in range(0, len(g["ids"]), 9):
        jy, jx = orc.NearestPoint(g["pSG"][k], m["latT"], m["lonT"], rd_found_km=orc.rFoundKM, resolkm=m["resol"], max_itr=10)
        assert (jy, jx) == tuple(g["nearest"][k]), k
    nPn, oSG, oSC, oIDs, ojiT, overt, okeep = orc.SeedInit(g["ids"], g["pSG"], g["pSC"], m["latT"], m["lonT"], m["Yf"], m["Xf"],
                                                           m["resol"], tmask, sic, nthreads=8)
    assert nPn == int(g["nPn"]) and np.array_equal(okeep, g["okeep"]) and np.array_equal(oIDs, g["oIDs"]) and np.array_equal(ojiT, g["ojiT"])


@pytest.mark.parametrize("tag", ["curvi", "regular"])
@pytest.mark.parametrize("strat", [1, 0])
def test_g6_trajectories_bit_exact(golden, tag, strat):
    g = golden("g6_traj_%s.npz" % tag)
    grid = syn.make_grid(int(g["Nj"]), int(g["Ni"]), dkm=float(g["dkm"]), warp=float(g["warp"]))
    grid["tmask"] = g["tmask"]
    tr = orc.Tracker(grid, g["yx0"], g["jiT0"], vert0=g["vert0"], rec_first=g["rec_first"], rec_last=g["rec_last"],
                     rdt=float(g["rdt"]), uv_strategy=strat)
    K, kstrt, Nt = g["u"].shape[0], int(g["kstrt"]), int(g["Nt"])
    pos, msk, jit, alive = g["pos_s%d" % strat], g["msk_s%d" % strat], g["jiT_s%d" % strat], g["alive_s%d" % strat]
    for jt in range(Nt):
        jrec = jt + kstrt
        pn, mn = tr.step(jrec, g["u"][jrec % K], g["v"][jrec % K], g["sic"][jrec % K])
        # a buoy whose window opens at jt+1 has its seed position pre-written in the reference array
        opening = (g["rec_first"] - kstrt) == (jt + 1)
        assert np.array_equal(pn[~opening], pos[jt + 1][~opening]), (jt, "positions must be bit-exact")
        assert np.array_equal(mn[~opening], msk[jt + 1][~opening])
        assert np.array_equal(tr.jiT, jit[jt + 1])
        assert np.array_equal(tr.alive, alive[jt + 1])
    assert np.array_equal(tr.vert, g["vert_s%d" % strat])
    assert tr.ncross == int(g["codes_s%d" % strat].sum())


@pytest.mark.parametrize("strat", [1, 0])
def test_g6b_fast_flow_trajectories_bit_exact(golden, strat):
    """G6b: 1 500 buoys x 90 records of a flow of up to two cells per record through the restated reference loop -- 36 000 of the
    77 000 crossings return code 4, most of them `CrossedEdge`'s fall-through, after which the reference keeps a buoy in a cell
    that does not contain it.  The oracle must walk the same path: per-record digests (positions bit for bit, masks, cells, alive)."""
    from conftest import g6b_case, traj_digest_row
    g = golden("g6b_traj_fast.npz")
    grid, u, v, sic = g6b_case(g)
    K, kstrt, Nt = u.shape[0], int(g["kstrt"]), int(g["Nt"])
    tr = orc.Tracker(grid, g["yx0"], g["jiT0"].astype(np.int64), rdt=float(g["rdt"]), uv_strategy=strat, nthreads=4)
    dg = g["digest_s%d" % strat]
    last = g["yx0"].copy()
    for jt in range(Nt):
        jrec = jt + kstrt
        pn, mn = tr.step(jrec, u[jrec % K].astype(np.float64), v[jrec % K].astype(np.float64), sic[jrec % K].astype(np.float64))
        assert np.array_equal(traj_digest_row(pn, mn, tr.jiT, tr.alive), dg[jt + 1]), jt
        last[mn == 1] = pn[mn == 1]
    assert np.array_equal(last, g["last_pos_s%d" % strat]) and np.array_equal(tr.jiT, g["jiT_end_s%d" % strat])
    assert np.array_equal(tr.alive, g["alive_end_s%d" % strat]) and tr.ncross == int(g["codes_s%d" % strat].sum())


@pytest.mark.parametrize("name", ["g6c_c2cut.npz", "g6d_c3cut.npz", "g6f_c5shape_cut.npz", "g6e_c3warp_cut.npz"])
def test_g6cd_reference_trajectories_on_the_baseline_workloads(golden, name):
    """G6c / G6d: the REFERENCE's trajectories on inputs cut from BASELINE configs 2 and 3 themselves -- the first 10^3 buoys x 100
    records of bench.py's C2 / C3 workloads (same grid, same seeds, the 32 resident records cycled) -- both velocity rules, with and
    without per-buoy record windows.  The oracle walks them bit for bit: per-record digests, final positions, cells, alive.
    (The four variants advance in lockstep so that each record is promoted to fp64 once, into buffers that are reused.)
    G6e / G6f (round 4): the same on bench.py's CURVILINEAR workloads -- C3 sheared and stretched (`--warp 1.0`; host cells from the
    reference's FindContainingCell) and the NANUK4-shaped mesh of `--config c5shape` (seeds kept by the reference's SeedInit, an island,
    a polynya in which 2 % of the cut's buoys die, flow up to 0.9 m/s)."""
    from conftest import baseline_cut_case, traj_digest_row
    g = golden(name)
    grid, u, v, sic, yx0 = baseline_cut_case(g)
    K, kstrt, Nt = u.shape[0], int(g["kstrt"]), int(g["Nt"])
    runs = {}
    for strat in (1, 0):
        for tag in ("", "w"):
            kw = dict(rec_first=g["rec_first"], rec_last=g["rec_last"]) if tag else {}
            runs["s%d%s" % (strat, tag)] = (orc.Tracker(grid, yx0, g["jiT0"].astype(np.int64), rdt=float(g["rdt"]), uv_strategy=strat,
                                                        nthreads=4, **kw), yx0.copy())
    bu, bv = np.empty(u.shape[1:]), np.empty(u.shape[1:])
    bs = np.ones(u.shape[1:])                                       # (siconc is 1 everywhere in the BASELINE fields: checked by the rebuild)
    ice_varies = "kind" in g and str(g["kind"]) == "c5shape"        # ... but for the polynya of the C5-shape workload
    for jt in range(Nt):
        jrec = jt + kstrt
        np.copyto(bu, u[jrec % K]); np.copyto(bv, v[jrec % K])
        if ice_varies:
            np.copyto(bs, sic[jrec % K])
        for key, (tr, last) in runs.items():
            pn, mn = tr.step(jrec, bu, bv, bs)
            if key.endswith("w"):                                   # the driver pre-writes a late starter's seed position (:289-312)
                opening = (g["rec_first"] - kstrt) == (jt + 1)
                pn[opening] = yx0[opening]; mn[opening] = 1
            assert np.array_equal(traj_digest_row(pn, mn, tr.jiT, tr.alive), g["digest_" + key][jt + 1]), (key, jt)
            last[mn == 1] = pn[mn == 1]
    for key, (tr, last) in runs.items():
        assert np.array_equal(last, g["last_pos_" + key]) and np.array_equal(tr.jiT, g["jiT_end_" + key]), key
        assert np.array_equal(tr.alive, g["alive_end_" + key]) and tr.ncross == int(g["codes_" + key].sum()), key
        assert tr.ncross > 0.05 * Nt * len(yx0)
        assert (tr.alive == 0).sum() == (20 if ice_varies else 0)


def test_g5d_nearest_point_with_a_previous_position(golden):
    """G5d: the reference's `NearestPoint` with `ji_prv` / `np_box_r` (locate.py:241-271; off the tracker's path -- SeedInit passes no
    previous position -- but part of the function): the host logic of sitrack_amd/predicates.py over the oracle's Haversine reproduces
    the reference's 210 answers, found and given up alike, incl. the box-relative index into `resolkm` of the first pass."""
    from sitrack_amd.predicates import nearest_point_with_previous
    g, d = golden("g5_seedinit.npz"), golden("g5d_nearest_local.npz")
    assert (d["ji"][:, 0] < 0).sum() > 20 and (d["ji"][:, 0] >= 0).sum() > 100
    for (k, pj, pi, box_r, max_itr, use_res, rd), want in zip(d["cases"], d["ji"]):
        k = int(k)
        got = nearest_point_with_previous(orc.Haversine, (g["pSG"][k, 0], g["pSG"][k, 1]), g["latT"], g["lonT"], float(rd),
                                          g["resol"] if use_res else None, (int(pj), int(pi)), int(box_r), int(max_itr))
        assert got == tuple(want), (k, got, tuple(want))


def test_g7_forward_projection_matches_reference_fixture(golden):
    # tools/nc/sitrack_seeding_sidfex_19961215_00_HSS5.nc__KEEP: (lat,lon) f4 -> (y_pos,x_pos) f4 made by the
    # reference's cartopy forward projection from tools/sidfexloc.dat
    g = golden("g7_projection.npz")
    assert np.array_equal(g["id_buoy"], g["dat_id"])
    ll = np.stack([g["dat_lonlat"][:, 1], g["dat_lonlat"][:, 0]], axis=1)
    assert np.array_equal(ll[:, 0].astype(np.float32), g["latitude"])
    assert np.array_equal(ll[:, 1].astype(np.float32), g["longitude"])
    yx = orc.Geo2CartNPSkm1D(ll)
    assert np.array_equal(yx[:, 0].astype(np.float32), g["y_pos"])
    assert np.array_equal(yx[:, 1].astype(np.float32), g["x_pos"])


def test_g7_inverse_projection_round_trip(golden):
    g = golden("g7_projection.npz")
    ll = np.stack([g["dat_lonlat"][:, 1], g["dat_lonlat"][:, 0]], axis=1)
    back = orc.CartNPSkm2Geo1D(orc.Geo2CartNPSkm1D(ll))
    assert np.allclose(back, ll, rtol=0, atol=1e-8)
    rng = np.random.default_rng(7)
    ll = np.stack([rng.uniform(40, 89.999, 5000), rng.uniform(-180, 180, 5000)], axis=1)
    back = orc.CartNPSkm2Geo1D(orc.Geo2CartNPSkm1D(ll))
    assert np.allclose(back, ll, rtol=0, atol=1e-8)
    # dead buoys hold -9999 km (si3_part_tracker.py:327,493): still a finite lat/lon
    dead = orc.CartNPSkm2Geo1D(np.array([[-9999., -9999.]]))
    assert np.all(np.isfinite(dead)) and -180 <= dead[0, 1] <= 180


def test_g7_inverse_projection_pinned_by_the_reference_fixture(golden):
    """a12, direct pin: the reference's own seeding file holds (latitude, longitude) AND the (y_pos, x_pos) its cartopy
    projection made of them (tools/nc/sitrack_seeding_sidfex_19961215_00_HSS5.nc__KEEP, all f4).  Feeding the stored
    y_pos/x_pos to the inverse (sitrack/util.py:413-429) must give back the stored latitude/longitude at north_star's
    tolerance, 1e-5 relative -- reference-held data on both sides, no round trip through our own forward."""
    g = golden("g7_projection.npz")
    yx = np.stack([g["y_pos"].astype(np.float64), g["x_pos"].astype(np.float64)], axis=1)
    ll = orc.CartNPSkm2Geo1D(yx)
    assert ll.shape == (10, 2)
    assert np.allclose(ll[:, 0], g["latitude"].astype(np.float64), rtol=1e-5, atol=0)
    assert np.allclose(np.mod(ll[:, 1], 360.), np.mod(g["longitude"].astype(np.float64), 360.), rtol=1e-5, atol=0)
    # what the fixture allows to state: the error is that of the f4 storage, far below the bar
    assert np.abs(ll[:, 0] - g["latitude"]).max() < 2e-5 and np.abs(ll[:, 1] - g["longitude"]).max() < 2e-4


def test_g8_get_time_span(golden):
    g = golden("g8_timespan.npz")
    vt = g["vtime"]
    for c in g["cases"]:
        sd, stop = int(c[0]), (None if c[1] < 0 else int(c[1]))
        assert orc.GetTimeSpan(3600., vt, sd, vt[0], vt[-1], iStop=stop) == tuple(int(x) for x in c[2:])
