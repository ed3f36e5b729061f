"""Parity tests proper: the HIP path, called through the C ABI, against the CPU
oracle and the committed golden vectors.  Bit-exact for indices, masks AND fp64
positions (the kernels evaluate the reference's arithmetic in the same order with
no FMA contraction).  Run on the GPU box with `-m gpu`."""
import numpy as np
import pytest

import sitrack_amd as sit
from sitrack_amd import synthetic as syn
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = sit.Context(0)
    yield c
    c.close()


def make_tracker(grid, tmask, nslots, **kw):
    return sit.IceTracker(grid["Yf"], grid["Xf"], grid["Yu"], grid["Xu"], grid["Yv"], grid["Xv"], tmask,
                          nslots=nslots, **kw)


@pytest.mark.parametrize("tag", ["curvi", "regular"])
@pytest.mark.parametrize("strat", [1, 0])
@pytest.mark.parametrize("sort", [True, False])
def test_g6_golden_trajectories(golden, tag, strat, sort):
    g = golden("g6_traj_%s.npz" % tag)
    grid = syn.make_grid(int(g["Nj"]), int(g["Ni"]), dkm=float(g["dkm"]), warp=float(g["warp"]))
    K, kstrt, Nt = g["u"].shape[0], int(g["kstrt"]), int(g["Nt"])
    trk = make_tracker(grid, g["tmask"], K, rdt=float(g["rdt"]), iUVstrategy=strat)
    try:
        trk.set_buoys(g["yx0"], g["jiT0"], g["rec_first"], g["rec_last"], sort=sort)
        if sort:
            trk.ctx.set_resort(5)
        for k in range(K):
            trk.load_record(k, g["u"][k], g["v"][k], g["sic"][k])
        pos, msk, jit, alive = g["pos_s%d" % strat], g["msk_s%d" % strat], g["jiT_s%d" % strat], g["alive_s%d" % strat]
        for jt in range(Nt):
            jrec = jt + kstrt
            trk.step(jrec, jrec % K)
            pn, mn = trk.record(jrec)
            opening = (g["rec_first"] - kstrt) == (jt + 1)     # seed position pre-written by the driver
            assert np.array_equal(pn[~opening], pos[jt + 1][~opening]), jt
            assert np.array_equal(mn[~opening], msk[jt + 1][~opening]), jt
            st = trk.state()
            assert np.array_equal(st["vJIt"], jit[jt + 1]), jt
            assert np.array_equal(st["iAlive"], alive[jt + 1]), jt
        assert np.array_equal(trk.state()["VRTCS"], g["vert_s%d" % strat])
        assert trk.alive_count() == int(alive[-1].sum())
    finally:
        trk.close()


@pytest.mark.parametrize("warp,field_dtype", [(1.0, np.float32), (0.0, np.float32), (1.0, np.float64)])
def test_random_cloud_vs_oracle(warp, field_dtype):
    Nj, Ni, nP, K, Nt = 192, 224, 60000, 5, 40
    grid = syn.make_grid(Nj, Ni, dkm=4.0, warp=warp)
    u, v, sic = syn.make_fields(grid, K=K, seed=11, umax=0.8, drift=0.25, ripple=0.15, dtype=field_dtype)
    tmask = grid["tmask"].copy()
    tmask[90:100, 100:120] = 0
    sic[:, 40:60, 30:70] = 0.03
    _, yx = syn.make_buoys(grid, nP, seed=5, frac=0.8)
    trk = make_tracker(grid, tmask, K, field_dtype=field_dtype)
    try:
        guess = syn.nearest_t_plane(grid, yx).astype(np.int32)
        found, ji, _ = sit.FindContainingCell(yx, guess, ctx=trk.ctx)
        # the oracle's FindContainingCell agrees buoy by buoy
        for b in range(0, nP, 997):
            ok, ji_o, _ = orc.FindContainingCell(yx[b], guess[b], grid["Yf"], grid["Xf"])
            assert ok == found[b] and (not ok or np.array_equal(ji_o, ji[b]))
        yx, ji = yx[found], ji[found]
        assert len(yx) > 0.9 * nP
        trk.set_buoys(yx, ji)
        trk.ctx.set_resort(7)
        g2 = dict(grid); g2["tmask"] = tmask
        ref = orc.Tracker(g2, yx, ji, nthreads=8)
        for k in range(K):
            trk.load_record(k, u[k], v[k], sic[k])
        for jrec in range(Nt):
            trk.step(jrec, jrec % K)
            rp, rm = ref.step(jrec, u[jrec % K], v[jrec % K], sic[jrec % K])
            if jrec % 9 == 0 or jrec == Nt - 1:
                pn, mn = trk.record(jrec)
                assert np.array_equal(mn, rm), jrec
                assert np.array_equal(pn, rp), jrec
        st = trk.state()
        assert np.array_equal(st["yx"], ref.pos)
        assert np.array_equal(st["vJIt"], ref.jiT)
        assert np.array_equal(st["iAlive"], ref.alive)
        assert ref.ncross > nP            # the crossing path was exercised
        assert 0 < trk.alive_count() < len(yx)
        kr = st["kill_rec"]
        assert np.array_equal(kr >= 0, ref.alive == 0)
    finally:
        trk.close()


def test_run_many_steps_equals_stepping(ctx):
    grid = syn.make_grid(96, 96, dkm=4.0, warp=1.0)
    u, v, sic = syn.make_fields(grid, K=3, seed=3, umax=0.6, drift=0.2)
    _, yx = syn.make_buoys(grid, 5000, seed=2, frac=0.6)
    res = []
    for mode in ("run", "step"):
        trk = make_tracker(grid, grid["tmask"], 3)
        found, ji, _ = sit.FindContainingCell(yx, syn.nearest_t_plane(grid, yx), ctx=trk.ctx)
        assert found.mean() > 0.9
        trk.set_buoys(yx[found], ji[found])
        for k in range(3):
            trk.load_record(k, u[k], v[k], sic[k])
        if mode == "run":
            trk.ctx.run(1, 10, 21)
        else:
            for s in range(21):
                trk.step(10 + s, (1 + s) % 3)
        res.append(trk.state())
        trk.close()
    for key in ("yx", "vJIt", "iAlive", "kill_rec"):
        assert np.array_equal(res[0][key], res[1][key])


def test_find_cells_matches_golden(golden, ctx):
    g = golden("g5_seedinit.npz")
    Nj, Ni = g["latT"].shape
    ctx.set_grid(g["Yf"], g["Xf"], g["Yf"], g["Xf"], g["Yf"], g["Xf"], g["tmask"])
    near = g["nearest"]
    sel = (near[:, 0] >= 2) & (near[:, 0] < Nj - 2) & (near[:, 1] >= 2) & (near[:, 1] < Ni - 2)
    found, ji = ctx.find_cells(g["pSC"][sel], near[sel].astype(np.int32))
    assert np.array_equal(found, g["fcc_ok"][sel])
    assert np.array_equal(ji[found], g["fcc_ji"][sel][found])


def test_seed_init_matches_golden(golden):
    g = golden("g5_seedinit.npz")
    out = sit.SeedInit(g["ids"], g["pSG"], g["pSC"], g["latT"], g["lonT"], g["Yf"], g["Xf"], g["resol"], g["tmask"],
                       xIceConc=g["sic"])
    nPn, oSG, oSC, oIDs, ojiT, overt, okeep = out
    assert nPn == int(g["nPn"])
    assert np.array_equal(okeep, g["okeep"])
    assert np.array_equal(oIDs, g["oIDs"]) and oIDs.dtype == np.int64
    assert np.array_equal(ojiT, g["ojiT"])
    assert np.array_equal(overt, g["overt"])
    assert np.array_equal(oSG, g["oSG"]) and np.array_equal(oSC, g["oSC"])


def test_projection(golden, ctx):
    g = golden("g7_projection.npz")
    ll = np.stack([g["dat_lonlat"][:, 1], g["dat_lonlat"][:, 0]], axis=1)
    yx = sit.Geo2CartNPSkm1D(ll, ctx=ctx)
    # the reference's own fixture (cartopy forward projection, stored as f4)
    assert np.array_equal(yx[:, 0].astype(np.float32), g["y_pos"])
    assert np.array_equal(yx[:, 1].astype(np.float32), g["x_pos"])
    rng = np.random.default_rng(3)
    pts = np.concatenate([rng.uniform(-4000, 4000, (20000, 2)), [[-9999., -9999.], [0., 0.]]])
    got = sit.CartNPSkm2Geo1D(pts, ctx=ctx)
    want = orc.CartNPSkm2Geo1D(pts)
    # north_star tolerance: 1e-5 relative on lat/lon; device libm differs from glibc in the last bits only
    assert np.allclose(got, want, rtol=1e-12, atol=1e-10)
    back = sit.Geo2CartNPSkm1D(got[:-1], ctx=ctx)
    assert np.allclose(back, pts[:-1], rtol=1e-11, atol=1e-8)


def test_record_latlon_of_dead_and_live(ctx):
    grid = syn.make_grid(32, 32, dkm=4.0)
    u, v, sic = syn.make_fields(grid, K=1, umax=0.3)
    trk = make_tracker(grid, grid["tmask"], 1)
    yx = np.array([[0.5, 0.5], [10.2, -7.1]])
    found, ji, _ = sit.FindContainingCell(yx, syn.nearest_t_guess(grid, yx), ctx=trk.ctx)
    assert found.all()
    trk.set_buoys(yx, ji, np.array([0, 5]), np.array([9, 9]))
    trk.load_record(0, u[0], v[0], sic[0])
    trk.step(0, 0)
    pos, msk, ll = trk.record(0, latlon=True)
    assert list(msk) == [1, 0] and np.all(pos[1] == -9999.)
    assert np.allclose(ll, orc.CartNPSkm2Geo1D(pos), rtol=1e-12, atol=1e-10)
    trk.close()


def test_errors(ctx):
    grid = syn.make_grid(16, 16)
    ctx.set_grid(grid["Yf"], grid["Xf"], grid["Yu"], grid["Xu"], grid["Yv"], grid["Xv"], grid["tmask"])
    with pytest.raises(IndexError):
        ctx.set_buoys(np.zeros((1, 2)), np.array([[0, 5]]))
    with pytest.raises(IndexError):
        ctx.set_buoys(np.zeros((1, 2)), np.array([[5, 15]]))
    ctx.set_buoys(np.zeros((0, 2)), np.zeros((0, 2), dtype=np.int32))
    with pytest.raises(sit.SitrkError):
        ctx.step(0, 0)                      # no records allocated
    ctx.alloc_records(2, np.float32)
    ctx.step(0, 0)                          # nP == 0: no-op
    with pytest.raises(sit.SitrkError):
        ctx.step(2, 0)
    with pytest.raises(sit.SitrkError):
        ctx.set_params(3600., 2, 0.1)
    with pytest.raises(ValueError):
        ctx.push_record(0, np.zeros((3, 3)), np.zeros((3, 3)), np.zeros((3, 3)))
