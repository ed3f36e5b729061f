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


def test_g1_is_inside_quadrangle_on_device(golden, ctx):
    g = golden("g1_inside.npz")
    got = ctx.eval_inside(g["pts"], g["quads"])
    assert list(got[:4]) == [True, False, False, True]          # reference tools/tests/test_pnt_inside_quad.py:16-24
    assert np.array_equal(got, g["inside"])


def test_g2_intersect2seg_and_ccw_on_device(golden, ctx):
    g = golden("g2_intersect.npz")
    inter, ccw = ctx.eval_intersect(g["P"])
    assert np.array_equal(inter, g["intersect"]) and np.array_equal(ccw, g["ccw"])


def test_g3_crossing_chain_on_device(golden, ctx):
    g = golden("g3_crossing.npz")
    Nj, Ni = g["Yf"].shape
    ctx.set_grid(g["Yf"], g["Xf"], g["Yf"], g["Xf"], g["Yf"], g["Xf"], np.ones((Nj, Ni), dtype=np.int8))
    got, icross, inhc = ctx.eval_crossing(g["P1"], g["P2"], g["jiT"])
    assert np.array_equal(icross, g["icross"])                   # CrossedEdge, incl. the fall-through to 4
    assert np.array_equal(inhc, g["inhc"])                       # NewHostCell, all 8 codes
    assert np.array_equal(got, g["jiT_out"])                     # UpdtInd4NewCell
    assert set(np.unique(g["inhc"])) == set(range(1, 9))
    with pytest.raises(IndexError):
        ctx.eval_crossing(g["P1"][:1], g["P2"][:1], np.array([[0, 3]]))


def test_g4_survive_on_device(golden, ctx):
    g = golden("g4_survive.npz")
    Nj, Ni = g["tmask"].shape
    z = np.zeros((Nj, Ni))
    for tm, sic, key in ((g["tmask"], g["sic"], "kill_a"), (g["tmask"], g["sic32"].astype(np.float64), "kill_b"),
                         (np.ones_like(g["tmask"]), g["sic32"].astype(np.float64), "kill_c")):
        ctx.set_grid(z, z, z, z, z, z, tm)
        ctx.set_params(3600., 1, 0.1)
        mask = ctx.survive_mask(sic)
        assert np.array_equal(mask.ravel(), g[key].astype(np.int8)), key     # jiT in G4 enumerates the grid in C order


@pytest.mark.parametrize("tile", [0, 1])
@pytest.mark.parametrize("tag", ["a", "b"])
def test_g4b_survive_reference_golden_both_kernels(golden, ctx, tag, tile):
    """G4b: the REFERENCE's `Survive` for every cell of a 44 x 252 and a 40 x 48 mesh (rows 16-byte aligned: the register-rolling
    kernel, with a strip boundary inside the wide one) through `sitrk_survive_mask`, i.e. the very kernel that derives a resident
    record's bytes -- and through the LDS-tile kernel (knob)."""
    g = golden("g4b_survive_wide.npz")
    tm = g[tag + "_tmask"]
    z = np.zeros(tm.shape)
    ctx.set_grid(z, z, z, z, z, z, tm)
    ctx.set_params(3600., 1, 0.1)
    ctx.set_tuning(survive_tile=tile)
    try:
        for sic, key in ((g[tag + "_sic"], "_kill"), (g[tag + "_sic32"].astype(np.float64), "_kill32")):
            assert np.array_equal(ctx.survive_mask(sic), g[tag + key]), key
    finally:
        ctx.set_tuning(survive_tile=0)


@pytest.mark.parametrize("shape", [(37, 44), (70, 252), (66, 500), (40, 46)])
def test_survive_kernels_agree_with_the_oracle_on_every_cell(ctx, shape):
    """`Survive` for every cell through both device forms -- the register-rolling kernel (meshes with Ni % 4 == 0: a wave walks
    down a strip of 248 columns, so 252 and 500 columns cross strip boundaries; more rows than one 32-row chunk) and the LDS-tile
    kernel (any Ni; forced by the knob `survive_tile`) -- against the oracle's scalar `Survive` (tracking.py:62-93): land
    stencils incl. the asymmetric [j-1,i-1] point and mask values other than 0/1, ice around the threshold, NaN and huge ice."""
    Nj, Ni = shape
    rng = np.random.default_rng(Nj * 1000 + Ni)
    tmask = (rng.random((Nj, Ni)) > 0.08).astype(np.int8)
    tmask[rng.integers(2, Nj - 2, 6), rng.integers(2, Ni - 2, 6)] = 2          # a 5-point sum can reach 5 with a land point in it
    tmask[rng.integers(2, Nj - 2, 6), rng.integers(2, Ni - 2, 6)] = -1
    sic = rng.choice([0.0, 0.05, 0.0999, 0.1, 0.1001, 0.12, 0.5, 1.0], size=(Nj, Ni)).astype(np.float64)
    sic[rng.integers(0, Nj, 8), rng.integers(0, Ni, 8)] = np.nan
    sic[rng.integers(0, Nj, 4), rng.integers(0, Ni, 4)] = 1e20
    z = np.zeros((Nj, Ni))
    ctx.set_grid(z, z, z, z, z, z, tmask)
    ctx.set_params(3600., 1, 0.1)
    want = np.array([[orc.Survive((j, i), tmask, sic) for i in range(Ni)] for j in range(Nj)], dtype=np.int8)
    assert 0.2 < want[2:-2, 2:-2].mean() < 0.9
    got = {}
    for tile in (0, 1):
        ctx.set_tuning(survive_tile=tile)
        got[tile] = ctx.survive_mask(sic)
    ctx.set_tuning(survive_tile=0)
    assert np.array_equal(got[1], want), "LDS-tile kernel"
    assert np.array_equal(got[0], want), "register-rolling kernel" if Ni % 4 == 0 else "LDS-tile kernel (Ni % 4 != 0)"


@pytest.mark.parametrize("fdt", [np.float32, np.float64])
def test_survive_bytes_of_resident_records_both_kernels(fdt):
    """The Survive bytes AND the packed 8-neighbour bytes of resident records, as the trajectories see them: a fast flow over a
    500-column mesh (two strip boundaries of the register-rolling kernel), holes in the ice and land -- one-record launches read
    the bytes, fused launches the packed neighbourhoods; both kernels, float32 and float64 records, against the oracle."""
    Nj, Ni, K, Nt = 70, 500, 4, 24
    grid = syn.make_grid(Nj, Ni, dkm=4.0, warp=1.0)
    u, v, sic = syn.make_fields(grid, K=K, seed=5, umax=1.0, drift=0.6, ripple=0.1, dtype=fdt)
    rng = np.random.default_rng(3)
    tmask = grid["tmask"].copy()
    for _ in range(40):
        j, i = int(rng.integers(3, Nj - 4)), int(rng.integers(3, Ni - 4))
        tmask[j, i] = 0
    for k in range(K):
        for _ in range(30):
            j, i = int(rng.integers(3, Nj - 6)), int(rng.integers(3, Ni - 8))
            sic[k, j:j + 2, i:i + 3] = 0.04
    _, yx = syn.make_buoys(grid, 40000, seed=8, frac=0.92)
    g2 = dict(grid); g2["tmask"] = tmask
    res = {}
    for tile in (0, 1):
        for mode in ("run", "step"):
            trk = sit.IceTracker(grid["Yf"], grid["Xf"], grid["Yu"], grid["Xu"], grid["Yv"], grid["Xv"], tmask, nslots=K, field_dtype=fdt)
            trk.ctx.set_tuning(survive_tile=tile)
            found, ji, _ = sit.FindContainingCell(yx, syn.nearest_t_plane(grid, yx), ctx=trk.ctx)
            p, c = yx[found], ji[found]
            trk.set_buoys(p, c)
            for k in range(K):
                trk.load_record(k, u[k], v[k], sic[k])
            if mode == "run":
                trk.ctx.run(0, 0, Nt)
            else:
                for s_ in range(Nt):
                    trk.step(s_, s_ % K)
            res[(tile, mode)] = trk.state()
            trk.close()
    ref = orc.Tracker(g2, p, c, nthreads=8)
    for s_ in range(Nt):
        ref.step(s_, u[s_ % K].astype(np.float64), v[s_ % K].astype(np.float64), sic[s_ % K].astype(np.float64), want_out=False)
    assert 0.02 < (ref.alive == 0).mean() < 0.9
    for key, st in res.items():
        assert np.array_equal(st["yx"], ref.pos) and np.array_equal(st["vJIt"], ref.jiT) and np.array_equal(st["iAlive"], ref.alive), key


@pytest.mark.parametrize("strat", [1, 0])
@pytest.mark.parametrize("mode", ["step", "run", "run_unsorted"])
def test_g6b_fast_flow_golden_trajectories(golden, strat, mode):
    """G6b on the device: the reference's fast-flow trajectories (up to two cells per record; tens of thousands of fall-through
    crossings) record by record with per-record digests, and through fused launches (sorted with periodic re-sorts: buoys leave
    the LDS patch all the time; unsorted: no patch) by the final state."""
    from conftest import g6b_case, traj_digest_row
    g = golden("g6b_traj_fast.npz")
    grid, u, v, sic = g6b_case(g)
    K, kstrt, Nt = u.shape[0], int(g["kstrt"]), int(g["Nt"])
    trk = make_tracker(grid, grid["tmask"], K, rdt=float(g["rdt"]), iUVstrategy=strat)
    try:
        trk.set_buoys(g["yx0"], g["jiT0"], sort=(mode != "run_unsorted"))
        trk.ctx.set_resort(0 if mode == "run_unsorted" else 7)
        for k in range(K):
            trk.load_record(k, u[k], v[k], sic[k])
        if mode == "step":
            dg = g["digest_s%d" % strat]
            for jt in range(Nt):
                jrec = jt + kstrt
                trk.step(jrec, jrec % K)
                pn, mn = trk.record(jrec)
                st = trk.state()
                assert np.array_equal(traj_digest_row(pn, mn, st["vJIt"], st["iAlive"]), dg[jt + 1]), jt
        else:
            trk.ctx.run(kstrt % K, kstrt, Nt)
        st = trk.state()
        alive = st["iAlive"] == 1
        assert np.array_equal(st["vJIt"], g["jiT_end_s%d" % strat]) and np.array_equal(st["iAlive"], g["alive_end_s%d" % strat])
        assert np.array_equal(st["yx"], g["last_pos_s%d" % strat]) and 0.3 < alive.mean() < 0.8
    finally:
        trk.close()


_CUT_TRK = {}


def _cut_tracker(g):
    """one tracker per BASELINE mesh with the workload's 32 records resident (6.4 GB at 4096^2), shared by the G6c / G6d cases"""
    from conftest import baseline_cut_case
    grid, u, v, sic, yx0 = baseline_cut_case(g)
    key = grid["Yf"].shape + (str(g["kind"]) if "kind" in g else "regular",)
    if key not in _CUT_TRK:
        for t in _CUT_TRK.values():
            t.close()
        _CUT_TRK.clear()
        trk = make_tracker(grid, grid["tmask"], u.shape[0], rdt=float(g["rdt"]))
        for k in range(u.shape[0]):
            trk.load_record(k, u[k], v[k], sic[k])
        _CUT_TRK[key] = trk
    return _CUT_TRK[key], yx0, u.shape[0]


@pytest.mark.parametrize("name", ["g6c_c2cut.npz", "g6d_c3cut.npz", "g6f_c5shape_cut.npz", "g6e_c3warp_cut.npz"])   # (outermost: one tracker per mesh)
@pytest.mark.parametrize("mode", ["step", "step_w", "run", "run_unsorted", "run_w", "run_w_unsorted"])
@pytest.mark.parametrize("strat", [1, 0])
def test_g6cd_reference_trajectories_on_the_baseline_workloads(golden, name, mode, strat):
    """G6c / G6d on the device: REFERENCE output on the BASELINE grids themselves -- the first 10^3 buoys x 100 records of bench.py's
    C2 (512^2) and C3 (4096^2) workloads, 32 resident records cycled, both velocity rules -- record by record with per-record digests
    (`step`), through fused launches sorted / unsorted (`run`), and -- the case rounds 1-3 tied to the reference only transitively --
    FUSED LAUNCHES WITH PER-BUOY RECORD WINDOWS (`run_w`: 12 late starters, 12 early stoppers; the launches that cut through a
    window run the windowed kernel form, the others the plain one), by the final state.
    G6e / G6f (round 4): the same on bench.py's curvilinear workloads (`--warp 1.0` C3; `--config c5shape` with its island and polynya)."""
    from conftest import traj_digest_row
    g = golden(name)
    trk, yx0, K = _cut_tracker(g)
    kstrt, Nt = int(g["kstrt"]), int(g["Nt"])
    windows = mode.split("_")[1:2] == ["w"]
    key = "s%d%s" % (strat, "w" if windows else "")
    trk.ctx.set_params(float(g["rdt"]), strat, 0.1)
    win = (g["rec_first"], g["rec_last"]) if windows else (None, None)
    trk.set_buoys(yx0, g["jiT0"], *win, sort=not mode.endswith("unsorted"))
    trk.ctx.set_resort(0 if mode.endswith("unsorted") else 24)
    if mode.startswith("step"):
        dg = g["digest_" + key]
        for jt in range(Nt):
            jrec = jt + kstrt
            trk.step(jrec, jrec % K)
            pn, mn = trk.record(jrec)
            if windows:                                              # the driver pre-writes a late starter's seed position (:289-312)
                opening = (g["rec_first"] - kstrt) == (jt + 1)
                pn[opening] = yx0[opening]; mn[opening] = 1
            st = trk.state()
            assert np.array_equal(traj_digest_row(pn, mn, st["vJIt"], st["iAlive"]), dg[jt + 1]), jt
    else:
        trk.ctx.set_tuning(fuse=32)
        trk.ctx.launch_stats(reset=True)
        trk.ctx.run(kstrt % K, kstrt, Nt)
        ls = trk.ctx.launch_stats()
        assert ls["fused_records"] + ls["step_launches"] == Nt and ls["fused_launches"] >= 3
    st = trk.state()
    assert np.array_equal(st["vJIt"], g["jiT_end_" + key]) and np.array_equal(st["iAlive"], g["alive_end_" + key])
    assert np.array_equal(st["yx"], g["last_pos_" + key])
    assert int(g["codes_" + key].sum()) > 0.05 * Nt * len(yx0)


def test_g6ef_the_products_own_seeding_gives_the_cuts_host_cells(golden):
    """the buoys of G6e / G6f are seeded by the REFERENCE (FindContainingCell; SeedInit): the product's own seeding of the same
    candidates -- what bench.py does at 1e7 -- keeps the same buoys in the same host cells"""
    from conftest import baseline_cut_case
    g = golden("g6e_c3warp_cut.npz")
    trk, yx0, _ = _cut_tracker(g)
    grid = baseline_cut_case(g)[0]
    found, ji = trk.ctx.find_cells(yx0, syn.nearest_t_index(grid, yx0).astype(np.int32))
    assert found.all() and np.array_equal(ji, g["jiT0"])
    g = golden("g6f_c5shape_cut.npz")
    trk, yx0, _ = _cut_tracker(g)
    grid = baseline_cut_case(g)[0]
    Nj, Ni = grid["Yf"].shape
    nAll, bseed, nP = (int(x) for x in g["buoys"])
    ncand = int(g["cand_idx"][-1]) + 1
    rng = np.random.default_rng(bseed)
    cand = np.stack([rng.uniform(grid["Yt"].min() + 30, grid["Yt"].max() - 30, nAll), rng.uniform(grid["Xt"].min() + 30, grid["Xt"].max() - 30, nAll)],
                    axis=1)[:ncand].astype(np.float32).astype(np.float64)
    ll = orc.CartNPSkm2Geo1D(np.stack([grid["Yt"].ravel(), grid["Xt"].ravel()], axis=1))
    latT, lonT = np.ascontiguousarray(ll[:, 0].reshape(Nj, Ni)), np.ascontiguousarray(np.mod(ll[:, 1], 360.).reshape(Nj, Ni))
    assert latT.sum() == float(g["lat_sum"]) and lonT.sum() == float(g["lon_sum"])
    sll = orc.CartNPSkm2Geo1D(cand); sll[:, 1] = np.mod(sll[:, 1], 360.)
    sic0 = np.ones((Nj, Ni)); j0, j1, i0, i1 = (int(x) for x in g["polynya"]); sic0[j0:j1, i0:i1] = 0.03
    ji, keep, why = trk.ctx.seed_init(sll, cand, latT, lonT, grid["resol"], sic0)
    k = np.where(keep == 1)[0]
    assert np.array_equal(k, g["cand_idx"]) and np.array_equal(ji[k], g["jiT0"]) and ncand - len(k) == int(g["seed_cancelled"])


def test_g6cd_release_the_shared_trackers():
    for t in _CUT_TRK.values():
        t.close()
    _CUT_TRK.clear()
    from conftest import _CUT_CACHE
    _CUT_CACHE.clear()


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


@pytest.mark.parametrize("windowed", [False, True])
def test_fused_run_equals_stepping(windowed):
    """sitrk_run with several records per launch == one launch per record == the oracle."""
    grid = syn.make_grid(120, 128, dkm=4.0, warp=1.0)
    K, Nt = 35, 81
    u, v, sic = syn.make_fields(grid, K=K, seed=13, umax=0.9, drift=0.3, ripple=0.1)
    tmask = grid["tmask"].copy(); tmask[50:56, 60:70] = 0
    sic[:, 20:30, 20:50] = 0.02
    _, yx = syn.make_buoys(grid, 20000, seed=4, frac=0.75)
    rng = np.random.default_rng(1)
    res = []
    for fuse in (1, 3, 8, 32):
        trk = make_tracker(grid, tmask, K)
        found, ji, _ = sit.FindContainingCell(yx, syn.nearest_t_plane(grid, yx), ctx=trk.ctx)
        n = int(found.sum())
        first = (2 + rng.integers(0, 9, n)) if windowed else None
        last = (2 + Nt - 1 - rng.integers(0, 9, n)) if windowed else None
        rng = np.random.default_rng(1)                      # same windows for every variant
        trk.set_buoys(yx[found], ji[found], first, last)
        trk.ctx.set_tuning(fuse=fuse)
        trk.ctx.set_resort(11)
        for k in range(K):
            trk.load_record(k, u[k], v[k], sic[k])
        trk.ctx.run(2 % K, 2, Nt)
        res.append(trk.state())
        trk.close()
    for r in res[1:]:
        for key in ("yx", "vJIt", "iAlive", "kill_rec"):
            assert np.array_equal(r[key], res[0][key]), key
    g2 = dict(grid); g2["tmask"] = tmask
    ref = orc.Tracker(g2, yx[found], ji[found], rec_first=first, rec_last=last, nthreads=8)
    for s in range(Nt):
        ref.step(2 + s, u[(2 + s) % K], v[(2 + s) % K], sic[(2 + s) % K], want_out=False)
    assert np.array_equal(res[0]["yx"], ref.pos) and np.array_equal(res[0]["vJIt"], ref.jiT)
    assert np.array_equal(res[0]["iAlive"], ref.alive)
    assert 0 < ref.alive.sum() < len(ref.alive)


def test_rim_cells_and_negative_index_wrap():
    """Buoys hosted by row/column 1 (possible after FindContainingCell moved them off the nearest T-point):
    NewHostCell then reads F[jT-2] = F[-1], which numpy wraps to the last row (reference tracking.py:219,240-242);
    leaving towards row 0 kills them (Survive rim test)."""
    Nj, Ni = 24, 26
    grid = syn.make_grid(Nj, Ni, dkm=4.0, warp=1.0)
    tmask = np.ones((Nj, Ni), dtype=np.int8)
    u = np.full((1, Nj, Ni), -0.55, dtype=np.float32); v = np.full((1, Nj, Ni), -0.7, dtype=np.float32)
    u[0, :, ::3] = 0.4; v[0, ::4, :] = 0.3
    sic = np.ones((1, Nj, Ni), dtype=np.float32)
    ji = np.array([[1, i] for i in range(1, Ni - 1)] + [[j, 1] for j in range(2, Nj - 1)] + [[Nj - 2, i] for i in range(2, Ni - 1)]
                  + [[j, Ni - 2] for j in range(2, Nj - 2)], dtype=np.int64)
    yx = np.stack([grid["Yt"][ji[:, 0], ji[:, 1]], grid["Xt"][ji[:, 0], ji[:, 1]]], axis=1)
    ok = np.array([orc.IsInsideQuadrangle(yx[b, 0], yx[b, 1], np.array(
        [[grid["Yf"][j - 1, i - 1], grid["Xf"][j - 1, i - 1]], [grid["Yf"][j - 1, i], grid["Xf"][j - 1, i]],
         [grid["Yf"][j, i], grid["Xf"][j, i]], [grid["Yf"][j, i - 1], grid["Xf"][j, i - 1]]])) for b, (j, i) in enumerate(ji)])
    yx, ji = yx[ok], ji[ok]
    trk = make_tracker(grid, tmask, 1)
    trk.set_buoys(yx, ji)
    trk.load_record(0, u[0], v[0], sic[0])
    g2 = dict(grid); g2["tmask"] = tmask
    ref = orc.Tracker(g2, yx, ji)
    for jrec in range(6):
        trk.step(jrec, 0)
        pn, mn = trk.record(jrec)
        rp, rm = ref.step(jrec, u[0], v[0], sic[0])
        assert np.array_equal(mn, rm) and np.array_equal(pn, rp), jrec
    st = trk.state()
    assert np.array_equal(st["vJIt"], ref.jiT) and np.array_equal(st["iAlive"], ref.alive)
    assert (ref.alive == 0).sum() > 10 and ref.jiT.min() == 0
    trk.close()
    # The fused entry point never lets such a buoy set into advect_run_kernel (its table-driven crossing path has no
    # negative-index wrap): sitrk_run steps it record by record -- same answers, and the launch statistics say so.  A set
    # that starts two cells off the rim does go through the fused kernel, and buoys that walk into row/column 1 die there.
    for inner in (False, True):
        trk = make_tracker(grid, tmask, 4)
        if inner:
            jj = np.array([[2, i] for i in range(2, Ni - 2)] + [[j, 2] for j in range(3, Nj - 2)], dtype=np.int64)
            yy = np.stack([grid["Yt"][jj[:, 0], jj[:, 1]], grid["Xt"][jj[:, 0], jj[:, 1]]], axis=1)
            found, jj2, _ = sit.FindContainingCell(yy, jj, ctx=trk.ctx)
            yy, jj = yy[found & np.all(jj2 == jj, axis=1)], jj[found & np.all(jj2 == jj, axis=1)]
        else:
            yy, jj = yx, ji
        for k in range(4):
            trk.load_record(k, u[0], v[0], sic[0])
        trk.set_buoys(yy, jj)
        ref = orc.Tracker(g2, yy, jj)
        trk.ctx.launch_stats(reset=True)
        trk.ctx.run(0, 0, 8)
        for jrec in range(8):
            ref.step(jrec, u[0], v[0], sic[0], want_out=False)
        st = trk.state()
        assert np.array_equal(st["yx"], ref.pos) and np.array_equal(st["vJIt"], ref.jiT) and np.array_equal(st["iAlive"], ref.alive)
        ls = trk.ctx.launch_stats()
        assert (ls["fused_launches"] > 0) == inner and (ls["step_launches"] == 8) == (not inner), ls
        assert (ref.alive == 0).sum() > 5
        trk.close()


def _step_ulps(x, k):
    for _ in range(abs(k)):
        x = np.nextafter(x, np.inf if k > 0 else -np.inf)
    return x


def test_division_free_cell_test_at_the_edges(ctx):
    """The hot loop decides IsInsideQuadrangle without dividing unless the buoy is within ~1e-11 km of an edge
    (sitrk_geom.h::inside_quad_hot).  Points ON the reference's own `xints`, a few ulps either side of it, on
    vertices and on the y of vertices, for warped, regular (vertical/horizontal edges) and far-from-origin cells:
    identical to the oracle, and the probe itself reports any disagreement with the plain form."""
    rng = np.random.default_rng(5)
    n = 6000
    quads = np.empty((n, 4, 2))
    for k in range(n):
        cy, cx = rng.uniform(-4000, 4000, 2)
        d = rng.choice([0.8, 4.0, 12.0])
        w = 0.0 if k % 3 == 0 else 0.25 * d                       # every third cell is an exact rectangle
        base = np.array([[0, 0], [0, d], [d, d], [d, 0]], dtype=np.float64)      # bl, br, ur, ul as (y, x)
        quads[k] = base + [cy, cx] + rng.uniform(-w, w, (4, 2))
        if k % 5 == 0:
            quads[k] = np.round(quads[k] * 4) / 4                 # coordinates on a lattice: exact ties happen
    pts, qq = [], []
    for k in range(n):
        q = quads[k]
        for e in range(4):
            A, B = q[e], q[(e + 1) % 4]
            lo, hi = min(A[0], B[0]), max(A[0], B[0])
            if lo == hi:
                continue
            for y in (rng.uniform(lo, hi), hi, _step_ulps(lo, 1), 0.5 * (lo + hi)):
                X = (y - A[0]) * (B[1] - A[1]) / (B[0] - A[0]) + A[1]        # the reference's expression, same order
                for ku in (-3, -1, 0, 1, 3):
                    pts.append((y, _step_ulps(X, ku))); qq.append(k)
        for vtx in q:                                             # on the vertices and level with them
            pts.append((vtx[0], vtx[1])); qq.append(k)
            pts.append((vtx[0], q[:, 1].mean())); qq.append(k)
    pts = np.array(pts); qq = np.array(qq)
    got = ctx.eval_inside(pts, quads[qq])                         # raises if the two device forms ever differ
    ref = np.array([orc.IsInsideQuadrangle(pts[m, 0], pts[m, 1], quads[qq[m]]) for m in range(0, len(pts), 7)])
    assert np.array_equal(got[::7], ref)
    assert 0.2 < got.mean() < 0.8
    # bulk: random points around random cells, device forms against each other (5e6 points)
    m = 5_000_000
    qi = rng.integers(0, n, m)
    c = quads[qi].mean(axis=1)
    ctx.eval_inside(c + rng.uniform(-8, 8, (m, 2)), quads[qi])
    # non-finite inputs take the plain path
    bad = np.array([[np.nan, 0.], [0., np.nan], [np.inf, 0.], [0., -np.inf], [0., np.inf]])
    assert not ctx.eval_inside(bad + quads[:5].mean(axis=1), quads[:5]).any()


def test_euler_update_without_division(ctx):
    """`r + (vel*rdt)/1000.` through the hot loop's constant division (sitrk_geom.h::div1000): bit-identical to numpy
    for ordinary speeds, signed zeros, subnormals, huge values, inf and NaN."""
    rng = np.random.default_rng(12)
    n = 4_000_000
    vel = rng.standard_normal(n) * 10.0 ** rng.uniform(-12, 4, n)
    vel[:200000] = rng.standard_normal(200000).astype(np.float32)              # f32 velocities, as on disk
    special = np.array([0.0, -0.0, 5e-324, -5e-324, 1e-310, 2.0 ** -1022, 2.0 ** -905, 2.0 ** -912, 2.0 ** -911 * 1.7,
                        2.0 ** 888, 2.0 ** 889, -2.0 ** 1000, 1e300, 1.7e308, np.inf, -np.inf, np.nan, 1e20, 9.96921e36])
    vel[-special.size:] = special
    # binary32 values take the fused loop's own range test (v_cmp_class on the value as loaded): zeros, subnormals, the
    # largest and smallest normals, NetCDF's default fill, infinities
    f32s = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-40, -3e-39, 1.1754944e-38, -1.1754944e-38, 3.4028235e38, -3.4028235e38,
                     9.96921e36, 1e20, np.inf, -np.inf, np.nan, 0.3, -0.25], dtype=np.float32).astype(np.float64)
    vel[-special.size - f32s.size:-special.size] = f32s
    r = rng.uniform(-5000, 5000, n)
    nsp = special.size + f32s.size
    r[-nsp:] = np.where(np.arange(nsp) % 2, -0.0, 0.0)                            # keeps the sign of a zero quotient visible
    for rdt in (3600., 1.0, 900., -3600., 2.0 ** -705, 2.0 ** 701, 2.0 ** -699, 2.0 ** 699):
        with np.errstate(all="ignore"):
            want = r + (vel * rdt) / 1000.
        got = ctx.eval_euler(r, vel, rdt)
        nan = np.isnan(want)
        assert np.array_equal(np.isnan(got), nan)
        assert np.array_equal(got[~nan].view(np.int64), want[~nan].view(np.int64))


def test_extra_linear_interpolation_rule():
    """uv_strategy = 2 is not in the reference; the GPU implementation is held against its independent C restatement,
    and it must differ from the two reference rules while staying between the two face values."""
    grid = syn.make_grid(96, 112, dkm=4.0, warp=1.0)
    K, Nt = 3, 20
    u, v, sic = syn.make_fields(grid, K=K, seed=31, umax=0.6, drift=0.2, ripple=0.15)
    _, yx = syn.make_buoys(grid, 12000, seed=9, frac=0.7)
    out = {}
    for strat in (1, 2):
        trk = make_tracker(grid, grid["tmask"], K, iUVstrategy=strat)
        found, ji, _ = sit.FindContainingCell(yx, syn.nearest_t_plane(grid, yx), ctx=trk.ctx)
        trk.set_buoys(yx[found], ji[found])
        for k in range(K):
            trk.load_record(k, u[k], v[k], sic[k])
        ref = orc.Tracker(grid, yx[found], ji[found], uv_strategy=strat, nthreads=8)
        for s in range(Nt):
            trk.step(s, s % K) if s % 2 else trk.ctx.run(s % K, s, 1)
            rp, rm = ref.step(s, u[s % K], v[s % K], sic[s % K])
        st = trk.state()
        assert np.array_equal(st["yx"], ref.pos) and np.array_equal(st["vJIt"], ref.jiT) and np.array_equal(st["iAlive"], ref.alive)
        out[strat] = st["yx"]
        trk.close()
    assert not np.array_equal(out[1], out[2]) and np.abs(out[1] - out[2]).max() < 30.0


def test_fill_values_nan_and_empty_windows():
    """Land points of NEMO output carry raw fill data (reference si3_part_tracker.py:372-374 assigns masked slabs
    into plain arrays): huge values, NaN.  Buoys that pick them up must behave exactly like in the reference
    arithmetic (NaN positions propagate, every comparison with NaN is false).  Windows may be empty."""
    Nj, Ni = 64, 72
    grid = syn.make_grid(Nj, Ni, dkm=4.0, warp=1.0)
    u, v, sic = syn.make_fields(grid, K=2, seed=6, umax=0.8, drift=0.3)
    tmask = grid["tmask"].copy()
    u[:, 20:26, 20:30] = np.nan; v[:, 30:36, 40:50] = 1.0e20; u[:, 40:44, 10:20] = -1.0e20; v[:, 10:14, 50:60] = np.inf
    _, yx = syn.make_buoys(grid, 6000, seed=8, frac=0.8)
    trk = make_tracker(grid, tmask, 2)
    found, ji, _ = sit.FindContainingCell(yx, syn.nearest_t_plane(grid, yx), ctx=trk.ctx)
    yx, ji = yx[found], ji[found]
    n = len(yx)
    first = np.zeros(n, dtype=np.int64); last = np.full(n, 100, dtype=np.int64)
    first[::9] = 5; last[::9] = 3              # empty window: never steps
    first[1::9] = 4; last[1::9] = 4            # one single record
    trk.set_buoys(yx, ji, first, last)
    trk.ctx.set_tuning(fuse=1)
    g2 = dict(grid); g2["tmask"] = tmask
    ref = orc.Tracker(g2, yx, ji, rec_first=first, rec_last=last)
    for k in range(2):
        trk.load_record(k, u[k], v[k], sic[k])
    for jrec in range(10):
        trk.step(jrec, jrec % 2)
        pn, mn = trk.record(jrec)
        rp, rm = ref.step(jrec, u[jrec % 2], v[jrec % 2], sic[jrec % 2])
        assert np.array_equal(mn, rm), jrec
        assert np.array_equal(pn, rp, equal_nan=True), jrec
    st = trk.state()
    assert np.array_equal(st["vJIt"], ref.jiT) and np.array_equal(st["iAlive"], ref.alive)
    assert np.array_equal(st["yx"], ref.pos, equal_nan=True)
    assert np.isnan(ref.pos).any() and np.array_equal(st["yx"][::9], yx[::9])
    # the fused path agrees too
    trk2 = make_tracker(grid, tmask, 2)
    trk2.set_buoys(yx, ji, first, last)
    for k in range(2):
        trk2.load_record(k, u[k], v[k], sic[k])
    trk2.ctx.run(0, 0, 10)
    st2 = trk2.state()
    assert np.array_equal(st2["yx"], st["yx"], equal_nan=True) and np.array_equal(st2["vJIt"], st["vJIt"])
    assert np.array_equal(st2["kill_rec"], st["kill_rec"])
    trk.close(); trk2.close()


def test_empty_buoy_sets_and_empty_seed_sets():
    """A rank of a multi-GPU run may own no buoy at all (more ranks than seeds, or every seed of its range cancelled):
    every entry point on its path takes n = 0 -- set_buoys, step, run (fused entry), sort, count_alive, buoy_rows, fetch,
    fetch_record, the locate functions, the projections and nemoSeed on an ice-free field."""
    grid = syn.make_grid(48, 56, dkm=4.0, warp=0.5)
    K = 3
    u, v, sic = syn.make_fields(grid, K=K, seed=3, umax=0.3)
    trk = make_tracker(grid, grid["tmask"], K)
    for k in range(K):
        trk.load_record(k, u[k], v[k], sic[k])
    none2 = np.zeros((0, 2))
    for windows in (False, True):
        kw = dict(rec_first=np.zeros(0, np.int32), rec_last=np.zeros(0, np.int32)) if windows else {}
        trk.ctx.set_buoys(none2, np.zeros((0, 2), dtype=np.int32), **kw)
        trk.ctx.step(0, 0)
        trk.ctx.run(0, 1, 5)
        trk.ctx.sort_buoys()
        assert trk.ctx.count_alive() == 0
        st = trk.ctx.fetch()
        assert st["yx"].shape == (0, 2) and st["jiT"].shape == (0, 2) and st["alive"].shape == (0,) and st["kill_rec"].shape == (0,)
        pos, msk = trk.ctx.fetch_record(1)
        assert pos.shape == (0, 2) and msk.shape == (0,)
    ctx = trk.ctx
    found, ji = ctx.find_cells(none2, np.zeros((0, 2), dtype=np.int32))
    assert found.shape == (0,) and ji.shape == (0, 2)
    assert ctx.cart2geo(none2).shape == (0, 2) and ctx.geo2cart(none2).shape == (0, 2)
    # SeedInit with no seed at all, and with seeds that are all cancelled (open water everywhere)
    latT = np.full(grid["Yf"].shape, 80.0); lonT = np.full(grid["Yf"].shape, 10.0)
    out = sit.SeedInit(np.zeros(0, dtype=np.int64), none2, none2, latT, lonT, grid["Yf"], grid["Xf"], np.full(grid["Yf"].shape, 4.0), grid["tmask"],
                       xIceConc=np.zeros(grid["Yf"].shape), ctx=ctx)
    assert out[0] == 0 and out[4].shape == (0, 2) and out[5].shape == (0, 2, 4) and out[6].size == 0
    sg = np.column_stack([np.full(7, 80.0), np.full(7, 10.0)])
    _, yx7 = syn.make_buoys(grid, 7, seed=5, frac=0.3)
    out = sit.SeedInit(np.arange(7), sg, yx7, latT, lonT, grid["Yf"], grid["Xf"], np.full(grid["Yf"].shape, 4.0), grid["tmask"],
                       xIceConc=np.zeros(grid["Yf"].shape), ctx=ctx)
    assert out[0] == 0 and out[3].size == 0 and out[6].size == 0
    # a rank that had buoys and lost them to another rank: back to a real set afterwards
    _, yx = syn.make_buoys(grid, 500, seed=4, frac=0.5)
    found, ji, _ = sit.FindContainingCell(yx, syn.nearest_t_plane(grid, yx), ctx=ctx)
    trk.set_buoys(yx[found], ji[found])
    trk.ctx.run(0, 0, 3)
    ref = orc.Tracker(grid, yx[found], ji[found], nthreads=4)
    for s_ in range(3):
        ref.step(s_, u[s_ % K].astype("f8"), v[s_ % K].astype("f8"), sic[s_ % K].astype("f8"), want_out=False)
    st = trk.ctx.fetch()
    assert np.array_equal(st["yx"], ref.pos) and np.array_equal(st["jiT"], ref.jiT)


def test_everything_dies_and_stays_dead():
    grid = syn.make_grid(32, 32, dkm=4.0)
    u, v, sic = syn.make_fields(grid, K=1, umax=0.3)
    sic[:] = 0.0                                  # open water everywhere: any crossing kills
    _, yx = syn.make_buoys(grid, 500, seed=2, frac=0.5)
    trk = make_tracker(grid, grid["tmask"], 1)
    found, ji, _ = sit.FindContainingCell(yx, syn.nearest_t_guess(grid, yx), ctx=trk.ctx)
    trk.set_buoys(yx[found], ji[found])
    trk.load_record(0, 10 * u[0], 10 * v[0], sic[0])
    ref = orc.Tracker(grid, yx[found], ji[found])
    trk.ctx.run(0, 0, 40)
    for jrec in range(40):
        ref.step(jrec, 10 * u[0], 10 * v[0], sic[0], want_out=False)
    st = trk.state()
    assert np.array_equal(st["iAlive"], ref.alive) and np.array_equal(st["yx"], ref.pos) and np.array_equal(st["vJIt"], ref.jiT)
    assert trk.alive_count() == int(ref.alive.sum()) and ref.alive.sum() < 0.5 * len(ref.alive)
    pos, msk = trk.record(39)
    assert np.all(pos[st["iAlive"] == 0][st["kill_rec"][st["iAlive"] == 0] < 39] == -9999.)
    trk.close()


@pytest.mark.parametrize("fdt", [np.float32, np.float64])
def test_row_band_ingest_equals_full_ingest(fdt):
    """Only rows [jmin-2, jmax+3) of a record are uploaded; the rest of the slot is poisoned (NaN velocities, zero ice
    => Survive kills).  Trajectories must equal those of full uploads: nothing outside the band is ever read."""
    Nj, Ni = 200, 96
    grid = syn.make_grid(Nj, Ni, dkm=4.0, warp=1.0)
    K, Nt = 6, 45
    u, v, sic = syn.make_fields(grid, K=K, seed=21, umax=1.1, drift=0.7, ripple=0.1, dtype=fdt)   # fast: ~1 cell per record
    tmask = grid["tmask"].copy(); tmask[88:92, 30:50] = 0
    sic[:, 60:70, 10:30] = 0.03
    _, yx = syn.make_buoys(grid, 15000, seed=12, frac=0.9)
    yx = yx[np.abs(yx[:, 0]) < 120.]                         # a band of rows in the middle of the mesh
    res = {}
    for mode in ("full", "band"):
        trk = make_tracker(grid, tmask, 1, field_dtype=fdt)
        found, ji, _ = sit.FindContainingCell(yx, syn.nearest_t_plane(grid, yx), ctx=trk.ctx)
        trk.set_buoys(yx[found], ji[found])
        poison = (np.full((Nj, Ni), np.nan, dtype=fdt), np.full((Nj, Ni), np.nan, dtype=fdt), np.zeros((Nj, Ni), dtype=fdt))
        bands = []
        for s in range(Nt):
            k = s % K
            if mode == "full":
                trk.ctx.push_record(0, u[k], v[k], sic[k])
            else:
                trk.ctx.push_record(0, *poison)
                j0, j1 = trk.ctx.band()
                bands.append((j0, j1))
                trk.ctx.push_record_rows(0, j0, j1, u[k][j0:j1], v[k][j0:j1], sic[k][j0:j1])
            trk.step(s, 0)
        res[mode] = trk.state()
        if mode == "band":
            assert max(b[1] - b[0] for b in bands) < 0.8 * Nj and bands[0] != bands[-1]      # a real band, and it moved
            jmin, jmax = trk.ctx.buoy_rows()
            alive = res[mode]["iAlive"] == 1
            assert (jmin, jmax) == (res[mode]["vJIt"][alive, 0].min(), res[mode]["vJIt"][alive, 0].max())
        trk.close()
    for key in ("yx", "vJIt", "iAlive", "kill_rec"):
        assert np.array_equal(res["band"][key], res["full"][key]), key
    assert not np.isnan(res["band"]["yx"]).any() and 0 < res["band"]["iAlive"].sum() < len(res["band"]["iAlive"])

@pytest.mark.parametrize("fdt", [np.float32, np.float64])
@pytest.mark.parametrize("Ni,tile", [(160, 0), (160, 1), (158, 0)])
@pytest.mark.parametrize("mode", ["step", "run", "commit", "commit_run", "commit_run_async"])
def test_box_ingest_equals_full_ingest(fdt, Ni, tile, mode):
    """Round 4: only the BOX rows [jmin-2, jmax+3) x columns [imin-2, imax+3) of a record is uploaded (three strided DMAs out of
    the pinned staging) and only its Survive bytes are derived; everything outside -- above, below, LEFT and RIGHT -- is poisoned
    (NaN velocities, zero ice => Survive kills).  Trajectories must equal those of full uploads, record by record (`step`),
    through fused launches whose boxes are widened by the records of a launch (`run`), and when the slab arrives in device
    memory whole and only the box is committed (`commit`: what an RCCL broadcast or bench.py's fresh-records leg does).
    Both Survive kernels (Ni = 158: rows not 16-byte aligned -> the LDS-tile form with unaligned box edges)."""
    Nj = 200
    grid = syn.make_grid(Nj, Ni, dkm=4.0, warp=1.0)
    K, Nt = 6, 42
    u, v, sic = syn.make_fields(grid, K=K, seed=21, umax=1.1, drift=0.7, ripple=0.1, dtype=fdt)   # fast: ~1 cell per record
    tmask = grid["tmask"].copy(); tmask[88:92, 60:80] = 0
    sic[:, 60:70, 70:95] = 0.03
    _, yx = syn.make_buoys(grid, 30000, seed=12, frac=0.9)
    yx = yx[(np.abs(yx[:, 0]) < 120.) & (np.abs(yx[:, 1]) < 100.)]      # a box in the middle of the mesh
    res = {}
    nfuse = 3
    for ingest in ("full", "box"):
        trk = make_tracker(grid, tmask, nfuse, field_dtype=fdt)
        ctx = trk.ctx
        ctx.set_tuning(survive_tile=tile, fuse=nfuse)
        found, ji, _ = sit.FindContainingCell(yx, syn.nearest_t_plane(grid, yx), ctx=ctx)
        trk.set_buoys(yx[found], ji[found])
        poison = (np.full((Nj, Ni), np.nan, dtype=fdt), np.full((Nj, Ni), np.nan, dtype=fdt), np.zeros((Nj, Ni), dtype=fdt))
        boxes = []

        def deliver(slot, k, age):
            if ingest == "full":
                ctx.push_record(slot, u[k], v[k], sic[k])
                return
            j0, j1, i0, i1 = ctx.box(age)
            boxes.append((j0, j1, i0, i1))
            if mode in ("commit", "commit_run", "commit_run_async"):
                # the whole slab is written in device memory, poisoned outside the box; only the box is committed
                import torch
                from sitrack_amd import distributed as sd
                comp = [p_.copy() for p_ in poison]
                for dst, src in zip(comp, (u[k], v[k], sic[k])):
                    dst[j0:j1, i0:i1] = src[j0:j1, i0:i1]
                ctx.sync()
                sd.slot_tensor(ctx, slot).copy_(torch.from_numpy(sd.pack_slab(*comp, fdt)))
                torch.cuda.synchronize()
                if mode == "commit":
                    ctx.commit_record_box(slot, j0, j1, i0, i1)
                elif slot == nfuse - 1:                             # the launch's records in ONE Survive launch
                    assert len(set(boxes[-nfuse:])) == 1
                    # (_async: on the library's ingest stream, behind the launch that last read these slots' bytes)
                    ctx.commit_records_box(0, nfuse, j0, j1, i0, i1, on_ingest_stream=(mode == "commit_run_async"))
            else:
                ctx.push_record(slot, *poison)
                ctx.push_record_box(slot, j0, j1, i0, i1, u[k][j0:j1, i0:i1], v[k][j0:j1, i0:i1], sic[k][j0:j1, i0:i1])

        if mode in ("run", "commit_run", "commit_run_async"):
            for b in range(Nt // nfuse):
                for r in range(nfuse):
                    deliver(r, (b * nfuse + r) % K, nfuse - 1)      # record r of a launch is stepped r records after the evaluation
                ctx.run(0, b * nfuse, nfuse)
        else:
            for s in range(Nt):
                deliver(0, s % K, 0)
                trk.step(s, 0)
        res[ingest] = trk.state()
        if ingest == "box":
            assert max((b[1] - b[0]) * (b[3] - b[2]) for b in boxes) < 0.6 * Nj * Ni and boxes[0] != boxes[-1]   # a real box, and it moved
            assert all(b[2] % 4 == 0 and (b[3] % 4 == 0 or b[3] == Ni) for b in boxes)
            jmin, jmax, imin, imax = ctx.buoy_box()
            alive = res[ingest]["iAlive"] == 1
            cells = res[ingest]["vJIt"][alive]
            assert (jmin, jmax, imin, imax) == (cells[:, 0].min(), cells[:, 0].max(), cells[:, 1].min(), cells[:, 1].max())
            assert ctx.launch_stats()["fused_launches"] == (Nt // nfuse if mode.endswith(("run", "run_async")) else 0)
        trk.close()
    for key in ("yx", "vJIt", "iAlive", "kill_rec"):
        assert np.array_equal(res["box"][key], res["full"][key]), key
    assert not np.isnan(res["box"]["yx"]).any() and 0 < res["box"]["iAlive"].sum() < len(res["box"]["iAlive"])


@pytest.mark.parametrize("on_ingest", [False, True])
def test_batched_box_commit_of_twelve_records(on_ingest):
    """`sitrk_commit_records_box` on a batch larger than the tests of the fused launch use elsewhere (3 records): twelve slabs written whole into device memory with everything outside the box poisoned, their stale
    Survive bytes those of an ice-free record, ONE commit for all twelve, one fused launch of twelve records -- against the oracle."""
    import torch
    from sitrack_amd import distributed as sd
    Nj, Ni, K = 150, 256, 12
    grid = syn.make_grid(Nj, Ni, dkm=4.0, warp=1.0)
    u, v, sic = syn.make_fields(grid, K=K, seed=31, umax=1.0, drift=0.6, ripple=0.1)
    tmask = grid["tmask"].copy(); tmask[70:74, 100:130] = 0
    for k in range(K):
        sic[k, 40:52, 60 + 3 * k:90 + 3 * k] = 0.03
    _, yx = syn.make_buoys(grid, 40000, seed=5, frac=0.9)
    yx = yx[(np.abs(yx[:, 0]) < 150.) & (np.abs(yx[:, 1]) < 300.)]
    trk = make_tracker(grid, tmask, K)
    try:
        ctx = trk.ctx
        found, ji, _ = sit.FindContainingCell(yx, syn.nearest_t_plane(grid, yx), ctx=ctx)
        yx, ji = yx[found], ji[found]
        trk.set_buoys(yx, ji)
        zero = np.zeros((Nj, Ni), dtype=np.float32)
        for k in range(K):
            ctx.push_record(k, zero, zero, zero)                  # every Survive byte of every slot: kill
        box = ctx.box(K - 1)
        assert (box[1] - box[0]) * (box[3] - box[2]) < 0.8 * Nj * Ni
        ctx.sync()
        for k in range(K):
            comp = [np.full((Nj, Ni), np.nan, dtype=np.float32), np.full((Nj, Ni), np.nan, dtype=np.float32), zero.copy()]
            for dst, src in zip(comp, (u[k], v[k], sic[k])):
                dst[box[0]:box[1], box[2]:box[3]] = src[box[0]:box[1], box[2]:box[3]]
            sd.slot_tensor(ctx, k).copy_(torch.from_numpy(sd.pack_slab(*comp, np.float32)))
        torch.cuda.synchronize()
        ctx.commit_records_box(0, K, *box, on_ingest_stream=on_ingest)
        ctx.set_tuning(fuse=K)
        ctx.run(0, 0, K)
        st = trk.state()
        g2 = dict(grid); g2["tmask"] = tmask
        ref = orc.Tracker(g2, yx, ji, nthreads=4)
        for k in range(K):
            ref.step(k, u[k].astype(np.float64), v[k].astype(np.float64), sic[k].astype(np.float64), want_out=False)
        assert np.array_equal(st["yx"], ref.pos) and np.array_equal(st["vJIt"], ref.jiT) and np.array_equal(st["iAlive"], ref.alive)
        assert 0 < (ref.alive == 0).sum() < len(yx) and ctx.launch_stats()["fused_launches"] == 1
    finally:
        trk.close()


def test_survive_bytes_of_a_box_equal_those_of_the_whole_record(ctx):
    """The Survive bytes a box commit derives are those of the whole-record pass wherever the box determines them, for boxes at
    every alignment of their four edges, through the trajectories' own reader of the bytes: a cloud stepped once per box."""
    Nj, Ni = 64, 96
    grid = syn.make_grid(Nj, Ni, dkm=4.0, warp=0.0)
    rng = np.random.default_rng(5)
    tmask = (rng.random((Nj, Ni)) > 0.05).astype(np.int8)
    sic = rng.choice([0.0, 0.08, 0.1, 0.12, 1.0], size=(Nj, Ni)).astype(np.float32)
    u = np.full((Nj, Ni), 1.2, dtype=np.float32); v = np.full((Nj, Ni), -1.2, dtype=np.float32)     # every buoy crosses: 4.3 km per record
    g2 = dict(grid); g2["tmask"] = tmask
    for (j0, j1, i0, i1) in [(10, 40, 8, 60), (11, 41, 9, 61), (9, 50, 10, 62), (12, 39, 11, 63), (5, 60, 4, 92), (20, 30, 31, 49)]:
        # buoys whose reach [j-2, j+3) x [i-2, i+3) lies inside the box
        jj, ii = np.meshgrid(np.arange(j0 + 2, j1 - 2), np.arange(i0 + 2, i1 - 2), indexing="ij")
        ji = np.stack([jj.ravel(), ii.ravel()], axis=1).astype(np.int32)
        yx = np.stack([grid["Yt"][ji[:, 0], ji[:, 1]] + 0.3, grid["Xt"][ji[:, 0], ji[:, 1]] - 0.4], axis=1)
        out = {}
        for tile in (0, 1):
            trk = make_tracker(grid, tmask, 1)
            trk.ctx.set_tuning(survive_tile=tile)
            trk.set_buoys(yx, ji)
            assert trk.ctx.box() == (j0, j1, i0 - i0 % 4, min(Ni, -(-i1 // 4) * 4))
            trk.ctx.push_record_box(0, j0, j1, i0, i1, u[j0:j1, i0:i1], v[j0:j1, i0:i1], sic[j0:j1, i0:i1])
            trk.ctx.step(0, 0)                                   # exactly the box: fine
            out[tile] = trk.state()
            if out[tile]["iAlive"].any():
                with pytest.raises(sit.SitrkError, match="can touch"):
                    trk.ctx.step(0, 1)                               # one record later the box is one cell too narrow all around
            trk.close()
        ref = orc.Tracker(g2, yx, ji, nthreads=4)
        ref.step(0, u.astype(np.float64), v.astype(np.float64), sic.astype(np.float64), want_out=False)
        assert ref.ncross == len(yx) and 0 < ref.alive.sum() < len(yx)
        for tile in (0, 1):
            assert np.array_equal(out[tile]["iAlive"], ref.alive) and np.array_equal(out[tile]["vJIt"], ref.jiT), (j0, j1, i0, i1, tile)


@pytest.mark.parametrize("async_survive", [0, 1])
def test_async_ingest_ring_and_staging(async_survive):
    """Library-owned ingest (include/sitrk.h: pinned staging, copy stream, events): records are pushed from TEMPORARY host
    arrays that are scribbled over right after the call, pushed two batches ahead of the launches that use them, read
    straight into the pinned staging, and a slot is re-uploaded while the launch that used it may still be running --
    the trajectories equal the oracle's record by record."""
    Nj, Ni, K, Nt = 96, 120, 8, 64
    grid = syn.make_grid(Nj, Ni, dkm=4.0, warp=1.0)
    u, v, sic = syn.make_fields(grid, K=Nt, seed=5, umax=0.9, drift=0.4, ripple=0.1)
    sic[:, 30:40, 50:80] = 0.04
    _, yx = syn.make_buoys(grid, 40000, seed=8, frac=0.8)
    trk = make_tracker(grid, grid["tmask"], K)
    try:
        found, ji, _ = sit.FindContainingCell(yx, syn.nearest_t_plane(grid, yx), ctx=trk.ctx)
        yx, ji = yx[found], ji[found]
        trk.set_buoys(yx, ji)
        ref = orc.Tracker(grid, yx, ji, nthreads=4)
        ctx = trk.ctx
        ctx.set_tuning(async_survive=async_survive)         # 1: the uploads' Survive bytes are derived on the ingest stream (round 4 option)
        ref_done[0] = 0

        def upload(s):
            if s % 3 == 0:                                  # through a temporary that is destroyed at once
                tu, tv, ts = u[s].copy(), v[s].copy(), sic[s].copy()
                ctx.push_record(s % K, tu, tv, ts)
                tu[:] = np.nan; tv[:] = 1e30; ts[:] = 0.
            elif s % 3 == 1:                                # straight into the library's pinned staging
                if s % 2:
                    bu, bv, bs = ctx.stage()
                    bu[...] = u[s]; bv[...] = v[s]; bs[...] = sic[s]
                    ctx.submit(s % K)
                else:
                    # a reader that fails half-way must leave the context usable (sitrk_stage_release), then the real read
                    def bad(bu, bv, bs):
                        bu[...] = np.nan
                        raise IOError("file vanished")
                    with pytest.raises(IOError):
                        ctx.stage_fill(s % K, 0, Nj, bad)

                    def good(bu, bv, bs):
                        bu[...] = u[s]; bv[...] = v[s]; bs[...] = sic[s]
                    ctx.stage_fill(s % K, 0, Nj, good)
            else:                                           # as a row band that happens to be the whole record
                ctx.push_record_rows(s % K, 0, Nj, u[s], v[s], sic[s])
        m = K // 2
        for s in range(m):
            upload(s)
        for b in range(Nt // m):
            ctx.run((b * m) % K, b * m, m)                  # asynchronous
            if b + 1 < Nt // m:
                for s in range((b + 1) * m, (b + 2) * m):   # overwrites the slots of batch b-1 while batch b runs
                    upload(s)
            if b % 3 == 2:
                for s in range(max(0, b - 2) * m, (b + 1) * m):
                    if s >= ref_done[0]:
                        ref.step(s, u[s], v[s], sic[s], want_out=False)
                        ref_done[0] = s + 1
                st = trk.state()
                assert np.array_equal(st["yx"], ref.pos) and np.array_equal(st["vJIt"], ref.jiT) and np.array_equal(st["iAlive"], ref.alive)
        while ref_done[0] < Nt:
            ref.step(ref_done[0], u[ref_done[0]], v[ref_done[0]], sic[ref_done[0]], want_out=False)
            ref_done[0] += 1
        st = trk.state()
        assert np.array_equal(st["yx"], ref.pos) and np.array_equal(st["vJIt"], ref.jiT) and np.array_equal(st["iAlive"], ref.alive)
        assert 0 < st["iAlive"].sum() < len(yx) and ref.ncross > 10000
        stats = ctx.launch_stats()
        assert stats["fused_records"] + stats["step_launches"] == Nt
    finally:
        trk.close()


ref_done = [0]


def test_partly_uploaded_slot_is_checked_against_the_buoys_band():
    """a band that is too narrow must not silently read stale rows.  The library knows which rows of a slot are
    valid and refuses a step whose buoys can reach outside them; untouched rows hold NaN velocities and kill bytes."""
    Nj, Ni = 120, 64
    grid = syn.make_grid(Nj, Ni, dkm=4.0, warp=0.0)
    u, v, sic = syn.make_fields(grid, K=2, seed=2, umax=0.5)
    _, yx = syn.make_buoys(grid, 5000, seed=4, frac=0.5)
    trk = make_tracker(grid, grid["tmask"], 4)
    try:
        found, ji, _ = sit.FindContainingCell(yx, syn.nearest_t_plane(grid, yx), ctx=trk.ctx)
        trk.set_buoys(yx[found], ji[found])
        ctx = trk.ctx
        jlo, jhi = int(ji[found][:, 0].min()), int(ji[found][:, 0].max())
        ctx.push_record_rows(0, jlo - 2, jhi + 3, u[0][jlo - 2:jhi + 3], v[0][jlo - 2:jhi + 3], sic[0][jlo - 2:jhi + 3])
        with pytest.raises(sit.SitrkError, match="sitrk_buoy_rows"):
            ctx.step(0, 0)                                   # band never evaluated since set_buoys
        assert ctx.buoy_rows() == (jlo, jhi)
        ctx.step(0, 0)                                       # exactly the band: fine
        with pytest.raises(sit.SitrkError, match="can touch rows"):
            ctx.step(0, 1)                                   # one record later the band is one row too narrow on each side
        # a fused run ages the band by one row per record: 3 records need 2 more rows than 1 record
        ctx.set_buoys(yx[found], ji[found])
        ctx.buoy_rows()
        for k in range(3):
            ctx.push_record_rows(k, jlo - 3, jhi + 4, u[k % 2][jlo - 3:jhi + 4], v[k % 2][jlo - 3:jhi + 4], sic[k % 2][jlo - 3:jhi + 4])
        with pytest.raises(sit.SitrkError, match="can touch rows"):
            ctx.run(0, 0, 3)
        ctx.run(0, 0, 2)
        # the same in the columns (round 4): rows wide enough, columns one short on the right
        ctx.set_buoys(yx[found], ji[found])
        jmin, jmax, imin, imax = ctx.buoy_box()
        assert (imin, imax) == (int(ji[found][:, 1].min()), int(ji[found][:, 1].max()))
        b = (jmin - 2, jmax + 3, imin - 2, imax + 2)
        ctx.push_record_box(0, *b, u[0][b[0]:b[1], b[2]:b[3]], v[0][b[0]:b[1], b[2]:b[3]], sic[0][b[0]:b[1], b[2]:b[3]])
        with pytest.raises(sit.SitrkError, match="can touch columns"):
            ctx.step(0, 0)
        b = (jmin - 2, jmax + 3, imin - 2, imax + 3)
        ctx.push_record_box(0, *b, u[0][b[0]:b[1], b[2]:b[3]], v[0][b[0]:b[1], b[2]:b[3]], sic[0][b[0]:b[1], b[2]:b[3]])
        ctx.step(0, 0)
        with pytest.raises(sit.SitrkError, match="box out of range"):
            z = np.zeros((Nj, Ni + 4), dtype=np.float32)
            ctx.push_record_box(0, 0, Nj, -4, Ni, z, z, z)
    finally:
        trk.close()


def test_restore_state_keeps_the_alive_means_kill_rec_minus_one_invariant():
    """the re-sort rebuilds a live buoy's kill record as -1 (permute_state_kernel gathers it for dead buoys only), so
    `sitrk_restore_state` must not plant another value on a live buoy: what sitrk_fetch answers may not change across a sort."""
    grid = syn.make_grid(64, 64, dkm=4.0, warp=0.0)
    _, yx = syn.make_buoys(grid, 3000, seed=3, frac=0.6)
    trk = make_tracker(grid, grid["tmask"], 1)
    try:
        found, ji, _ = sit.FindContainingCell(yx, syn.nearest_t_plane(grid, yx), ctx=trk.ctx)
        yx, ji = yx[found], ji[found]
        n = len(yx)
        trk.ctx.set_buoys(yx, ji, sort=False)
        alive = np.ones(n, dtype=np.int8); alive[::5] = 0
        kill_rec = np.full(n, -1, dtype=np.int32); kill_rec[::5] = 7
        kill_rec[1::5] = 3                                # a caller's stale value on LIVE buoys: ignored, stored as -1
        trk.ctx.restore_state(alive, kill_rec)
        before = trk.ctx.fetch()
        trk.ctx.sort_buoys()
        after = trk.ctx.fetch()
        for k in ("yx", "jiT", "alive", "kill_rec"):
            assert np.array_equal(before[k], after[k]), k
        assert np.array_equal(after["alive"], alive)
        assert np.all(after["kill_rec"][alive == 1] == -1) and np.all(after["kill_rec"][alive == 0] == 7)
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


def test_tuning_knobs_do_not_change_results():
    grid = syn.make_grid(128, 160, dkm=4.0, warp=1.0)
    u, v, sic = syn.make_fields(grid, K=3, seed=8, umax=0.7, drift=0.2, ripple=0.1)
    _, yx = syn.make_buoys(grid, 30011, seed=21, frac=0.7)
    res = []
    for knobs in ({}, {"xcd_remap": 1}, {"nt_state": 1}, {"xcd_remap": 1, "nt_state": 1}, {"sort_tile": 8 * 256 + 32},
                  {"sort_tile": 5 * 256 + 7, "nt_state": 1, "xcd_remap": 1},
                  # the fused kernel's LDS patch: none, tiny (most buoys leave it: global fallback), large; XCD grouping; one-record launches
                  {"patch_kb": 0}, {"patch_kb": 1, "patch_margin": 0}, {"patch_kb": 60, "patch_margin": 40}, {"xcd_group": 0},
                  {"xcd_group": 5, "patch_kb": 3}, {"fuse": 1}, {"fuse": 2, "patch_kb": 2}, {"survive_tile": 1},
                  # round 4: where an uploaded record's Survive bytes are derived (ingest stream / compute stream), host fill threads
                  {"async_survive": 1}, {"async_survive": 1, "survive_tile": 1, "fill_threads": 1}, {"fill_threads": 16}):
        trk = make_tracker(grid, grid["tmask"], 3)
        found, ji, _ = sit.FindContainingCell(yx, syn.nearest_t_plane(grid, yx), ctx=trk.ctx)
        trk.ctx.set_tuning(**knobs)
        trk.set_buoys(yx[found], ji[found])
        trk.ctx.set_resort(6)
        for k in range(3):
            trk.load_record(k, u[k], v[k], sic[k])
        trk.ctx.run(0, 0, 25)
        res.append(trk.state())
        trk.close()
    for r in res[1:]:
        for key in ("yx", "vJIt", "iAlive", "kill_rec"):
            assert np.array_equal(r[key], res[0][key])
    with pytest.raises(sit.SitrkError):
        sit.Context(0).set_tuning(bogus=1)


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


def test_seed_init_bruteforce_mode_matches_golden(golden, ctx):
    """The whole-grid scan (the reference's own algorithm) stays available behind a knob and agrees too."""
    g = golden("g5_seedinit.npz")
    ctx.set_grid(g["Yf"], g["Xf"], g["Yf"], g["Xf"], g["Yf"], g["Xf"], g["tmask"])
    ctx.set_tuning(locate_bruteforce=1)
    try:
        out = sit.SeedInit(g["ids"], g["pSG"], g["pSC"], g["latT"], g["lonT"], g["Yf"], g["Xf"], g["resol"], g["tmask"],
                           xIceConc=g["sic"], ctx=ctx)
    finally:
        ctx.set_tuning(locate_bruteforce=0)
    assert out[0] == int(g["nPn"]) and np.array_equal(out[6], g["okeep"]) and np.array_equal(out[4], g["ojiT"])


@pytest.mark.parametrize("bruteforce", [0, 1])
@pytest.mark.parametrize("prec", ["f4", "f8"])
@pytest.mark.parametrize("mesh", ["w", "r", "l"])
def test_seed_init_on_t_and_f_points_matches_reference_golden(golden, ctx, mesh, prec, bruteforce):
    """G5b through `sitrk_seed_init`, both locate modes, bit for bit: seeds exactly on T- and F-points as the reference's
    `nemoSeed(..., platF, plonF)` emits them (tracking.py:426-440).  Mesh "l" holds 581 seeds whose two nearest T-points
    are at EXACTLY equal Haversine distance in numpy: the device's libm must produce the tie as well and break it toward
    the lowest flat index (locate.py:13-20); on meshes "w"/"r" the nearest gaps go down to 1e-6 km."""
    g = golden("g5b_seeds_on_points.npz")
    G = lambda k: g[mesh + "_" + k]                                       # noqa: E731
    P = lambda k: g[mesh + "_" + prec + "_" + k]                          # noqa: E731
    ctx.set_grid(G("Yf"), G("Xf"), G("Yf"), G("Xf"), G("Yf"), G("Xf"), G("tmask"))
    ctx.set_tuning(locate_bruteforce=bruteforce)
    try:
        out = sit.SeedInit(G("ids"), P("pSG"), P("pSC"), G("latT"), G("lonT"), G("Yf"), G("Xf"), G("resol"), G("tmask"),
                           xIceConc=G("sic"), ctx=ctx)
        near, dmin = ctx.nearest_point(P("pSG"), G("latT"), G("lonT"), resolkm=G("resol"), rd_found_km=2.5, max_itr=10)
    finally:
        ctx.set_tuning(locate_bruteforce=0)
    nPn, oSG, oSC, oIDs, ojiT, overt, okeep = out
    assert nPn == int(P("nPn")) and np.array_equal(okeep, P("okeep")) and np.array_equal(oIDs, P("oIDs"))
    assert np.array_equal(ojiT, P("ojiT")) and np.array_equal(overt, P("overt"))
    assert np.array_equal(oSG, P("oSG")) and np.array_equal(oSC, P("oSC"))
    assert np.array_equal(near, P("nearest"))
    assert np.allclose(dmin, P("dmin"), rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("bruteforce", [0, 1])
def test_seed_init_on_the_300x330_mesh_matches_reference_golden(golden, ctx, bruteforce):
    """G5c through `sitrk_seed_init` / `sitrk_nearest_point`: REFERENCE output on a mesh large enough for the bounding-sphere
    hierarchy (2 x 2 superblocks of 16 x 16 blocks of 16 x 16 points), with the pole inside the mesh; both locate modes."""
    from conftest import g5c_mesh
    g = golden("g5c_seedinit_300x330.npz")
    m, tmask, sic = g5c_mesh(g)
    ctx.set_grid(m["Yf"], m["Xf"], m["Yf"], m["Xf"], m["Yf"], m["Xf"], tmask)
    ctx.set_tuning(locate_bruteforce=bruteforce)
    try:
        out = sit.SeedInit(g["ids"], g["pSG"], g["pSC"], m["latT"], m["lonT"], m["Yf"], m["Xf"], m["resol"], tmask, xIceConc=sic, ctx=ctx)
        near, _ = ctx.nearest_point(g["pSG"], m["latT"], m["lonT"], resolkm=m["resol"], rd_found_km=2.5, max_itr=10)
    finally:
        ctx.set_tuning(locate_bruteforce=0)
    assert out[0] == int(g["nPn"]) and np.array_equal(out[6], g["okeep"]) and np.array_equal(out[3], g["oIDs"]) and np.array_equal(out[4], g["ojiT"])
    assert np.array_equal(near, g["nearest"])


@pytest.mark.parametrize("yc,xc", [(-300., 200.), (0., 0.), (-2500., 1800.)])
def test_seed_search_equals_whole_grid_scan(ctx, yc, xc):
    """Bounding-sphere search == exhaustive Haversine argmin, incl. seeds exactly on T- and F-points
    (exact ties between T-points), seeds far outside the mesh, and a mesh that contains the pole."""
    Nj, Ni, dkm = 150, 170, 9.0
    grid = syn.make_grid(Nj, Ni, dkm=dkm, warp=1.0)
    for k in ("Yt", "Yu", "Yv", "Yf"):
        grid[k] = grid[k] + yc
    for k in ("Xt", "Xu", "Xv", "Xf"):
        grid[k] = grid[k] + xc
    llT = orc.CartNPSkm2Geo1D(np.stack([grid["Yt"].ravel(), grid["Xt"].ravel()], axis=1))
    latT = np.ascontiguousarray(llT[:, 0].reshape(Nj, Ni))
    lonT = np.ascontiguousarray(np.mod(llT[:, 1], 360.).reshape(Nj, Ni))
    rng = np.random.default_rng(17)
    n_r = 6000
    yx = np.stack([rng.uniform(grid["Yt"].min() - 60, grid["Yt"].max() + 60, n_r),
                   rng.uniform(grid["Xt"].min() - 60, grid["Xt"].max() + 60, n_r)], axis=1)
    jj, ii = rng.integers(0, Nj, 1500), rng.integers(0, Ni, 1500)
    onT = np.stack([grid["Yt"][jj, ii], grid["Xt"][jj, ii]], axis=1)
    onF = np.stack([grid["Yf"][jj, ii], grid["Xf"][jj, ii]], axis=1)
    far = np.array([[grid["Yt"].min() - 900., xc], [yc, grid["Xt"].max() + 1500.], [yc + 4000., xc - 4000.]])
    yx = np.concatenate([yx, onT, onF, far])
    ll = orc.CartNPSkm2Geo1D(yx)
    ll[:, 1] = np.mod(ll[:, 1], 360.)
    # seeds on T-points get the grid's own lat/lon bit for bit (distance exactly 0)
    ll[n_r:n_r + 1500, 0] = latT[jj, ii]
    ll[n_r:n_r + 1500, 1] = lonT[jj, ii]
    tmask = grid["tmask"].copy(); tmask[60:70, 80:95] = 0
    sic = np.ones((Nj, Ni)); sic[20:40, 20:50] = 0.05
    ctx.set_grid(grid["Yf"], grid["Xf"], grid["Yf"], grid["Xf"], grid["Yf"], grid["Xf"], tmask)
    res = {}
    for mode in (0, 1):
        ctx.set_tuning(locate_bruteforce=mode)
        res[mode] = ctx.seed_init(ll, yx, latT, lonT, grid["resol"], sic)
    ctx.set_tuning(locate_bruteforce=0)
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)
    assert 0 < res[0][1].sum() < len(yx)
    # and the oracle agrees on a subsample (nearest T-point, Survive, containing cell)
    sel = np.r_[0:200, n_r:n_r + 60, n_r + 1500:n_r + 1560, len(yx) - 3:len(yx)]
    o = orc.SeedInit(np.arange(len(sel)), ll[sel], yx[sel], latT, lonT, grid["Yf"], grid["Xf"], grid["resol"], tmask, sic,
                     return_why=True)
    keep_o = np.zeros(len(sel), dtype=np.int8); keep_o[o[6]] = 1
    assert np.array_equal(keep_o, res[0][1][sel])
    assert np.array_equal(o[7], res[0][2][sel])
    assert np.array_equal(o[4], res[0][0][sel][keep_o == 1])


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


def test_inverse_projection_pinned_by_the_reference_fixture(golden, ctx):
    """a12 on the device, direct pin against reference-held data: cart2geo(fixture y_pos, x_pos) == fixture latitude,
    longitude (lon mod 360) at north_star's 1e-5 relative, all 10 buoys of tools/nc/...HSS5.nc__KEEP."""
    g = golden("g7_projection.npz")
    yx = np.stack([g["y_pos"].astype(np.float64), g["x_pos"].astype(np.float64)], axis=1)
    ll = sit.CartNPSkm2Geo1D(yx, ctx=ctx)
    assert np.allclose(ll[:, 0], g["latitude"].astype(np.float64), rtol=1e-5, atol=0)
    assert np.allclose(np.mod(ll[:, 1], 360.), np.mod(g["longitude"].astype(np.float64), 360.), rtol=1e-5, atol=0)
    # and through the path the driver uses at output records (sitrk_fetch_record's latlon, si3_part_tracker.py:493)
    assert np.allclose(ll, orc.CartNPSkm2Geo1D(yx), rtol=1e-12, atol=1e-10)


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
    # size limits are checked before any array is read: 32767 x 65535, and 2^29 cells (32-bit byte offsets in a field)
    import ctypes as C
    dummy = (C.c_double * 4)()
    dm = C.cast(dummy, C.c_void_p)
    for nj, ni in ((32768, 16), (16, 65536), (3, 16), (32767, 16385), (23171, 23171)):
        assert ctx._L.sitrk_set_grid(ctx._h, nj, ni, dm, dm, dm, dm, dm, dm, dm) != 0, (nj, ni)
        assert b"sitrk_set_grid" in ctx._L.sitrk_last_error(ctx._h)
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
        ctx.set_params(3600., 3, 0.1)           # 0, 1 = the reference's rules, 2 = the extra
    with pytest.raises(ValueError):
        ctx.push_record(0, np.zeros((3, 3)), np.zeros((3, 3)), np.zeros((3, 3)))


def test_contexts_release_their_device_memory():
    """40 full life cycles (grid, records, buoys, sort, fused run, SeedInit-style search, fetch, destroy): the free
    device memory at the end is where it was (torch only reads the counter)."""
    import torch
    grid = syn.make_grid(384, 400, dkm=4.0, warp=1.0)
    u, v, sic = syn.make_fields(grid, K=4, seed=3, umax=0.5, drift=0.1)
    _, yx = syn.make_buoys(grid, 150000, seed=4, frac=0.7)
    guess = syn.nearest_t_plane(grid, yx).astype(np.int32)

    def cycle():
        c = sit.Context(0)
        c.set_grid(grid["Yf"], grid["Xf"], grid["Yu"], grid["Xu"], grid["Yv"], grid["Xv"], grid["tmask"])
        c.alloc_records(4, np.float32)
        for k in range(4):
            c.push_record(k, u[k], v[k], sic[k])
        found, ji = c.find_cells(yx, guess)
        c.set_buoys(yx[found], ji[found])
        c.run(0, 0, 20)
        c.nearest_point(np.array([[80., 10.], [75., -30.]]), np.full(grid["Yf"].shape, 80.), np.full(grid["Yf"].shape, 10.), None, 2.5, 10)
        st = c.fetch()
        c.close()
        return st["yx"]

    first = cycle()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(0)
    for _ in range(40):
        assert np.array_equal(cycle(), first)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(0)
    assert free0 - free1 < 64 << 20, (free0, free1)


def test_two_contexts_in_two_threads():
    """One host thread per context, no global state: two trackers stepped concurrently from two
    threads (ctypes releases the GIL during the calls) give what each gives alone."""
    import threading
    grid = syn.make_grid(160, 176, dkm=4.0, warp=1.0)
    cases = []
    for seed in (1, 2):
        u, v, sic = syn.make_fields(grid, K=4, seed=seed, umax=0.8, drift=0.2, ripple=0.1)
        _, yx = syn.make_buoys(grid, 80000, seed=10 + seed, frac=0.75)
        cases.append((u, v, sic, yx))

    def run(case, out, idx, nrep):
        u, v, sic, yx = case
        for _ in range(nrep):
            trk = make_tracker(grid, grid["tmask"], 4, iUVstrategy=idx % 2)
            found, ji, _ = sit.FindContainingCell(yx, syn.nearest_t_plane(grid, yx), ctx=trk.ctx)
            trk.set_buoys(yx[found], ji[found])
            for k in range(4):
                trk.load_record(k, u[k], v[k], sic[k])
            for s in range(0, 48, 6):
                trk.ctx.run(s % 4, s, 5)
                trk.step(s + 5, (s + 5) % 4)
                trk.record(s + 5, latlon=True)
            out[idx] = trk.state()
            trk.close()

    alone = [None, None]
    for i in (0, 1):
        run(cases[i], alone, i, 1)
    both = [None, None]
    th = [threading.Thread(target=run, args=(cases[i], both, i, 3)) for i in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in (0, 1):
        for key in ("yx", "vJIt", "iAlive", "kill_rec"):
            assert np.array_equal(both[i][key], alone[i][key]), (i, key)
