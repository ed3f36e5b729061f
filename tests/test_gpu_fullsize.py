"""Parity at BASELINE.json's full C3 size (4096x4096 grid, 1e7 buoys) through size-independent properties:
independence of the buoys (any partition of the set gives the same trajectories), equivalence of the fused
multi-record launches with record-by-record stepping, invariance under the internal cell sort, plus the oracle
on a subsample that the CPU finishes in seconds."""
import numpy as np
import pytest

import sitrack_amd as sit
from sitrack_amd import synthetic as syn
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c3():
    N, nP, K = 4096, 10_000_000, 4
    grid = syn.make_grid(N, N, dkm=4.0, warp=0.0)
    u, v, sic = syn.make_fields(grid, K=K, seed=2024, umax=0.3, drift=0.05)
    sic[:, 1500:1600, 1500:1700] = 0.02            # some open water so that buoys die
    _, yx = syn.make_buoys(grid, nP, seed=1234, frac=0.6)
    ji = syn.regular_host_cell(grid, yx).astype(np.int32)
    return dict(grid=grid, u=u, v=v, sic=sic, yx=yx, ji=ji, K=K)


def run(c3, sel, nsteps, **knobs):
    g = c3["grid"]
    ctx = sit.Context(0)
    ctx.set_grid(g["Yf"], g["Xf"], g["Yu"], g["Xu"], g["Yv"], g["Xv"], g["tmask"])
    ctx.alloc_records(c3["K"], np.float32)
    for k in range(c3["K"]):
        ctx.push_record(k, c3["u"][k], c3["v"][k], c3["sic"][k])
    sort = knobs.pop("sort", True)
    ctx.set_tuning(**knobs)
    ctx.set_buoys(c3["yx"][sel], c3["ji"][sel], sort=sort)
    ctx.run(0, 0, nsteps)
    out = ctx.fetch()
    ctx.close()
    return out


def test_full_size_properties(c3):
    nP = len(c3["yx"])
    nsteps = 24
    whole = run(c3, slice(None), nsteps)                           # fused launches, tile-major sort
    # (1) fused == record by record
    single = run(c3, slice(None), nsteps, fuse=1)
    for k in ("yx", "jiT", "alive", "kill_rec"):
        assert np.array_equal(whole[k], single[k]), k
    # (2) independence: two halves processed separately == the whole set (what the multi-GPU partition relies on)
    h = nP // 2
    lo, hi = run(c3, slice(0, h), nsteps), run(c3, slice(h, nP), nsteps)
    for k in ("yx", "jiT", "alive", "kill_rec"):
        assert np.array_equal(np.concatenate([lo[k], hi[k]]), whole[k]), k
    # (3) the internal order does not matter: unsorted, row-major key
    plain = run(c3, slice(None), nsteps, sort=False, sort_tile=0, fuse=3)
    assert np.array_equal(plain["yx"], whole["yx"]) and np.array_equal(plain["jiT"], whole["jiT"])
    # (4) checksum of checksums: outputs are in the caller's order, IDs never move
    assert whole["alive"].sum() < nP and (whole["kill_rec"] >= 0).sum() == nP - whole["alive"].sum()
    # (5) the oracle on a subsample spread over the whole set
    sel = np.arange(0, nP, 997)
    ref = orc.Tracker(c3["grid"], c3["yx"][sel], c3["ji"][sel], nthreads=8)
    f64 = [(c3["u"][k].astype(np.float64), c3["v"][k].astype(np.float64), c3["sic"][k].astype(np.float64)) for k in range(c3["K"])]
    for s in range(nsteps):
        ref.step(s, *f64[s % c3["K"]], want_out=False)
    assert np.array_equal(whole["yx"][sel], ref.pos)
    assert np.array_equal(whole["jiT"][sel], ref.jiT) and np.array_equal(whole["alive"][sel], ref.alive)


@pytest.mark.parametrize("N,fused", [(9400, True), (9600, False)])
def test_largest_meshes_at_the_32_bit_offset_limit(N, fused):
    """Maximum sizes.  The fused kernel addresses the geometry with 32-bit byte offsets: 9400 x 9400 cells x 48 B = 4.24e9
    is the last size class below 2^32, 9600 x 9600 (4.42e9) is beyond it and `sitrk_run` steps it record by record
    (64-bit indexing).  Buoys sit in the LAST rows of the mesh, where the offsets are largest; both meshes are held
    against the oracle on every buoy, through the fused entry point."""
    K, nsteps = 2, 12
    grid = syn.make_grid(N, N, dkm=4.0, warp=0.0)
    u, v, sic = syn.make_fields(grid, K=K, seed=11, umax=0.6, drift=0.2)
    sic[:, N - 200:N - 190, N - 400:N - 100] = 0.02
    rng = np.random.default_rng(2)
    nP = 150_000
    j = rng.uniform(N - 330, N - 20, nP); i = rng.uniform(N - 900, N - 20, nP)     # cell indices, last rows and columns
    yx = np.stack([np.interp(j, np.arange(N), grid["Yt"][:, 0]), np.interp(i, np.arange(N), grid["Xt"][0, :])], axis=1)
    ji = syn.regular_host_cell(grid, yx).astype(np.int32)
    assert (ji[:, 0].astype(np.int64) * N + ji[:, 1]).max() * 48 > (3.9e9 if fused else 2 ** 32)
    ctx = sit.Context(0)
    try:
        ctx.set_grid(grid["Yf"], grid["Xf"], grid["Yu"], grid["Xu"], grid["Yv"], grid["Xv"], grid["tmask"])
        found, ji2 = ctx.find_cells(yx, ji)
        assert found.all() and np.array_equal(ji2, ji)
        ctx.alloc_records(K, np.float32)
        for k in range(K):
            ctx.push_record(k, u[k], v[k], sic[k])
        ctx.set_buoys(yx, ji)
        ctx.launch_stats(reset=True)
        for s in range(0, nsteps, K):
            ctx.run(0, s, K)
        st = ctx.launch_stats()
        assert (st["fused_launches"] > 0) == fused and (st["step_launches"] == 0) == fused, st
        out = ctx.fetch()
    finally:
        ctx.close()
    ref = orc.Tracker(grid, yx, ji, nthreads=8)
    f64 = [(u[k].astype(np.float64), v[k].astype(np.float64), sic[k].astype(np.float64)) for k in range(K)]
    for s in range(nsteps):
        ref.step(s, *f64[s % K], want_out=False)
    assert np.array_equal(out["yx"], ref.pos) and np.array_equal(out["jiT"], ref.jiT) and np.array_equal(out["alive"], ref.alive)
    assert 0 < out["alive"].sum() < nP and (out["jiT"] != ji).any()


@pytest.mark.parametrize("Nj,Ni", [(12, 65535), (32767, 12)])
def test_longest_rows_and_columns_of_the_packed_cell(Nj, Ni):
    """Maximum sizes, the other way: the host cell travels packed as 15 bits of jT and 16 bits of iT, so a mesh may have
    65 535 columns or 32 767 rows.  Buoys drift along the far end of such a strip (indices next to the packing limits)
    and off its rim; fused and record-by-record launches against the oracle."""
    K, nsteps = 3, 24
    grid = syn.make_grid(Nj, Ni, dkm=4.0, warp=0.0)
    u, v, sic = syn.make_fields(grid, K=K, seed=5, umax=0.5, drift=0.3)
    rng = np.random.default_rng(8)
    nP = 20_000
    if Ni > Nj:
        j = rng.uniform(3, Nj - 4, nP); i = rng.uniform(Ni - 3000, Ni - 4, nP)
    else:
        j = rng.uniform(Nj - 3000, Nj - 4, nP); i = rng.uniform(3, Ni - 4, nP)
    yx = np.stack([np.interp(j, np.arange(Nj), grid["Yt"][:, 0]), np.interp(i, np.arange(Ni), grid["Xt"][0, :])], axis=1)
    ji = syn.regular_host_cell(grid, yx).astype(np.int32)
    assert ji[:, 1].max() > 65000 or ji[:, 0].max() > 32500
    f64 = [(u[k].astype(np.float64), v[k].astype(np.float64), sic[k].astype(np.float64)) for k in range(K)]
    ref = orc.Tracker(grid, yx, ji, nthreads=8)
    for s in range(nsteps):
        ref.step(s, *f64[s % K], want_out=False)
    for fuse in (32, 1):
        ctx = sit.Context(0)
        try:
            ctx.set_grid(grid["Yf"], grid["Xf"], grid["Yu"], grid["Xu"], grid["Yv"], grid["Xv"], grid["tmask"])
            ctx.alloc_records(K, np.float32)
            for k in range(K):
                ctx.push_record(k, u[k], v[k], sic[k])
            ctx.set_tuning(fuse=fuse)
            ctx.set_buoys(yx, ji)
            ctx.set_resort(7)
            for s in range(0, nsteps, K):
                ctx.run(0, s, K)
            out = ctx.fetch()
        finally:
            ctx.close()
        assert np.array_equal(out["yx"], ref.pos) and np.array_equal(out["jiT"], ref.jiT) and np.array_equal(out["alive"], ref.alive)
    assert (out["jiT"] != ji).any()
