#!/usr/bin/env python3
"""Whole pipeline at scale on one GPU, in memory (no NetCDF): the reference driver's stages with synthetic inputs of
BASELINE.json config-5 shape -- a polar 4096^2 mesh (1 km), 1e7 seeds given as (lat,lon) + (y,x) like a seeding file,
ice mask with open water and land, hourly records.

    python tests/sweeps/demo_pipeline.py [--grid 4096] [--seeds 10000000] [--records 240]

Stages timed: grid upload, SeedInit (GPU locate), compaction, set_buoys (+ cell sort), record upload, tracking
(sitrk_run, 8 records per launch), final fetch with lat/lon.  A subsample goes through the CPU oracle end to
end (SeedInit + every record) and must agree bit for bit.  (Subsample = 256 seeds: the oracle's SeedInit is the reference's
O(Nj*Ni) scan per seed.)"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import sitrack_amd as sit                      # noqa: E402
from sitrack_amd import synthetic as syn       # noqa: E402
from oracle import oracle as orc               # noqa: E402  (checker only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=4096)
    ap.add_argument("--seeds", type=int, default=10_000_000)
    ap.add_argument("--records", type=int, default=240)
    a = ap.parse_args()
    N, K = a.grid, 8
    T = {}
    t = time.perf_counter()
    grid = syn.make_grid(N, N, dkm=1.0, warp=1.0)
    tmask = grid["tmask"].copy()
    tmask[N // 3:N // 3 + N // 16, N // 2:N // 2 + N // 8] = 0                   # an island
    u, v, sic = syn.make_fields(grid, K=K, seed=2024, umax=0.3, drift=0.05)
    sic[:, : N // 5, :] = 0.05                                                   # open water in the south of the mesh
    T["synthetic inputs (host)"] = time.perf_counter() - t

    t = time.perf_counter()
    ctx = sit.Context(0)
    ctx.set_grid(grid["Yf"], grid["Xf"], grid["Yu"], grid["Xu"], grid["Yv"], grid["Xv"], tmask)
    llT = ctx.cart2geo(np.stack([grid["Yt"].ravel(), grid["Xt"].ravel()], axis=1))
    latT = np.ascontiguousarray(llT[:, 0].reshape(N, N)); lonT = np.ascontiguousarray(np.mod(llT[:, 1], 360.).reshape(N, N))
    T["grid upload + T-point lat/lon"] = time.perf_counter() - t

    _, yx = syn.make_buoys(grid, a.seeds, seed=1234, frac=0.9)
    ll = ctx.cart2geo(yx); ll[:, 1] = np.mod(ll[:, 1], 360.)
    # a seeding file stores f4 and the driver promotes to f8 (reference ncio.py:294-309)
    pSG = ll.astype(np.float32).astype(np.float64); pSC = yx.astype(np.float32).astype(np.float64)
    ids = np.arange(1, a.seeds + 1, dtype=np.int64)
    sic0 = sic[0].astype(np.float64)

    t = time.perf_counter()
    nP, oSG, oSC, oIDs, vJIt, VRTCS, idxK = sit.SeedInit(ids, pSG, pSC, latT, lonT, grid["Yf"], grid["Xf"], grid["resol"], tmask,
                                                        xIceConc=sic0, ctx=ctx)
    T["SeedInit (%d seeds -> %d kept)" % (a.seeds, nP)] = time.perf_counter() - t

    t = time.perf_counter()
    ctx.set_grid(grid["Yf"], grid["Xf"], grid["Yu"], grid["Xu"], grid["Yv"], grid["Xv"], tmask)
    ctx.alloc_records(K, np.float32)
    for k in range(K):
        ctx.push_record(k, u[k], v[k], sic[k])
    T["records upload (%d x 201 MB)" % K] = time.perf_counter() - t
    t = time.perf_counter()
    ctx.set_buoys(oSC, vJIt)
    ctx.sync()
    T["set_buoys + cell sort"] = time.perf_counter() - t
    t = time.perf_counter()
    ctx.run(0, 0, a.records)
    ctx.sync()
    T["tracking %d records" % a.records] = time.perf_counter() - t
    rate = nP * a.records / T["tracking %d records" % a.records]
    t = time.perf_counter()
    st = ctx.fetch()
    pos, msk, latlon = ctx.fetch_record(a.records - 1, latlon=True)
    T["fetch state + last record with lat/lon"] = time.perf_counter() - t

    for k, val in T.items():
        print("  %-48s %8.3f s" % (k, val), file=sys.stderr, flush=True)
    # ---- oracle on a subsample, end to end (its SeedInit scans the whole mesh per seed, like the reference)
    sub = np.arange(0, a.seeds, max(1, a.seeds // 256))[:256]
    print("  oracle: SeedInit of %d seeds by exhaustive scan ..." % len(sub), file=sys.stderr, flush=True)
    o = orc.SeedInit(ids[sub], pSG[sub], pSC[sub], latT, lonT, grid["Yf"], grid["Xf"], grid["resol"], tmask, sic0, nthreads=16)
    print("  oracle: tracking ...", file=sys.stderr, flush=True)
    kept_sub = np.isin(sub, idxK)
    assert o[0] == int(kept_sub.sum()) and np.array_equal(o[3], ids[sub][kept_sub])
    where = np.searchsorted(idxK, sub[kept_sub])
    assert np.array_equal(o[4], vJIt[where])
    g2 = dict(grid); g2["tmask"] = tmask
    ref = orc.Tracker(g2, o[2], o[4], nthreads=8)
    f64 = [(u[k].astype(np.float64), v[k].astype(np.float64), sic[k].astype(np.float64)) for k in range(K)]
    for s in range(a.records):
        ref.step(s, *f64[s % K], want_out=False)
    ok = (np.array_equal(st["yx"][where], ref.pos) and np.array_equal(st["jiT"][where], ref.jiT)
          and np.array_equal(st["alive"][where], ref.alive))
    want_ll = orc.CartNPSkm2Geo1D(pos[where])
    ok_ll = bool(np.allclose(latlon[where], want_ll, rtol=1e-12, atol=1e-10))
    print(json.dumps({"grid": [N, N], "seeds": a.seeds, "kept": int(nP), "records": a.records,
                      "alive_after": int(st["alive"].sum()), "particle_steps_per_s_tracking": rate,
                      "seconds": {k: round(v, 4) for k, v in T.items()},
                      "oracle_subsample": {"seeds": len(sub), "kept": int(o[0]), "bit_exact": bool(ok), "latlon_1e-10": ok_ll}}))
    assert ok and ok_ll
    ctx.close()


if __name__ == "__main__":
    main()
