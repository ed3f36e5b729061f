#!/usr/bin/env python3
"""Time SeedInit (nearest T-point by Haversine + Survive + FindContainingCell) at C3 scale on one GPU.

    python tools/bench_locate.py [--grid 4096] [--seeds 10000000]
The reference does O(nP*Nj*Ni) Haversine evaluations here (1.7e14 at this size); the library's bounding-sphere
search visits a few hundred mesh points per seed.  Checks the result against the analytic host cell of the regular
grid and, on a subsample, against the exhaustive scan (`locate_bruteforce`)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import sitrack_amd as sit                      # noqa: E402
from sitrack_amd import synthetic as syn       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=4096)
    ap.add_argument("--seeds", type=int, default=10_000_000)
    ap.add_argument("--dkm", type=float, default=1.0, help="grid spacing [km] (4096 x 1 km keeps the mesh north of ~55N)")
    a = ap.parse_args()
    N = a.grid
    grid = syn.make_grid(N, N, dkm=a.dkm, warp=0.0)
    ctx = sit.Context(0)
    ctx.set_grid(grid["Yf"], grid["Xf"], grid["Yf"], grid["Xf"], grid["Yf"], grid["Xf"], grid["tmask"])
    llT = ctx.cart2geo(np.stack([grid["Yt"].ravel(), grid["Xt"].ravel()], axis=1))
    latT = np.ascontiguousarray(llT[:, 0].reshape(N, N))
    lonT = np.ascontiguousarray(np.mod(llT[:, 1], 360.).reshape(N, N))
    _, yx = syn.make_buoys(grid, a.seeds, seed=1234, frac=0.6)
    ll = ctx.cart2geo(yx)
    ll[:, 1] = np.mod(ll[:, 1], 360.)
    sic = np.ones((N, N))
    ctx.seed_init(ll[:1000], yx[:1000], latT, lonT, grid["resol"], sic)            # warm-up (allocations)
    t0 = time.perf_counter()
    ji, keep, why = ctx.seed_init(ll, yx, latT, lonT, grid["resol"], sic)
    dt = time.perf_counter() - t0
    want = syn.regular_host_cell(grid, yx)
    ok = bool(keep.all()) and bool(np.array_equal(ji, want))
    # exhaustive scan on a subsample
    sub = np.arange(0, a.seeds, max(1, a.seeds // 2000))[:2000]
    ctx.set_tuning(locate_bruteforce=1)
    t1 = time.perf_counter()
    jb, kb, wb = ctx.seed_init(ll[sub], yx[sub], latT, lonT, grid["resol"], sic)
    dtb = time.perf_counter() - t1
    same = bool(np.array_equal(jb, ji[sub]) and np.array_equal(kb, keep[sub]))
    print(json.dumps({"what": "SeedInit", "grid": [N, N], "seeds": a.seeds, "seconds": dt, "seeds_per_s": a.seeds / dt,
                      "includes": "H2D of seeds and of latT/lonT/resol/sic, sphere set-up, search, D2H",
                      "all_found_in_expected_cell": ok,
                      "bruteforce_subsample": {"seeds": len(sub), "seconds": dtb, "seeds_per_s": len(sub) / dtb, "identical": same},
                      "reference_haversines_avoided": float(a.seeds) * N * N}))
    ctx.close()


if __name__ == "__main__":
    main()
