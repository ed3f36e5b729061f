#!/usr/bin/env python3
"""Randomised sweep of SeedInit: the bounding-sphere search must return exactly what the exhaustive Haversine scan
returns (same device formula, ties -> lowest flat index) on random meshes anywhere on the northern hemisphere -- across
the pole, down to low latitudes, strongly warped -- with seeds on T-points, on F-points (exact ties), inside cells and
far outside the mesh; a subsample is also held against the CPU oracle (glibc libm).

    python tests/sweeps/fuzz_locate.py [--cases 80] [--seed 0]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import sitrack_amd as sit                      # noqa: E402
from sitrack_amd import synthetic as syn       # noqa: E402
from oracle import oracle as orc               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=80)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    ctx = sit.Context(0)
    t0 = time.time()
    n_seeds = n_oracle = n_diff_oracle = 0
    for idx in range(a.cases):
        Nj, Ni = int(rng.integers(20, 200)), int(rng.integers(20, 230))
        dkm = float(rng.choice([1.0, 4.0, 12.5, 50.0]))
        warp = float(rng.choice([0.0, 1.0, 2.0]))
        yc, xc = float(rng.uniform(-4500, 4500)), float(rng.uniform(-4500, 4500))
        g = syn.make_grid(Nj, Ni, dkm=dkm, warp=warp)
        for k in ("Yt", "Yf"):
            g[k] = g[k] + yc
        for k in ("Xt", "Xf"):
            g[k] = g[k] + xc
        llT = orc.CartNPSkm2Geo1D(np.stack([g["Yt"].ravel(), g["Xt"].ravel()], axis=1))
        latT = np.ascontiguousarray(llT[:, 0].reshape(Nj, Ni)); lonT = np.ascontiguousarray(np.mod(llT[:, 1], 360.).reshape(Nj, Ni))
        n_r = 1500
        ext = 8 * dkm
        yx = np.stack([rng.uniform(g["Yt"].min() - ext, g["Yt"].max() + ext, n_r), rng.uniform(g["Xt"].min() - ext, g["Xt"].max() + ext, n_r)], axis=1)
        jj, ii = rng.integers(0, Nj, 300), rng.integers(0, Ni, 300)
        onT = np.stack([g["Yt"][jj, ii], g["Xt"][jj, ii]], axis=1)
        onF = np.stack([g["Yf"][jj, ii], g["Xf"][jj, ii]], axis=1)
        far = np.array([[g["Yt"].min() - 200 * dkm, xc], [yc, g["Xt"].max() + 500 * dkm]])
        yx = np.concatenate([yx, onT, onF, far])
        ll = orc.CartNPSkm2Geo1D(yx); ll[:, 1] = np.mod(ll[:, 1], 360.)
        ll[n_r:n_r + 300, 0] = latT[jj, ii]; ll[n_r:n_r + 300, 1] = lonT[jj, ii]
        if rng.random() < 0.5:                        # seeding files store float32
            ll = ll.astype(np.float32).astype(np.float64); yx = yx.astype(np.float32).astype(np.float64)
        tmask = (rng.random((Nj, Ni)) > 0.03).astype(np.int8)
        sic = rng.choice([0.0, 0.2, 1.0], size=(Nj, Ni), p=[0.05, 0.05, 0.9])
        resol = np.full((Nj, Ni), np.sqrt(2.) * dkm)
        ctx.set_grid(g["Yf"], g["Xf"], g["Yf"], g["Xf"], g["Yf"], g["Xf"], tmask)
        res = {}
        for mode in (0, 1):
            ctx.set_tuning(locate_bruteforce=mode)
            res[mode] = ctx.seed_init(ll, yx, latT, lonT, resol, sic)
        for x, y in zip(res[0], res[1]):
            assert np.array_equal(x, y), ("sphere search != exhaustive scan", idx)
        sel = rng.choice(len(yx), 40, replace=False)
        o = orc.SeedInit(np.arange(40), ll[sel], yx[sel], latT, lonT, g["Yf"], g["Xf"], resol, tmask, sic, return_why=True, nthreads=8)
        keep_o = np.zeros(40, dtype=np.int8); keep_o[o[6]] = 1
        same = np.array_equal(keep_o, res[0][1][sel]) and np.array_equal(o[4], res[0][0][sel][keep_o == 1])
        n_diff_oracle += 0 if same else 1
        n_seeds += len(yx); n_oracle += 40
        print("case %3d: mesh %dx%d dkm %.1f warp %.0f centre (%.0f,%.0f) km lat %.1f..%.1f: %d seeds, %d kept; oracle subsample %s" % (
            idx, Nj, Ni, dkm, warp, yc, xc, latT.min(), latT.max(), len(yx), int(res[0][1].sum()), "same" if same else "DIFFERS (libm tie)"),
            flush=True)
    print("ALL %d CASES: sphere search == exhaustive scan on %d seeds; oracle (glibc) agreed on %d/%d subsamples (%.1f s)" % (
        a.cases, n_seeds, a.cases - n_diff_oracle, a.cases, time.time() - t0), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
