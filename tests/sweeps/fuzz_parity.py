#!/usr/bin/env python3
"""Randomised parity sweep: GPU (through the C ABI) vs the CPU oracle on many random configurations --
grid size and warp, velocity scale, land/ice patterns incl. NaN/huge fill values, per-buoy windows, both
velocity rules, fp32/fp64 records, fused and per-record launches, re-sort cadence, tile order.  Every case must
be bit-identical (positions, cells, alive, kill record, per-record masks on sampled records).

    python tests/sweeps/fuzz_parity.py [--cases 60] [--seed 0]"""
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


def one_case(rng, idx, nstrat=2):
    Nj, Ni = int(rng.integers(24, 260)), int(rng.integers(24, 300))
    warp = float(rng.choice([0.0, 0.5, 1.0]))
    dkm = float(rng.choice([1.0, 4.0, 12.5]))
    K = int(rng.integers(1, 9)) if rng.random() < 0.8 else int(rng.integers(9, 41))      # up to 40 resident records
    nP = int(rng.integers(1, 40000))
    Nt = int(rng.integers(1, 60))
    strat = int(rng.integers(0, nstrat))       # 0/1 = the reference's rules; 2 = the linear-interpolation extra
    fdt = np.float64 if rng.random() < 0.25 else np.float32
    umax = float(rng.choice([0.1, 0.3, 0.9, 2.5])) * dkm / 4.0
    grid = syn.make_grid(Nj, Ni, dkm=dkm, warp=warp)
    u, v, sic = syn.make_fields(grid, K=K, seed=int(rng.integers(1 << 30)), umax=umax, drift=0.3 * umax, ripple=0.2 * umax, dtype=fdt)
    tmask = grid["tmask"].copy()
    for _ in range(int(rng.integers(0, 4))):
        j, i = int(rng.integers(2, Nj - 6)), int(rng.integers(2, Ni - 6))
        tmask[j:j + int(rng.integers(1, 5)), i:i + int(rng.integers(1, 8))] = 0
    for _ in range(int(rng.integers(0, 3))):
        j, i = int(rng.integers(2, Nj - 8)), int(rng.integers(2, Ni - 8))
        sic[:, j:j + int(rng.integers(1, 8)), i:i + int(rng.integers(1, 10))] = float(rng.choice([0.0, 0.05, 0.0999, 0.1]))
    if rng.random() < 0.3:
        j, i = int(rng.integers(2, Nj - 4)), int(rng.integers(2, Ni - 4))
        u[:, j:j + 2, i:i + 3] = float(rng.choice([np.nan, 1e20, -1e20, np.inf]))
    _, yx = syn.make_buoys(grid, nP, seed=int(rng.integers(1 << 30)), frac=float(rng.uniform(0.3, 0.95)))
    trk = sit.IceTracker(grid["Yf"], grid["Xf"], grid["Yu"], grid["Xu"], grid["Yv"], grid["Xv"], tmask, iUVstrategy=strat,
                         nslots=K, field_dtype=fdt)
    found, ji, _ = sit.FindContainingCell(yx, syn.nearest_t_plane(grid, yx), ctx=trk.ctx)
    yx, ji = yx[found], ji[found]
    n = len(yx)
    if n == 0:
        trk.close()
        return "empty"
    windowed = rng.random() < 0.4
    first = last = None
    if windowed:
        first = rng.integers(0, max(1, Nt // 2), n); last = first + rng.integers(-2, Nt, n)
    fuse = int(rng.choice([1, 2, 8, 32]))
    tile = int(rng.choice([0, 8 * 256 + 16, 4 * 256 + 4, 32 * 256 + 32]))
    trk.ctx.set_tuning(fuse=fuse, sort_tile=tile, nt_state=int(rng.integers(0, 2)), xcd_remap=int(rng.integers(0, 2)),
                       step_block=int(rng.choice([256, 512, 1024])),
                       patch_kb=int(rng.choice([0, 2, 16, 40])), patch_margin=int(rng.choice([0, 2, 8])),
                       xcd_group=int(rng.choice([0, 3, 16])), survive_tile=int(rng.integers(0, 2)))
    trk.set_buoys(yx, ji, first, last, sort=bool(rng.random() < 0.8))
    trk.ctx.set_resort(int(rng.choice([0, 3, 17])))
    g2 = dict(grid); g2["tmask"] = tmask
    ref = orc.Tracker(g2, yx, ji, rec_first=first, rec_last=last, uv_strategy=strat, nthreads=8)
    for k in range(K):
        trk.load_record(k, u[k], v[k], sic[k])
    j0 = int(rng.integers(0, 5))
    boxed = rng.random() < 0.35
    if boxed:
        # round 4: records arrive as BOXES (rows x columns the buoys can touch), everything else of the slot poisoned; a third of the
        # boxed cases commit the box of a slab that sits in device memory whole (commit_records_box, one Survive launch per batch)
        poison = (np.full((Nj, Ni), np.nan, dtype=fdt), np.full((Nj, Ni), np.nan, dtype=fdt), np.zeros((Nj, Ni), dtype=fdt))
        commit = rng.random() < 0.33
        m = max(1, min(fuse, K))
        s = 0
        while s < Nt:
            cnt = min(m, Nt - s)
            box = trk.ctx.box(cnt - 1)
            for r in range(cnt):
                k = (j0 + s + r) % K
                if commit:
                    comp = [p_.copy() for p_ in poison]
                    for dst, src in zip(comp, (u[k], v[k], sic[k])):
                        dst[box[0]:box[1], box[2]:box[3]] = src[box[0]:box[1], box[2]:box[3]]
                    trk.ctx.push_record(k, *comp)
                else:
                    trk.ctx.push_record(k, *poison)
                    if box[1] > box[0]:
                        trk.ctx.push_record_box(k, *box, u[k][box[0]:box[1], box[2]:box[3]], v[k][box[0]:box[1], box[2]:box[3]],
                                                sic[k][box[0]:box[1], box[2]:box[3]])
            if commit:
                trk.ctx.commit_records_box((j0 + s) % K, cnt, *box)
            trk.ctx.run((j0 + s) % K, j0 + s, cnt)
            for r in range(cnt):
                ref.step(j0 + s + r, u[(j0 + s + r) % K], v[(j0 + s + r) % K], sic[(j0 + s + r) % K], want_out=False)
            s += cnt
    elif fuse == 1 and rng.random() < 0.5:
        for s in range(Nt):
            trk.step(j0 + s, (j0 + s) % K)
            rp, rm = ref.step(j0 + s, u[(j0 + s) % K], v[(j0 + s) % K], sic[(j0 + s) % K])
            if s % 7 == 0 or s == Nt - 1:
                pn, mn = trk.record(j0 + s)
                assert np.array_equal(mn, rm) and np.array_equal(pn, rp, equal_nan=True), ("record", idx, s)
    else:
        trk.ctx.run(j0 % K, j0, Nt)
        for s in range(Nt):
            ref.step(j0 + s, u[(j0 + s) % K], v[(j0 + s) % K], sic[(j0 + s) % K], want_out=False)
    st = trk.state()
    assert np.array_equal(st["yx"], ref.pos, equal_nan=True), ("pos", idx)
    assert np.array_equal(st["vJIt"], ref.jiT), ("cell", idx)
    assert np.array_equal(st["iAlive"], ref.alive), ("alive", idx)
    assert np.array_equal(st["kill_rec"] >= 0, ref.alive == 0), ("kill_rec", idx)
    desc = "grid %dx%d warp %.1f dkm %.1f nP %d Nt %d K %d strat %d %s win %d fuse %d tile %d box %d: crossings %d dead %d" % (
        Nj, Ni, warp, dkm, n, Nt, K, strat, np.dtype(fdt).name, windowed, fuse, tile, boxed, ref.ncross, int((ref.alive == 0).sum()))
    trk.close()
    return desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--with-extra", action="store_true", help="also draw uv_strategy 2 (not in the reference)")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    for idx in range(a.cases):
        print("case %3d: %s" % (idx, one_case(rng, idx, 3 if a.with_extra else 2)), flush=True)
    print("ALL %d CASES BIT-IDENTICAL (%.1f s)" % (a.cases, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
