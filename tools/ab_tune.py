#!/usr/bin/env python3
"""Interleaved A/B of libsitrk performance knobs on one GPU, one process (C3 workload).

    python tools/ab_tune.py [--steps 200] [--rounds 5] [--knobs xcd_remap,nt_state] [--tiles 8x16,16x16]
                            [--fuse 2,4,8] [--values step_block:256:512:1024] [--base fuse=1,nt_state=1]
Knobs (sitrk_set_tuning): xcd_remap, nt_state (0/1), sort_tile (tile_j*256+tile_i), fuse (1..32), step_block
(256/512/1024), locate_bruteforce; with `make -C sitrack_amd/csrc -B DIAG=1` also the ablation kernels
diag_memonly / diag_nocross (--singles).  Prints median / min ms per record for every variant; results must not
depend on knobs (checked on the final state against the first variant).  This is how the defaults in
sitrk_internal.h were chosen."""
import argparse
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import sitrack_amd as sit                      # noqa: E402
from sitrack_amd import synthetic as syn       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=4096)
    ap.add_argument("--buoys", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--knobs", default="xcd_remap,nt_state")
    ap.add_argument("--tiles", default="", help="comma list of tile_jxtile_i sort orders to add as variants, e.g. 8x32,16x16")
    ap.add_argument("--fuse", default="", help="comma list of records-per-launch values to add as variants, e.g. 2,4,8")
    ap.add_argument("--values", default="", help="knob:v1:v2:... extra variants sweeping one knob's values, e.g. step_block:512:1024")
    ap.add_argument("--base", default="", help="knob=value,... applied to every variant")
    ap.add_argument("--singles", default="", help="extra variants, one knob each (e.g. diag_memonly,diag_nocross); no result check")
    a = ap.parse_args()
    N, nP, K = a.grid, a.buoys, 8
    grid = syn.make_grid(N, N, dkm=4.0, warp=0.0)
    _, yx = syn.make_buoys(grid, nP, seed=1234, frac=0.6)
    ji = syn.regular_host_cell(grid, yx).astype(np.int32)
    u, v, sic = syn.make_fields(grid, K=K, seed=2024, umax=0.3, drift=0.05)
    ctx = sit.Context(0)
    ctx.set_grid(grid["Yf"], grid["Xf"], grid["Yu"], grid["Xu"], grid["Yv"], grid["Xv"], grid["tmask"])
    ctx.alloc_records(K, np.float32)
    for k in range(K):
        ctx.push_record(k, u[k], v[k], sic[k])
    knobs = [k for k in a.knobs.split(",") if k]
    variants = [dict(zip(knobs, bits)) for bits in itertools.product((0, 1), repeat=len(knobs))]
    singles = [k for k in a.singles.split(",") if k]
    allk = knobs + singles
    variants = [dict({k: 0 for k in allk}, **v) for v in variants] + [dict({k: 0 for k in allk}, **{k: 1}) for k in singles]
    base = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.base.split(",") if kv)
    for t in [t for t in a.tiles.split(",") if t]:
        tj, ti = (int(x) for x in t.split("x"))
        variants.append(dict({k: 0 for k in allk}, sort_tile=tj * 256 + ti))
    for fz in [int(x) for x in a.fuse.split(",") if x]:
        variants.append(dict({k: 0 for k in allk}, sort_tile=8 * 256 + 16, fuse=fz))
    if a.values:
        kn, *vals = a.values.split(":")
        for vv in vals:
            variants.append({kn: int(vv)})
    variants = [dict(dict(fuse=1), **dict(v, **base)) for v in variants]
    times = {i: [] for i in range(len(variants))}
    ref = None
    for rnd in range(a.rounds):
        for i, var in enumerate(variants):
            ctx.set_tuning(**var)
            ctx.set_buoys(yx, ji)                  # same start every time (sorted with the variant's key)
            ctx.run(0, 0, 10)
            ctx.sync()
            ctx.timer_start()
            ctx.run(10 % K, 10, a.steps)
            ms = ctx.timer_stop()
            times[i].append(ms / a.steps)
            if rnd == 0 and not any(var.get(k) for k in singles):
                st = ctx.fetch(("yx", "jiT"))
                if ref is None:
                    ref = st
                else:
                    assert np.array_equal(st["yx"], ref["yx"]) and np.array_equal(st["jiT"], ref["jiT"]), var
    for i, var in enumerate(variants):
        t = np.array(times[i])
        print("%-40s median %.4f ms  min %.4f ms  (%.3e p-steps/s)" % (var, np.median(t), t.min(), nP / np.median(t) * 1e3))


if __name__ == "__main__":
    main()
