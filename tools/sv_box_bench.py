#!/usr/bin/env python3
"""Time the per-record Survive derivation on the C3 mesh: whole record vs the box the C3 buoys can touch, one launch per record and one launch
per batch of 16 records (GPU box; prints one JSON line per case).  tools/sv_box_bench.py [--n 200]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sitrack_amd as sit                                    # noqa: E402
from sitrack_amd import synthetic as syn                     # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=320)
ap.add_argument("--grid", type=int, default=4096)
a = ap.parse_args()
N = a.grid
g = syn.make_grid(N, N, dkm=4.0, warp=0.0)
u, v, sic = syn.make_fields(g, K=2, seed=2024, umax=0.3, drift=0.05)
ctx = sit.Context(0)
ctx.set_grid(g["Yf"], g["Xf"], g["Yu"], g["Xu"], g["Yv"], g["Xv"], g["tmask"])
NS = 16
ctx.alloc_records(NS, np.float32)
for k in range(NS):
    ctx.push_record(k, u[k % 2], v[k % 2], sic[k % 2])
lo, hi = int(0.2 * N) - 2, int(0.8 * N) + 3
box = (lo, hi, lo - lo % 4, -(-hi // 4) * 4)
for name, fn in (("whole", lambda k: ctx.commit_record(k)), ("box", lambda k: ctx.commit_record_box(k, *box)),
                 ("box_batch16", lambda k: (ctx.commit_records_box(0, NS, *box) if k % NS == 0 else None))):
    for smw in (0,):
        for tile in (0, 1):
            ctx.set_tuning(survive_tile=tile)
            for k in range(10):
                fn(k % NS)
            ctx.sync()
            ctx.timer_start()
            for k in range(a.n):
                fn(k % NS)
            ms = ctx.timer_stop()
            cells = N * N if name == "whole" else (box[1] - box[0]) * (box[3] - box[2])
            print(json.dumps({"case": name, "survive_tile": tile, "us_per_record": 1e3 * ms / a.n,
                              "cells": cells, "GBps_7B_per_cell": 7.0 * cells / (ms / a.n * 1e-3) / 1e9}), flush=True)
ctx.close()
