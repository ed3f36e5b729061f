#!/usr/bin/env python3
"""Why is the first fused launch after a pause slower ("What a 20-step run measures")?  Times 20-record launches of the C3
workload: back to back, after host-side pauses of different lengths, on record slots that no kernel has read since their upload and
on slots read before (GPU box).  Prints one JSON line per series."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sitrack_amd as sit                                    # noqa: E402
from sitrack_amd import synthetic as syn                     # noqa: E402

N, nP, K = 4096, 10_000_000, 64
g = syn.make_grid(N, N, dkm=4.0, warp=0.0)
_, yx = syn.make_buoys(g, nP, seed=1234, frac=0.6)
ji = syn.regular_host_cell(g, yx).astype(np.int32)
u, v, sic = syn.make_fields(g, K=8, seed=2024, umax=0.3, drift=0.05)
ctx = sit.Context(0)
ctx.set_grid(g["Yf"], g["Xf"], g["Yu"], g["Xu"], g["Yv"], g["Xv"], g["tmask"])
ctx.alloc_records(K, np.float32)
for k in range(K):
    ctx.push_record(k, u[k % 8], v[k % 8], sic[k % 8])
ctx.set_buoys(yx, ji)
ctx.set_tuning(fuse=20)
ctx.sync()
step = [0]


def launch(slot0):
    ctx.timer_start()
    ctx.run(slot0, step[0], 20)
    ms = ctx.timer_stop()
    step[0] += 20
    return ms


out = {}
# (a) the very first launch reads slots 0..19 for the first time; then slots 20..39 (first touch again), then 0..19 again (touched before)
out["first_ever_slots_0_19"] = launch(0)
out["next_untouched_slots_20_39"] = launch(20)
out["again_slots_0_19"] = launch(0)
out["again_slots_20_39"] = launch(20)
out["untouched_slots_40_59"] = launch(40)
# (b) back to back on touched slots
out["back_to_back_x8"] = [launch((20 * i) % 40) for i in range(8)]
# (c) after pauses
for pause in (0.01, 0.1, 1.0, 3.0):
    time.sleep(pause)
    a = launch(0)
    b = launch(20)
    out["after_%.2fs_pause" % pause] = [a, b]
print(json.dumps(out))
ctx.close()
