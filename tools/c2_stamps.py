#!/usr/bin/env python3
"""GPU box, DIAGNOSTIC library (tools/build_variant.sh diag -DSITRK_DIAG; SITRK_LIB_PATH=build_ab/libsitrk_diag.so): where one
wave's time goes inside a record of the fused loop -- s_memtime stamps accumulated per wave over one 32-record launch.

    SITRK_LIB_PATH=$PWD/build_ab/libsitrk_diag.so python tools/c2_stamps.py [--config c2|c3] [--buoys N]
"""
import argparse
import ctypes as C
import json
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sitrack_amd as sit                           # noqa: E402
from sitrack_amd import synthetic as syn, _lib      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2")
ap.add_argument("--buoys", type=int, default=0)
a = ap.parse_args()
Nj, Ni, nP = {"c2": (512, 512, 100_000), "c3": (4096, 4096, 10_000_000)}[a.config]
nP = a.buoys or nP
K = 32
grid = syn.make_grid(Nj, Ni, dkm=4.0, warp=0.0)
_, yx = syn.make_buoys(grid, nP, seed=1234, frac=0.6)
ji = syn.regular_host_cell(grid, yx).astype(np.int32)
u, v, sic = syn.make_fields(grid, K=K, seed=2024, umax=0.3, drift=0.05)
ctx = sit.Context(0)
ctx.set_grid(grid["Yf"], grid["Xf"], grid["Yu"], grid["Xu"], grid["Yv"], grid["Xv"], grid["tmask"])
ctx.set_params(3600., 1, 0.1)
ctx.alloc_records(K, np.float32)
for k in range(K):
    ctx.push_record(k, u[k], v[k], sic[k])
ctx.set_buoys(yx, ji)
ctx.set_tuning(fuse=32)
ctx.run(0, 0, 64)                                   # warm
ctx.sync()
ctx.set_tuning(stamps=1)
ctx.timer_start()
ctx.run(0, 64, 32)                                  # ONE stamped launch of 32 records
ms = ctx.timer_stop()
L = _lib.lib()
f = L.sitrk_diag_stamps
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.POINTER(C.c_longlong)]
nw = (nP + 63) // 64
out = np.zeros((nw + 8, 8), dtype=np.uint64)
n = C.c_longlong(0)
assert f(ctx._h, out.ctypes.data_as(C.c_void_p), nw + 8, C.byref(n)) == 0
st = out[:nw].astype(np.float64) / 32.0             # cycles per record
names = ["context there (wait for the U/V points + orientation byte requested by the previous record's crossing path; issues the record's loads)",
         "velocity pick: orientation tests", "velocities there (wait)", "Euler update", "cell test", "crossing path (resolution, requests the new context)"]
med = np.median(st, axis=0)
res = {"config": a.config, "buoys": nP, "waves": int(nw), "launch_ms_with_stamps": ms, "cycles_per_record_median_over_waves": {}}
tot = 0.0
for k, nm in enumerate(names):
    res["cycles_per_record_median_over_waves"]["%d %s" % (k, nm)] = float(med[k])
    tot += float(med[k])
res["sum_cycles_per_record"] = tot
res["us_per_record_if_2.1GHz"] = tot / 2100.0
print(json.dumps(res, indent=1))
