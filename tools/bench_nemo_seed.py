#!/usr/bin/env python3
"""Times `sitrk_nemo_seed` (idealised seeding on the device: mask logic, ordered compaction, projection) at the scale of
BASELINE config 5: a 4096 x 4096 polar mesh under a synthetic ice mask, T- and F-seeds -> ~1e7 seeds.  Prints one JSON line;
a 64 x 64 corner of the same inputs is checked against the numpy statement of the reference's rule."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sitrack_amd as sit                      # noqa: E402
from sitrack_amd import synthetic as syn       # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    g = syn.make_grid(N, N, dkm=1.5, warp=0.5)
    ctx = sit.Context(0)
    llT = ctx.cart2geo(np.stack([g["Yt"].ravel(), g["Xt"].ravel()], axis=1))
    llF = ctx.cart2geo(np.stack([g["Yf"].ravel(), g["Xf"].ravel()], axis=1))
    latT, lonT = llT[:, 0].reshape(N, N).copy(), np.mod(llT[:, 1], 360.).reshape(N, N).copy()
    latF, lonF = llF[:, 0].reshape(N, N).copy(), np.mod(llF[:, 1], 360.).reshape(N, N).copy()
    rng = np.random.default_rng(5)
    tmask = g["tmask"].copy()
    sic = np.ones((N, N))
    for _ in range(40):                                   # islands and open water
        j, i, h, w = rng.integers(0, N - 300), rng.integers(0, N - 300), rng.integers(20, 300), rng.integers(20, 300)
        (tmask if rng.random() < 0.5 else sic)[j:j + h, i:i + w] = 0
    ctx.nemo_seed(tmask[:64, :64], latT[:64, :64], lonT[:64, :64], sic[:64, :64])          # warm up
    out = {}
    for tag, kw in (("T_only", {}), ("T_and_F", dict(latF=latF, lonF=lonF)), ("T_every_2nd", dict(khss=2))):
        t0 = time.perf_counter()
        ll, yx, nT, nF = ctx.nemo_seed(tmask, latT, lonT, sic, **kw)
        dt = time.perf_counter() - t0
        out[tag] = {"seeds": int(nT + nF), "T": int(nT), "F": int(nF), "wall_s": round(dt, 4), "seeds_per_s": (nT + nF) / dt}
    # check a corner against numpy
    n = 64
    m = tmask[:n, :n].copy(); m[latT[:n, :n] < 55.] = 0; m[sic[:n, :n] < 0.9] = 0
    ll, _, nT, nF = ctx.nemo_seed(tmask[:n, :n], latT[:n, :n], lonT[:n, :n], sic[:n, :n])
    assert nT == int((m == 1).sum()) and np.array_equal(ll[:, 0], latT[:n, :n][m == 1])
    print(json.dumps({"case": "%dx%d mesh incl. host<->device copies of the mesh arrays and of the seeds" % (N, N), **out}))
    ctx.close()


if __name__ == "__main__":
    main()
