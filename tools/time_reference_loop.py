#!/usr/bin/env python3
"""Times the REFERENCE's own Python functions in the loop shape of si3_part_tracker.py:378-490 on inputs cut from the C2 and
C3 synthetic workloads (SURVEY.md 8d "CPU baseline (1)"): 10^3 buoys x 100 records, one core.

    python tools/time_reference_loop.py [--buoys 1000] [--records 100] [--configs c2,c3]

Runs in the BUILD CONTAINER ONLY: it imports /root/reference/sitrack/{util,locate,tracking}.py through
tests/golden/refload.py (the package itself needs netCDF4) and drives them with tests/golden/gen_golden.py::reference_loop,
the restatement of the loop body that also generates golden set G6 (the loop lives under `__main__` in the reference and
cannot be imported).  Nothing of the reference is copied; only the measured numbers leave the container (BASELINE.md
section 2, bench.py `cpu_baseline.reference_python`).  The GPU box has no /root/reference: bench.py never calls this.

The trajectories of such a cut are no longer thrown away: `python tests/golden/gen_golden.py g6cd` runs the same loop on the same
cuts (with the workload's 32 resident records) and commits final states + per-record digests as tests/golden/g6c_c2cut.npz and
g6d_c3cut.npz, which the oracle (CPU) and the device (step / fused / fused + windows) are held against (round 4).

Same inputs as bench.py: regular 4-km grid, buoys default_rng(1234) in the central 60 %, fields default_rng(2024) with
umax 0.3 m/s, fp32 records promoted into fp64 arrays once per record like the reference does at :372-374 (three whole-grid
copies per record: at 4096^2 that copy, not the per-buoy work, dominates a 10^3-buoy sample -- reported both ways).
"""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--buoys", type=int, default=1000)
    ap.add_argument("--records", type=int, default=100)
    ap.add_argument("--configs", default="c2,c3")
    a = ap.parse_args()
    import gen_golden as gg                                  # imports the reference (refload) at module scope
    from sitrack_amd import synthetic as syn
    from sitrack_amd.tracking import vertices_of          # VRTCS from vJIt (locate.py:320-321); no GPU needed
    shapes = {"c2": (512, 512, 100_000), "c3": (4096, 4096, 10_000_000)}
    out = {"cpu": cpu_model(), "cores_used": 1, "python": platform.python_version(), "numpy": np.__version__,
           "date": time.strftime("%Y-%m-%d"), "buoys": a.buoys, "records": a.records,
           "how": "reference functions (intersect2Seg, IsInsideQuadrangle, CrossedEdge, NewHostCell, UpdtInd4NewCell, Survive) "
                  "in the loop order of si3_part_tracker.py:378-490, iUVstrategy = 1, tests/golden/gen_golden.py::reference_loop"}
    for cfg in a.configs.split(","):
        Nj, Ni, nP = shapes[cfg]
        K = 4
        grid = syn.make_grid(Nj, Ni, dkm=4.0, warp=0.0)
        _, yx = syn.make_buoys(grid, nP, seed=1234, frac=0.6)
        yx = yx[:a.buoys]
        ji = syn.regular_host_cell(grid, yx).astype(np.int64)
        vert = vertices_of(ji)
        u, v, sic = syn.make_fields(grid, K=K, seed=2024, umax=0.3, drift=0.05)
        first = np.zeros(a.buoys, dtype=np.int64); last = np.full(a.buoys, 10**9, dtype=np.int64)
        t0 = time.perf_counter()
        pos, msk, jit_rec, alive_rec, _, codes = gg.reference_loop(grid, grid["tmask"], u, v, sic, yx, ji, vert, first, last, 0,
                                                                   a.records, 3600., 1)
        dt = time.perf_counter() - t0
        steps = int(msk[1:].sum())
        # the three whole-grid record assignments of :372-374 alone (what reference_loop does once per record)
        xa = np.zeros_like(grid["Yf"])
        t1 = time.perf_counter()
        for jt in range(a.records):
            xa[:, :] = sic[jt % K]; xa[:, :] = u[jt % K]; xa[:, :] = v[jt % K]
        dt_copy = time.perf_counter() - t1
        out[cfg] = {"grid": [Nj, Ni], "particle_steps": steps, "seconds": dt, "particle_steps_per_s": steps / dt,
                    "seconds_in_record_assignments": dt_copy,
                    "particle_steps_per_s_without_record_assignments": steps / max(dt - dt_copy, 1e-9),
                    "crossings": int(codes.sum()), "alive_at_end": int(alive_rec[-1].sum())}
        print("%s: %d particle-steps in %.2f s = %.3e /s (%.2f s of it the per-record grid assignments; %.3e /s without)"
              % (cfg, steps, dt, steps / dt, dt_copy, out[cfg]["particle_steps_per_s_without_record_assignments"]), file=sys.stderr)
        del grid, u, v, sic
    print(json.dumps(out))


if __name__ == "__main__":
    main()
