#!/usr/bin/env python3
"""The command line end to end on a case of the size the reference is used at (its demo is the NANUK4 configuration:
a ~566 x 492 regional mesh, hourly SI3 records, 10^3..10^4 buoys), with synthetic NetCDF inputs of that shape.

    python tests/sweeps/demo_cli.py [--nj 566 --ni 492 --dkm 12.5 --buoys 20000 --records 120]

Times `si3_part_tracker.py -i ... -m ... -s ... -F` (files in, files out) and checks every output array against a run of
the whole driver restated on the CPU oracle (exhaustive-scan SeedInit + every record; 16 threads).  Prints one JSON object.
The reference itself cannot run here (netCDF4, cartopy and mojito are absent); its own functions in its loop shape were
timed at 3.5e4 particle-steps/s on one core of the build container (tests/golden/gen_golden.py::reference_loop)."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_driver as td                       # noqa: E402  (case generator + oracle restatement of the driver)
from sitrack_amd import driver as drv, ncio    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nj", type=int, default=566)
    ap.add_argument("--ni", type=int, default=492)
    ap.add_argument("--dkm", type=float, default=12.5)
    ap.add_argument("--buoys", type=int, default=20000)
    ap.add_argument("--records", type=int, default=120)
    ap.add_argument("--big", action="store_true",
                    help="config-5 shape: the reference's default 2-D-time mode (first/last positions written), any number "
                         "of seeds; the check is an oracle replay of every 4000th kept buoy from the seed cache the run wrote")
    a = ap.parse_args()
    if a.big:
        return big(a)
    tmp = tempfile.mkdtemp(prefix="sitrk_cli_")
    os.chdir(tmp)
    t = time.perf_counter()
    c = td.make_case(tmp, nrec=a.records, nP=a.buoys, Nj=a.nj, Ni=a.ni, dkm=a.dkm)
    t_make = time.perf_counter() - t
    print("inputs written (%.1f s): icemod %.0f MB" % (t_make, os.path.getsize(c["si3"]) / 1e6), file=sys.stderr, flush=True)
    argv = ["-i", c["si3"], "-m", c["mm"], "-s", c["seed"], "-N", "TEST4", "-F"]
    t = time.perf_counter()
    out = drv.main(argv)
    t_cli = time.perf_counter() - t
    print("command line done (%.2f s); oracle run of the same case ..." % t_cli, file=sys.stderr, flush=True)
    t = time.perf_counter()
    ref = td.oracle_run(c, False, nthreads=16)
    t_ref = time.perf_counter() - t
    assert out["nP"] == ref["nP"] and np.array_equal(out["IDs"], ref["ids"])
    assert np.array_equal(out["vJIt"], ref["jiT"]) and np.array_equal(out["iAlive"], ref["alive"])
    _, ids, _, yxo, mko = ncio.LoadNCdata(out["files"][0], krec=-1, lmask=True)
    assert np.array_equal(ids, ref["ids"]) and np.array_equal(mko, ref["msk"])
    assert np.array_equal(yxo.astype('f4'), ref["pos"].astype('f4'))
    psteps = float(ref["msk"][1:].sum())
    print(json.dumps({
        "case": "synthetic %dx%d mesh (%.1f km), %d seeds -> %d buoys kept, %d hourly records, -F (full series written)"
                % (a.nj, a.ni, a.dkm, a.buoys, int(out["nP"]), a.records),
        "particle_steps": psteps, "alive_at_end": int((ref["alive"] == 1).sum()),
        "cli_wall_s": t_cli, "cli_particle_steps_per_s": psteps / t_cli,
        "oracle_driver_16_threads_wall_s": t_ref,
        "reference_python_estimate_s": psteps / 3.5e4,
        "outputs_identical_to_oracle_run": True, "files": [os.path.basename(f) for f in out["files"]]}))


def big(a):
    import glob
    from oracle import oracle as orc
    tmp = tempfile.mkdtemp(prefix="sitrk_cli_")
    os.chdir(tmp)
    t = time.perf_counter()
    c = td.make_case(tmp, nrec=a.records, nP=a.buoys, Nj=a.nj, Ni=a.ni, dkm=a.dkm, two_d_time=True)
    print("inputs written (%.1f s)" % (time.perf_counter() - t), file=sys.stderr, flush=True)
    argv = ["-i", c["si3"], "-m", c["mm"], "-s", c["seed"], "-N", "TEST4"]
    t = time.perf_counter()
    out = drv.main(argv)
    t_cli = time.perf_counter() - t
    print("command line done (%.2f s); oracle replay of a subsample ..." % t_cli, file=sys.stderr, flush=True)
    with np.load(glob.glob("./seed/Initialized_buoys_*.npz")[0]) as z:
        nP, xPosC0, vJIt0, keep = int(z["nP"]), z["xPosC0"], z["vJIt"], z["idxKeep"]
    imaskt, _, _, _, _, xYf, xXf, _ = ncio.GetModelGrid(c["mm"])
    xYv, xXv, xYu, xXu = ncio.GetModelUVGrid(c["mm"])
    grid = dict(Yf=xYf, Xf=xXf, Yu=xYu, Xu=xXu, Yv=xYv, Xv=xXv, tmask=imaskt)
    tc, base, n0 = c["tc"], c["base"], len(c["ids"])
    tp0 = np.full(n0, base); tp0[::7] = base + 4 * 3600
    tp1 = np.full(n0, tc[-1] + 1800); tp1[::5] = base + 9 * 3600
    z1, zL = drv.record_windows(np.stack([tp0, tp1]), tc, 0, len(tc) - 1, tc[0], tc[-1], n0)
    z1, zL = z1[keep], zL[keep]
    sub = np.arange(0, nP, 4000)
    ref = orc.Tracker(grid, xPosC0[sub], vJIt0[sub], rec_first=z1[sub], rec_last=zL[sub], nthreads=8)
    last = xPosC0[sub].copy()
    for jt in range(len(tc)):
        pn, mn = ref.step(jt, c["u"][jt].astype('f8'), c["v"][jt].astype('f8'), c["sic"][jt].astype('f8'))
        last[mn == 1] = pn[mn == 1]
    assert np.array_equal(out["vJIt"][sub], ref.jiT) and np.array_equal(out["iAlive"][sub], ref.alive)
    _, _, _, yx2, mk2 = ncio.LoadNCdata(out["files"][0], krec=-1, lmask=True)
    ok = mk2[1][sub] == 1
    assert np.array_equal(yx2[1][sub][ok].astype('f4'), last[ok].astype('f4'))
    psteps = float(np.clip(zL - z1 + 1, 0, None).sum())              # upper bound: buoys killed early step fewer records
    print(json.dumps({
        "case": "synthetic %dx%d mesh (%.1f km), %d seeds -> %d buoys kept, %d hourly records, 2-D-time mode (first/last written)"
                % (a.nj, a.ni, a.dkm, a.buoys, nP, a.records),
        "cli_wall_s": t_cli, "particle_steps_upper_bound": psteps, "alive_at_end": int((out["iAlive"] == 1).sum()),
        "where_the_time_goes_s": {k: round(v, 3) for k, v in out["timing"].items()},
        "launches": out["launches"],
        "subsample_checked_against_oracle": int(len(sub)), "files": [os.path.basename(f) for f in out["files"]]}))


if __name__ == "__main__":
    main()
