#!/usr/bin/env python3
"""bench.py -- particle-steps/s of the per-buoy advection hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one model record applied to every buoy of the batch.  `sitrk_run` advances 32 resident records per launch
(advect_run_kernel: loop interchange, every buoy still takes every step -- the form the command line runs too); the same
K steps with one launch of advect_step_kernel per record and with 8 records per launch are timed in the same run and
reported under `per_record_launch` / `eight_records_per_launch`.

Workload: N = 1 -> BASELINE.json configs[2] (C3, the one the metric is quoted on): synthetic regular 4096x4096 C-grid
(4 km), 1e7 random buoys in the central 60 %, 32 device-resident fp32 records (solid-body rotation + per-record drift,
SURVEY.md 8d) cycled.  N > 1 -> configs[3] (C4): one process per GPU, each rank owns 1.25e7 buoys (1e8 over 8 GPUs; weak
scaling), the record slabs are generated on rank 0 and broadcast over RCCL in place into every rank's resident slots;
stepping needs no collective.  `c2` = configs[1] measured on the same GPU in the same process.  `--warp W` shears the mesh
(SURVEY 8d's optional curvilinear variant), `--config c5shape` is the NANUK4-shaped 566 x 492 curvilinear mesh with >= 1e7
seeds kept by the product's own SeedInit under a synthetic land / ice mask (BASELINE configs[4]'s shape).

Prints ONE JSON line (rank 0).
 * `value`: the K resident records are cycled, so each record's Survive bytes (tracking.py:62-93 once per cell) are derived
   once and reused K/records_resident times OUTSIDE the clock.  `fresh_records.value` = the same steps with every record
   committed afresh (Survive re-derived over the box the buoys can touch, one launch per batch of records) INSIDE the clock.
 * `roofline` of the fused kernel: bound "valu_issue+wave_chain"; achieved = VALU instructions per second (instructions per
   wave and record from the committed rocprofv3 counters in profiles/traffic.json x waves x records really advanced, counted
   by the library: sitrk_launch_stats) against a class-weighted issue ceiling (a model) and the uniform 4-cycle ceiling;
   `lane_utilisation` = share of the issued lanes that do work; `stale` = the profiled binary is not the one that ran
   (kernel fingerprints, tools/kernel_fingerprint.py) -- `frac` is dropped then.  `traffic` = measured fabric bytes per launch
   (FETCH_SIZE x 2 + WRITE_SIZE of the same command, profiles/); `hbm` = the algorithmic-byte view.  Algorithmic bytes follow
   SURVEY.md 8(d) -- 50 B of state per buoy + 56 B (48 B geometry + u,v) per grid cell a step needs.
 * `per_record_launch.roofline`: bound "hbm", algorithmic bytes of one record over the measured launch time.
 * `cpu_baseline`: the CPU oracle (oracle/sitrk_oracle.c, a port of the reference loop, OpenMP over buoys) timed on this
   box's host cores on a bounded sample of the same workload.
 * N = 1: `c4_shard` = C4's per-rank shard (1.25e7 buoys) on this one GPU -- the N = 1 point of the weak-scaling curve --
   and `e2e_upload` = one record per step from host memory through the library's pinned staging (whole records and boxes).
 * N > 1: `solo_same_shard` (rank 0 alone, the others idle) and `efficiency_vs_n1_same_shard`; `e2e_broadcast` = a short
   segment with ONE broadcast of every record's slab per step overlapped with the stepping (RCCL broadcast and scatter +
   all-gather), next to the resident `value`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 measured achievable

VALU_PEAK_GINST = 1024 * 2.4 / 4.0   # 256 CUs x 4 SIMDs, 2.4 GHz max clock, one wave64 VALU instruction per 4 cycles per SIMD
                                     # (the rate SQ_ACTIVE_INST_VALU counts in; MI355X_MICROARCH.md, cycle constants) = 614.4 G/s

# ---- one line, whatever happens (N > 1) --------------------------------------------------------------------------
# A multi-rank run must never leave the driver guessing: the JSON line is printed exactly once, and any failure of the
# exchange path (a collective that raises, one that never completes, a rank that dies and has the launcher terminate the
# others) prints it with "degraded": "<reason>" and "value": null, and the process exits NON-ZERO.  Nothing is retried
# in-process and nothing is re-exec'ed.
import threading

_STATE = {"base": {}, "phase": "start", "t_phase": time.time(), "printed": False, "rank": 0, "multi": False}
_EMIT_LOCK = threading.Lock()

EXIT_E2E_FAILED = 3        # the resident numbers are in the line; the end-to-end broadcast segment failed or hung
EXIT_DEGRADED = 4          # the record broadcast (or another collective `value` depends on) failed: value is null
EXIT_EXCEPTION = 5         # any other exception on a multi-rank run
EXIT_NO_PROGRESS = 6       # watchdog: a phase did not end (a collective that never completes)
EXIT_TERMINATED = 7        # the launcher terminated this rank (another rank failed)


def _phase(name):
    _STATE["phase"] = name
    _STATE["t_phase"] = time.time()


def _emit(line):
    """print the JSON line once, whoever gets here first (main flow, watchdog thread or signal watcher)"""
    with _EMIT_LOCK:
        if _STATE["printed"]:
            return False
        _STATE["printed"] = True
        sys.stdout.write(json.dumps(line) + "\n")
        sys.stdout.flush()
        return True


def _degraded_exit(reason, code):
    """rank 0: the line with value null and the reason; every rank: leave with a non-zero code, at once, without touching
    a communicator that may be stuck (no orderly teardown is possible then)"""
    try:
        if _STATE["rank"] == 0:
            line = dict(_STATE["base"])
            line.update({"value": None, "degraded": reason, "phase": _STATE["phase"]})
            _emit(line)
        sys.stderr.write("bench.py rank %d: %s (phase %s) -> exit %d\n" % (_STATE["rank"], reason, _STATE["phase"], code))
        sys.stderr.flush()
    finally:
        os._exit(code)


def _start_guards():
    """multi-rank runs only.  (1) a watcher on a signal wake-up pipe: SIGTERM from the launcher reaches it even while the
    main thread sits inside a collective (Python-level handlers only run between bytecodes of the main thread);
    (2) a no-progress watchdog, longer than the process group's own timeout so that torch's error surfaces first."""
    import signal
    rfd, wfd = os.pipe()
    os.set_blocking(wfd, False)
    signal.set_wakeup_fd(wfd, warn_on_full_buffer=False)
    signal.signal(signal.SIGTERM, lambda *_: None)        # installs the C-level handler that writes to the pipe

    def _sig_watch():
        while True:
            b = os.read(rfd, 1)
            if b and b[0] == signal.SIGTERM:
                _degraded_exit("terminated by the launcher (SIGTERM): another rank failed or the job was cancelled", EXIT_TERMINATED)

    def _watchdog():
        limit = float(os.environ.get("SITRK_BENCH_PHASE_TIMEOUT", "420"))
        while True:
            time.sleep(min(5.0, limit / 4))
            if _STATE["phase"] == "done":
                return
            if time.time() - _STATE["t_phase"] > limit:
                _degraded_exit("no progress for %.0f s (a collective that never completes?)" % limit, EXIT_NO_PROGRESS)

    threading.Thread(target=_sig_watch, daemon=True).start()
    threading.Thread(target=_watchdog, daemon=True).start()


def _device_identity(torch, dev):
    """what tells two GPUs apart: index, name, UUID (or PCI address) -- gathered from every rank into the line"""
    out = {"device": int(dev)}
    try:
        pr = torch.cuda.get_device_properties(dev)
        out["name"] = pr.name
        u = getattr(pr, "uuid", None)
        out["uuid"] = str(u) if u is not None else None
        for k in ("pci_bus_id", "pci_device_id", "pci_domain_id"):
            if hasattr(pr, k):
                out[k] = int(getattr(pr, k))
    except Exception as e:                                          # noqa: BLE001
        out["error"] = repr(e)
    return out


def _slot_checksum(torch, sd, ctx, slot):
    """exact checksum of a resident slab as it sits in HBM: sum of its 32-bit words as int64 (then the mask is re-derived,
    because handing out the slot's pointer marks it as rewritten)"""
    t = sd.slot_tensor(ctx, slot)
    c = int(t.view(torch.int32).to(torch.int64).sum().item())
    ctx.commit_record(slot)
    return c


CONFIGS = {
    # name: (Nj, Ni, buoys per GPU, label)
    "c3": (4096, 4096, 10_000_000, "C3: synthetic 4096x4096 C-grid, 1e7 buoys/GPU, fp32 records"),
    "c2": (512, 512, 100_000, "C2: synthetic 512x512 C-grid, 1e5 buoys/GPU, fp32 records"),
    # BASELINE.json configs[3]: 1e8 buoys over 8 GPUs = 1.25e7 per rank.  Selected when --gpus > 1 (weak scaling: the
    # per-rank shard is C4's at every N, the total is C4's 1e8 at N = 8)
    "c4": (4096, 4096, 12_500_000, "C4: synthetic 4096x4096 C-grid, 1.25e7 buoys/GPU (1e8 over 8 GPUs), fp32 records, "
                                   "buoy-range partition + RCCL record broadcast"),
    # BASELINE.json configs[4]'s SHAPE: the real NANUK4 mesh_mask is not in the container -- a curvilinear 566 x 492 mesh at
    # 12.5 km (the NANUK4-shaped mesh of tests/test_gpu_configs.py), land + open water, 1.13e7 candidate seeds of which the
    # product's SeedInit keeps those under the mask (>= 1e7)
    "c5shape": (566, 492, 11_300_000, "C5 shape: NANUK4-shaped curvilinear 566x492 mesh (12.5 km, synthetic), >= 1e7 seeds kept by "
                                      "SeedInit under a land / ice mask, hourly fp32 records"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", default="auto", choices=["auto"] + sorted(CONFIGS),
                    help="auto = C3 (the headline configuration) on one GPU, C4's per-rank shard on several")
    ap.add_argument("--warp", type=float, default=0.0,
                    help="shear / stretch of the synthetic mesh (SURVEY 8d's curvilinear variant; 0 = axis-aligned squares, 1 = the "
                         "warp of the parity tests); c5shape is always warped")
    ap.add_argument("--buoys", type=int, default=0, help="override the configuration's buoys per GPU (capacity runs; not the metric's workload)")
    ap.add_argument("--records", type=int, default=32, help="device-resident records, cycled (32 x 201 MB at 4096^2)")
    ap.add_argument("--resort", type=int, default=-1, help="re-sort buoys by cell every R steps (0 never, -1 default)")
    ap.add_argument("--uv-strategy", type=int, default=1)
    ap.add_argument("--no-sort", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--check", action="store_true", help="verify a subsample against the oracle after the run")
    ap.add_argument("--fuse", type=int, default=32,
                    help="resident records advanced per launch by sitrk_run (loop interchange; 1 = one launch per record)")
    ap.add_argument("--e2e-full", action="store_true", help="e2e regime: upload whole records instead of the box the buoys can touch")
    ap.add_argument("--e2e-rows", action="store_true", help="e2e regime: upload row bands (round 1-3's form) instead of boxes")
    ap.add_argument("--e2e-library", action="store_true",
                    help="e2e regime through the library's own pinned staging and copy stream (sitrk_push_record_box), host fill included")
    ap.add_argument("--only-fused", action="store_true",
                    help="skip every leg but the timed fused run (one record per launch, 8 records per launch, fresh records, C4 "
                         "shard, e2e upload): profiling runs, so that every dispatch of advect_run_kernel in the trace is a launch of "
                         "the timed configuration")
    ap.add_argument("--tune", default="", help="library tuning knobs for A/B runs, e.g. patch_kb=0,sort_tile=2080 (never change results)")
    ap.add_argument("--no-c2", action="store_true", help="skip the C2 (512x512, 1e5 buoys, 1000 steps) sub-measurement")
    ap.add_argument("--no-fresh", action="store_true", help="skip the fresh-records leg (every record committed once, inside the clock)")
    ap.add_argument("--fresh-overlap", action="store_true",
                    help="fresh-records leg with the commits on the library's ingest stream, next to the launch that steps with the other "
                         "half of the slot ring (measured slower than committing on the compute stream: not the default)")
    ap.add_argument("--no-c4-shard", action="store_true", help="N = 1: skip the C4 per-rank shard sub-measurement")
    ap.add_argument("--no-e2e-upload", action="store_true", help="N = 1: skip the short end-to-end upload segment")
    ap.add_argument("--no-e2e-broadcast", action="store_true",
                    help="N > 1: skip the short end-to-end segment (one RCCL broadcast per record, overlapped with stepping)")
    ap.add_argument("--regime", default="resident", choices=["resident", "e2e"],
                    help="resident: records live in HBM (the metric). e2e: every step's record is uploaded from pinned host "
                         "memory on rank 0 (+ RCCL broadcast), double-buffered against the stepping; reported for context only")
    return ap.parse_args()


# The reference's OWN Python loop cannot run on the GPU box (no /root/reference there): it was timed in the build container
# by tools/time_reference_loop.py -- the imported reference functions in the loop order of si3_part_tracker.py:378-490, 10^3
# buoys x 100 records cut from the C2 / C3 inputs, one core -- and travels as constants with their provenance
# (profiles/r03b_reference_python_loop.json).
REFERENCE_PYTHON = {
    "c2": 40867.0, "c3": 10692.1, "c3_without_the_per_record_grid_assignments": 24476.0,
    "unit": "particle-steps/s", "cores": 1, "cpu": 'Intel(R) Xeon(R) Processor @ 2.10GHz', "date": '2026-10-05',
    "sample": "first 1000 buoys x 100 records of the C2 / C3 inputs, iUVstrategy = 1",
    "source": "tools/time_reference_loop.py -> profiles/r03b_reference_python_loop.json (build container; the GPU box has no reference)",
}


def cpu_baseline(grid, u, v, sic, yx, ji, target_s, uv_strategy):
    """Oracle on host cores: bounded sample of the same workload (first nS buoys x a few records)."""
    from oracle import oracle as orc
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)      # the GPU box's CPU share for one GPU is 16 cores
    K = u.shape[0]
    f64 = [(u[k].astype(np.float64), v[k].astype(np.float64), sic[k].astype(np.float64)) for k in range(min(K, 2))]
    out = {}
    for nthreads, nS in ((1, min(len(yx), 200_000)), (cores, min(len(yx), 2_000_000))):
        trk = orc.Tracker(grid, yx[:nS], ji[:nS], uv_strategy=uv_strategy, nthreads=nthreads)
        trk.step(0, *f64[0], want_out=False)                     # warm
        t0 = time.perf_counter()
        nrec = 0
        while True:
            trk.step(1 + nrec, *f64[nrec % len(f64)], want_out=False)
            nrec += 1
            if time.perf_counter() - t0 > target_s / 2 or nrec >= 200:
                break
        dt = time.perf_counter() - t0
        out[nthreads] = (nS * nrec / dt, nS, nrec)
        cross = trk.ncross / float(nS * (nrec + 1))      # (the warm-up record counts too)
    rate, nS, nrec = out[cores]
    return {"value": rate, "unit": "particle-steps/s", "cores": cores, "kind": "port", "crossing_rate": cross,
            "sample": "first %d buoys x %d records of the same workload, fp64 oracle with OpenMP over buoys" % (nS, nrec),
            "value_1core": out[1][0], "sample_1core": "%d buoys x %d records" % (out[1][1], out[1][2]),
            "reference_python": REFERENCE_PYTHON}


def cpu_baseline_check(ctx, grid, u, v, sic, yx, ji, nsteps, uv_strategy, nS=20000):
    """cpu_baseline leg, optional part (--check): the oracle replays the run on the first nS buoys and the GPU state must
    agree bit for bit (checker only -- nothing here is timed or fed back)."""
    from oracle import oracle as orc
    K = u.shape[0]
    ref = orc.Tracker(grid, yx[:nS], ji[:nS], uv_strategy=uv_strategy, nthreads=8)
    f64 = [(u[k].astype(np.float64), v[k].astype(np.float64), sic[k].astype(np.float64)) for k in range(K)]
    for s in range(nsteps):
        ref.step(s, *f64[s % K], want_out=False)
        if s % 200 == 199:
            _phase("oracle check")                      # (keeps the no-progress watchdog of multi-rank runs quiet)
        if s % 2000 == 1999:
            print("check: oracle at step %d / %d" % (s + 1, nsteps), file=sys.stderr, flush=True)
    st = ctx.fetch()
    assert np.array_equal(st["yx"][:nS], ref.pos) and np.array_equal(st["jiT"][:nS], ref.jiT)
    assert np.array_equal(st["alive"][:nS], ref.alive)
    print("check OK: first %d buoys bit-exact vs oracle after %d steps" % (nS, nsteps), file=sys.stderr)


def make_workload(a, syn, ctx, rank, config):
    """Grid, buoys (positions + host cells through the product's own locate) and the parameters of the synthetic fields for a
    configuration.  Returns a dict; the grid is set on `ctx`."""
    Nj, Ni, nP, label = CONFIGS[config]
    if a.buoys > 0:
        nP = a.buoys
        label += " [--buoys %d override]" % nP
    w = {"config": config, "Nj": Nj, "Ni": Ni, "label": label, "seeding": "uniform in the central 60 %"}
    if config == "c5shape":
        grid = syn.shift_grid(syn.make_grid(Nj, Ni, dkm=12.5, warp=1.0), -250., 150.)
        tmask = grid["tmask"]
        tmask[Nj // 3:Nj // 3 + Nj // 12, Ni // 2:Ni // 2 + Ni // 10] = 0            # an island
        w["fields"] = dict(seed=77, umax=0.9, drift=0.3, ripple=0.1)
        w["polynya"] = (slice(Nj // 2, Nj // 2 + Nj // 10), slice(Ni // 5, Ni // 5 + Ni // 6))   # open water: buoys that drift in die
    else:
        grid = syn.make_grid(Nj, Ni, dkm=4.0, warp=a.warp)
        w["fields"] = dict(seed=2024, umax=0.3, drift=0.05)
        if a.warp:
            w["label"] = label = label.replace("synthetic ", "synthetic curvilinear (warp %g) " % a.warp)
    w["grid"] = grid
    ctx.set_grid(grid["Yf"], grid["Xf"], grid["Yu"], grid["Xu"], grid["Yv"], grid["Xv"], grid["tmask"])
    ctx.set_params(3600., a.uv_strategy, 0.1)
    w["yx"], w["ji"] = make_shard(a, syn, ctx, w, nP, 1234 + rank)
    w["nP"] = len(w["yx"])
    return w


def make_shard(a, syn, ctx, w, nP, seed):
    """nP buoys of workload `w` and their host cells"""
    grid, Nj, Ni = w["grid"], w["Nj"], w["Ni"]
    if w["config"] == "c5shape":
        # the reference's own path: seeds (lat/lon + km, float32 like a seeding file) -> SeedInit (nearest T-point by Haversine,
        # Survive, FindContainingCell: sitrack/tracking.py:98-178) on the device -> the kept ones are tracked
        rng = np.random.default_rng(seed)
        yx = np.stack([rng.uniform(grid["Yt"].min() + 30, grid["Yt"].max() - 30, nP), rng.uniform(grid["Xt"].min() + 30, grid["Xt"].max() - 30, nP)], axis=1)
        yx = yx.astype(np.float32).astype(np.float64)
        latlonT = ctx.cart2geo(np.stack([grid["Yt"].ravel(), grid["Xt"].ravel()], axis=1))
        latT, lonT = latlonT[:, 0].reshape(Nj, Ni), np.mod(latlonT[:, 1], 360.).reshape(Nj, Ni)
        sll = ctx.cart2geo(yx)
        sll[:, 1] = np.mod(sll[:, 1], 360.)
        sic0 = np.ones((Nj, Ni))
        sic0[w["polynya"]] = 0.03
        ji, keep, why = ctx.seed_init(sll, yx, latT, lonT, grid["resol"], sic0)
        k = keep == 1
        w["seeding"] = ("%d candidate seeds -> SeedInit kept %d (no nearest point %d, Survive %d, no containing cell %d)"
                        % (nP, int(k.sum()), int((why == 1).sum()), int((why == 2).sum()), int((why == 3).sum())))
        return np.ascontiguousarray(yx[k]), np.ascontiguousarray(ji[k]).astype(np.int32)
    _, yx = syn.make_buoys(grid, nP, seed=seed, frac=0.6)
    if grid["warp"] == 0.0:
        ji = syn.regular_host_cell(grid, yx).astype(np.int32)
        found, ji2 = ctx.find_cells(yx, ji)              # the product's own FindContainingCell also validates the analytic guess
        assert found.all() and np.array_equal(ji2, ji), "host-cell seeding failed"
        return yx, ji
    found, ji = ctx.find_cells(yx, syn.nearest_t_index(grid, yx).astype(np.int32))
    assert found.mean() > 0.999, "host-cell seeding failed on the warped mesh (%.4f found)" % found.mean()
    return np.ascontiguousarray(yx[found]), np.ascontiguousarray(ji[found])


def make_records(syn, w, K):
    u, v, sic = syn.make_fields(w["grid"], K=K, **w["fields"])
    if "polynya" in w:
        sic[(slice(None),) + w["polynya"]] = 0.03
    return u, v, sic


def c2_subrun(sit, syn, dev, a, steps=1000, warmup=64):
    """BASELINE.json configs[1] (C2: 512x512, 1e5 buoys, 1000 steps) on the same GPU, same library defaults: the grid and
    the state are cache-resident and a launch is under-filled (391 workgroups), so this is a latency figure -- reported
    as particle-steps/s only (the HBM fraction is not meaningful there)."""
    Nj, Ni, nP, label = CONFIGS["c2"]
    K = 32
    grid = syn.make_grid(Nj, Ni, dkm=4.0, warp=0.0)
    _, yx = syn.make_buoys(grid, nP, seed=1234, frac=0.6)
    ji = syn.regular_host_cell(grid, yx).astype(np.int32)
    u, v, sic = syn.make_fields(grid, K=K, seed=2024, umax=0.3, drift=0.05)
    ctx = sit.Context(dev)
    try:
        ctx.set_grid(grid["Yf"], grid["Xf"], grid["Yu"], grid["Xu"], grid["Yv"], grid["Xv"], grid["tmask"])
        ctx.set_params(3600., a.uv_strategy, 0.1)
        ctx.alloc_records(K, np.float32)
        for k in range(K):
            ctx.push_record(k, u[k], v[k], sic[k])
        ctx.set_buoys(yx, ji)
        ctx.set_tuning(fuse=32)
        ctx.run(0, 0, warmup)
        ctx.sync()
        ctx.launch_stats(reset=True)
        ctx.timer_start()
        t0 = time.perf_counter()
        ctx.run(warmup % K, warmup, steps)
        ms = ctx.timer_stop()
        ctx.sync()
        dt = time.perf_counter() - t0
        st = ctx.launch_stats(reset=True)
        return {"workload": label, "steps": steps, "warmup": warmup, "value": nP * steps / dt, "unit": "particle-steps/s",
                "ms_per_step": 1e3 * dt / steps, "event_ms_per_step": ms / steps, "launches": st["fused_launches"] + st["step_launches"],
                "records_per_launch": st["fused_records"] / max(st["fused_launches"], 1), "alive_after": ctx.count_alive(),
                "note": "cache-resident and bound by ONE wave's dependent chain per record (1 563 waves for 1 024 SIMDs): no "
                        "roofline fraction is claimed"}
    finally:
        ctx.close()


def fresh_records_leg(ctx, K, fuse, s0, nsteps, sync, overlap=False):
    """The timed run once more with NOTHING amortised over the record cycling: every record stepped with is committed afresh
    right before the launch that uses it (sitrk_commit_records_box: its Survive bytes re-derived from its siconc, over the box the
    buoys can touch, one Survive launch per fused launch), inside the clock.  What a run over distinct records pays per record
    when the slabs arrive in device memory (an RCCL broadcast, a device-side producer); an upload over PCIe hides it (e2e_upload).
    The box comes from sitrk_buoy_box_begin/_end: the evaluation is queued one launch ahead and collected while the next launch
    runs, so the stream never drains; a box is therefore up to 2 x fuse - 1 records old and that many cells wider all around.
    overlap=True (bench.py --fresh-overlap; measured, NOT the default: profiles/r04r_*): launches of half the slot ring, the other
    half committed on the library's ingest stream next to the running launch (sitrk_commit_records_box_async) -- the fused loop
    needs its seven waves per SIMD, a co-running memory-bound kernel costs it more than the 11 us per record it hides."""
    m = max(1, min(fuse, K // 2 if K > 1 else 1)) if overlap else fuse
    ctx.set_tuning(fuse=m)
    sync()
    ctx.launch_stats(reset=True)
    ev, ev_age = ctx.buoy_box(), 0                       # the evaluation in hand and the records stepped since its begin
    pending = False
    cells = []
    ctx.timer_start()
    t0 = time.perf_counter()

    def commit(k, cnt, age):
        box = ctx.box_of(*ev, age + cnt - 1)
        ctx.commit_records_box((s0 + k) % K, cnt, *box, on_ingest_stream=overlap)
        cells.append((box[1] - box[0]) * (box[3] - box[2]))

    if overlap:
        commit(0, min(m, nsteps), ev_age)
    k = 0
    while k < nsteps:
        mk = min(m, nsteps - k)
        if pending:
            r = ctx.buoy_box_end()                       # queued before the previous launch: does not wait for it
            ev, ev_age, pending = r[:4], r[4], False
        k2 = k + mk
        m2 = min(m, nsteps - k2)
        if m2 > 0 and (not overlap or k2 + m2 < nsteps): # (the last launches need no fresher box)
            ctx.buoy_box_begin()
            pending = True
        if not overlap:
            commit(k, mk, ev_age)
        ctx.run((s0 + k) % K, s0 + k, mk)
        ev_age += mk
        if overlap and m2 > 0:                           # the other half of the ring, prepared while the launch above runs
            commit(k2, m2, ev_age)
        k = k2
    ms = ctx.timer_stop()
    sync()
    dt = time.perf_counter() - t0
    if pending:
        ctx.buoy_box_end()
    st = ctx.launch_stats(reset=True)
    # the Survive pass alone, same box, same batch size (its share of the time above)
    mm = min(m, K)
    box = ctx.box(mm - 1)
    ctx.commit_records_box(0, mm, *box)
    ctx.sync()
    ctx.timer_start()
    for _ in range(5):
        ctx.commit_records_box(0, mm, *box)
    sv_ms = ctx.timer_stop()
    for slot in range(K):
        ctx.commit_record(slot)                          # the resident legs that follow see whole records again
    ctx.sync()
    ctx.set_tuning(fuse=fuse)
    return {"dt": dt, "event_ms": ms, "stats": st, "box_cells_mean": float(np.mean(cells)), "survive_us_per_record": 1e3 * sv_ms / (5 * mm),
            "box": list(box), "records_per_launch": m, "overlap": overlap}


def c4_shard_subrun(a, syn, ctx, w, K, fuse, s0, barrier):
    """N = 1 only: BASELINE configs[3]'s per-rank shard (1.25e7 buoys, rank 0's seeds) on this GPU, same grid, same resident
    records, same launch shape -- the N = 1 point that the N > 1 lines' per-GPU workload can be compared with."""
    nP4 = CONFIGS["c4"][2]
    yx, ji = make_shard(a, syn, ctx, dict(w, config="c4"), nP4, 1234)
    ctx.set_buoys(yx, ji, sort=not a.no_sort)
    ctx.set_tuning(fuse=fuse)
    ctx.run(s0 % K, s0, a.warmup)
    barrier()
    ctx.launch_stats(reset=True)
    ctx.timer_start()
    t0 = time.perf_counter()
    ctx.run((s0 + a.warmup) % K, s0 + a.warmup, a.steps)
    ms = ctx.timer_stop()
    barrier()
    dt = time.perf_counter() - t0
    st = ctx.launch_stats(reset=True)
    return {"workload": CONFIGS["c4"][3], "buoys": len(yx), "steps": a.steps, "warmup": a.warmup, "value": len(yx) * a.steps / dt,
            "unit": "particle-steps/s", "ms_per_step": 1e3 * dt / a.steps, "event_ms_per_step": ms / a.steps,
            "launches": st["fused_launches"] + st["step_launches"], "alive_after": ctx.count_alive(),
            "note": "what `bench.py --gpus N` gives every rank (rank 0's seeds): the same-shard N = 1 point of the weak-scaling curve"}


def e2e_upload_segment(ctx, w, u, v, sic, K, s0, nsteps=48):
    """N = 1 counterpart of `e2e_broadcast` (same keys per mode): every step's record comes from ordinary host arrays through
    the library's pinned staging and copy stream (sitrk_push_record / sitrk_push_record_box: host gather by a few threads, then
    DMA), double-buffered against the stepping -- whole records, and the box the buoys can touch.  PCIe bound; never `value`."""
    Nj, Ni = w["Nj"], w["Ni"]
    out = {"steps": nsteps, "slab_bytes": int(3 * Nj * Ni * 4), "note": "one record per step from host memory, overlapped with the stepping; not `value`"}
    for mode in ("whole_record", "box"):
        st = {"eval": -10**9, "box": (0, Nj, 0, Ni), "bytes": 0, "pending": False}

        def deliver(sidx):
            k = sidx % K
            if mode == "whole_record":
                ctx.push_record(sidx % 2, u[k], v[k], sic[k])
                st["bytes"] += 3 * Nj * Ni * 4
                return
            j0, j1, i0, i1 = ctx.box_of(*st["box"], sidx - st["eval"])
            ctx.push_record_box(sidx % 2, j0, j1, i0, i1, u[k][j0:j1, i0:i1], v[k][j0:j1, i0:i1], sic[k][j0:j1, i0:i1])
            st["bytes"] += 3 * (j1 - j0) * (i1 - i0) * 4

        def run(first, n):
            for sidx in range(first, first + n):
                if mode == "box":
                    if st["pending"] and sidx - st["begin"] >= 8:
                        r = ctx.buoy_box_end()                       # queued 8 records ago: no wait to speak of
                        st["box"], st["eval"], st["pending"] = r[:4], st["begin"], False
                    if not st["pending"] and sidx - st["eval"] >= 16:
                        ctx.buoy_box_begin()
                        st["pending"], st["begin"] = True, sidx
                if sidx == first:
                    deliver(sidx)
                ctx.step(sidx % 2, sidx)
                if sidx + 1 < first + n:
                    deliver(sidx + 1)                  # its DMA waits for the step above only if it reuses that slot: it does not
                if sidx % 16 == 0:
                    _phase("e2e upload segment")

        if mode == "box":
            st["box"], st["eval"] = ctx.buoy_box(), s0
        run(s0, 4)
        ctx.sync()
        st["bytes"] = 0
        t0 = time.perf_counter()
        run(s0 + 4, nsteps)
        ctx.sync()
        dt = time.perf_counter() - t0
        if st["pending"]:
            ctx.buoy_box_end()
        s0 += 4 + nsteps
        out[mode] = {"ms_per_step": 1e3 * dt / nsteps, "slab_GBps_per_rank": st["bytes"] / nsteps / (dt / nsteps) / 1e9,
                     "upload_bytes_per_step": st["bytes"] / nsteps, "particle_steps_per_s": ctx.nP * nsteps / dt}
    return out, s0


_HIP = []


def _memcpy2d_h2d(dst, dpitch, src, spitch, width, height, stream):
    """hipMemcpy2DAsync host -> device on `stream` (the HIP runtime the process already runs on; torch has no strided async copy
    that does not stage through a pageable temporary)"""
    import ctypes as C
    if not _HIP:
        # the copy of the runtime this process already runs on (torch's bundled one, or the system's): never a second one
        path = next((l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64.so" in l), "libamdhip64.so")
        _HIP.append(C.CDLL(path))
        _HIP[0].hipMemcpy2DAsync.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
        _HIP[0].hipMemcpy2DAsync.restype = C.c_int
    rc = _HIP[0].hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, 1, stream)         # 1 = hipMemcpyHostToDevice
    if rc:
        raise RuntimeError("hipMemcpy2DAsync -> %d" % rc)


def isa_fingerprint(kernel="advect_run_kernel<float,1,false>"):
    """fingerprint of a kernel (default: the fused one) in the library that is loaded (written by the build: tools/kernel_fingerprint.py)"""
    try:
        import sitrack_amd._lib as L
        d = json.load(open(os.path.splitext(L.SO_PATH)[0] + ".isa.json"))
        return d["kernels"][kernel]
    except Exception:                                                # noqa: BLE001
        return None


def roofline_is_stale(profiled, shipped):
    """the counter passes behind profiles/traffic.json describe the binary `profiled`; True unless it is the one that ran"""
    return not (profiled and shipped and profiled.get("sha256") and profiled.get("sha256") == shipped.get("sha256"))


def e2e_broadcast_segment(ctx, dist, torch, sd, rank, world, slabs_host, K, s0, Nj, Ni, nsteps=24):
    """N > 1 only, reported next to the resident value, never as `value`: the regime north_star describes -- every record
    is delivered to every rank by ONE broadcast of its [u|v|siconc] slab (RCCL over xGMI) into a resident slot,
    double-buffered against the stepping: the broadcast of record s+1 runs on its own stream while record s is stepped
    with.  Two transports are timed back to back: RCCL's broadcast and the scatter + all-gather form (every link of the
    full mesh carries 1/N-th of the slab).  The slabs already resident on rank 0 are the source (device to device: the
    PCIe leg is what `--regime e2e` measures on one GPU).  A failure is reported, not fatal."""
    out = {"steps": nsteps, "slab_bytes": int(ctx.slab_elems * 4), "note": "one broadcast per record, overlapped with the stepping; not `value`"}
    try:
        comp, comm_s = torch.cuda.Stream(), torch.cuda.Stream()
        ctx.set_stream(comp.cuda_stream)
        views = [sd.slot_tensor(ctx, k) for k in range(K)]
        # two scratch slots are the broadcast targets; the source on rank 0 are the resident slots 2..K-1
        ready = [torch.cuda.Event(), torch.cuda.Event()]
        free = [torch.cuda.Event(), torch.cuda.Event()]
        for mode in ("broadcast", "scatter_allgather"):
            for e in free:
                e.record(comp)

            def deliver(sidx):
                b = sidx % 2
                with torch.cuda.stream(comm_s):
                    comm_s.wait_event(free[b])
                    if rank == 0:
                        views[b].copy_(views[2 + sidx % (K - 2)], non_blocking=True)
                    if mode == "broadcast":
                        dist.broadcast(views[b], src=0)
                    else:
                        sd.scatter_allgather(views[b], src=0)
                    ready[b].record(comm_s)

            def run(first, n):
                for sidx in range(first, first + n):
                    if sidx == first:
                        deliver(sidx)
                    if sidx + 1 < first + n:
                        deliver(sidx + 1)
                    comp.wait_event(ready[sidx % 2])
                    ctx.record_ptr(sidx % 2)                     # marks the slot as rewritten in place
                    ctx.commit_record(sidx % 2)
                    ctx.step(sidx % 2, sidx)
                    free[sidx % 2].record(comp)

            run(s0, 4)
            ctx.sync(); torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            run(s0 + 4, nsteps)
            ctx.sync(); torch.cuda.synchronize(); dist.barrier()
            dt = time.perf_counter() - t0
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
            s0 += 4 + nsteps
            out[mode] = {"ms_per_step": 1e3 * dt / nsteps, "slab_GBps_per_rank": out["slab_bytes"] / (dt / nsteps) / 1e9,
                         "particle_steps_per_s": ctx.nP * world * nsteps / dt}
        # round 4: the same with ONE broadcast of the BOX the ranks' buoys can touch (the union of every rank's box, one
        # all-reduce of four integers before the segment, wide enough for all its records): the source packs the box out of
        # its slot, the broadcast moves 3 x rows x columns elements instead of the slab, the receivers unpack into their slots
        # and commit that box.  Its own try: a failure here leaves the two whole-slab figures standing.
        try:
            for e in free:
                e.record(comp)
            box = sd.union_box(ctx.box(4 + nsteps + 1), Nj, Ni)
            buf = torch.empty(max(1, 3 * (box[1] - box[0]) * (box[3] - box[2])), dtype=views[0].dtype, device=views[0].device)
            moved = [0]

            def deliver_box(sidx):
                b = sidx % 2
                with torch.cuda.stream(comm_s):
                    comm_s.wait_event(free[b])
                    if rank == 0:
                        views[b].copy_(views[2 + sidx % (K - 2)], non_blocking=True)
                    moved[0] = sd.broadcast_box(views[b], Nj, Ni, box, buf, src=0)
                    ready[b].record(comm_s)

            def run_box(first, n):
                for sidx in range(first, first + n):
                    if sidx == first:
                        deliver_box(sidx)
                    comp.wait_event(ready[sidx % 2])
                    ctx.record_ptr(sidx % 2)
                    ctx.commit_record_box(sidx % 2, *box)
                    ctx.step(sidx % 2, sidx)
                    free[sidx % 2].record(comp)
                    if sidx + 1 < first + n:
                        deliver_box(sidx + 1)               # (one packing buffer: the next box is packed behind this record's unpack)

            run_box(s0, 4)
            ctx.sync(); torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            run_box(s0 + 4, nsteps)
            ctx.sync(); torch.cuda.synchronize(); dist.barrier()
            dt = time.perf_counter() - t0
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
            s0 += 4 + nsteps
            out["box_broadcast"] = {"ms_per_step": 1e3 * dt / nsteps, "slab_GBps_per_rank": moved[0] / (dt / nsteps) / 1e9,
                                    "particle_steps_per_s": ctx.nP * world * nsteps / dt, "box": list(box), "bytes_per_step": moved[0],
                                    "share_of_slab": moved[0] / float(out["slab_bytes"])}
        except Exception as e:                                      # noqa: BLE001
            out["box_broadcast"] = {"error": repr(e)}
        ctx.set_stream(None)                                        # (nothing steps with the two scratch slots after the segment)
    except Exception as e:                                          # noqa: BLE001
        out["error"] = repr(e)
        try:
            ctx.set_stream(None)
        except Exception:                                           # noqa: BLE001
            pass
    return out


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))

    import torch
    import sitrack_amd as sit
    from sitrack_amd import synthetic as syn

    # backend "nccl" = RCCL over xGMI.  SITRK_DIST_BACKEND=gloo (+ SITRK_DEVICE) rehearses the N>1 code path with several
    # ranks on ONE GPU (RCCL refuses two ranks per device): the slabs then travel through host memory.
    dist = None
    backend = os.environ.get("SITRK_DIST_BACKEND", "nccl")
    dev = int(os.environ.get("SITRK_DEVICE", str(local_rank)))
    force = world == 1 and os.environ.get("SITRK_FORCE_DIST") == "1"      # rehearsal of the N > 1 code path with one RCCL rank
    if force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    _STATE["rank"] = rank
    _STATE["base"] = {"metric": "particle-steps/s", "value": None, "unit": "particle-steps/s", "n_gpus": world, "steps": a.steps,
                      "warmup": a.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
                      "data": "synthetic"}
    if world > 1 or force:
        _STATE["multi"] = True
        _start_guards()
        _phase("init_process_group")
        import torch.distributed as dist
        torch.cuda.set_device(dev)
        import datetime
        tmo = datetime.timedelta(seconds=300)           # a stuck collective ends the run instead of hanging it
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev), timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    red_dev = "cuda" if backend == "nccl" else "cpu"        # where the small reduction tensors live

    # who takes part: every rank's (rank, local rank, device index, device UUID, host, pid), gathered through the process
    # group itself -- the line proves that N ranks on N different devices exchanged data, or says which did not
    rccl = None
    if dist is not None:
        _phase("gather identities")
        import socket
        me = dict(_device_identity(torch, dev), rank=rank, local_rank=local_rank, host=socket.gethostname(), pid=os.getpid())
        ids = [None] * dist.get_world_size()
        dist.all_gather_object(ids, me)
        uu = [(d.get("host"), d.get("uuid") or d.get("pci_bus_id") or d.get("device")) for d in ids]
        rccl = {"backend": "rccl (torch.distributed nccl)" if backend == "nccl" else backend, "world": dist.get_world_size(),
                "ranks": ids, "distinct_devices": len(set(uu))}

    if a.config == "auto":
        a.config = "c3" if world == 1 else "c4"
    K = a.records
    ctx = sit.Context(dev)
    _phase("workload")
    w = make_workload(a, syn, ctx, rank, a.config)
    Nj, Ni, nP, label, grid, yx, ji = w["Nj"], w["Ni"], w["nP"], w["label"], w["grid"], w["yx"], w["ji"]
    ctx.alloc_records(K, np.float32)
    if a.tune:
        ctx.set_tuning(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.tune.split(",")})

    from sitrack_amd import distributed as sd

    # records: generated on rank 0; every rank receives them by ONE RCCL broadcast per record, written
    # in place into its resident slot (the path's only exchange step).  No collective while stepping.
    u = v = sic = None
    slabs_host = None
    if rank == 0:
        u, v, sic = make_records(syn, w, K)
        slabs_host = [sd.pack_slab(u[k], v[k], sic[k], np.float32) for k in range(K)]
    records_via = "host upload" if dist is None else ("RCCL broadcast from rank 0" if backend == "nccl" else backend + " broadcast from rank 0")
    _STATE["base"]["config"] = {"workload": label, "grid": [Nj, Ni], "buoys_per_gpu": nP, "buoys_total": nP * world, "regime": a.regime,
                                "partition": "buoy-range x%d" % world, "records_via": records_via}
    if a.regime == "resident":
        _phase("record broadcast")
        try:
            for k in range(K):
                if dist is not None and os.environ.get("SITRK_BENCH_FAIL_BCAST") == str(rank) and k == K // 2:
                    raise RuntimeError("injected failure of the record broadcast (SITRK_BENCH_FAIL_BCAST)")     # tests only
                if dist is not None and backend == "nccl":
                    sd.broadcast_record(ctx, k, slabs_host[k] if rank == 0 else None, src=0)
                elif world > 1:
                    slab = sd.broadcast_record_host(slabs_host[k] if rank == 0 else None, ctx.slab_elems, np.float32, src=0)
                    ctx.push_record(k, *sd.split_slab(slab, Nj, Ni))
                else:
                    ctx.push_record(k, u[k], v[k], sic[k])
        except Exception as e:                          # noqa: BLE001
            if dist is None:
                raise
            # The exchange is what a multi-rank run is there to show: no local regeneration behind the driver's back.
            # Rank 0 prints the line with value null and the reason; every rank that got here leaves non-zero (the
            # launcher then terminates the others, whose SIGTERM watcher does the same).
            import traceback
            traceback.print_exc()
            _degraded_exit("record broadcast failed on rank %d: %r" % (rank, e), EXIT_DEGRADED)
        if dist is not None:
            # proof that every rank holds rank 0's bytes: exact checksums of the resident slabs, as they sit in HBM on each
            # rank, gathered and compared with rank 0's (and, on rank 0, with the host arrays they were generated as)
            _phase("slab checksums")
            ctx.sync(); torch.cuda.synchronize()
            probe = sorted({0, K // 2, K - 1})
            mine = torch.tensor([_slot_checksum(torch, sd, ctx, k) for k in probe], dtype=torch.int64, device=red_dev)
            allc = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
            dist.all_gather(allc, mine)
            allc = [[int(x) for x in t.cpu()] for t in allc]
            rccl["slab_checksum_slots"] = probe
            rccl["slab_checksums"] = allc
            rccl["slab_checksum_ok"] = all(c == allc[0] for c in allc)
            if rank == 0:
                host = [int(slabs_host[k].view(np.int32).astype(np.int64).sum()) for k in probe]
                rccl["slab_checksum_matches_source"] = host == allc[0]
            ctx.sync(); torch.cuda.synchronize()

    _phase("set_buoys")
    ctx.set_buoys(yx, ji, sort=not a.no_sort)
    resort = a.resort if a.resort >= 0 else 512        # measured over 6000 steps: 512 > 256 > 128 > never
    ctx.set_resort(0 if a.no_sort else resort)

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    per_record = None
    eight = None
    stats = None
    fresh = None
    solo = None
    nrun = 0
    if a.regime == "resident":
        fuse = max(1, min(a.fuse, K, 32))               # a launch advances distinct resident records only

        def timed_run(s0, nsteps, nfuse, sync=barrier):
            """nsteps records from step s0 at `nfuse` records per launch: (wall s, HIP-event ms, what was really launched)"""
            _phase("timed stepping, %d records per launch" % nfuse)
            ctx.set_tuning(fuse=nfuse)
            sync()
            ctx.launch_stats(reset=True)
            ctx.timer_start()
            t = time.perf_counter()
            ctx.run(s0 % K, s0, nsteps)
            ms = ctx.timer_stop()
            sync()
            return time.perf_counter() - t, ms, ctx.launch_stats(reset=True)

        def local_sync():
            ctx.sync()
            torch.cuda.synchronize()

        ctx.set_tuning(fuse=fuse)
        ctx.run(0, 0, a.warmup)
        nrun = a.warmup
        if dist is not None and not a.only_fused:
            # the same-shard N = 1 point, measured in THIS job: rank 0 steps alone while every other rank idles at the barrier
            # (what follows is then the ratio of two runs of one binary on one node, not of two workloads)
            _phase("solo leg (rank 0 alone)")
            barrier()
            if rank == 0:
                solo = timed_run(nrun, a.steps, fuse, sync=local_sync)
            barrier()
            if rank != 0:
                ctx.run(nrun % K, nrun, a.steps)       # every rank's buoys take the same steps (the check replays them)
            nrun += a.steps
        dt, ev_ms, stats = timed_run(nrun, a.steps, fuse)
        nrun += a.steps
        if fuse > 1 and not a.only_fused and not a.no_fresh:
            # right behind the headline leg (the box the buoys can touch grows as the cloud rotates: same state, same launches)
            _phase("fresh-records leg")
            fresh = fresh_records_leg(ctx, K, fuse, nrun, a.steps, barrier, overlap=a.fresh_overlap)
            nrun += a.steps
        if fuse > 1 and not a.only_fused:
            # for reference: the same K steps with one launch per record (the HBM-bound form of the kernel)
            dt1, ev1_ms, st1 = timed_run(nrun, a.steps, 1)
            per_record = (dt1, ev1_ms, st1)
            nrun += a.steps
        if fuse > 8 and not a.only_fused:
            # for reference: 8 records per launch (the K = 8 of SURVEY 8d's synthetic set-up allows no more)
            eight = timed_run(nrun, a.steps, 8)[0]
            nrun += a.steps
        ctx.set_tuning(fuse=fuse)
    else:
        # End-to-end: every step's record comes from host memory, double-buffered against the stepping (never `value`).
        #   --e2e-library : ordinary host arrays -> the library's pinned staging (gathered by a few host threads: the part a
        #                   NetCDF reader plays in the driver, which reads into the staging directly) -> its copy stream.  One GPU.
        #   default       : pre-pinned torch tensors on a torch copy stream (rank 0), + one RCCL broadcast per record at N > 1.
        # What travels: the BOX the buoys can touch (rows x columns, round 4; sitrk_buoy_box evaluated asynchronously every 16
        # records), row bands (--e2e-rows, rounds 1-3), or whole records (--e2e-full, and always at N > 1: a broadcast).
        fuse = 1
        assert K >= 2
        assert world == 1 or backend == "nccl", "the e2e regime broadcasts device slabs: needs RCCL"
        assert not (a.e2e_library and world > 1)
        whole = a.e2e_full or dist is not None
        st = {"eval": 0, "box": (0, Nj - 1, 0, Ni - 1), "bytes": 0, "pending": False, "begin": 0, "shape": [None, None]}
        if not a.e2e_library:
            comp, copy = torch.cuda.Stream(), torch.cuda.Stream()
            ctx.set_stream(comp.cuda_stream)
            slots = [sd.slot_tensor(ctx, 0).view(3, Nj, Ni), sd.slot_tensor(ctx, 1).view(3, Nj, Ni)]
            pinned = [torch.from_numpy(x).pin_memory().view(3, Nj, Ni) for x in slabs_host] if rank == 0 else None
            ready = [torch.cuda.Event(), torch.cuda.Event()]
            free = [torch.cuda.Event(), torch.cuda.Event()]
            for e in free:
                e.record(comp)

        def box_for(sidx):
            if whole:
                return 0, Nj, 0, Ni
            j0, j1, i0, i1 = ctx.box_of(*st["box"], sidx - st["eval"])
            return (j0, j1, 0, Ni) if a.e2e_rows else (j0, j1, i0, i1)

        def deliver(sidx):
            b, k = sidx % 2, sidx % K
            j0, j1, i0, i1 = box_for(sidx)
            st["shape"][b] = (j0, j1, i0, i1)
            st["bytes"] += 3 * (j1 - j0) * (i1 - i0) * 4
            if a.e2e_library:
                ctx.push_record_box(b, j0, j1, i0, i1, u[k][j0:j1, i0:i1], v[k][j0:j1, i0:i1], sic[k][j0:j1, i0:i1])
                return
            with torch.cuda.stream(copy):
                copy.wait_event(free[b])
                if rank == 0:
                    if i0 == 0 and i1 == Ni:
                        for f in range(3):                           # three contiguous row ranges
                            slots[b][f, j0:j1].copy_(pinned[k][f, j0:j1], non_blocking=True)
                    else:
                        for f in range(3):                           # strided DMA straight out of the pinned whole fields
                            _memcpy2d_h2d(slots[b][f, j0, i0:].data_ptr(), Ni * 4, pinned[k][f, j0, i0:].data_ptr(), Ni * 4,
                                          (i1 - i0) * 4, j1 - j0, copy.cuda_stream)
                if dist is not None:
                    dist.broadcast(slots[b], src=0)
                ready[b].record(copy)

        def run_e2e(s0, n):
            for sidx in range(s0, s0 + n):
                if not whole:
                    if st["pending"] and sidx - st["begin"] >= 8:
                        r = ctx.buoy_box_end()                       # queued 8 records ago: no wait to speak of
                        st["box"], st["eval"], st["pending"] = r[:4], st["begin"], False
                    if not st["pending"] and sidx - st["eval"] >= 16:
                        ctx.buoy_box_begin()
                        st["pending"], st["begin"] = True, sidx
                if sidx == s0:
                    deliver(sidx)
                if a.e2e_library:
                    ctx.step(sidx % 2, sidx)
                    if sidx + 1 < s0 + n:
                        deliver(sidx + 1)          # its DMA waits for the step above only if it reuses that slot: it does not
                else:
                    if sidx + 1 < s0 + n:
                        deliver(sidx + 1)
                    comp.wait_event(ready[sidx % 2])
                    ctx.record_ptr(sidx % 2)                         # written in place: the Survive bytes of the box just written
                    ctx.commit_record_box(sidx % 2, *st["shape"][sidx % 2])
                    ctx.step(sidx % 2, sidx)
                    free[sidx % 2].record(comp)
                if sidx % 32 == 0:
                    _phase("e2e stepping")                           # (keeps the no-progress watchdog of multi-rank runs quiet)

        if not whole:
            st["box"], st["eval"] = ctx.buoy_box(), 0
        run_e2e(0, a.warmup)
        barrier()
        st["bytes"] = 0
        ctx.timer_start()
        t0 = time.perf_counter()
        run_e2e(a.warmup, a.steps)
        ev_ms = ctx.timer_stop()
        barrier()
        dt = time.perf_counter() - t0
        if st["pending"]:
            ctx.buoy_box_end()
        if not a.e2e_library:
            ctx.set_stream(None)
        e2e_bytes_per_step = st["bytes"] / float(a.steps)
        nrun = a.warmup + a.steps
    value_per_rank = None
    fresh_dt = fresh["dt"] if fresh else 0.0
    if dist is not None:                                 # the slowest rank sets every reported time
        _phase("reductions")
        mine = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        each = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(each, mine)
        per = [float(nP) * a.steps / float(x[0]) for x in each]          # every rank's own particle-steps/s over the timed region
        value_per_rank = {"min": min(per), "max": max(per), "all": per}
        extra = [per_record[0], per_record[1]] if per_record is not None else [0.0, 0.0]
        t = torch.tensor([dt, ev_ms] + extra + [eight or 0.0, fresh_dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, ev_ms = float(t[0]), float(t[1])
        if per_record is not None:
            per_record = (float(t[2]), float(t[3]), per_record[2])
        if eight is not None:
            eight = float(t[4])
        fresh_dt = float(t[5])

    _phase("count alive")
    nalive = ctx.count_alive()
    if dist is not None:
        t = torch.tensor([nalive], dtype=torch.int64, device=red_dev)
        dist.all_reduce(t)
        nalive = int(t[0])

    if a.check and rank == 0:
        cpu_baseline_check(ctx, grid, u, v, sic, yx, ji, nrun, a.uv_strategy)

    # ---- N = 1 extras on the same context, after everything `value` and the check need (they move the buoys on)
    e2e_upload = None
    c4_shard = None
    plain = world == 1 and dist is None and a.regime == "resident" and not a.only_fused
    if plain and not a.no_e2e_upload and K >= 2:
        _phase("e2e upload segment")
        e2e_upload, nrun = e2e_upload_segment(ctx, w, u, v, sic, K, nrun)
        for k in (0, 1):
            ctx.push_record(k, u[k], v[k], sic[k])           # the two slots the segment used hold whole records again
    if plain and a.config == "c3" and a.buoys == 0 and not a.no_c4_shard:
        _phase("c4 shard sub-run")
        c4_shard = c4_shard_subrun(a, syn, ctx, w, K, fuse, nrun, barrier)

    def measured_copy_GBps():
        """practical HBM ceiling of this GPU, measured now: a device-to-device copy of 1 GiB (read + write bytes / time)"""
        try:
            n = 1 << 28
            x = torch.empty(n, dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
            y.copy_(x); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                y.copy_(x)
            e1.record(); torch.cuda.synchronize()
            return 5 * 2 * 4.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9
        except Exception:                                          # noqa: BLE001
            return None

    copy_GBps = measured_copy_GBps() if (rank == 0 and per_record is not None) else None

    e2e_bcast = None
    e2e_hung = False
    if dist is not None and a.regime == "resident" and backend == "nccl" and not a.no_e2e_broadcast:
        # Everything `value` needs is measured and reduced by now.  The extra segment runs under a watchdog: if a collective
        # in it never completes on some rank, every rank gives up after the same wait, rank 0 still prints the line (with
        # the segment marked as timed out) and the processes leave without touching the stuck communicator.
        box = {}

        def _segment():
            torch.cuda.set_device(dev)
            box["out"] = e2e_broadcast_segment(ctx, dist, torch, sd, rank, world, slabs_host, K, nrun, Nj, Ni)

        _phase("e2e broadcast segment")
        th = threading.Thread(target=_segment, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("SITRK_E2E_SEGMENT_TIMEOUT", "150")))
        if th.is_alive():
            e2e_hung = True
            e2e_bcast = {"error": "timed out (a collective of the segment did not complete); the resident numbers were measured and "
                                  "reduced before the segment started; exit code %d" % EXIT_E2E_FAILED}
        else:
            e2e_bcast = box.get("out", {"error": "segment thread ended without a result"})

    _phase("line")
    if rank == 0:
        total = float(nP) * world * a.steps
        step_s = (ev_ms / 1e3) / a.steps                     # avg time per record, HIP events, library stream
        # cells whose record a step needs: the union of every buoy's (j,i),(j,i-1),(j-1,i),(j-1,i-1)
        keys = ji[:, 0].astype(np.int64) * Ni + ji[:, 1]
        need = np.zeros(Nj * Ni, dtype=bool)
        for off in (0, 1, Ni, Ni + 1):
            need[keys - off] = True
        n_cells = int(need.sum())
        A1 = 50.0 * nP + 56.0 * n_cells                      # algorithmic bytes of ONE record per GPU
        A_survey = 50.0 * nP + 56.0 * Nj * Ni                # SURVEY 8d closed form (every cell of the grid)
        prof = {}
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        except Exception:                                    # noqa: BLE001
            prof = {}
        prof_key = "c3" if a.config == "c4" else a.config    # C4's per-rank shard runs the C3 kernels on the C3 grid
        if a.warp and (prof_key + "warp_fused") in prof:     # a counter profile of the curvilinear variant, when there is one
            prof_key += "warp"
        prof_step, prof_fused = prof.get(prof_key, {}), prof.get(prof_key + "_fused", {})
        nwaves = (nP + 63) // 64

        def roofline_step(ms_per_launch):
            """advect_step_kernel: one record per launch, HBM bound (SURVEY 8d's byte model, needed cells only)"""
            s_ = ms_per_launch / 1e3
            tr = prof_step.get("hbm_bytes_per_launch") if (a.buoys == 0 and a.uv_strategy == 1) else None
            # `achieved` / `frac` are algorithmic bytes over the time measured in THIS run; only `traffic` comes from a counter profile,
            # and is marked when that profile is of another build of the kernel than the one that ran
            return {"bound": "hbm", "achieved": A1 / s_ / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": A1 / s_ / 1e9 / HBM_PEAK_GBS,
                    "traffic": tr, "traffic_source": prof_step.get("source") if tr else None,
                    "traffic_stale": roofline_is_stale(prof_step.get("isa"), isa_fingerprint("advect_step_kernel<float,1,false,512>")) if tr else None,
                    "algorithmic_bytes_per_launch": A1, "cells_needed": n_cells, "kernel": "advect_step_kernel",
                    "records_per_launch": 1, "avg_launch_ms": ms_per_launch,
                    "survey_formula_bytes_per_record": A_survey, "survey_formula_frac": A_survey / s_ / 1e9 / HBM_PEAK_GBS}

        def roofline_fused(ev_ms_total, st):
            """advect_run_kernel: state and geometry are read once per launch, so it is bound by fp64 VALU issue, not by
            HBM.  achieved = VALU instructions issued per second: (fixed per launch + per record x records) per wave, both
            terms fitted from committed rocprofv3 counters of launches of different lengths (tools/fit_valu_terms.py ->
            profiles/traffic.json), for the launches and records COUNTED BY THE LIBRARY in this run.  The ceiling weights
            the instruction mix: a wave64 instruction of the 64-bit classes holds its SIMD 4 cycles, a 32-bit one 2
            (share of the 64-bit classes from the per-class counters of the same profile; static ISA count: 0.83,
            tools/isa_valu_classes.py).  frac is quoted against the 2.4 GHz maximum clock and against the clock the chip
            held under this kernel in the profiled run."""
            nl, nr, ns = st["fused_launches"], st["fused_records"], st["step_launches"]
            rpl = nr / nl if nl else 0.0
            fixed, per = prof_fused.get("valu_per_wave_fixed"), prof_fused.get("valu_per_wave_per_record")
            if fixed is not None and per is not None and rpl:
                ipwr = (fixed + per * rpl) / rpl                      # for THIS run's launch length
                ipwr_src = "%.1f fixed per launch + %.2f per record, %s" % (fixed, per, prof_fused.get("valu_terms_source"))
            else:
                ipwr = prof_fused.get("valu_per_wave_record")         # one launch length only (older profiles)
                ipwr_src = prof_fused.get("source")
            # the timed region = nl fused launches (+ ns one-record launches where a re-sort or the tail cut a launch short)
            launch_ms = ev_ms_total / max(nl + ns, 1)
            rec_s = (ev_ms_total / 1e3) / max(nr + ns, 1)
            A = 50.0 * nP + 48.0 * n_cells + 8.0 * n_cells * rpl          # bytes one launch of rpl records needs
            w64 = prof_fused.get("valu64_frac")
            mb = prof.get("valu_issue_microbench", {})
            c64, c32 = mb.get("cycles_64bit", 4.0), mb.get("cycles_32bit", 2.0)       # measured issue cost per wave64 instruction per SIMD
            cpi = c64 * w64 + c32 * (1.0 - w64) if w64 is not None else 4.0
            peak = 1024 * 2.4 / cpi
            held = prof_fused.get("sclk_ghz")
            shipped, profiled = isa_fingerprint(), prof_fused.get("isa")
            stale = roofline_is_stale(profiled, shipped)
            out = {"bound": "valu_issue+wave_chain", "unit": "Ginst/s", "peak": peak,
                   "bound_note": "the loop sits where two bounds meet (DESIGN 3.2 item 35): the vector pipe is ~88 % busy AND seven resident "
                                 "waves each need ~6 500 cycles per record (dependent chain); removing 38 % of the vector instructions "
                                 "bought 6 % -- `frac` is the share of an instruction-COUNT ceiling, not the distance to a hard limit",
                   "kernel": "advect_run_kernel", "launches": nl, "records_advanced": nr, "one_record_launches": ns,
                   "records_per_launch": rpl, "avg_launch_ms": launch_ms, "waves_per_launch": nwaves,
                   "valu_inst_per_wave_record": ipwr, "valu_inst_source": ipwr_src,
                   "valu64_class_share": w64, "cycles_per_valu_inst": cpi,
                   "cycles_64bit_class": c64, "cycles_32bit_class": c32,
                   "peak_note": "a MODEL built on measured issue costs: 1024 SIMDs x 2.4 GHz / (%.2f cycles x share of 64-bit-class VALU "
                                "instructions + %.2f cycles x the rest); peak_uniform_4_cycles prices every wave64 VALU instruction at 4 cycles.  %s"
                                % (c64, c32, mb.get("summary", "issue costs not pinned by a microbenchmark: nominal 4 / 2 cycles")),
                   "peak_source": mb.get("source"),
                   "peak_uniform_4_cycles": VALU_PEAK_GINST,
                   "stale": stale, "isa_shipped": shipped, "isa_profiled": profiled}
            # share of the issued lanes that do work: the main path runs with all 64 lanes, the crossing path -- every wave, every
            # record -- with the lanes whose buoy left its cell (p_cross of them)
            main_i, cross_i = prof_fused.get("valu_main_path_per_wave_record"), prof_fused.get("valu_crossing_path_per_wave_record")
            if main_i and cross_i and p_cross is not None:
                out["lane_utilisation"] = (main_i + cross_i * p_cross) / (main_i + cross_i)
                out["lane_utilisation_note"] = ("(%.0f main-path instructions x 64 lanes + %.0f crossing-path instructions x %.1f lanes) / "
                                                "(all instructions x 64): %.3f of the buoy-records leave their cell (%s)"
                                                % (main_i, cross_i, 64 * p_cross, p_cross, p_cross_src))
            died = 1.0 - nalive / float(nP * world)
            if ipwr and a.uv_strategy != 1:
                # the counter passes are of advect_run_kernel<float, 1, false> (the reference's default rule): another instantiation ran
                out.update({"achieved": None, "frac": None, "frac_note": "iUVstrategy = %d runs another instantiation of the kernel than the "
                            "profiled one (iUVstrategy = 1): its instruction count per wave and record is not known, no fraction is claimed"
                            % a.uv_strategy})
            elif ipwr and died > 0.02:
                # dead buoys' lanes (and whole waves of them) stop issuing: the per-wave instruction count of the profiled run no
                # longer describes this one -- no fraction is claimed
                out.update({"achieved": None, "frac": None, "frac_note": "%.0f %% of the buoys died during the run: the instruction count per "
                            "wave and record of the counter profile (%s buoys alive throughout) does not describe it" % (100 * died, "nearly all")})
            elif ipwr:
                ach = ipwr * nwaves / rec_s / 1e9
                out.update({"achieved": ach, "frac": None if stale else ach / peak,
                            "frac_uniform_4_cycles": None if stale else ach / VALU_PEAK_GINST,
                            "clock_held_ghz": held, "clock_held_source": prof_fused.get("source"),
                            "frac_at_held_clock": (ach / (1024 * held / cpi)) if (held and not stale) else None})
                if stale:
                    out["stale_note"] = ("the rocprofv3 counter passes behind valu_inst_per_wave_record / valu64_class_share / clock_held_ghz "
                                         "(profiles/traffic.json) describe another build of advect_run_kernel than the one that ran: frac is "
                                         "withheld; `achieved` uses the OLD instruction count and is indicative only")
            else:
                out.update({"achieved": None, "frac": None})
            tr = prof_fused.get("hbm_bytes_per_launch") if a.buoys == 0 else None
            tr_rpl = prof_fused.get("records_per_launch", 32)
            out["traffic"] = tr * rpl / tr_rpl if (tr and rpl) else None
            out["traffic_note"] = ("rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE of a %d-record launch (%s), scaled to %.1f records"
                                   % (tr_rpl, prof_fused.get("source"), rpl)) if tr else None
            s_launch = launch_ms / 1e3
            out["hbm"] = {"algorithmic_bytes_per_launch": A, "achieved": A / s_launch / 1e9 if nl else None,
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": A / s_launch / 1e9 / HBM_PEAK_GBS if nl else None,
                          "cells_needed": n_cells,
                          "survey_formula_bytes_per_launch": A_survey * rpl,
                          "survey_formula_frac": A_survey * rpl / s_launch / 1e9 / HBM_PEAK_GBS if nl else None,
                          "note": "secondary: the fused kernel is not HBM bound; per_record_launch holds the HBM-bound form.  "
                                  "survey_formula_frac charges SURVEY 8d's per-step bytes (50 B per buoy + 56 B for EVERY cell of the "
                                  "grid) to every record of the launch and comes out ABOVE 1: the launch reads state and geometry once "
                                  "for all its records (loop interchange), and only the cells that host buoys (cells_needed) are read"}
            return out

        # crossing rate of THIS workload (share of the buoy-records that leave their cell): counted by the CPU oracle on its
        # baseline sample when that leg runs, else the figure stored with the profile
        cpu_base = None
        if world == 1 and not a.no_cpu_baseline:
            _phase("cpu baseline")
            cpu_base = cpu_baseline(grid, u, v, sic, yx, ji, a.cpu_seconds, a.uv_strategy)
        p_cross, p_cross_src = prof_fused.get("crossing_rate"), "profiles/traffic.json"
        if cpu_base is not None and cpu_base.get("crossing_rate") is not None:
            p_cross, p_cross_src = cpu_base["crossing_rate"], "counted by the oracle on the cpu_baseline sample of this run"
        if a.regime == "resident" and fuse > 1 and stats["fused_launches"] > 0:
            roof = roofline_fused(ev_ms, stats)
        else:
            roof = roofline_step(1e3 * step_s)
        amort = (a.steps + a.warmup) / float(K) if a.regime == "resident" else None
        line = {
            "metric": "particle-steps/s", "value": total / dt, "unit": "particle-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": label, "grid": [Nj, Ni], "buoys_per_gpu": nP, "buoys_total": nP * world,
                       "warp": float(grid["warp"]), "seeding": w["seeding"],
                       "records_resident": K, "record_dtype": "f32", "uv_strategy": a.uv_strategy,
                       "sorted": not a.no_sort, "resort_every": 0 if a.no_sort else resort, "regime": a.regime,
                       "records_per_launch": fuse,
                       "e2e_upload_bytes_per_step": (e2e_bytes_per_step if a.regime == "e2e" else None),
                       "partition": "buoy-range x%d" % world, "records_via": records_via, "alive_after": nalive,
                       "tune": a.tune or None},
            "roofline": roof,
        }
        if a.regime == "resident":
            line["note"] = ("the %d resident records are cycled: each record's Survive bytes (sitrack/tracking.py:62-93 once per cell) were "
                            "derived once at ingest and reused ~%.1f times OUTSIDE the timed region; fresh_records.value has every record "
                            "committed once, inside the clock" % (K, amort))
        if fresh is not None:
            fs = fresh["stats"]
            line["fresh_records"] = {
                "value": total / fresh_dt, "unit": "particle-steps/s", "ms_per_step": 1e3 * fresh_dt / a.steps,
                "event_ms_per_step": fresh["event_ms"] / a.steps, "survive_us_per_record": fresh["survive_us_per_record"],
                "survive_box_cells": fresh["box_cells_mean"], "survive_box_share_of_grid": fresh["box_cells_mean"] / float(Nj * Ni),
                "launches": fs["fused_launches"], "records_advanced": fs["fused_records"] + fs["step_launches"],
                "records_per_launch": fresh["records_per_launch"], "commits_overlap_the_stepping": fresh["overlap"],
                "note": "same steps, same launches; before each launch its records are committed afresh (sitrk_commit_records_box: "
                        "Survive re-derived from siconc over the box the buoys can touch, ONE Survive launch per fused launch; the box "
                        "from sitrk_buoy_box_begin/_end queued one launch ahead), all inside the timed region"}
            line["value_fresh_records"] = line["fresh_records"]["value"]
            line["survive_us_per_record"] = fresh["survive_us_per_record"]
        if per_record is not None:
            dt1, ev1, st1 = per_record
            line["per_record_launch"] = {
                "value": total / dt1, "ms_per_step": 1e3 * dt1 / a.steps, "kernel": "advect_step_kernel",
                "launches": st1["step_launches"], "roofline": roofline_step(ev1 / max(st1["step_launches"], 1))}
            cp = copy_GBps
            if cp:
                rr = line["per_record_launch"]["roofline"]
                rr["device_copy_GBps_measured"] = cp                 # the practical ceiling next to the 8 TB/s spec
                rr["frac_of_measured_copy"] = rr["achieved"] / cp
        if eight is not None:
            line["eight_records_per_launch"] = {"value": total / eight, "ms_per_step": 1e3 * eight / a.steps,
                                                "note": "same kernel, 8 records per launch (SURVEY 8d keeps K = 8 records resident)"}
        if solo is not None:
            sdt, sms, sst = solo
            line["solo_same_shard"] = {"value": float(nP) * a.steps / sdt, "ms_per_step": 1e3 * sdt / a.steps, "event_ms_per_step": sms / a.steps,
                                       "launches": sst["fused_launches"] + sst["step_launches"],
                                       "note": "rank 0 stepping its shard ALONE (every other rank idle at a barrier), same job, same binary: the "
                                               "N = 1 point of this very workload"}
            line["efficiency_vs_n1_same_shard"] = (total / dt) / (world * line["solo_same_shard"]["value"])
        if c4_shard is not None:
            line["c4_shard"] = c4_shard
        if e2e_upload is not None:
            line["e2e_upload"] = e2e_upload
        if rccl is not None:
            rccl["value_per_rank"] = value_per_rank
            line["rccl"] = rccl
        if e2e_bcast is not None:
            line["e2e_broadcast"] = e2e_bcast
        # (nothing below touches the GPU when the segment's thread is stuck inside a collective)
        if world == 1 and not e2e_hung and a.regime == "resident" and a.config == "c3" and a.buoys == 0 and not a.no_c2 and not a.warp:
            _phase("c2 sub-run")
            line["c2"] = c2_subrun(sit, syn, dev, a)
        if cpu_base is not None:
            line["cpu_baseline"] = cpu_base
        if rccl is not None and rccl.get("slab_checksum_ok") is False:
            line["degraded"] = "the resident slabs differ between ranks (slab_checksums)"
        _emit(line)
    e2e_failed = e2e_bcast is not None and "error" in e2e_bcast
    if e2e_hung:
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(EXIT_E2E_FAILED)                       # a thread is stuck inside a collective: no orderly teardown possible
    _phase("teardown")
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    _phase("done")
    if e2e_failed:
        sys.exit(EXIT_E2E_FAILED)                       # a collective of the segment raised: the line says so, the code too
    if rccl is not None and rccl.get("slab_checksum_ok") is False:
        sys.exit(EXIT_DEGRADED)


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:                          # noqa: BLE001
        if not _STATE["multi"]:
            raise
        import traceback
        traceback.print_exc()
        _degraded_exit("exception on rank %d: %r" % (_STATE["rank"], e), EXIT_EXCEPTION)
