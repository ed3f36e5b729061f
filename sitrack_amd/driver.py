"""Command-line driver: the reference's `si3_part_tracker.py` surface on top of libsitrk.

Same flags and defaults (reference si3_part_tracker.py:42-73), same module constants
(`rdt = 3600`, `iUVstrategy = 1`, :31,37), same file-name logic (:115-148,510-517,565-568), same
`./seed/Initialized_buoys_<SeedName>_<CONF>.npz` cache keys (:205-255), same per-buoy record windows in
2-D-time mode (:264-318), same NetCDF outputs (:515-571) -- the per-buoy loop (:378-490) and the
per-record inverse projection (:493) run on the GPU.  Differences, all opt-in or forced:
extra flags `--device`, `--uv-strategy`; under `torchrun` (WORLD_SIZE > 1) the buoys are range-partitioned over the
ranks by latitude band and every rank reads only the rows of each record its own buoys can touch (row-band ingest, no
collective; `--full-records`: rank 0 reads, RCCL broadcast), rank 0 writes the files; errors raise instead of
`print; exit(0)`; maps need the
optional `mojito` package and are skipped without it; the full (Nt+1,nP,2) series is NEVER held in host memory (the
reference keeps it twice, si3_part_tracker.py:324-330: 320 GB at 1e7 buoys x 1000 records): with `-F` every record is
appended to the series file as it is fetched (ncio.CloudBuoysStream: O(nP) memory), `--out-stride K` writes every K-th
record only, and the always-written 2-record file keeps records 0 and Nt.
"""
import argparse
import os
from datetime import datetime, timezone
from os import path

import numpy as np

from . import _lib, ncio
from .distributed import Comm, all_ranges
from .tracking import GetTimeSpan, IceTracker, SeedInit

rdt = 3600.          # time step [s] = model output period (reference :31)
FILL = ncio.FillValue


def epoch2clock(it):
    """reference util.py:18-34 (precision 's')"""
    return datetime.fromtimestamp(int(it), timezone.utc).strftime("%Y-%m-%d_%H:%M:%S")


def clock2epoch(cdate):
    """'YYYY-MM-DD_hh:mm:ss' (reference util.py:36-40) or a bare day 'YYYY-MM-DD' / 'YYYYMMDD' (mojito's 'guess')."""
    for fmt in ("%Y-%m-%d_%H:%M:%S", "%Y-%m-%d", "%Y%m%d"):
        try:
            return int(datetime.strptime(cdate, fmt).replace(tzinfo=timezone.utc).timestamp())
        except ValueError:
            pass
    raise ValueError("cannot parse date '%s'" % cdate)


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description='SITRACK ICE PARTICULES TRACKER (MI355X build)')
    rq = ap.add_argument_group('required arguments')
    rq.add_argument('-i', '--fsi3', required=True, help='output file of SI3 containing ice velocities ans co')
    rq.add_argument('-m', '--fmmm', required=True, help='model `mesh_mask` file of NEMO config used in SI3 run')
    rq.add_argument('-s', '--fsdg', required=True, help='seeding file')
    ap.add_argument('-k', '--krec', type=int, default=0, help='record of seeding file to use to seed from')
    ap.add_argument('-e', '--dend', default=None, help='date at which to stop')
    ap.add_argument('-F', '--fxdt', action="store_true", help='fixed tracking time (1D time array)')
    ap.add_argument('-N', '--ncnf', default='NANUK4', help='name of the horizontak NEMO config used')
    ap.add_argument('-p', '--plot', type=int, default=0, help='how often, in terms of model records, we plot the positions on a map')
    ap.add_argument('--device', type=int, default=0, help='GPU to use (extra)')
    ap.add_argument('--uv-strategy', type=int, default=1, choices=(0, 1, 2),
                    help='iUVstrategy of the reference (0 cell mean, 1 nearest U/V point = default); 2 = linear interpolation, an extra the reference does not have')
    ap.add_argument('--slots', type=int, default=32,
                    help='model records resident on the GPU (extra): up to half of them are advanced by one fused launch '
                         'whenever no per-record output is due, while the next ones are read and uploaded')
    ap.add_argument('--rebalance', type=int, default=256,
                    help='under torchrun: every so many records the buoys are re-partitioned over the ranks by their CURRENT host '
                         'row (migration of the states between ranks), so that every rank keeps a compact latitude band to read '
                         '(extra; 0 = never; same results)')
    ap.add_argument('--out-stride', type=int, default=1,
                    help='with -F: write every so many-th record of the series only (records 0, K, 2K, ...; extra, default 1 = the '
                         'reference\'s full series); the records in between go through fused launches of up to K records')
    ap.add_argument('--full-records', action='store_true',
                    help='read and upload whole records (under torchrun: rank 0 reads, RCCL broadcast) instead of only the rows '
                         'each rank\'s buoys can touch (extra; same results)')
    return ap.parse_args(argv)


_IDEALISED_KINDS = ('nemoTsi3', 'nemoTmm', 'sidfex')


def seed_name_tokens(seed_file_name):
    """The two tags the output names inherit from the seeding file's NAME (behaviour of reference
    si3_part_tracker.py:115-148): the time-bin tag and the resolution tag, each with its leading underscore.

    * idealised seeding (`sitrack_seeding_<kind>_...`, kind = nemoTsi3 / nemoTmm / sidfex): time-bin tag `_idlSeed`;
      the resolution tag is the last of the final three `_`-separated tokens that ends in `km`, if any;
    * otherwise (RGPS-style selections, `..._dt72_..._10km.nc`): the time-bin token `dt<h>` sits three places before the
      resolution token `<n>km`, which is among the last four tokens; a name without resolution token has its `dt<h>`
      third from the end and yields an empty resolution tag.  Anything else is an error, like in the reference."""
    head = seed_file_name.split('.')[0].split('_')
    if len(head) > 2 and head[2] in _IDEALISED_KINDS:
        km = next((tok for tok in reversed(head[-3:]) if tok.endswith('km')), None)
        return '_idlSeed', ('_' + km) if km else ''
    toks = seed_file_name.split('.')[-2].split('_')
    if toks[-3].startswith('dt'):                      # no resolution token at the end
        return '_' + toks[-3], ''
    for back in range(1, 5):
        res, dtbin = toks[-back], toks[-back - 3]
        if res.endswith('km') and dtbin.startswith('dt'):
            return '_' + dtbin, '_' + res
    raise ValueError('could not figure out the resolution and time-bin tags from the file name %s' % seed_file_name)


def date_tag(it):
    """'1996-12-15_00:00:00' -> '19961215h00' (reference :510-512)"""
    c = epoch2clock(it).split(':')[0]
    return c.replace('-', '').replace('_', 'h')


def _savez_deflate(fname, **arrays):
    """The `.npz` the reference writes with `np.savez_compressed` (:255) -- same members, same `np.load` -- deflated at
    level 1 in 4-MB pieces on a thread pool (independent raw-deflate blocks closed with a sync flush concatenate into one
    valid deflate stream, the way `pigz -i` does it): the cache of 10^7 seeds (1.3 GB of arrays) took 36 s of zlib at
    numpy's level 6, 9 s at level 1 on one thread.  Written as ZIP64 throughout (members of 10^8 seeds exceed 4 GB)."""
    import io
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    from numpy.lib import format as npfmt
    PIECE = 4 << 20

    def deflate(args):
        piece, last = args
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        return co.compress(piece) + co.flush(zlib.Z_FINISH if last else zlib.Z_SYNC_FLUSH)

    central = []
    with open(fname, 'wb') as f, ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:
        for key, val in arrays.items():
            arr = np.asanyarray(val)
            if not arr.flags.c_contiguous:
                arr = np.ascontiguousarray(arr)                # (never for a 0-d array: it would become 1-d)
            hdr = io.BytesIO()
            npfmt.write_array_header_1_0(hdr, npfmt.header_data_from_array_1_0(arr))        # the .npy header, then the data
            head = hdr.getvalue()
            body = memoryview(arr.reshape(-1).view(np.uint8)) if arr.size else memoryview(b'')
            name = (key + '.npy').encode()
            n = len(head) + len(body)
            pieces = [(head, len(body) == 0)] + [(body[o:o + PIECE], o + PIECE >= len(body)) for o in range(0, len(body), PIECE)]
            crc_job = ex.submit(lambda: zlib.crc32(body, zlib.crc32(head)))
            blobs = list(ex.map(deflate, pieces))
            crc, csize, off = crc_job.result() & 0xffffffff, sum(len(b) for b in blobs), f.tell()
            extra = struct.pack('<HHQQ', 1, 16, n, csize)                       # zip64: sizes
            f.write(struct.pack('<IHHHHHIIIHH', 0x04034b50, 45, 0, 8, 0, 0x21, crc, 0xffffffff, 0xffffffff, len(name), len(extra)))
            f.write(name); f.write(extra)
            for b in blobs:
                f.write(b)
            central.append((name, crc, csize, n, off))
        cd_off = f.tell()
        for name, crc, csize, n, off in central:
            extra = struct.pack('<HHQQQ', 1, 24, n, csize, off)
            f.write(struct.pack('<IHHHHHHIIIHHHHHII', 0x02014b50, 45, 45, 0, 8, 0, 0x21, crc, 0xffffffff, 0xffffffff, len(name),
                                len(extra), 0, 0, 0, 0, 0xffffffff))
            f.write(name); f.write(extra)
        cd_size = f.tell() - cd_off
        z64 = f.tell()
        f.write(struct.pack('<IQHHIIQQQQ', 0x06064b50, 44, 45, 45, 0, 0, len(central), len(central), cd_size, cd_off))
        f.write(struct.pack('<IIQI', 0x07064b50, 0, z64, 1))
        f.write(struct.pack('<IHHHHIIH', 0x06054b50, 0, 0, 0xffff, 0xffff, 0xffffffff, 0xffffffff, 0))


def record_windows(zTpos, ztime_model, kstrt, kstop, iTmA, iTmB, nP):
    """Per-buoy first / last model record in 2-D-time mode (reference :264-312).

    The reference loops over the late starters / early stoppers with one `np.where` over the model time axis each
    (`idx[-1]+1`, `idx[0]-1`).  On an increasing time axis those are counts, i.e. two `searchsorted` for all buoys at
    once (10^7 buoys: 5 s -> 0.1 s); any other axis takes the reference's loop.  Same IndexError when the search
    comes back empty."""
    z1st = np.zeros(nP, dtype=int) + kstrt
    zLst = np.zeros(nP, dtype=int) + kstop
    half = int(rdt / 2)
    tm = np.asarray(ztime_model)
    # (entries the seeding file holds as _FillValue arrive masked, like netCDF4 hands them to the reference: a masked
    # comparison is no match for np.where, so such a buoy keeps the full window)
    late = np.where(np.ma.filled(zTpos[0, :] >= iTmA + half, False))[0]
    early = np.where(np.ma.filled(zTpos[1, :] < iTmB - half, False))[0]
    if tm.ndim == 1 and (tm.size < 2 or np.all(np.diff(tm) > 0)):
        if late.size:
            cnt = np.searchsorted(tm + half, zTpos[0, late], side='left')        # how many tm+half < T
            if np.any(cnt == 0):
                raise IndexError("index -1 is out of bounds for axis 0 with size 0")
            z1st[late] = cnt
        if early.size:
            cnt = np.searchsorted(tm - half, zTpos[1, early], side='right')      # how many tm-half <= T
            if np.any(cnt == tm.size):
                raise IndexError("index 0 is out of bounds for axis 0 with size 0")
            zLst[early] = cnt - 1
        return z1st, zLst
    return _record_windows_loop(zTpos, tm, z1st, zLst, late, early, half)


def _record_windows_loop(zTpos, tm, z1st, zLst, late, early, half):
    """the reference's own form (:289-312), one search per buoy"""
    for jb in late:
        (idx,) = np.where(tm + half < zTpos[0, jb])
        z1st[jb] = idx[-1] + 1
    for jb in early:
        (idx,) = np.where(tm - half > zTpos[1, jb])
        zLst[jb] = idx[0] - 1
    return z1st, zLst


class _Clock:
    """wall-clock shares of a run (where the command line spends its time: reading, the GPU, writing)"""

    def __init__(self):
        import time
        self.now = time.perf_counter
        self.t = {}
        self.t0 = self.now()

    def add(self, key, since):
        self.t[key] = self.t.get(key, 0.0) + (self.now() - since)
        return self.now()


def _tame_malloc():
    """The command line's host memory is O(nP) by design (the series is streamed); glibc would still keep hundreds of MB of
    freed record-sized buffers in per-thread arenas (the output is deflated on up to 16 threads, each allocating and freeing
    4-MB pieces): allocations of 1 MB and more go straight to mmap (returned to the system when freed, no dynamic threshold)
    and the arenas are capped.  Best effort, Linux / glibc only; the library itself never touches the allocator."""
    try:
        import ctypes
        libc = ctypes.CDLL(None)
        libc.mallopt(-3, 1 << 20)        # M_MMAP_THRESHOLD
        libc.mallopt(-8, 4)              # M_ARENA_MAX
    except Exception:                    # noqa: BLE001
        pass


def main(argv=None):
    a = parse_args(argv)
    _tame_malloc()
    clk = _Clock()
    cf_uv, cf_mm, fNCseed, jrecSeed, cdate_stop, CONF = a.fsi3, a.fmmm, a.fsdg, a.krec, a.dend, a.ncnf
    lUse2DTime = not a.fxdt
    iUVstrategy = a.uv_strategy
    comm = Comm()
    say = print if comm.root else (lambda *args, **kw: None)
    say('\n *** SITRACK ice particule tracker, GPU build; NetCDF backend = ' + ncio.backend()
        + ('; %d ranks (%s)' % (comm.world, comm.backend) if comm.multi else ''))
    say(' *** SI3 file =>', cf_uv, '\n *** mesh_mask =>', cf_mm, '\n *** seeding  =>', fNCseed, jrecSeed)

    cdtbin, csfkm = seed_name_tokens(path.basename(fNCseed))
    idateSeedA, idateSeedB, SeedName, SeedBatch, zTpos = ncio.SeedFileTimeInfo(fNCseed, ltime2d=lUse2DTime)
    Nt0, ztime_model, idateModA, idateModB, ModConf, ModExp = ncio.ModelFileTimeInfo(cf_uv)

    date_stop = None
    if cdate_stop:
        date_stop = clock2epoch(cdate_stop)
    elif idateSeedB - idateSeedA >= 3600.:
        date_stop = idateSeedB                      # several records in the seeding file: replicate its time span (:168-172)
    Nt, kstrt, kstop, iTmA, iTmB = GetTimeSpan(rdt, ztime_model, idateSeedA, idateModA, idateModB, iStop=date_stop)
    if Nt < 1:
        say(' QUITTING since no matching model records!')
        comm.close()
        return 0
    say(' *** model records %d..%d (%d records): %s -> %s' % (kstrt, kstop, Nt, epoch2clock(iTmA), epoch2clock(iTmB)))
    if comm.root:
        for cd in ('seed', 'nc', 'npz'):
            os.makedirs(cd, exist_ok=True)

    ctx = _lib.Context(comm.device if comm.multi else a.device)
    tk = clk.now()
    imaskt, xlatT, xlonT, xYt, xXt, xYf, xXf, xResKM = ncio.GetModelGrid(cf_mm, ctx=ctx)
    if iUVstrategy >= 1:
        xYv, xXv, xYu, xXu = ncio.GetModelUVGrid(cf_mm, ctx=ctx)
    else:
        xYv, xXv, xYu, xXu = xYf, xXf, xYf, xXf     # never read by the cell-mean rule
    (Nj, Ni) = np.shape(imaskt)
    records = ncio.ModelRecords(cf_uv)
    tk = clk.add("read_mesh_s", tk)

    # ---- seeding, with the reference's intermediate cache (:205-255).  Seeds are independent: with several ranks
    #      each one locates its own contiguous range and the results are concatenated in rank (= seed) order.
    cf_npz_itm = './seed/Initialized_buoys_' + SeedName + '_' + CONF + '.npz'
    if comm.bcast_obj(path.exists(cf_npz_itm)):
        say(' *** using cached seed initialisation ' + cf_npz_itm)
        pack = None
        if comm.root:
            with np.load(cf_npz_itm) as data:
                pack = (data['xPosG0'], data['xPosC0'], data['IDs'], data['vJIt'], data['VRTCS'], data['idxKeep'])
        xPosG0, xPosC0, IDs, vJIt, VRTCS, idxK = comm.bcast_arrays(pack)          # tensors, not pickles
        nP = len(IDs)
    else:
        (xIC,) = records.fields(kstrt, ('siconc',))
        zt, zIDs, XseedG, XseedC = ncio.LoadNCdata(fNCseed, krec=jrecSeed)
        (nP0, _) = np.shape(XseedG)
        IDs = np.array(zIDs, dtype=int)
        lo, hi = comm.range(nP0)
        part = SeedInit(IDs[lo:hi], XseedG[lo:hi], XseedC[lo:hi], xlatT, xlonT, xYf, xXf, xResKM, imaskt,
                        xIceConc=np.asarray(xIC, dtype=np.float64), ctx=ctx)
        # every rank's kept seeds, concatenated in rank (= seed) order: tensor all-gathers
        xPosG0, xPosC0, IDs, vJIt, VRTCS, idxK = comm.allgather_rows(part[1:6] + (lo + part[6],))
        nP = len(IDs)
        if nP < nP0:
            say(' *** `SeedInit()` had to cancel ' + str(nP0 - nP) + ' buoys! => nP = ' + str(nP))
        if comm.root:
            _savez_deflate(cf_npz_itm, nP=nP, xPosG0=xPosG0, xPosC0=xPosC0, IDs=IDs, vJIt=vJIt, VRTCS=VRTCS, idxKeep=idxK)

    tk = clk.add("seed_init_and_cache_s", tk)
    # ---- per-buoy record windows (:264-318)
    z1stModelRec = np.zeros(nP, dtype=int) + kstrt
    zLstModelRec = np.zeros(nP, dtype=int) + kstop
    if lUse2DTime:
        zTpos = zTpos if np.ma.isMaskedArray(zTpos) else np.asarray(zTpos)
        if zTpos.shape != (2, nP) and zTpos.ndim == 2 and zTpos.shape[1] > nP and len(idxK) == nP:
            # SeedInit cancelled buoys: keep the time positions of the survivors.  (The reference has this line
            # commented out, si3_part_tracker.py:250, and then stops on the shape check of :269-272.)
            say(' *** adjusting `zTpos` to the %d buoys kept by SeedInit' % nP)
            zTpos = zTpos[:, idxK]
        if zTpos.shape != (2, nP):
            raise ValueError('wrong shape for the 2D time array `zTpos`: %s vs nP=%d' % (zTpos.shape, nP))
        z1stModelRec, zLstModelRec = record_windows(zTpos, ztime_model, kstrt, kstop, iTmA, iTmB, nP)
    k0 = z1stModelRec - kstrt

    # ---- device state.  Under torchrun every rank owns a contiguous range of the buoys ordered by host row, i.e. a
    #      latitude band: with row-band ingest each rank then reads only its own rows of every record, no collective.
    order = np.argsort(vJIt[:, 0], kind='stable') if comm.multi else np.arange(nP)
    lo, hi = comm.range(nP)
    part = {"order": order, "mine": order[lo:hi]}          # who owns what; re-made by rebalance()
    mine = part["mine"]

    def to_caller_order(rows):            # rows gathered in rank order -> the caller's buoy order
        if rows is None or not comm.multi:
            return rows
        out = np.empty_like(rows)
        out[part["order"]] = rows
        return out

    (u0,) = records.fields(kstrt, ('u_ice',))
    fdt = np.float64 if np.asarray(u0).dtype == np.float64 else np.float32
    K = int(max(2, min(a.slots, 64)))
    trk = IceTracker(xYf, xXf, xYu, xXu, xYv, xXv, imaskt, rdt=rdt, iUVstrategy=iUVstrategy, nslots=K, field_dtype=fdt, ctx=ctx)
    trk.set_buoys(xPosC0[mine], vJIt[mine], z1stModelRec[mine] if lUse2DTime else None, zLstModelRec[mine] if lUse2DTime else None)

    # ---- outputs (rank 0).  The reference allocates the whole series -- xPosC, xPosG (Nt+1,nP,2) f8, xmask -- and writes it at the
    #      end (:324-330,515-519).  Here the series file is opened now and every record is appended when it is fetched: host
    #      memory stays O(nP).  Its time axis is known in advance (record times of the model file).
    lFull = not lUse2DTime
    stride = max(1, int(a.out_stride)) if lFull else 1
    vTime = np.zeros(Nt + 1, dtype=int)
    for jt_ in range(Nt):
        vTime[jt_] = records.time(jt_ + kstrt) - int(rdt / 2.)
    vTime[Nt] = vTime[Nt - 1] + int(rdt)
    corgn = 'NEMO-SI3_' + ModConf + '_' + ModExp
    series, cf_series = None, None
    if lFull and comm.root:
        cf_series = ('./nc/' + corgn + '_tracking_' + SeedBatch + cdtbin + '_' + date_tag(vTime[0]) + '_' + date_tag(vTime[Nt]) + csfkm
                     + ('_stride%d' % stride if stride > 1 else '') + '.nc')
        series = ncio.CloudBuoysStream(cf_series, vTime[::stride], IDs, with_mask=True, corigin=corgn)
        # record 0: the seeds as the seeding file gave them (:331-333; in 1-D-time mode every window opens at record 0)
        series.put(0, xPosC0[:, 0], xPosC0[:, 1], xPosG0[:, 0], xPosG0[:, 1], np.ones(nP, dtype='i1'))
        z2XY, z2GC = np.zeros((2, nP, 2)) + FILL, np.zeros((2, nP, 2)) + FILL
        zMSK = np.zeros((2, nP), dtype='i1')
        z2XY[0], z2GC[0], zMSK[0] = xPosC0, xPosG0, 1
    if lUse2DTime:
        ends = set(np.unique(zLstModelRec).tolist())
        if comm.root:
            z2XY, z2GC = np.zeros((2, nP, 2)) + FILL, np.zeros((2, nP, 2)) + FILL
            zMSK, zTim = np.zeros((2, nP), dtype='i1'), np.zeros((2, nP), dtype=int) + int(FILL)
            z2XY[0], z2GC[0], zMSK[0] = xPosC0, xPosG0, 1
            zTim[0] = ztime_model[z1stModelRec] - int(rdt / 2)
            # a buoy whose window opens later has its seed position pre-written at record k0, and the reference converts
            # that row to lat/lon when it passes over record k0-1 (:493)
            late = k0 > 0
            if np.any(late):
                z2GC[0, late] = ctx.cart2geo(xPosC0[late])

    # ---- the record loop (:361-496).  The reference steps record by record; here consecutive records go through ONE
    #      fused launch (sitrk_run: every buoy still takes every step, only the loop nest is interchanged) whenever no
    #      per-record output is due between them -- always in the default 2-D-time mode, which writes first/last positions
    #      only (:565-571).  Records are read straight into the library's pinned staging and uploaded on its copy stream
    #      while the previous batch is stepped with; `-F` / `-p` need every record's positions: batches of one.
    def need_output(jrec):
        k = jrec - kstrt + 1                               # the record of the series this step produces
        return (lFull and (k % stride == 0 or k == Nt)) or (lUse2DTime and jrec in ends)

    bcast = None
    if a.full_records and comm.multi and comm.backend == "nccl":
        from .distributed import RecordBroadcaster
        bcast = RecordBroadcaster(ctx)                 # rank 0 reads; one RCCL broadcast per record, overlapped with the stepping
    batches, jt = [], 0
    while jt < Nt:
        m = 1
        while m < K // 2 and jt + m < Nt and not need_output(jt + m - 1 + kstrt):
            m += 1
        batches.append((jt, m))
        jt += m
    band = {"box": None, "age": None, "bytes": 0}
    esz = np.dtype(fdt).itemsize

    def upload(jt0, m):
        """records jt0..jt0+m-1 -> slots (jt0+r) % K, asynchronously"""
        t_up = clk.now()
        if not a.full_records:
            # the box (rows x columns, round 4; rows only before) this rank's buoys can touch during those records: their host
            # cells at the last evaluation, widened by one cell per record stepped or queued since (sitrk_buoy_box waits for
            # the GPU: only every so often).  Only that hyperslab of the record is read from the file and uploaded.
            if band["age"] is None or band["age"] + m > 96:
                band["box"] = ctx.buoy_box()
                band["age"] = 0
            j0, j1, i0, i1 = ctx.box_of(*band["box"], band["age"] + m - 1)
            band["age"] += m
        for r in range(m):
            jrec, slot = jt0 + r + kstrt, (jt0 + r) % K
            if not a.full_records:
                if j1 > j0:
                    ctx.stage_fill(slot, j0, j1 - j0, lambda *outs: records.fields_box_into(jrec, j0, j1, i0, i1, outs), i0=i0, ncols=i1 - i0)
                    band["bytes"] += 3 * (j1 - j0) * (i1 - i0) * esz
                else:
                    ctx.commit_record_rows(slot, 0, 0)                  # no live buoy: nothing to read
            elif not comm.multi:
                ctx.stage_fill(slot, 0, Nj, lambda *outs: records.fields_rows_into(jrec, 0, Nj, outs))   # the whole record (:372-374)
            elif bcast is not None:
                bcast.deliver(slot, records.fields(jrec) if comm.root else None)
            else:
                comm.deliver_record(ctx, slot, records.fields(jrec) if comm.root else None)   # gloo rehearsal: host broadcast
        clk.add("read_and_stage_records_s", t_up)

    tk = clk.add("setup_s", tk)
    t_loop = clk.now()
    def rebalance():
        """Re-partition the buoys over the ranks by their CURRENT host row (migration / re-balancing).
        Each rank fetches its buoys' state, a global histogram of host rows fixes every buoy's place in the new order --
        by (row, owning rank, local order) -- hence its new owner; the states travel in one all-to-all of 50-byte rows and
        every rank re-creates its buoy set with their history (dead flags, kill records).  Trajectories do not depend on
        who steps a buoy; only the rows a rank has to read do."""
        st = ctx.fetch()
        mine_g = part["mine"]
        rows = st["jiT"][:, 0].astype(np.int64)
        hist = np.bincount(rows, minlength=Nj).astype(np.int64)
        allh = comm.allgather_rows((hist[None, :],))[0]                        # (world, Nj)
        before_row = np.concatenate([[0], np.cumsum(allh.sum(axis=0))[:-1]])    # buoys in lower rows
        before_rank = allh[:comm.rank].sum(axis=0)                              # same row, lower ranks
        o = np.argsort(rows, kind='stable')
        within = np.empty(len(rows), dtype=np.int64)
        starts = np.concatenate([[0], np.cumsum(hist)[:-1]])
        within[o] = np.arange(len(rows)) - starts[rows[o]]                      # same row, this rank, local order
        newpos = before_row[rows] + before_rank[rows] + within                  # place in the new global order
        ends = np.array([hi_ for _, hi_ in all_ranges(nP, comm.world)])         # the new owner: whose range the place falls in
        dest = np.searchsorted(ends, newpos, side='right')
        f8 = np.concatenate([st["yx"], newpos[:, None].astype(np.float64)], axis=1)          # positions travel exactly
        win = (z1stModelRec[mine_g], zLstModelRec[mine_g]) if lUse2DTime else (np.zeros(len(rows), dtype=int),) * 2
        i8 = np.stack([mine_g.astype(np.int64), st["jiT"][:, 0].astype(np.int64), st["jiT"][:, 1].astype(np.int64),
                       st["alive"].astype(np.int64), st["kill_rec"].astype(np.int64), np.asarray(win[0], dtype=np.int64),
                       np.asarray(win[1], dtype=np.int64)], axis=1)
        part["rebalances"] = part.get("rebalances", 0) + 1
        part["migrated"] = part.get("migrated", 0) + comm.sum_int(int((dest != comm.rank).sum()))
        chunks = [(f8[dest == d], i8[dest == d]) for d in range(comm.world)]
        rf8, ri8 = comm.alltoall_rows(chunks)
        k = np.argsort(rf8[:, 2], kind='stable')                               # the new order inside this rank's range
        rf8, ri8 = rf8[k], ri8[k]
        part["mine"] = ri8[:, 0].copy()
        ctx.set_buoys(np.ascontiguousarray(rf8[:, :2]), ri8[:, 1:3].astype(np.int32),
                      ri8[:, 5].astype(np.int32) if lUse2DTime else None, ri8[:, 6].astype(np.int32) if lUse2DTime else None, sort=False)
        ctx.restore_state(ri8[:, 3].astype(np.int8), ri8[:, 4].astype(np.int32))
        ctx.sort_buoys()
        allmine = comm.gather_rows(part["mine"], nP)
        if comm.root:
            part["order"] = allmine
        band["age"] = None                                                      # the rows this rank reads start over

    due, since = [], 0
    for (jt0, m) in batches:              # after which batches the ranks re-balance: a function of the record count alone
        since += m
        due.append(comm.multi and a.rebalance > 0 and since >= a.rebalance)
        if due[-1]:
            since = 0
    if due:
        due[-1] = False
    if batches:
        upload(*batches[0])
    for ib, (jt0, m) in enumerate(batches):
        jrec0, jrecN = jt0 + kstrt, jt0 + m - 1 + kstrt
        if m == 1:
            nalive = comm.sum_int(trk.alive_count())
            say(' *** record #%d/%d  date = %s   buoys alive = %d' % (jrec0 + 1, Nt0, epoch2clock(vTime[jt0]), nalive))
        else:
            say(' *** records #%d..#%d/%d  dates = %s .. %s   (one fused launch)' % (jrec0 + 1, jrecN + 1, Nt0, epoch2clock(vTime[jt0]),
                                                                                   epoch2clock(vTime[jt0 + m - 1])))
        t_q = clk.now()
        used = [(jt0 + r) % K for r in range(m)]
        if bcast is not None:
            bcast.before_run(used)
        trk.run(jrec0, jt0 % K, m)
        if bcast is not None:
            bcast.after_run(used)
        clk.add("enqueue_stepping_s", t_q)
        if ib + 1 < len(batches) and not due[ib]:
            upload(*batches[ib + 1])               # travels while the launch above runs
        jt, jrec, itime = jt0 + m - 1, jrecN, vTime[jt0 + m - 1]
        need = need_output(jrec)
        t_f = clk.now()
        if need:
            # one record at a time, never the series: xPosC[jt+1], xmask[jt+1] (:459-460) and, for the series, xPosG[jt+1] =
            # CartNPSkm2Geo1D of that row, FillValue rows included (:493), converted on the device by the rank that owns the rows
            if lFull:
                pos_l, msk_l, ll_l = trk.record(jrec, latlon=True)
                ll = to_caller_order(comm.gather_rows(ll_l, nP))
            else:
                pos_l, msk_l = trk.record(jrec)
            pos, msk = to_caller_order(comm.gather_rows(pos_l, nP)), to_caller_order(comm.gather_rows(msk_l, nP))
        if need and comm.root:
            stepped = msk == 1
            if lFull:
                k = jt + 1
                try:
                    if k % stride == 0:
                        series.put(k // stride, pos[:, 0], pos[:, 1], ll[:, 0], ll[:, 1], msk)
                except BaseException:
                    series.abort()
                    raise
                if k == Nt:
                    z2XY[1], z2GC[1], zMSK[1] = pos, ll, msk
            if lUse2DTime and jrec in ends:
                sel = np.where(zLstModelRec == jrec)[0]
                z2XY[1, sel] = pos[sel]
                z2GC[1, sel] = ctx.cart2geo(pos[sel])
                zMSK[1, sel] = msk[sel]
                zTim[1, sel[stepped[sel]]] = int(itime + rdt)
        clk.add("fetch_and_store_outputs_s", t_f)
        if due[ib]:
            t_r = clk.now()
            rebalance()
            clk.add("rebalance_s", t_r)
            upload(*batches[ib + 1])
    ctx.sync()
    if bcast is not None:
        bcast.close()
    clk.add("record_loop_s", t_loop)
    if not a.full_records:
        # what box ingest bought, per rank (every rank prints its own line)
        print(' *** rank %d read and uploaded %.1f MB of the %.1f MB of its %d records (box ingest: the rows x columns its own buoys can touch)'
              % (comm.rank, band["bytes"] / 1e6, 3.0 * Nj * Ni * esz * Nt / 1e6, Nt), flush=True)
    tk = clk.now()
    records.close()
    launches = ctx.launch_stats()
    state = trk.state()
    vJIt_end, alive_end = to_caller_order(comm.gather_rows(state["vJIt"], nP)), to_caller_order(comm.gather_rows(state["iAlive"], nP))
    trk.close()
    if not comm.root:
        comm.close()
        return {"rank": comm.rank, "nP": nP, "range": (lo, hi), "upload_bytes": band["bytes"]}

    # ---- outputs (:509-571)
    outs = []
    if not lUse2DTime:
        series.close()                                       # every record is in the file already
        outs.append(cf_series)
        zTim = []
        zvt = np.array([vTime[0], vTime[Nt]])
    else:
        zvt = np.array([np.mean(zTim[0, :]), np.mean(zTim[1, :])])
    cf_nc_out = './nc/' + corgn + '_tracking12_' + SeedBatch + cdtbin + '_' + date_tag(zvt[0]) + '_' + date_tag(zvt[1]) + csfkm + '.nc'
    ncio.ncSaveCloudBuoys(cf_nc_out, zvt, IDs, z2XY[:, :, 0], z2XY[:, :, 1], z2GC[:, :, 0], z2GC[:, :, 1], mask=zMSK, xtime=zTim,
                          corigin=corgn)
    outs.append(cf_nc_out)

    if a.plot > 0:
        try:
            import mojito as mjt          # noqa: F401  optional, as in the reference (:20,587-600)
            print(' *** `mojito` found: map plotting is left to the reference tooling (positions are in ' + outs[0] + ')')
        except Exception:                 # noqa: BLE001
            print(' *** `-p`: package `mojito` not available here => no maps (positions are in ' + outs[0] + ')')
    print(' *** first and final dates in simulated trajectories:', epoch2clock(zvt[0]), epoch2clock(zvt[1]))
    for f in outs:
        print('      ===> ' + f + ' saved!')
    clk.add("write_outputs_s", tk)
    clk.t["total_s"] = clk.now() - clk.t0
    comm.close()
    return {"files": outs, "nP": nP, "IDs": IDs, "vJIt": vJIt_end, "iAlive": alive_end, "Nt": Nt, "kstrt": kstrt,
            "timing": dict(clk.t), "launches": launches, "upload_bytes": band["bytes"], "record_bytes": 3 * Nj * Ni * esz,
            "rebalances": part.get("rebalances", 0), "migrated": part.get("migrated", 0)}
