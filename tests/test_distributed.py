"""N>1 path on CPU: world_size-2 gloo rehearsal of the buoy-range partition, the per-record
slab broadcast and the end-of-run gather.  The per-rank stepping is done here by the CPU
oracle (the GPU kernels cannot run in this container); the exchange code is the product's."""
import os
import socket

import numpy as np
import pytest

from sitrack_amd import distributed as sd
from sitrack_amd import synthetic as syn


def test_buoy_ranges_cover_in_order():
    for nP in (0, 1, 7, 8, 9, 1000, 10_000_001):
        for world in (1, 2, 3, 8):
            r = sd.all_ranges(nP, world)
            assert r[0][0] == 0 and r[-1][1] == nP
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1


def test_pack_and_split_slab():
    rng = np.random.default_rng(0)
    u, v, s = (rng.random((5, 7)).astype(np.float32) for _ in range(3))
    slab = sd.pack_slab(u, v, s, np.float32)
    assert slab.shape == (3 * 35,)
    uu, vv, ss = sd.split_slab(slab, 5, 7)
    assert np.array_equal(uu, u) and np.array_equal(vv, v) and np.array_equal(ss, s)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Nj, Ni, nP, K, Nt = 40, 48, 901, 3, 12
        grid = syn.make_grid(Nj, Ni, dkm=4.0, warp=1.0)           # geometry replicated: every rank builds it
        _, yx = syn.make_buoys(grid, nP, seed=9, frac=0.6)
        guess = syn.nearest_t_plane(grid, yx)
        ok = np.zeros(nP, dtype=bool); ji = np.zeros((nP, 2), dtype=np.int64)
        for b in range(nP):
            ok[b], ji[b], _ = orc.FindContainingCell(yx[b], guess[b], grid["Yf"], grid["Xf"])
        yx, ji = yx[ok], ji[ok]
        nP = len(yx)
        lo, hi = sd.buoy_range(nP, rank, world)
        mine = orc.Tracker(grid, yx[lo:hi], ji[lo:hi])
        fields = syn.make_fields(grid, K=K, seed=4, umax=0.7, drift=0.2, ripple=0.1) if rank == 0 else None
        for jrec in range(Nt):
            slab = sd.pack_slab(fields[0][jrec % K], fields[1][jrec % K], fields[2][jrec % K], np.float32) if rank == 0 else None
            slab = sd.broadcast_record_host(slab, 3 * Nj * Ni, np.float32, src=0)      # the path's only exchange
            u, v, s = sd.split_slab(slab, Nj, Ni)
            mine.step(jrec, u, v, s, want_out=False)
        pos = sd.gather_ranges(mine.pos, nP)
        cells = sd.gather_ranges(mine.jiT, nP)
        alive = sd.gather_ranges(mine.alive, nP)
        if rank == 0:
            ref = orc.Tracker(grid, yx, ji)
            for jrec in range(Nt):
                ref.step(jrec, fields[0][jrec % K], fields[1][jrec % K], fields[2][jrec % K], want_out=False)
            good = np.array_equal(pos, ref.pos) and np.array_equal(cells, ref.jiT) and np.array_equal(alive, ref.alive)
            q.put(("ok" if good else "mismatch", int(ref.ncross)))
    except Exception as e:                                           # pragma: no cover
        if rank == 0:
            q.put(("error: %r" % (e,), 0))
        raise
    finally:
        dist.destroy_process_group()


def _sag_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        for n in (10, 3 * 7 * 11, 1000):                      # divisible and not divisible by the world size
            want = torch.arange(n, dtype=torch.float32) * 0.5 + 1.0
            t = want.clone() if rank == 1 else torch.full((n,), -7.0)
            sd.scatter_allgather(t, src=1)
            ok = ok and bool(torch.equal(t, want))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def _gather_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        for nP in (0, 1, 2, 10, 1001):                      # fewer buoys than ranks, uneven ranges
            lo, hi = sd.buoy_range(nP, rank, world)
            full = {"f8": np.arange(2 * nP, dtype=np.float64).reshape(nP, 2) * 0.25, "i1": (np.arange(nP) % 3).astype(np.int8),
                    "i8": (np.arange(2 * nP).reshape(nP, 2) + 2 ** 40).astype(np.int64), "i4": np.arange(nP, dtype=np.int32)}
            for key, arr in full.items():
                got = sd.gather_ranges(arr[lo:hi], nP)
                if rank == 0:
                    ok = ok and got.dtype == arr.dtype and got.shape == arr.shape and np.array_equal(got, arr)
                else:
                    ok = ok and got is None
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def _a2a_worker(rank, world, port, q):
    import torch.distributed as dist
    from sitrack_amd.distributed import Comm
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), SITRK_DIST_BACKEND="gloo")
    comm = Comm()
    try:
        rng = np.random.default_rng(100 + rank)
        n = [0, 7, 1000][rank]                                  # one rank has nothing to send
        f8 = rng.random((n, 3)); i8 = rng.integers(0, 1 << 40, (n, 2)); tag = np.full(n, rank, dtype=np.int64)
        dest = rng.integers(0, world, n)
        got_f8, got_i8, got_tag = comm.alltoall_rows([(f8[dest == d], i8[dest == d], tag[dest == d]) for d in range(world)])
        # everybody tells everybody what they sent where: the expected result on each rank
        everything = comm.allgather_obj((f8, i8, dest))
        want_f8 = np.concatenate([e[0][e[2] == rank] for e in everything])
        want_i8 = np.concatenate([e[1][e[2] == rank] for e in everything])
        ok = np.array_equal(got_f8, want_f8) and np.array_equal(got_i8, want_i8) and np.all(np.diff(got_tag) >= 0)
        s = comm.allreduce_sum(np.array([rank + 1, 10], dtype=np.int64))
        ok = ok and list(s) == [6, 30]
        q.put((rank, bool(ok)))
    finally:
        comm.close()


def _world8_worker(rank, world, port, q):
    """everything the 8-GPU run of BASELINE config 4 exchanges, with 8 gloo ranks on the CPU: identity gather, slab broadcast and
    scatter + all-gather of a slab whose length is not a multiple of 8, buoy ranges incl. fewer buoys than ranks, the
    all-to-all migration, reductions -- and the oracle stepping every rank's range"""
    import torch
    import torch.distributed as dist
    from oracle import oracle as orc
    from sitrack_amd.distributed import Comm
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), SITRK_DIST_BACKEND="gloo")
    comm = Comm()
    try:
        ok = comm.world == 8 and comm.multi
        ids = comm.allgather_obj((rank, os.getpid()))
        ok = ok and [r for r, _ in ids] == list(range(8)) and len({p for _, p in ids}) == 8
        # slab exchange both ways
        for n in (3 * 37 * 41, 1000):
            want = torch.arange(n, dtype=torch.float32) * 0.25 - 3.0
            t = want.clone() if rank == 0 else torch.zeros(n)
            sd.scatter_allgather(t, src=0)
            t2 = want.clone() if rank == 0 else torch.zeros(n)
            dist.broadcast(t2, src=0)
            ok = ok and bool(torch.equal(t, want)) and bool(torch.equal(t2, want))
        # ranges, gathers
        for nP in (0, 5, 8, 1003):
            lo, hi = sd.buoy_range(nP, rank, world)
            arr = np.arange(2 * nP, dtype=np.float64).reshape(nP, 2) * 0.5
            got = sd.gather_ranges(arr[lo:hi], nP)
            ok = ok and ((got is None) if rank else np.array_equal(got, arr))
        assert sd.all_ranges(1003, 8)[-1][1] == 1003 and sum(h - l for l, h in sd.all_ranges(1003, 8)) == 1003
        # migration
        rng = np.random.default_rng(200 + rank)
        n = [0, 7, 1000, 3, 250, 1, 64, 500][rank]
        f8 = rng.random((n, 3)); dest = rng.integers(0, world, n)
        (got_f8,) = comm.alltoall_rows([(f8[dest == d],) for d in range(world)])
        everything = comm.allgather_obj((f8, dest))
        ok = ok and np.array_equal(got_f8, np.concatenate([e[0][e[1] == rank] for e in everything]))
        ok = ok and list(comm.allreduce_sum(np.array([rank + 1], dtype=np.int64))) == [36]
        # partition + per-record slab broadcast + gather, the oracle stepping each range
        Nj, Ni, K, Nt = 40, 48, 3, 8
        grid = syn.make_grid(Nj, Ni, dkm=4.0, warp=1.0)
        _, yx = syn.make_buoys(grid, 700, seed=9, frac=0.6)
        guess = syn.nearest_t_plane(grid, yx)
        good = np.zeros(len(yx), dtype=bool); ji = np.zeros((len(yx), 2), dtype=np.int64)
        for b in range(len(yx)):
            good[b], ji[b], _ = orc.FindContainingCell(yx[b], guess[b], grid["Yf"], grid["Xf"])
        yx, ji = yx[good], ji[good]
        lo, hi = sd.buoy_range(len(yx), rank, world)
        mine = orc.Tracker(grid, yx[lo:hi], ji[lo:hi])
        fields = syn.make_fields(grid, K=K, seed=4, umax=0.7, drift=0.2, ripple=0.1) if rank == 0 else None
        for jrec in range(Nt):
            slab = sd.pack_slab(fields[0][jrec % K], fields[1][jrec % K], fields[2][jrec % K], np.float32) if rank == 0 else None
            u, v, s_ = sd.split_slab(sd.broadcast_record_host(slab, 3 * Nj * Ni, np.float32, src=0), Nj, Ni)
            mine.step(jrec, u, v, s_, want_out=False)
        pos = sd.gather_ranges(mine.pos, len(yx)); cells = sd.gather_ranges(mine.jiT, len(yx))
        if rank == 0:
            ref = orc.Tracker(grid, yx, ji)
            for jrec in range(Nt):
                ref.step(jrec, fields[0][jrec % K], fields[1][jrec % K], fields[2][jrec % K], want_out=False)
            ok = ok and np.array_equal(pos, ref.pos) and np.array_equal(cells, ref.jiT) and ref.ncross > 50
        q.put((rank, bool(ok)))
    finally:
        comm.close()


def test_world8_gloo_rehearsal_of_the_8_gpu_exchange():
    """BASELINE config 4 runs on 8 ranks; the driver launches that, not the builder.  Its rank arithmetic and every exchange
    primitive with EIGHT ranks (gloo, CPU): see _world8_worker."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_world8_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(8)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _ in res) == list(range(8)) and all(ok for _, ok in res), res


def test_alltoall_rows_world3_gloo():
    """the migration primitive of the re-balancing (Comm.alltoall_rows): rows for every destination, received in
    source-rank order; gloo has no all-to-all, so this is the isend / irecv form"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_a2a_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(3)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _ in res) == [0, 1, 2] and all(ok for _, ok in res)


def test_tensor_gather_of_buoy_ranges_world3():
    """gather_ranges moves every rank's rows as one tensor (no pickled objects): three gloo ranks, uneven and empty
    ranges, the dtypes the driver gathers (positions f8, masks i1, cells i8, ...)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(3)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _ in res) == [0, 1, 2] and all(ok for _, ok in res)


def test_scatter_allgather_equals_broadcast_world3():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sag_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(3)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _ in res) == [0, 1, 2] and all(ok for _, ok in res)


def test_world2_gloo_partition_broadcast_gather():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    status, ncross = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert status == "ok", status
    assert ncross > 100


def _box_bcast_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from sitrack_amd import distributed as sd
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Nj, Ni = 40, 52
        rng = np.random.default_rng(11)
        rec = rng.normal(size=3 * Nj * Ni).astype(np.float32)                 # the record (rank 0 holds it)
        old = np.full(3 * Nj * Ni, -7.0, dtype=np.float32)                     # what the other ranks' slots held before
        slot = torch.from_numpy(rec.copy() if rank == 0 else old.copy())
        # every rank's own box; rank 2 has no live buoy (empty box)
        mine = [(5, 20, 8, 24), (12, 31, 4, 16), (0, 0, 0, 0)][rank]
        box = sd.union_box(mine, Nj, Ni)
        ok = box == (5, 31, 4, 24)
        buf = torch.empty(3 * (box[1] - box[0]) * (box[3] - box[2]), dtype=torch.float32)
        moved = sd.broadcast_box(slot, Nj, Ni, box, buf, src=0)
        ok = ok and moved == 3 * 26 * 20 * 4
        want = (rec if rank == 0 else old).reshape(3, Nj, Ni).copy()
        want[:, box[0]:box[1], box[2]:box[3]] = rec.reshape(3, Nj, Ni)[:, box[0]:box[1], box[2]:box[3]]
        ok = ok and np.array_equal(slot.numpy().reshape(3, Nj, Ni), want)
        ok = ok and sd.union_box((0, 0, 0, 0), Nj, Ni) == (0, 0, 0, 0)          # nobody has a live buoy
        t = torch.tensor([1 if ok else 0])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if rank == 0:
            q.put("ok" if int(t[0]) == 1 else "wrong box or contents on some rank")
    finally:
        dist.destroy_process_group()


def test_box_broadcast_world3_gloo():
    """round 4: one broadcast of the union of the ranks' boxes instead of the whole slab (bench.py's `box_broadcast` segment):
    three ranks, one of them without a live buoy; the box arrives in every slot, everything else of the slots is untouched."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_box_bcast_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    status = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert status == "ok", status


class _CpuLayer:
    """Stand-in for torch.cuda in RecordBroadcaster: the host is the device, work runs when it is queued.  Events carry the
    sequence number of their last record(), so the test can audit WHAT each wait fences (the protocol), while the slot contents
    show that the right bytes were there when they were stepped with."""

    def __init__(self, torch):
        self._torch, self.seq, self.log = torch, 0, []
        layer = self

        class Stream:
            cuda_stream = None

            def wait_event(self, e):
                assert e.seq is not None, "wait on an event that was never recorded"
                layer.log.append(("wait", e.seq))
                self.last_waited = e.seq

            def synchronize(self):
                pass

        class Event:
            seq = None

            def record(self, stream=None):
                layer.seq += 1
                self.seq = layer.seq

            def synchronize(self):
                pass

        self.Stream, self.Event = Stream, Event

    def tick(self):
        self.seq += 1
        return self.seq

    def stream(self, s):
        import contextlib
        return contextlib.nullcontext()

    def pinned(self, n, dtype):
        return self._torch.empty(n, dtype=dtype)


class _FakeCtx:
    """what RecordBroadcaster touches of a sitrack_amd Context: slots, their geometry, the in-place-write / commit protocol"""

    def __init__(self, torch, nslots, Nj, Ni):
        self.nslots, self.Nj, self.Ni, self.slab_elems = nslots, Nj, Ni, 3 * Nj * Ni
        self.mem = [torch.full((self.slab_elems,), float("nan"), dtype=torch.float32) for _ in range(nslots)]
        self.dirty = [False] * nslots
        self.committed = [None] * nslots                 # checksum of the slab at its last commit
        self.stream = "own"

    def set_stream(self, s):
        self.stream = s

    def record_ptr(self, slot):
        self.dirty[slot] = True

    def commit_record(self, slot):
        assert self.dirty[slot], "commit of a slot nobody wrote"
        self.dirty[slot] = False
        self.committed[slot] = float(self.mem[slot].double().sum())


def _bcaster_worker(rank, world, port, q):
    """RecordBroadcaster with a real second rank (gloo) and its stream / event layer on the CPU: 13 records through 4 slots in
    batches of 2, every slot rewritten three times while the protocol decides when."""
    import torch
    import torch.distributed as dist
    from sitrack_amd import distributed as sd
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Nj, Ni, S, Nrec, m = 6, 10, 4, 13, 2
        n = Nj * Ni
        rng = np.random.default_rng(3)
        recs = [tuple(rng.integers(-99, 99, (Nj, Ni)).astype(np.float64) * 0.25 for _ in range(3)) for _ in range(Nrec)]   # exact in f4
        ctx = _FakeCtx(torch, S, Nj, Ni)
        dev = _CpuLayer(torch)
        bc = sd.RecordBroadcaster(ctx, src=0, dev=dev, slot_view=lambda c, k: c.mem[k])
        assert ctx.stream is None                         # the library's stream was handed over to the layer's compute stream
        last_read, last_write, holds = [0] * S, [0] * S, [None] * S
        ok = True

        def deliver(jt):
            slot = jt % S
            bc.deliver(slot, recs[jt] if rank == 0 else None)
            # the write waited for an event recorded AFTER the last launch that read the slot ...
            nonlocal ok
            ok = ok and bc.comm.last_waited > last_read[slot]
            last_write[slot] = bc.ready[slot].seq
            holds[slot] = jt

        def run(jt0, cnt):
            nonlocal ok
            used = [(jt0 + r) % S for r in range(cnt)]
            bc.before_run(used)
            for r, slot in enumerate(used):
                # ... the launch waited for the slot's delivery, committed it, and the slot holds the record it is stepped with
                want = np.concatenate([a.reshape(-1) for a in recs[jt0 + r]]).astype(np.float32)
                ok = ok and holds[slot] == jt0 + r and bc.ready[slot].seq >= last_write[slot] and not ctx.dirty[slot]
                ok = ok and np.array_equal(ctx.mem[slot].numpy(), want) and ctx.committed[slot] == float(want.astype(np.float64).sum())
                last_read[slot] = dev.tick()
            bc.after_run(used)
            for slot in used:
                ok = ok and bc.free[slot].seq > last_read[slot]

        batches = [(j, min(m, Nrec - j)) for j in range(0, Nrec, m)]
        for r in range(batches[0][1]):
            deliver(r)
        for ib, (jt0, cnt) in enumerate(batches):
            if ib + 1 < len(batches):                   # the next batch travels "while" this one is stepped with
                for r in range(batches[ib + 1][1]):
                    deliver(batches[ib + 1][0] + r)
            run(jt0, cnt)
        # a field that does not survive the cast to the slot's type is refused on the source rank, before anything is sent
        if rank == 0:
            bad = list(recs[0]); bad[2] = bad[2] + 1e-9
            try:
                bc.deliver(0, tuple(bad))
                ok = False
            except ValueError as e:
                ok = ok and "siconc is not exactly representable" in str(e)
        bc.close()
        ok = ok and ctx.stream is None
        t = torch.tensor([1 if ok else 0])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if rank == 0:
            q.put("ok" if int(t[0]) == 1 else "protocol or contents wrong on some rank")
    except Exception as e:                                # noqa: BLE001
        if rank == 0:
            q.put("rank 0: %r" % (e,))
        raise
    finally:
        dist.destroy_process_group()


def test_record_broadcaster_slot_recycling_world2_gloo():
    """`RecordBroadcaster` (distributed.py) -- pinned staging -> slot on a communication stream, broadcast in
    place, the compute stream ordered behind the slot's event only, slots recycled behind the launches that read them -- had only
    ever run with ONE rank.  Here its stream / event layer is a CPU stand-in and the broadcast is a real gloo broadcast between
    two processes: the slot-recycling logic runs with a second rank, contents and fences audited on both."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bcaster_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    status = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert status == "ok", status


@pytest.mark.gpu
def test_slot_tensor_broadcast_world1_rccl():
    """RCCL path on one GPU: the broadcast writes the resident slot in place."""
    import torch
    import torch.distributed as dist
    import sitrack_amd as sit
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        grid = syn.make_grid(64, 64, dkm=4.0, warp=1.0)
        u, v, s = syn.make_fields(grid, K=2, seed=1, umax=0.6, drift=0.2)
        _, yx = syn.make_buoys(grid, 3000, seed=3)
        outs = []
        for mode in ("push", "bcast"):
            trk = sit.IceTracker(grid["Yf"], grid["Xf"], grid["Yu"], grid["Xu"], grid["Yv"], grid["Xv"], grid["tmask"], nslots=2)
            found, ji, _ = sit.FindContainingCell(yx, syn.nearest_t_plane(grid, yx), ctx=trk.ctx)
            trk.set_buoys(yx[found], ji[found])
            for k in range(2):
                if mode == "push":
                    trk.load_record(k, u[k], v[k], s[k])
                else:
                    sd.broadcast_record(trk.ctx, k, sd.pack_slab(u[k], v[k], s[k], np.float32), src=0)
            trk.ctx.run(0, 0, 9)
            outs.append(trk.state())
            trk.close()
        for key in ("yx", "vJIt", "iAlive"):
            assert np.array_equal(outs[0][key], outs[1][key])
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_record_broadcaster_overlapped_delivery_world1_rccl():
    """distributed.RecordBroadcaster (what `--full-records` uses under torchrun): pinned staging -> slot on a
    communication stream, RCCL broadcast in place, the compute stream ordered by events only; batches are delivered one
    ahead of the launches that use them and slots are recycled.  One rank is all a one-GPU box allows RCCL; the result
    must equal the oracle's."""
    import torch
    import torch.distributed as dist
    import sitrack_amd as sit
    from oracle import oracle as orc
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        Nj, Ni, K, Nt = 80, 96, 6, 36
        grid = syn.make_grid(Nj, Ni, dkm=4.0, warp=1.0)
        u, v, s = syn.make_fields(grid, K=Nt, seed=1, umax=0.8, drift=0.3)
        s[:, 20:30, 40:60] = 0.02
        _, yx = syn.make_buoys(grid, 20000, seed=3, frac=0.8)
        trk = sit.IceTracker(grid["Yf"], grid["Xf"], grid["Yu"], grid["Xu"], grid["Yv"], grid["Xv"], grid["tmask"], nslots=K)
        found, ji, _ = sit.FindContainingCell(yx, syn.nearest_t_plane(grid, yx), ctx=trk.ctx)
        yx, ji = yx[found], ji[found]
        trk.set_buoys(yx, ji)
        ref = orc.Tracker(grid, yx, ji, nthreads=4)
        bc = sd.RecordBroadcaster(trk.ctx)
        m = K // 2
        for r in range(m):
            bc.deliver(r % K, (u[r], v[r], s[r]))
        for b in range(Nt // m):
            used = [(b * m + r) % K for r in range(m)]
            bc.before_run(used)
            trk.run(b * m, (b * m) % K, m)
            bc.after_run(used)
            if b + 1 < Nt // m:
                for r in range((b + 1) * m, (b + 2) * m):
                    bc.deliver(r % K, (u[r], v[r], s[r]))
        bc.close()
        for r in range(Nt):
            ref.step(r, u[r], v[r], s[r], want_out=False)
        st = trk.state()
        assert np.array_equal(st["yx"], ref.pos) and np.array_equal(st["vJIt"], ref.jiT) and np.array_equal(st["iAlive"], ref.alive)
        assert 0 < st["iAlive"].sum() < len(yx)
        # the tensor gather on the device through RCCL (one rank: the whole set)
        got = sd.gather_ranges(st["yx"], len(yx))
        assert np.array_equal(got, st["yx"])
        trk.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_bench_two_ranks_as_the_driver_launches_it():
    """bench.py under `python -m torch.distributed.run --nproc-per-node 2` (the driver's launch line), both ranks on the
    box's one GPU with gloo as transport (RCCL refuses two ranks per device): one JSON line from rank 0, whole-job value."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SITRK_DIST_BACKEND="gloo", SITRK_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "40", "--warmup", "8", "--config", "c2",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 40 and d["warmup"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["buoys_per_gpu"] == 100000 and "broadcast from rank 0" in d["config"]["records_via"]
    assert abs(d["value"] - 2 * 100000 * 40 / (d["ms_per_step"] * 40e-3)) < 1e-6 * d["value"]
    # the line proves who took part and that the exchange delivered rank 0's bytes to every rank
    rc = d["rccl"]
    assert rc["world"] == 2 and rc["backend"] == "gloo" and [x["rank"] for x in rc["ranks"]] == [0, 1]
    assert all(x["device"] == 0 and x["pid"] > 0 and "name" in x for x in rc["ranks"]) and rc["ranks"][0]["pid"] != rc["ranks"][1]["pid"]
    assert rc["distinct_devices"] == 1                       # the rehearsal: both ranks on the box's one GPU
    assert rc["slab_checksum_ok"] is True and rc["slab_checksum_matches_source"] is True
    assert len(rc["slab_checksums"]) == 2 and rc["slab_checksums"][0] == rc["slab_checksums"][1] and len(rc["slab_checksum_slots"]) == 3
    vr = rc["value_per_rank"]
    assert len(vr["all"]) == 2 and 0 < vr["min"] <= vr["max"] and d["value"] <= 2 * vr["min"] * (1 + 1e-9)
    assert "degraded" not in d
    # round 4: the curve is readable before it is run -- the same-shard N = 1 point is measured in THIS job (rank 0 alone, the
    # other rank idle at a barrier) and the line carries the ratio; the fresh-records leg runs on every rank
    so = d["solo_same_shard"]
    assert so["value"] > 0 and so["launches"] >= 1
    assert abs(d["efficiency_vs_n1_same_shard"] - d["value"] / (2 * so["value"])) < 1e-9
    assert 0.2 < d["efficiency_vs_n1_same_shard"] < 1.5          # (two ranks share ONE GPU here: anything but a scaling figure)
    assert d["fresh_records"]["records_advanced"] == 40 and d["value_fresh_records"] > 0
    assert "c4_shard" not in d and "e2e_upload" not in d          # the N = 1 extras stay out of the N > 1 line


@pytest.mark.gpu
def test_bench_multi_gpu_code_path_with_one_rccl_rank():
    """What `bench.py --gpus N` does beyond one GPU -- RCCL process group, records broadcast in place into the resident
    slots, max-over-ranks reductions, and the short end-to-end segment (one broadcast per record overlapped with the
    stepping, RCCL broadcast and scatter + all-gather) -- rehearsed with ONE RCCL rank (all a one-GPU box allows), checked
    against the oracle by the run's own --check."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SITRK_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, "bench.py", "--config", "c2", "--steps", "96", "--warmup", "8", "--no-cpu-baseline", "--check"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "check OK" in r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert "RCCL broadcast" in d["config"]["records_via"]
    rc = d["rccl"]
    assert rc["world"] == 1 and rc["backend"].startswith("rccl") and rc["ranks"][0]["rank"] == 0 and rc["distinct_devices"] == 1
    assert rc["slab_checksum_ok"] is True and rc["slab_checksum_matches_source"] is True and "degraded" not in d
    assert d["solo_same_shard"]["value"] > 0 and 0.5 < d["efficiency_vs_n1_same_shard"] < 1.5      # one rank: the two legs are the same run
    e = d["e2e_broadcast"]
    assert "error" not in e, e
    for mode in ("broadcast", "scatter_allgather", "box_broadcast"):
        assert e[mode]["ms_per_step"] > 0 and e[mode]["particle_steps_per_s"] > 0, e[mode]
    assert 0.2 < e["box_broadcast"]["share_of_slab"] < 0.9
    assert e["slab_bytes"] == 3 * 512 * 512 * 4


@pytest.mark.gpu
def test_bench_line_survives_a_stuck_end_to_end_segment():
    """The extra segment of `bench.py --gpus N` runs under a watchdog: when it does not come back in time (here: a wait of
    a millisecond) rank 0 still prints the line with the resident numbers and the segment marked as timed out -- and the
    process leaves with a NON-ZERO code (3): a stuck communicator must not look like a clean run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SITRK_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), SITRK_E2E_SEGMENT_TIMEOUT="0.001")
    r = subprocess.run([sys.executable, "bench.py", "--config", "c2", "--steps", "64", "--warmup", "8", "--no-cpu-baseline"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 3, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert "timed out" in d["e2e_broadcast"]["error"] and d["value"] > 0 and d["roofline"]["launches"] >= 2
    assert d["rccl"]["slab_checksum_ok"] is True


@pytest.mark.gpu
def test_bench_broadcast_failure_is_visible_one_rccl_rank():
    """A record broadcast that raises: no local regeneration behind the driver's back -- the line is printed with
    "degraded" and value null, the exit code is 4."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SITRK_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), SITRK_BENCH_FAIL_BCAST="0")
    r = subprocess.run([sys.executable, "bench.py", "--config", "c2", "--steps", "64", "--warmup", "8", "--no-cpu-baseline"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 4, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] is None and "record broadcast failed on rank 0" in d["degraded"] and d["phase"] == "record broadcast"
    assert d["metric"] == "particle-steps/s" and d["n_gpus"] == 1 and d["config"]["workload"].startswith("C2")


@pytest.mark.gpu
def test_bench_broadcast_failure_on_another_rank_is_visible():
    """Two ranks as the driver launches them (gloo on the one GPU); the broadcast fails on rank 1 only.  Rank 1 leaves
    non-zero, the launcher terminates rank 0 -- which may sit inside the collective -- and rank 0 still prints ONE line
    with "degraded" and value null; the launcher's exit code is non-zero."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SITRK_DIST_BACKEND="gloo", SITRK_DEVICE="0", SITRK_BENCH_FAIL_BCAST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "40", "--warmup", "8", "--config", "c2",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["value"] is None and d["degraded"] and d["n_gpus"] == 2


def _run_guard_script(body, env_extra=None, timeout=60):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = ("import os, sys, time, signal, json\nsys.path.insert(0, %r)\nimport bench\n"
              "bench._STATE['multi'] = True\nbench._STATE['base'] = {'metric': 'particle-steps/s', 'value': None, 'n_gpus': 2}\n" % root) + body
    env = dict(os.environ, **(env_extra or {}))
    return subprocess.run([sys.executable, "-c", script], cwd=root, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_guards_sigterm_prints_the_degraded_line():
    """rank 0 terminated by the launcher while its main thread is busy in native code (here: a long sleep stands in for a
    collective): the wake-up-pipe watcher prints the line and leaves with code 7."""
    import json
    r = _run_guard_script("bench._start_guards()\nbench._phase('record broadcast')\n"
                          "os.kill(os.getpid(), signal.SIGTERM)\ntime.sleep(30)\nprint('not reached')\n")
    assert r.returncode == 7, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and "not reached" not in r.stdout
    d = json.loads(lines[0])
    assert d["value"] is None and "SIGTERM" in d["degraded"] and d["phase"] == "record broadcast" and d["n_gpus"] == 2


def test_bench_guards_no_progress_watchdog_and_single_line():
    """a phase that never ends: the watchdog prints the line and leaves with code 6; ranks other than 0 print no line;
    and the line is printed at most once however many paths reach _emit."""
    import json
    r = _run_guard_script("bench._start_guards()\nbench._phase('reductions')\ntime.sleep(30)\n", {"SITRK_BENCH_PHASE_TIMEOUT": "0.6"})
    assert r.returncode == 6, r.stdout + r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert "no progress" in d["degraded"] and d["phase"] == "reductions"
    r = _run_guard_script("bench._STATE['rank'] = 1\nbench._start_guards()\ntime.sleep(30)\n", {"SITRK_BENCH_PHASE_TIMEOUT": "0.6"})
    assert r.returncode == 6 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    r = _run_guard_script("assert bench._emit({'a': 1}) is True\nassert bench._emit({'a': 2}) is False\nbench._degraded_exit('x', 4)\n")
    assert r.returncode == 4 and [l for l in r.stdout.splitlines() if l.startswith("{")] == ['{"a": 1}']


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--e2e-full"], ["--e2e-library"], ["--e2e-library", "--e2e-full"]])
def test_bench_end_to_end_regime_checks_against_the_oracle(extra):
    """bench.py --regime e2e (every record uploaded from pinned host memory on a copy stream, double-buffered against the
    stepping on a second stream, row bands or whole records): the run's own --check replays it on the oracle."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "bench.py", "--config", "c2", "--regime", "e2e", "--steps", "60", "--warmup", "5", "--check",
           "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "check OK" in r.stderr, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["regime"] == "e2e" and d["value"] > 0 and d["config"]["e2e_upload_bytes_per_step"] > 0
    full = 3 * 512 * 512 * 4
    assert (d["config"]["e2e_upload_bytes_per_step"] == full) == ("--e2e-full" in extra)
