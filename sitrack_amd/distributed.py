"""Multi-GPU plumbing: one process per GPU, buoy-range partition, record broadcast.

Buoys never interact (no term of reference si3_part_tracker.py:378-490 reads another
buoy's state), so rank r owns the r-th contiguous range of the buoy arrays and steps it
with its own context.  Geometry is replicated.  The path's single exchange step is the
broadcast of each model record's `[u|v|siconc]` slab from the rank that ingested it
(reference :372-374 reads it from NetCDF) -- `torch.distributed.broadcast`, backend
"nccl" (= RCCL over xGMI) straight into the resident slot of libsitrk, or "gloo" on host
arrays for CPU rehearsals.  Results are gathered back in the caller's buoy order.
"""
import numpy as np


def buoy_range(nP, rank, world):
    """Contiguous [lo,hi) of rank `rank`: sizes differ by at most one, order preserved."""
    base, rem = divmod(int(nP), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def all_ranges(nP, world):
    return [buoy_range(nP, r, world) for r in range(world)]


class _DeviceSpan:
    """Exposes a raw device pointer through __cuda_array_interface__ so that torch can
    wrap libsitrk's record slot without a copy."""

    def __init__(self, ptr, nelem, typestr):
        self.__cuda_array_interface__ = {"shape": (int(nelem),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def slot_tensor(ctx, slot):
    """torch view (1-D, 3*Nj*Ni elements) of a resident record slot of `ctx`."""
    import torch
    typestr = "<f8" if ctx.field_dtype == np.dtype(np.float64) else "<f4"
    return torch.as_tensor(_DeviceSpan(ctx.record_ptr(slot), ctx.slab_elems, typestr), device="cuda:%d" % ctx.device)


def pack_slab(u, v, sic, dtype):
    """[u|v|siconc] as one contiguous 1-D array (the layout of a record slot)."""
    return np.concatenate([np.ascontiguousarray(u, dtype=dtype).ravel(), np.ascontiguousarray(v, dtype=dtype).ravel(),
                           np.ascontiguousarray(sic, dtype=dtype).ravel()])


def broadcast_record(ctx, slot, slab_host, src=0, group=None, async_op=False):
    """Broadcast one record slab into slot `slot` of every rank's context.

    `slab_host`: packed [u|v|siconc] numpy array on rank `src`, ignored elsewhere.
    With backend nccl the broadcast writes the resident slot directly (no staging copy).
    Returns the torch work handle when async_op, else None (stream-synchronised)."""
    import os
    import torch
    import torch.distributed as dist
    staged = False
    try:
        t = slot_tensor(ctx, slot)                     # zero-copy view of the resident slot
    except Exception:                                  # noqa: BLE001 -- e.g. a torch build without __cuda_array_interface__ import
        t = torch.empty(ctx.slab_elems, dtype=torch.float64 if ctx.field_dtype == np.dtype(np.float64) else torch.float32,
                        device="cuda:%d" % ctx.device)
        staged = True
    if dist.get_rank(group) == src:
        t.copy_(torch.from_numpy(slab_host), non_blocking=False)
    if os.environ.get("SITRK_BCAST", "") == "scatter_allgather" and not async_op and dist.get_world_size(group) > 2:
        scatter_allgather(t, src=src, group=group)
        work = None
    else:
        work = dist.broadcast(t, src=src, group=group, async_op=async_op and not staged)
    if async_op and not staged:
        return work                    # caller: work.wait(), order the streams, then ctx.commit_record(slot)
    torch.cuda.current_stream().synchronize()
    if staged:
        ctx.push_record_dev(slot, t.data_ptr())        # device-to-device copy into the slot + Survive mask
        ctx.sync()
    else:
        ctx.commit_record(slot)        # derive the record's Survive mask from the new slab
    return None


def scatter_allgather(t, src=0, group=None):
    """Broadcast of the 1-D tensor `t` as scatter + all-gather: the root sends a different 1/N-th of the slab to every
    rank and the ranks then exchange their pieces, so every xGMI link of the full mesh carries 1/N-th of the bytes
    instead of one link carrying the whole slab (~2S/(7B) instead of S/B for a ring at N = 8).
    Opt-in (env SITRK_BCAST=scatter_allgather): whether it beats RCCL's own broadcast is for the 8-GPU node to say.
    Works with any backend; in place on every rank."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = t.numel()
    chunk = (n + world - 1) // world
    pad = chunk * world
    buf = t if pad == n else torch.empty(pad, dtype=t.dtype, device=t.device)
    if buf is not t and rank == src:
        buf[:n].copy_(t)
    pieces = [buf[r * chunk:(r + 1) * chunk] for r in range(world)]
    mine = torch.empty(chunk, dtype=t.dtype, device=t.device)
    dist.scatter(mine, scatter_list=pieces if rank == src else None, src=src, group=group)
    if hasattr(dist, "all_gather_into_tensor") and t.is_cuda:
        dist.all_gather_into_tensor(buf, mine, group=group)
    else:
        dist.all_gather(pieces, mine, group=group)
    if buf is not t:
        t.copy_(buf[:n])
    return t


def union_box(box, Nj, Ni, group=None, device=None):
    """The smallest box that contains every rank's box (j0, j1, i0, i1) (empty boxes -- j0 == j1 -- do not count): what ONE
    broadcast has to carry so that every rank finds the cells its own buoys can touch.  One all-reduce of four integers."""
    import torch
    import torch.distributed as dist
    j0, j1, i0, i1 = box
    empty = j1 <= j0 or i1 <= i0
    t = torch.tensor([-(Nj if empty else j0), 0 if empty else j1, -(Ni if empty else i0), 0 if empty else i1], dtype=torch.int64,
                     device=device or ("cuda" if dist.get_backend(group) == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    j0, j1, i0, i1 = -int(t[0]), int(t[1]), -int(t[2]), int(t[3])
    return (0, 0, 0, 0) if (j1 <= j0 or i1 <= i0) else (j0, j1, i0, i1)


def broadcast_box(slot_t, Nj, Ni, box, buf, src=0, group=None):
    """ONE broadcast of the box rows [j0,j1) x columns [i0,i1) of a record's three fields instead of the whole [u|v|siconc] slab:
    the source packs the box out of its slot `slot_t` (1-D, 3*Nj*Ni) into the contiguous buffer `buf`, the broadcast moves
    3*(j1-j0)*(i1-i0) elements, every other rank unpacks them into its own slot (strided device copies on the current stream).
    The rest of the receivers' slots keeps what it held: commit the slot with sitrk_commit_record_box(box)."""
    import torch.distributed as dist
    j0, j1, i0, i1 = box
    nr, nc = j1 - j0, i1 - i0
    if nr <= 0 or nc <= 0:
        return 0
    view = slot_t.view(3, Nj, Ni)[:, j0:j1, i0:i1]
    flat = buf[:3 * nr * nc]
    packed = flat.view(3, nr, nc)
    if dist.get_rank(group) == src:
        packed.copy_(view)
    dist.broadcast(flat, src=src, group=group)
    if dist.get_rank(group) != src:
        view.copy_(packed)
    return flat.numel() * flat.element_size()


def broadcast_record_host(slab_host, nelem, dtype, src=0, group=None):
    """gloo rehearsal of the same exchange on host memory: returns the slab on every rank."""
    import torch
    import torch.distributed as dist
    if dist.get_rank(group) == src:
        t = torch.from_numpy(np.ascontiguousarray(slab_host, dtype=dtype))
    else:
        t = torch.from_numpy(np.empty(nelem, dtype=dtype))
    dist.broadcast(t, src=src, group=group)
    return t.numpy()


def gather_ranges(local, nP, group=None, dst=0):
    """Gather per-rank arrays (first axis = the rank's buoy range of `nP`, buoy_range()) back into buoy order on `dst`.

    A TENSOR gather: every rank's rows travel as one contiguous buffer (padded to the longest range; the ranges are a
    function of (nP, world) alone, so no sizes are exchanged) -- through RCCL as device tensors with backend nccl,
    through gloo as host tensors.  At C4 a gathered record is 1.6 GB of positions; as pickled objects
    (`gather_object`) it would be serialised on every rank and deserialised on rank 0."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    ranges = all_ranges(nP, world)
    lo, hi = ranges[rank]
    a = np.ascontiguousarray(local)
    if a.shape[0] != hi - lo:
        raise ValueError("gather_ranges: rank %d holds %d rows, its range of %d buoys has %d" % (rank, a.shape[0], nP, hi - lo))
    nmax = max(h - l for l, h in ranges)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.zeros((nmax,) + a.shape[1:], dtype=torch.from_numpy(a[:0]).dtype, device=dev)
    if hi > lo:
        t[:hi - lo].copy_(torch.from_numpy(a))
    bufs = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
    dist.gather(t, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    out = np.empty((nP,) + a.shape[1:], dtype=a.dtype)
    for r, (l, h) in enumerate(ranges):
        if h > l:
            out[l:h] = bufs[r][:h - l].cpu().numpy()
    return out


class _CudaLayer:
    """what RecordBroadcaster needs of a device: streams, events, a current-stream scope, pinned host buffers (= torch.cuda)"""

    def __init__(self, torch):
        self.Stream, self.Event, self.stream = torch.cuda.Stream, torch.cuda.Event, torch.cuda.stream
        self._torch = torch

    def pinned(self, n, dtype):
        return self._torch.empty(n, dtype=dtype).pin_memory()


class RecordBroadcaster:
    """Delivery of whole records to every rank, overlapped with the stepping (driver `--full-records` under torchrun,
    SURVEY 8e): rank 0 copies the record from pinned host memory into its resident slot on a communication stream,
    ONE broadcast of the [u|v|siconc] slab (RCCL over xGMI) writes every other rank's slot in place, and the compute
    stream -- the library's, adopted from torch for the ordering -- only waits for the slot's event before it derives the
    record's Survive mask and steps.  A slot is rewritten only behind the launches that read it.  `deliver()` returns at
    once; the records of the next batch travel while the current batch is stepped with."""

    def __init__(self, ctx, src=0, group=None, dev=None, slot_view=None):
        """`dev`: the stream / event layer -- torch.cuda by default; tests hand in a CPU stand-in with the same four names
        (Stream, Event, stream, and pinned(n, dtype)) so that the slot-recycling protocol runs with a real second rank over gloo.
        `slot_view(ctx, k)`: the resident slot k as a tensor (default: zero-copy view of the library's device memory)."""
        import torch
        import torch.distributed as dist
        self.ctx, self.src, self.group, self.dist, self.torch = ctx, src, group, dist, torch
        self.dev = dev if dev is not None else _CudaLayer(torch)
        self.rank = dist.get_rank(group)
        self.comp, self.comm = self.dev.Stream(), self.dev.Stream()
        ctx.set_stream(self.comp.cuda_stream)
        self.slots = [(slot_view or slot_tensor)(ctx, k) for k in range(ctx.nslots)]
        self.ready = [self.dev.Event() for _ in range(ctx.nslots)]
        self.free = [self.dev.Event() for _ in range(ctx.nslots)]
        self.pinned = [None, None]                       # two staging buffers on the source rank
        self.pin_done = [self.dev.Event(), self.dev.Event()]
        self.npush = 0
        for e in self.free:
            e.record(self.comp)

    def deliver(self, slot, fields):
        """`fields` = (u, v, sic) on the source rank, None elsewhere"""
        with self.dev.stream(self.comm):
            self.comm.wait_event(self.free[slot])        # the launches that read the slot have finished
            if self.rank == self.src:
                b = self.npush % 2
                if self.pinned[b] is None:
                    self.pinned[b] = self.dev.pinned(self.ctx.slab_elems, self.slots[slot].dtype)
                self.pin_done[b].synchronize()           # the copy that last read this staging buffer is done
                n = self.ctx.Nj * self.ctx.Ni
                host = self.pinned[b].numpy()
                for f, a in enumerate(fields):
                    a = np.asarray(a)
                    dst = host[f * n:(f + 1) * n]
                    dst[...] = a.reshape(-1)             # ONE cast, straight into the pinned buffer
                    # the slot's dtype is the context's; a field of another type must survive the cast exactly, as in
                    # IceTracker.load_record / RecordReader.fields_box_into (an f8 siconc next to f4 velocities would
                    # otherwise be rounded silently).  Checked by casting BACK what was just written (no temporary of the
                    # slot's type): this sits on the critical path of every record, in front of the copy and the broadcast.
                    if a.dtype.newbyteorder('=') != host.dtype:
                        if not np.array_equal(dst.astype(a.dtype).reshape(a.shape), a, equal_nan=True):
                            raise ValueError("%s is not exactly representable as %s; allocate float64 records"
                                             % (("u_ice", "v_ice", "siconc")[f], host.dtype))
                self.slots[slot].copy_(self.pinned[b], non_blocking=True)
                self.pin_done[b].record(self.comm)
                self.npush += 1
            self.dist.broadcast(self.slots[slot], src=self.src, group=self.group)
            self.ready[slot].record(self.comm)
        self.ctx.record_ptr(slot)                        # the slab was rewritten in place: its mask must be derived again

    def before_run(self, slots):
        """order the compute stream behind the deliveries of `slots` and derive their Survive masks"""
        for k in slots:
            self.comp.wait_event(self.ready[k])
            self.ctx.commit_record(k)

    def after_run(self, slots):
        for k in slots:
            self.free[k].record(self.comp)

    def close(self):
        self.comp.synchronize(); self.comm.synchronize()
        self.ctx.set_stream(None)


def split_slab(slab, Nj, Ni):
    n = Nj * Ni
    return slab[:n].reshape(Nj, Ni), slab[n:2 * n].reshape(Nj, Ni), slab[2 * n:3 * n].reshape(Nj, Ni)


class Comm:
    """What the driver needs from `torch.distributed`, with a trivial single-process form.

    One process per GPU (`torchrun --nproc-per-node N si3_part_tracker.py ...`).  Backend "nccl" (RCCL over xGMI)
    broadcasts each record slab in place into the resident slot; backend "gloo" (env SITRK_DIST_BACKEND=gloo) moves
    host arrays instead and exists to rehearse the N>1 logic where RCCL cannot run (several ranks on one GPU, CPU)."""

    def __init__(self):
        import os
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = None
        self.dist = None
        # SITRK_FORCE_DIST=1: ONE rank takes every multi-rank code path (process group of one, RCCL collectives with themselves):
        # the rehearsal a one-GPU box allows for the nccl branches, which RCCL refuses to run with two ranks on one device
        force = self.world == 1 and os.environ.get("SITRK_FORCE_DIST") == "1"
        if self.world > 1 or force:
            import torch
            import torch.distributed as dist
            self.backend = os.environ.get("SITRK_DIST_BACKEND", "nccl")
            self.device = int(os.environ.get("SITRK_DEVICE", str(self.local_rank)))
            if force:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29541")
                os.environ.setdefault("RANK", "0")
                os.environ.setdefault("WORLD_SIZE", "1")
            if self.backend == "nccl":
                torch.cuda.set_device(self.device)
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.device))
            else:
                dist.init_process_group(self.backend)
            self.dist = dist
        else:
            self.device = None

    @property
    def multi(self):
        """the multi-rank code paths are in use (several ranks, or one rank rehearsing them)"""
        return self.dist is not None

    @property
    def root(self):
        return self.rank == 0

    def range(self, n):
        return buoy_range(n, self.rank, self.world)

    def bcast_obj(self, obj, src=0):
        if self.dist is None:
            return obj
        box = [obj if self.rank == src else None]
        self.dist.broadcast_object_list(box, src=src)
        return box[0]

    def allgather_obj(self, obj):
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def sum_int(self, n):
        if self.dist is None:
            return int(n)
        return int(sum(self.allgather_obj(int(n))))

    def _dev(self):
        return "cuda" if self.backend == "nccl" else "cpu"

    def allgather_rows(self, arrays):
        """Every rank contributes a tuple of arrays with one common (rank-dependent) number of rows; every rank gets back
        the tuple of their concatenations in rank order.  Tensor collectives (sizes first, then one padded all-gather per
        array): what SeedInit's per-range results travel with -- 10^7..10^8 seeds are GBs, not something to pickle."""
        if self.dist is None:
            return tuple(np.asarray(a) for a in arrays)
        import torch
        n = int(np.shape(arrays[0])[0])
        sizes = [int(x) for x in self.allgather_obj(n)]
        nmax, out = max(sizes), []
        for a in arrays:
            a = np.ascontiguousarray(a)
            t = torch.zeros((nmax,) + a.shape[1:], dtype=torch.from_numpy(a[:0]).dtype, device=self._dev())
            if n:
                t[:n].copy_(torch.from_numpy(a))
            bufs = [torch.empty_like(t) for _ in range(self.world)]
            self.dist.all_gather(bufs, t)
            out.append(np.concatenate([bufs[r][:sizes[r]].cpu().numpy() for r in range(self.world)], axis=0).astype(a.dtype, copy=False))
        return tuple(out)

    def alltoall_rows(self, chunks):
        """`chunks[d]` = tuple of arrays (common number of rows) for rank d; returns, per array position, the
        concatenation of what every rank sent here, in source-rank order.  RCCL: one `all_to_all_single` per array with
        the row counts exchanged first; gloo has no all-to-all: pairwise isend / irecv."""
        import torch
        W, me = self.world, self.rank
        narr = len(chunks[0])
        send_n = [int(np.shape(chunks[d][0])[0]) for d in range(W)]
        if self.dist is None:
            return tuple(np.asarray(a) for a in chunks[0])
        cnt = torch.tensor(send_n, dtype=torch.int64, device=self._dev())
        rcv = torch.empty(W, dtype=torch.int64, device=self._dev())
        if self.backend == "nccl":
            self.dist.all_to_all_single(rcv, cnt)
        else:
            bufs = [torch.empty(W, dtype=torch.int64) for _ in range(W)]
            self.dist.all_gather(bufs, cnt)
            rcv = torch.stack([b[me] for b in bufs])
        recv_n = [int(x) for x in rcv.cpu()]
        out = []
        for k in range(narr):
            parts = [np.ascontiguousarray(chunks[d][k]) for d in range(W)]
            tail, dt = parts[0].shape[1:], parts[0].dtype
            width = int(np.prod(tail)) if tail else 1
            if self.backend == "nccl":
                tin = torch.from_numpy(np.concatenate(parts, axis=0)).cuda()
                tout = torch.empty((sum(recv_n),) + tail, dtype=tin.dtype, device="cuda")
                self.dist.all_to_all_single(tout, tin, output_split_sizes=recv_n, input_split_sizes=send_n)
                out.append(tout.cpu().numpy())
            else:
                got = [np.empty((recv_n[s],) + tail, dtype=dt) for s in range(W)]
                got[me] = parts[me]
                reqs = []
                for off in range(1, W):                       # round `off`: send to me+off, receive from me-off
                    d, src = (me + off) % W, (me - off) % W
                    if send_n[d] * width:
                        reqs.append(self.dist.isend(torch.from_numpy(parts[d]), dst=d))
                    if recv_n[src] * width:
                        reqs.append(self.dist.irecv(torch.from_numpy(got[src]), src=src))
                for r in reqs:
                    r.wait()
                out.append(np.concatenate(got, axis=0))
        return tuple(out)

    def allreduce_sum(self, a):
        """element-wise sum over the ranks of an int64 / float64 array (returned on every rank)"""
        if self.dist is None:
            return np.asarray(a)
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a)).to(self._dev())
        self.dist.all_reduce(t)
        return t.cpu().numpy()

    def bcast_arrays(self, arrays, src=0):
        """tuple of numpy arrays from `src` to everyone as tensors (shapes and dtypes travel as a small object first)"""
        if self.dist is None:
            return arrays
        import torch
        meta = self.bcast_obj([(np.asarray(a).shape, np.asarray(a).dtype.str) for a in arrays] if self.rank == src else None, src)
        out = []
        for k, (shape, dt) in enumerate(meta):
            a = np.ascontiguousarray(arrays[k]) if self.rank == src else np.empty(shape, dtype=np.dtype(dt))
            t = torch.from_numpy(a).to(self._dev()) if a.size else torch.from_numpy(a)
            if a.size:
                self.dist.broadcast(t, src=src)
            out.append(t.cpu().numpy() if a.size else a)
        return tuple(out)

    def gather_rows(self, local, n_total):
        """rank 0: rows of every rank concatenated in rank (= buoy) order; others: None."""
        if self.dist is None:
            return local
        return gather_ranges(local, n_total)

    def deliver_record(self, ctx, slot, fields):
        """`fields` = (u, v, sic) arrays on rank 0, None elsewhere: make the record resident in `slot` everywhere."""
        if self.dist is None:
            ctx.push_record(slot, *fields)
        elif self.backend == "nccl":
            slab = pack_slab(*fields, dtype=ctx.field_dtype) if self.root else None
            broadcast_record(ctx, slot, slab, src=0)
        else:
            slab = pack_slab(*fields, dtype=ctx.field_dtype) if self.root else None
            slab = broadcast_record_host(slab, ctx.slab_elems, ctx.field_dtype, src=0)
            ctx.push_record(slot, *split_slab(slab, ctx.Nj, ctx.Ni))

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
