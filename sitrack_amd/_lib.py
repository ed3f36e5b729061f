"""ctypes binding of libsitrk.so (include/sitrk.h).

The extension is built in-tree (sitrack_amd/csrc/Makefile -> sitrack_amd/libsitrk.so)
and loaded from there.  There is no CPU fallback: if the library is missing, or no
HIP device is usable, the product raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# the in-tree build; SITRK_LIB_PATH points A/B tooling (tools/build_variant.sh) at another build of the same sources
SO_PATH = os.environ.get("SITRK_LIB_PATH") or os.path.join(_HERE, "libsitrk.so")
CSRC = os.path.join(_HERE, "csrc")

SITRK_F32, SITRK_F64 = 0, 1
FillValue = -9999.0

_vp = C.c_void_p
_i64 = C.c_int64
_int = C.c_int
_dbl = C.c_double

# every symbol include/sitrk.h declares: (restype, argtypes)
_SIGNATURES = {
    "sitrk_version": (_int, []),
    "sitrk_create": (_int, [C.POINTER(_vp), _int]),
    "sitrk_destroy": (_int, [_vp]),
    "sitrk_last_error": (C.c_char_p, [_vp]),
    "sitrk_sync": (_int, [_vp]),
    "sitrk_set_stream": (_int, [_vp, _vp]),
    "sitrk_set_grid": (_int, [_vp, _int, _int] + [_vp] * 7),
    "sitrk_set_params": (_int, [_vp, _dbl, _int, _dbl]),
    "sitrk_set_tuning": (_int, [_vp, C.c_char_p, _int]),
    "sitrk_alloc_records": (_int, [_vp, _int, _int]),
    "sitrk_push_record": (_int, [_vp, _int, _vp, _vp, _vp]),
    "sitrk_push_record_dev": (_int, [_vp, _int, _vp]),
    "sitrk_stage_acquire": (_int, [_vp, _int, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "sitrk_stage_submit": (_int, [_vp, _int, _int, _int]),
    "sitrk_stage_release": (_int, [_vp]),
    "sitrk_launch_stats": (_int, [_vp, _int, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "sitrk_buoy_rows": (_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sitrk_push_record_rows": (_int, [_vp, _int, _int, _int, _vp, _vp, _vp]),
    "sitrk_commit_record_rows": (_int, [_vp, _int, _int, _int]),
    "sitrk_buoy_box": (_int, [_vp] + [C.POINTER(C.c_int32)] * 4),
    "sitrk_push_record_box": (_int, [_vp, _int, _int, _int, _int, _int, _vp, _vp, _vp, _i64]),
    "sitrk_buoy_box_begin": (_int, [_vp]),
    "sitrk_buoy_box_end": (_int, [_vp] + [C.POINTER(C.c_int32)] * 5),
    "sitrk_stage_acquire_box": (_int, [_vp, _int, _int, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "sitrk_stage_submit_box": (_int, [_vp, _int, _int, _int, _int, _int]),
    "sitrk_commit_record_box": (_int, [_vp, _int, _int, _int, _int, _int]),
    "sitrk_commit_records_box": (_int, [_vp, _int, _int, _int, _int, _int, _int]),
    "sitrk_commit_records_box_async": (_int, [_vp, _int, _int, _int, _int, _int, _int]),
    "sitrk_record_ptr": (_vp, [_vp, _int]),
    "sitrk_commit_record": (_int, [_vp, _int]),
    "sitrk_set_buoys": (_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "sitrk_restore_state": (_int, [_vp, _vp, _vp]),
    "sitrk_sort_buoys": (_int, [_vp]),
    "sitrk_set_resort": (_int, [_vp, _int]),
    "sitrk_step": (_int, [_vp, _int, _int]),
    "sitrk_run": (_int, [_vp, _int, _int, _int]),
    "sitrk_fetch": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "sitrk_fetch_record": (_int, [_vp, _int, _vp, _vp, _vp]),
    "sitrk_count_alive": (_int, [_vp, C.POINTER(_i64)]),
    "sitrk_find_cells": (_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "sitrk_seed_init": (_int, [_vp, _i64] + [_vp] * 9),
    "sitrk_nemo_seed": (_int, [_vp, _int, _int, _int] + [_vp] * 7 + [_dbl, _dbl, _i64, _vp, _vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "sitrk_nearest_point": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _dbl, _int, _vp, _vp]),
    "sitrk_eval_haversine": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "sitrk_eval_inside": (_int, [_vp, _i64, _vp, _vp, _vp]),
    "sitrk_eval_euler": (_int, [_vp, _i64, _vp, _vp, _dbl, _vp]),
    "sitrk_eval_intersect": (_int, [_vp, _i64, _vp, _vp, _vp]),
    "sitrk_eval_crossing": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "sitrk_survive_mask": (_int, [_vp, _vp, _vp]),
    "sitrk_cart2geo": (_int, [_vp, _i64, _vp, _dbl, _dbl, _vp]),
    "sitrk_geo2cart": (_int, [_vp, _i64, _vp, _dbl, _dbl, _vp]),
    "sitrk_timer_start": (_int, [_vp]),
    "sitrk_timer_stop": (_int, [_vp, C.POINTER(C.c_float)]),
}


class SitrkError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """Compile libsitrk.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC] + (["-B"] if force else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout + r.stderr)
    if r.returncode:
        raise SitrkError("building libsitrk.so failed (hipcc --offload-arch=gfx950)")
    return SO_PATH


_lib = None


def _one_hip_runtime():
    """A process must run on ONE HIP runtime.  PyTorch-ROCm wheels bundle their own (same soname as /opt/rocm's), and
    whichever copy is loaded first serves both libsitrk.so and torch: with the system's copy first, torch's bundled
    RCCL/HSA libraries come up next to it and the runtime that initialises second sees no device.  So if PyTorch is
    installed and not loaded yet, its copy is loaded before libsitrk.so (found without importing torch;
    SITRK_HIP_RUNTIME=system keeps the system's)."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("SITRK_HIP_RUNTIME", "") == "system":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    for loc in (spec.submodule_search_locations or []) if spec else []:
        p = os.path.join(loc, "lib", "libamdhip64.so")
        if os.path.exists(p):
            try:
                C.CDLL(p, mode=C.RTLD_GLOBAL)
            except OSError:
                pass
            return


def lib():
    """Load the in-tree extension; raise loudly if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise SitrkError("%s not found: build it with `make -C %s` (or __graft_entry__.build()); "
                             "sitrack_amd has no CPU fallback" % (SO_PATH, CSRC))
        _one_hip_runtime()
        L = C.CDLL(SO_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


def as_c(a, dtype, shape=None, name="array"):
    """C-contiguous array of `dtype` (copy only when needed); validates the shape."""
    b = np.ascontiguousarray(a, dtype=dtype)
    if shape is not None and tuple(b.shape) != tuple(shape):
        raise ValueError("%s: expected shape %s, got %s" % (name, tuple(shape), tuple(b.shape)))
    return b


class Context:
    """Owns one sitrk_t handle (= one GPU)."""

    def __init__(self, device=0):
        self._L = lib()
        h = _vp()
        rc = self._L.sitrk_create(C.byref(h), int(device))
        if rc:
            raise SitrkError("sitrk_create: %s" % self._L.sitrk_last_error(None).decode())
        self._h = h
        self.device = int(device)
        self.Nj = self.Ni = 0
        self.nP = 0
        self.nslots = 0
        self.field_dtype = None

    # -- plumbing
    def _chk(self, rc):
        if rc:
            msg = self._L.sitrk_last_error(self._h).decode()
            if rc == -2:
                raise IndexError(msg)
            raise SitrkError(msg)

    def close(self):
        if getattr(self, "_h", None):
            self._L.sitrk_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def sync(self):
        self._chk(self._L.sitrk_sync(self._h))

    def set_stream(self, hip_stream):
        self._chk(self._L.sitrk_set_stream(self._h, hip_stream))

    # -- grid / params / records
    def set_grid(self, Yf, Xf, Yu, Xu, Yv, Xv, tmask):
        Yf = as_c(Yf, np.float64)
        Nj, Ni = Yf.shape
        arrs = [Yf] + [as_c(a, np.float64, (Nj, Ni), n) for a, n in
                       ((Xf, "Xf"), (Yu, "Yu"), (Xu, "Xu"), (Yv, "Yv"), (Xv, "Xv"))]
        tm = as_c(tmask, np.int8, (Nj, Ni), "tmask")
        self._chk(self._L.sitrk_set_grid(self._h, Nj, Ni, *[_ptr(a) for a in arrs], _ptr(tm)))
        self.Nj, self.Ni = Nj, Ni
        self.nP = 0
        self.nslots = 0

    def set_params(self, rdt=3600., uv_strategy=1, rmin_conc=0.1):
        self._chk(self._L.sitrk_set_params(self._h, float(rdt), int(uv_strategy), float(rmin_conc)))

    def set_tuning(self, **knobs):
        for k, v in knobs.items():
            self._chk(self._L.sitrk_set_tuning(self._h, k.encode(), int(v)))

    def alloc_records(self, nslots, dtype=np.float32):
        dt = np.dtype(dtype)
        if dt not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError("records must be float32 or float64")
        self._chk(self._L.sitrk_alloc_records(self._h, int(nslots), SITRK_F64 if dt == np.float64 else SITRK_F32))
        self.nslots = int(nslots)
        self.field_dtype = dt

    def push_record(self, slot, u, v, sic):
        shp = (self.Nj, self.Ni)
        u = as_c(u, self.field_dtype, shp, "u")
        v = as_c(v, self.field_dtype, shp, "v")
        sic = as_c(sic, self.field_dtype, shp, "sic")
        # asynchronous: the library copies the fields into its pinned staging before returning (temporaries are fine)
        self._chk(self._L.sitrk_push_record(self._h, int(slot), _ptr(u), _ptr(v), _ptr(sic)))

    def stage(self, nrows=None, ncols=None):
        """The library's next pinned staging buffer as three (nrows, ncols) arrays of the records' dtype (ncols = Ni unless
        a box is staged): read the record straight into them, then submit(slot, j0[, i0]).  The arrays are VIEWS of pinned
        host memory the library owns: valid until submit() / stage_release(), and dangling after alloc_records(),
        set_grid() or close(), which free it.  Prefer stage_fill(), which also releases the buffer when the read fails."""
        nrows = self.Nj if nrows is None else int(nrows)
        ncols = self.Ni if ncols is None else int(ncols)
        pu, pv, ps = _vp(), _vp(), _vp()
        if ncols == self.Ni:
            self._chk(self._L.sitrk_stage_acquire(self._h, nrows, C.byref(pu), C.byref(pv), C.byref(ps)))
        else:
            self._chk(self._L.sitrk_stage_acquire_box(self._h, nrows, ncols, C.byref(pu), C.byref(pv), C.byref(ps)))
        nb = nrows * ncols * self.field_dtype.itemsize

        def view(p):
            return np.frombuffer((C.c_char * nb).from_address(p.value), dtype=self.field_dtype).reshape(nrows, ncols)
        self._staged_rows, self._staged_cols = nrows, ncols
        return view(pu), view(pv), view(ps)

    def stage_release(self):
        """Give the buffer handed out by stage() back without uploading it (the read into it failed): the views must not
        be used any more, the next stage() hands out the same buffer."""
        self._chk(self._L.sitrk_stage_release(self._h))

    def stage_fill(self, slot, j0, nrows, fill, i0=0, ncols=None):
        """stage() + fill(u, v, sic) + submit(slot, j0, i0), exception safe: if `fill` raises, the buffer is released, so the
        context stays usable (a later stage()/push_record is not refused with 'not submitted')."""
        bufs = self.stage(nrows, ncols)
        try:
            fill(*bufs)
        except BaseException:
            self.stage_release()
            raise
        finally:
            del bufs                                     # the views die with the hand-out
        self.submit(slot, j0, i0)

    def submit(self, slot, j0=0, i0=0):
        """Queue the staged box as rows [j0, j0+nrows) x columns [i0, i0+ncols) of `slot` (asynchronous, see
        sitrk_stage_submit / sitrk_stage_submit_box)."""
        j1, i1 = int(j0) + int(self._staged_rows), int(i0) + int(self._staged_cols)
        if int(i0) == 0 and i1 == self.Ni:
            self._chk(self._L.sitrk_stage_submit(self._h, int(slot), int(j0), j1))
        else:
            self._chk(self._L.sitrk_stage_submit_box(self._h, int(slot), int(j0), j1, int(i0), i1))

    def launch_stats(self, reset=False):
        a, b, c = _i64(0), _i64(0), _i64(0)
        self._chk(self._L.sitrk_launch_stats(self._h, int(bool(reset)), C.byref(a), C.byref(b), C.byref(c)))
        return {"fused_launches": a.value, "fused_records": b.value, "step_launches": c.value}

    def buoy_rows(self):
        """(jmin, jmax) of the host rows of the buoys still alive; jmin > jmax when there is none."""
        a, b = C.c_int32(0), C.c_int32(0)
        self._chk(self._L.sitrk_buoy_rows(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def band(self, age=0):
        """Rows [j0,j1) of a record the next step(s) can touch: [jmin-2-age, jmax+3+age) clipped to the grid, where `age`
        = number of records stepped since buoy_rows() was evaluated (a host cell moves at most one row per record)."""
        jmin, jmax = self.buoy_rows()
        if jmin > jmax:
            return 0, 0
        return max(0, jmin - 2 - age), min(self.Nj, jmax + 3 + age)

    def buoy_box(self):
        """(jmin, jmax, imin, imax) of the host cells of the buoys still alive; jmin > jmax when there is none."""
        v = [C.c_int32(0) for _ in range(4)]
        self._chk(self._L.sitrk_buoy_box(self._h, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def box(self, age=0, align=4):
        """The box (j0, j1, i0, i1) of a record the next step(s) can touch: rows [jmin-2-age, jmax+3+age) x columns
        [imin-2-age, imax+3+age) clipped to the grid, the columns widened to multiples of `align` (16-byte lines of an fp32
        row).  `age` = records stepped since (a host cell moves at most one row and one column per record)."""
        jmin, jmax, imin, imax = self.buoy_box()
        return self.box_of(jmin, jmax, imin, imax, age, align)

    def box_of(self, jmin, jmax, imin, imax, age=0, align=4):
        if jmin > jmax:
            return 0, 0, 0, 0
        i0, i1 = max(0, imin - 2 - age), min(self.Ni, imax + 3 + age)
        i0 -= i0 % align
        i1 = min(self.Ni, -(-i1 // align) * align)
        return max(0, jmin - 2 - age), min(self.Nj, jmax + 3 + age), i0, i1

    def push_record_box(self, slot, j0, j1, i0, i1, u_box, v_box, sic_box):
        """The box rows [j0,j1) x columns [i0,i1) of a record from three (j1-j0, i1-i0) arrays.  Views into whole fields
        (`u[j0:j1, i0:i1]`: rows contiguous, one common row pitch) are handed over as they are -- the library gathers
        them into its pinned staging -- anything else is made contiguous first."""
        shp = (j1 - j0, i1 - i0)
        es = self.field_dtype.itemsize
        arrs = [np.asarray(x) for x in (u_box, v_box, sic_box)]
        for x, n in zip(arrs, ("u box", "v box", "sic box")):
            if tuple(x.shape) != shp:
                raise ValueError("%s: expected shape %s, got %s" % (n, shp, tuple(x.shape)))
        pitched = all(x.dtype == self.field_dtype and x.ndim == 2 and x.strides[1] == es and x.strides[0] % es == 0
                      and x.strides[0] >= shp[1] * es for x in arrs) and len({x.strides[0] for x in arrs}) == 1
        if pitched and shp[0] > 0 and shp[1] > 0:
            ld = arrs[0].strides[0] // es
        else:
            arrs = [as_c(x, self.field_dtype, shp) for x in arrs]
            ld = shp[1]
        self._chk(self._L.sitrk_push_record_box(self._h, int(slot), int(j0), int(j1), int(i0), int(i1), *[_ptr(x) for x in arrs], int(ld)))

    def buoy_box_begin(self):
        """queue the evaluation of buoy_box() on the compute stream without waiting for it (see sitrk_buoy_box_begin)"""
        self._chk(self._L.sitrk_buoy_box_begin(self._h))

    def buoy_box_end(self):
        """(jmin, jmax, imin, imax, age): the box as it was when buoy_box_begin() was queued, and the records stepped since"""
        v = [C.c_int32(0) for _ in range(5)]
        self._chk(self._L.sitrk_buoy_box_end(self._h, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def commit_record_box(self, slot, j0, j1, i0, i1):
        self._chk(self._L.sitrk_commit_record_box(self._h, int(slot), int(j0), int(j1), int(i0), int(i1)))

    def commit_records_box(self, slot0, nrec, j0, j1, i0, i1, on_ingest_stream=False):
        """commit_record_box for the nrec slots (slot0 + k) % nslots in one launch; on_ingest_stream: next to the stepping of other
        slots (sitrk_commit_records_box_async: the slabs must be complete in device memory)"""
        fn = self._L.sitrk_commit_records_box_async if on_ingest_stream else self._L.sitrk_commit_records_box
        self._chk(fn(self._h, int(slot0), int(nrec), int(j0), int(j1), int(i0), int(i1)))

    def push_record_rows(self, slot, j0, j1, u_rows, v_rows, sic_rows):
        shp = (j1 - j0, self.Ni)
        u = as_c(u_rows, self.field_dtype, shp, "u rows")
        v = as_c(v_rows, self.field_dtype, shp, "v rows")
        s = as_c(sic_rows, self.field_dtype, shp, "sic rows")
        self._chk(self._L.sitrk_push_record_rows(self._h, int(slot), int(j0), int(j1), _ptr(u), _ptr(v), _ptr(s)))

    def commit_record_rows(self, slot, j0, j1):
        self._chk(self._L.sitrk_commit_record_rows(self._h, int(slot), int(j0), int(j1)))

    def push_record_dev(self, slot, dev_ptr):
        self._chk(self._L.sitrk_push_record_dev(self._h, int(slot), _vp(dev_ptr)))

    def record_ptr(self, slot):
        p = self._L.sitrk_record_ptr(self._h, int(slot))
        if not p:
            raise SitrkError("sitrk_record_ptr: bad slot %d" % slot)
        return p

    def commit_record(self, slot):
        self._chk(self._L.sitrk_commit_record(self._h, int(slot)))

    @property
    def slab_elems(self):
        return 3 * self.Nj * self.Ni

    # -- buoys
    def set_buoys(self, yx, jiT, rec_first=None, rec_last=None, sort=True):
        yx = as_c(yx, np.float64)
        nP = yx.shape[0]
        yx = as_c(yx, np.float64, (nP, 2), "yx")
        ji = np.asarray(jiT)
        if ji.shape != (nP, 2):
            raise ValueError("jiT: expected shape %s, got %s" % ((nP, 2), ji.shape))
        ji32 = as_c(ji, np.int32)
        if not np.array_equal(ji32, ji):
            raise IndexError("jiT does not fit int32")
        f = None if rec_first is None else as_c(rec_first, np.int32, (nP,), "rec_first")
        l = None if rec_last is None else as_c(rec_last, np.int32, (nP,), "rec_last")
        self._chk(self._L.sitrk_set_buoys(self._h, nP, _ptr(yx), _ptr(ji32), _ptr(f), _ptr(l)))
        self.nP = nP
        if sort:
            self.sort_buoys()

    def restore_state(self, alive, kill_rec):
        """after set_buoys(sort=False): dead flags and kill records of buoys that were stepped elsewhere before"""
        al = as_c(alive, np.int8, (self.nP,), "alive")
        kr = as_c(kill_rec, np.int32, (self.nP,), "kill_rec")
        self._chk(self._L.sitrk_restore_state(self._h, _ptr(al), _ptr(kr)))

    def sort_buoys(self):
        self._chk(self._L.sitrk_sort_buoys(self._h))

    def set_resort(self, every):
        self._chk(self._L.sitrk_set_resort(self._h, int(every)))

    def step(self, slot, jrec):
        self._chk(self._L.sitrk_step(self._h, int(slot), int(jrec)))

    def run(self, slot0, jrec0, nsteps):
        self._chk(self._L.sitrk_run(self._h, int(slot0), int(jrec0), int(nsteps)))

    def fetch(self, want=("yx", "jiT", "alive", "kill_rec")):
        nP = self.nP
        out = {}
        if "yx" in want:
            out["yx"] = np.empty((nP, 2), dtype=np.float64)
        if "jiT" in want:
            out["jiT"] = np.empty((nP, 2), dtype=np.int32)
        if "alive" in want:
            out["alive"] = np.empty(nP, dtype=np.int8)
        if "kill_rec" in want:
            out["kill_rec"] = np.empty(nP, dtype=np.int32)
        self._chk(self._L.sitrk_fetch(self._h, _ptr(out.get("yx")), _ptr(out.get("jiT")), _ptr(out.get("alive")),
                                      _ptr(out.get("kill_rec"))))
        return out

    def fetch_record(self, jrec, latlon=False):
        nP = self.nP
        yx = np.empty((nP, 2), dtype=np.float64)
        mask = np.empty(nP, dtype=np.int8)
        ll = np.empty((nP, 2), dtype=np.float64) if latlon else None
        self._chk(self._L.sitrk_fetch_record(self._h, int(jrec), _ptr(yx), _ptr(mask), _ptr(ll)))
        return (yx, mask, ll) if latlon else (yx, mask)

    def count_alive(self):
        n = _i64(0)
        self._chk(self._L.sitrk_count_alive(self._h, C.byref(n)))
        return n.value

    # -- locate / projection
    def find_cells(self, yx, jiT_guess):
        yx = as_c(yx, np.float64)
        n = yx.shape[0]
        g = as_c(jiT_guess, np.int32, (n, 2), "jiT_guess")
        out = np.empty((n, 2), dtype=np.int32)
        found = np.empty(n, dtype=np.int8)
        self._chk(self._L.sitrk_find_cells(self._h, n, _ptr(yx), _ptr(g), _ptr(out), _ptr(found)))
        return found.astype(bool), out

    def seed_init(self, latlon, yx, latT, lonT, resolkm, sic):
        latlon = as_c(latlon, np.float64)
        nP = latlon.shape[0]
        yx = as_c(yx, np.float64, (nP, 2), "pSC")
        shp = (self.Nj, self.Ni)
        latT = as_c(latT, np.float64, shp, "latT")
        lonT = as_c(lonT, np.float64, shp, "lonT")
        res = None if resolkm is None else as_c(resolkm, np.float64, shp, "resolkm")
        sic = as_c(sic, np.float64, shp, "sic")
        jiT = np.zeros((nP, 2), dtype=np.int32)
        keep = np.zeros(nP, dtype=np.int8)
        why = np.zeros(nP, dtype=np.int8)
        self._chk(self._L.sitrk_seed_init(self._h, nP, _ptr(latlon), _ptr(yx), _ptr(latT), _ptr(lonT), _ptr(res), _ptr(sic),
                                          _ptr(jiT), _ptr(keep), _ptr(why)))
        return jiT, keep, why

    def nemo_seed(self, tmask, latT, lonT, sic, khss=1, rmask=None, latF=None, lonF=None, lat0=70., lon0=-45.):
        """sitrk_nemo_seed: (latlon (n,2), yx (n,2) km, nT, nF) -- T-seeds first, then F-seeds, each in C order."""
        tm = as_c(tmask, np.int8)
        Nj, Ni = tm.shape
        shp = (Nj, Ni)
        la, lo, ic = (as_c(x, np.float64, shp, n) for x, n in ((latT, "latT"), (lonT, "lonT"), (sic, "sic")))
        rm = None if rmask is None else as_c(rmask, np.int8, shp, "rmask")
        lf = None if latF is None else as_c(latF, np.float64, shp, "latF")
        of = None if lonF is None else as_c(lonF, np.float64, shp, "lonF")
        nT, nF = _i64(0), _i64(0)
        args = (self._h, Nj, Ni, int(khss), _ptr(tm), _ptr(rm), _ptr(la), _ptr(lo), _ptr(ic), _ptr(lf), _ptr(of), float(lat0), float(lon0))
        self._chk(self._L.sitrk_nemo_seed(*args, 0, None, None, C.byref(nT), C.byref(nF)))
        n = nT.value + nF.value
        ll = np.empty((n, 2), dtype=np.float64)
        yx = np.empty((n, 2), dtype=np.float64)
        if n:
            self._chk(self._L.sitrk_nemo_seed(*args, n, _ptr(ll), _ptr(yx), C.byref(nT), C.byref(nF)))
        return ll, yx, nT.value, nF.value

    def nearest_point(self, latlon, latT, lonT, resolkm=None, rd_found_km=10., max_itr=5):
        """NearestPoint of the reference for an array of points: (ji (n,2) int32 with -1,-1 = not found, dmin km)."""
        latlon = as_c(latlon, np.float64)
        n = latlon.shape[0]
        latlon = as_c(latlon, np.float64, (n, 2), "latlon")
        latT = as_c(latT, np.float64, (self.Nj, self.Ni), "latT")
        lonT = as_c(lonT, np.float64, (self.Nj, self.Ni), "lonT")
        res = None if resolkm is None else as_c(resolkm, np.float64, (self.Nj, self.Ni), "resolkm")
        ji = np.empty((n, 2), dtype=np.int32)
        dmin = np.empty(n, dtype=np.float64)
        self._chk(self._L.sitrk_nearest_point(self._h, n, _ptr(latlon), _ptr(latT), _ptr(lonT), _ptr(res), float(rd_found_km),
                                              int(max_itr), _ptr(ji), _ptr(dmin)))
        return ji, dmin

    def eval_haversine(self, plat, plon, xlat, xlon):
        plat, plon, xlat, xlon = (np.ascontiguousarray(a, dtype=np.float64) for a in np.broadcast_arrays(plat, plon, xlat, xlon))
        out = np.empty(plat.shape, dtype=np.float64)
        self._chk(self._L.sitrk_eval_haversine(self._h, plat.size, _ptr(plat), _ptr(plon), _ptr(xlat), _ptr(xlon), _ptr(out)))
        return out

    # -- predicate probes (parity tests)
    def eval_inside(self, pts, quads):
        pts = as_c(pts, np.float64)
        n = pts.shape[0]
        quads = as_c(quads, np.float64, (n, 4, 2), "quads")
        out = np.empty(n, dtype=np.int8)
        self._chk(self._L.sitrk_eval_inside(self._h, n, _ptr(pts), _ptr(quads), _ptr(out)))
        if (out & 2).any():
            raise SitrkError("eval_inside: the division-free cell test and the plain one disagree for %d point(s), first at %d"
                             % (int((out & 2).astype(bool).sum()), int(np.flatnonzero(out & 2)[0])))
        return out.astype(bool)

    def eval_euler(self, r, vel, rdt=3600.):
        r = as_c(r, np.float64)
        vel = as_c(vel, np.float64, r.shape, "vel")
        out = np.empty_like(r)
        self._chk(self._L.sitrk_eval_euler(self._h, r.size, _ptr(r), _ptr(vel), float(rdt), _ptr(out)))
        return out

    def eval_intersect(self, segs):
        segs = as_c(segs, np.float64)
        n = segs.shape[0]
        segs = as_c(segs, np.float64, (n, 4, 2), "segs")
        inter = np.empty(n, dtype=np.int8)
        ccw = np.empty(n, dtype=np.int8)
        self._chk(self._L.sitrk_eval_intersect(self._h, n, _ptr(segs), _ptr(inter), _ptr(ccw)))
        return inter.astype(bool), ccw.astype(bool)

    def eval_crossing(self, P1, P2, jiT):
        P1 = as_c(P1, np.float64)
        n = P1.shape[0]
        P2 = as_c(P2, np.float64, (n, 2), "P2")
        ji = as_c(jiT, np.int32, (n, 2), "jiT")
        out = np.empty((n, 2), dtype=np.int32)
        codes = np.empty((n, 2), dtype=np.int32)
        self._chk(self._L.sitrk_eval_crossing(self._h, n, _ptr(P1), _ptr(P2), _ptr(ji), _ptr(out), _ptr(codes)))
        return out, codes[:, 0], codes[:, 1]

    def survive_mask(self, sic):
        sic = as_c(sic, np.float64, (self.Nj, self.Ni), "sic")
        out = np.empty((self.Nj, self.Ni), dtype=np.int8)
        self._chk(self._L.sitrk_survive_mask(self._h, _ptr(sic), _ptr(out)))
        return out

    def cart2geo(self, yx, lat0=70., lon0=-45.):
        yx = as_c(yx, np.float64)
        out = np.empty_like(yx)
        self._chk(self._L.sitrk_cart2geo(self._h, yx.shape[0], _ptr(yx), float(lat0), float(lon0), _ptr(out)))
        return out

    def geo2cart(self, latlon, lat0=70., lon0=-45.):
        g = as_c(latlon, np.float64)
        out = np.empty_like(g)
        self._chk(self._L.sitrk_geo2cart(self._h, g.shape[0], _ptr(g), float(lat0), float(lon0), _ptr(out)))
        return out

    # -- measurement
    def timer_start(self):
        self._chk(self._L.sitrk_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float(0)
        self._chk(self._L.sitrk_timer_stop(self._h, C.byref(ms)))
        return ms.value
