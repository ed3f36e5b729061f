// sitrk.hip -- C ABI of libsitrk.so (see include/sitrk.h) and kernel launches.
// MI355X / gfx950 only.  No CPU fallback: every entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "../../include/sitrk.h"
#include "sitrk_internal.h"
#include "sitrk_kernels.h"
#include "sitrk_locate.h"
#include "sitrk_seed.h"

using namespace sitrk;

#define SITRK_API extern "C" __attribute__((visibility("default")))

static thread_local char g_create_err[512] = {0};

static int fail(sitrk_ctx *h, int code, const char *fmt, ...)
{
    char *dst = h ? h->err : g_create_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) return fail(h, SITRK_EHIP, "%s -> %s", #call, hipGetErrorString(e_)); \
    } while (0)

#define NEED(cond, msg)                                \
    do {                                               \
        if (!(cond)) return fail(h, SITRK_EINVAL, msg); \
    } while (0)

static inline unsigned nblocks(int64_t n, int bs = kBlock) { return (unsigned)((n + bs - 1) / bs); }
static inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

template <typename T>
static hipError_t dev_alloc(T **p, size_t count)
{
    return hipMalloc((void **)p, count ? count * sizeof(T) : sizeof(T));
}

static void dev_free(void *p)
{
    if (p) (void)hipFree(p);
}

static int ensure_scratch(sitrk_ctx *h, size_t bytes)
{
    if (h->scratch_bytes >= bytes) return SITRK_OK;
    dev_free(h->scratch);
    h->scratch = nullptr;
    h->scratch_bytes = 0;
    HIPCHK(hipMalloc(&h->scratch, bytes));
    h->scratch_bytes = bytes;
    return SITRK_OK;
}

static void free_buoys(sitrk_ctx *h)
{
    for (int b = 0; b < 2; b++) {
        dev_free(h->st[b].pos); dev_free(h->st[b].cell); dev_free(h->st[b].kill_rec);
        dev_free(h->st[b].win); dev_free(h->st[b].perm);
        h->st[b] = BuoyState();
        dev_free(h->keys[b]); dev_free(h->vals[b]);
        h->keys[b] = nullptr; h->vals[b] = nullptr;
    }
    dev_free(h->sort_tmp);
    h->sort_tmp = nullptr; h->sort_tmp_bytes = 0;
    h->nP = 0;
}

static void free_records(sitrk_ctx *h);

// --------------------------------------------------------------------------- context
SITRK_API int sitrk_version(void) { return SITRK_VERSION; }

SITRK_API const char *sitrk_last_error(sitrk_t *h) { return h ? h->err : g_create_err; }

SITRK_API int sitrk_create(sitrk_t **out, int device)
{
    sitrk_ctx *h = nullptr;
    if (!out) return fail(h, SITRK_EINVAL, "sitrk_create: null output pointer");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(h, SITRK_EHIP, "sitrk_create: no HIP device (%s); libsitrk has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(h, SITRK_EINVAL, "sitrk_create: device %d out of range [0,%d)", device, ndev);
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(h, SITRK_EHIP, "hipSetDevice(%d) -> %s", device, hipGetErrorString(e));
    sitrk_ctx *c = new (std::nothrow) sitrk_ctx();
    if (!c) return fail(h, SITRK_ENOMEM, "sitrk_create: out of host memory");
    c->device = device;
    for (int k = 0; k < 4096; k++) c->slot_used_seq[k] = -1;
    e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->sv_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    for (int b = 0; b < sitrk_ctx::kStage && e == hipSuccess; b++) e = hipEventCreateWithFlags(&c->stage_done[b], hipEventDisableTiming);
    for (int k = 0; k < sitrk_ctx::kLaunchRing && e == hipSuccess; k++) e = hipEventCreateWithFlags(&c->launch_ev[k], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->box_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->box_host, 4 * sizeof(int), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void **)&c->counter, 4 * sizeof(unsigned long long));     // reductions: 2 x 4 ints / one 64-bit count
    if (e != hipSuccess) {
        int rc = fail(h, SITRK_EHIP, "sitrk_create: %s", hipGetErrorString(e));
        (void)sitrk_destroy(c);                 // releases whatever was created before the failure
        return rc;
    }
    c->stream = c->own_stream;
    *out = c;
    return SITRK_OK;
}

SITRK_API int sitrk_destroy(sitrk_t *h)
{
    if (!h) return SITRK_OK;
    (void)hipSetDevice(h->device);
    if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    if (h->sv_stream) (void)hipStreamSynchronize(h->sv_stream);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    free_buoys(h);
    free_records(h);
    dev_free(h->geo); dev_free(h->geoF); dev_free(h->orient); dev_free(h->tmask); dev_free(h->scratch); dev_free(h->counter);
    dev_free(h->stamps);
    if (h->box_ev) (void)hipEventDestroy(h->box_ev);
    if (h->box_host) (void)hipHostFree(h->box_host);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    for (int b = 0; b < sitrk_ctx::kStage; b++) if (h->stage_done[b]) (void)hipEventDestroy(h->stage_done[b]);
    for (int k = 0; k < sitrk_ctx::kLaunchRing; k++) if (h->launch_ev[k]) (void)hipEventDestroy(h->launch_ev[k]);
    for (int k = 0; k < 4096; k++) if (h->slot_ready[k]) (void)hipEventDestroy(h->slot_ready[k]);
    for (int k = 0; k < 4096; k++) if (h->slot_sv[k]) (void)hipEventDestroy(h->slot_sv[k]);
    if (h->sv_stream) (void)hipStreamDestroy(h->sv_stream);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return SITRK_OK;
}

SITRK_API int sitrk_sync(sitrk_t *h)
{
    NEED(h, "null handle");
    HIPCHK(hipStreamSynchronize(h->copy_stream));
    HIPCHK(hipStreamSynchronize(h->sv_stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SITRK_OK;
}

SITRK_API int sitrk_set_stream(sitrk_t *h, void *hip_stream)
{
    NEED(h, "null handle");
    HIPCHK(hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return SITRK_OK;
}

// --------------------------------------------------------------------------- grid
SITRK_API int sitrk_set_grid(sitrk_t *h, int Nj, int Ni, const double *Yf, const double *Xf, const double *Yu,
                             const double *Xu, const double *Yv, const double *Xv, const int8_t *tmask)
{
    NEED(h, "null handle");
    NEED(Yf && Xf && Yu && Xu && Yv && Xv && tmask, "sitrk_set_grid: null array");
    if (Nj < 4 || Nj > 32767 || Ni < 4 || Ni > 65535)
        return fail(h, SITRK_EINVAL, "sitrk_set_grid: grid %dx%d outside 4..32767 x 4..65535", Nj, Ni);
    if ((int64_t)Nj * Ni > ((int64_t)1 << 29))      // byte offsets inside one fp64 field stay below 2^32 (CellCtx)
        return fail(h, SITRK_EINVAL, "sitrk_set_grid: grid %dx%d has more than 2^29 cells", Nj, Ni);
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->copy_stream));
    HIPCHK(hipStreamSynchronize(h->sv_stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    dev_free(h->geo); dev_free(h->geoF); dev_free(h->orient); dev_free(h->tmask);
    h->geo = nullptr; h->geoF = nullptr; h->orient = nullptr; h->tmask = nullptr;
    free_records(h);
    free_buoys(h);
    const size_t n = (size_t)Nj * Ni;
    HIPCHK(dev_alloc(&h->geo, n));
    HIPCHK(dev_alloc(&h->geoF, n));
    HIPCHK(dev_alloc(&h->orient, n));
    HIPCHK(dev_alloc(&h->tmask, n));
    // stage the six arrays in scratch, interleave on the device
    int rc = ensure_scratch(h, 6 * n * sizeof(double));
    if (rc) return rc;
    double *s = (double *)h->scratch;
    const double *src[6] = {Yf, Xf, Yu, Xu, Yv, Xv};
    for (int a = 0; a < 6; a++) HIPCHK(hipMemcpyAsync(s + a * n, src[a], n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->tmask, tmask, n, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(build_geo_kernel, dim3(nblocks((int64_t)n)), dim3(kBlock), 0, h->stream, n, s, s + n, s + 2 * n, s + 3 * n,
                       s + 4 * n, s + 5 * n, h->geo, h->geoF);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(cell_orient_kernel, dim3(nblocks((int64_t)n)), dim3(kBlock), 0, h->stream, Nj, Ni, h->geo, h->orient);
    HIPCHK(hipGetLastError());
    // margin scale of the division-free cell test: every vertex coordinate is <= mg in magnitude
    // (a non-finite vertex makes it inf/NaN: nothing is decided by the filter and the plain test runs)
    double mg = 0.0;
    for (size_t k = 0; k < n; k++) {
        const double ay = fabs(Yf[k]), ax = fabs(Xf[k]);
        if (!(ay <= mg)) mg = ay;
        if (!(ax <= mg)) mg = ax;
    }
    h->eps_mg = 0x1p-48 * mg;
    HIPCHK(hipStreamSynchronize(h->stream));
    h->Nj = Nj; h->Ni = Ni;
    return SITRK_OK;
}

SITRK_API int sitrk_set_params(sitrk_t *h, double rdt, int uv_strategy, double rmin_conc)
{
    NEED(h, "null handle");
    NEED(uv_strategy >= 0 && uv_strategy <= 2,
         "sitrk_set_params: uv_strategy must be 0 (cell mean), 1 (nearest U/V point) or 2 (linear interpolation, not in the reference)");
    NEED(rdt > 0.0, "sitrk_set_params: rdt must be > 0");
    if (rmin_conc != h->rmin_conc) memset(h->slot_dirty, 1, sizeof(h->slot_dirty));     // masks depend on it
    h->rdt = rdt; h->uv_strategy = uv_strategy; h->rmin_conc = rmin_conc;
    return SITRK_OK;
}

SITRK_API int sitrk_set_tuning(sitrk_t *h, const char *knob, int value)
{
    NEED(h, "null handle");
    NEED(knob, "sitrk_set_tuning: null knob");
    int bit = 0;
    if (!strcmp(knob, "xcd_remap")) bit = TUNE_XCD_REMAP;
    else if (!strcmp(knob, "nt_state")) bit = TUNE_NT_STATE;
    else if (!strcmp(knob, "survive_tile")) bit = TUNE_SURVIVE_TILE;
    else if (!strcmp(knob, "step_block")) {          // workgroup size of advect_step_kernel
        if (value != 256 && value != 512 && value != 1024) return fail(h, SITRK_EINVAL, "sitrk_set_tuning: step_block must be 256, 512 or 1024");
        h->step_block = value;
        return SITRK_OK;
    }
    else if (!strcmp(knob, "fuse")) {                // records per launch in sitrk_run (1..32)
        if (value < 1 || value > kMaxFuse) return fail(h, SITRK_EINVAL, "sitrk_set_tuning: fuse must be 1..%d", kMaxFuse);
        h->fuse = value;
        return SITRK_OK;
    }
    else if (!strcmp(knob, "sort_tile")) {           // value = tile_j * 256 + tile_i, 0 = row-major
        const int tj = value >> 8, ti = value & 255;
        if (value != 0 && (tj < 1 || ti < 1)) return fail(h, SITRK_EINVAL, "sitrk_set_tuning: sort_tile = tile_j*256 + tile_i");
        h->tile_j = tj; h->tile_i = ti;
        return SITRK_OK;
    }
    else if (!strcmp(knob, "patch_kb")) {            // LDS bytes (KiB) per workgroup for the fused kernel's geometry patch
        if (value < 0 || value > 63) return fail(h, SITRK_EINVAL, "sitrk_set_tuning: patch_kb must be 0..63");
        h->patch_kb = value;
        return SITRK_OK;
    }
    else if (!strcmp(knob, "xcd_group")) {
        if (value < 0 || value > 4096) return fail(h, SITRK_EINVAL, "sitrk_set_tuning: xcd_group must be 0..4096");
        h->xcd_group = value;
        return SITRK_OK;
    }
    else if (!strcmp(knob, "async_survive")) {       // uploads derive their Survive bytes on the ingest stream (1) or on the compute stream (0)
        h->async_survive = value != 0;
        return SITRK_OK;
    }
    else if (!strcmp(knob, "fill_threads")) {        // host threads that copy a pushed record into the pinned staging
        if (value < 1 || value > 16) return fail(h, SITRK_EINVAL, "sitrk_set_tuning: fill_threads must be 1..16");
        h->fill_threads = value;
        return SITRK_OK;
    }
    else if (!strcmp(knob, "patch_margin")) {
        if (value < 0 || value > 64) return fail(h, SITRK_EINVAL, "sitrk_set_tuning: patch_margin must be 0..64");
        h->patch_margin = value;
        return SITRK_OK;
    }
    else if (!strcmp(knob, "locate_bruteforce")) bit = TUNE_LOCATE_BRUTEFORCE;
#ifdef SITRK_DIAG
    else if (!strcmp(knob, "stamps")) { h->stamps_on = value != 0; return SITRK_OK; }
    else if (!strcmp(knob, "diag_memonly")) bit = TUNE_DIAG_MEMONLY;
    else if (!strcmp(knob, "diag_nocross")) bit = TUNE_DIAG_NOCROSS;
#endif
    else return fail(h, SITRK_EINVAL, "sitrk_set_tuning: unknown knob '%s'", knob);
    h->tune = value ? (h->tune | bit) : (h->tune & ~bit);
    return SITRK_OK;
}

// --------------------------------------------------------------------------- records
static inline size_t elem_size(int dtype) { return dtype == SITRK_F64 ? 8 : 4; }

static void free_records(sitrk_ctx *h)
{
    dev_free(h->slabs); dev_free(h->kill9);
    h->slabs = nullptr; h->kill9 = nullptr; h->nslots = 0;
    for (int b = 0; b < sitrk_ctx::kStage; b++) {
        if (h->stage[b]) (void)hipHostFree(h->stage[b]);
        h->stage[b] = nullptr;
    }
    h->stage_bytes = 0; h->stage_rows = -1; h->stage_cols = 0; h->stage_next = 0;
    memset(h->slot_pending, 0, sizeof(h->slot_pending));
    memset(h->slot_sv_pending, 0, sizeof(h->slot_sv_pending));
    memset(h->slot_dirty, 1, sizeof(h->slot_dirty));
    for (int k = 0; k < 4096; k++) {
        h->slot_used_seq[k] = -1;
        h->slot_row_lo[k] = h->slot_row_hi[k] = 0;
        h->slot_col_lo[k] = h->slot_col_hi[k] = 0;
    }
}

SITRK_API int sitrk_alloc_records(sitrk_t *h, int nslots, int dtype)
{
    NEED(h, "null handle");
    NEED(h->geo, "sitrk_alloc_records: call sitrk_set_grid first");
    NEED(nslots >= 1 && nslots <= 4096, "sitrk_alloc_records: nslots out of range");
    NEED(dtype == SITRK_F32 || dtype == SITRK_F64, "sitrk_alloc_records: dtype must be SITRK_F32 or SITRK_F64");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->copy_stream));
    HIPCHK(hipStreamSynchronize(h->sv_stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    free_records(h);
    const size_t n = (size_t)h->Nj * h->Ni;
    h->slab_bytes = 3 * n * elem_size(dtype);
    HIPCHK(hipMalloc(&h->slabs, h->slab_bytes * nslots));
    HIPCHK(hipMalloc((void **)&h->kill9, n * nslots * sizeof(uint8_t)));
    // sentinels: a read outside the rows that were uploaded must be detectable, not silent garbage --
    // every Survive byte starts as "kill", every field value as NaN (0xff.. is a NaN in fp32 and fp64)
    HIPCHK(hipMemsetAsync(h->kill9, 0xff, n * nslots * sizeof(uint8_t), h->stream));
    HIPCHK(hipMemsetAsync(h->slabs, 0xff, h->slab_bytes * nslots, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->nslots = nslots; h->dtype = dtype;
    return SITRK_OK;
}

static inline char *slab_of(sitrk_ctx *h, int slot) { return (char *)h->slabs + (size_t)slot * h->slab_bytes; }

SITRK_API void *sitrk_record_ptr(sitrk_t *h, int slot)
{
    if (!h || !h->slabs || slot < 0 || slot >= h->nslots) return nullptr;
    h->slot_dirty[slot] = 1;            // the caller is about to write the slab
    h->slot_row_lo[slot] = 0; h->slot_row_hi[slot] = h->Nj;     // until a commit says otherwise
    h->slot_col_lo[slot] = 0; h->slot_col_hi[slot] = h->Ni;
    return slab_of(h, slot);
}

// The compute stream is about to read `slot`: order it behind an upload still in flight on the copy stream.
static int slot_wait_upload(sitrk_ctx *h, int slot)
{
    if (h->slot_pending[slot]) {
        HIPCHK(hipStreamWaitEvent(h->stream, h->slot_ready[slot], 0));
        h->slot_pending[slot] = 0;
    }
    return SITRK_OK;
}

// ... and behind a Survive derivation of the slot still in flight on the ingest stream
static int slot_wait_sv(sitrk_ctx *h, int slot)
{
    if (h->slot_sv_pending[slot]) {
        HIPCHK(hipStreamWaitEvent(h->stream, h->slot_sv[slot], 0));
        h->slot_sv_pending[slot] = 0;
    }
    return SITRK_OK;
}

// One event per launch (a ring of them): an upload into a slot waits for the last launch that read it, not for the
// whole compute stream, so the next records travel while the current ones are stepped with.
static int launch_mark(sitrk_ctx *h, const int *slots, int nslots_used)
{
    const long long seq = ++h->launch_seq;
    HIPCHK(hipEventRecord(h->launch_ev[seq % sitrk_ctx::kLaunchRing], h->stream));
    for (int k = 0; k < nslots_used; k++) h->slot_used_seq[slots[k]] = seq;
    return SITRK_OK;
}

// Survive bytes + packed neighbourhoods of the box rows [j0,j1) x columns [i0,i1) of `nb` records whose siconc is valid in
// exactly that box, one pass, ONE launch: record k's siconc field is sic + slots[k] * sic_stride elements, its Survive arrays
// kill / kill9 + slots[k] * kill_stride bytes (f64 or f32; a probe passes one record with strides 0)
static int launch_survive(sitrk_ctx *h, bool f64, const void *sic, int8_t *kill, uint8_t *kill9, int j0, int j1, int i0, int i1,
                          const int *slots = nullptr, int nb = 1, long long sic_stride = 0, long long kill_stride = 0,
                          hipStream_t stream = nullptr)
{
    if (!stream) stream = h->stream;
    SvBox bx;
    bx.j_lo = j0; bx.j_hi = j1; bx.v_lo = j0; bx.v_hi = j1;
    bx.cv_lo = i0; bx.cv_hi = i1;
    bx.c_lo = i0 & ~3;                                   // 4-byte aligned output strips (both kernels store 32-bit words when Ni % 4 == 0)
    SvBatch sb;
    sb.sic_stride = sic_stride; sb.kill_stride = kill_stride;
    for (int k = 0; k < kSvMaxBatch; k++) sb.slot[k] = (slots && k < nb) ? slots[k] : 0;
    if ((h->Ni & 3) == 0 && !(h->tune & TUNE_SURVIVE_TILE)) {
        // 16-byte aligned rows: the register-rolling form (no LDS, no barriers; sitrk_kernels.h)
        bx.c_hi = (i1 + 3) & ~3;
        const unsigned gx = (unsigned)((bx.c_hi - bx.c_lo + kSvRowsCols - 1) / kSvRowsCols);
        // (8 rows per wave instead of 16 for small boxes was measured: no gain alone, 15 % slower in a batch -- profiles/r04b_sv_box.jsonl)
        const dim3 g(gx, (unsigned)((j1 - j0 + 4 * kSvRowsR - 1) / (4 * kSvRowsR)), (unsigned)nb);
        if (f64)
            hipLaunchKernelGGL((survive_kill9_rows_kernel<double>), g, dim3(256), 0, stream, h->Nj, h->Ni, bx, sb, h->tmask,
                               (const double *)sic, h->rmin_conc, kill, kill9);
        else
            hipLaunchKernelGGL((survive_kill9_rows_kernel<float>), g, dim3(256), 0, stream, h->Nj, h->Ni, bx, sb, h->tmask,
                               (const float *)sic, h->rmin_conc, kill, kill9);
    } else {
        bx.c_hi = i1;
        const dim3 grid((unsigned)((bx.c_hi - bx.c_lo + kSvTC - 1) / kSvTC), (unsigned)((j1 - j0 + kSvTR - 1) / kSvTR), (unsigned)nb);
        if (f64)
            hipLaunchKernelGGL((survive_kill9_kernel<double>), grid, dim3(kSvBlock), 0, stream, h->Nj, h->Ni, bx, sb, h->tmask,
                               (const double *)sic, h->rmin_conc, kill, kill9);
        else
            hipLaunchKernelGGL((survive_kill9_kernel<float>), grid, dim3(kSvBlock), 0, stream, h->Nj, h->Ni, bx, sb, h->tmask,
                               (const float *)sic, h->rmin_conc, kill, kill9);
    }
    HIPCHK(hipGetLastError());
    return SITRK_OK;
}

// derive the Survive bytes of the box rows [j0,j1) x columns [i0,i1) of `nb` slots from their siconc there, in one launch.
//   on_ingest = false: on the compute stream, behind the slots' uploads and behind whatever the caller queued there before (a slab
//                      written in place through sitrk_record_ptr is ordered against the compute stream by its writer);
//   on_ingest = true : on the ingest stream (sv_stream), next to the stepping of OTHER slots: behind the slots' uploads, behind the
//                      last launch that read these slots' bytes, and in front of the first launch that will (slot_sv events).
static int derive_mask_box_batch(sitrk_ctx *h, const int *slots, int nb, int j0, int j1, int i0, int i1, bool on_ingest = false)
{
    const size_t n = (size_t)h->Nj * h->Ni, es = elem_size(h->dtype);
    const bool empty = (j0 >= j1 || i0 >= i1);           // (a slot that holds nothing: check_band refuses to step with it)
    if (!on_ingest) {
        for (int k = 0; k < nb; k++) {
            int rc = slot_wait_upload(h, slots[k]);
            if (rc) return rc;
            rc = slot_wait_sv(h, slots[k]);             // an earlier derivation of the same slot writes the same bytes
            if (rc) return rc;
        }
        if (!empty) {
            int rc = launch_survive(h, h->dtype == SITRK_F64, (const char *)h->slabs + 2 * n * es, nullptr, h->kill9, j0, j1, i0, i1,
                                    slots, nb, (long long)(h->slab_bytes / es), (long long)n);
            if (rc) return rc;
            rc = launch_mark(h, slots, nb);             // it reads the slots' siconc and writes their bytes: uploads and ingest-side
            if (rc) return rc;                          // derivations of these slots stay behind it
        }
    } else if (!empty) {
        for (int k = 0; k < nb; k++) {
            const int slot = slots[k];
            if (h->slot_pending[slot]) HIPCHK(hipStreamWaitEvent(h->sv_stream, h->slot_ready[slot], 0));     // (the compute stream waits for it too)
            if (h->slot_used_seq[slot] >= 0)
                HIPCHK(hipStreamWaitEvent(h->sv_stream, h->launch_ev[h->slot_used_seq[slot] % sitrk_ctx::kLaunchRing], 0));
        }
        int rc = launch_survive(h, h->dtype == SITRK_F64, (const char *)h->slabs + 2 * n * es, nullptr, h->kill9, j0, j1, i0, i1,
                                slots, nb, (long long)(h->slab_bytes / es), (long long)n, h->sv_stream);
        if (rc) return rc;
        for (int k = 0; k < nb; k++) {
            const int slot = slots[k];
            if (!h->slot_sv[slot]) HIPCHK(hipEventCreateWithFlags(&h->slot_sv[slot], hipEventDisableTiming));
            HIPCHK(hipEventRecord(h->slot_sv[slot], h->sv_stream));
            h->slot_sv_pending[slot] = 1;
        }
    }
    for (int k = 0; k < nb; k++) h->slot_dirty[slots[k]] = 0;
    return SITRK_OK;
}

static int derive_mask_box(sitrk_ctx *h, int slot, int j0, int j1, int i0, int i1)
{
    return derive_mask_box_batch(h, &slot, 1, j0, j1, i0, i1);
}

// a slot that was marked as rewritten in place is re-derived over the box it holds
static int derive_mask(sitrk_ctx *h, int slot)
{
    return derive_mask_box(h, slot, h->slot_row_lo[slot], h->slot_row_hi[slot], h->slot_col_lo[slot], h->slot_col_hi[slot]);
}

static inline void slot_holds(sitrk_ctx *h, int slot, int j0, int j1, int i0, int i1)
{
    h->slot_row_lo[slot] = j0; h->slot_row_hi[slot] = j1;
    h->slot_col_lo[slot] = i0; h->slot_col_hi[slot] = i1;
}

SITRK_API int sitrk_commit_record(sitrk_t *h, int slot)
{
    NEED(h, "null handle");
    NEED(h->slabs, "sitrk_commit_record: call sitrk_alloc_records first");
    NEED(slot >= 0 && slot < h->nslots, "sitrk_commit_record: slot out of range");
    HIPCHK(hipSetDevice(h->device));
    slot_holds(h, slot, 0, h->Nj, 0, h->Ni);
    return derive_mask(h, slot);
}

// ---- pinned staging + copy stream -------------------------------------------------------------------------
static int stage_acquire_box(sitrk_ctx *h, int nrows, int ncols, void **u, void **v, void **sic)
{
    NEED(h->stage_rows < 0, "sitrk_stage_acquire: the buffer handed out before was not submitted");
    HIPCHK(hipSetDevice(h->device));
    const int b = h->stage_next;
    if (!h->stage[b]) {                 // first use: one whole slab of pinned host memory per buffer
        HIPCHK(hipHostMalloc(&h->stage[b], h->slab_bytes, hipHostMallocDefault));
        h->stage_bytes = h->slab_bytes;
    }
    HIPCHK(hipEventSynchronize(h->stage_done[b]));      // the DMA that last read this buffer has finished
    const size_t nb = (size_t)nrows * ncols * elem_size(h->dtype);
    *u = h->stage[b];
    *v = (char *)h->stage[b] + nb;
    *sic = (char *)h->stage[b] + 2 * nb;
    h->stage_rows = nrows; h->stage_cols = ncols;
    return SITRK_OK;
}

SITRK_API int sitrk_stage_acquire(sitrk_t *h, int nrows, void **u, void **v, void **sic)
{
    NEED(h, "null handle");
    NEED(h->slabs, "sitrk_stage_acquire: call sitrk_alloc_records first");
    NEED(nrows >= 1 && nrows <= h->Nj, "sitrk_stage_acquire: nrows out of range");
    NEED(u && v && sic, "sitrk_stage_acquire: null output");
    return stage_acquire_box(h, nrows, h->Ni, u, v, sic);
}

SITRK_API int sitrk_stage_acquire_box(sitrk_t *h, int nrows, int ncols, void **u, void **v, void **sic)
{
    NEED(h, "null handle");
    NEED(h->slabs, "sitrk_stage_acquire_box: call sitrk_alloc_records first");
    NEED(nrows >= 1 && nrows <= h->Nj && ncols >= 1 && ncols <= h->Ni, "sitrk_stage_acquire_box: box out of range");
    NEED(u && v && sic, "sitrk_stage_acquire_box: null output");
    return stage_acquire_box(h, nrows, ncols, u, v, sic);
}

// the staged fields travel as the box rows [j0,j1) x columns [i0,i1) of `slot`: full-width boxes as three linear copies,
// others as three strided (2-D) copies out of the densely packed staging
static int stage_submit_box(sitrk_ctx *h, int slot, int j0, int j1, int i0, int i1)
{
    HIPCHK(hipSetDevice(h->device));
    const int b = h->stage_next;
    const size_t n = (size_t)h->Nj * h->Ni, es = elem_size(h->dtype);
    const int nr = j1 - j0, nc = i1 - i0;
    const size_t nb = (size_t)nr * nc * es;
    char *d = slab_of(h, slot);
    const char *src = (const char *)h->stage[b];
    // the copy may not overtake kernels that still read the slot -- the last launch that stepped with it, the last Survive derivation
    // that read its siconc on the ingest stream; nothing else on the compute stream holds it back
    if (h->slot_used_seq[slot] >= 0) HIPCHK(hipStreamWaitEvent(h->copy_stream, h->launch_ev[h->slot_used_seq[slot] % sitrk_ctx::kLaunchRing], 0));
    if (h->slot_sv[slot]) HIPCHK(hipStreamWaitEvent(h->copy_stream, h->slot_sv[slot], 0));
    for (int f = 0; f < 3; f++) {
        char *df = d + (size_t)f * n * es + ((size_t)j0 * h->Ni + i0) * es;
        if (nc == h->Ni) HIPCHK(hipMemcpyAsync(df, src + (size_t)f * nb, nb, hipMemcpyHostToDevice, h->copy_stream));
        else HIPCHK(hipMemcpy2DAsync(df, (size_t)h->Ni * es, src + (size_t)f * nb, (size_t)nc * es, (size_t)nc * es, (size_t)nr,
                                     hipMemcpyHostToDevice, h->copy_stream));
    }
    HIPCHK(hipEventRecord(h->stage_done[b], h->copy_stream));
    if (!h->slot_ready[slot]) HIPCHK(hipEventCreateWithFlags(&h->slot_ready[slot], hipEventDisableTiming));
    HIPCHK(hipEventRecord(h->slot_ready[slot], h->copy_stream));
    h->slot_pending[slot] = 1;
    h->stage_rows = -1;
    h->stage_next = (b + 1) % sitrk_ctx::kStage;
    slot_holds(h, slot, j0, j1, i0, i1);
    // the Survive bytes this box determines, behind the upload: on the ingest stream (next to the stepping of the resident records;
    // the first launch that reads the slot waits for it) or, knob async_survive = 0, on the compute stream as in rounds 1-3
    return derive_mask_box_batch(h, &slot, 1, j0, j1, i0, i1, h->async_survive != 0);
}

SITRK_API int sitrk_stage_submit(sitrk_t *h, int slot, int j0, int j1)
{
    NEED(h, "null handle");
    NEED(h->slabs, "sitrk_stage_submit: call sitrk_alloc_records first");
    NEED(slot >= 0 && slot < h->nslots, "sitrk_stage_submit: slot out of range");
    NEED(h->stage_rows >= 0, "sitrk_stage_submit: nothing acquired");
    NEED(j0 >= 0 && j1 <= h->Nj && j1 - j0 == h->stage_rows && h->stage_cols == h->Ni, "sitrk_stage_submit: rows [j0,j1) do not match the acquired buffer");
    return stage_submit_box(h, slot, j0, j1, 0, h->Ni);
}

SITRK_API int sitrk_stage_submit_box(sitrk_t *h, int slot, int j0, int j1, int i0, int i1)
{
    NEED(h, "null handle");
    NEED(h->slabs, "sitrk_stage_submit_box: call sitrk_alloc_records first");
    NEED(slot >= 0 && slot < h->nslots, "sitrk_stage_submit_box: slot out of range");
    NEED(h->stage_rows >= 0, "sitrk_stage_submit_box: nothing acquired");
    NEED(j0 >= 0 && j1 <= h->Nj && j1 - j0 == h->stage_rows && i0 >= 0 && i1 <= h->Ni && i1 - i0 == h->stage_cols,
         "sitrk_stage_submit_box: the box does not match the acquired buffer");
    return stage_submit_box(h, slot, j0, j1, i0, i1);
}

SITRK_API int sitrk_stage_release(sitrk_t *h)
{
    NEED(h, "null handle");
    h->stage_rows = -1;                 // stage_next is unchanged: the same buffer goes out again
    return SITRK_OK;
}

// one row of a box into the (write-only, 16-byte aligned) pinned staging with non-temporal stores: a memcpy of a 10-KB row uses
// ordinary stores, whose read-for-ownership doubles the traffic of the destination (measured on the box rows of C3, profiles/r04e_e2e.txt:
// 4 threads 29 -> 36 GB/s, 8 threads 32 -> 47 GB/s; the PCIe link moves 55)
static inline void copy_row_stream(char *d, const char *s, size_t n)
{
    typedef long long v2ll __attribute__((vector_size(16)));
    typedef long long v2ll_u __attribute__((vector_size(16), aligned(1)));
    size_t k = 0;
    if (((uintptr_t)d & 15u) == 0) {
        for (; k + 64 <= n; k += 64) {
            const v2ll a = *(const v2ll_u *)(s + k), b = *(const v2ll_u *)(s + k + 16), c = *(const v2ll_u *)(s + k + 32),
                       e = *(const v2ll_u *)(s + k + 48);
            __builtin_nontemporal_store(a, (v2ll *)(d + k));
            __builtin_nontemporal_store(b, (v2ll *)(d + k + 16));
            __builtin_nontemporal_store(c, (v2ll *)(d + k + 32));
            __builtin_nontemporal_store(e, (v2ll *)(d + k + 48));
        }
    }
    if (k < n) memcpy(d + k, s + k, n - k);
}

// host arrays -> staging -> slot.  u, v, sic address element (j0,i0) of the box; consecutive rows of the box are `ld` elements
// apart in the caller's arrays (ld = i1-i0: the arrays hold exactly the box, densely packed; ld = Ni: views into whole fields).
static int push_box(sitrk_ctx *h, int slot, int j0, int j1, int i0, int i1, const void *u, const void *v, const void *sic, int64_t ld)
{
    void *su, *sv, *ss;
    int rc = stage_acquire_box(h, j1 - j0, i1 - i0, &su, &sv, &ss);
    if (rc) return rc;
    const size_t es = elem_size(h->dtype);
    const size_t nr = (size_t)(j1 - j0), rowb = (size_t)(i1 - i0) * es, srcb = (size_t)ld * es;
    const size_t nb = nr * rowb;
    // from here on the caller's buffers are its own again.  One thread copies ~10 GB/s into pinned memory, a fifth of what
    // the PCIe link then moves: large records are copied by a few threads (201 MB at 4096^2: 20 ms -> 6 ms)
    const void *src[3] = {u, v, sic};
    void *dst[3] = {su, sv, ss};
    const int nthr = nb >= ((size_t)8 << 20) ? h->fill_threads : 1;
    const bool dense = (srcb == rowb);
    // thread t takes rows [t*part, (t+1)*part) of every field
    const size_t part = (nr + nthr - 1) / nthr;
    auto copy_part = [=](int t) {
        const size_t r0 = (size_t)t * part, r1 = std::min(nr, r0 + part);
        if (r0 >= r1) return;
        for (int f = 0; f < 3; f++) {
            if (dense) memcpy((char *)dst[f] + r0 * rowb, (const char *)src[f] + r0 * rowb, (r1 - r0) * rowb);
            else
                for (size_t r = r0; r < r1; r++) copy_row_stream((char *)dst[f] + r * rowb, (const char *)src[f] + r * srcb, rowb);
        }
        if (!dense) __atomic_thread_fence(__ATOMIC_SEQ_CST);      // (mfence: the streamed rows are globally visible before the DMA is queued)
    };
    if (nthr == 1) {
        copy_part(0);
    } else {
        std::thread pool[15];
        int started = 0;
        try {
            for (; started < nthr - 1; started++) pool[started] = std::thread(copy_part, started + 1);
        } catch (...) {                                 // no more threads to be had (no C++ exception crosses the C ABI)
        }
        copy_part(0);
        for (int t = started; t < nthr - 1; t++) copy_part(t + 1);       // the parts nobody took
        for (int t = 0; t < started; t++) pool[t].join();
    }
    return stage_submit_box(h, slot, j0, j1, i0, i1);
}

SITRK_API int sitrk_push_record(sitrk_t *h, int slot, const void *u, const void *v, const void *sic)
{
    NEED(h, "null handle");
    NEED(h->slabs, "sitrk_push_record: call sitrk_alloc_records first");
    NEED(slot >= 0 && slot < h->nslots, "sitrk_push_record: slot out of range");
    NEED(u && v && sic, "sitrk_push_record: null field");
    return push_box(h, slot, 0, h->Nj, 0, h->Ni, u, v, sic, h->Ni);
}

// rows and columns of the live buoys' host cells (one small kernel + one synchronisation of the compute stream)
static int eval_buoy_box(sitrk_ctx *h)
{
    h->box_pending = false;             // a synchronous evaluation supersedes one that was begun and not collected
    h->band_jmin = 1; h->band_jmax = 0; h->band_imin = 1; h->band_imax = 0;
    h->band_age = -1;                   // no box is known until this evaluation has succeeded (check_band refuses partial slots meanwhile)
    if (h->nP == 0) { h->band_age = 0; return SITRK_OK; }
    NEED(h->st[0].pos, "sitrk_buoy_rows: call sitrk_set_buoys first");
    HIPCHK(hipSetDevice(h->device));
    int res[4];
    int *d = (int *)h->counter;                                     // first half of the 32-byte reduction scratch
    HIPCHK(hipMemsetAsync(d, 0x80, 4 * sizeof(int), h->stream));    // four maxima start at -2139062144
    hipLaunchKernelGGL(buoy_box_kernel, dim3(std::min(nblocks(h->nP, 4 * kBlock), 2048u)), dim3(kBlock), 0, h->stream, h->nP, h->st[h->cur].cell, d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(res, d, sizeof(res), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (res[0] >= 0) { h->band_jmax = res[0]; h->band_jmin = -res[1]; h->band_imax = res[2]; h->band_imin = -res[3]; }
    h->band_age = 0;
    return SITRK_OK;
}

SITRK_API int sitrk_buoy_rows(sitrk_t *h, int32_t *jmin, int32_t *jmax)
{
    NEED(h, "null handle");
    NEED(jmin && jmax, "sitrk_buoy_rows: null output");
    *jmin = 1; *jmax = 0;
    int rc = eval_buoy_box(h);
    if (rc) return rc;
    *jmin = h->band_jmin; *jmax = h->band_jmax;
    return SITRK_OK;
}

SITRK_API int sitrk_buoy_box(sitrk_t *h, int32_t *jmin, int32_t *jmax, int32_t *imin, int32_t *imax)
{
    NEED(h, "null handle");
    NEED(jmin && jmax && imin && imax, "sitrk_buoy_box: null output");
    *jmin = 1; *jmax = 0; *imin = 1; *imax = 0;
    int rc = eval_buoy_box(h);
    if (rc) return rc;
    *jmin = h->band_jmin; *jmax = h->band_jmax; *imin = h->band_imin; *imax = h->band_imax;
    return SITRK_OK;
}

// The same without stalling the stream: _begin queues the reduction (and the copy of its result into pinned host memory) behind
// the work already queued, _end waits for THAT point only -- launches queued after the begin keep the GPU busy meanwhile -- and
// adopts the result: the box then counts as evaluated at the begin (records stepped since the begin are its age).
SITRK_API int sitrk_buoy_box_begin(sitrk_t *h)
{
    NEED(h, "null handle");
    NEED(!h->box_pending, "sitrk_buoy_box_begin: the evaluation begun before was not collected (sitrk_buoy_box_end)");
    h->box_host[0] = -1; h->box_host[1] = 0; h->box_host[2] = -1; h->box_host[3] = 0;      // {max j, max -j, max i, max -i}: none alive
    h->box_pending_age = 0;
    if (h->nP == 0) { h->box_pending = true; return SITRK_OK; }
    NEED(h->st[0].pos, "sitrk_buoy_box_begin: call sitrk_set_buoys first");
    HIPCHK(hipSetDevice(h->device));
    int *d = (int *)h->counter + 4;                                 // second half of the 32-byte reduction scratch
    HIPCHK(hipMemsetAsync(d, 0x80, 4 * sizeof(int), h->stream));
    hipLaunchKernelGGL(buoy_box_kernel, dim3(std::min(nblocks(h->nP, 4 * kBlock), 2048u)), dim3(kBlock), 0, h->stream, h->nP, h->st[h->cur].cell, d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h->box_host, d, 4 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipEventRecord(h->box_ev, h->stream));
    h->box_pending = true;              // only now: an evaluation that failed to queue leaves nothing to collect
    return SITRK_OK;
}

SITRK_API int sitrk_buoy_box_end(sitrk_t *h, int32_t *jmin, int32_t *jmax, int32_t *imin, int32_t *imax, int32_t *age)
{
    NEED(h, "null handle");
    NEED(h->box_pending, "sitrk_buoy_box_end: nothing begun");
    NEED(jmin && jmax && imin && imax, "sitrk_buoy_box_end: null output");
    if (h->nP > 0) HIPCHK(hipEventSynchronize(h->box_ev));
    h->box_pending = false;
    if (h->nP > 0 && h->box_host[0] >= 0) {
        h->band_jmax = h->box_host[0]; h->band_jmin = -h->box_host[1]; h->band_imax = h->box_host[2]; h->band_imin = -h->box_host[3];
    } else {
        h->band_jmin = 1; h->band_jmax = 0; h->band_imin = 1; h->band_imax = 0;
    }
    h->band_age = h->box_pending_age;
    *jmin = h->band_jmin; *jmax = h->band_jmax; *imin = h->band_imin; *imax = h->band_imax;
    if (age) *age = h->band_age;
    return SITRK_OK;
}

SITRK_API int sitrk_push_record_rows(sitrk_t *h, int slot, int j0, int j1, const void *u_rows, const void *v_rows, const void *sic_rows)
{
    NEED(h, "null handle");
    NEED(h->slabs, "sitrk_push_record_rows: call sitrk_alloc_records first");
    NEED(slot >= 0 && slot < h->nslots, "sitrk_push_record_rows: slot out of range");
    NEED(j0 >= 0 && j1 <= h->Nj && j0 <= j1, "sitrk_push_record_rows: rows out of range");
    if (j0 == j1) { slot_holds(h, slot, 0, 0, 0, 0); return SITRK_OK; }
    NEED(u_rows && v_rows && sic_rows, "sitrk_push_record_rows: null field");
    return push_box(h, slot, j0, j1, 0, h->Ni, u_rows, v_rows, sic_rows, h->Ni);
}

SITRK_API int sitrk_push_record_box(sitrk_t *h, int slot, int j0, int j1, int i0, int i1, const void *u_box, const void *v_box,
                                    const void *sic_box, int64_t ld)
{
    NEED(h, "null handle");
    NEED(h->slabs, "sitrk_push_record_box: call sitrk_alloc_records first");
    NEED(slot >= 0 && slot < h->nslots, "sitrk_push_record_box: slot out of range");
    NEED(j0 >= 0 && j1 <= h->Nj && j0 <= j1 && i0 >= 0 && i1 <= h->Ni && i0 <= i1, "sitrk_push_record_box: box out of range");
    if (j0 == j1 || i0 == i1) { slot_holds(h, slot, 0, 0, 0, 0); return SITRK_OK; }
    NEED(u_box && v_box && sic_box, "sitrk_push_record_box: null field");
    NEED(ld >= (int64_t)(i1 - i0), "sitrk_push_record_box: ld is smaller than the box is wide");
    return push_box(h, slot, j0, j1, i0, i1, u_box, v_box, sic_box, ld);
}

SITRK_API int sitrk_commit_record_rows(sitrk_t *h, int slot, int j0, int j1)
{
    NEED(h, "null handle");
    NEED(h->slabs, "sitrk_commit_record_rows: call sitrk_alloc_records first");
    NEED(slot >= 0 && slot < h->nslots, "sitrk_commit_record_rows: slot out of range");
    NEED(j0 >= 0 && j1 <= h->Nj && j0 <= j1, "sitrk_commit_record_rows: rows out of range");
    if (j0 == j1) { slot_holds(h, slot, 0, 0, 0, 0); return SITRK_OK; }
    slot_holds(h, slot, j0, j1, 0, h->Ni);
    HIPCHK(hipSetDevice(h->device));
    return derive_mask_box(h, slot, j0, j1, 0, h->Ni);
}

SITRK_API int sitrk_commit_record_box(sitrk_t *h, int slot, int j0, int j1, int i0, int i1)
{
    NEED(h, "null handle");
    NEED(h->slabs, "sitrk_commit_record_box: call sitrk_alloc_records first");
    NEED(slot >= 0 && slot < h->nslots, "sitrk_commit_record_box: slot out of range");
    NEED(j0 >= 0 && j1 <= h->Nj && j0 <= j1 && i0 >= 0 && i1 <= h->Ni && i0 <= i1, "sitrk_commit_record_box: box out of range");
    if (j0 == j1 || i0 == i1) { slot_holds(h, slot, 0, 0, 0, 0); return SITRK_OK; }
    slot_holds(h, slot, j0, j1, i0, i1);
    HIPCHK(hipSetDevice(h->device));
    return derive_mask_box(h, slot, j0, j1, i0, i1);
}

static int commit_records_box(sitrk_ctx *h, int slot0, int nrec, int j0, int j1, int i0, int i1, bool on_ingest);

SITRK_API int sitrk_commit_records_box(sitrk_t *h, int slot0, int nrec, int j0, int j1, int i0, int i1)
{
    return commit_records_box(h, slot0, nrec, j0, j1, i0, i1, false);
}

SITRK_API int sitrk_commit_records_box_async(sitrk_t *h, int slot0, int nrec, int j0, int j1, int i0, int i1)
{
    return commit_records_box(h, slot0, nrec, j0, j1, i0, i1, true);
}

static int commit_records_box(sitrk_ctx *h, int slot0, int nrec, int j0, int j1, int i0, int i1, bool on_ingest)
{
    NEED(h, "null handle");
    NEED(h->slabs, "sitrk_commit_records_box: call sitrk_alloc_records first");
    NEED(slot0 >= 0 && slot0 < h->nslots, "sitrk_commit_records_box: slot0 out of range");
    NEED(nrec >= 0 && nrec <= h->nslots, "sitrk_commit_records_box: more records than slots");
    NEED(j0 >= 0 && j1 <= h->Nj && j0 <= j1 && i0 >= 0 && i1 <= h->Ni && i0 <= i1, "sitrk_commit_records_box: box out of range");
    HIPCHK(hipSetDevice(h->device));
    const bool empty = (j0 == j1 || i0 == i1);
    for (int k = 0; k < nrec; k += kSvMaxBatch) {
        int slots[kSvMaxBatch];
        const int nb = std::min(kSvMaxBatch, nrec - k);
        for (int q = 0; q < nb; q++) {
            slots[q] = (slot0 + k + q) % h->nslots;
            if (empty) slot_holds(h, slots[q], 0, 0, 0, 0);
            else slot_holds(h, slots[q], j0, j1, i0, i1);
        }
        if (empty) continue;
        int rc = derive_mask_box_batch(h, slots, nb, j0, j1, i0, i1, on_ingest);
        if (rc) return rc;
    }
    return SITRK_OK;
}

SITRK_API int sitrk_push_record_dev(sitrk_t *h, int slot, const void *slab_dev)
{
    NEED(h, "null handle");
    NEED(h->slabs, "sitrk_push_record_dev: call sitrk_alloc_records first");
    NEED(slot >= 0 && slot < h->nslots, "sitrk_push_record_dev: slot out of range");
    NEED(slab_dev, "sitrk_push_record_dev: null slab");
    HIPCHK(hipSetDevice(h->device));
    void *d = slab_of(h, slot);
    int rc = slot_wait_upload(h, slot);
    if (rc) return rc;
    rc = slot_wait_sv(h, slot);                         // (a derivation still reading the slot's siconc on the ingest stream)
    if (rc) return rc;
    if (d != slab_dev) HIPCHK(hipMemcpyAsync(d, slab_dev, h->slab_bytes, hipMemcpyDeviceToDevice, h->stream));
    slot_holds(h, slot, 0, h->Nj, 0, h->Ni);
    return derive_mask(h, slot);
}

// A slot that holds only a box of its record may be stepped with only while every live buoy is provably inside the box:
// rows [jmin-2-age, jmax+3+age) and columns [imin-2-age, imax+3+age) with (jmin,jmax,imin,imax) from the last
// sitrk_buoy_rows() / sitrk_buoy_box() and age = records stepped since.
static int check_band(sitrk_ctx *h, int slot, int extra_age)
{
    const int lo = h->slot_row_lo[slot], hi = h->slot_row_hi[slot], clo = h->slot_col_lo[slot], chi = h->slot_col_hi[slot];
    if (lo == 0 && hi == h->Nj && clo == 0 && chi == h->Ni) return SITRK_OK;
    if (h->band_age < 0)
        return fail(h, SITRK_EINVAL, "slot %d holds rows [%d,%d) x columns [%d,%d) only: call sitrk_buoy_rows() / sitrk_buoy_box() after "
                    "sitrk_set_buoys() so that the box can be checked", slot, lo, hi, clo, chi);
    if (h->band_jmin > h->band_jmax) return SITRK_OK;       // no live buoy
    const int age = h->band_age + extra_age;
    const int need_lo = std::max(0, h->band_jmin - 2 - age), need_hi = std::min(h->Nj, h->band_jmax + 3 + age);
    if (lo > need_lo || hi < need_hi)
        return fail(h, SITRK_EINVAL, "slot %d holds rows [%d,%d) but the buoys (rows %d..%d, %d records ago) can touch rows [%d,%d)",
                    slot, lo, hi, h->band_jmin, h->band_jmax, age, need_lo, need_hi);
    const int cneed_lo = std::max(0, h->band_imin - 2 - age), cneed_hi = std::min(h->Ni, h->band_imax + 3 + age);
    if (clo > cneed_lo || chi < cneed_hi)
        return fail(h, SITRK_EINVAL, "slot %d holds columns [%d,%d) but the buoys (columns %d..%d, %d records ago) can touch columns [%d,%d)",
                    slot, clo, chi, h->band_imin, h->band_imax, age, cneed_lo, cneed_hi);
    return SITRK_OK;
}

// --------------------------------------------------------------------------- buoys
SITRK_API int sitrk_set_buoys(sitrk_t *h, int64_t nP, const double *yx, const int32_t *jiT, const int32_t *rec_first,
                              const int32_t *rec_last)
{
    NEED(h, "null handle");
    NEED(h->geo, "sitrk_set_buoys: call sitrk_set_grid first");
    NEED(nP >= 0 && nP < 2147483647LL, "sitrk_set_buoys: nP out of range");
    NEED(nP == 0 || (yx && jiT), "sitrk_set_buoys: null array");
    NEED((rec_first == nullptr) == (rec_last == nullptr), "sitrk_set_buoys: rec_first and rec_last go together");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    free_buoys(h);
    h->windowed = (rec_first != nullptr);
    h->cur = 0; h->steps_since_sort = 0; h->sorted_once = false;
    h->band_age = -1; h->box_pending = false;
    // host-side validation + packing of the host cell
    std::vector<int32_t> packed;
    try {
        packed.resize((size_t)nP);
    } catch (const std::bad_alloc &) {          // no C++ exception may cross the C ABI
        return fail(h, SITRK_ENOMEM, "sitrk_set_buoys: out of host memory for %lld buoys", (long long)nP);
    }
    bool rim = false;
    for (int64_t p = 0; p < nP; p++) {
        int j = jiT[2 * p], i = jiT[2 * p + 1];
        if (j < 1 || j > h->Nj - 2 || i < 1 || i > h->Ni - 2)
            return fail(h, SITRK_EINDEX, "sitrk_set_buoys: buoy %lld host cell (%d,%d) outside 1..%d x 1..%d "
                        "(the reference would index out of range)", (long long)p, j, i, h->Nj - 2, h->Ni - 2);
        packed[(size_t)p] = pack_cell(j, i);
        rim = rim || j < 2 || i < 2;
    }
    h->rim_buoys = rim;
    h->win_first_max = INT32_MIN; h->win_last_min = INT32_MAX;
    std::vector<int2> win;                          // (first, last) interleaved: one 8-byte word per buoy on the device
    if (h->windowed) {
        try {
            win.resize((size_t)nP);
        } catch (const std::bad_alloc &) {
            return fail(h, SITRK_ENOMEM, "sitrk_set_buoys: out of host memory for %lld buoys", (long long)nP);
        }
        for (int64_t p = 0; p < nP; p++) {
            h->win_first_max = std::max(h->win_first_max, rec_first[p]);
            h->win_last_min = std::min(h->win_last_min, rec_last[p]);
            win[(size_t)p].x = rec_first[p]; win[(size_t)p].y = rec_last[p];
        }
    }
    for (int b = 0; b < 2; b++) {
        HIPCHK(dev_alloc(&h->st[b].pos, (size_t)nP));
        HIPCHK(dev_alloc(&h->st[b].cell, (size_t)nP));
        HIPCHK(dev_alloc(&h->st[b].kill_rec, (size_t)nP));
        HIPCHK(dev_alloc(&h->st[b].perm, (size_t)nP));
        if (h->windowed) {
            HIPCHK(dev_alloc(&h->st[b].win, (size_t)nP));
        }
        HIPCHK(dev_alloc(&h->keys[b], (size_t)nP));
        HIPCHK(dev_alloc(&h->vals[b], (size_t)nP));
    }
    h->nP = nP;
    if (nP == 0) return SITRK_OK;
    BuoyState &s = h->st[0];
    HIPCHK(hipMemcpyAsync(s.pos, yx, (size_t)nP * sizeof(pt), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(s.cell, packed.data(), (size_t)nP * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemsetAsync(s.kill_rec, 0xff, (size_t)nP * 4, h->stream));          // -1
    hipLaunchKernelGGL(iota_kernel, dim3(nblocks(nP)), dim3(kBlock), 0, h->stream, nP, s.perm);
    HIPCHK(hipGetLastError());
    if (h->windowed) {
        HIPCHK(hipMemcpyAsync(s.win, win.data(), (size_t)nP * sizeof(int2), hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));      // `packed` must outlive the copy
    return SITRK_OK;
}

SITRK_API int sitrk_restore_state(sitrk_t *h, const int8_t *alive, const int32_t *kill_rec)
{
    NEED(h, "null handle");
    NEED(h->st[0].pos, "sitrk_restore_state: call sitrk_set_buoys first");
    NEED(alive && kill_rec, "sitrk_restore_state: null array");
    const int64_t nP = h->nP;
    if (nP == 0) return SITRK_OK;
    HIPCHK(hipSetDevice(h->device));
    const size_t b_a = align256((size_t)nP), b_k = align256((size_t)nP * 4);
    int rc = ensure_scratch(h, b_a + b_k);
    if (rc) return rc;
    char *w = (char *)h->scratch;
    HIPCHK(hipMemcpyAsync(w, alive, (size_t)nP, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(w + b_a, kill_rec, (size_t)nP * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemsetAsync(h->counter, 0, sizeof(unsigned long long), h->stream));
    hipLaunchKernelGGL(restore_state_kernel, dim3(nblocks(nP)), dim3(kBlock), 0, h->stream, nP, h->st[h->cur], (const int8_t *)w,
                       (const int32_t *)(w + b_a), h->counter);
    HIPCHK(hipGetLastError());
    unsigned long long rim = 0;
    HIPCHK(hipMemcpyAsync(&rim, h->counter, sizeof(rim), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));          // the caller's arrays are its own again
    h->rim_buoys = rim != 0;                          // dead buoys sit where they died, often in the rim: they never step
    return SITRK_OK;
}

SITRK_API int sitrk_set_resort(sitrk_t *h, int resort_every)
{
    NEED(h, "null handle");
    NEED(resort_every >= 0, "sitrk_set_resort: resort_every must be >= 0");
    h->resort_every = resort_every;
    return SITRK_OK;
}

SITRK_API int sitrk_sort_buoys(sitrk_t *h)
{
    NEED(h, "null handle");
    NEED(h->st[0].pos, "sitrk_sort_buoys: call sitrk_set_buoys first");
    const int64_t nP = h->nP;
    h->steps_since_sort = 0;
    if (nP <= 1) return SITRK_OK;
    HIPCHK(hipSetDevice(h->device));
    BuoyState &in = h->st[h->cur], &out = h->st[h->cur ^ 1];
    // keys are < (#tiles) * tile cells (row-major: Nj*Ni); the dead key sits just above
    uint32_t dead_key = (uint32_t)h->Nj * (uint32_t)h->Ni;
    if (h->tile_j) {
        const uint64_t ntj = ((uint64_t)h->Nj + h->tile_j - 1) / h->tile_j, nti = ((uint64_t)h->Ni + h->tile_i - 1) / h->tile_i;
        const uint64_t lim = ntj * nti * (uint64_t)(h->tile_j * h->tile_i);
        if (lim >= 0xffffffffull) return fail(h, SITRK_EINVAL, "sitrk_sort_buoys: tile-major key does not fit 32 bits");
        dead_key = (uint32_t)lim;
    }
    unsigned end_bit = 1;
    while (end_bit < 32 && (dead_key >> end_bit)) end_bit++;
    hipLaunchKernelGGL(make_keys_kernel, dim3(nblocks(nP)), dim3(kBlock), 0, h->stream, nP, h->Ni, h->tile_j, h->tile_i, dead_key, in.cell, h->keys[0], h->vals[0]);
    HIPCHK(hipGetLastError());
    size_t need = 0;
    HIPCHK(sort_pairs_u32(nullptr, &need, h->keys[0], h->keys[1], h->vals[0], h->vals[1], (size_t)nP, end_bit, h->stream));
    if (need > h->sort_tmp_bytes) {
        HIPCHK(hipStreamSynchronize(h->stream));
        dev_free(h->sort_tmp);
        h->sort_tmp = nullptr; h->sort_tmp_bytes = 0;
        HIPCHK(hipMalloc(&h->sort_tmp, need));
        h->sort_tmp_bytes = need;
    }
    size_t tb = h->sort_tmp_bytes;
    HIPCHK(sort_pairs_u32(h->sort_tmp, &tb, h->keys[0], h->keys[1], h->vals[0], h->vals[1], (size_t)nP, end_bit, h->stream));
    hipLaunchKernelGGL(permute_state_kernel, dim3(nblocks(nP)), dim3(kBlock), 0, h->stream, nP, h->vals[1], h->keys[1], dead_key, h->Ni, h->tile_j, h->tile_i,
                       in, out, h->windowed);
    HIPCHK(hipGetLastError());
    h->cur ^= 1;
    h->sorted_once = true;
    return SITRK_OK;
}

// records [jrec0, jrec0 + n) need the per-buoy window test unless they lie inside every buoy's window
static inline bool window_test_needed(const sitrk_ctx *h, int jrec0, int n)
{
    return h->windowed && !(jrec0 >= h->win_first_max && (int64_t)jrec0 + n - 1 <= h->win_last_min);
}

template <typename FT, int BLOCK>
static void launch_step_b(sitrk_ctx *h, const StepArgs &a)
{
    const bool windowed = window_test_needed(h, a.jrec, 1);
    dim3 grid(nblocks(a.nP, BLOCK)), block(BLOCK);
    if (h->uv_strategy == 1) {
        if (windowed) hipLaunchKernelGGL((advect_step_kernel<FT, 1, true, BLOCK>), grid, block, 0, h->stream, a);
        else hipLaunchKernelGGL((advect_step_kernel<FT, 1, false, BLOCK>), grid, block, 0, h->stream, a);
    } else if (h->uv_strategy == 2) {
        if (windowed) hipLaunchKernelGGL((advect_step_kernel<FT, 2, true, BLOCK>), grid, block, 0, h->stream, a);
        else hipLaunchKernelGGL((advect_step_kernel<FT, 2, false, BLOCK>), grid, block, 0, h->stream, a);
    } else {
        if (windowed) hipLaunchKernelGGL((advect_step_kernel<FT, 0, true, BLOCK>), grid, block, 0, h->stream, a);
        else hipLaunchKernelGGL((advect_step_kernel<FT, 0, false, BLOCK>), grid, block, 0, h->stream, a);
    }
}

template <typename FT>
static void launch_step(sitrk_ctx *h, const StepArgs &a)
{
#ifdef SITRK_DIAG
    dim3 grid(nblocks(a.nP)), block(kBlock);
    if (h->tune & TUNE_DIAG_MEMONLY) { hipLaunchKernelGGL((advect_memonly_kernel<FT>), grid, block, 0, h->stream, a); return; }
    if (h->tune & TUNE_DIAG_NOCROSS) { hipLaunchKernelGGL((advect_nocross_kernel<FT>), grid, block, 0, h->stream, a); return; }
#endif
    if (h->step_block == 1024) launch_step_b<FT, 1024>(h, a);
    else if (h->step_block == 512) launch_step_b<FT, 512>(h, a);
    else launch_step_b<FT, 256>(h, a);
}

SITRK_API int sitrk_step(sitrk_t *h, int slot, int jrec)
{
    NEED(h, "null handle");
    NEED(h->st[0].pos, "sitrk_step: call sitrk_set_buoys first");
    NEED(h->slabs, "sitrk_step: call sitrk_alloc_records first");
    NEED(slot >= 0 && slot < h->nslots, "sitrk_step: slot out of range");
    if (h->nP == 0) return SITRK_OK;
    int rc = check_band(h, slot, 0);
    if (rc) return rc;
    if (h->resort_every > 0 && h->steps_since_sort >= h->resort_every) {
        rc = sitrk_sort_buoys(h);
        if (rc) return rc;
    }
    if (h->slot_dirty[slot]) {            // slab written through sitrk_record_ptr and not committed yet
        rc = derive_mask(h, slot);
        if (rc) return rc;
    }
    rc = slot_wait_upload(h, slot);
    if (rc) return rc;
    rc = slot_wait_sv(h, slot);
    if (rc) return rc;
    const size_t n = (size_t)h->Nj * h->Ni, es = elem_size(h->dtype);
    const char *slab = slab_of(h, slot);
    BuoyState &s = h->st[h->cur];
    StepArgs a;
    a.nP = h->nP; a.tune = h->tune; a.Nj = h->Nj; a.Ni = h->Ni; a.jrec = jrec;
    a.rdt = h->rdt; a.rmin_conc = h->rmin_conc; a.eps_mg = h->eps_mg;
    a.geo = h->geo; a.orient = h->orient; a.kill = h->kill9 + (size_t)slot * n;
    a.u = slab; a.v = slab + n * es;
    a.pos = s.pos; a.cell = s.cell; a.kill_rec = s.kill_rec; a.win = s.win;
    if (h->dtype == SITRK_F64) launch_step<double>(h, a);
    else launch_step<float>(h, a);
    HIPCHK(hipGetLastError());
    rc = launch_mark(h, &slot, 1);
    if (rc) return rc;
    h->steps_since_sort++;
    h->n_step_launches++;
    if (h->band_age >= 0) h->band_age++;
    if (h->box_pending) h->box_pending_age++;
    return SITRK_OK;
}

// div1000_of_f32's premise on the time step (sitrk_geom.h); outside it every lane takes the division
static int f32_class_for(double rdt)
{
    const double ar = std::fabs(rdt);
    return (ar >= 0x1p-700 && ar <= 0x1p700) ? kClassFiniteNonzeroF32 : 0;
}

template <typename FT>
static void launch_run(sitrk_ctx *h, const RunArgs &ra)
{
    dim3 grid(nblocks(ra.s.nP, kRunBlock)), block(kRunBlock);
    // dynamic LDS: tables + the patch's F-points
    const size_t lds = kRunLdsFixed + (size_t)ra.patch_cells * sizeof(pt);
    const bool windowed = window_test_needed(h, ra.s.jrec, ra.nrec);
#define SITRK_LAUNCH_RUN(KERNEL)                                                                          \
    do {                                                                                                  \
        if (h->uv_strategy == 1) {                                                                        \
            if (windowed) hipLaunchKernelGGL((KERNEL<FT, 1, true>), grid, block, lds, h->stream, ra);     \
            else hipLaunchKernelGGL((KERNEL<FT, 1, false>), grid, block, lds, h->stream, ra);             \
        } else if (h->uv_strategy == 2) {                                                                 \
            if (windowed) hipLaunchKernelGGL((KERNEL<FT, 2, true>), grid, block, lds, h->stream, ra);     \
            else hipLaunchKernelGGL((KERNEL<FT, 2, false>), grid, block, lds, h->stream, ra);             \
        } else {                                                                                          \
            if (windowed) hipLaunchKernelGGL((KERNEL<FT, 0, true>), grid, block, lds, h->stream, ra);     \
            else hipLaunchKernelGGL((KERNEL<FT, 0, false>), grid, block, lds, h->stream, ra);             \
        }                                                                                                 \
    } while (0)
    SITRK_LAUNCH_RUN(advect_run_kernel);
#undef SITRK_LAUNCH_RUN
}

SITRK_API int sitrk_run(sitrk_t *h, int slot0, int jrec0, int nsteps)
{
    NEED(h, "null handle");
    NEED(nsteps >= 0, "sitrk_run: nsteps must be >= 0");
    NEED(h->nslots > 0, "sitrk_run: call sitrk_alloc_records first");
    NEED(slot0 >= 0 && slot0 < h->nslots, "sitrk_run: slot0 out of range");
    NEED(h->st[0].pos, "sitrk_run: call sitrk_set_buoys first");
    if (h->nP == 0) return SITRK_OK;
    int fuse = std::max(1, std::min(std::min(h->fuse, kMaxFuse), h->nslots));    // a launch never wraps the slot ring
    // the fused kernel addresses the geometry with 32-bit byte offsets and has no negative-index wrap: buoy sets seeded in
    // the two outermost rows/columns (the reference itself cancels such seeds, tracking.py:73) and meshes beyond
    // 2^32 / 48 cells (9 460 x 9 460) are stepped record by record
    if (h->rim_buoys || (uint64_t)h->Nj * h->Ni * sizeof(CellGeo) >= ((uint64_t)1 << 32)) fuse = 1;
    const size_t n = (size_t)h->Nj * h->Ni, es = elem_size(h->dtype);
    int k = 0;
    while (k < nsteps) {
        if (h->resort_every > 0 && h->steps_since_sort >= h->resort_every) {
            int rc = sitrk_sort_buoys(h);
            if (rc) return rc;
        }
        int m = std::min(fuse, nsteps - k);
        if (h->resort_every > 0) m = std::min(m, h->resort_every - h->steps_since_sort);
        if (m <= 1) {
            int rc = sitrk_step(h, (slot0 + k) % h->nslots, jrec0 + k);
            if (rc) return rc;
            k += 1;
            continue;
        }
        // m consecutive records, all resident in distinct slots, in one launch
        BuoyState &s = h->st[h->cur];
        RunArgs ra;
        ra.s.nP = h->nP; ra.s.tune = h->tune; ra.s.Nj = h->Nj; ra.s.Ni = h->Ni; ra.s.jrec = jrec0 + k;
        ra.s.rdt = h->rdt; ra.s.rmin_conc = h->rmin_conc; ra.s.eps_mg = h->eps_mg; ra.s.geo = h->geo; ra.s.orient = h->orient; ra.s.kill = nullptr; ra.s.u = ra.s.v = nullptr;
        ra.s.pos = s.pos; ra.s.cell = s.cell; ra.s.kill_rec = s.kill_rec; ra.s.win = s.win;
        ra.nrec = m;
        make_cross_tab(h->Ni, ra.tab, ra.dji);
        ra.geoF = h->geoF;
        ra.patch_cells = (int)((size_t)h->patch_kb * 1024 / sizeof(pt));
        ra.patch_margin = h->patch_margin;
        ra.xcd_group = h->xcd_group;
        ra.f32_class = f32_class_for(h->rdt);
#ifdef SITRK_DIAG
        ra.stamps = nullptr;
        if (h->stamps_on) {
            const size_t nw = (size_t)nblocks(h->nP, kRunBlock) * (kRunBlock / 64);
            if (nw > h->stamps_waves) {
                dev_free(h->stamps); h->stamps = nullptr; h->stamps_waves = 0;
                HIPCHK(hipMalloc((void **)&h->stamps, nw * 8 * sizeof(unsigned long long)));
                h->stamps_waves = nw;
            }
            HIPCHK(hipMemsetAsync(h->stamps, 0, h->stamps_waves * 8 * sizeof(unsigned long long), h->stream));
            ra.stamps = h->stamps;
        }
#endif
        int used[kMaxFuse];
        for (int r = 0; r < m; r++) {
            const int slot = (slot0 + k + r) % h->nslots;
            used[r] = slot;
            int rc = check_band(h, slot, r);
            if (rc) return rc;
            if (h->slot_dirty[slot]) {
                rc = derive_mask(h, slot);
                if (rc) return rc;
            }
            rc = slot_wait_upload(h, slot);
            if (rc) return rc;
            rc = slot_wait_sv(h, slot);
            if (rc) return rc;
            const char *slab = slab_of(h, slot);
            ra.u[r] = slab; ra.v[r] = slab + n * es; ra.kill9[r] = h->kill9 + (size_t)slot * n;
        }
        if (h->dtype == SITRK_F64) launch_run<double>(h, ra);
        else launch_run<float>(h, ra);
        HIPCHK(hipGetLastError());
        int rc = launch_mark(h, used, m);
        if (rc) return rc;
        h->steps_since_sort += m;
        h->n_fused_launches++;
        h->n_fused_records += m;
        if (h->band_age >= 0) h->band_age += m;
        if (h->box_pending) h->box_pending_age += m;
        k += m;
    }
    return SITRK_OK;
}

#ifdef SITRK_DIAG
// diagnostic builds only (not in include/sitrk.h): the s_memtime intervals of the last fused launch, 8 per wave
SITRK_API int sitrk_diag_stamps(sitrk_t *h, unsigned long long *out, long long max_waves, long long *nwaves)
{
    NEED(h, "null handle");
    NEED(out && nwaves, "sitrk_diag_stamps: null output");
    HIPCHK(hipStreamSynchronize(h->stream));
    const long long n = std::min<long long>((long long)h->stamps_waves, max_waves);
    *nwaves = n;
    if (n > 0) HIPCHK(hipMemcpy(out, h->stamps, (size_t)n * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return SITRK_OK;
}
#endif

SITRK_API int sitrk_launch_stats(sitrk_t *h, int reset, int64_t *fused_launches, int64_t *fused_records, int64_t *step_launches)
{
    NEED(h, "null handle");
    if (fused_launches) *fused_launches = h->n_fused_launches;
    if (fused_records) *fused_records = h->n_fused_records;
    if (step_launches) *step_launches = h->n_step_launches;
    if (reset) h->n_fused_launches = h->n_fused_records = h->n_step_launches = 0;
    return SITRK_OK;
}

// --------------------------------------------------------------------------- fetch
SITRK_API int sitrk_fetch(sitrk_t *h, double *yx, int32_t *jiT, int8_t *alive, int32_t *kill_rec)
{
    NEED(h, "null handle");
    NEED(h->st[0].pos, "sitrk_fetch: call sitrk_set_buoys first");
    const int64_t nP = h->nP;
    if (nP == 0) return SITRK_OK;
    HIPCHK(hipSetDevice(h->device));
    const size_t b_yx = align256((size_t)nP * sizeof(pt)), b_ji = align256((size_t)nP * 8), b_al = align256((size_t)nP),
                 b_kr = align256((size_t)nP * 4);
    int rc = ensure_scratch(h, b_yx + b_ji + b_al + b_kr);
    if (rc) return rc;
    char *s = (char *)h->scratch;
    pt *d_yx = yx ? (pt *)s : nullptr;
    int32_t *d_ji = jiT ? (int32_t *)(s + b_yx) : nullptr;
    int8_t *d_al = alive ? (int8_t *)(s + b_yx + b_ji) : nullptr;
    int32_t *d_kr = kill_rec ? (int32_t *)(s + b_yx + b_ji + b_al) : nullptr;
    hipLaunchKernelGGL(fetch_state_kernel, dim3(nblocks(nP)), dim3(kBlock), 0, h->stream, nP, h->st[h->cur], d_yx, d_ji, d_al, d_kr);
    HIPCHK(hipGetLastError());
    if (yx) HIPCHK(hipMemcpyAsync(yx, d_yx, (size_t)nP * sizeof(pt), hipMemcpyDeviceToHost, h->stream));
    if (jiT) HIPCHK(hipMemcpyAsync(jiT, d_ji, (size_t)nP * 8, hipMemcpyDeviceToHost, h->stream));
    if (alive) HIPCHK(hipMemcpyAsync(alive, d_al, (size_t)nP, hipMemcpyDeviceToHost, h->stream));
    if (kill_rec) HIPCHK(hipMemcpyAsync(kill_rec, d_kr, (size_t)nP * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SITRK_OK;
}

static ProjParams make_proj(double lat0, double lon0)
{
    ProjParams pp;
    const double f = 1.0 / 298.257223563;            // WGS84
    pp.a = 6378137.0;
    pp.e = std::sqrt(2.0 * f - f * f);
    pp.lon0 = lon0;
    const double phits = std::fabs(lat0) * (M_PI / 180.0);
    if (std::fabs(phits - M_PI_2) < 1e-10) {
        pp.akm1 = 2.0 / std::sqrt(std::pow(1 + pp.e, 1 + pp.e) * std::pow(1 - pp.e, 1 - pp.e));
    } else {
        double t = std::sin(phits);
        double es = pp.e * t;
        double tsfn = std::tan(0.5 * (M_PI_2 - phits)) / std::pow((1.0 - es) / (1.0 + es), 0.5 * pp.e);
        pp.akm1 = std::cos(phits) / tsfn;
        pp.akm1 /= std::sqrt(1.0 - es * es);
    }
    return pp;
}

SITRK_API int sitrk_fetch_record(sitrk_t *h, int jrec, double *yx_rec, int8_t *mask, double *latlon)
{
    NEED(h, "null handle");
    NEED(h->st[0].pos, "sitrk_fetch_record: call sitrk_set_buoys first");
    const int64_t nP = h->nP;
    if (nP == 0) return SITRK_OK;
    HIPCHK(hipSetDevice(h->device));
    const size_t b_yx = align256((size_t)nP * sizeof(pt)), b_mk = align256((size_t)nP);
    int rc = ensure_scratch(h, 2 * b_yx + b_mk);
    if (rc) return rc;
    char *s = (char *)h->scratch;
    pt *d_yx = (pt *)s;
    int8_t *d_mk = (int8_t *)(s + b_yx);
    ll *d_ll = (ll *)(s + b_yx + b_mk);
    hipLaunchKernelGGL(fetch_record_kernel, dim3(nblocks(nP)), dim3(kBlock), 0, h->stream, nP, jrec, h->st[h->cur], h->windowed, d_yx, d_mk);
    HIPCHK(hipGetLastError());
    if (latlon) {
        hipLaunchKernelGGL(cart2geo_kernel, dim3(nblocks(nP)), dim3(kBlock), 0, h->stream, nP, make_proj(70., -45.), d_yx, d_ll);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(latlon, d_ll, (size_t)nP * sizeof(ll), hipMemcpyDeviceToHost, h->stream));
    }
    if (yx_rec) HIPCHK(hipMemcpyAsync(yx_rec, d_yx, (size_t)nP * sizeof(pt), hipMemcpyDeviceToHost, h->stream));
    if (mask) HIPCHK(hipMemcpyAsync(mask, d_mk, (size_t)nP, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SITRK_OK;
}

SITRK_API int sitrk_count_alive(sitrk_t *h, int64_t *nalive)
{
    NEED(h, "null handle");
    NEED(nalive, "sitrk_count_alive: null output");
    *nalive = 0;
    if (h->nP == 0) return SITRK_OK;
    NEED(h->st[0].pos, "sitrk_count_alive: call sitrk_set_buoys first");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemsetAsync(h->counter, 0, sizeof(unsigned long long), h->stream));
    hipLaunchKernelGGL(count_alive_kernel, dim3(std::min(nblocks(h->nP), 2048u)), dim3(kBlock), 0, h->stream, h->nP, h->st[h->cur].cell, h->counter);
    HIPCHK(hipGetLastError());
    unsigned long long v = 0;
    HIPCHK(hipMemcpyAsync(&v, h->counter, sizeof(v), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *nalive = (int64_t)v;
    return SITRK_OK;
}

// --------------------------------------------------------------------------- locate
SITRK_API int sitrk_find_cells(sitrk_t *h, int64_t n, const double *yx, const int32_t *jiT_guess, int32_t *jiT_out, int8_t *found)
{
    NEED(h, "null handle");
    NEED(h->geo, "sitrk_find_cells: call sitrk_set_grid first");
    NEED(n >= 0, "sitrk_find_cells: n < 0");
    if (n == 0) return SITRK_OK;
    NEED(yx && jiT_guess && jiT_out && found, "sitrk_find_cells: null array");
    HIPCHK(hipSetDevice(h->device));
    const size_t b_yx = align256((size_t)n * sizeof(pt)), b_ji = align256((size_t)n * 8), b_f = align256((size_t)n);
    int rc = ensure_scratch(h, b_yx + 2 * b_ji + b_f);
    if (rc) return rc;
    char *s = (char *)h->scratch;
    pt *d_yx = (pt *)s;
    int32_t *d_g = (int32_t *)(s + b_yx), *d_o = (int32_t *)(s + b_yx + b_ji);
    int8_t *d_f = (int8_t *)(s + b_yx + 2 * b_ji);
    HIPCHK(hipMemcpyAsync(d_yx, yx, (size_t)n * sizeof(pt), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(d_g, jiT_guess, (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(find_cells_kernel, dim3(nblocks(n)), dim3(kBlock), 0, h->stream, n, h->Nj, h->Ni, h->geo, d_yx, d_g, d_o, d_f);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(jiT_out, d_o, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(found, d_f, (size_t)n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SITRK_OK;
}

// Nearest T-point of nP seeds by exact branch-and-bound over bounding spheres of the mesh (sitrk_locate.h):
// kernels are enqueued on h->stream; *kb / *db (flat index, Haversine distance) live in the work buffer *w, which the
// caller frees after synchronising.  Seeds farther than anything NearestPoint's acceptance loop can accept get
// kb = 0xffffffff without a search.
static int nearest_search(sitrk_ctx *h, int64_t nP, const ll *d_ll, const double *d_lat, const double *d_lon, const double *resolkm_host,
                          double rd_found_km, int max_itr, char **w_out, uint32_t **kb_out, double **db_out)
{
    const size_t n = (size_t)h->Nj * h->Ni;
    const int nbj = (h->Nj + kLB - 1) / kLB, nbi = (h->Ni + kLB - 1) / kLB, sbf = 16;
    const int nsj = (nbj + sbf - 1) / sbf, nsi = (nbi + sbf - 1) / sbf;
    const size_t b_u = align256(n * 8), b_blk = align256((size_t)nbj * nbi * sizeof(Sphere)),
                 b_sb = align256((size_t)nsj * nsi * sizeof(Sphere)), b_kb = align256((size_t)nP * 4), b_db = align256((size_t)nP * 8);
    char *w = nullptr;
    HIPCHK(hipMalloc((void **)&w, 3 * b_u + b_blk + b_sb + b_kb + b_db));
    double *ux = (double *)w, *uy = (double *)(w + b_u), *uz = (double *)(w + 2 * b_u);
    Sphere *blk = (Sphere *)(w + 3 * b_u), *sblk = (Sphere *)(w + 3 * b_u + b_blk);
    uint32_t *kb = (uint32_t *)(w + 3 * b_u + b_blk + b_sb);
    double *db = (double *)(w + 3 * b_u + b_blk + b_sb + b_kb);
    hipLaunchKernelGGL(unitvec_kernel, dim3(nblocks((int64_t)n)), dim3(kBlock), 0, h->stream, n, d_lat, d_lon, ux, uy, uz);
    hipLaunchKernelGGL(block_sphere_kernel, dim3((unsigned)(nbj * nbi)), dim3(kBlock), 0, h->stream, h->Nj, h->Ni, nbi, ux, uy, uz, blk);
    hipLaunchKernelGGL(superblock_sphere_kernel, dim3((unsigned)(nsj * nsi)), dim3(kBlock), 0, h->stream, nbj, nbi, sbf, nsi, blk, sblk);
    SearchArgs sa;
    sa.nP = nP; sa.Nj = h->Nj; sa.Ni = h->Ni; sa.nbj = nbj; sa.nbi = nbi; sa.sbf = sbf; sa.nsj = nsj; sa.nsi = nsi;
    sa.latlon = d_ll; sa.ux = ux; sa.uy = uy; sa.uz = uz; sa.blk = blk; sa.sblk = sblk; sa.latT = d_lat; sa.lonT = d_lon;
    sa.kbest = kb; sa.dbest = db;
    {
        // largest distance the acceptance loop of NearestPoint can ever accept (locate.py:253-266):
        // rfnd starts at 0.5*resol (or rd_found_km) and is multiplied by 1.2 at most max_itr-2 times
        double rmax = rd_found_km;
        if (resolkm_host) {
            rmax = 0.0;
            for (size_t c = 0; c < n; c++) rmax = std::max(rmax, 0.5 * resolkm_host[c]);
        }
        const double dmax = rmax * std::pow(1.2, std::max(0, max_itr - 2)) * 1.02;   // km on the R = 6360 km Haversine sphere, +2 %
        const double half = std::min(dmax / (2.0 * 6360.0), M_PI_2);
        const double chord = 2.0 * std::sin(half);
        sa.far2 = (std::isfinite(dmax) && dmax >= 0.0) ? chord * chord : __builtin_inf();
    }
    hipLaunchKernelGGL(seed_search_kernel, dim3(nblocks(nP, kBlock / 64)), dim3(kBlock), 0, h->stream, sa);
    *w_out = w; *kb_out = kb; *db_out = db;
    return SITRK_OK;
}

SITRK_API int sitrk_seed_init(sitrk_t *h, int64_t nP, const double *latlon, const double *yx, const double *latT,
                              const double *lonT, const double *resolkm, const double *sic, int32_t *jiT_out, int8_t *keep,
                              int8_t *why)
{
    NEED(h, "null handle");
    NEED(h->geo, "sitrk_seed_init: call sitrk_set_grid first");
    NEED(nP >= 0, "sitrk_seed_init: nP < 0");
    if (nP == 0) return SITRK_OK;
    NEED(latlon && yx && latT && lonT && sic && jiT_out && keep, "sitrk_seed_init: null array");
    HIPCHK(hipSetDevice(h->device));
    const size_t n = (size_t)h->Nj * h->Ni;
    const size_t b_pt = align256((size_t)nP * sizeof(pt)), b_g = align256(n * 8), b_ji = align256((size_t)nP * 8),
                 b_k = align256((size_t)nP);
    int rc = ensure_scratch(h, 2 * b_pt + 4 * b_g + b_ji + 2 * b_k);
    if (rc) return rc;
    char *s = (char *)h->scratch;
    ll *d_ll = (ll *)s;                 s += b_pt;
    pt *d_yx = (pt *)s;                 s += b_pt;
    double *d_lat = (double *)s;        s += b_g;
    double *d_lon = (double *)s;        s += b_g;
    double *d_res = (double *)s;        s += b_g;
    double *d_sic = (double *)s;        s += b_g;
    int32_t *d_ji = (int32_t *)s;       s += b_ji;
    int8_t *d_keep = (int8_t *)s;       s += b_k;
    int8_t *d_why = (int8_t *)s;
    HIPCHK(hipMemcpyAsync(d_ll, latlon, (size_t)nP * sizeof(ll), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(d_yx, yx, (size_t)nP * sizeof(pt), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(d_lat, latT, n * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(d_lon, lonT, n * 8, hipMemcpyHostToDevice, h->stream));
    if (resolkm) HIPCHK(hipMemcpyAsync(d_res, resolkm, n * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(d_sic, sic, n * 8, hipMemcpyHostToDevice, h->stream));
    // rFoundKM = 2.5 (tracking.py:5), max_itr = 10 (tracking.py:134)
    if (h->tune & TUNE_LOCATE_BRUTEFORCE) {
        hipLaunchKernelGGL(seed_init_bruteforce_kernel, dim3((unsigned)nP), dim3(kBlock), 0, h->stream, nP, h->Nj, h->Ni, d_ll, d_yx,
                           d_lat, d_lon, resolkm ? d_res : nullptr, d_sic, h->tmask, h->geo, h->rmin_conc, 2.5, 10, d_ji, d_keep, d_why);
        HIPCHK(hipGetLastError());
    } else {
        // exact branch-and-bound over bounding spheres of the mesh (sitrk_locate.h)
        char *w = nullptr;
        uint32_t *kb = nullptr;
        double *db = nullptr;
        rc = nearest_search(h, nP, d_ll, d_lat, d_lon, resolkm, 2.5, 10, &w, &kb, &db);
        if (rc) return rc;
        hipLaunchKernelGGL(seed_finish_kernel, dim3(nblocks(nP)), dim3(kBlock), 0, h->stream, nP, h->Nj, h->Ni, kb, db, d_yx,
                           resolkm ? d_res : nullptr, d_sic, h->tmask, h->geo, h->rmin_conc, 2.5, 10, d_ji, d_keep, d_why);
        hipError_t le = hipGetLastError();
        hipError_t se = hipStreamSynchronize(h->stream);
        (void)hipFree(w);
        if (le != hipSuccess) return fail(h, SITRK_EHIP, "seed search launch -> %s", hipGetErrorString(le));
        if (se != hipSuccess) return fail(h, SITRK_EHIP, "seed search -> %s", hipGetErrorString(se));
    }
    HIPCHK(hipMemcpyAsync(jiT_out, d_ji, (size_t)nP * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(keep, d_keep, (size_t)nP, hipMemcpyDeviceToHost, h->stream));
    if (why) HIPCHK(hipMemcpyAsync(why, d_why, (size_t)nP, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SITRK_OK;
}

SITRK_API int sitrk_nearest_point(sitrk_t *h, int64_t nP, const double *latlon, const double *latT, const double *lonT,
                                  const double *resolkm, double rd_found_km, int max_itr, int32_t *ji_out, double *dmin_out)
{
    NEED(h, "null handle");
    NEED(h->geo, "sitrk_nearest_point: call sitrk_set_grid first");
    NEED(nP >= 0, "sitrk_nearest_point: nP < 0");
    NEED(max_itr >= 1, "sitrk_nearest_point: max_itr must be >= 1");
    if (nP == 0) return SITRK_OK;
    NEED(latlon && latT && lonT && ji_out, "sitrk_nearest_point: null array");
    HIPCHK(hipSetDevice(h->device));
    const size_t n = (size_t)h->Nj * h->Ni;
    const size_t b_pt = align256((size_t)nP * sizeof(ll)), b_g = align256(n * 8), b_ji = align256((size_t)nP * 8),
                 b_d = align256((size_t)nP * 8);
    int rc = ensure_scratch(h, b_pt + 3 * b_g + b_ji + b_d);
    if (rc) return rc;
    char *s = (char *)h->scratch;
    ll *d_ll = (ll *)s;                 s += b_pt;
    double *d_lat = (double *)s;        s += b_g;
    double *d_lon = (double *)s;        s += b_g;
    double *d_res = (double *)s;        s += b_g;
    int32_t *d_ji = (int32_t *)s;       s += b_ji;
    double *d_dm = (double *)s;
    HIPCHK(hipMemcpyAsync(d_ll, latlon, (size_t)nP * sizeof(ll), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(d_lat, latT, n * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(d_lon, lonT, n * 8, hipMemcpyHostToDevice, h->stream));
    if (resolkm) HIPCHK(hipMemcpyAsync(d_res, resolkm, n * 8, hipMemcpyHostToDevice, h->stream));
    char *w = nullptr;
    uint32_t *kb = nullptr;
    double *db = nullptr;
    rc = nearest_search(h, nP, d_ll, d_lat, d_lon, resolkm, rd_found_km, max_itr, &w, &kb, &db);
    if (rc) return rc;
    hipLaunchKernelGGL(nearest_finish_kernel, dim3(nblocks(nP)), dim3(kBlock), 0, h->stream, nP, h->Ni, kb, db,
                       resolkm ? d_res : nullptr, rd_found_km, max_itr, d_ji, d_dm);
    hipError_t le = hipGetLastError();
    hipError_t se = hipStreamSynchronize(h->stream);
    (void)hipFree(w);
    if (le != hipSuccess) return fail(h, SITRK_EHIP, "nearest-point launch -> %s", hipGetErrorString(le));
    if (se != hipSuccess) return fail(h, SITRK_EHIP, "nearest-point search -> %s", hipGetErrorString(se));
    HIPCHK(hipMemcpyAsync(ji_out, d_ji, (size_t)nP * 8, hipMemcpyDeviceToHost, h->stream));
    if (dmin_out) HIPCHK(hipMemcpyAsync(dmin_out, d_dm, (size_t)nP * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SITRK_OK;
}

SITRK_API int sitrk_eval_haversine(sitrk_t *h, int64_t n, const double *plat, const double *plon, const double *xlat,
                                   const double *xlon, double *dist)
{
    NEED(h, "null handle");
    NEED(n >= 0, "sitrk_eval_haversine: n < 0");
    if (n == 0) return SITRK_OK;
    NEED(plat && plon && xlat && xlon && dist, "sitrk_eval_haversine: null array");
    HIPCHK(hipSetDevice(h->device));
    const size_t b = align256((size_t)n * 8);
    int rc = ensure_scratch(h, 5 * b);
    if (rc) return rc;
    char *s = (char *)h->scratch;
    const double *src[4] = {plat, plon, xlat, xlon};
    for (int a = 0; a < 4; a++) HIPCHK(hipMemcpyAsync(s + a * b, src[a], (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(eval_haversine_kernel, dim3(nblocks(n)), dim3(kBlock), 0, h->stream, n, (const double *)s, (const double *)(s + b),
                       (const double *)(s + 2 * b), (const double *)(s + 3 * b), (double *)(s + 4 * b));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(dist, s + 4 * b, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SITRK_OK;
}

// --------------------------------------------------------------------------- predicate probes
SITRK_API int sitrk_eval_inside(sitrk_t *h, int64_t n, const double *pts, const double *quads, int8_t *inside)
{
    NEED(h, "null handle");
    NEED(n >= 0, "sitrk_eval_inside: n < 0");
    if (n == 0) return SITRK_OK;
    NEED(pts && quads && inside, "sitrk_eval_inside: null array");
    HIPCHK(hipSetDevice(h->device));
    const size_t b_p = align256((size_t)n * sizeof(pt)), b_q = align256((size_t)n * 4 * sizeof(pt)), b_o = align256((size_t)n);
    int rc = ensure_scratch(h, b_p + b_q + b_o);
    if (rc) return rc;
    char *s = (char *)h->scratch;
    HIPCHK(hipMemcpyAsync(s, pts, (size_t)n * sizeof(pt), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(s + b_p, quads, (size_t)n * 4 * sizeof(pt), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(eval_inside_kernel, dim3(nblocks(n)), dim3(kBlock), 0, h->stream, n, (const pt *)s, (const pt *)(s + b_p),
                       (int8_t *)(s + b_p + b_q));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(inside, s + b_p + b_q, (size_t)n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SITRK_OK;
}

SITRK_API int sitrk_eval_euler(sitrk_t *h, int64_t n, const double *r, const double *vel, double rdt, double *out)
{
    NEED(h, "null handle");
    NEED(n >= 0, "sitrk_eval_euler: n < 0");
    if (n == 0) return SITRK_OK;
    NEED(r && vel && out, "sitrk_eval_euler: null array");
    HIPCHK(hipSetDevice(h->device));
    const size_t b = align256((size_t)n * sizeof(double));
    int rc = ensure_scratch(h, 3 * b);
    if (rc) return rc;
    char *s = (char *)h->scratch;
    HIPCHK(hipMemcpyAsync(s, r, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(s + b, vel, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(eval_euler_kernel, dim3(nblocks(n)), dim3(kBlock), 0, h->stream, n, (const double *)s, (const double *)(s + b),
                       rdt, f32_class_for(rdt), (double *)(s + 2 * b));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, s + 2 * b, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SITRK_OK;
}

SITRK_API int sitrk_eval_intersect(sitrk_t *h, int64_t n, const double *segs, int8_t *intersect, int8_t *ccw_abc)
{
    NEED(h, "null handle");
    NEED(n >= 0, "sitrk_eval_intersect: n < 0");
    if (n == 0) return SITRK_OK;
    NEED(segs && intersect, "sitrk_eval_intersect: null array");
    HIPCHK(hipSetDevice(h->device));
    const size_t b_s = align256((size_t)n * 4 * sizeof(pt)), b_o = align256((size_t)n);
    int rc = ensure_scratch(h, b_s + 2 * b_o);
    if (rc) return rc;
    char *s = (char *)h->scratch;
    HIPCHK(hipMemcpyAsync(s, segs, (size_t)n * 4 * sizeof(pt), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(eval_intersect_kernel, dim3(nblocks(n)), dim3(kBlock), 0, h->stream, n, (const pt *)s, (int8_t *)(s + b_s),
                       ccw_abc ? (int8_t *)(s + b_s + b_o) : nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(intersect, s + b_s, (size_t)n, hipMemcpyDeviceToHost, h->stream));
    if (ccw_abc) HIPCHK(hipMemcpyAsync(ccw_abc, s + b_s + b_o, (size_t)n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SITRK_OK;
}

SITRK_API int sitrk_eval_crossing(sitrk_t *h, int64_t n, const double *P1, const double *P2, const int32_t *jiT, int32_t *jiT_new,
                                  int32_t *codes)
{
    NEED(h, "null handle");
    NEED(h->geo, "sitrk_eval_crossing: call sitrk_set_grid first");
    NEED(n >= 0, "sitrk_eval_crossing: n < 0");
    if (n == 0) return SITRK_OK;
    NEED(P1 && P2 && jiT && jiT_new, "sitrk_eval_crossing: null array");
    for (int64_t p = 0; p < n; p++)
        if (jiT[2 * p] < 1 || jiT[2 * p] > h->Nj - 2 || jiT[2 * p + 1] < 1 || jiT[2 * p + 1] > h->Ni - 2)
            return fail(h, SITRK_EINDEX, "sitrk_eval_crossing: host cell (%d,%d) outside 1..%d x 1..%d", jiT[2 * p], jiT[2 * p + 1],
                        h->Nj - 2, h->Ni - 2);
    HIPCHK(hipSetDevice(h->device));
    const size_t b_p = align256((size_t)n * sizeof(pt)), b_j = align256((size_t)n * 8);
    int rc = ensure_scratch(h, 2 * b_p + 3 * b_j);
    if (rc) return rc;
    char *s = (char *)h->scratch;
    pt *d1 = (pt *)s, *d2 = (pt *)(s + b_p);
    int32_t *dj = (int32_t *)(s + 2 * b_p), *dn = (int32_t *)(s + 2 * b_p + b_j), *dc = (int32_t *)(s + 2 * b_p + 2 * b_j);
    HIPCHK(hipMemcpyAsync(d1, P1, (size_t)n * sizeof(pt), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(d2, P2, (size_t)n * sizeof(pt), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(dj, jiT, (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
    CrossTab tab;
    make_cross_tab(h->Ni, tab);
    hipLaunchKernelGGL(eval_crossing_kernel, dim3(nblocks(n)), dim3(kBlock), 0, h->stream, n, h->Nj, h->Ni, h->geo, d1, d2, dj, dn,
                       codes ? dc : nullptr, tab);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(jiT_new, dn, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
    if (codes) HIPCHK(hipMemcpyAsync(codes, dc, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SITRK_OK;
}

SITRK_API int sitrk_survive_mask(sitrk_t *h, const double *sic, int8_t *mask)
{
    NEED(h, "null handle");
    NEED(h->geo, "sitrk_survive_mask: call sitrk_set_grid first");
    NEED(sic && mask, "sitrk_survive_mask: null array");
    HIPCHK(hipSetDevice(h->device));
    const size_t cells = (size_t)h->Nj * h->Ni;
    const size_t b_s = align256(cells * 8), b_m = align256(cells);
    int rc = ensure_scratch(h, b_s + 2 * b_m);
    if (rc) return rc;
    char *s = (char *)h->scratch;
    HIPCHK(hipMemcpyAsync(s, sic, cells * 8, hipMemcpyHostToDevice, h->stream));
    // the very kernel that derives a resident record's bytes (the packed neighbourhoods go to scratch and are dropped)
    rc = launch_survive(h, true, s, (int8_t *)(s + b_s), (uint8_t *)(s + b_s + b_m), 0, h->Nj, 0, h->Ni);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(mask, s + b_s, cells, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SITRK_OK;
}

// --------------------------------------------------------------------------- projection
static int project(sitrk_ctx *h, int64_t n, const double *in, double lat0, double lon0, double *out, bool inverse)
{
    NEED(h, "null handle");
    NEED(n >= 0, "projection: n < 0");
    if (n == 0) return SITRK_OK;
    NEED(in && out, "projection: null array");
    HIPCHK(hipSetDevice(h->device));
    const size_t b = align256((size_t)n * 16);
    int rc = ensure_scratch(h, 2 * b);
    if (rc) return rc;
    char *s = (char *)h->scratch;
    HIPCHK(hipMemcpyAsync(s, in, (size_t)n * 16, hipMemcpyHostToDevice, h->stream));
    ProjParams pp = make_proj(lat0, lon0);
    if (inverse) hipLaunchKernelGGL(cart2geo_kernel, dim3(nblocks(n)), dim3(kBlock), 0, h->stream, n, pp, (const pt *)s, (ll *)(s + b));
    else hipLaunchKernelGGL(geo2cart_kernel, dim3(nblocks(n)), dim3(kBlock), 0, h->stream, n, pp, (const ll *)s, (pt *)(s + b));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, s + b, (size_t)n * 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SITRK_OK;
}

SITRK_API int sitrk_cart2geo(sitrk_t *h, int64_t n, const double *yx, double lat0, double lon0, double *latlon)
{
    return project(h, n, yx, lat0, lon0, latlon, true);
}

SITRK_API int sitrk_geo2cart(sitrk_t *h, int64_t n, const double *latlon, double lat0, double lon0, double *yx)
{
    return project(h, n, latlon, lat0, lon0, yx, false);
}

// --------------------------------------------------------------------------- idealised seeding
SITRK_API int sitrk_nemo_seed(sitrk_t *h, int Nj, int Ni, int khss, const int8_t *tmask, const int8_t *rmask, const double *latT,
                              const double *lonT, const double *sic, const double *latF, const double *lonF, double lat0, double lon0,
                              int64_t capacity, double *latlon, double *yx, int64_t *nT, int64_t *nF)
{
    NEED(h, "null handle");
    NEED(Nj >= 1 && Ni >= 1 && (int64_t)Nj * Ni < ((int64_t)1 << 31), "sitrk_nemo_seed: bad mesh shape");
    NEED(khss >= 1, "sitrk_nemo_seed: khss must be >= 1");
    NEED(tmask && latT && lonT && sic, "sitrk_nemo_seed: null array");
    NEED((latF == nullptr) == (lonF == nullptr), "sitrk_nemo_seed: latF and lonF go together");
    NEED(nT && nF, "sitrk_nemo_seed: null count output");
    NEED(capacity >= 0 && (capacity == 0 || latlon), "sitrk_nemo_seed: capacity without an output array");
    HIPCHK(hipSetDevice(h->device));
    SeedArgs s;
    s.Nj = Nj; s.Ni = Ni; s.khss = khss;
    s.Njs = (Nj + khss - 1) / khss; s.Nis = (Ni + khss - 1) / khss;        // shape of array[::khss, ::khss]
    s.with_f = latF ? 1 : 0;
    const size_t n = (size_t)Nj * Ni;
    const int64_t ns = (int64_t)s.Njs * s.Nis, nblk_t = (ns + kSeedBlock - 1) / kSeedBlock, nblk = 2 * nblk_t;
    const size_t b_d = align256(n * 8), b_m = align256(n), b_c = align256((size_t)nblk * 4), b_o = align256((size_t)nblk * 8),
                 b_out = align256((size_t)capacity * 16);
    int rc = ensure_scratch(h, 5 * b_d + 2 * b_m + b_c + b_o + 256 + 2 * b_out);
    if (rc) return rc;
    char *w = (char *)h->scratch;
    double *d_latT = (double *)w;           w += b_d;
    double *d_lonT = (double *)w;           w += b_d;
    double *d_sic = (double *)w;            w += b_d;
    double *d_latF = (double *)w;           w += b_d;
    double *d_lonF = (double *)w;           w += b_d;
    int8_t *d_tm = (int8_t *)w;             w += b_m;
    int8_t *d_rm = (int8_t *)w;             w += b_m;
    unsigned *d_cnt = (unsigned *)w;        w += b_c;
    int64_t *d_off = (int64_t *)w;          w += b_o;
    int64_t *d_tot = (int64_t *)w;          w += 256;
    ll *d_ll = (ll *)w;                     w += b_out;
    pt *d_yx = (pt *)w;
    HIPCHK(hipMemcpyAsync(d_latT, latT, n * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(d_lonT, lonT, n * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(d_sic, sic, n * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(d_tm, tmask, n, hipMemcpyHostToDevice, h->stream));
    if (rmask) HIPCHK(hipMemcpyAsync(d_rm, rmask, n, hipMemcpyHostToDevice, h->stream));
    if (latF) {
        HIPCHK(hipMemcpyAsync(d_latF, latF, n * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(d_lonF, lonF, n * 8, hipMemcpyHostToDevice, h->stream));
    }
    s.tmask = d_tm; s.rmask = rmask ? d_rm : nullptr;
    s.latT = d_latT; s.lonT = d_lonT; s.sic = d_sic; s.latF = d_latF; s.lonF = d_lonF;
    hipLaunchKernelGGL(seed_count_kernel, dim3((unsigned)nblk), dim3(kSeedBlock), 0, h->stream, s, nblk_t, d_cnt);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(seed_scan_kernel, dim3(1), dim3(kSeedBlock), 0, h->stream, nblk, nblk_t, d_cnt, d_off, d_tot);
    HIPCHK(hipGetLastError());
    int64_t tot[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(tot, d_tot, sizeof(tot), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *nT = tot[0]; *nF = tot[1];
    if (capacity == 0) return SITRK_OK;                 // counting call
    if (capacity < tot[0] + tot[1])
        return fail(h, SITRK_EINVAL, "sitrk_nemo_seed: %lld seeds do not fit the capacity %lld", (long long)(tot[0] + tot[1]), (long long)capacity);
    hipLaunchKernelGGL(seed_emit_kernel, dim3((unsigned)nblk), dim3(kSeedBlock), 0, h->stream, s, nblk_t, d_off, make_proj(lat0, lon0), capacity,
                       d_ll, yx ? d_yx : nullptr);
    HIPCHK(hipGetLastError());
    const size_t nout = (size_t)(tot[0] + tot[1]);
    HIPCHK(hipMemcpyAsync(latlon, d_ll, nout * 16, hipMemcpyDeviceToHost, h->stream));
    if (yx) HIPCHK(hipMemcpyAsync(yx, d_yx, nout * 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SITRK_OK;
}

// --------------------------------------------------------------------------- measurement
SITRK_API int sitrk_timer_start(sitrk_t *h)
{
    NEED(h, "null handle");
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    return SITRK_OK;
}

SITRK_API int sitrk_timer_stop(sitrk_t *h, float *ms)
{
    NEED(h, "null handle");
    NEED(ms, "sitrk_timer_stop: null output");
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipEventSynchronize(h->ev1));
    HIPCHK(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return SITRK_OK;
}
