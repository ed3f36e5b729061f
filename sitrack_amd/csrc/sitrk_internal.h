// sitrk_internal.h -- context layout shared by the translation units of libsitrk.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "sitrk_geom.h"

namespace sitrk {

hipError_t sort_pairs_u32(void *tmp, size_t *tmp_bytes, const uint32_t *kin, uint32_t *kout,
                          const int32_t *vin, int32_t *vout, size_t n, unsigned end_bit, hipStream_t s);

// Device-resident buoy state, structure of arrays, in SORTED slot order.
// perm[s] = index of slot s in the caller's order.
struct BuoyState {
    pt      *pos      = nullptr;   // (y,x) km, current position           16 B
    int32_t *cell     = nullptr;   // packed (jT,iT) | dead bit              4 B
    int32_t *kill_rec = nullptr;   // model record of the kill, -1 alive     4 B
    int2    *win      = nullptr;   // (z1stModelRec, zLstModelRec), only when windowed: one 8-byte word, so that the
                                   // re-sort gathers it with one request                          8 B
    int32_t *perm     = nullptr;   //                                        4 B
};

}  // namespace sitrk

struct sitrk_ctx {
    int device = 0;
    char err[512] = {0};

    hipStream_t own_stream = nullptr;   // created by the library
    hipStream_t stream = nullptr;       // compute stream in use (own or adopted)
    hipStream_t copy_stream = nullptr;  // host -> device record uploads (overlap with stepping)
    hipStream_t sv_stream = nullptr;    // Survive derivation of records that just arrived (round 4): ingest work -- the DMA on copy_stream,
                                        // then the record's Survive bytes here -- runs NEXT TO the stepping of the resident records;
                                        // the compute stream waits for a slot's event right before the first launch that reads the slot
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    // record ingest: library-owned pinned staging, double-buffered (sitrk_stage_acquire / sitrk_stage_submit).
    // The caller's buffers are copied (or read by the caller) into pinned memory before a push returns, the DMA into
    // the slot runs on copy_stream, ordered against the compute stream by events only.
    static constexpr int kStage = 2;
    void *stage[kStage] = {nullptr, nullptr};
    size_t stage_bytes = 0;                     // capacity of each buffer (= one whole slab)
    hipEvent_t stage_done[kStage] = {nullptr, nullptr};   // recorded on copy_stream behind the last DMA out of the buffer
    int stage_next = 0;                         // buffer the next acquire hands out
    int stage_rows = -1;                        // rows of the buffer handed out by acquire and not submitted yet (-1: none)
    int stage_cols = 0;                         // ... and its columns (Ni for the row-band entry points)

    // grid
    int Nj = 0, Ni = 0;
    sitrk::CellGeo *geo = nullptr;      // (Nj*Ni) 48-byte records
    sitrk::pt *geoF = nullptr;          // (Nj*Ni) F-points alone: the fused kernel fills its LDS patches from contiguous rows of it
    int8_t *orient = nullptr;           // (Nj*Ni) orientation bits of the velocity pick (cell_orient_kernel)
    int8_t *tmask = nullptr;

    // parameters
    double rdt = 3600.0;
    int uv_strategy = 1;
    double rmin_conc = 0.1;
    double eps_mg = 0.0;                // 2^-48 * max |Yf|,|Xf| (inside_quad_hot)
    // performance knobs and their measured defaults (tools/ab_tune.py on MI355X, C3):
    // non-temporal state streams -2 %, tile-major 8x16 cell order -8 %, XCD-chunked block order +5 % (off)
    int tune = sitrk::TUNE_NT_STATE;    // TUNE_* bits
    int step_block = 512;               // workgroup size of advect_step_kernel (512: -3.6 % vs 256, 1024: +1.8 %)
    int fuse = 32;                      // sitrk_run: consecutive resident records advanced per launch, <= nslots (1 = one launch per record)
    int tile_j = 8, tile_i = 16;        // sort order: 0 = row-major cells, else tile-major tiles of tile_j x tile_i cells
    int patch_kb = 16;                  // fused kernel: LDS bytes per workgroup for its geometry patch (0 = none, all reads global)
    int xcd_group = 16;                 // fused kernel: runs of that many consecutive workgroups on one XCD (0/1 = hardware order)
    int patch_margin = 8;               // ... and the widest margin of cells around the buoys' bounding box it may take
    int fill_threads = 8;               // host threads copying a pushed record (>= 8 MB) into the pinned staging (2 / 4 / 8: 26 / 36 / 47 GB/s on the box rows of C3)

    // records
    int nslots = 0, dtype = 0;
    size_t slab_bytes = 0;
    void *slabs = nullptr;              // nslots * [u|v|sic]
    uint8_t *kill9 = nullptr;           // nslots * (Nj*Ni): per cell the Survive bytes (tmask, sic, rmin_conc) of its 8 neighbours, one bit each
    unsigned char slot_dirty[4096] = {0};   // slab (re)written since its mask was derived
    // per slot: upload still in flight on copy_stream (the compute stream waits for slot_ready before it reads the
    // slot; events are created on first use), and the sequence number of the last launch that reads the slot (an upload
    // into it waits for that launch's event in the ring below, not for the whole compute stream)
    unsigned char slot_pending[4096] = {0};
    hipEvent_t slot_ready[4096] = {nullptr};
    // ... and a Survive derivation in flight on sv_stream: slot_sv[k] is recorded behind it (created on first use); the compute stream
    // waits for it before it reads the slot's bytes (slot_sv_pending), an upload into the slot before it overwrites the siconc it reads
    hipEvent_t slot_sv[4096] = {nullptr};
    unsigned char slot_sv_pending[4096] = {0};
    int async_survive = 0;              // knob: uploads derive their Survive bytes on sv_stream (1) or on the compute stream (0, default:
                                        // next to the fused loop, which needs its seven waves per SIMD, the co-running kernel costs more
                                        // than it hides -- profiles/r04r_*; behind a PCIe upload there is nothing to hide)
    long long slot_used_seq[4096];
    static constexpr int kLaunchRing = 64;
    hipEvent_t launch_ev[kLaunchRing] = {nullptr};
    long long launch_seq = 0;
    // rows [row_lo,row_hi) of the slot's u,v hold this record (whole record: 0..Nj); everything else is stale.
    // Survive bytes are valid for rows (row_lo, row_hi-1) and the domain rim.
    int slot_row_lo[4096] = {0}, slot_row_hi[4096] = {0};
    // the same for the columns (box ingest, round 4): columns [col_lo,col_hi) of those rows hold this record
    int slot_col_lo[4096] = {0}, slot_col_hi[4096] = {0};
    // host rows and columns of the live buoys at the last sitrk_buoy_rows() / sitrk_buoy_box(), and the records stepped
    // since (a host cell moves at most one row and one column per record): what a partly uploaded slot is checked against
    int band_jmin = 0, band_jmax = -1, band_age = -1;      // band_age < 0: not evaluated since sitrk_set_buoys
    int band_imin = 0, band_imax = -1;
    // asynchronous evaluation (sitrk_buoy_box_begin / _end): the reduction is queued on the compute stream, its result lands in
    // pinned host memory behind an event; records stepped after the begin are counted separately until the end adopts the result
    int *box_host = nullptr;            // 4 ints, pinned
    hipEvent_t box_ev = nullptr;
    bool box_pending = false;
    int box_pending_age = 0;
    // launch accounting (sitrk_launch_stats)
    long long n_fused_launches = 0, n_fused_records = 0, n_step_launches = 0;

    // buoys
    int64_t nP = 0;
    bool windowed = false;
    int32_t win_first_max = 0, win_last_min = 0;   // over all buoys: a launch whose records lie in [first_max, last_min] steps
                                                   // every live buoy at every record -> the form without the window test
    bool rim_buoys = false;             // some buoy was set in a cell with jT < 2 or iT < 2 (numpy negative-index wrap possible)
    sitrk::BuoyState st[2];             // double buffer for the sort
    int cur = 0;
    uint32_t *keys[2] = {nullptr, nullptr};
    int32_t *vals[2] = {nullptr, nullptr};
    void *sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    int resort_every = 512;             // re-sort cadence in steps (0 = never); measured best over 6000 steps at C3
    int steps_since_sort = 0;
    bool sorted_once = false;

    // diagnostic builds (make DIAG=1, knob "stamps"): per-wave s_memtime intervals of the fused loop's last launch
    unsigned long long *stamps = nullptr;
    size_t stamps_waves = 0;
    bool stamps_on = false;

    // scratch for fetch / locate
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    unsigned long long *counter = nullptr;   // device scalar for reductions
};
