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
    int32_t *first    = nullptr;   // z1stModelRec (only when windowed)      4 B
    int32_t *last     = nullptr;   // zLstModelRec                           4 B
    int32_t *perm     = nullptr;   //                                        4 B
};

}  // namespace sitrk

struct sitrk_ctx {
    int device = 0;
    char err[512] = {0};

    hipStream_t own_stream = nullptr;   // created by the library
    hipStream_t stream = nullptr;       // compute stream in use (own or adopted)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    // grid
    int Nj = 0, Ni = 0;
    sitrk::CellGeo *geo = nullptr;      // (Nj*Ni) 48-byte records
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

    // records
    int nslots = 0, dtype = 0;
    size_t slab_bytes = 0;
    void *slabs = nullptr;              // nslots * [u|v|sic]
    int8_t *kill = nullptr;             // nslots * (Nj*Ni) Survive masks derived from (tmask, sic, rmin_conc)
    unsigned char slot_dirty[4096] = {0};   // slab (re)written since its mask was derived

    // buoys
    int64_t nP = 0;
    bool windowed = false;
    sitrk::BuoyState st[2];             // double buffer for the sort
    int cur = 0;
    uint32_t *keys[2] = {nullptr, nullptr};
    int32_t *vals[2] = {nullptr, nullptr};
    void *sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    int resort_every = 512;             // re-sort cadence in steps (0 = never); measured best over 6000 steps at C3
    int steps_since_sort = 0;
    bool sorted_once = false;

    // scratch for fetch / locate
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    unsigned long long *counter = nullptr;   // device scalar for reductions
};
