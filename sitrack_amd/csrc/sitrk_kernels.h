// sitrk_kernels.h -- HIP kernels of libsitrk.so (gfx950, wave64).
//
// The hot path is advance_record(): everything one buoy does for one model record, one buoy per
// lane.  Buoys are kept sorted by host cell (tile-major), so a wavefront's gathers fall on a few
// contiguous 48-byte cell records (F,U,V plane coordinates interleaved) and short runs of the u/v
// slabs, which L1/L2 coalesce; nothing is a contraction (no MFMA).  Two launch forms:
//   advect_step_kernel : one record per launch  -- HBM bound (~84 B per particle-step, 0.62 of the 8 TB/s spec)
//   advect_run_kernel  : up to 32 resident records per launch; buoy and cell context in registers, the workgroup's
//                        F-points in LDS, crossing path driven by a table in LDS -- fp64 VALU-issue bound (86 % busy)
// Measurements and the optimisation history are in DESIGN.md section 3.2.
#pragma once
#include "sitrk_internal.h"

#pragma clang fp contract(off)

// The ablation hooks below give WRONG results by design (timing only, profiles/r03s_*): they exist in diagnostic builds
// (`make DIAG=1`) and nowhere else -- a stray -D must not ship a wrong tracker that still exports every symbol.
#if (defined(SITRK_ABL_NOEDGE) || defined(SITRK_ABL_NODIAG) || defined(SITRK_ABL_VEL2) || defined(SITRK_ABL_NOK9) || \
     defined(SITRK_ABL_VELCONST) || defined(SITRK_ABL_VELLDS) || defined(SITRK_NO_LAZY_HIT)) && !defined(SITRK_DIAG)
#error "SITRK_ABL_* / SITRK_NO_LAZY_HIT are diagnostic ablations (wrong results): they need -DSITRK_DIAG (make DIAG=1)"
#endif

namespace sitrk {

static constexpr int kBlock = 256;

// ---------------------------------------------------------------------------
// grid re-layout: six (Nj,Ni) fp64 arrays -> one CellGeo record per cell
// ---------------------------------------------------------------------------
__global__ void build_geo_kernel(size_t n, const double *__restrict__ Yf, const double *__restrict__ Xf,
                                 const double *__restrict__ Yu, const double *__restrict__ Xu,
                                 const double *__restrict__ Yv, const double *__restrict__ Xv,
                                 CellGeo *__restrict__ geo, pt *__restrict__ geoF)
{
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    CellGeo g;
    g.f = make_pt(Yf[k], Xf[k]);
    geoF[k] = g.f;
    g.u = make_pt(Yu[k], Xu[k]);
    g.v = make_pt(Yv[k], Xv[k]);
    geo[k] = g;
}

// The two orientation terms of the velocity pick (:427-441) that do not depend on the buoy,
//   ccw(F[j,i], V[j-1,i], V[j,i])  (bit 0)   and   ccw(F[j,i], U[j,i-1], U[j,i])  (bit 1),
// once per cell: the fused kernel reads one byte when a buoy enters a cell instead of evaluating them.
__global__ void cell_orient_kernel(int Nj, int Ni, const CellGeo *__restrict__ geo, int8_t *__restrict__ orient)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (size_t)Nj * Ni) return;
    const int j = (int)(k / Ni), i = (int)(k % Ni);
    int8_t o = 0;
    if (j >= 1 && i >= 1) {
        const CellGeo g = geo[k];
        o = (ccw(g.f, geo[k - Ni].v, g.v) ? 1 : 0) | (ccw(g.f, geo[k - 1].u, g.u) ? 2 : 0);
    }
    orient[k] = o;
}

// ---------------------------------------------------------------------------
// Survive   reference sitrack/tracking.py:62-93   (returns true = kill)
// ---------------------------------------------------------------------------
template <typename FT>
__device__ __forceinline__ bool survive_kill(int jT, int iT, int Nj, int Ni, const int8_t *__restrict__ tmask,
                                             const FT *__restrict__ sic, double rmin_conc)
{
    // too close to the domain boundaries (:73)
    if (jT <= 1 || jT >= Nj - 2 || iT <= 1 || iT >= Ni - 2) return true;
    size_t k = (size_t)jT * Ni + iT;
    // land-sea mask, 5 points; note [j-1,i-1] (:79)
    int zmt = (int)tmask[k] + (int)tmask[k + 1] + (int)tmask[k + Ni] + (int)tmask[k - 1] + (int)tmask[k - Ni - 1];
    if (zmt < 5) return true;
    // sea-ice concentration, same stencil, summed left to right in fp64 (:87-89)
    double zic = 0.2 * ((double)sic[k] + (double)sic[k + 1] + (double)sic[k + Ni] + (double)sic[k - 1] +
                        (double)sic[k - Ni - 1]);
    return zic < rmin_conc;
}

// ---------------------------------------------------------------------------
// Per-record Survive bytes, ONE pass over a record (or over an uploaded row band).  `Survive` (tracking.py:62-93)
// depends only on the cell and on the record's ice concentration, so it is evaluated ONCE PER CELL when a record
// becomes resident (same tests, same order, same left-to-right fp64 sum) and the crossing path reads one byte instead
// of chasing two dependent 5-point stencils through memory.  Outputs per cell:
//   kill9[j,i]  the Survive bytes of the cell's 8 neighbours, bit b = cell (j+dj, i+di) with (dj+1)*3 + (di+1) = b for
//               b < 4 and b + 1 otherwise (the centre is never a destination): both stepping kernels request this ONE byte
//               per buoy and record together with the velocities, at the top of a record's iteration, not behind the
//               crossing test -- it is the one crossing-path operand that is new with every record, i.e. never in cache.
//               (Round 4: the one-record kernel gathered three bytes of `kill` before; a record now has one Survive product:
//               6 bytes of traffic per cell instead of 7 and one array per slot instead of two.)
//   kill [j,i]  the Survive byte itself, only when asked for (the probe sitrk_survive_mask: `kill` may be null);
// A workgroup owns a tile of 32 x 64 cells: the tile's siconc and tmask with a halo of two cells go to LDS by row-wise
// coalesced loads, the Survive bytes of the tile and a halo of one cell are evaluated from LDS into LDS, and every
// thread then packs strips of four cells (one 32-bit store per output).  Rows: [j_lo, j_hi) are written, of which the
// siconc rows [v_lo, v_hi) are valid (row-band ingest; the whole record: 0, Nj, 0, Nj).  A Survive byte is derived
// when its 3-row stencil is valid or when its row belongs to the domain rim (killed whatever the ice is: slots are
// born with "kill" everywhere), else it is left alone; a kill9 byte whose three byte rows are not all derivable gets
// the sentinel "everything kills".
// (round 2 did this in two passes, survive_mask_kernel + pack_kill9_kernel: 58 + 52 us per 4096^2 record.)
// ---------------------------------------------------------------------------
// Round 4: the same for a BOX of the record (rows x columns: what the buoys of this GPU can touch, sitrk_buoy_box).  Columns
// follow the rows' rules: the kernel writes columns [c_lo, c_hi) (multiples of 4 on meshes with Ni % 4 == 0), of which the
// siconc columns [cv_lo, cv_hi) are valid; a Survive byte is derived when its 3 x 3 stencil is valid in BOTH directions or
// the cell belongs to the rim, a kill9 byte whose 3 x 3 byte neighbourhood is not all derivable gets the sentinel.
struct SvBox {
    int j_lo, j_hi, v_lo, v_hi;         // rows written / rows of siconc that are valid
    int c_lo, c_hi, cv_lo, cv_hi;       // columns written / columns of siconc that are valid
};
// Several records in ONE launch (blockIdx.z): a fused launch of advect_run_kernel steps with up to 32 resident records, and their
// Survive bytes are derived the same way -- one dispatch for the batch instead of one per record (a box of a 4096^2 record is 10 us
// of memory traffic, of the order of a dependent dispatch itself).
static constexpr int kSvMaxBatch = 32;
struct SvBatch {
    long long sic_stride, kill_stride;  // elements between two slots' siconc fields / bytes between their Survive arrays
    int slot[kSvMaxBatch];
};
// Survive byte of column i derivable from the valid columns (rim columns always are: killed whatever the ice is)
__device__ __forceinline__ bool sv_col_ok(int i, int Ni, int cv_lo, int cv_hi)
{
    return (i <= 1) | (i >= Ni - 2) | ((i - 1 >= cv_lo) & (i + 1 < cv_hi));
}

static constexpr int kSvTR = 32, kSvTC = 64;                   // cells per workgroup
static constexpr int kSvSC = kSvTC + 4, kSvSR = kSvTR + 4;     // siconc / tmask tile: halo of 2 (two nested 3-row stencils)
static constexpr int kSvKC = kSvTC + 2, kSvKR = kSvTR + 2;     // Survive bytes: halo of 1
static constexpr int kSvKP = 72;                               // their row pitch in LDS: 8-byte aligned rows of 66 (+2 read past)
static constexpr int kSvBlock = 256;

template <typename FT>
__global__ __launch_bounds__(kSvBlock) void survive_kill9_kernel(int Nj, int Ni, SvBox bx, SvBatch sb,
                                                                const int8_t *__restrict__ tmask, const FT *__restrict__ sic0,
                                                                double rmin_conc, int8_t *__restrict__ kill0, uint8_t *__restrict__ kill90)
{
    const FT *__restrict__ sic = sic0 + (size_t)sb.slot[blockIdx.z] * (size_t)sb.sic_stride;
    int8_t *__restrict__ kill = kill0 ? kill0 + (size_t)sb.slot[blockIdx.z] * (size_t)sb.kill_stride : nullptr;    // (probes only)
    uint8_t *__restrict__ kill9 = kill90 + (size_t)sb.slot[blockIdx.z] * (size_t)sb.kill_stride;
    __shared__ FT s_sic[kSvSR * kSvSC];
    __shared__ int8_t s_tm[kSvSR * kSvSC];
    __shared__ __attribute__((aligned(8))) uint8_t s_k[kSvKR * kSvKP];
    const int j_lo = bx.j_lo, j_hi = bx.j_hi, v_lo = bx.v_lo, v_hi = bx.v_hi;
    const int jt0 = j_lo + (int)blockIdx.y * kSvTR, it0 = bx.c_lo + (int)blockIdx.x * kSvTC;
    (void)j_lo;
    // ---- the tile's inputs, halo of 2, row-wise coalesced; outside the mesh: anything (such cells are rim or beyond)
    for (int t = threadIdx.x; t < kSvSR * kSvSC; t += kSvBlock) {
        const int r = t / kSvSC, c = t - r * kSvSC;
        const int j = jt0 - 2 + r, i = it0 - 2 + c;
        const bool in = (j >= 0) & (j < Nj) & (i >= 0) & (i < Ni);
        const size_t k = (size_t)(in ? j : 0) * Ni + (in ? i : 0);
        s_sic[t] = sic[k];
        s_tm[t] = tmask[k];
    }
    __syncthreads();
    // ---- Survive (tracking.py:62-93) for the tile and a halo of 1; true = kill
    for (int t = threadIdx.x; t < kSvKR * kSvKC; t += kSvBlock) {
        const int r = t / kSvKC, c = t - r * kSvKC;
        const int j = jt0 - 1 + r, i = it0 - 1 + c;
        bool kl = true;                                                              // too close to the domain boundaries (:73)
        if (!(j <= 1 || j >= Nj - 2 || i <= 1 || i >= Ni - 2)) {
            const int s = (r + 1) * kSvSC + (c + 1);
            // land-sea mask, 5 points; note [j-1,i-1] (:79)
            const int zmt = (int)s_tm[s] + (int)s_tm[s + 1] + (int)s_tm[s + kSvSC] + (int)s_tm[s - 1] + (int)s_tm[s - kSvSC - 1];
            kl = zmt < 5;
            if (!kl) {
                // sea-ice concentration, same stencil, summed left to right in fp64 (:87-89)
                const double zic = 0.2 * ((double)s_sic[s] + (double)s_sic[s + 1] + (double)s_sic[s + kSvSC] + (double)s_sic[s - 1] +
                                          (double)s_sic[s - kSvSC - 1]);
                kl = zic < rmin_conc;
            }
        }
        s_k[r * kSvKP + c] = kl ? 1 : 0;
    }
    __syncthreads();
    // ---- strips of four cells: their Survive bytes and the packed neighbourhoods
    const bool vec = (Ni & 3) == 0;                                                  // rows are 4-byte aligned: one store per strip
    for (int g = threadIdx.x; g < kSvTR * (kSvTC / 4); g += kSvBlock) {
        const int r = g / (kSvTC / 4), c4 = (g - r * (kSvTC / 4)) * 4;
        const int j = jt0 + r, i0 = it0 + c4;
        if (j >= j_hi || i0 >= Ni || i0 >= bx.c_hi) continue;
        // byte rows j-1, j, j+1, columns i0-1 .. i0+4 (bytes 0..5 of each word; 6, 7 unused)
        // (two 4-byte aligned words each)
        const uint32_t *p0 = (const uint32_t *)&s_k[r * kSvKP + c4], *p1 = p0 + kSvKP / 4, *p2 = p1 + kSvKP / 4;
        const uint64_t R0 = p0[0] | ((uint64_t)p0[1] << 32), R1 = p1[0] | ((uint64_t)p1[1] << 32), R2 = p2[0] | ((uint64_t)p2[1] << 32);
        // which byte rows can be derived from the rows that were uploaded (the rim rows always can)
        bool ok3 = (j >= 1) & (j <= Nj - 2);
#pragma unroll
        for (int dj = -1; dj <= 1; dj++) {
            const int rr = j + dj;
            ok3 = ok3 & ((rr <= 1) | (rr >= Nj - 2) | ((rr - 1 >= v_lo) & (rr + 1 < v_hi)));
        }
        const bool wr_kill = (j <= 1) | (j >= Nj - 2) | ((j - 1 >= v_lo) & (j + 1 < v_hi));
        // the same in the columns: bit q + 1 of dv6 = the Survive byte of column i0 + q (q = -1..4) can be derived
        unsigned dv6 = 0;
#pragma unroll
        for (int q = -1; q <= 4; q++) dv6 |= (sv_col_ok(i0 + q, Ni, bx.cv_lo, bx.cv_hi) ? 1u : 0u) << (q + 1);
        const unsigned okc = dv6 & (dv6 >> 1) & (dv6 >> 2);                           // columns q-1, q, q+1 all derivable
        const unsigned wrm = wr_kill ? ((dv6 >> 1) & 0xfu) : 0u;
        unsigned w9 = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned a = (unsigned)(R0 >> (8 * q)) & 0x010101u, b = (unsigned)(R1 >> (8 * q)) & 0x010001u,
                           c = (unsigned)(R2 >> (8 * q)) & 0x010101u;
            unsigned w = (a & 1u) | ((a >> 7) & 2u) | ((a >> 14) & 4u) | ((b & 1u) << 3) | ((b >> 12) & 16u) |
                         ((c & 1u) << 5) | ((c >> 2) & 64u) | ((c >> 9) & 128u);
            const int i = i0 + q;
            if (!(ok3 && ((okc >> q) & 1u) && i >= 1 && i <= Ni - 2)) w = 0xffu;
            w9 |= w << (8 * q);
        }
        const unsigned wk = (unsigned)(R1 >> 8);                                      // the strip's own four Survive bytes
        const size_t k = (size_t)j * Ni + i0;
        if (vec && i0 + 4 <= bx.c_hi) {
            if (kill) {
                if (wrm == 0xfu) *(unsigned *)(kill + k) = wk;
                else
                    for (int q = 0; q < 4; q++)
                        if ((wrm >> q) & 1u) kill[k + q] = (int8_t)((wk >> (8 * q)) & 0xffu);
            }
            *(unsigned *)(kill9 + k) = w9;
        } else {
            for (int q = 0; q < 4 && i0 + q < Ni && i0 + q < bx.c_hi; q++) {
                if (kill && ((wrm >> q) & 1u)) kill[k + q] = (int8_t)((wk >> (8 * q)) & 0xffu);
                kill9[k + q] = (uint8_t)((w9 >> (8 * q)) & 0xffu);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// The same two outputs WITHOUT LDS and without barriers, for meshes whose rows are 16-byte aligned (Ni % 4 == 0: every
// configuration of BASELINE.json): a wavefront walks DOWN a strip of the mesh, one mesh row per iteration, each lane owning
// four adjacent columns (one 16-byte load of siconc and one 4-byte load of tmask per lane and row; 16 B loads, 4 B stores).
// A Survive byte needs the rows j-1, j, j+1 of both inputs: they are the last three rows the wave loaded (registers); the
// columns i-1 and i+1 of a lane's edge cells are the neighbouring lanes' edge values (two shuffles per input and row).
// The packed neighbourhoods of row j need the Survive bytes of rows j-1..j+1: the last three rows the wave derived.  So row
// r is loaded, row r-1's Survive bytes are derived, row r-2's outputs are stored.  Lanes 0 and 63 are halo (a strip yields
// 62 x 4 = 248 columns per row), a chunk of kSvRowsR rows costs 4 extra rows.  Loads run kSvRowsD rows ahead (unconditional,
// from clamped addresses).  Per row and wave: 38 fp64-class + 105 32-bit vector instructions for 248 cells, against ~1.5
// wave-instructions per cell of the LDS-tile kernel above, which stays the general form (any Ni).  35 us per 4096^2 record
// (58 us the LDS-tile kernel, 110 us round 2's two passes): 117 MB -> 3.3 TB/s.
// Same tests, same order, same left-to-right fp64 sum; same validity rules for row bands.
// ---------------------------------------------------------------------------
#ifndef SITRK_SV_R
#define SITRK_SV_R 16                   // measured at 4096^2 (tools/sv_ab.sh, profiles/r03m_*): 8 / 12 / 16 / 32 / 64 rows per wave = 34.7 / 35.4 /
#endif                                  // 35.3 / 39.3 / 57.1 us -- more waves beat fewer halo rows; 16 keeps the halo at 25 %
#ifndef SITRK_SV_D
#define SITRK_SV_D 4
#endif
static constexpr int kSvRowsR = SITRK_SV_R;  // output rows per wave
static constexpr int kSvRowsD = SITRK_SV_D;  // rows in flight per wave; (kSvRowsR + 4) % kSvRowsD == 0
static_assert((kSvRowsR + 4) % kSvRowsD == 0, "the row loop is unrolled by the prefetch depth");
static constexpr int kSvRowsCols = 62 * 4;   // output columns per wave and row

template <typename FT> struct Row4 { FT x, y, z, w; };

template <typename FT>
__device__ __forceinline__ Row4<FT> sv_load4(const FT *__restrict__ p)
{
    Row4<FT> r;
    if (sizeof(FT) == 4) {
        const float4 t = *(const float4 *)p;
        r.x = (FT)t.x; r.y = (FT)t.y; r.z = (FT)t.z; r.w = (FT)t.w;
    } else {
        const double2 a = *(const double2 *)p, b = *(const double2 *)(p + 2);
        r.x = (FT)a.x; r.y = (FT)a.y; r.z = (FT)b.x; r.w = (FT)b.y;
    }
    return r;
}

// lane i <- lane i-1 / lane i+1 across the whole wavefront as ONE vector instruction each (DPP wave_shr:1 / wave_shl:1, gfx9
// family) instead of a round trip through the LDS crossbar (ds_bpermute + lgkmcnt wait): three such hops sit on every row's
// dependent chain.  Lane 0 (63) keeps its own value: the strip's halo lanes never use what they would receive.
__device__ __forceinline__ int sv_up1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int sv_dn1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ float sv_shfl_up(float v) { return __int_as_float(sv_up1(__float_as_int(v))); }
__device__ __forceinline__ float sv_shfl_dn(float v) { return __int_as_float(sv_dn1(__float_as_int(v))); }
__device__ __forceinline__ double sv_shfl_up(double v)
{
    return __hiloint2double(sv_up1(__double2hiint(v)), sv_up1(__double2loint(v)));
}
__device__ __forceinline__ double sv_shfl_dn(double v)
{
    return __hiloint2double(sv_dn1(__double2hiint(v)), sv_dn1(__double2loint(v)));
}

template <typename FT>
__global__ __launch_bounds__(256) void survive_kill9_rows_kernel(int Nj, int Ni, SvBox bx, SvBatch sb,
                                                                const int8_t *__restrict__ tmask, const FT *__restrict__ sic0,
                                                                double rmin_conc, int8_t *__restrict__ kill0, uint8_t *__restrict__ kill90)
{
    const FT *__restrict__ sic = sic0 + (size_t)sb.slot[blockIdx.z] * (size_t)sb.sic_stride;
    int8_t *__restrict__ kill = kill0 ? kill0 + (size_t)sb.slot[blockIdx.z] * (size_t)sb.kill_stride : nullptr;    // (probes only)
    uint8_t *__restrict__ kill9 = kill90 + (size_t)sb.slot[blockIdx.z] * (size_t)sb.kill_stride;
    static_assert(sizeof(FT) == 4 || sizeof(FT) == 8, "records are binary32 or binary64");
#if !defined(__gfx9__) && !defined(__GFX9__) && defined(__HIP_DEVICE_COMPILE__)
#error "survive_kill9_rows_kernel uses the wave64 DPP row shifts of the gfx9 family (wave_shr:1 / wave_shl:1): build for gfx950"
#endif
    const int j_lo = bx.j_lo, j_hi = bx.j_hi, v_lo = bx.v_lo, v_hi = bx.v_hi;
    const int lane = (int)(threadIdx.x & 63u), wv = (int)(threadIdx.x >> 6);
    const int jr0 = j_lo + ((int)blockIdx.y * 4 + wv) * kSvRowsR;            // first output row of this wave
    if (jr0 >= j_hi) return;                                                  // (wave-uniform: no barrier in this kernel)
    const int g0 = bx.c_lo + (int)blockIdx.x * kSvRowsCols + 4 * (lane - 1); // first of the lane's four columns (lane 0: halo left); c_lo % 4 == 0
    const bool in_cols = g0 >= 0 && g0 < Ni;                                  // Ni % 4 == 0: a group is inside the mesh or outside, whole
    const bool out_lane = lane >= 1 && lane <= 62 && in_cols && g0 < bx.c_hi;
    // columns whose Survive byte can be derived from the valid columns: bit q + 1 = column g0 + q, q = -1..4
    unsigned dv6 = 0;
#pragma unroll
    for (int q = -1; q <= 4; q++) dv6 |= (sv_col_ok(g0 + q, Ni, bx.cv_lo, bx.cv_hi) ? 1u : 0u) << (q + 1);
    const unsigned okc = dv6 & (dv6 >> 1) & (dv6 >> 2);                       // packed byte of column q: columns q-1, q, q+1 all derivable
    const unsigned dvc = (dv6 >> 1) & 0xfu;
    // columns of the domain rim (iT <= 1 or iT >= Ni-2, tracking.py:73), one bit per column of the group
    unsigned rimc = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) rimc |= ((g0 + q <= 1 || g0 + q >= Ni - 2) ? 1u : 0u) << q;
    // interior columns for the packed byte (1 <= i <= Ni-2)
    unsigned intc = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) intc |= ((g0 + q >= 1 && g0 + q <= Ni - 2) ? 1u : 0u) << q;

    const int r_first = jr0 - 2, r_last = min(jr0 + kSvRowsR, j_hi) + 1;     // rows to load: two above the first output row .. two below the last
    Row4<FT> ring_s[kSvRowsD];
    unsigned ring_t[kSvRowsD];
    // (always a load, from a clamped address: a load inside a branch makes the compiler wait for ALL outstanding loads at every
    // use -- `s_waitcnt vmcnt(0)` -- i.e. one row in flight instead of kSvRowsD; rows and columns outside the mesh read row 0 /
    // column group 0 and are never used: such cells are rim or beyond)
    const int gload = in_cols ? g0 : 0;
    auto fetch = [&](int r, Row4<FT> &sv, unsigned &tv) {
        const int rr = min(max(r, 0), min(Nj - 1, r_last));
        const size_t k = (size_t)rr * Ni + gload;
        sv = sv_load4<FT>(sic + k);
        tv = *(const unsigned *)(tmask + k);
    };
#pragma unroll
    for (int d = 0; d < kSvRowsD; d++) fetch(r_first + d, ring_s[d], ring_t[d]);

    // rolling state: inputs of rows r-1 (S1, T1 with their edge neighbours) and r-2 (shifted: column q-1), Survive nibbles of rows r-3, r-2
    Row4<FT> S1 = {(FT)0, (FT)0, (FT)0, (FT)0};
    FT S1up = (FT)0, S1dn = (FT)0;
    FT S0m1 = (FT)0, S0x = (FT)0, S0y = (FT)0, S0z = (FT)0;                  // row r-2 at columns q-1 for q = 0..3
    unsigned T1 = 0, T1up = 0, T1dn = 0, T0sh = 0;                           // T0sh: row r-2's bytes at columns q-1 (4 bytes)
    unsigned K0 = 0, K1 = 0;                                                 // 6-bit windows (columns -1..4) of the Survive bits of rows r-3, r-2

    for (int it = 0; it < kSvRowsR + 4; it += kSvRowsD) {
#pragma unroll
        for (int d = 0; d < kSvRowsD; d++) {
            const int r = r_first + it + d;                                   // the row that arrives now
            const Row4<FT> S2 = ring_s[d];
            const unsigned T2 = ring_t[d];
            fetch(r + kSvRowsD, ring_s[d], ring_t[d]);                        // keep kSvRowsD rows in flight
            // ---- Survive bytes of row jm = r-1 (tracking.py:62-93), four columns
            const int jm = r - 1;
            const bool rimrow = (jm <= 1) | (jm >= Nj - 2);
            unsigned kn = 0;                                                  // nibble: bit q = kill(jm, g0+q)
            {
                const FT c1[6] = {S1up, S1.x, S1.y, S1.z, S1.w, S1dn};        // row jm, columns -1..4
                const FT c0[4] = {S0m1, S0x, S0y, S0z};                       // row jm-1, columns q-1
                const FT c2[4] = {S2.x, S2.y, S2.z, S2.w};                    // row jm+1, columns q
                const unsigned long long w1 = ((unsigned long long)(T1up >> 24)) | ((unsigned long long)T1 << 8) | ((unsigned long long)(T1dn & 0xffu) << 40);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    // land-sea mask, 5 points; note [j-1,i-1] (:79): m[j,i] + m[j,i+1] + m[j+1,i] + m[j,i-1] + m[j-1,i-1]
                    const int zmt = (int)(int8_t)(w1 >> (8 * (q + 1))) + (int)(int8_t)(w1 >> (8 * (q + 2))) + (int)(int8_t)(T2 >> (8 * q)) +
                                    (int)(int8_t)(w1 >> (8 * q)) + (int)(int8_t)(T0sh >> (8 * q));
                    // sea-ice concentration, same stencil, summed left to right in fp64 (:87-89)
                    const double zic = 0.2 * ((double)c1[q + 1] + (double)c1[q + 2] + (double)c2[q] + (double)c1[q] + (double)c0[q]);
                    const bool kl = rimrow | (((rimc >> q) & 1u) != 0) | (zmt < 5) | (zic < rmin_conc);
                    kn |= (kl ? 1u : 0u) << q;
                }
            }
            // the row's 6-bit window: neighbours' edge bits on both sides
            const unsigned kup = (unsigned)sv_up1((int)kn), kdn = (unsigned)sv_dn1((int)kn);
            const unsigned K2 = ((kup >> 3) & 1u) | (kn << 1) | ((kdn & 1u) << 5);
            // ---- outputs of row jo = r-2
            const int jo = r - 2;
            if (jo >= jr0 && jo < j_hi && jo < jr0 + kSvRowsR && out_lane) {   // (row conditions are wave-uniform)
                bool ok3 = (jo >= 1) & (jo <= Nj - 2);
#pragma unroll
                for (int dj = -1; dj <= 1; dj++) {
                    const int rr = jo + dj;
                    ok3 = ok3 & ((rr <= 1) | (rr >= Nj - 2) | ((rr - 1 >= v_lo) & (rr + 1 < v_hi)));
                }
                const bool wr_kill = (jo <= 1) | (jo >= Nj - 2) | ((jo - 1 >= v_lo) & (jo + 1 < v_hi));
                unsigned w9 = 0, wk = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    unsigned w = ((K0 >> q) & 7u) | (((K1 >> q) & 1u) << 3) | (((K1 >> (q + 2)) & 1u) << 4) | (((K2 >> q) & 7u) << 5);
                    if (!(ok3 && (((intc & okc) >> q) & 1u))) w = 0xffu;
                    w9 |= w << (8 * q);
                    wk |= ((K1 >> (q + 1)) & 1u) << (8 * q);
                }
                const size_t k = (size_t)jo * Ni + g0;
                if (wr_kill && kill) {
                    if (dvc == 0xfu) *(unsigned *)(kill + k) = wk;
                    else                                                              // (a box's edge groups only)
                        for (int q = 0; q < 4; q++)
                            if ((dvc >> q) & 1u) kill[k + q] = (int8_t)((wk >> (8 * q)) & 0xffu);
                }
                *(unsigned *)(kill9 + k) = w9;
            }
            // ---- roll: row r becomes row r-1
            S0m1 = S1up; S0x = S1.x; S0y = S1.y; S0z = S1.z;
            T0sh = (T1 << 8) | (T1up >> 24);
            S1 = S2; T1 = T2;
            S1up = sv_shfl_up(S2.w); S1dn = sv_shfl_dn(S2.x);
            T1up = (unsigned)sv_up1((int)T2); T1dn = (unsigned)sv_dn1((int)T2);
            K0 = K1; K1 = K2;
        }
    }
}

// rows and columns of the host cells of the buoys that are still alive, as FOUR MAXIMA: out = {max jT, max -jT, max iT, max -iT}
// (one identity for all four: the host fills `out` with a very negative pattern by one memset).  16-byte loads; a workgroup
// touches the result only where it improves on what is there (same-address atomics serialise at the memory side: 2048 workgroups
// x 4 atomics cost 100 us at 1e7 buoys, the reads 15 us; a stale read can only make a workgroup try an atomic it did not need).
__global__ __launch_bounds__(kBlock) void buoy_box_kernel(int64_t n, const int32_t *__restrict__ cell, int *out)
{
    __shared__ int sm[4][kBlock / 64];
    int m0 = INT32_MIN, m1 = INT32_MIN, m2 = INT32_MIN, m3 = INT32_MIN;
    auto take = [&](int32_t c) {
        if (c >= 0) {
            const int j = cell_j(c), i = cell_i(c);
            m0 = max(m0, j); m1 = max(m1, -j); m2 = max(m2, i); m3 = max(m3, -i);
        }
    };
    const int64_t n4 = n >> 2;
    for (int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x; s < n4; s += (int64_t)gridDim.x * kBlock) {
        const int4 c = ((const int4 *)cell)[s];                     // (hipMalloc'ed: 256-byte aligned)
        take(c.x); take(c.y); take(c.z); take(c.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n & 3)) take(cell[4 * n4 + threadIdx.x]);
    for (int off = 32; off > 0; off >>= 1) {
        m0 = max(m0, __shfl_down(m0, off)); m1 = max(m1, __shfl_down(m1, off));
        m2 = max(m2, __shfl_down(m2, off)); m3 = max(m3, __shfl_down(m3, off));
    }
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; sm[0][w] = m0; sm[1][w] = m1; sm[2][w] = m2; sm[3][w] = m3; }
    __syncthreads();
    if (threadIdx.x < 4) {
        int m = sm[threadIdx.x][0];
        for (int w = 1; w < kBlock / 64; w++) m = max(m, sm[threadIdx.x][w]);
        if (m > __atomic_load_n(&out[threadIdx.x], __ATOMIC_RELAXED)) atomicMax(&out[threadIdx.x], m);
    }
}

__device__ __forceinline__ pt load_f(const CellGeo *__restrict__ geo, int j, int i, int Nj, int Ni)
{
    return geo[(unsigned)(pywrap(j, Nj) * Ni + pywrap(i, Ni))].f;          // Nj*Ni < 2^31 (sitrk_set_grid)
}

// CrossedEdge + NewHostCell + UpdtInd4NewCell + Survive   reference sitrack/tracking.py:62-93,182-305
// quad = [bl, br, ur, ul] of the current cell (jT,iT).  Returns the new packed cell (dead bit set when
// Survive kills).  CrossedEdge needs only the quad (registers); what must come from memory afterwards --
// the two grid-line extension points of NewHostCell -- is loaded in one batch of independent loads; the
// Survive bytes of the three cells the buoy can have entered are bits of k9, the host cell's packed
// neighbourhood byte of this record (loaded with the velocities: round 4, three byte gathers before).
__device__ __forceinline__ int32_t resolve_crossing(pt P1, pt P2, pt bl, pt br, pt ur, pt ul, int jT, int iT, int Nj, int Ni,
                                                    const CellGeo *__restrict__ geo, unsigned k9,
                                                    bool &killed, int *codes = nullptr)
{
    // CrossedEdge (:189-200): first of bottom, right, upper, left hit; falls through to 4.
    // intersect2Seg(P1,P2,C,D) = (ccw(P1,C,D) != ccw(P2,C,D)) and (ccw(P1,P2,C) != ccw(P1,P2,D)); the second
    // pair only involves one quad vertex each, so the four vertex terms are shared by adjacent edges.
    const bool sbl = ccw(P1, P2, bl), sbr = ccw(P1, P2, br), sur = ccw(P1, P2, ur), sul = ccw(P1, P2, ul);
    const bool h1 = (ccw(P1, bl, br) != ccw(P2, bl, br)) && (sbl != sbr);
    const bool h2 = (ccw(P1, br, ur) != ccw(P2, br, ur)) && (sbr != sur);
    const bool h3 = (ccw(P1, ur, ul) != ccw(P2, ur, ul)) && (sur != sul);
    const int kc = h1 ? 1 : (h2 ? 2 : (h3 ? 3 : 4));
    // NewHostCell (:217-243): segment vs the grid line prolonging the crossed edge beyond its two ends.
    //   kc  first test (vertex -> ext. point) => code      second test                       => code   straight
    //   1   bl -> F[jT-2,iT-1]  => 5 (-1,-1)                br -> F[jT-2,iT  ]  => 6 (-1,+1)           (-1, 0)
    //   2   br -> F[jT-1,iT+1]  => 6 (-1,+1)                ur -> F[jT  ,iT+1]  => 7 (+1,+1)           ( 0,+1)
    //   3   ul -> F[jT+1,iT-1]  => 8 (+1,-1)                ur -> F[jT+1,iT  ]  => 7 (+1,+1)           (+1, 0)
    //   4   ul -> F[jT  ,iT-2]  => 8 (+1,-1)                bl -> F[jT-1,iT-2]  => 5 (-1,-1)           ( 0,-1)
    const pt va = (kc == 1) ? bl : (kc == 2) ? br : ul;
    const pt vb = (kc == 1) ? br : (kc == 4) ? bl : ur;
    // the ten index offsets of the table above come out of 2-bit lookup words (entry kc-1 holds offset+2): one
    // bit-field extract each instead of a chain of selects -- the crossing path runs in every wave on every record
    // with a few lanes active, so its instruction count is paid in full
#define SITRK_TAB4(e1, e2, e3, e4) ((unsigned)(((e1) + 2) | (((e2) + 2) << 2) | (((e3) + 2) << 4) | (((e4) + 2) << 6)))
#define SITRK_LOOKUP(word) ((int)(((word) >> sh) & 3u) - 2)
    const unsigned sh = 2u * (unsigned)(kc - 1);
    const int jA = jT + SITRK_LOOKUP(SITRK_TAB4(-2, -1, 1, 0)), iA = iT + SITRK_LOOKUP(SITRK_TAB4(-1, 1, -1, -2));
    const int jB = jT + SITRK_LOOKUP(SITRK_TAB4(-2, 0, 1, -1)), iB = iT + SITRK_LOOKUP(SITRK_TAB4(0, 1, 0, -2));
    const int djS = SITRK_LOOKUP(SITRK_TAB4(-1, 0, 1, 0)), diS = SITRK_LOOKUP(SITRK_TAB4(0, 1, 0, -1));
    const int djA = SITRK_LOOKUP(SITRK_TAB4(-1, -1, 1, 1)), diA = SITRK_LOOKUP(SITRK_TAB4(-1, 1, -1, -1));      // codes 5,6,8,8
    const int djB = SITRK_LOOKUP(SITRK_TAB4(-1, 1, 1, -1)), diB = SITRK_LOOKUP(SITRK_TAB4(1, 1, 1, -1));        // codes 6,7,7,5
#undef SITRK_LOOKUP
#undef SITRK_TAB4
    // one batch of independent loads (only the extension points can index below zero, and then numpy wraps)
    const pt eA = load_f(geo, jA, iA, Nj, Ni);
    const pt eB = load_f(geo, jB, iB, Nj, Ni);
    // the Survive bytes of the three cells the buoy can have entered: bits of the host cell's packed neighbourhood (kill9),
    // bit b = cell (jT+dj, iT+di) with (dj+1)*3 + (di+1) = b for b < 4 and b + 1 otherwise
    auto nb_bit = [](int dj, int di) { const int b9 = (dj + 1) * 3 + (di + 1); return 1u << (b9 < 4 ? b9 : b9 - 1); };
    const bool kS = (k9 & nb_bit(djS, diS)) != 0, kA = (k9 & nb_bit(djA, diA)) != 0, kB = (k9 & nb_bit(djB, diB)) != 0;
    const bool sva = (kc == 1) ? sbl : (kc == 2) ? sbr : sul;               // ccw(P1,P2,va), already known
    const bool svb = (kc == 1) ? sbr : (kc == 4) ? sbl : sur;
    const bool hitA = (ccw(P1, va, eA) != ccw(P2, va, eA)) && (sva != ccw(P1, P2, eA));
    const bool hitB = (ccw(P1, vb, eB) != ccw(P2, vb, eB)) && (svb != ccw(P1, P2, eB));
    // UpdtInd4NewCell (:257-300); first match wins (if / elif)
    const int dj = hitA ? djA : (hitB ? djB : djS);
    const int di = hitA ? diA : (hitB ? diB : diS);
    killed = hitA ? kA : (hitB ? kB : kS);                                  // Survive (:483-484)
    if (codes) {                                                            // probes only: CrossedEdge / NewHostCell return values
        const int cA = (kc == 1) ? 5 : (kc == 2) ? 6 : 8, cB = (kc == 1) ? 6 : (kc == 4) ? 5 : 7;
        codes[0] = kc;
        codes[1] = hitA ? cA : (hitB ? cB : kc);
    }
    int32_t cn = pack_cell(jT + dj, iT + di);
    if (killed) cn |= SITRK_DEAD_BIT;
    return cn;
}


// ---------------------------------------------------------------------------
// The same chain for the fused kernel, driven by a 4-row table in LDS (one row per crossed edge) instead of chains of
// selects and index arithmetic: the crossing path runs in every wave on every record with ~9 of 64 lanes active, so
// every instruction in it is paid in full.  Everything memory-side is an offset from the host cell's own geometry
// record (the F-point is its first member) or from its cell index; cells with jT < 2 or iT < 2, where numpy's negative
// index wraps (tracking.py:219), never reach this function (sitrk_run steps such buoy sets record by record).
//   row kc-1:  [0] va  [1] vb  [2] eA  [3] eB   byte offsets into `geo` relative to the host cell's record
//              [4] S   [5] A   [6] B            packed-cell increments (dj << 16) + di   straight / first / second diagonal
//              [7] S   [8] A   [9] B            cell-index increments dj*Ni + di
//              [10] S  [11] A  [12] B           mask (1 << bit) of the destination in the neighbours' Survive byte (pack_kill9_kernel)
// ---------------------------------------------------------------------------
struct CrossTab { int v[4][16]; };

__host__ inline void make_cross_tab(int Ni, CrossTab &t, int (*dji)[7][2] = nullptr)
{
    // kc: 1 bottom, 2 right, 3 upper, 4 left.  (dj,di) of va, vb (the crossed edge's ends as F-points relative to F[jT,iT]),
    // of the two extension points, and of the straight / A / B destination cells -- the table of resolve_crossing above
    static const int va[4][2] = {{-1, -1}, {-1, 0}, {0, -1}, {0, -1}}, vb[4][2] = {{-1, 0}, {0, 0}, {0, 0}, {-1, -1}};
    static const int eA[4][2] = {{-2, -1}, {-1, 1}, {1, -1}, {0, -2}}, eB[4][2] = {{-2, 0}, {0, 1}, {1, 0}, {-1, -2}};
    static const int dS[4][2] = {{-1, 0}, {0, 1}, {1, 0}, {0, -1}}, dA[4][2] = {{-1, -1}, {-1, 1}, {1, -1}, {1, -1}},
                     dB[4][2] = {{-1, 1}, {1, 1}, {1, 1}, {-1, -1}};
    for (int e = 0; e < 4; e++) {
        int *r = t.v[e];
        for (int q = 0; q < 16; q++) r[q] = 0;
        r[0] = (va[e][0] * Ni + va[e][1]) * (int)sizeof(CellGeo);
        r[1] = (vb[e][0] * Ni + vb[e][1]) * (int)sizeof(CellGeo);
        r[2] = (eA[e][0] * Ni + eA[e][1]) * (int)sizeof(CellGeo);
        r[3] = (eB[e][0] * Ni + eB[e][1]) * (int)sizeof(CellGeo);
        if (dji) {
            const int(*all[7])[2] = {va, vb, eA, eB, dS, dA, dB};
            for (int q = 0; q < 7; q++) { dji[e][q][0] = all[q][e][0]; dji[e][q][1] = all[q][e][1]; }
        }
        const int(*d[3])[2] = {dS, dA, dB};
        for (int q = 0; q < 3; q++) {
            const int dj = d[q][e][0], di = d[q][e][1];
            r[4 + q] = dj * 65536 + di;
            r[7 + q] = dj * Ni + di;
            const int b9 = (dj + 1) * 3 + (di + 1);
            r[10 + q] = 1 << (b9 < 4 ? b9 : b9 - 1);
        }
    }
}

// a point of the geometry at (scalar base + 32-bit byte offset + immediate): no 64-bit address arithmetic per lane
__device__ __forceinline__ pt geo_pt(const char *__restrict__ gb, unsigned off, int imm = 0)
{
    return *(const pt *)(gb + (size_t)off + (ptrdiff_t)imm);
}

// P1 -> P2 leaves the cell whose quad is (bl, br, ur, ul); k48 = byte offset of the cell's geometry record, k9 = the
// Survive bits of the cell's 8 neighbours for this record (pack_kill9_kernel).  Returns the crossed edge kc (1..4) and
// fills the increments of the destination cell; `killed` = the destination's Survive byte.
// Same predicates on the same operands in the same order as resolve_crossing().
__device__ __forceinline__ int resolve_crossing_tab(pt P1, pt P2, pt bl, pt br, pt ur, pt ul, unsigned k48, unsigned k9,
                                                    const char *__restrict__ gb,
                                                    const int *__restrict__ tab /* LDS */, int &dcell, int &dk, bool &killed,
                                                    int *codes = nullptr)
{
    const bool sbl = ccw(P1, P2, bl), sbr = ccw(P1, P2, br), sur = ccw(P1, P2, ur), sul = ccw(P1, P2, ul);
    const bool h1 = (ccw(P1, bl, br) != ccw(P2, bl, br)) && (sbl != sbr);
    const bool h2 = (ccw(P1, br, ur) != ccw(P2, br, ur)) && (sbr != sur);
    const bool h3 = (ccw(P1, ur, ul) != ccw(P2, ur, ul)) && (sur != sul);
    const unsigned ro = h1 ? 0u : (h2 ? 64u : (h3 ? 128u : 192u));            // byte offset of row kc-1 (selected as such)
    const int kc = (int)(ro >> 6) + 1;                                       // (only the probes look at it)
    const int *row = (const int *)((const char *)tab + ro);
    const int4 r0 = *(const int4 *)(row), r1 = *(const int4 *)(row + 4), r2 = *(const int4 *)(row + 8);
    const int bB = row[12];
    // one batch of independent loads: the crossed edge's two ends and the two extension points
    const pt va = geo_pt(gb, k48 + (unsigned)r0.x), vb = geo_pt(gb, k48 + (unsigned)r0.y);
    const pt eA = geo_pt(gb, k48 + (unsigned)r0.z), eB = geo_pt(gb, k48 + (unsigned)r0.w);
    // NewHostCell (:217-243): first the side test of the two ends of the move against each prolongation ...
    bool hitA = ccw(P1, va, eA) != ccw(P2, va, eA);
    bool hitB = ccw(P1, vb, eB) != ccw(P2, vb, eB);
    // ... and only for lanes that pass it the second half of intersect2Seg (a diagonal move is rare: most waves skip this)
#ifdef SITRK_NO_LAZY_HIT
    {
#else
    if (hitA || hitB) {
#endif
        hitA = hitA && (ccw(P1, P2, va) != ccw(P1, P2, eA));
        hitB = hitB && (ccw(P1, P2, vb) != ccw(P1, P2, eB));
    }
    // UpdtInd4NewCell (:257-300); first match wins (if / elif)
    dcell = hitA ? r1.y : (hitB ? r1.z : r1.x);
    dk = hitA ? r2.x : (hitB ? r2.y : r1.w);
    const int msk = hitA ? r2.w : (hitB ? bB : r2.z);
    killed = (k9 & (unsigned)msk) != 0;                                      // Survive (:483-484)
    if (codes) {
        const int cA = (kc == 1) ? 5 : (kc == 2) ? 6 : 8, cB = (kc == 1) ? 6 : (kc == 4) ? 5 : 7;
        codes[0] = kc;
        codes[1] = hitA ? cA : (hitB ? cB : kc);
    }
    return kc;
}

typedef double v2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ pt load_pt_nt(const pt *p)
{
    v2d t = __builtin_nontemporal_load((const v2d *)p);
    return make_pt(t.x, t.y);
}
__device__ __forceinline__ void store_pt_nt(pt *p, pt v)
{
    v2d t;
    t.x = v.y; t.y = v.x;
    __builtin_nontemporal_store(t, (v2d *)p);
}

// Workgroups are dealt round-robin over the 8 XCDs (each with its own 4 MiB L2).  Map the hardware block id
// to a logical one so that XCD x walks the contiguous chunk x of the cell-sorted buoys: the row j-1 records a
// workgroup needs were fetched a few workgroups earlier BY THE SAME XCD.  Bijective for any grid size.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg)
{
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// The same idea at a finer grain: runs of `g` consecutive workgroups (neighbouring tiles, which share 128-byte lines of
// the fields) go to one XCD, the runs themselves stay dealt round-robin so that all HBM channels see one stream.
__device__ __forceinline__ unsigned xcd_group(unsigned bid, unsigned nwg, unsigned g)
{
    const unsigned span = 8u * g, full = (nwg / span) * span;
    if (bid >= full) return bid;
    const unsigned base = (bid / span) * span, r = bid - base;       // r = position inside a span of 8 runs
    return base + (r & 7u) * g + (r >> 3);                            // hardware deals r round-robin: XCD = r & 7
}

// The fused loop evaluates `/1000.` and IsInsideQuadrangle without fp64 divisions (sitrk_geom.h: div1000,
// inside_quad_hot - same results for every input): 0.111 -> 0.101 ms per record on C3.  The one-record kernel is
// bound by memory latency, not by issue, and was measured 4-6 % SLOWER with them (and with pinned loads): it keeps
// the plain forms.  `make EXACTDIV=1` builds the plain forms everywhere for A/B timing.
#ifdef SITRK_EXACT_DIV
#define SITRK_DIV1000(x, ...) ((x) / 1000.)
#define SITRK_INSIDE(y, x, q0, q1, q2, q3, eps) inside_quad(y, x, q0, q1, q2, q3)
#else
#define SITRK_DIV1000(x, ...) div1000(x, ##__VA_ARGS__)
#define SITRK_INSIDE(y, x, q0, q1, q2, q3, eps) inside_quad_hot(y, x, q0, q1, q2, q3, eps)
#endif

// A value that must be loaded where the source loads it: the empty asm is a use the compiler cannot move the load below.
template <typename T> __device__ __forceinline__ void pin_load(T &v) { asm volatile("" : "+v"(v)); }

struct StepArgs {
    int64_t nP;
    int tune;
    int Nj, Ni;
    int jrec;
    double rdt, rmin_conc;
    double eps_mg;                      // 2^-48 * max |F-point coordinate| : margin scale of inside_quad_hot
    const CellGeo *geo;
    const int8_t *orient;               // per-cell orientation bits (cell_orient_kernel)
    const uint8_t *kill;                // the record's packed Survive neighbourhoods, one byte per cell (kill9 of survive_kill9_*_kernel)
    const void *u, *v;
    pt *pos;
    int32_t *cell;
    int32_t *kill_rec;
    const int2 *win;                    // per-buoy (first, last) model record, 2-D time mode only
};

// ---------------------------------------------------------------------------
// Everything ONE buoy does for ONE model record: the body of the reference's per-buoy loop
// (si3_part_tracker.py:382-484).  P = (ry,rx) and the packed host cell c are updated in place;
// returns false when Survive killed the buoy (its position is still advanced: the reference writes
// xPosC[jt+1] before the kill test, :459-460 vs :483-484).
//   UVS : iUVstrategy (:37-40)   1 nearest U/V point, 0 cell mean
// ---------------------------------------------------------------------------
template <typename FT, int UVS>
__device__ __forceinline__ bool advance_record(const StepArgs &a, const FT *__restrict__ u, const FT *__restrict__ v,
                                               const uint8_t *__restrict__ kill9, pt &P, int32_t &c)
{
    const int Ni = a.Ni, Nj = a.Nj;
    const int jT = cell_j(c), iT = cell_i(c);
    const size_t k = (size_t)jT * Ni + iT;
    // cell (jT,iT): F = upper-right vertex, U = right U-point, V = upper V-point
    const CellGeo g11 = a.geo[k];
    const pt F10 = a.geo[k - 1].f;                       // F[jT  ,iT-1]  upper-left
    const pt F01 = a.geo[k - Ni].f;                      // F[jT-1,iT  ]  bottom-right
    const pt F00 = a.geo[k - Ni - 1].f;                  // F[jT-1,iT-1]  bottom-left

    double zU, zV;
    if (UVS == 0) {                                      // :423-425
        zU = 0.5 * ((double)u[k] + (double)u[k - 1]);
        zV = 0.5 * ((double)v[k] + (double)v[k - Ni]);
    } else if (UVS == 2) {                               // extra: linear interpolation (not in the reference)
        zU = lerp_on_segment(P, a.geo[k - 1].u, g11.u, (double)u[k - 1], (double)u[k]);
        zV = lerp_on_segment(P, a.geo[k - Ni].v, g11.v, (double)v[k - Ni], (double)v[k]);
    } else {                                             // :427-441
        const pt U10 = a.geo[k - 1].u;                   // U[jT,iT-1]
        const pt V01 = a.geo[k - Ni].v;                  // V[jT-1,iT]
        // (the compiler sinks the two "other side" loads into the branch that selects them: a wave where no buoy
        // picks the far point skips them, measured faster here than loading all four up front)
        const double u1 = (double)u[k], u0 = (double)u[k - 1];
        const double v1 = (double)v[k], v0 = (double)v[k - Ni];
        const bool llum1 = intersect2seg(P, g11.f, V01, g11.v);
        const bool llvm1 = intersect2seg(P, g11.f, U10, g11.u);
        zU = llum1 ? u0 : u1;
        zV = llvm1 ? v0 : v1;
    }

    // forward Euler, one step per record (:452-458): km += (m/s * s) / 1000   (true divisions, like the reference)
    const double dx = zU * a.rdt;
    const double dy = zV * a.rdt;
    pt Pn;
    Pn.x = P.x + dx / 1000.;
    Pn.y = P.y + dy / 1000.;

    bool killed = false;
    // still inside the host cell? (:466) quad = [bl, br, ur, ul]
    if (!inside_quad(Pn.y, Pn.x, F00, F01, g11.f, F10))
        c = resolve_crossing(P, Pn, F00, F01, g11.f, F10, jT, iT, Nj, Ni, a.geo, (unsigned)kill9[k], killed);   // :474-484
    P = Pn;
    return !killed;
}

// ---------------------------------------------------------------------------
// One model record for every buoy, one buoy per lane (sitrk_step).
//   WINDOW : per-buoy first/last model record (2-D time mode, :264-318,380)
// ---------------------------------------------------------------------------
template <typename FT, int UVS, bool WINDOW, int BLOCK>
__global__ __launch_bounds__(BLOCK, 8) void advect_step_kernel(StepArgs a)
{
    const unsigned blk = (a.tune & TUNE_XCD_REMAP) ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int64_t p = (int64_t)blk * BLOCK + threadIdx.x;
    if (p >= a.nP) return;
    const bool nt = (a.tune & TUNE_NT_STATE) != 0;
    int32_t c = nt ? __builtin_nontemporal_load(&a.cell[p]) : a.cell[p];
    if (c < 0) return;                                   // iAlive != 1 (:380)
    if (WINDOW) {
        const int2 w = a.win[p];
        if (a.jrec < w.x || a.jrec > w.y) return;
    }
    pt P = nt ? load_pt_nt(&a.pos[p]) : a.pos[p];        // (ry, rx)
    const int32_t c0 = c;
    const bool alive = advance_record<FT, UVS>(a, (const FT *)a.u, (const FT *)a.v, a.kill, P, c);
    if (nt) store_pt_nt(&a.pos[p], P);
    else a.pos[p] = P;
    if (!alive) a.kill_rec[p] = a.jrec;
    if (c != c0) a.cell[p] = c;
}

// ---------------------------------------------------------------------------
// Several consecutive records in ONE launch (sitrk_run).  Buoys never interact, so the loop nest
// "for record: for buoy" of the reference can be interchanged for the records that are resident
// together: each lane keeps its buoy's position and cell in registers across up to kMaxFuse
// records, the geometry lines it needs stay in L2 between iterations (a buoy moves < 1 cell per
// record), and the once-per-record position/cell streams are read and written once per launch.
// Per buoy the sequence of operations is exactly the one of advect_step_kernel -> identical results.
// ---------------------------------------------------------------------------
static constexpr int kMaxFuse = 32;

struct RunArgs {
    StepArgs s;                         // s.u/s.v/s.kill unused; s.jrec = first record
    int nrec;
    const void *u[kMaxFuse], *v[kMaxFuse];
    const uint8_t *kill9[kMaxFuse];     // the neighbours' Survive bits of each record, one byte per cell (pack_kill9_kernel)
    const pt *geoF;                     // F-points alone, 16 B per cell (what the LDS patch is filled from)
    CrossTab tab;                       // crossing table (make_cross_tab), copied to LDS by every workgroup
    int dji[4][7][2];                   // its (dj,di) pairs: va, vb, eA, eB, S, A, B per crossed edge (for the patch's own offsets)
    int patch_cells, patch_margin;      // LDS patch: capacity in cells (0 = no patch) and the largest margin to try
    int xcd_group;                      // > 1: runs of that many consecutive workgroups share an XCD
#ifdef SITRK_DIAG
    unsigned long long *stamps;         // diagnostic builds: per wave 8 accumulated s_memtime intervals of the record loop (or null)
#endif
    int f32_class;                      // v_cmp_class mask of div1000_of_f32: finite and non-zero, or 0 when |rdt| is outside
                                        // [2^-700, 2^700] (every lane then divides)
};

// ---------------------------------------------------------------------------
// The fused kernel keeps the HOST CELL'S CONTEXT in registers across records: the 8 geometry points and
// the two record-independent orientation terms of the velocity pick (ccw(F,V01,V11), ccw(F,U10,U11)) are
// (re)loaded only when the buoy changes cell (3-16 % of the records); per record only the four velocity
// candidates are loaded.  Same operations on the same operands in the same order as advance_record.
// 72 VGPRs, no scratch -> 7 waves/SIMD (78 -> 6 with per-buoy record windows; round 1: 91 VGPRs, 5 waves; requesting the
// next record's velocities one record ahead was measured 7 % slower then: more instructions in the loop).
// ---------------------------------------------------------------------------
struct CellCtx {
    unsigned o1;                        // byte offset of cell (jT,iT) inside a field of the record (32 bits: sitrk_set_grid keeps
                                        // Nj*Ni*8 below 2^32) -> a record's velocities are loaded as (scalar record pointer + this
                                        // offset [- one row for v[jT-1,iT]]), next to no address arithmetic per record
    pt F11, U11, V11, F10, U10, F01, V01, F00;
    unsigned ori;                       // the cell's orientation byte as loaded: bit 0 = ccw(F11,V01,V11), bit 1 = ccw(F11,U10,U11);
                                        // unpacked where it is used, one record later, not behind its own load in the crossing path
};

// all eight points from two 32-bit byte offsets (the cell's record and the one a row below) + immediates
template <unsigned ES>
__device__ __forceinline__ void load_ctx(const StepArgs &a, const char *__restrict__ gb, unsigned kcell, CellCtx &x)
{
    const unsigned Ni = (unsigned)a.Ni;
    x.o1 = kcell * ES;
    const unsigned k48 = kcell * (unsigned)sizeof(CellGeo), k48b = k48 - Ni * (unsigned)sizeof(CellGeo);
    x.F11 = geo_pt(gb, k48, 0); x.U11 = geo_pt(gb, k48, 16); x.V11 = geo_pt(gb, k48, 32);
    x.F10 = geo_pt(gb, k48, -48); x.U10 = geo_pt(gb, k48, -32);
    x.F01 = geo_pt(gb, k48b, 0); x.V01 = geo_pt(gb, k48b, 32);
    x.F00 = geo_pt(gb, k48b, -48);
    x.ori = (unsigned)(uint8_t)a.orient[kcell];
}

// ---------------------------------------------------------------------------
// The F-point PATCH of a workgroup in LDS.
// What bounds the fused loop (profiles/r02d_pmc_old_vs_table.txt, r02e A/B): neither VALU issue (58-80 % busy) nor HBM
// (2.9 TB/s) but the CHAIN of dependent memory round trips every wave walks every record -- velocities of the (new) cell,
// then the crossing path's extension points, then the new cell's context -- because nearly every wave has a buoy that
// changes cell in every record; fewer instructions at the same chain length bought nothing (-29 % VALU, +9 % time).
// Buoys are sorted by host cell, so a workgroup's 256 buoys sit in a compact patch of cells (one or two 8x16 tiles) and
// move a few cells per launch: the patch's F-points (16 B per cell: a few KB, so occupancy is not touched) are copied
// ONCE per launch into LDS, and CrossedEdge / NewHostCell / the new cell's quad read them with ds_read_b128 (two short
// LDS hops instead of two global round trips).  The new cell's U/V points are then requested together with the next
// record's velocities: one global round trip per record is left on the chain.  A buoy that leaves the patch (or a
// workgroup whose buoys do not fit one: unsorted sets, the end of a tile row) reads global memory exactly as before.
// ---------------------------------------------------------------------------
struct Patch {
    int R0, C0, PR, PC;                 // rows [R0, R0+PR) x columns [C0, C0+PC) of the mesh; PR = PC = 3: no patch (covers nothing)
};

// LDS byte offset of the F-point of cell (R0 + jr, C0 + ir)
__device__ __forceinline__ unsigned patch_off(const Patch &pa, int jr, int ir) { return (unsigned)(jr * pa.PC + ir) * (unsigned)sizeof(pt); }

// every F-point a buoy hosted by cell (R0+jr, C0+ir) can ask for -- its quad (rows jr-1..jr, columns ir-1..ir) and the
// extension points of NewHostCell (rows jr-2..jr+1, columns ir-2..ir+1) -- lies inside the patch
__device__ __forceinline__ bool patch_covers(const Patch &pa, int jr, int ir)
{
    return (unsigned)(jr - 2) < (unsigned)(pa.PR - 3) && (unsigned)(ir - 2) < (unsigned)(pa.PC - 3);
}

// A patch point by its LDS ADDRESS (32 bits, less kLdsBias so that the immediates -16 .. stay non-negative and fold into
// the instruction's offset field): `la` already contains the patch's base -- no address arithmetic per access
static constexpr int kLdsBias = 64;
typedef __attribute__((address_space(3))) const v2d lds_cv2d;
__device__ __forceinline__ pt lds_pt_at(unsigned address)
{
    const v2d t = *(lds_cv2d *)(uintptr_t)address;       // one ds_read_b128; members in the order of struct pt (y, x)
    return make_pt(t.x, t.y);
}
__device__ __forceinline__ pt lds_pt(unsigned la, int imm = 0) { return lds_pt_at(la + (unsigned)(imm + kLdsBias)); }

// the context of cell `kcell`: its quad from the patch (biased LDS address lo of its own F-point), the U/V points and the
// orientation byte from global memory
template <unsigned ES>
__device__ __forceinline__ void load_ctx_lds(const StepArgs &a, const Patch &pa, const char *__restrict__ gb,
                                             unsigned kcell, unsigned lo, CellCtx &x)
{
    const unsigned Ni = (unsigned)a.Ni;
    x.o1 = kcell * ES;
    const unsigned k48 = kcell * (unsigned)sizeof(CellGeo), k48b = k48 - Ni * (unsigned)sizeof(CellGeo);
    x.U11 = geo_pt(gb, k48, 16); x.V11 = geo_pt(gb, k48, 32); x.U10 = geo_pt(gb, k48, -32); x.V01 = geo_pt(gb, k48b, 32);
    x.ori = (unsigned)(uint8_t)a.orient[kcell];
    const unsigned lob = lo - (unsigned)pa.PC * (unsigned)sizeof(pt);
    x.F11 = lds_pt(lo, 0); x.F10 = lds_pt(lo, -16);
    x.F01 = lds_pt(lob, 0); x.F00 = lds_pt(lob, -16);
}

// resolve_crossing_tab() with every point read from the patch: same predicates, same operands, same order.
// tabL row kc-1 (16 ints like a row of tab): [0] va [1] vb [2] eA [3] eB  LDS byte offsets relative to the host cell's
// biased address (i.e. + kLdsBias); [4] S [5] A [6] B  LDS byte increments to the destination cell's record
__device__ __forceinline__ void resolve_crossing_lds(pt P1, pt P2, pt bl, pt br, pt ur, pt ul, unsigned lo, unsigned k9,
                                                     const int *__restrict__ tab, const int *__restrict__ tabL, int &dcell, int &dk,
                                                     int &dlo, bool &killed)
{
#ifdef SITRK_ABL_NOEDGE                 // ablation (timing only, WRONG results): the crossed edge from one comparison instead of CrossedEdge
    const unsigned ro = (P2.x > ur.x) ? 64u : ((P2.y > ur.y) ? 128u : ((P2.y <= bl.y) ? 0u : 192u));
#else
    const bool sbl = ccw(P1, P2, bl), sbr = ccw(P1, P2, br), sur = ccw(P1, P2, ur), sul = ccw(P1, P2, ul);
    const bool h1 = (ccw(P1, bl, br) != ccw(P2, bl, br)) && (sbl != sbr);
    const bool h2 = (ccw(P1, br, ur) != ccw(P2, br, ur)) && (sbr != sur);
    const bool h3 = (ccw(P1, ur, ul) != ccw(P2, ur, ul)) && (sur != sul);
    const unsigned ro = h1 ? 0u : (h2 ? 64u : (h3 ? 128u : 192u));            // both tables have 64-byte rows
#endif
    const int *row = (const int *)((const char *)tab + ro), *rowL = (const int *)((const char *)tabL + ro);
    const int4 r1 = *(const int4 *)(row + 4), r2 = *(const int4 *)(row + 8);
    const int bB = row[12];
    const int4 l0 = *(const int4 *)(rowL), l1 = *(const int4 *)(rowL + 4);
#ifdef SITRK_ABL_NODIAG                 // ablation (timing only, WRONG results): no NewHostCell tests, no extension points
    bool hitA = false, hitB = false;
    (void)l0;
#else
    const pt va = lds_pt_at(lo + (unsigned)l0.x), vb = lds_pt_at(lo + (unsigned)l0.y);     // (the table's offsets carry the bias)
    const pt eA = lds_pt_at(lo + (unsigned)l0.z), eB = lds_pt_at(lo + (unsigned)l0.w);
    bool hitA = ccw(P1, va, eA) != ccw(P2, va, eA);
    bool hitB = ccw(P1, vb, eB) != ccw(P2, vb, eB);
    if (hitA || hitB) {
        hitA = hitA && (ccw(P1, P2, va) != ccw(P1, P2, eA));
        hitB = hitB && (ccw(P1, P2, vb) != ccw(P1, P2, eB));
    }
#endif
    dcell = hitA ? r1.y : (hitB ? r1.z : r1.x);
    dk = hitA ? r2.x : (hitB ? r2.y : r1.w);
    dlo = hitA ? l1.y : (hitB ? l1.z : l1.x);
    const int msk = hitA ? r2.w : (hitB ? bB : r2.z);
    killed = (k9 & (unsigned)msk) != 0;
}

static constexpr int kRunLdsFixed = 256 + 256 + 64;      // crossing table, its LDS-offset twin (same row stride), bounding box / patch header

// Wave priorities inside the fused loop (s_setprio: a scheduling hint, results untouched).  The loop sits where vector issue and
// the per-wave dependency chain meet; which of the seven waves of a SIMD issues next is the arbiter's choice.
// Raised priority for the record's loads (they go out before other waves' arithmetic: their latency overlaps more of it) and
// for the crossing path (the long stretch of a wave's chain, after which it can request its next operands): +1.1 ... +2.2 % on C3,
// +1.6 % on C2, five interleaved rounds (profiles/r04y_prio_ab.txt); (3,3), (3,1), (1,3), (2,1) are within 0.3 % of (3,2), main
// path high / the rest low is no better than no hint.  -DSITRK_PRIO_LOADS=0 -DSITRK_PRIO_CROSS=0 builds the loop without the hints.
#ifndef SITRK_PRIO_LOADS
#define SITRK_PRIO_LOADS 3
#endif
#ifndef SITRK_PRIO_CROSS
#define SITRK_PRIO_CROSS 2
#endif
#ifndef SITRK_PRIO_BASE
#define SITRK_PRIO_BASE 0
#endif
#ifndef SITRK_RUN_BLOCK
#define SITRK_RUN_BLOCK 256             // workgroup size of the fused kernel (A/B: tools/build_variant.sh x -DSITRK_RUN_BLOCK=128 ...)
#endif
static constexpr int kRunBlock = SITRK_RUN_BLOCK;
#ifndef SITRK_RUN_WAVES
#define SITRK_RUN_WAVES 7               // 72 VGPRs, no scratch (the cell is stored behind a lane flag, crel and the row-below offset are
#endif                                  // derived where they are used): +2.6 % over 6 waves; 8 waves (64 VGPRs) spill 32 registers
#ifndef SITRK_RUN_WAVES_WINDOW
#define SITRK_RUN_WAVES_WINDOW 6        // the form with per-buoy record windows carries two more registers: 9 spilled at 7 waves (13-25
                                        // with the window packed into one register, as a single test, or as a predicate on the body)
#endif
// In-kernel stamps (diagnostic builds, `make DIAG=1`, knob "stamps"): where ONE wave's time goes inside a record.  s_memtime ticks
// are shader cycles; the values go to a buffer nothing else reads.  SITRK_STAMP(k) closes interval k.
#if defined(SITRK_DIAG) && defined(SITRK_NO_STAMPS)      // ablation builds that must keep the shipped register budget
#define SITRK_STAMP_DECL const bool st_on = false;
#define SITRK_STAMP_START
#define SITRK_STAMP(k)
#define SITRK_STAMP_FLUSH(widx)
#elif defined(SITRK_DIAG)
#define SITRK_STAMP_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_t = 0; const bool st_on = ra.stamps != nullptr;
#define SITRK_STAMP_START if (st_on) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t) :: "memory"); }
#define SITRK_STAMP(k) if (st_on) { unsigned long long st_n; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_n) :: "memory"); st_acc[k] += st_n - st_t; st_t = st_n; }
#define SITRK_STAMP_FLUSH(widx) if (st_on && (threadIdx.x & 63u) == 0) { for (int q_ = 0; q_ < 8; q_++) ra.stamps[(size_t)(widx) * 8 + q_] = st_acc[q_]; }
#else
#define SITRK_STAMP_DECL
#define SITRK_STAMP_START
#define SITRK_STAMP(k)
#define SITRK_STAMP_FLUSH(widx)
#endif

template <typename FT, int UVS, bool WINDOW>
__global__ __launch_bounds__(kRunBlock, WINDOW ? SITRK_RUN_WAVES_WINDOW : SITRK_RUN_WAVES) void advect_run_kernel(RunArgs ra)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *s_tab = (int *)smem;                            // CrossTab, 64 ints
    int *s_tabL = s_tab + 64;                            // 4 rows of 16 ints (7 used)
    int *s_box = s_tabL + 64;                            // [0..3] jmin jmax imin imax of the live buoys; [4..7] R0 C0 PR PC
    const StepArgs &a = ra.s;
    const unsigned blk = (a.tune & TUNE_XCD_REMAP) ? xcd_remap(blockIdx.x, gridDim.x)
                         : (ra.xcd_group > 1 ? xcd_group(blockIdx.x, gridDim.x, (unsigned)ra.xcd_group) : blockIdx.x);
    const int64_t p = (int64_t)blk * kRunBlock + threadIdx.x;
    const bool nt = (a.tune & TUNE_NT_STATE) != 0;
    int32_t c = -1;
    if (p < a.nP) c = nt ? __builtin_nontemporal_load(&a.cell[p]) : a.cell[p];
    const bool live = c >= 0;                            // iAlive == 1 (:380)
    if (threadIdx.x < 64) s_tab[threadIdx.x] = ((const int *)&ra.tab)[threadIdx.x];
    if (threadIdx.x == 0) { s_box[0] = 0x7fffffff; s_box[1] = -1; s_box[2] = 0x7fffffff; s_box[3] = -1; }
    __syncthreads();
    // ---- the workgroup's patch: bounding box of its live buoys' host cells, widened by the stencil and by as many
    //      cells of margin as the LDS budget allows (a buoy moves < 1 cell per record)
    {
        int jlo = live ? cell_j(c) : 0x7fffffff, jhi = live ? cell_j(c) : -1, ilo = live ? cell_i(c) : 0x7fffffff, ihi = live ? cell_i(c) : -1;
        for (int off = 32; off > 0; off >>= 1) {
            jlo = min(jlo, __shfl_xor(jlo, off)); jhi = max(jhi, __shfl_xor(jhi, off));
            ilo = min(ilo, __shfl_xor(ilo, off)); ihi = max(ihi, __shfl_xor(ihi, off));
        }
        if ((threadIdx.x & 63) == 0 && jhi >= 0) {
            atomicMin(&s_box[0], jlo); atomicMax(&s_box[1], jhi); atomicMin(&s_box[2], ilo); atomicMax(&s_box[3], ihi);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int R0 = 0, C0 = 0, PR = 3, PC = 3;              // "no patch": patch_covers() is false for every cell
        if (s_box[1] >= 0 && ra.patch_cells > 0) {
            const int nr = s_box[1] - s_box[0] + 1 + 3, nc = s_box[3] - s_box[2] + 1 + 3;      // rows jmin-2 .. jmax+1
            int m = -1;
            for (int t = 0; t <= ra.patch_margin; t++)
                if ((int64_t)(nr + 2 * t) * (nc + 2 * t) <= ra.patch_cells) m = t;
            if (m >= 0) {
                R0 = max(0, s_box[0] - 2 - m); C0 = max(0, s_box[2] - 2 - m);
                PR = min(a.Nj, s_box[1] + 2 + m) - R0; PC = min(a.Ni, s_box[3] + 2 + m) - C0;
            }
        }
        s_box[4] = R0; s_box[5] = C0; s_box[6] = PR; s_box[7] = PC;
    }
    __syncthreads();
    Patch pa;
    pa.R0 = s_box[4]; pa.C0 = s_box[5]; pa.PR = s_box[6]; pa.PC = s_box[7];
    char *s_geo = smem + kRunLdsFixed;
    if (threadIdx.x < 28) {
        // LDS twins of the table's geometry offsets: (dj*PC + di) * 16 for va, vb, eA, eB, S, A, B of each crossed edge
        const int e = threadIdx.x / 7, q = threadIdx.x % 7;
        s_tabL[16 * e + q] = (ra.dji[e][q][0] * pa.PC + ra.dji[e][q][1]) * (int)sizeof(pt) + (q < 4 ? kLdsBias : 0);
    }
    const char *__restrict__ gb = (const char *)a.geo;
    if (pa.PR > 3) {
        // a row of the patch is contiguous in the F-only copy of the geometry
        const int ncell = pa.PR * pa.PC;
        for (int t = threadIdx.x; t < ncell; t += kRunBlock) {
            const int r = t / pa.PC, cc = t - r * pa.PC;
            *(v2d *)(s_geo + (size_t)t * sizeof(pt)) = *(const v2d *)(ra.geoF + ((size_t)(pa.R0 + r) * a.Ni + pa.C0 + cc));
        }
    }
    __syncthreads();                                     // last barrier: from here on lanes may leave
    if (!live) return;
    int first = 0, last = 0x7fffffff;
    if (WINDOW) { const int2 w = a.win[p]; first = w.x; last = w.y; }
    pt P = nt ? load_pt_nt(&a.pos[p]) : a.pos[p];
    bool moved = false, recelled = false;               // (lane flags: the cell is stored only if it changed)
    CellCtx x;
    // position of the host cell inside the patch, packed like the cell itself (jr << 16 | ir), and the LDS offset of its record
    const int porg = (pa.R0 << 16) | pa.C0;
    int crel = c - porg;
    const unsigned geo_la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)s_geo - (unsigned)kLdsBias;
    unsigned lo = geo_la + patch_off(pa, crel >> 16, crel & 0xffff);
    bool inl = patch_covers(pa, crel >> 16, crel & 0xffff);
    {
        const unsigned kcell = (unsigned)(cell_j(c) * a.Ni + cell_i(c));
        if (inl) load_ctx_lds<sizeof(FT)>(a, pa, gb, kcell, lo, x);
        else load_ctx<sizeof(FT)>(a, gb, kcell, x);
    }
    // the record pointers (scalar loads from the kernel arguments) are fetched one record ahead: a record's vector
    // loads go out at the top of its iteration instead of behind a scalar load and its wait (-0.7 %)
    const char *ub_next = (const char *)ra.u[0], *vb_next = (const char *)ra.v[0], *kb_next = (const char *)ra.kill9[0];
    double k1000 = 1000.;                                // div1000's constant, kept in scalar registers
    asm volatile("" : "+s"(k1000));
    SITRK_STAMP_DECL
#pragma unroll 1
    for (int r = 0; r < ra.nrec; r++) {
        const int jrec = a.jrec + r;
        SITRK_STAMP_START
        const char *ub = ub_next, *vb = vb_next, *kb = kb_next;
        const int rn = (r + 1 < ra.nrec) ? r + 1 : r;
        ub_next = (const char *)ra.u[rn]; vb_next = (const char *)ra.v[rn]; kb_next = (const char *)ra.kill9[rn];
        if (WINDOW) {
            if (jrec < first) continue;
            if (jrec > last) break;
        }
        // the four velocity candidates u[jT,iT-1], u[jT,iT], v[jT-1,iT], v[jT,iT]
#if SITRK_PRIO_LOADS > 0                // the record's loads issued at raised wave priority
        __builtin_amdgcn_s_setprio(SITRK_PRIO_LOADS);
#endif
#ifdef SITRK_ABL_VEL2                   // ablation (timing only, WRONG results): two velocity loads instead of three
        FT fu1 = *(const FT *)(ub + x.o1), fu0 = fu1;
        FT fv1 = *(const FT *)(vb + x.o1), fv0 = fv1;
#else
        FT fu0 = *(const FT *)(ub + x.o1 - sizeof(FT)), fu1 = *(const FT *)(ub + x.o1);
        FT fv0 = *(const FT *)(vb + (x.o1 - (unsigned)a.Ni * (unsigned)sizeof(FT))), fv1 = *(const FT *)(vb + x.o1);
#endif
        // ... and the Survive byte of the cell's 8 neighbours for this record (used only if the buoy leaves the cell)
#ifdef SITRK_ABL_NOK9                   // ablation (timing only, WRONG results): no Survive byte
        unsigned k9 = 0; (void)kb;
#else
        unsigned k9 = *(const uint8_t *)(kb + (x.o1 >> (sizeof(FT) == 4 ? 2 : 3)));
#endif
#if SITRK_PRIO_LOADS > 0
        __builtin_amdgcn_s_setprio(SITRK_PRIO_BASE);
#endif
#ifdef SITRK_ABL_VELCONST               // ablation (timing only, WRONG results): a uniform drift instead of the record's velocities, the
        {                                   // loads kept as dependencies -- from global memory as shipped, or (SITRK_ABL_VELLDS) as LDS
#ifdef SITRK_ABL_VELLDS                     // reads at a cell-dependent address: what staging u/v patches through LDS could buy at best
            typedef __attribute__((address_space(3))) const float lds_cf;
            const unsigned la_ = inl ? lo : geo_la;
            const float l0 = *(lds_cf *)(uintptr_t)(la_ + kLdsBias), l1 = *(lds_cf *)(uintptr_t)(la_ + kLdsBias + 4),
                        l2 = *(lds_cf *)(uintptr_t)(la_ + kLdsBias + 8), l3 = *(lds_cf *)(uintptr_t)(la_ + kLdsBias + 12);
#else
            const FT l0 = fu0, l1 = fu1, l2 = fv0, l3 = fv1;
#endif
            float cu0, cu1, cv0, cv1;       // 0.1 and 0.05 m/s, each "computed from" one loaded value
            asm volatile("v_mov_b32 %0, 0x3dcccccd" : "=v"(cu0) : "v"(l0));
            asm volatile("v_mov_b32 %0, 0x3dcccccd" : "=v"(cu1) : "v"(l1));
            asm volatile("v_mov_b32 %0, 0x3d4ccccd" : "=v"(cv0) : "v"(l2));
            asm volatile("v_mov_b32 %0, 0x3d4ccccd" : "=v"(cv1) : "v"(l3));
            fu0 = (FT)cu0; fu1 = (FT)cu1; fv0 = (FT)cv0; fv1 = (FT)cv1;
        }
#endif
        double zU, zV;
        FT su = 0, sv = 0;                               // UVS == 1: the selected candidates as loaded
        if (UVS == 0) {                                  // :423-425
            zU = 0.5 * ((double)fu1 + (double)fu0);
            zV = 0.5 * ((double)fv1 + (double)fv0);
        } else if (UVS == 2) {                           // extra: linear interpolation (not in the reference)
            zU = lerp_on_segment(P, x.U10, x.U11, (double)fu0, (double)fu1);
            zV = lerp_on_segment(P, x.V01, x.V11, (double)fv0, (double)fv1);
        } else {                                         // :427-441
            // all four candidates are requested up front (pin_load: none is sunk into the branch that selects it)
            // intersect2Seg(P,F,C,D) = (ccw(P,C,D) != ccw(F,C,D)) and (ccw(P,F,C) != ccw(P,F,D)); ccw(F,C,D) is per cell
#ifdef SITRK_DIAG
            if (st_on) { pin_load(x.ori); pin_load(x.U11.y); pin_load(x.V11.y); pin_load(x.U10.y); pin_load(x.V01.y); pin_load(x.F11.y); pin_load(x.F00.y); }
            SITRK_STAMP(0)                               // the cell's context is there (requested by the previous record's crossing path)
#endif
            const bool sFV = (x.ori & 1u) != 0, sFU = (x.ori & 2u) != 0;
            const bool llum1 = (ccw(P, x.V01, x.V11) != sFV) && (ccw(P, x.F11, x.V01) != ccw(P, x.F11, x.V11));
            const bool llvm1 = (ccw(P, x.U10, x.U11) != sFU) && (ccw(P, x.F11, x.U10) != ccw(P, x.F11, x.U11));
#ifdef SITRK_DIAG
            if (st_on) { int b_ = (llum1 ? 1 : 0) | (llvm1 ? 2 : 0); asm volatile("" : "+v"(b_)); }
            SITRK_STAMP(1)                               // the pick's orientation tests
#endif
            pin_load(fu0); pin_load(fv0);
#ifdef SITRK_DIAG
            if (st_on) { pin_load(fu1); pin_load(fv1); }
            SITRK_STAMP(2)                               // the record's velocities are there
#endif
            su = llum1 ? fu0 : fu1;
            sv = llvm1 ? fv0 : fv1;
            zU = (double)su;
            zV = (double)sv;
        }
        const double dx = zU * a.rdt;                    // :452-458
        const double dy = zV * a.rdt;
        pt Pn;
#ifndef SITRK_EXACT_DIV
        if (UVS == 1 && sizeof(FT) == 4) {               // binary32 records: the range test of div1000 on the value as loaded
            Pn.x = P.x + div1000_of_f32(dx, (float)su, ra.f32_class, k1000);
            Pn.y = P.y + div1000_of_f32(dy, (float)sv, ra.f32_class, k1000);
        } else
#endif
        {
            Pn.x = P.x + SITRK_DIV1000(dx, k1000);
            Pn.y = P.y + SITRK_DIV1000(dy, k1000);
        }
        moved = true;
        bool killed = false;
#ifdef SITRK_DIAG
        if (st_on) { pin_load(Pn.x); pin_load(Pn.y); }
        SITRK_STAMP(3)                                   // Euler update
#endif
        const bool still_in = SITRK_INSIDE(Pn.y, Pn.x, x.F00, x.F01, x.F11, x.F10, a.eps_mg);
#ifdef SITRK_DIAG
        if (st_on) { int b_ = still_in ? 1 : 0; asm volatile("" : "+v"(b_)); }
        SITRK_STAMP(4)                                   // cell test
#endif
        if (!still_in) {      // :466-484
#if SITRK_PRIO_CROSS > 0                // the crossing path -- the long part of a wave's chain -- at raised priority
            __builtin_amdgcn_s_setprio(SITRK_PRIO_CROSS);
#endif
            const unsigned kcell = x.o1 / (unsigned)sizeof(FT);
            int dcell, dk, dlo = 0;
            pin_load(k9);
            if (inl) {
                resolve_crossing_lds(P, Pn, x.F00, x.F01, x.F11, x.F10, lo, k9, s_tab, s_tabL, dcell, dk, dlo, killed);
            } else {
                resolve_crossing_tab(P, Pn, x.F00, x.F01, x.F11, x.F10, kcell * (unsigned)sizeof(CellGeo), k9, gb, s_tab, dcell, dk, killed);
            }
            c += dcell;
            recelled = true;
            const int crel = c - porg;                  // (derived, not carried from record to record; c is a live cell here)
            // (a killed buoy's destination is inside the mesh too: its context is loaded unconditionally, which keeps the
            // loads out of the shadow of the Survive test)
            if (inl) {
                lo += (unsigned)dlo;
            } else {
                lo = geo_la + patch_off(pa, crel >> 16, crel & 0xffff);
            }
            inl = patch_covers(pa, crel >> 16, crel & 0xffff);
            if (inl) load_ctx_lds<sizeof(FT)>(a, pa, gb, kcell + (unsigned)dk, lo, x);
            else load_ctx<sizeof(FT)>(a, gb, kcell + (unsigned)dk, x);
#if SITRK_PRIO_CROSS > 0
            __builtin_amdgcn_s_setprio(SITRK_PRIO_BASE);
#endif
        }
        P = Pn;
#ifdef SITRK_DIAG
        if (st_on) { int b_ = c; asm volatile("" : "+v"(b_)); }
        SITRK_STAMP(5)                                   // crossing path (resolution; the new context is requested, not waited for)
#endif
        if (killed) {
            c |= SITRK_DEAD_BIT;
            unsigned tk = threadIdx.x;
            asm volatile("" : "+v"(tk));
            a.kill_rec[(int64_t)blk * kRunBlock + tk] = jrec;
            break;                                       // dead buoys never step again
        }
    }
    SITRK_STAMP_FLUSH((size_t)blk * (kRunBlock / 64) + (threadIdx.x >> 6))
    // (the buoy's index is recomputed here rather than kept: 16 bytes of state addresses per lane would be spilled to
    // scratch across the loop, i.e. written and read back through HBM)
    unsigned tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int64_t pe = (int64_t)blk * kRunBlock + tid;
    if (moved) {
        if (nt) store_pt_nt(&a.pos[pe], P);
        else a.pos[pe] = P;
    }
    if (recelled) a.cell[pe] = c;
}

#ifdef SITRK_DIAG
// ---------------------------------------------------------------------------
// ABLATION KERNELS (diagnostic builds only, `make DIAG=1`; results are WRONG by design).
//   memonly : issues exactly the loads/stores of the hot path, no predicates  -> memory-side time
//   nocross : the hot path without CrossedEdge/NewHostCell/Survive            -> price of the crossing path
// ---------------------------------------------------------------------------
template <typename FT>
__global__ __launch_bounds__(kBlock) void advect_memonly_kernel(StepArgs a)
{
    int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= a.nP) return;
    int32_t c = a.cell[p];
    if (c < 0) return;
    const int Ni = a.Ni;
    const size_t k = (size_t)cell_j(c) * Ni + cell_i(c);
    const FT *__restrict__ u = (const FT *)a.u;
    const FT *__restrict__ v = (const FT *)a.v;
    const pt P = a.pos[p];
    const CellGeo g11 = a.geo[k];
    const pt F10 = a.geo[k - 1].f, U10 = a.geo[k - 1].u, F01 = a.geo[k - Ni].f, V01 = a.geo[k - Ni].v, F00 = a.geo[k - Ni - 1].f;
    const double s = (double)u[k] + (double)u[k - 1] + (double)v[k] + (double)v[k - Ni];
    pt Pn;
    Pn.y = P.y + 1e-300 * (g11.f.y + g11.u.y + g11.v.y + F10.y + U10.y + F01.y + V01.y + F00.y + s);
    Pn.x = P.x + 1e-300 * (g11.f.x + g11.u.x + g11.v.x + F10.x + U10.x + F01.x + V01.x + F00.x);
    a.pos[p] = Pn;
}

template <typename FT>
__global__ __launch_bounds__(kBlock) void advect_nocross_kernel(StepArgs a)
{
    int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= a.nP) return;
    int32_t c = a.cell[p];
    if (c < 0) return;
    const int Ni = a.Ni;
    const size_t k = (size_t)cell_j(c) * Ni + cell_i(c);
    const FT *__restrict__ u = (const FT *)a.u;
    const FT *__restrict__ v = (const FT *)a.v;
    const pt P = a.pos[p];
    const CellGeo g11 = a.geo[k];
    const pt F10 = a.geo[k - 1].f, U10 = a.geo[k - 1].u, F01 = a.geo[k - Ni].f, V01 = a.geo[k - Ni].v, F00 = a.geo[k - Ni - 1].f;
    const double u1 = (double)u[k], u0 = (double)u[k - 1], v1 = (double)v[k], v0 = (double)v[k - Ni];
    const bool llum1 = intersect2seg(P, g11.f, V01, g11.v);
    const bool llvm1 = intersect2seg(P, g11.f, U10, g11.u);
    const double zU = llum1 ? u0 : u1, zV = llvm1 ? v0 : v1;
    pt Pn;
    Pn.x = P.x + (zU * a.rdt) / 1000.;
    Pn.y = P.y + (zV * a.rdt) / 1000.;
    a.pos[p] = Pn;
    if (!inside_quad(Pn.y, Pn.x, F00, F01, g11.f, F10)) a.cell[p] = c ^ 1;      // keep the test alive, skip the rest
}
#endif  // SITRK_DIAG

// ---------------------------------------------------------------------------
// Predicate probes: the device functions of the hot path evaluated on plain arrays, so that the parity
// tests can hold them against the reference's golden vectors one predicate at a time (sitrk_eval_*).
// ---------------------------------------------------------------------------
__global__ void eval_inside_kernel(int64_t n, const pt *__restrict__ pts, const pt *__restrict__ quads, int8_t *__restrict__ out)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const pt q0 = quads[4 * k], q1 = quads[4 * k + 1], q2 = quads[4 * k + 2], q3 = quads[4 * k + 3];
    // the hot loop's division-free form, with the margin scale of this quad alone; bit 1 flags a disagreement with
    // the plain form (never set: the tests compare the byte with 0/1)
    const double mg = fmax(fmax(fmax(fabs(q0.x), fabs(q1.x)), fmax(fabs(q2.x), fabs(q3.x))),
                           fmax(fmax(fabs(q0.y), fabs(q1.y)), fmax(fabs(q2.y), fabs(q3.y))));
    const bool hot = inside_quad_hot(pts[k].y, pts[k].x, q0, q1, q2, q3, 0x1p-48 * mg);
    const bool ref = inside_quad(pts[k].y, pts[k].x, q0, q1, q2, q3);
    out[k] = (hot ? 1 : 0) | (hot != ref ? 2 : 0);
}

// r + (vel * rdt) / 1000. as the hot loop evaluates it (si3_part_tracker.py:452-458): a velocity that is a binary32 value
// (what a float32 record holds) goes the way the fused loop takes for such records, everything else the general way
__global__ void eval_euler_kernel(int64_t n, const double *__restrict__ r, const double *__restrict__ vel, double rdt, int f32_class,
                                  double *__restrict__ out)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double w = vel[k];
    const double d = w * rdt;
    const float wf = (float)w;
    double q = SITRK_DIV1000(d);
#ifndef SITRK_EXACT_DIV
    double k1000 = 1000.;
    asm volatile("" : "+s"(k1000));
    if ((double)wf == w) q = div1000_of_f32(d, wf, f32_class, k1000);
#endif
    out[k] = r[k] + q;
}

__global__ void eval_intersect_kernel(int64_t n, const pt *__restrict__ segs, int8_t *__restrict__ inter, int8_t *__restrict__ ccw_abc)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const pt A = segs[4 * k], B = segs[4 * k + 1], Cc = segs[4 * k + 2], D = segs[4 * k + 3];
    inter[k] = intersect2seg(A, B, Cc, D) ? 1 : 0;
    if (ccw_abc) ccw_abc[k] = ccw(A, B, Cc) ? 1 : 0;
}

__global__ void eval_crossing_kernel(int64_t n, int Nj, int Ni, const CellGeo *__restrict__ geo,
                                     const pt *__restrict__ P1, const pt *__restrict__ P2, const int32_t *__restrict__ jiT,
                                     int32_t *__restrict__ jiT_new, int32_t *__restrict__ codes_out, CrossTab tab)
{
    __shared__ __attribute__((aligned(16))) int s_tab[64];
    if (threadIdx.x < 64) s_tab[threadIdx.x] = ((const int *)&tab)[threadIdx.x];
    __syncthreads();
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int jT = jiT[2 * p], iT = jiT[2 * p + 1];
    const size_t k = (size_t)jT * Ni + iT;
    bool killed;
    int codes[2];
    const pt bl = geo[k - Ni - 1].f, br = geo[k - Ni].f, ur = geo[k].f, ul = geo[k - 1].f;
    int32_t cn = resolve_crossing(P1[p], P2[p], bl, br, ur, ul, jT, iT, Nj, Ni, geo, 0u, killed, codes);
    // the table-driven form of the fused kernel (cells it is used for: no negative-index wrap; 32-bit byte offsets)
    if (jT >= 2 && iT >= 2 && (size_t)Nj * Ni * sizeof(CellGeo) < ((size_t)1 << 32)) {
        int dcell, dk, codes2[2];
        bool killed2;
        resolve_crossing_tab(P1[p], P2[p], bl, br, ur, ul, (unsigned)k * (unsigned)sizeof(CellGeo), 0u, (const char *)geo,
                             s_tab, dcell, dk, killed2, codes2);
        const int32_t cn2 = pack_cell(jT, iT) + dcell;
        const bool same = (cn2 == cn) && (codes2[0] == codes[0]) && (codes2[1] == codes[1]) && !killed2 &&
                          ((int64_t)k + dk == (int64_t)cell_j(cn) * Ni + cell_i(cn));
        if (!same) { cn = pack_cell(0, 0); codes[0] = codes[1] = -1; }      // the two forms disagree: make the tests fail loudly
    }
    jiT_new[2 * p] = cell_j(cn);
    jiT_new[2 * p + 1] = cell_i(cn);
    if (codes_out) {
        codes_out[2 * p] = codes[0];                     // CrossedEdge: 1 bottom, 2 right, 3 upper, 4 left
        codes_out[2 * p + 1] = codes[1];                 // NewHostCell: 1..4 or the diagonals 5..8
    }
}

// ---------------------------------------------------------------------------
// state upload helpers / sort support
// ---------------------------------------------------------------------------
__global__ void iota_kernel(int64_t n, int32_t *v)
{
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) v[s] = (int32_t)s;
}

// Sort key of a host cell; dead buoys last.  tj == 0: row-major cell index.  Otherwise tile-major:
// tiles of tj x ti cells in row-major order, row-major inside a tile, so that a workgroup's 256 buoys
// cover a compact 2-D patch (rows j-1..j of the patch are shared inside the workgroup instead of being
// fetched twice by workgroups a whole grid row apart).
__host__ __device__ __forceinline__ uint32_t cell_key(int j, int i, int Ni, int tj, int ti)
{
    if (tj == 0) return (uint32_t)j * (uint32_t)Ni + (uint32_t)i;
    const uint32_t nti = ((uint32_t)Ni + (uint32_t)ti - 1u) / (uint32_t)ti;
    const uint32_t tile = ((uint32_t)j / (uint32_t)tj) * nti + (uint32_t)i / (uint32_t)ti;
    return tile * (uint32_t)(tj * ti) + ((uint32_t)j % (uint32_t)tj) * (uint32_t)ti + (uint32_t)i % (uint32_t)ti;
}

__global__ void make_keys_kernel(int64_t n, int Ni, int tj, int ti, uint32_t dead_key, const int32_t *__restrict__ cell,
                                 uint32_t *__restrict__ keys, int32_t *__restrict__ vals)
{
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    int32_t c = cell[s];
    keys[s] = (c < 0) ? dead_key : cell_key(cell_j(c), cell_i(c), Ni, tj, ti);
    vals[s] = (int32_t)s;
}

// inverse of cell_key
__host__ __device__ __forceinline__ int32_t cell_of_key(uint32_t key, int Ni, int tj, int ti)
{
    if (tj == 0) return pack_cell((int)(key / (uint32_t)Ni), (int)(key % (uint32_t)Ni));
    const uint32_t nti = ((uint32_t)Ni + (uint32_t)ti - 1u) / (uint32_t)ti, tcells = (uint32_t)(tj * ti);
    const uint32_t tile = key / tcells, w = key - tile * tcells;
    const uint32_t jt = tile / nti, it = tile - jt * nti, jr = w / (uint32_t)ti, ir = w - jr * (uint32_t)ti;
    return pack_cell((int)(jt * (uint32_t)tj + jr), (int)(it * (uint32_t)ti + ir));
}

// The re-sort's tail: slot s of the new order takes the state of slot src[s] of the old one.  A gather costs one 64-byte
// fabric request per array whatever it needs of it (measured on the first sort of a random set: 4.0 requests per buoy for
// pos, cell, kill_rec, perm = 2.5 GB for 280 MB of state, 766 us), so as few arrays as possible are gathered:
//   the host cell IS the sorted key (cell_of_key), the kill record of a live buoy is -1 by definition -- only dead
//   buoys (key == dead_key, the tail of the order) gather those two; the record window is one 8-byte word.
// Left: pos and perm (+ win): 2 (3) requests per buoy instead of 4 (6).
__global__ void permute_state_kernel(int64_t n, const int32_t *__restrict__ src, const uint32_t *__restrict__ skeys, uint32_t dead_key,
                                     int Ni, int tj, int ti, BuoyState in, BuoyState out, bool windowed)
{
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int32_t q = src[s];
    const uint32_t key = skeys[s];
    out.pos[s] = in.pos[q];
    out.perm[s] = in.perm[q];
    if (windowed) out.win[s] = in.win[q];
    if (key == dead_key) {
        out.cell[s] = in.cell[q];
        out.kill_rec[s] = in.kill_rec[q];
    } else {
        out.cell[s] = cell_of_key(key, Ni, tj, ti);
        out.kill_rec[s] = -1;
    }
}

// buoys handed over with a history (migration between ranks): slot s holds the caller's buoy perm[s]
__global__ void restore_state_kernel(int64_t n, BuoyState st, const int8_t *__restrict__ alive, const int32_t *__restrict__ kill_rec,
                                     unsigned long long *__restrict__ rim_alive)
{
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int32_t o = st.perm[s];
    int32_t c = st.cell[s];
    if (!alive[o]) st.cell[s] = c | SITRK_DEAD_BIT;
    else if (cell_j(c) < 2 || cell_i(c) < 2) atomicOr(rim_alive, 1ull);      // a LIVE buoy in the two outermost rows/columns
    // invariant the re-sort relies on (permute_state_kernel): alive <=> kill_rec == -1
    st.kill_rec[s] = alive[o] ? -1 : kill_rec[o];
}

// ---------------------------------------------------------------------------
// fetch: sorted slots -> caller order
// ---------------------------------------------------------------------------
__global__ void fetch_state_kernel(int64_t n, BuoyState st, pt *yx, int32_t *jiT, int8_t *alive, int32_t *kill_rec)
{
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    int32_t o = st.perm[s];
    int32_t c = st.cell[s];
    if (yx) yx[o] = st.pos[s];
    if (jiT) {
        jiT[2 * (int64_t)o] = cell_j(c);
        jiT[2 * (int64_t)o + 1] = cell_i(c);
    }
    if (alive) alive[o] = (c < 0) ? 0 : 1;
    if (kill_rec) kill_rec[o] = st.kill_rec[s];
}

// xPosC[jt+1], xmask[jt+1] of si3_part_tracker.py:459-460 for the record just stepped
__global__ void fetch_record_kernel(int64_t n, int jrec, BuoyState st, bool windowed, pt *yx, int8_t *mask)
{
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    int32_t o = st.perm[s];
    int32_t c = st.cell[s];
    bool in_window = true;
    if (windowed) { const int2 w = st.win[s]; in_window = (jrec >= w.x) && (jrec <= w.y); }
    // stepped at jrec  <=>  was alive before it: still alive, or killed by this very record
    bool stepped = in_window && (c >= 0 || st.kill_rec[s] == jrec);
    if (yx) yx[o] = stepped ? st.pos[s] : make_pt(-9999.0, -9999.0);      // sitrack/ncio.py:19 FillValue
    if (mask) mask[o] = stepped ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void count_alive_kernel(int64_t n, const int32_t *__restrict__ cell, unsigned long long *out)
{
    // grid-stride count, wave ballot, one atomic per workgroup (same-address atomics serialise at the memory side)
    __shared__ unsigned int sw[kBlock / 64];
    unsigned int cnt = 0;
    for (int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x; s < n; s += (int64_t)gridDim.x * kBlock)
        cnt += (cell[s] >= 0) ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int t = 0;
        for (int w = 0; w < kBlock / 64; w++) t += sw[w];
        if (t) atomicAdd(out, (unsigned long long)t);
    }
}

// ---------------------------------------------------------------------------
// FindContainingCell   reference sitrack/locate.py:280-330
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool find_containing_cell(double zy, double zx, int kj, int ki, int Nj, int Ni,
                                                     const CellGeo *__restrict__ geo, int &jT, int &iT)
{
    const int dj[5] = {0, 0, 1, 0, -1};
    const int di[5] = {0, 1, 0, -1, 0};
    bool lPin = false;
    jT = kj; iT = ki;
#pragma unroll 1
    for (int kp = 0; kp < 5 && !lPin; kp++) {
        jT = kj + dj[kp];
        iT = ki + di[kp];
        pt bl = load_f(geo, jT - 1, iT - 1, Nj, Ni);
        pt br = load_f(geo, jT - 1, iT, Nj, Ni);
        pt ur = load_f(geo, jT, iT, Nj, Ni);
        pt ul = load_f(geo, jT, iT - 1, Nj, Ni);
        lPin = inside_quad(zy, zx, bl, br, ur, ul);
    }
    return lPin;
}

__global__ void find_cells_kernel(int64_t n, int Nj, int Ni, const CellGeo *__restrict__ geo, const pt *__restrict__ yx,
                                  const int32_t *__restrict__ guess, int32_t *__restrict__ jiT, int8_t *__restrict__ found)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int kj = guess[2 * p], ki = guess[2 * p + 1];
    int jT = kj, iT = ki;
    bool ok = false;
    // candidates must keep every vertex index inside the arrays (the reference would raise IndexError)
    if (kj >= 1 && kj <= Nj - 2 && ki >= 1 && ki <= Ni - 2) ok = find_containing_cell(yx[p].y, yx[p].x, kj, ki, Nj, Ni, geo, jT, iT);
    jiT[2 * p] = jT;
    jiT[2 * p + 1] = iT;
    found[p] = ok ? 1 : 0;
}

// ---------------------------------------------------------------------------
// Haversine   reference sitrack/util.py:85-103   (R = 6360 km)
// ---------------------------------------------------------------------------
__device__ __forceinline__ double haversine(double plat, double plon, double cos_plat, double xlat, double xlon)
{
    const double to_rad = 3.141592653589793 / 180.;
    const double R = 6360.;
    double a1 = sin(0.5 * ((xlat - plat) * to_rad));
    double a2 = sin(0.5 * ((xlon - plon) * to_rad));
    double a3 = cos(xlat * to_rad) * cos_plat;
    return 2. * R * asin(sqrt(a1 * a1 + a3 * a2 * a2));
}

// SeedInit per-seed part (tracking.py:120-160): one workgroup per seed scans the
// whole T grid (exact first-minimum argmin like find_ji_of_min, locate.py:13-20),
// then lane 0 applies the acceptance rule of NearestPoint (locate.py:253-271) as
// called with max_itr = 10 and a 2-D resolkm, Survive and FindContainingCell.
__global__ __launch_bounds__(kBlock) void seed_init_bruteforce_kernel(
    int64_t nP, int Nj, int Ni, const ll *__restrict__ latlon, const pt *__restrict__ yx,
    const double *__restrict__ latT, const double *__restrict__ lonT, const double *__restrict__ resol,
    const double *__restrict__ sic, const int8_t *__restrict__ tmask, const CellGeo *__restrict__ geo,
    double rmin_conc, double rd_found_km, int max_itr, int32_t *__restrict__ jiT, int8_t *__restrict__ keep,
    int8_t *__restrict__ why)
{
    __shared__ double sd[kBlock / 64];
    __shared__ unsigned int sk[kBlock / 64];
    const int64_t p = blockIdx.x;
    if (p >= nP) return;
    const double plat = latlon[p].lat, plon = latlon[p].lon;
    const double to_rad = 3.141592653589793 / 180.;
    const double cos_plat = cos(plat * to_rad);
    const unsigned int n = (unsigned int)Nj * (unsigned int)Ni;
    double best = __builtin_inf();
    unsigned int kbest = 0xffffffffu;
    for (unsigned int k = threadIdx.x; k < n; k += kBlock) {
        double d = haversine(plat, plon, cos_plat, latT[k], lonT[k]);
        if (d < best) { best = d; kbest = k; }          // strided scan keeps the lowest k per lane
    }
    // wave reduction: smaller distance wins, ties -> lower flat index
    for (int off = 32; off > 0; off >>= 1) {
        double od = __shfl_down(best, off);
        unsigned int ok = __shfl_down(kbest, off);
        if (od < best || (od == best && ok < kbest)) { best = od; kbest = ok; }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sd[wave] = best; sk[wave] = kbest; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int w = 1; w < kBlock / 64; w++)
        if (sd[w] < best || (sd[w] == best && sk[w] < kbest)) { best = sd[w]; kbest = sk[w]; }

    int jy = (int)(kbest / (unsigned int)Ni), jx = (int)(kbest % (unsigned int)Ni);
    // acceptance loop of NearestPoint, whole-domain form (locate.py:253-266)
    double rfnd = rd_found_km;
    bool lfound = false;
    int igo = 0;
    while (!lfound && igo < max_itr) {
        igo = igo + 1;
        if (igo == 1 && resol) rfnd = 0.5 * resol[kbest];
        if (igo == 1) igo = 2;
        lfound = (best < rfnd);
        if (igo > 1 && !lfound) rfnd = 1.2 * rfnd;
    }
    int8_t kp = 1, wy = 0;
    int jT = 0, iT = 0;
    if (igo == max_itr) { kp = 0; wy = 1; }              // locate.py:271 -> (-1,-1), tracking.py:136-138
    if (kp && survive_kill<double>(jy, jx, Nj, Ni, tmask, sic, rmin_conc)) { kp = 0; wy = 2; }   // tracking.py:146-149
    if (kp && !find_containing_cell(yx[p].y, yx[p].x, jy, jx, Nj, Ni, geo, jT, iT)) { kp = 0; wy = 3; }   // :154-160
    jiT[2 * p] = kp ? jT : 0;
    jiT[2 * p + 1] = kp ? iT : 0;
    keep[p] = kp;
    if (why) why[p] = wy;
}

// ---------------------------------------------------------------------------
// Polar stereographic (WGS84), north-pole mode with lat_ts.
// Reference call sites: CartNPSkm2Geo1D / Geo2CartNPSkm1D, sitrack/util.py:394-429
// (si3_part_tracker.py:493).  The arithmetic there is cartopy -> PROJ `stere`
// (ellipsoidal); this follows that published algorithm (Snyder 1987, 21-33..21-40).
// ---------------------------------------------------------------------------
struct ProjParams { double e, akm1, a, lon0; };

__device__ __forceinline__ double nps_tsfn(double phi, double sinphi, double e)
{
    double es = e * sinphi;
    return tan(0.5 * (M_PI_2 - phi)) / pow((1.0 - es) / (1.0 + es), 0.5 * e);
}

__global__ void cart2geo_kernel(int64_t n, ProjParams pp, const pt *__restrict__ yx, ll *__restrict__ latlon)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double r2d = 180.0 / M_PI;
    double x = 1000. * yx[k].x / pp.a;
    double y = 1000. * yx[k].y / pp.a;
    double rho = hypot(x, y);
    y = -y;
    double tp = -rho / pp.akm1;
    double phi_l = M_PI_2 - 2. * atan(tp);
    const double halfpi = -M_PI_2, halfe = -.5 * pp.e;
    double phi = phi_l;
    bool ok = false;
    for (int i = 0; i < 8 && !ok; i++) {
        double sinphi = pp.e * sin(phi_l);
        phi = 2. * atan(tp * pow((1. + sinphi) / (1. - sinphi), halfe)) - halfpi;
        ok = fabs(phi_l - phi) < 1.e-10;
        phi_l = phi;
    }
    double lam = (x == 0. && y == 0.) ? 0. : atan2(x, y);
    double lon = lam * r2d + pp.lon0;
    if (lon < -180.0 || lon > 180.0) {
        lon = lon + 180.0;
        lon = lon - 360.0 * floor(lon / 360.0);
        lon = lon - 180.0;
    }
    const double nan = __builtin_nan("");
    ll o;
    o.lat = ok ? phi * r2d : nan;
    o.lon = ok ? lon : nan;
    latlon[k] = o;
}

__global__ void geo2cart_kernel(int64_t n, ProjParams pp, const ll *__restrict__ latlon, pt *__restrict__ yx)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double d2r = M_PI / 180.0;
    double phi = latlon[k].lat * d2r;
    double lam = (latlon[k].lon - pp.lon0) * d2r;
    double rho = (fabs(phi - M_PI_2) < 1e-15) ? 0.0 : pp.akm1 * nps_tsfn(phi, sin(phi), pp.e);
    double x = rho * sin(lam);
    double y = -rho * cos(lam);
    yx[k] = make_pt(pp.a * y / 1000., pp.a * x / 1000.);
}

}  // namespace sitrk
