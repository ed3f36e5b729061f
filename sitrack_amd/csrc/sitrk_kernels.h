// sitrk_kernels.h -- HIP kernels of libsitrk.so (gfx950, wave64).
//
// advect_step_kernel is the hot path: one buoy per lane, buoys kept sorted by
// host cell so that a wavefront's gathers fall on a few contiguous 48-byte cell
// records (F,U,V plane coordinates interleaved) and on short runs of the u/v
// slabs.  The work is HBM/latency bound (~100 fp64 flops vs ~140 B per
// particle-step): no LDS reuse to exploit beyond what L1/L2 already give for
// sorted buoys, no MFMA.
#pragma once
#include "sitrk_internal.h"

#pragma clang fp contract(off)

namespace sitrk {

static constexpr int kBlock = 256;

// ---------------------------------------------------------------------------
// grid re-layout: six (Nj,Ni) fp64 arrays -> one CellGeo record per cell
// ---------------------------------------------------------------------------
__global__ void build_geo_kernel(size_t n, const double *__restrict__ Yf, const double *__restrict__ Xf,
                                 const double *__restrict__ Yu, const double *__restrict__ Xu,
                                 const double *__restrict__ Yv, const double *__restrict__ Xv,
                                 CellGeo *__restrict__ geo)
{
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    CellGeo g;
    g.f = make_pt(Yf[k], Xf[k]);
    g.u = make_pt(Yu[k], Xu[k]);
    g.v = make_pt(Yv[k], Xv[k]);
    geo[k] = g;
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

__device__ __forceinline__ pt load_f(const CellGeo *__restrict__ geo, int j, int i, int Nj, int Ni)
{
    return geo[(size_t)pywrap(j, Nj) * Ni + pywrap(i, Ni)].f;
}

// CrossedEdge + NewHostCell + UpdtInd4NewCell   reference sitrack/tracking.py:182-305
// quad = [bl, br, ur, ul] of the current cell (jT,iT); returns the move (dj,di).
__device__ __forceinline__ void new_host_cell(pt P1, pt P2, pt bl, pt br, pt ur, pt ul, int jT, int iT, int Nj, int Ni,
                                              const CellGeo *__restrict__ geo, int &dj, int &di)
{
    // CrossedEdge (:189-200): first of bottom, right, upper, left hit; falls through to 4
    int kc;
    if (intersect2seg(P1, P2, bl, br)) kc = 1;
    else if (intersect2seg(P1, P2, br, ur)) kc = 2;
    else if (intersect2seg(P1, P2, ur, ul)) kc = 3;
    else kc = 4;
    // NewHostCell (:217-243): the grid line prolonging the crossed edge beyond each end
    int knhc = kc;
    if (kc == 1) {
        if (intersect2seg(P1, P2, bl, load_f(geo, jT - 2, iT - 1, Nj, Ni))) knhc = 5;
        else if (intersect2seg(P1, P2, br, load_f(geo, jT - 2, iT, Nj, Ni))) knhc = 6;
    } else if (kc == 2) {
        if (intersect2seg(P1, P2, br, load_f(geo, jT - 1, iT + 1, Nj, Ni))) knhc = 6;
        else if (intersect2seg(P1, P2, ur, load_f(geo, jT, iT + 1, Nj, Ni))) knhc = 7;
    } else if (kc == 3) {
        if (intersect2seg(P1, P2, ul, load_f(geo, jT + 1, iT - 1, Nj, Ni))) knhc = 8;
        else if (intersect2seg(P1, P2, ur, load_f(geo, jT + 1, iT, Nj, Ni))) knhc = 7;
    } else {
        if (intersect2seg(P1, P2, ul, load_f(geo, jT, iT - 2, Nj, Ni))) knhc = 8;
        else if (intersect2seg(P1, P2, bl, load_f(geo, jT - 1, iT - 2, Nj, Ni))) knhc = 5;
    }
    // UpdtInd4NewCell (:257-300)
    //            1   2   3   4   5   6   7   8
    dj = (knhc == 1 || knhc == 5 || knhc == 6) ? -1 : ((knhc == 3 || knhc == 7 || knhc == 8) ? 1 : 0);
    di = (knhc == 4 || knhc == 5 || knhc == 8) ? -1 : ((knhc == 2 || knhc == 6 || knhc == 7) ? 1 : 0);
}

// performance knobs (never change results)
enum : int {
    TUNE_XCD_REMAP = 1,    // give each XCD a contiguous chunk of the sorted buoys (neighbour rows hit the same L2)
    TUNE_NT_STATE = 2,     // stream pos/cell with non-temporal loads/stores: they are touched once per step
    TUNE_COMPACT = 4,      // workgroup compaction of the crossing path (advect_step_compact_kernel)
    TUNE_LOCATE_BRUTEFORCE = 8,   // SeedInit: whole-grid Haversine scan per seed (the reference's algorithm) instead of the sphere search
};

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

struct StepArgs {
    int64_t nP;
    int tune;
    int Nj, Ni;
    int jrec;
    double rdt, rmin_conc;
    const CellGeo *geo;
    const int8_t *tmask;
    const void *u, *v, *sic;
    pt *pos;
    int32_t *cell;
    int32_t *kill_rec;
    const int32_t *first, *last;
};

// ---------------------------------------------------------------------------
// One model record for every buoy -- body of si3_part_tracker.py:378-490.
//   UVS    : iUVstrategy (:37-40)   1 nearest U/V point, 0 cell mean
//   WINDOW : per-buoy first/last model record (2-D time mode, :264-318,380)
// ---------------------------------------------------------------------------
template <typename FT, int UVS, bool WINDOW>
__global__ __launch_bounds__(kBlock) void advect_step_kernel(StepArgs a)
{
    const unsigned blk = (a.tune & TUNE_XCD_REMAP) ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    int64_t p = (int64_t)blk * kBlock + threadIdx.x;
    if (p >= a.nP) return;
    const bool nt = (a.tune & TUNE_NT_STATE) != 0;
    int32_t c = nt ? __builtin_nontemporal_load(&a.cell[p]) : a.cell[p];
    if (c < 0) return;                                   // iAlive != 1
    if (WINDOW) {
        if (a.jrec < a.first[p] || a.jrec > a.last[p]) return;
    }
    const int Ni = a.Ni, Nj = a.Nj;
    const int jT = cell_j(c), iT = cell_i(c);
    const size_t k = (size_t)jT * Ni + iT;
    const FT *__restrict__ u = (const FT *)a.u;
    const FT *__restrict__ v = (const FT *)a.v;

    const pt P = nt ? load_pt_nt(&a.pos[p]) : a.pos[p];  // (ry, rx)
    // cell (jT,iT): F = upper-right vertex, U = right U-point, V = upper V-point
    const CellGeo g11 = a.geo[k];
    const pt F10 = a.geo[k - 1].f;                       // F[jT  ,iT-1]  upper-left
    const pt F01 = a.geo[k - Ni].f;                      // F[jT-1,iT  ]  bottom-right
    const pt F00 = a.geo[k - Ni - 1].f;                  // F[jT-1,iT-1]  bottom-left

    double zU, zV;
    if (UVS == 0) {                                      // :423-425
        zU = 0.5 * ((double)u[k] + (double)u[k - 1]);
        zV = 0.5 * ((double)v[k] + (double)v[k - Ni]);
    } else {                                             // :427-441
        const pt U10 = a.geo[k - 1].u;                   // U[jT,iT-1]
        const pt V01 = a.geo[k - Ni].v;                  // V[jT-1,iT]
        const double u1 = (double)u[k], u0 = (double)u[k - 1];
        const double v1 = (double)v[k], v0 = (double)v[k - Ni];
        const bool llum1 = intersect2seg(P, g11.f, V01, g11.v);
        const bool llvm1 = intersect2seg(P, g11.f, U10, g11.u);
        zU = llum1 ? u0 : u1;
        zV = llvm1 ? v0 : v1;
    }

    // forward Euler, one step per record (:452-458): km += (m/s * s) / 1000
    const double dx = zU * a.rdt;
    const double dy = zV * a.rdt;
    pt Pn;
    Pn.x = P.x + dx / 1000.;
    Pn.y = P.y + dy / 1000.;
    if (nt) store_pt_nt(&a.pos[p], Pn);                  // written before the kill test (:459-460)
    else a.pos[p] = Pn;

    // still inside the host cell? (:466) quad = [bl, br, ur, ul]
    if (!inside_quad(Pn.y, Pn.x, F00, F01, g11.f, F10)) {
        int dj, di;
        new_host_cell(P, Pn, F00, F01, g11.f, F10, jT, iT, Nj, Ni, a.geo, dj, di);
        const int jN = jT + dj, iN = iT + di;
        int32_t cn = pack_cell(jN, iN);
        if (survive_kill<FT>(jN, iN, Nj, Ni, a.tmask, (const FT *)a.sic, a.rmin_conc)) {   // :483-484
            cn |= SITRK_DEAD_BIT;
            a.kill_rec[p] = a.jrec;
        }
        a.cell[p] = cn;
    }
}

// ---------------------------------------------------------------------------
// Same record step with WORKGROUP COMPACTION OF THE CROSSING PATH.
//
// Only 3-16 % of buoys leave their cell in a step, but almost every wavefront has at least
// one that does, so in the one-pass kernel every wave pays the whole CrossedEdge /
// NewHostCell / Survive instruction stream (~45 % of its fp64 work; fp64 issues at 16
// lanes/clk, 4 cycles per wave64 instruction, and this kernel is as VALU-heavy as it is
// HBM-heavy) for a handful of active lanes.  Here the lanes that crossed append
// (buoy, old position, new position, cell) to an LDS queue (one LDS atomic per wave, ballot +
// popcount for the slot), and after one barrier the queue is drained by densely packed lanes:
// typically ONE wave of the four does the crossing work of the whole workgroup.
// Buoys are independent, so the queue order does not matter: results are bit-identical.
// ---------------------------------------------------------------------------
struct CrossItem {
    pt P, Pn;          // position before / after the Euler step
    int32_t cell;      // packed host cell before the move
    int32_t slot;      // index within the workgroup's 256 buoys
};

template <typename FT, int UVS, bool WINDOW>
__global__ __launch_bounds__(kBlock) void advect_step_compact_kernel(StepArgs a)
{
    __shared__ CrossItem queue[kBlock];
    __shared__ int qcount;
    if (threadIdx.x == 0) qcount = 0;
    __syncthreads();

    const unsigned blk = (a.tune & TUNE_XCD_REMAP) ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int64_t p0 = (int64_t)blk * kBlock;
    const int64_t p = p0 + threadIdx.x;
    const bool nt = (a.tune & TUNE_NT_STATE) != 0;
    const int Ni = a.Ni, Nj = a.Nj;

    bool active = p < a.nP;
    int32_t c = 0;
    if (active) {
        c = nt ? __builtin_nontemporal_load(&a.cell[p]) : a.cell[p];
        active = c >= 0;                                   // iAlive == 1
    }
    if (WINDOW) {
        if (active) active = (a.jrec >= a.first[p]) && (a.jrec <= a.last[p]);
    }

    bool crossed = false;
    pt P = make_pt(0., 0.), Pn = P;
    if (active) {
        const int jT = cell_j(c), iT = cell_i(c);
        const size_t k = (size_t)jT * Ni + iT;
        const FT *__restrict__ u = (const FT *)a.u;
        const FT *__restrict__ v = (const FT *)a.v;
        P = nt ? load_pt_nt(&a.pos[p]) : a.pos[p];
        const CellGeo g11 = a.geo[k];
        const pt F10 = a.geo[k - 1].f;
        const pt F01 = a.geo[k - Ni].f;
        const pt F00 = a.geo[k - Ni - 1].f;
        double zU, zV;
        if (UVS == 0) {                                      // :423-425
            zU = 0.5 * ((double)u[k] + (double)u[k - 1]);
            zV = 0.5 * ((double)v[k] + (double)v[k - Ni]);
        } else {                                             // :427-441
            const pt U10 = a.geo[k - 1].u;
            const pt V01 = a.geo[k - Ni].v;
            const double u1 = (double)u[k], u0 = (double)u[k - 1];
            const double v1 = (double)v[k], v0 = (double)v[k - Ni];
            const bool llum1 = intersect2seg(P, g11.f, V01, g11.v);
            const bool llvm1 = intersect2seg(P, g11.f, U10, g11.u);
            zU = llum1 ? u0 : u1;
            zV = llvm1 ? v0 : v1;
        }
        const double dx = zU * a.rdt;                        // :452-458
        const double dy = zV * a.rdt;
        Pn.x = P.x + dx / 1000.;
        Pn.y = P.y + dy / 1000.;
        if (nt) store_pt_nt(&a.pos[p], Pn);                  // :459-460
        else a.pos[p] = Pn;
        crossed = !inside_quad(Pn.y, Pn.x, F00, F01, g11.f, F10);   // :466
    }

    // ---- enqueue the lanes that left their cell
    const unsigned long long m = __ballot(crossed);
    if (m) {
        const int lane = threadIdx.x & 63;
        int base = 0;
        if (lane == 0) base = atomicAdd(&qcount, __popcll(m));
        base = __shfl(base, 0);
        if (crossed) {
            const int at = base + __popcll(m & ((1ull << lane) - 1ull));
            CrossItem it;
            it.P = P; it.Pn = Pn; it.cell = c; it.slot = (int32_t)threadIdx.x;
            queue[at] = it;
        }
    }
    __syncthreads();

    // ---- drain: CrossedEdge / NewHostCell / UpdtInd4NewCell / Survive (:474-484) on dense lanes
    const int n = qcount;
    for (int t = threadIdx.x; t < n; t += kBlock) {
        const CrossItem it = queue[t];
        const int jT = cell_j(it.cell), iT = cell_i(it.cell);
        const size_t k = (size_t)jT * Ni + iT;
        const pt F11 = a.geo[k].f, F10 = a.geo[k - 1].f, F01 = a.geo[k - Ni].f, F00 = a.geo[k - Ni - 1].f;
        int dj, di;
        new_host_cell(it.P, it.Pn, F00, F01, F11, F10, jT, iT, Nj, Ni, a.geo, dj, di);
        const int jN = jT + dj, iN = iT + di;
        int32_t cn = pack_cell(jN, iN);
        const int64_t q = p0 + it.slot;
        if (survive_kill<FT>(jN, iN, Nj, Ni, a.tmask, (const FT *)a.sic, a.rmin_conc)) {
            cn |= SITRK_DEAD_BIT;
            a.kill_rec[q] = a.jrec;
        }
        a.cell[q] = cn;
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

// key = row-major cell index; dead buoys last
__global__ void make_keys_kernel(int64_t n, int Ni, uint32_t dead_key, const int32_t *__restrict__ cell,
                                 uint32_t *__restrict__ keys, int32_t *__restrict__ vals)
{
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    int32_t c = cell[s];
    keys[s] = (c < 0) ? dead_key : (uint32_t)cell_j(c) * (uint32_t)Ni + (uint32_t)cell_i(c);
    vals[s] = (int32_t)s;
}

__global__ void permute_state_kernel(int64_t n, const int32_t *__restrict__ src, BuoyState in, BuoyState out, bool windowed)
{
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    int32_t q = src[s];
    out.pos[s] = in.pos[q];
    out.cell[s] = in.cell[q];
    out.kill_rec[s] = in.kill_rec[q];
    out.perm[s] = in.perm[q];
    if (windowed) {
        out.first[s] = in.first[q];
        out.last[s] = in.last[q];
    }
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
    if (windowed) in_window = (jrec >= st.first[s]) && (jrec <= st.last[s]);
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
