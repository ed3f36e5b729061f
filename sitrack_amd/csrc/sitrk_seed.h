// sitrk_seed.h -- idealised seeding on the model grid (sitrk_nemo_seed): which points of the sub-sampled mesh carry a
// seed, their ordered compaction and their projection, on the device.
//
// What it computes is specified by reference sitrack/tracking.py:365-442 (`nemoSeed`) and util.py:394-410
// (`Geo2CartNPSkm1D`); restated here from that behaviour:
//   * the sub-sampled mesh takes every khss-th row and column of the model arrays: point (a,b) is model point (a*khss, b*khss);
//   * a T-seed sits on sub-sampled point (a,b) iff  m(a,b) == 1  with  m = tmask * restriction (int8 product; restriction
//     optional), reset to 0 where the T-latitude is below 55 degrees or the ice concentration below 0.9 (comparisons that a
//     NaN does not satisfy, so a NaN keeps the point, like numpy's `<`);
//   * optionally an F-seed sits on interior sub-sampled point (a,b), 1 <= a <= Njs-2, 1 <= b <= Nis-2, iff the four
//     neighbours m(a+1,b), m(a,b+1), m(a-1,b), m(a,b-1) sum to 4..7 (the reference stores sum/4 into an int8 and asks for 1);
//     its coordinates are the sub-sampled F-point arrays at (a,b);
//   * output order: all T-seeds in C order of the sub-sampled mesh, then all F-seeds in C order; each as (lat, lon) and as
//     its polar-stereographic (y, x) in km.
// Three kernels: flags counted per 1024-element block, one workgroup scans the block counts, and the last kernel ranks
// each flagged element inside its block (ballot + popcount), writes its coordinates and projects them.
#pragma once
#include "sitrk_kernels.h"

namespace sitrk {

static constexpr int kSeedBlock = 1024;

struct SeedArgs {
    int Nj, Ni, khss, Njs, Nis;
    int with_f;
    const int8_t *tmask, *rmask;                 // rmask may be null
    const double *latT, *lonT, *sic, *latF, *lonF;
};

// m(a,b) of the header comment
__device__ __forceinline__ int seed_mask(const SeedArgs &s, int a, int b)
{
    const size_t k = (size_t)(a * s.khss) * s.Ni + (size_t)(b * s.khss);
    int8_t m = s.tmask[k];
    if (s.rmask) m = (int8_t)(m * s.rmask[k]);
    if (s.latT[k] < 55.) m = 0;
    if (s.sic[k] < 0.9) m = 0;
    return (int)m;
}

// The flag sequence: blocks [0, nblk_t) hold the T-points of the sub-sampled mesh in C order, blocks [nblk_t, 2 nblk_t) its
// F-points (each part padded to whole blocks, so that the number of T-seeds is a prefix sum at a block boundary).
__device__ __forceinline__ bool seed_flag(const SeedArgs &s, bool isF, int64_t q)
{
    const int64_t ns = (int64_t)s.Njs * s.Nis;
    if (q >= ns) return false;
    if (!isF) return seed_mask(s, (int)(q / s.Nis), (int)(q % s.Nis)) == 1;
    if (!s.with_f) return false;
    const int a = (int)(q / s.Nis), b = (int)(q % s.Nis);
    if (a < 1 || a > s.Njs - 2 || b < 1 || b > s.Nis - 2) return false;
    const int sum = seed_mask(s, a + 1, b) + seed_mask(s, a, b + 1) + seed_mask(s, a - 1, b) + seed_mask(s, a, b - 1);
    return sum >= 4 && sum < 8;
}

__global__ __launch_bounds__(kSeedBlock) void seed_count_kernel(SeedArgs s, int64_t nblk_t, unsigned *__restrict__ block_count)
{
    __shared__ unsigned sw[kSeedBlock / 64];
    const bool isF = (int64_t)blockIdx.x >= nblk_t;
    const int64_t q = ((int64_t)blockIdx.x - (isF ? nblk_t : 0)) * kSeedBlock + threadIdx.x;
    const bool f = seed_flag(s, isF, q);
    const unsigned long long bal = __ballot(f);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = (unsigned)__popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
        for (int w = 0; w < kSeedBlock / 64; w++) t += sw[w];
        block_count[blockIdx.x] = t;
    }
}

// exclusive scan of the block counts by ONE workgroup (a few thousand blocks at most: 4096^2 points -> 32 768 blocks);
// offsets are 64-bit, total[0] = number of T-seeds, total[1] = number of F-seeds
__global__ __launch_bounds__(kSeedBlock) void seed_scan_kernel(int64_t nblk, int64_t nblk_t, const unsigned *__restrict__ block_count,
                                                               int64_t *__restrict__ block_off, int64_t *__restrict__ total)
{
    __shared__ int64_t s_part[kSeedBlock];
    __shared__ int64_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nblk; base += kSeedBlock) {
        const int64_t k = base + threadIdx.x;
        const int64_t v = k < nblk ? (int64_t)block_count[k] : 0;
        s_part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < kSeedBlock; off <<= 1) {         // Hillis-Steele inclusive scan of this chunk
            int64_t t = threadIdx.x >= off ? s_part[threadIdx.x - off] : 0;
            __syncthreads();
            s_part[threadIdx.x] += t;
            __syncthreads();
        }
        const int64_t carry = s_carry;
        if (k < nblk) {
            block_off[k] = carry + s_part[threadIdx.x] - v;
            if (k == nblk_t - 1) total[0] = carry + s_part[threadIdx.x];          // everything up to the last T block
        }
        __syncthreads();
        if (threadIdx.x == kSeedBlock - 1) s_carry = carry + s_part[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) total[1] = s_carry - total[0];
}

__global__ __launch_bounds__(kSeedBlock) void seed_emit_kernel(SeedArgs s, int64_t nblk_t, const int64_t *__restrict__ block_off, ProjParams pp,
                                                               int64_t capacity, ll *__restrict__ latlon, pt *__restrict__ yx)
{
    __shared__ unsigned sw[kSeedBlock / 64];
    const bool isF = (int64_t)blockIdx.x >= nblk_t;
    const int64_t q = ((int64_t)blockIdx.x - (isF ? nblk_t : 0)) * kSeedBlock + threadIdx.x;
    const bool f = seed_flag(s, isF, q);
    const unsigned long long bal = __ballot(f);
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sw[wave] = (unsigned)__popcll(bal);
    __syncthreads();
    if (!f) return;
    unsigned rank = (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
    for (unsigned w = 0; w < wave; w++) rank += sw[w];
    const int64_t o = block_off[blockIdx.x] + rank;
    if (o >= capacity) return;
    const size_t k = (size_t)((q / s.Nis) * s.khss) * s.Ni + (size_t)((q % s.Nis) * s.khss);
    ll g;
    g.lat = isF ? s.latF[k] : s.latT[k];
    g.lon = isF ? s.lonF[k] : s.lonT[k];
    latlon[o] = g;
    if (yx) {                                                    // Geo2CartNPSkm1D, same arithmetic as geo2cart_kernel
        const double d2r = M_PI / 180.0;
        const double phi = g.lat * d2r, lam = (g.lon - pp.lon0) * d2r;
        const double rho = (fabs(phi - M_PI_2) < 1e-15) ? 0.0 : pp.akm1 * nps_tsfn(phi, sin(phi), pp.e);
        yx[o] = make_pt(pp.a * (-rho * cos(lam)) / 1000., pp.a * (rho * sin(lam)) / 1000.);
    }
}

}  // namespace sitrk
