// sitrk_locate.h -- exact nearest-T-point search for SeedInit at scale (gfx950, wave64).
//
// The reference finds the nearest T-point of every seed by evaluating Haversine against
// the WHOLE grid and taking the first minimum (sitrack/locate.py:257-258, util.py:85-103):
// O(nP*Nj*Ni), 1.7e14 evaluations at 1e7 seeds on 4096^2.  Here the same answer comes from a
// branch-and-bound over bounding spheres of the structured mesh:
//
//   * every T-point becomes a unit vector; the Haversine argument
//     sin^2(dphi/2) + cos(phi)cos(phi_p)sin^2(dlam/2) equals |p - q|^2 / 4 (chord), so
//     the Euclidean 3-D metric orders points exactly like the great-circle distance and
//     obeys the triangle inequality used for pruning;
//   * blocks of 16x16 mesh points and superblocks of SBFxSBF blocks carry a bounding
//     sphere (centre, radius) with conservative rounding slack;
//   * one WAVEFRONT per seed: the 64 lanes evaluate 64 sphere bounds / 64 mesh points at a
//     time, `__ballot` selects the spheres that can still contain a closer point and
//     shuffles reduce the minimum;
//   * pass 1 yields the minimum chord^2; pass 2 re-visits the (1-4) blocks within
//     (1 + 1e-9) of it and evaluates the reference's Haversine formula, in its operation order,
//     on those candidates only, keeping the smallest distance and, on exact ties, the lowest
//     flat index -- i.e. what argmin over the whole distance field returns, because any point
//     outside that margin is farther by ~1e-9 relative while the formula's rounding error is
//     ~1e-15.
#pragma once
#include "sitrk_kernels.h"

#pragma clang fp contract(off)

namespace sitrk {

static constexpr int kLB = 16;             // mesh points per block edge

struct __attribute__((aligned(16))) Sphere { double cx, cy, cz, r; };

__device__ __forceinline__ void unit_vec(double lat, double lon, double &x, double &y, double &z)
{
    const double to_rad = 3.141592653589793 / 180.;
    double sl, cl, sp, cp;
    sincos(lon * to_rad, &sl, &cl);
    sincos(lat * to_rad, &sp, &cp);
    x = cp * cl; y = cp * sl; z = sp;
}

__global__ void unitvec_kernel(size_t n, const double *__restrict__ lat, const double *__restrict__ lon,
                               double *__restrict__ ux, double *__restrict__ uy, double *__restrict__ uz)
{
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double x, y, z;
    unit_vec(lat[k], lon[k], x, y, z);
    ux[k] = x; uy[k] = y; uz[k] = z;
}

__device__ __forceinline__ double wave_sum(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return __shfl(v, 0);
}
__device__ __forceinline__ double wave_max(double v)
{
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off));
    return __shfl(v, 0);
}
__device__ __forceinline__ double wave_min(double v)
{
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off));
    return __shfl(v, 0);
}

// block reduction helpers over kBlock = 256 threads (4 waves)
__device__ __forceinline__ double block_sum(double v, double *sh)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ double block_max(double v, double *sh)
{
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}

// one workgroup per 16x16 block of mesh points -> bounding sphere
__global__ __launch_bounds__(kBlock) void block_sphere_kernel(int Nj, int Ni, int nbi, const double *__restrict__ ux,
                                                              const double *__restrict__ uy, const double *__restrict__ uz,
                                                              Sphere *__restrict__ blk)
{
    __shared__ double sh[4];
    const int b = blockIdx.x, bj = b / nbi, bi = b % nbi;
    const int j = bj * kLB + (int)(threadIdx.x >> 4), i = bi * kLB + (int)(threadIdx.x & 15);
    const bool in = (j < Nj) && (i < Ni);
    double x = 0, y = 0, z = 0;
    if (in) { size_t k = (size_t)j * Ni + i; x = ux[k]; y = uy[k]; z = uz[k]; }
    double sx = block_sum(x, sh), sy = block_sum(y, sh), sz = block_sum(z, sh);
    double nrm = sqrt(sx * sx + sy * sy + sz * sz);
    double cx, cy, cz;
    if (nrm > 1e-9) { cx = sx / nrm; cy = sy / nrm; cz = sz / nrm; }
    else {                                   // degenerate spread: fall back on the block's first point
        size_t k0 = (size_t)(bj * kLB) * Ni + bi * kLB;
        cx = ux[k0]; cy = uy[k0]; cz = uz[k0];
    }
    double d2 = in ? ((x - cx) * (x - cx) + (y - cy) * (y - cy) + (z - cz) * (z - cz)) : 0.0;
    double r = sqrt(block_max(d2, sh)) * (1.0 + 1e-12) + 1e-14;
    if (threadIdx.x == 0) { Sphere s; s.cx = cx; s.cy = cy; s.cz = cz; s.r = r; blk[b] = s; }
}

// one workgroup per superblock (sbf x sbf blocks, sbf <= 16)
__global__ __launch_bounds__(kBlock) void superblock_sphere_kernel(int nbj, int nbi, int sbf, int nsi, const Sphere *__restrict__ blk,
                                                                   Sphere *__restrict__ sblk)
{
    __shared__ double sh[4];
    const int s = blockIdx.x, sj = s / nsi, si = s % nsi;
    const int tj = (int)threadIdx.x / sbf, ti = (int)threadIdx.x % sbf;
    const int bj = sj * sbf + tj, bi = si * sbf + ti;
    const bool in = (tj < sbf) && (bj < nbj) && (bi < nbi);
    Sphere m; m.cx = m.cy = m.cz = m.r = 0;
    if (in) m = blk[bj * nbi + bi];
    double sx = block_sum(in ? m.cx : 0., sh), sy = block_sum(in ? m.cy : 0., sh), sz = block_sum(in ? m.cz : 0., sh);
    double nrm = sqrt(sx * sx + sy * sy + sz * sz);
    Sphere f = blk[(sj * sbf) * nbi + si * sbf];
    double cx = f.cx, cy = f.cy, cz = f.cz;
    if (nrm > 1e-9) { cx = sx / nrm; cy = sy / nrm; cz = sz / nrm; }
    double reach = in ? (sqrt((m.cx - cx) * (m.cx - cx) + (m.cy - cy) * (m.cy - cy) + (m.cz - cz) * (m.cz - cz)) + m.r) : 0.0;
    double r = block_max(reach, sh) * (1.0 + 1e-12) + 1e-14;
    if (threadIdx.x == 0) { Sphere o; o.cx = cx; o.cy = cy; o.cz = cz; o.r = r; sblk[s] = o; }
}

struct SearchArgs {
    int64_t nP;
    int Nj, Ni, nbj, nbi, sbf, nsj, nsi;
    const ll *latlon;
    const double *ux, *uy, *uz;
    const Sphere *blk, *sblk;
    const double *latT, *lonT;
    uint32_t *kbest;
    double *dbest;
    double far2;                        // chord^2 beyond which no T-point can pass NearestPoint's acceptance test
};

// squared lower bound of the chord from s to any point of the sphere (0 when s may be inside)
__device__ __forceinline__ double sphere_lb2(double sx, double sy, double sz, const Sphere &q)
{
    double d = sqrt((sx - q.cx) * (sx - q.cx) + (sy - q.cy) * (sy - q.cy) + (sz - q.cz) * (sz - q.cz));
    double lb = d - q.r - 1e-14;
    return lb > 0.0 ? lb * lb * (1.0 - 1e-12) : 0.0;
}
// squared upper bound of the chord from s to the FARTHEST point of the sphere
__device__ __forceinline__ double sphere_ub2(double sx, double sy, double sz, const Sphere &q)
{
    double d = sqrt((sx - q.cx) * (sx - q.cx) + (sy - q.cy) * (sy - q.cy) + (sz - q.cz) * (sz - q.cz)) + q.r;
    return d * d * (1.0 + 1e-12) + 1e-28;
}

// visits every mesh point whose chord^2 to the seed may be <= lim2, wave-cooperatively.
//   MODE 0: lim2 shrinks to the running minimum chord^2 (returned)
//   MODE 1: lim2 fixed; evaluates the reference Haversine on points with chord^2 <= lim2 and keeps the
//           per-lane (distance, lowest flat index) minimum in (hd, hk)
template <int MODE>
__device__ __forceinline__ double scan_mesh(const SearchArgs &a, double sx, double sy, double sz, double lim2, int lane,
                                            double plat, double plon, double cos_plat, double &hd, uint32_t &hk)
{
    const int nsb = a.nsj * a.nsi;
    for (int s0 = 0; s0 < nsb; s0 += 64) {
        const int s = s0 + lane;
        double lb2 = __builtin_inf();
        if (s < nsb) lb2 = sphere_lb2(sx, sy, sz, a.sblk[s]);
        unsigned long long ms = __ballot(lb2 <= lim2);
        while (ms) {
            const int sl = __ffsll((long long)ms) - 1;
            ms &= ms - 1;
            if (__shfl(lb2, sl) > lim2) continue;            // the limit may have shrunk since the ballot
            const int sb = s0 + sl, sj = sb / a.nsi, si = sb % a.nsi;
            // blocks of this superblock: sbf*sbf <= 256, 64 per round
            const int nb_in = a.sbf * a.sbf;
            for (int b0 = 0; b0 < nb_in; b0 += 64) {
                const int t = b0 + lane, tj = t / a.sbf, ti = t % a.sbf;
                const int bj = sj * a.sbf + tj, bi = si * a.sbf + ti;
                double bl2 = __builtin_inf();
                if (t < nb_in && bj < a.nbj && bi < a.nbi) bl2 = sphere_lb2(sx, sy, sz, a.blk[bj * a.nbi + bi]);
                unsigned long long mb = __ballot(bl2 <= lim2);
                while (mb) {
                    const int bl = __ffsll((long long)mb) - 1;
                    mb &= mb - 1;
                    if (__shfl(bl2, bl) > lim2) continue;
                    const int tt = b0 + bl, bjj = sj * a.sbf + tt / a.sbf, bii = si * a.sbf + tt % a.sbf;
                    // 16x16 points, 4 rows of 16 per round
                    double m2 = __builtin_inf();
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int j = bjj * kLB + r * 4 + (lane >> 4), i = bii * kLB + (lane & 15);
                        if (j < a.Nj && i < a.Ni) {
                            const size_t k = (size_t)j * a.Ni + i;
                            const double dx = sx - a.ux[k], dy = sy - a.uy[k], dz = sz - a.uz[k];
                            const double c2 = dx * dx + dy * dy + dz * dz;
                            if (MODE == 0) {
                                m2 = fmin(m2, c2);
                            } else if (c2 <= lim2) {
                                const double d = haversine(plat, plon, cos_plat, a.latT[k], a.lonT[k]);
                                if (d < hd || (d == hd && (uint32_t)k < hk)) { hd = d; hk = (uint32_t)k; }
                            }
                        }
                    }
                    if (MODE == 0) lim2 = fmin(lim2, wave_min(m2));
                }
            }
        }
    }
    return lim2;
}

// one wavefront per seed
__global__ __launch_bounds__(kBlock) void seed_search_kernel(SearchArgs a)
{
    const int64_t p = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (p >= a.nP) return;                                   // wave-uniform
    const int lane = threadIdx.x & 63;
    const double plat = a.latlon[p].lat, plon = a.latlon[p].lon;
    const double to_rad = 3.141592653589793 / 180.;
    const double cos_plat = cos(plat * to_rad);
    double sx, sy, sz;
    unit_vec(plat, plon, sx, sy, sz);

    // upper bound of the minimum chord^2: the farthest point of the most promising superblock,
    // then of its most promising block
    const int nsb = a.nsj * a.nsi;
    double ub2 = __builtin_inf(), lbmin2 = __builtin_inf();
    int sbest = 0;
    for (int s = lane; s < nsb; s += 64) {
        const Sphere q = a.sblk[s];
        double u = sphere_ub2(sx, sy, sz, q);
        if (u < ub2) { ub2 = u; sbest = s; }
        lbmin2 = fmin(lbmin2, sphere_lb2(sx, sy, sz, q));
    }
    // A seed farther from every superblock than the largest distance NearestPoint can accept is rejected whatever its
    // nearest T-point is (locate.py:264-271): skip the search -- it would have to visit most of the mesh, since from far
    // away all T-points are about equally distant.  `far2` carries a 2 % margin; closer seeds take the exact path.
    if (wave_min(lbmin2) > a.far2) {
        if (lane == 0) { a.kbest[p] = 0xffffffffu; a.dbest[p] = __builtin_inf(); }
        return;
    }
    {
        const double w = wave_min(ub2);
        unsigned long long mm = __ballot(ub2 == w);
        sbest = __shfl(sbest, __ffsll((long long)mm) - 1);
        ub2 = w;
    }
    {
        const int sj = sbest / a.nsi, si = sbest % a.nsi, nb_in = a.sbf * a.sbf;
        double bu2 = __builtin_inf();
        for (int t = lane; t < nb_in; t += 64) {
            const int bj = sj * a.sbf + t / a.sbf, bi = si * a.sbf + t % a.sbf;
            if (bj < a.nbj && bi < a.nbi) bu2 = fmin(bu2, sphere_ub2(sx, sy, sz, a.blk[bj * a.nbi + bi]));
        }
        ub2 = fmin(ub2, wave_min(bu2));
    }

    double hd = __builtin_inf();
    uint32_t hk = 0xffffffffu;
    // pass 1: exact minimum chord^2
    const double min2 = scan_mesh<0>(a, sx, sy, sz, ub2, lane, plat, plon, cos_plat, hd, hk);
    // pass 2: reference Haversine on every point within the rounding margin of the minimum
    // margin >> rounding error of the chord (abs ~7e-16*d on d^2) and of the Haversine formula (~1e-15 relative)
    const double thr2 = min2 * (1.0 + 1e-9) + 4e-15 * sqrt(min2) + 1e-28;
    scan_mesh<1>(a, sx, sy, sz, thr2, lane, plat, plon, cos_plat, hd, hk);
    for (int off = 32; off > 0; off >>= 1) {
        const double od = __shfl_down(hd, off);
        const uint32_t ok = __shfl_down(hk, off);
        if (od < hd || (od == hd && ok < hk)) { hd = od; hk = ok; }
    }
    if (lane == 0) { a.kbest[p] = hk; a.dbest[p] = hd; }
}

// second half of SeedInit per seed, one lane each: NearestPoint's acceptance loop
// (locate.py:253-271 as called with max_itr = 10 and a 2-D resolkm), Survive on the nearest T-point
// (tracking.py:146-149), FindContainingCell (:154-160)
__global__ void seed_finish_kernel(int64_t nP, int Nj, int Ni, const uint32_t *__restrict__ kbest, const double *__restrict__ dbest,
                                   const pt *__restrict__ yx, const double *__restrict__ resol, const double *__restrict__ sic,
                                   const int8_t *__restrict__ tmask, const CellGeo *__restrict__ geo, double rmin_conc,
                                   double rd_found_km, int max_itr, int32_t *__restrict__ jiT, int8_t *__restrict__ keep,
                                   int8_t *__restrict__ why)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nP) return;
    const uint32_t k = kbest[p];
    const double best = dbest[p];
    if (k == 0xffffffffu) {                              // farther than any acceptance radius (seed_search_kernel)
        jiT[2 * p] = 0; jiT[2 * p + 1] = 0;
        keep[p] = 0;
        if (why) why[p] = 1;
        return;
    }
    const int jy = (int)(k / (uint32_t)Ni), jx = (int)(k % (uint32_t)Ni);
    double rfnd = rd_found_km;
    bool lfound = false;
    int igo = 0;
    while (!lfound && igo < max_itr) {
        igo = igo + 1;
        if (igo == 1 && resol) rfnd = 0.5 * resol[k];
        if (igo == 1) igo = 2;
        lfound = (best < rfnd);
        if (igo > 1 && !lfound) rfnd = 1.2 * rfnd;
    }
    int8_t kp = 1, wy = 0;
    int jT = 0, iT = 0;
    if (igo == max_itr) { kp = 0; wy = 1; }
    if (kp && survive_kill<double>(jy, jx, Nj, Ni, tmask, sic, rmin_conc)) { kp = 0; wy = 2; }
    if (kp && !find_containing_cell(yx[p].y, yx[p].x, jy, jx, Nj, Ni, geo, jT, iT)) { kp = 0; wy = 3; }
    jiT[2 * p] = kp ? jT : 0;
    jiT[2 * p + 1] = kp ? iT : 0;
    keep[p] = kp;
    if (why) why[p] = wy;
}

// NearestPoint alone (locate.py:222-276, whole-domain form): nearest T-point of every seed, or (-1,-1) when the
// acceptance loop gives up; same loop as in seed_finish_kernel
__global__ void nearest_finish_kernel(int64_t nP, int Ni, const uint32_t *__restrict__ kbest, const double *__restrict__ dbest,
                                      const double *__restrict__ resol, double rd_found_km, int max_itr,
                                      int32_t *__restrict__ ji, double *__restrict__ dmin)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nP) return;
    const uint32_t k = kbest[p];
    int jy = -1, jx = -1;
    double best = __builtin_inf();
    if (k != 0xffffffffu) {
        best = dbest[p];
        double rfnd = rd_found_km;
        bool lfound = false;
        int igo = 0;
        while (!lfound && igo < max_itr) {
            igo = igo + 1;
            if (igo == 1 && resol) rfnd = 0.5 * resol[k];
            if (igo == 1) igo = 2;
            lfound = (best < rfnd);
            if (igo > 1 && !lfound) rfnd = 1.2 * rfnd;
        }
        if (igo != max_itr) { jy = (int)(k / (uint32_t)Ni); jx = (int)(k % (uint32_t)Ni); }
    }
    ji[2 * p] = jy; ji[2 * p + 1] = jx;
    if (dmin) dmin[p] = best;
}

// Haversine (util.py:85-103) element by element
__global__ void eval_haversine_kernel(int64_t n, const double *__restrict__ plat, const double *__restrict__ plon,
                                      const double *__restrict__ xlat, const double *__restrict__ xlon, double *__restrict__ out)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double to_rad = 3.141592653589793 / 180.;
    out[k] = haversine(plat[k], plon[k], cos(plat[k] * to_rad), xlat[k], xlon[k]);
}

}  // namespace sitrk
