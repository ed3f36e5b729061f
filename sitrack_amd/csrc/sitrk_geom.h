// sitrk_geom.h -- fp64 device predicates of the advection hot path (gfx950).
//
// Every comparison below decides a host cell, hence a velocity, hence the whole
// trajectory, so the arithmetic is IEEE double in the reference's operation
// order with NO fused multiply-add (Python/numpy never fuse).  The build passes
// -ffp-contract=off; the pragma repeats it for anyone compiling this header alone.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace sitrk {

// points are (y,x) pairs in the reference's array order [y,x]; 16-byte aligned so
// a point is one dwordx4 load.  `ll` is the same for [lat,lon].
struct __attribute__((aligned(16))) pt { double y, x; };
struct __attribute__((aligned(16))) ll { double lat, lon; };
__host__ __device__ __forceinline__ pt make_pt(double y, double x) { pt p; p.y = y; p.x = x; return p; }

// one 48-byte record per T-cell: F-, U-, V-point plane coordinates [km]
struct __attribute__((aligned(16))) CellGeo {
    pt f, u, v;
};

// _ccw_   reference sitrack/tracking.py:44-49
__device__ __forceinline__ bool ccw(pt A, pt B, pt C)
{
    double lhs = (C.y - A.y) * (B.x - A.x);
    double rhs = (B.y - A.y) * (C.x - A.x);
    return lhs > rhs;
}

// intersect2Seg   reference sitrack/tracking.py:51-58
__device__ __forceinline__ bool intersect2seg(pt A, pt B, pt C, pt D)
{
    return (ccw(A, C, D) != ccw(B, C, D)) && (ccw(A, B, C) != ccw(A, B, D));
}

// EXTRA, not in the reference (iUVstrategy = 2): linear interpolation of a C-grid component between the two points
// that carry it in the cell -- u between the left and right U-points, v between the lower and upper V-points -- at the
// buoy's projection on the segment joining them, clamped to the segment.  The reference's rules are piecewise constant
// (nearest point, or cell mean); this is the "interpolated" velocity BASELINE.json's north_star speaks of.
__device__ __forceinline__ double lerp_on_segment(pt P, pt A, pt B, double fa, double fb)
{
    const double dy = B.y - A.y, dx = B.x - A.x;
    const double den = dy * dy + dx * dx;
    double s = ((P.y - A.y) * dy + (P.x - A.x) * dx) / den;
    s = (s < 0.0) ? 0.0 : ((s > 1.0) ? 1.0 : s);
    return (1.0 - s) * fa + s * fb;
}

// one edge visit of the ray cast of IsInsideQuadrangle (locate.py:66-76);
// `xints` is carried between visits when the edge is horizontal (:63,71-73)
__device__ __forceinline__ void ray_edge(double y, double x, pt z1, pt z2, double &xints, bool &inside)
{
    // Python min()/max() of two finite numbers; only compared against, so the sign of a zero does not matter
    double ymin = fmin(z1.y, z2.y);
    double ymax = fmax(z1.y, z2.y);
    double xmax = fmax(z1.x, z2.x);
    if ((y > ymin) & (y <= ymax) & (x <= xmax)) {
        if (z1.y != z2.y) xints = (y - z1.y) * (z2.x - z1.x) / (z2.y - z1.y) + z1.x;
        if ((z1.x == z2.x) | (x <= xints)) inside = !inside;
    }
}

// IsInsideQuadrangle   reference sitrack/locate.py:49-78
// The reference visits n+1 = 5 edges starting with the degenerate quad[0]->quad[0],
// which can never toggle (y > q0y and y <= q0y), so four visits remain.
__device__ __forceinline__ bool inside_quad(double y, double x, pt q0, pt q1, pt q2, pt q3)
{
    bool inside = false;
    double xints = 0.0;
    ray_edge(y, x, q0, q1, xints, inside);
    ray_edge(y, x, q1, q2, xints, inside);
    ray_edge(y, x, q2, q3, xints, inside);
    ray_edge(y, x, q3, q0, xints, inside);
    return inside;
}

// Python-style index: a negative index wraps, as numpy does for the reference
// (e.g. pY[jbl-1,ibl] with jbl = 0, tracking.py:219).  Indices >= n cannot occur
// for the cells the library accepts (1 <= jT <= Nj-2, 1 <= iT <= Ni-2).
__device__ __forceinline__ int pywrap(int k, int n) { return k < 0 ? k + n : k; }

// packed host cell: bit 31 = dead, bits 30..16 = jT, bits 15..0 = iT
__host__ __device__ __forceinline__ int32_t pack_cell(int jT, int iT) { return (int32_t)((jT << 16) | iT); }
__host__ __device__ __forceinline__ int cell_j(int32_t c) { return (c >> 16) & 0x7fff; }
__host__ __device__ __forceinline__ int cell_i(int32_t c) { return c & 0xffff; }
#define SITRK_DEAD_BIT ((int32_t)0x80000000)

// performance knobs (never change results)
enum : int {
    TUNE_XCD_REMAP = 1,            // give each XCD a contiguous chunk of the sorted buoys
    TUNE_NT_STATE = 2,             // non-temporal loads/stores for the once-per-step pos/cell streams
    TUNE_LOCATE_BRUTEFORCE = 8,    // SeedInit: whole-grid Haversine scan per seed (the reference's algorithm)
    TUNE_DIAG_MEMONLY = 16,        // ablation kernels, diagnostic builds only (make DIAG=1)
    TUNE_DIAG_NOCROSS = 32,
};

}  // namespace sitrk
