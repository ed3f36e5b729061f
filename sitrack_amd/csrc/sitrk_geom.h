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

// ---------------------------------------------------------------------------
// Division-free evaluation of the same two results for the fused loop (advect_run_kernel), which is bound by
// VALU issue: an fp64 division is an 11-instruction sequence on gfx950.  Both replacements return, for every
// input, exactly what the expressions above return; the arguments are in DESIGN.md section 3.1.
// ---------------------------------------------------------------------------

// x / 1000. , correctly rounded, without a division.
//   z = RN(1/1000);  q = RN(x z);  r = x - 1000 q (exact in an fma);  result = RN(q + r z) = RN(v + (v-q) e_z),  v = x/1000.
// |v-q| <= 2u|v| and |e_z| <= u (u = 2^-53), so the argument of the last rounding is within 2^-105 |v| of v, while
// x/1000 = X 2^e / (8 * 125) is never closer than 2^-61 |v| to a midpoint of two doubles (125 (2M+1) is odd, X 2^g is
// even) -> the rounding lands where RN(v) does.  Needs no over/underflow in q, r: guaranteed for 2^-900 <= |x| < 2^900;
// everything else (zeros with their sign, subnormals, huge values, inf, NaN) takes the division.
// `k` is the constant 1000.: the fused loop hands it over in a scalar register pair (as a literal the compiler needs a
// register copy of x in front of the fma).
__device__ __forceinline__ double div1000_core(double x, double k, bool in_range)
{
    const double z = 0x1.0624dd2f1a9fcp-10;                       // RN(1/1000)
    const double q = x * z;
    const double r = __builtin_fma(-q, k, x);
    double f = __builtin_fma(r, z, q);
    if (__builtin_expect(!in_range, 0)) {
        double t = x;
        asm volatile("" : "+v"(t));                               // keeps the division in a branch of its own (not if-converted)
        f = t / 1000.;
    }
    return f;
}

__device__ __forceinline__ double div1000(double x, double k = 1000.)
{
    const unsigned e = ((unsigned)__double2hiint(x) >> 20) & 0x7ffu;
    return div1000_core(x, k, (e - 123u) < 1800u);
}

// x = RN((double)src * rdt) with `src` a binary32 value and 2^-700 <= |rdt| <= 2^700 (checked by the host, which hands over
// class_mask = 0 otherwise: every lane then takes the division): x lies in [2^-900, 2^900) exactly when src is finite and
// not zero (2^-149 <= |src| < 2^128), i.e. when v_cmp_class says normal or subnormal -- one instruction on the value as
// loaded instead of three on the product's exponent.
static constexpr int kClassFiniteNonzeroF32 = 0x008 | 0x010 | 0x080 | 0x100;   // -normal, -subnormal, +subnormal, +normal
__device__ __forceinline__ double div1000_of_f32(double x, float src, int class_mask, double k)
{
    return div1000_core(x, k, __builtin_amdgcn_classf(src, class_mask));
}

// IsInsideQuadrangle without the divisions.  Per edge A->B the reference toggles when
//     y in (min(Ay,By), max(Ay,By)]  and  x <= max(Ax,Bx)  and  ( Ax == Bx  or  x <= X ),
//     X = RN(RN(RN(a b) / c) + Ax),   a = RN(y-Ay), b = RN(Bx-Ax), c = RN(By-Ay)
// (the stale `xints` of a horizontal edge is never read: such an edge fails the strict y-range test).
//  * the range tests are rewritten on the vertices: (y > Ay) != (y > By), (x <= Ax) | (x <= Bx) - identities for
//    finite vertices, and one comparison per vertex instead of min/max per edge;
//  * sign(x - X) is taken from E = fma(RN(x-Ax), c, -RN(a b)) whenever |E| > |c| * 32u (|x| + Mg), Mg >= every
//    |vertex coordinate|: x - X = E'/c - eta with |eta| <= u (|x| + 6.01 Mg) (|a| <= |c| inside the y-range bounds the
//    quotient by |b|), so beyond that margin E/c and x - X have the same sign and the toggle is (E < 0) == (y > Ay)
//    (c > 0 <=> y > Ay there).  Ax == Bx needs no case of its own (x < Ax decides, x == Ax is not decided);
//  * a lane with any edge not decided that way (within ~1e-11 km of the edge, or non-finite) re-evaluates the
//    whole test with inside_quad() above;
//  * each edge sits in a branch of its own (the empty asm keeps it from being if-converted): an edge no lane of the
//    wave is level with costs nothing, as in the plain form.
// `eps_mg` = 2^-48 * Mg (sitrk_set_grid).
__device__ __forceinline__ bool inside_quad_hot(double y, double x, pt q0, pt q1, pt q2, pt q3, double eps_mg)
{
    const double mx = __builtin_fma(fabs(x), 0x1p-48, eps_mg);
    const bool g0 = y > q0.y, g1 = y > q1.y, g2 = y > q2.y, g3 = y > q3.y;
    const bool l0 = x <= q0.x, l1 = x <= q1.x, l2 = x <= q2.x, l3 = x <= q3.x;
    bool inside = false, decided = true;                 // logical operators: the predicates stay lane masks
    // (the bookkeeping in bitwise form: lane masks combined by a handful of scalar instructions, no nested exec region)
#define SITRK_EDGE(A, B, gA, gB, lA, lB)                                              \
    if ((gA != gB) && (lA || lB)) {                                                   \
        double c = B.y - A.y;                                                         \
        asm volatile("" : "+v"(c));                                                   \
        const double p = (y - A.y) * (B.x - A.x);                                     \
        const double E = __builtin_fma(x - A.x, c, -p);                               \
        const bool dec = fabs(E) > __builtin_fma(fabs(c), mx, 0x1p-1000);             \
        const bool neg = __double2hiint(E) < 0;                                       \
        decided = decided & dec;                                                      \
        inside = inside != (dec & (neg == gA));                                       \
    }
    SITRK_EDGE(q0, q1, g0, g1, l0, l1)
    SITRK_EDGE(q1, q2, g1, g2, l1, l2)
    SITRK_EDGE(q2, q3, g2, g3, l2, l3)
    SITRK_EDGE(q3, q0, g3, g0, l3, l0)
#undef SITRK_EDGE
    const bool undecided = !decided;
    if (undecided) inside = inside_quad(y, x, q0, q1, q2, q3);
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
    TUNE_SURVIVE_TILE = 4,         // derive the Survive bytes with the LDS-tile kernel even where the register-rolling one applies
    TUNE_DIAG_MEMONLY = 16,        // ablation kernels, diagnostic builds only (make DIAG=1)
    TUNE_DIAG_NOCROSS = 32,
};

}  // namespace sitrk
