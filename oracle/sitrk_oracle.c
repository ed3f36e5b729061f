/*
 * sitrk_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A scalar fp64 restatement, in plain C, of the per-buoy advection hot path of
 * the reference tracker.  It exists only to CHECK the HIP
 * path: it may be imported/linked/executed by tests/, by
 * __graft_entry__.smoke() and by bench.py's `cpu_baseline` leg, and by nothing
 * else.  The product (sitrack_amd/) never calls into it and has no CPU
 * fallback.
 *
 * Pinning: every function below is checked against golden vectors generated
 * from the reference's own Python functions (tests/golden/gen_golden.py,
 * fixtures G1..G9) and against the reference's own known answers
 * (tools/tests/test_pnt_inside_quad.py:16-24 and the seeding NetCDF fixture
 * tools/nc/sitrack_seeding_sidfex_19961215_00_HSS5.nc__KEEP).
 * EXCEPTION: the inverse polar-stereographic map (orc_cart2geo) follows the
 * published PROJ `stere` ellipsoidal algorithm because the reference delegates
 * to cartopy/pyproj/PROJ (un-vendored, no version pin, absent here): it is
 * pinned only by round trip against the forward map, which itself reproduces
 * the reference fixture at float32 -> "inverse-projection parity unpinned by
 * reference tests" (see DESIGN.md).
 *
 * Arithmetic contract: IEEE double, the reference's operation order, no FMA
 * contraction (build with -ffp-contract=off, no -ffast-math).  Array order is
 * always [y,x] / [lat,lon] / [j,i] like the reference.
 *
 * Indexing contract: the reference indexes numpy arrays with Python ints, so a
 * negative index wraps (k -> k+N) and an index >= N raises IndexError.
 * PYIDX() reproduces the wrap; an index that would raise is reported through
 * the function's status (ORC_EINDEX) instead of crashing.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK      0
#define ORC_EINDEX  (-2)   /* reference would raise IndexError            */
#define ORC_EDIR    (-3)   /* reference would print 'unknown direction'   */

#define ORC_FILL   (-9999.0)   /* sitrack/ncio.py:19 FillValue */

typedef int64_t i64;

/* ---- Python-style index ------------------------------------------------ */
static inline i64 pyidx(i64 k, i64 n, int *err)
{
    if (k < 0) k += n;
    if (k < 0 || k >= n) { *err = ORC_EINDEX; return 0; }
    return k;
}
#define AT(arr, j, i) ((arr)[pyidx((j), Nj, &err) * Ni + pyidx((i), Ni, &err)])

/* ------------------------------------------------------------------------
 * _ccw_            sitrack/tracking.py:44-49
 * points are [y,x]
 * ---------------------------------------------------------------------- */
int orc_ccw(const double *A, const double *B, const double *C)
{
    double lhs = (C[0] - A[0]) * (B[1] - A[1]);
    double rhs = (B[0] - A[0]) * (C[1] - A[1]);
    return lhs > rhs;
}

/* intersect2Seg    sitrack/tracking.py:51-58 */
int orc_intersect2seg(const double *A, const double *B, const double *C, const double *D)
{
    return (orc_ccw(A, C, D) != orc_ccw(B, C, D)) && (orc_ccw(A, B, C) != orc_ccw(A, B, D));
}

/* ------------------------------------------------------------------------
 * IsInsideQuadrangle   sitrack/locate.py:49-78
 * quad is (4,2) row-major [[y0,x0],...].  Ray cast with n+1 = 5 edge visits,
 * the first on the degenerate edge quad[0]->quad[0]; `xints` is carried over
 * between visits when the edge is horizontal (locate.py:63,71-73).
 * ---------------------------------------------------------------------- */
int orc_is_inside_quadrangle(double y, double x, const double *quad)
{
    int inside = 0;
    double xints = 0.0;
    double z1y = quad[0], z1x = quad[1];
    for (int i = 0; i < 5; i++) {
        double z2y = quad[2 * (i % 4)], z2x = quad[2 * (i % 4) + 1];
        double ymin = (z2y < z1y) ? z2y : z1y;      /* Python min(z1y,z2y) */
        double ymax = (z2y > z1y) ? z2y : z1y;      /* Python max(z1y,z2y) */
        if (y > ymin) {
            if (y <= ymax) {
                double xmax = (z2x > z1x) ? z2x : z1x;
                if (x <= xmax) {
                    if (z1y != z2y)
                        xints = (y - z1y) * (z2x - z1x) / (z2y - z1y) + z1x;
                    if (z1x == z2x || x <= xints)
                        inside = !inside;
                }
            }
        }
        z1y = z2y; z1x = z2x;
    }
    return inside;
}

/* ------------------------------------------------------------------------
 * CrossedEdge      sitrack/tracking.py:182-200
 * ji4vert is (2,4) row-major: [[jbl,jbr,jur,jul],[ibl,ibr,iur,iul]].
 * Returns 1..4 (first edge hit; falls through to 4 when none intersects),
 * or a negative status.
 * ---------------------------------------------------------------------- */
int orc_crossed_edge(const double *P1, const double *P2, const i64 *ji4vert,
                     const double *Yf, const double *Xf, i64 Nj, i64 Ni)
{
    int err = 0, kk;
    for (kk = 0; kk < 4; kk++) {
        int kp1 = (kk + 1) % 4;
        i64 j1 = ji4vert[kk], i1 = ji4vert[4 + kk];
        i64 j2 = ji4vert[kp1], i2 = ji4vert[4 + kp1];
        double C[2] = { AT(Yf, j1, i1), AT(Xf, j1, i1) };
        double D[2] = { AT(Yf, j2, i2), AT(Xf, j2, i2) };
        if (err) return err;
        if (orc_intersect2seg(P1, P2, C, D)) break;
    }
    if (kk == 4) kk = 3;            /* Python loop variable after exhaustion */
    return kk + 1;
}

/* ------------------------------------------------------------------------
 * NewHostCell      sitrack/tracking.py:203-249
 * Refines the crossed edge 1..4 into the diagonal moves 5..8.
 * ---------------------------------------------------------------------- */
int orc_new_host_cell(int kcross, const double *P1, const double *P2, const i64 *ji4vert,
                      const double *Yf, const double *Xf, i64 Nj, i64 Ni)
{
    int err = 0, knhc = kcross;
    i64 jbl = ji4vert[0], jbr = ji4vert[1], jur = ji4vert[2], jul = ji4vert[3];
    i64 ibl = ji4vert[4], ibr = ji4vert[5], iur = ji4vert[6], iul = ji4vert[7];
#define SEG(ja, ia, jb, ib, code)                                              \
    {                                                                          \
        double C[2] = { AT(Yf, ja, ia), AT(Xf, ja, ia) };                      \
        double D[2] = { AT(Yf, jb, ib), AT(Xf, jb, ib) };                      \
        if (err) return err;                                                   \
        if (orc_intersect2seg(P1, P2, C, D)) return (code);                    \
    }
    if (kcross == 1) {                       /* bottom edge  :217-222 */
        SEG(jbl, ibl, jbl - 1, ibl, 5)
        SEG(jbr, ibr, jbr - 1, ibr, 6)
    } else if (kcross == 2) {                /* right edge   :224-229 */
        SEG(jbr, ibr, jbr, ibr + 1, 6)
        SEG(jur, iur, jur, iur + 1, 7)
    } else if (kcross == 3) {                /* upper edge   :231-236 */
        SEG(jul, iul, jul + 1, iul, 8)
        SEG(jur, iur, jur + 1, iur, 7)
    } else if (kcross == 4) {                /* left edge    :238-243 */
        SEG(jul, iul, jul, iul - 1, 8)
        SEG(jbl, ibl, jbl, ibl - 1, 5)
    }
#undef SEG
    return knhc;
}

/* ------------------------------------------------------------------------
 * UpdtInd4NewCell  sitrack/tracking.py:253-305   (mutates in place)
 * ---------------------------------------------------------------------- */
int orc_updt_ind4newcell(int knhc, i64 *ji4vert, i64 *jiT)
{
    int dj, di;
    switch (knhc) {
    case 1: dj = -1; di =  0; break;
    case 2: dj =  0; di =  1; break;
    case 3: dj =  1; di =  0; break;
    case 4: dj =  0; di = -1; break;
    case 5: dj = -1; di = -1; break;
    case 6: dj = -1; di =  1; break;
    case 7: dj =  1; di =  1; break;
    case 8: dj =  1; di = -1; break;
    default: return ORC_EDIR;
    }
    for (int k = 0; k < 4; k++) { ji4vert[k] += dj; ji4vert[4 + k] += di; }
    jiT[0] += dj; jiT[1] += di;
    return ORC_OK;
}

/* ------------------------------------------------------------------------
 * Survive          sitrack/tracking.py:62-93
 * Returns the kill code 0/1 (the reference increments `ikill` at most once);
 * *which (optional) tells which test fired: 1 rim, 2 mask, 3 ice.
 * `sic` must be a 2-D field (callers on the path always pass one).
 * ---------------------------------------------------------------------- */
int orc_survive(i64 jT, i64 iT, const int8_t *tmask, const double *sic,
                i64 Nj, i64 Ni, double rmin_conc, int *which)
{
    int err = 0;
    if (which) *which = 0;
    if (jT == 0 || jT == 1 || jT == Nj - 2 || jT == Nj - 1 ||
        iT == 0 || iT == 1 || iT == Ni - 2 || iT == Ni - 1) {
        if (which) *which = 1;
        return 1;
    }
    int zmt = AT(tmask, jT, iT) + AT(tmask, jT, iT + 1) + AT(tmask, jT + 1, iT)
            + AT(tmask, jT, iT - 1) + AT(tmask, jT - 1, iT - 1);
    if (err) return err;
    if (zmt < 5) { if (which) *which = 2; return 1; }
    double zic = 0.2 * (AT(sic, jT, iT) + AT(sic, jT, iT + 1) + AT(sic, jT + 1, iT)
                        + AT(sic, jT, iT - 1) + AT(sic, jT - 1, iT - 1));
    if (err) return err;
    if (zic < rmin_conc) { if (which) *which = 3; return 1; }
    return 0;
}

/* ------------------------------------------------------------------------
 * Haversine        sitrack/util.py:85-103   (R = 6360 km)
 * ---------------------------------------------------------------------- */
double orc_haversine(double plat, double plon, double xlat, double xlon)
{
    const double to_rad = 3.141592653589793 / 180.;
    const double R = 6360.;
    double a1 = sin(0.5 * ((xlat - plat) * to_rad));
    double a2 = sin(0.5 * ((xlon - plon) * to_rad));
    double a3 = cos(xlat * to_rad) * cos(plat * to_rad);
    return 2. * R * asin(sqrt(a1 * a1 + a3 * a2 * a2));
}

void orc_haversine_field(double plat, double plon, const double *xlat, const double *xlon,
                         i64 n, double *out)
{
    for (i64 k = 0; k < n; k++) out[k] = orc_haversine(plat, plon, xlat[k], xlon[k]);
}

/* ------------------------------------------------------------------------
 * NearestPoint     sitrack/locate.py:222-276  + find_ji_of_min :13-20
 * Whole-domain form (no ji_prv), as SeedInit calls it.  `resol` may be NULL
 * (then rd_found_km is the acceptance radius).  Returns (jy,jx) or (-1,-1);
 * *dmin receives the distance of the argmin.
 * ---------------------------------------------------------------------- */
void orc_nearest_point(double latP, double lonP, const double *latT, const double *lonT,
                       const double *resol, i64 Nj, i64 Ni, double rd_found_km, int max_itr,
                       i64 *jy_out, i64 *jx_out, double *dmin_out)
{
    i64 n = Nj * Ni, kmin = 0;
    double dmin = INFINITY;
    /* the distance field and its argmin do not change across `igo` rounds
     * (locate.py:257-258 recomputes the same thing); first minimum in C order */
    for (i64 k = 0; k < n; k++) {
        double d = orc_haversine(latP, lonP, latT[k], lonT[k]);
        if (d < dmin) { dmin = d; kmin = k; }
    }
    i64 jy = kmin / Ni, jx = kmin % Ni;
    double rfnd = rd_found_km;
    int lfound = 0, igo = 0;
    while (!lfound && igo < max_itr) {
        igo = igo + 1;
        if (igo == 1 && resol) rfnd = 0.5 * resol[jy * Ni + jx];
        if (igo == 1) igo = 2;                         /* not lbox, :262 */
        lfound = (dmin < rfnd);
        if (igo > 1 && !lfound) rfnd = 1.2 * rfnd;
    }
    if (igo == max_itr) { jy = -1; jx = -1; }          /* :271 */
    *jy_out = jy; *jx_out = jx;
    if (dmin_out) *dmin_out = dmin;
}

/* ------------------------------------------------------------------------
 * FindContainingCell   sitrack/locate.py:280-330
 * Candidates: centre, i+1, j+1, i-1, j-1.  Outputs jiT[2] and vert (2,4).
 * Returns 1 found / 0 not found / negative status.
 * ---------------------------------------------------------------------- */
int orc_find_containing_cell(double zy, double zx, i64 kj, i64 ki,
                             const double *Yf, const double *Xf, i64 Nj, i64 Ni,
                             i64 *jiT, i64 *vert)
{
    static const int dj[5] = { 0, 0, 1, 0, -1 };
    static const int di[5] = { 0, 1, 0, -1, 0 };
    int err = 0, lPin = 0, kp = 0;
    i64 jT = kj, iT = ki;
    i64 jf[4] = {0}, i_f[4] = {0};
    while (!lPin && kp < 5) {
        jT = kj + dj[kp]; iT = ki + di[kp];
        kp++;
        jf[0] = jT - 1; jf[1] = jT - 1; jf[2] = jT; jf[3] = jT;
        i_f[0] = iT - 1; i_f[1] = iT; i_f[2] = iT; i_f[3] = iT - 1;
        double quad[8];
        for (int k = 0; k < 4; k++) {
            quad[2 * k] = AT(Yf, jf[k], i_f[k]);
            quad[2 * k + 1] = AT(Xf, jf[k], i_f[k]);
        }
        if (err) return err;
        lPin = orc_is_inside_quadrangle(zy, zx, quad);
    }
    jiT[0] = jT; jiT[1] = iT;
    for (int k = 0; k < 4; k++) { vert[k] = jf[k]; vert[4 + k] = i_f[k]; }
    return lPin;
}

/* ------------------------------------------------------------------------
 * SeedInit         sitrack/tracking.py:98-178   (per-seed part :120-160)
 * Inputs  pSG (nP,2) lat/lon, pSC (nP,2) y/x.
 * Outputs jiT (nP,2), vert (nP,2,4), keep (nP) 0/1 -- un-compacted; the caller
 * compacts with where(keep==1) like tracking.py:166-178.
 * `why` (optional, nP): 0 kept, 1 no nearest point, 2 Survive, 3 no cell.
 * ---------------------------------------------------------------------- */
int orc_seed_init(i64 nP, const double *pSG, const double *pSC,
                  const double *latT, const double *lonT, const double *Yf, const double *Xf,
                  const double *resol, const int8_t *tmask, const double *sic,
                  i64 Nj, i64 Ni, double rmin_conc, double rd_found_km, int max_itr,
                  i64 *jiT, i64 *vert, int8_t *keep, int8_t *why, int nthreads)
{
    int status = ORC_OK;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1) reduction(min:status)
#endif
    for (i64 p = 0; p < nP; p++) {
        i64 jT, iT;
        keep[p] = 1;
        if (why) why[p] = 0;
        jiT[2 * p] = 0; jiT[2 * p + 1] = 0;
        memset(vert + 8 * p, 0, 8 * sizeof(i64));
        orc_nearest_point(pSG[2 * p], pSG[2 * p + 1], latT, lonT, resol, Nj, Ni,
                          rd_found_km, max_itr, &jT, &iT, NULL);
        if (jT < 0 || iT < 0) { keep[p] = 0; if (why) why[p] = 1; continue; }
        int ic = orc_survive(jT, iT, tmask, sic, Nj, Ni, rmin_conc, NULL);
        if (ic < 0) { if (ic < status) status = ic; continue; }
        if (ic > 0) { keep[p] = 0; if (why) why[p] = 2; continue; }
        int lPin = orc_find_containing_cell(pSC[2 * p], pSC[2 * p + 1], jT, iT, Yf, Xf, Nj, Ni,
                                            jiT + 2 * p, vert + 8 * p);
        if (lPin < 0) { if (lPin < status) status = lPin; continue; }
        if (!lPin) { keep[p] = 0; if (why) why[p] = 3; }
    }
    return status;
}

/* ------------------------------------------------------------------------
 * One model record of the hot loop:  si3_part_tracker.py:378-490
 *
 * State (mutated): pos (nP,2) [y,x] km -- the CURRENT position (xPosC[jt]);
 *   jiT (nP,2); vert (nP,2,4); alive (nP).
 * Outputs for record jt+1: pos_next (nP,2) (FillValue where the buoy did not
 *   step -- si3_part_tracker.py:327 initialisation) and mask_next (nP)
 *   (xmask[jt+1], :460).  Stepped buoys also get `pos` advanced.
 * rec_first/rec_last: per-buoy z1stModelRec/zLstModelRec (:264-265,380);
 * uv_strategy: iUVstrategy (:37-40).  u,v,sic: the record's (Nj,Ni) slabs as
 * fp64 (the reference assigns them into fp64 arrays, :372-374).
 * Note the kill timing (:459-460 then :483-484): position and mask are written
 * BEFORE the kill test.
 * ---------------------------------------------------------------------- */
/* EXTRA velocity rule (uv_strategy = 2), not in the reference: linear interpolation between the two points that
 * carry the component, at the buoy's clamped projection on the segment joining them.  Restated here only so that the
 * GPU implementation of the extra has an independent check; there is no reference behaviour to pin it to. */
static double lerp_on_segment(const double *P, const double *A, const double *B, double fa, double fb)
{
    const double dy = B[0] - A[0], dx = B[1] - A[1];
    const double den = dy * dy + dx * dx;
    double s = ((P[0] - A[0]) * dy + (P[1] - A[1]) * dx) / den;
    s = (s < 0.0) ? 0.0 : ((s > 1.0) ? 1.0 : s);
    return (1.0 - s) * fa + s * fb;
}

static int advect_one(i64 p, i64 jrec, double rdt, int uv_strategy, double rmin_conc,
                      i64 Nj, i64 Ni,
                      const double *Yf, const double *Xf, const double *Yu, const double *Xu,
                      const double *Yv, const double *Xv, const int8_t *tmask,
                      const double *u, const double *v, const double *sic,
                      const i64 *rec_first, const i64 *rec_last,
                      double *pos, i64 *jiT, i64 *vert, int8_t *alive,
                      double *pos_next, int8_t *mask_next, i64 *ncross)
{
    int err = 0;
    if (pos_next) { pos_next[2 * p] = ORC_FILL; pos_next[2 * p + 1] = ORC_FILL; }
    if (mask_next) mask_next[p] = 0;
    if (!(alive[p] == 1 && jrec >= rec_first[p] && jrec <= rec_last[p])) return ORC_OK;

    double ry = pos[2 * p], rx = pos[2 * p + 1];
    i64 *vr = vert + 8 * p;
    i64 jT = jiT[2 * p], iT = jiT[2 * p + 1];

    /* vMesh from VRTCS (:394-402); it is a pure cache of VRTCS */
    double quad[8];
    for (int k = 0; k < 4; k++) {
        quad[2 * k] = AT(Yf, vr[k], vr[4 + k]);
        quad[2 * k + 1] = AT(Xf, vr[k], vr[4 + k]);
    }

    double zU, zV;
    if (uv_strategy == 0) {                                 /* :423-425 */
        zU = 0.5 * (AT(u, jT, iT) + AT(u, jT, iT - 1));
        zV = 0.5 * (AT(v, jT, iT) + AT(v, jT - 1, iT));
    } else if (uv_strategy == 2) {                          /* extra, see lerp_on_segment */
        double P[2] = { ry, rx };
        double Va[2] = { AT(Yv, jT - 1, iT), AT(Xv, jT - 1, iT) };
        double Vb[2] = { AT(Yv, jT, iT), AT(Xv, jT, iT) };
        double Ua[2] = { AT(Yu, jT, iT - 1), AT(Xu, jT, iT - 1) };
        double Ub[2] = { AT(Yu, jT, iT), AT(Xu, jT, iT) };
        zU = lerp_on_segment(P, Ua, Ub, AT(u, jT, iT - 1), AT(u, jT, iT));
        zV = lerp_on_segment(P, Va, Vb, AT(v, jT - 1, iT), AT(v, jT, iT));
    } else {                                                /* :427-441 */
        double P[2] = { ry, rx };
        double F[2] = { AT(Yf, jT, iT), AT(Xf, jT, iT) };
        double Va[2] = { AT(Yv, jT - 1, iT), AT(Xv, jT - 1, iT) };
        double Vb[2] = { AT(Yv, jT, iT), AT(Xv, jT, iT) };
        double Ua[2] = { AT(Yu, jT, iT - 1), AT(Xu, jT, iT - 1) };
        double Ub[2] = { AT(Yu, jT, iT), AT(Xu, jT, iT) };
        int llum1 = orc_intersect2seg(P, F, Va, Vb);
        int llvm1 = orc_intersect2seg(P, F, Ua, Ub);
        zU = llum1 ? AT(u, jT, iT - 1) : AT(u, jT, iT);
        zV = llvm1 ? AT(v, jT - 1, iT) : AT(v, jT, iT);
    }
    if (err) return err;

    double dx = zU * rdt;                                   /* :452-458 */
    double dy = zV * rdt;
    double rx_nxt = rx + dx / 1000.;
    double ry_nxt = ry + dy / 1000.;
    pos[2 * p] = ry_nxt; pos[2 * p + 1] = rx_nxt;
    if (pos_next) { pos_next[2 * p] = ry_nxt; pos_next[2 * p + 1] = rx_nxt; }
    if (mask_next) mask_next[p] = 1;

    if (!orc_is_inside_quadrangle(ry_nxt, rx_nxt, quad)) {  /* :466-484 */
        double P1[2] = { ry, rx }, P2[2] = { ry_nxt, rx_nxt };
        int icross = orc_crossed_edge(P1, P2, vr, Yf, Xf, Nj, Ni);
        if (icross < 0) return icross;
        int inhc = orc_new_host_cell(icross, P1, P2, vr, Yf, Xf, Nj, Ni);
        if (inhc < 0) return inhc;
        int st = orc_updt_ind4newcell(inhc, vr, jiT + 2 * p);
        if (st < 0) return st;
        int icncl = orc_survive(jiT[2 * p], jiT[2 * p + 1], tmask, sic, Nj, Ni, rmin_conc, NULL);
        if (icncl < 0) return icncl;
        if (icncl > 0) alive[p] = 0;
        (*ncross)++;
    }
    return ORC_OK;
}

/* nthreads > 1 uses OpenMP over buoys when built with -fopenmp (buoys never
 * interact: no term of si3_part_tracker.py:378-490 reads another buoy). */
int orc_advect_record(i64 nP, i64 jrec, double rdt, int uv_strategy, double rmin_conc,
                      i64 Nj, i64 Ni,
                      const double *Yf, const double *Xf, const double *Yu, const double *Xu,
                      const double *Yv, const double *Xv, const int8_t *tmask,
                      const double *u, const double *v, const double *sic,
                      const i64 *rec_first, const i64 *rec_last,
                      double *pos, i64 *jiT, i64 *vert, int8_t *alive,
                      double *pos_next, int8_t *mask_next, i64 *ncross_out, int nthreads)
{
    i64 ncross = 0;
    int status = ORC_OK;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1) \
        reduction(+:ncross) reduction(min:status)
#endif
    for (i64 p = 0; p < nP; p++) {
        i64 nc = 0;
        int st = advect_one(p, jrec, rdt, uv_strategy, rmin_conc, Nj, Ni, Yf, Xf, Yu, Xu, Yv, Xv,
                            tmask, u, v, sic, rec_first, rec_last, pos, jiT, vert, alive,
                            pos_next, mask_next, &nc);
        ncross += nc;
        if (st < status) status = st;
    }
    if (ncross_out) *ncross_out = ncross;
    return status;
}

/* ------------------------------------------------------------------------
 * Polar stereographic, WGS84, lat_ts = 70, lon_0 = -45 (EPSG:3413 parameters)
 * Reference call sites: Geo2CartNPSkm1D sitrack/util.py:394-410 (forward),
 * CartNPSkm2Geo1D sitrack/util.py:413-429 (inverse, si3_part_tracker.py:493).
 * The arithmetic lives in cartopy -> pyproj -> PROJ `stere` (ellipsoidal,
 * north-pole mode); restated from PROJ's published algorithm (Snyder 1987,
 * eqs 21-33/21-34/21-39, 7-9 iteration) -- see header note on pinning.
 * ---------------------------------------------------------------------- */
#define WGS84_A   6378137.0
#define WGS84_RF  298.257223563
#define NPS_NITER 8
#define NPS_CONV  1.e-10

static double nps_e(void)
{
    double f = 1.0 / WGS84_RF;
    return sqrt(2.0 * f - f * f);
}
/* Snyder (15-9): t = tan(pi/4 - phi/2) / ((1-e sin)/(1+e sin))^(e/2) */
static double nps_tsfn(double phi, double sinphi, double e)
{
    double es = e * sinphi;
    return tan(0.5 * (M_PI_2 - phi)) / pow((1.0 - es) / (1.0 + es), 0.5 * e);
}
static double nps_akm1(double lat_ts_deg, double e)
{
    double phits = fabs(lat_ts_deg) * (M_PI / 180.0);
    if (fabs(phits - M_PI_2) < 1e-10)
        return 2.0 / sqrt(pow(1 + e, 1 + e) * pow(1 - e, 1 - e));
    double t = sin(phits);
    double akm1 = cos(phits) / nps_tsfn(phits, t, e);
    t *= e;
    return akm1 / sqrt(1.0 - t * t);
}

/* forward: (lat,lon)[deg] (n,2) -> (y,x)[km] (n,2) */
void orc_geo2cart(i64 n, const double *latlon, double lat_ts, double lon0, double *yx)
{
    double e = nps_e(), akm1 = nps_akm1(lat_ts, e);
    const double d2r = M_PI / 180.0;
    for (i64 k = 0; k < n; k++) {
        double phi = latlon[2 * k] * d2r;
        double lam = (latlon[2 * k + 1] - lon0) * d2r;
        /* PROJ normalises lam to [-pi,pi]; harmless for sin/cos */
        double rho = (fabs(phi - M_PI_2) < 1e-15) ? 0.0 : akm1 * nps_tsfn(phi, sin(phi), e);
        double x = rho * sin(lam);
        double y = -rho * cos(lam);
        yx[2 * k] = WGS84_A * y / 1000.;
        yx[2 * k + 1] = WGS84_A * x / 1000.;
    }
}

/* inverse: (y,x)[km] (n,2) -> (lat,lon)[deg] (n,2); lon in [-180,180] */
void orc_cart2geo(i64 n, const double *yx, double lat_ts, double lon0, double *latlon)
{
    double e = nps_e(), akm1 = nps_akm1(lat_ts, e);
    const double r2d = 180.0 / M_PI;
    for (i64 k = 0; k < n; k++) {
        double x = 1000. * yx[2 * k + 1] / WGS84_A;
        double y = 1000. * yx[2 * k] / WGS84_A;
        double rho = hypot(x, y);
        y = -y;                                        /* north-pole mode */
        double tp = -rho / akm1;
        double phi_l = M_PI_2 - 2. * atan(tp);
        const double halfpi = -M_PI_2, halfe = -.5 * e;
        double phi = phi_l, lam = 0.0;
        int ok = 0;
        for (int i = NPS_NITER; i-- > 0; phi_l = phi) {
            double sinphi = e * sin(phi_l);
            phi = 2. * atan(tp * pow((1. + sinphi) / (1. - sinphi), halfe)) - halfpi;
            if (fabs(phi_l - phi) < NPS_CONV) { ok = 1; break; }
        }
        lam = (x == 0. && y == 0.) ? 0. : atan2(x, y);
        double lon = lam * r2d + lon0;
        /* PROJ adjlon: wrap into [-180,180] */
        if (lon < -180.0 || lon > 180.0) {
            lon = lon + 180.0;
            lon = lon - 360.0 * floor(lon / 360.0);
            lon = lon - 180.0;
        }
        latlon[2 * k] = ok ? phi * r2d : NAN;
        latlon[2 * k + 1] = ok ? lon : NAN;
    }
}

/* ------------------------------------------------------------------------
 * GetTimeSpan      sitrack/tracking.py:8-37  (host logic; kept here so the
 * driver test has an independent restatement).  Returns 0, or -1 where the
 * reference prints PROBLEM and exits.
 * ---------------------------------------------------------------------- */
int orc_get_time_span(double dt, const i64 *vtime_mod, i64 nt, i64 iSdA, i64 iMdA, i64 iMdB,
                      int has_stop, i64 iStop, i64 *Nt, i64 *kt0, i64 *ktN, i64 *itM0, i64 *itMN)
{
    if ((double)iSdA < (double)iMdA - dt / 2 || (double)iSdA > (double)iMdB - dt / 2) return -1;
    i64 k0 = 0, best = llabs(vtime_mod[0] - iSdA);
    for (i64 k = 1; k < nt; k++) { i64 d = llabs(vtime_mod[k] - iSdA); if (d < best) { best = d; k0 = k; } }
    if (iSdA >= vtime_mod[k0]) k0 += 1;
    i64 kN;
    if (has_stop && iStop != 0) {            /* `ltStop = ( iStop )` truthiness */
        kN = 0; best = llabs(vtime_mod[0] - iStop);
        for (i64 k = 1; k < nt; k++) { i64 d = llabs(vtime_mod[k] - iStop); if (d < best) { best = d; kN = k; } }
    } else {
        kN = nt - 1;
    }
    *kt0 = k0; *ktN = kN; *Nt = kN - k0 + 1;
    *itM0 = (k0 >= 0 && k0 < nt) ? vtime_mod[k0] : 0;
    *itMN = vtime_mod[kN];
    return (k0 >= nt) ? ORC_EINDEX : 0;
}
