/*
 * sitrk.h -- C ABI of libsitrk.so: MI355X (gfx950) per-buoy advection for the
 * `sitrack` Lagrangian sea-ice tracker.
 *
 * The reference (stephanieleroux/sitrack) is pure Python and has no FFI of its
 * own; the boundary this library replaces is the body of the record loop of
 * si3_part_tracker.py:361-496 and the `sit.*` calls made from it.  Each entry
 * point below cites the reference code it stands for.  Plain pointers and
 * sizes only; host arrays are C-contiguous, order [j,i] / [y,x] / [lat,lon]
 * like the reference.  A maintainer binds it from Python with ctypes (see
 * INTEGRATION.md); sitrack_amd/_lib.py is that binding.
 *
 * Conventions
 *   - every function returns 0 on success or a negative SITRK_E* code; the
 *     text is available from sitrk_last_error().  The library never prints
 *     and never exits (the reference's `print(...); exit(0)` convention,
 *     e.g. sitrack/tracking.py:18-20,301-303, becomes an error return).
 *   - one context = one GPU = one host thread at a time (not re-entrant per
 *     handle, no global state).  Multi-GPU = one process per GPU, each with
 *     its own context and its own contiguous range of buoys.
 *   - the caller owns all host buffers; no host pointer is retained after a
 *     call returns: inputs are copied into device memory or into the library's
 *     own pinned staging before the call comes back (a C caller may free or
 *     overwrite them at once), outputs are complete on return.  Device pointers
 *     passed to *_dev entry points must stay valid until the next sitrk_sync()/fetch.
 *   - threading: one compute stream and one copy stream per context (+ an ingest
 *     stream for the opt-in asynchronous Survive derivation).  Record uploads
 *     (sitrk_push_record*, sitrk_stage_submit*) run on the copy stream out of
 *     double-buffered pinned staging and are ordered against the kernels by
 *     events only, so the next records travel while the current ones are
 *     stepped with (SURVEY 8b "async, double-buffered").
 *   - there is NO CPU fallback: without a usable HIP device sitrk_create()
 *     fails with SITRK_EHIP.
 */
#ifndef SITRK_H
#define SITRK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SITRK_VERSION 100          /* 0.1.0 */

#define SITRK_OK        0
#define SITRK_EINVAL  (-1)         /* bad argument / call order                        */
#define SITRK_EINDEX  (-2)         /* index the reference would fault on (IndexError)  */
#define SITRK_EHIP    (-3)         /* HIP runtime error (text in sitrk_last_error)     */
#define SITRK_ENOMEM  (-4)

#define SITRK_F32 0                /* field record element types */
#define SITRK_F64 1

#define SITRK_FILL (-9999.0)       /* sitrack/ncio.py:19 FillValue */

typedef struct sitrk_ctx sitrk_t;

/* ---- context ----------------------------------------------------------- */
int         sitrk_version(void);
int         sitrk_create(sitrk_t **h, int device);
int         sitrk_destroy(sitrk_t *h);
const char *sitrk_last_error(sitrk_t *h);           /* h may be NULL: last create() error */
int         sitrk_sync(sitrk_t *h);                 /* wait for all queued work          */
/* adopt an external compute stream (hipStream_t as void*), e.g. the caller's
 * torch stream; NULL restores the library's own stream */
int         sitrk_set_stream(sitrk_t *h, void *hip_stream);

/* ---- static model grid -------------------------------------------------
 * The arrays GetModelGrid / GetModelUVGrid hand to the loop
 * (sitrack/ncio.py:22-92; si3_part_tracker.py:192,196): F-, U-, V-point plane
 * coordinates [km] and the T-point land-sea mask.  Copied to the device once and
 * re-laid out as one 48-byte record per cell.  4 <= Nj <= 32767, 4 <= Ni <= 65535, Nj*Ni <= 2^29. */
int sitrk_set_grid(sitrk_t *h, int Nj, int Ni,
                   const double *Yf, const double *Xf,
                   const double *Yu, const double *Xu,
                   const double *Yv, const double *Xv,
                   const int8_t *tmask);

/* module-level constants of the reference, same defaults:
 * rdt = 3600 (si3_part_tracker.py:31), iUVstrategy = 1 (:37; 0 = cell mean, 1 = nearest U/V point),
 * rmin_conc = 0.1 (sitrack/tracking.py:4).
 * uv_strategy = 2 is an EXTRA the reference does not have (no parity claim against it): u is interpolated linearly
 * between the cell's left and right U-points and v between its lower and upper V-points, at the buoy's clamped
 * projection on the segment joining them. */
int sitrk_set_params(sitrk_t *h, double rdt, int uv_strategy, double rmin_conc);

/* performance knobs; they never change results.  "xcd_remap" (0/1): each XCD walks a
 * contiguous chunk of the cell-sorted buoys; "nt_state" (0/1): non-temporal loads/stores for
 * the once-per-step position/cell streams; "sort_tile" (tile_j*256 + tile_i, 0 = row-major):
 * order of the cell sort, tile-major tiles of tile_j x tile_i cells; "locate_bruteforce" (0/1):
 * SeedInit scans the whole grid per seed like the reference instead of the bounding-sphere search; "survive_tile" (0/1): derive a
 * record's Survive bytes with the LDS-tile kernel even where the register-rolling one applies (meshes with Ni % 4 == 0);
 * "async_survive" (0/1, default 0): an uploaded record's Survive bytes are derived on the compute stream (0) or on the library's
 * ingest stream, next to the stepping of the resident records (1: measured slower under the fused loop, kept as a knob);
 * "fill_threads" (1..16, default 8): host threads that copy a pushed record of 8 MB or more into the pinned staging;
 * "patch_kb" (0..63, default 16) / "patch_margin" (0..64, default 8): KB of LDS per workgroup that the fused kernel may fill with
 * the F-points of the cells around its buoys (0 = none: every geometry read goes to global memory), and the widest margin of
 * cells it takes around their bounding box; "xcd_group" (0..4096, default 16): runs of that many consecutive workgroups of the
 * fused kernel share an XCD (its L2); "step_block" (256/512/1024): workgroup size of the one-record kernel; "fuse" (1..32):
 * consecutive resident records advanced per launch by sitrk_run (loop interchange: the buoys are independent, each lane keeps
 * its buoy in registers across the records). */
int sitrk_set_tuning(sitrk_t *h, const char *knob, int value);

/* ---- model records (u_ice, v_ice, siconc) -------------------------------
 * si3_part_tracker.py:372-374 reads one (Nj,Ni) slab of each per record.
 * `nslots` records are resident on the device; a slot is one contiguous slab
 * [u | v | siconc] of 3*Nj*Ni elements of `dtype` (what one RCCL broadcast moves). */
int   sitrk_alloc_records(sitrk_t *h, int nslots, int dtype);
/* Host arrays -> slot.  Asynchronous and double-buffered: the three fields are copied into the library's pinned staging
 * (so u, v, sic may be freed or reused as soon as the call returns), the DMA into the slot is queued on the copy stream
 * behind the last kernel that reads the slot, and the record's Survive mask is queued on the compute stream behind the
 * DMA.  A second push proceeds while the first one's DMA is in flight; a third waits for the first buffer to drain. */
int   sitrk_push_record(sitrk_t *h, int slot, const void *u, const void *v, const void *sic);   /* host pointers   */
/* The same without the intermediate copy, for callers that can READ INTO pinned memory (a NetCDF reader):
 * sitrk_stage_acquire hands out the next staging buffer as three arrays of nrows x Ni elements of the records' dtype
 * (blocks until the upload that last used the buffer has drained); the caller fills them and sitrk_stage_submit queues
 * them as rows [j0, j0 + nrows) of `slot` exactly like sitrk_push_record_rows (whole record: nrows = Nj, j0 = 0).
 * The pointers belong to the library and are valid until the submit (or the release); sitrk_alloc_records and
 * sitrk_destroy free the buffers, so nothing handed out before them may be touched afterwards.
 * sitrk_stage_release gives an acquired buffer back WITHOUT uploading it (a reader that failed half-way): the next
 * acquire hands out the same buffer again.  Releasing when nothing is acquired is not an error. */
int   sitrk_stage_acquire(sitrk_t *h, int nrows, void **u, void **v, void **sic);
int   sitrk_stage_submit(sitrk_t *h, int slot, int j0, int j1);
int   sitrk_stage_release(sitrk_t *h);
int   sitrk_push_record_dev(sitrk_t *h, int slot, const void *slab_dev);                        /* device pointer: [u|v|sic] */
void *sitrk_record_ptr(sitrk_t *h, int slot);   /* device address of a slot's slab (broadcast target); NULL on error */
/* A slot whose slab was (re)written in place through sitrk_record_ptr must be committed before it is
 * stepped with: this derives the record's Survive mask (sitrack/tracking.py:62-93 evaluated once per cell:
 * rim, 5-point tmask sum, 5-point siconc mean < rmin_conc) on the library's stream.  push_record* commit
 * by themselves; a slot handed out by sitrk_record_ptr and never committed is committed by the next step. */
int   sitrk_commit_record(sitrk_t *h, int slot);

/* Row-band ingest.  A step of a buoy hosted by row jT reads u,v in rows jT-1..jT and Survive bytes in rows jT-1..jT+1,
 * which derive from siconc rows jT-2..jT+2; and a host cell moves by at most one row per record (UpdtInd4NewCell,
 * sitrack/tracking.py:257-300).  So with [jmin,jmax] = sitrk_buoy_rows() (rows of the buoys still alive; jmin > jmax
 * when none) the next step can only touch rows [jmin-2, jmax+3) of a record, and only those need to be uploaded:
 * sitrk_push_record_rows copies rows [j0,j1) of the (Nj,Ni) fields (host arrays holding just those rows) into the slot
 * and derives the Survive bytes they determine (rows j0+1..j1-2 and the domain rim); other rows keep what they held.
 * Each rank of a multi-GPU run can thus ingest only the band of its own buoys: no collective at all.
 * The library remembers which rows of a slot are valid and checks every step against them: stepping with a partly
 * uploaded slot needs sitrk_buoy_rows() to have been evaluated since sitrk_set_buoys(), and fails with SITRK_EINVAL when
 * [jmin-2-age, jmax+3+age) (age = records stepped since that evaluation; record r of a fused sitrk_run counts age+r) is
 * not inside the uploaded rows.  Slots are allocated with every Survive byte = kill and every field value = NaN. */
int sitrk_buoy_rows(sitrk_t *h, int32_t *jmin, int32_t *jmax);
int sitrk_push_record_rows(sitrk_t *h, int slot, int j0, int j1, const void *u_rows, const void *v_rows, const void *sic_rows);
/* same derivation for rows [j0,j1) that the caller wrote in place through sitrk_record_ptr (device-side copies) */
int sitrk_commit_record_rows(sitrk_t *h, int slot, int j0, int j1);

/* Box ingest (rows AND columns).  The same argument in both directions: a step of a buoy hosted by (jT,iT) reads u,v in
 * rows jT-1..jT / columns iT-1..iT and Survive bytes of the cells (jT-1..jT+1, iT-1..iT+1), which derive from siconc in
 * (jT-2..jT+2, iT-2..iT+2); UpdtInd4NewCell moves a host cell by at most one row and one column per record.  With
 * (jmin,jmax,imin,imax) = sitrk_buoy_box() the next step can only touch the box rows [jmin-2, jmax+3) x columns
 * [imin-2, imax+3) of a record (36 % of the cells of BASELINE config 3, whose buoys fill the central 60 % x 60 %):
 *   sitrk_push_record_box     host arrays -> slot (gathered into the library's pinned staging by a few threads, then three
 *                             strided DMAs), + the Survive bytes the box determines.  u_box, v_box, sic_box address element
 *                             (j0,i0); consecutive rows of the box are `ld` elements apart in the caller's arrays: ld = i1-i0
 *                             for arrays that hold exactly the box, ld = Ni for pointers into whole (Nj,Ni) fields
 *   sitrk_stage_acquire_box / sitrk_stage_submit_box   the same for a reader that fills the pinned staging itself
 *                             (a NetCDF hyperslab read of si3_part_tracker.py:372-374 restricted to the box)
 *   sitrk_commit_record_box   the slab already sits in device memory (written through sitrk_record_ptr: an RCCL broadcast, a
 *                             device-side producer): derive the Survive bytes of the box only, and treat the slot as holding
 *                             that box from now on
 * The library remembers the box a slot holds and checks every step against it exactly like the row bands: age records after
 * sitrk_buoy_box() the slot must hold rows [jmin-2-age, jmax+3+age) and columns [imin-2-age, imax+3+age), else SITRK_EINVAL.
 * sitrk_buoy_rows() evaluates the columns too (a row band is a box of full width). */
int sitrk_buoy_box(sitrk_t *h, int32_t *jmin, int32_t *jmax, int32_t *imin, int32_t *imax);
int sitrk_push_record_box(sitrk_t *h, int slot, int j0, int j1, int i0, int i1, const void *u_box, const void *v_box, const void *sic_box,
                          int64_t ld);
/* sitrk_buoy_box without stalling the stream: _begin queues the reduction behind the work already queued on the compute stream,
 * _end waits for that point only (launches queued after the begin keep the GPU busy) and adopts the result -- the box then
 * counts as evaluated at the begin; *age (optional) = records stepped since the begin.  One evaluation in flight at a time. */
int sitrk_buoy_box_begin(sitrk_t *h);
int sitrk_buoy_box_end(sitrk_t *h, int32_t *jmin, int32_t *jmax, int32_t *imin, int32_t *imax, int32_t *age);
int sitrk_stage_acquire_box(sitrk_t *h, int nrows, int ncols, void **u, void **v, void **sic);
int sitrk_stage_submit_box(sitrk_t *h, int slot, int j0, int j1, int i0, int i1);
int sitrk_commit_record_box(sitrk_t *h, int slot, int j0, int j1, int i0, int i1);
/* the same box of the nrec records in slots (slot0 + k) % nslots, k < nrec (the slots a sitrk_run of nrec records from slot0
 * steps with), derived by ONE launch: a box of a record is ~10 us of memory traffic, of the order of a dependent dispatch */
int sitrk_commit_records_box(sitrk_t *h, int slot0, int nrec, int j0, int j1, int i0, int i1);
/* the same on the library's INGEST stream instead of the compute stream: the derivation runs next to the launches that step with
 * OTHER slots -- behind the last launch that read these slots' bytes, in front of the first one that will (knob "async_survive": the
 * same for uploaded records).  For slabs that are COMPLETE in device memory when the call is made: nothing may still be writing them.
 * Measured (bench.py --fresh-overlap): slower than the plain call under the fused loop -- an option, not the default path. */
int sitrk_commit_records_box_async(sitrk_t *h, int slot0, int nrec, int j0, int j1, int i0, int i1);

/* ---- buoys ---------------------------------------------------------------
 * State of si3_part_tracker.py:324-330 reduced to what the loop reads:
 * yx = xPosC[jt] (nP,2) km; jiT = vJIt (nP,2); rec_first/rec_last =
 * z1stModelRec/zLstModelRec (:264-265), NULL = every record.  VRTCS is a pure
 * function of vJIt (sitrack/locate.py:320-321, tracking.py:257-300) and is not
 * stored.  All buoys start alive.  Requires 1 <= jT <= Nj-2, 1 <= iT <= Ni-2
 * (outside it the reference indexes out of range) -> SITRK_EINDEX. */
int sitrk_set_buoys(sitrk_t *h, int64_t nP, const double *yx, const int32_t *jiT,
                    const int32_t *rec_first, const int32_t *rec_last);

/* Buoys that arrive with a history (handed over by another rank when the ranks' latitude bands are re-balanced): right after
 * sitrk_set_buoys, mark the buoys with alive[k] == 0 as dead and give every DEAD buoy its kill record back (kill_rec[k]), so
 * that sitrk_fetch / sitrk_fetch_record answer as they did on the rank that stepped them before.  Invariant of the library:
 * a buoy is alive exactly when its kill record is -1 -- kill_rec[k] of a buoy with alive[k] != 0 is ignored (stored as -1). */
int sitrk_restore_state(sitrk_t *h, const int8_t *alive, const int32_t *kill_rec);

/* re-order the device-resident buoys by host cell (coalescing); results are
 * always returned in the caller's original order.  resort_every > 0 re-sorts
 * automatically every that many steps (default 512; 0 = never). */
int sitrk_sort_buoys(sitrk_t *h);
int sitrk_set_resort(sitrk_t *h, int resort_every);

/* One model record for every buoy: the body of `for jP in range(nP)`
 * (si3_part_tracker.py:378-490): gate (:380), velocity pick (:423-441,
 * sitrack/tracking.py:44-58), Euler update (:452-458), IsInsideQuadrangle
 * (:466, locate.py:49-78), CrossedEdge / NewHostCell / UpdtInd4NewCell / Survive
 * (:474-484, tracking.py:62-93,182-305).  `slot` = resident record used as
 * model record `jrec`.  Asynchronous. */
int sitrk_step(sitrk_t *h, int slot, int jrec);

/* nsteps records jrec0, jrec0+1, ... using slots (slot0 + k) % nslots; consecutive resident records go into one launch
 * (knob "fuse"; never across a re-sort or the slot ring).  Buoy sets with per-buoy record windows run the kernel form
 * without the window test for every launch whose records lie inside all windows. */
int sitrk_run(sitrk_t *h, int slot0, int jrec0, int nsteps);
/* What sitrk_run / sitrk_step really launched since the last reset: fused launches of advect_run_kernel, the records
 * they advanced in total (a launch is cut short at a re-sort and at the end of a run), and one-record launches of
 * advect_step_kernel.  Any pointer may be NULL.  bench.py prices its roofline per launch from these. */
int sitrk_launch_stats(sitrk_t *h, int reset, int64_t *fused_launches, int64_t *fused_records, int64_t *step_launches);

/* Current state in the caller's buoy order (any pointer may be NULL):
 * yx (nP,2) current position; jiT (nP,2) = vJIt; alive (nP) = iAlive;
 * kill_rec (nP) = model record at which the buoy was killed, -1 if alive. */
int sitrk_fetch(sitrk_t *h, double *yx, int32_t *jiT, int8_t *alive, int32_t *kill_rec);

/* What the reference stores for the record that followed model record `jrec`
 * (xPosC[jt+1], xmask[jt+1], si3_part_tracker.py:459-460): the position where
 * the buoy stepped at `jrec`, FillValue and mask 0 elsewhere.  Valid right
 * after the step of `jrec`.  latlon (optional) = CartNPSkm2Geo1D of yx_rec
 * (:493; dead buoys' -9999 km are converted too, like the reference). */
int sitrk_fetch_record(sitrk_t *h, int jrec, double *yx_rec, int8_t *mask, double *latlon);

/* ---- locate / seeding ----------------------------------------------------
 * FindContainingCell (sitrack/locate.py:280-330) for n points: from the guess
 * T-point tries centre, i+1, j+1, i-1, j-1.  found[k] 1/0; jiT_out = centre of
 * the found cell (last candidate tried when not found).  Needs set_grid. */
int sitrk_find_cells(sitrk_t *h, int64_t n, const double *yx, const int32_t *jiT_guess,
                     int32_t *jiT_out, int8_t *found);

/* SeedInit (sitrack/tracking.py:98-178), per-seed part: nearest T-point by
 * Haversine (locate.py:222-276, util.py:85-103) with the acceptance test of
 * NearestPoint as called there (rd_found_km = rFoundKM = 2.5, max_itr = 10, 2-D
 * resolkm), Survive on that T-point, FindContainingCell.  Outputs are NOT
 * compacted: keep[k] in {0,1}, why[k] (optional) 0 kept / 1 no nearest point /
 * 2 Survive / 3 no containing cell; the caller compacts with where(keep==1)
 * exactly like tracking.py:166-178.  sic: (Nj,Ni) fp64 ice concentration at
 * the seeding record (si3_part_tracker.py:228-229). */
int sitrk_seed_init(sitrk_t *h, int64_t nP, const double *latlon, const double *yx,
                    const double *latT, const double *lonT, const double *resolkm,
                    const double *sic, int32_t *jiT_out, int8_t *keep, int8_t *why);

/* NearestPoint alone (sitrack/locate.py:222-276, whole-domain form with find_ji_of_min :13-20 and Haversine
 * util.py:85-103): for each of nP points latlon (nP,2) [lat,lon] the (j,i) of the nearest T-point of the current grid's
 * latT/lonT (Nj,Ni), or (-1,-1) when the acceptance loop gives up (distance >= 0.5*resolkm[j,i] * 1.2^(max_itr-2), or
 * rd_found_km * 1.2^(max_itr-2) when resolkm is NULL).  SeedInit calls it with rd_found_km = 2.5, max_itr = 10.
 * dmin (nP) may be NULL: Haversine distance to that T-point in km (+inf for points rejected without a search). */
int sitrk_nearest_point(sitrk_t *h, int64_t nP, const double *latlon, const double *latT, const double *lonT,
                        const double *resolkm, double rd_found_km, int max_itr, int32_t *ji, double *dmin);

/* Idealised seeding on the model grid: nemoSeed (sitrack/tracking.py:365-442) + Geo2CartNPSkm1D (util.py:394-410), i.e. what
 * tools/generate_idealized_seeding.py computes (:201-386), on the device.  Every khss-th T-point of the (Nj,Ni) mesh
 * whose tmask x rmask (rmask may be NULL) is 1, whose latitude is not below 55 and whose ice concentration is not below
 * 0.9 carries a seed; with latF/lonF (both or neither) also every interior sub-sampled F-point whose four sub-sampled
 * T-neighbours carry one.  Outputs in the reference's order -- T-seeds in C order of the sub-sampled mesh, then F-seeds
 * -- as latlon (n,2) [lat,lon] and, if yx != NULL, yx (n,2) [y,x] km in the polar-stereographic plane (lat0, lon0).
 * Call once with capacity = 0 to learn the counts (*nT, *nF), then with arrays of nT + nF rows. */
int sitrk_nemo_seed(sitrk_t *h, int Nj, int Ni, int khss, const int8_t *tmask, const int8_t *rmask,
                    const double *latT, const double *lonT, const double *sic, const double *latF, const double *lonF,
                    double lat0, double lon0, int64_t capacity, double *latlon, double *yx, int64_t *nT, int64_t *nF);

/* ---- predicate probes ------------------------------------------------------
 * The device-side predicates of the hot path evaluated on plain arrays, so that each one can be held
 * against the reference function it restates (parity tests):
 *   sitrk_eval_inside    IsInsideQuadrangle (sitrack/locate.py:49-78): pts (n,2) [y,x], quads (n,4,2).  Evaluated in the
 *                        division-free form of the hot loop AND in the plain form: 0/1, +2 if the two ever disagreed
 *   sitrk_eval_euler     r + (vel * rdt) / 1000. (si3_part_tracker.py:452-458) as the hot loop evaluates it
 *   sitrk_eval_intersect intersect2Seg and _ccw_(A,B,C) (sitrack/tracking.py:44-58): segs (n,4,2) = A,B,C,D; ccw_abc may be NULL
 *   sitrk_eval_crossing  CrossedEdge + NewHostCell + UpdtInd4NewCell (tracking.py:182-305) on the current grid for a
 *                        move P1 -> P2 out of host cell jiT (n,2): new vJIt and, if codes != NULL, (n,2) = the return
 *                        values of CrossedEdge (1..4) and NewHostCell (1..8)
 *   sitrk_survive_mask   Survive (tracking.py:62-93) for every cell of the current grid with the given (Nj,Ni) fp64
 *                        ice concentration and the current rmin_conc: 1 = kill */
int sitrk_eval_inside(sitrk_t *h, int64_t n, const double *pts, const double *quads, int8_t *inside);
int sitrk_eval_euler(sitrk_t *h, int64_t n, const double *r, const double *vel, double rdt, double *out);
/* Haversine (sitrack/util.py:85-103, R = 6360 km) element by element: dist[k] between (plat[k],plon[k]) and (xlat[k],xlon[k]) */
int sitrk_eval_haversine(sitrk_t *h, int64_t n, const double *plat, const double *plon, const double *xlat,
                         const double *xlon, double *dist);
int sitrk_eval_intersect(sitrk_t *h, int64_t n, const double *segs, int8_t *intersect, int8_t *ccw_abc);
int sitrk_eval_crossing(sitrk_t *h, int64_t n, const double *P1, const double *P2, const int32_t *jiT, int32_t *jiT_new,
                        int32_t *codes);
int sitrk_survive_mask(sitrk_t *h, const double *sic, int8_t *mask);

/* ---- projection -----------------------------------------------------------
 * CartNPSkm2Geo1D / Geo2CartNPSkm1D (sitrack/util.py:394-429): WGS84 polar
 * stereographic, lat_ts = lat0, lon_0 = lon0 (defaults 70, -45 in the
 * reference), km <-> degrees, arrays (n,2) [y,x] <-> [lat,lon]. */
int sitrk_cart2geo(sitrk_t *h, int64_t n, const double *yx, double lat0, double lon0, double *latlon);
int sitrk_geo2cart(sitrk_t *h, int64_t n, const double *latlon, double lat0, double lon0, double *yx);

/* ---- measurement ----------------------------------------------------------
 * HIP events on the library's compute stream (torch.cuda.Event would only see
 * torch's stream).  timer_stop waits for the stop event and returns the elapsed ms. */
int sitrk_timer_start(sitrk_t *h);
int sitrk_timer_stop(sitrk_t *h, float *ms);
/* iAlive.sum() -- the per-record "current number of buoys alive" line of the
 * reference driver (si3_part_tracker.py:376) */
int sitrk_count_alive(sitrk_t *h, int64_t *nalive);

#ifdef __cplusplus
}
#endif
#endif /* SITRK_H */
