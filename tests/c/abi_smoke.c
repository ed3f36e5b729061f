/* Plain-C consumer of include/sitrk.h: proves the boundary is a C ABI (no C++/Python types in it).
 * Built and run by tests/test_abi.py with gcc.  Without a GPU it must fail cleanly at sitrk_create;
 * with one it pushes a tiny grid, one record and three buoys through a step and reads them back. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sitrk.h"

int main(void)
{
    sitrk_t *h = NULL;
    printf("version %d\n", sitrk_version());
    int rc = sitrk_create(&h, 0);
    if (rc != SITRK_OK) {
        printf("create failed (%d): %s\n", rc, sitrk_last_error(NULL));
        return rc == SITRK_EHIP ? 3 : 1;               /* 3 = no device: the expected outcome on a CPU box */
    }
    enum { N = 16 };
    static double Yf[N * N], Xf[N * N], Yu[N * N], Xu[N * N], Yv[N * N], Xv[N * N];
    static int8_t tmask[N * N];
    /* the record fields live on the heap: they are scribbled over and freed right after the push (include/sitrk.h:
     * "no host pointer is retained after a call returns" must hold for a C caller, not only behind Python's sync) */
    float *u = malloc(sizeof(float) * N * N), *v = malloc(sizeof(float) * N * N), *sic = malloc(sizeof(float) * N * N);
    if (!u || !v || !sic) return 1;
    for (int j = 0; j < N; j++)
        for (int i = 0; i < N; i++) {
            int k = j * N + i;
            Yf[k] = 4.0 * (j + 0.5); Xf[k] = 4.0 * (i + 0.5);
            Yu[k] = 4.0 * j;         Xu[k] = 4.0 * (i + 0.5);
            Yv[k] = 4.0 * (j + 0.5); Xv[k] = 4.0 * i;
            tmask[k] = 1; u[k] = 0.5f; v[k] = -0.25f; sic[k] = 1.0f;
        }
#define CHK(call) do { rc = (call); if (rc) { printf("%s -> %d: %s\n", #call, rc, sitrk_last_error(h)); return 1; } } while (0)
    CHK(sitrk_set_grid(h, N, N, Yf, Xf, Yu, Xu, Yv, Xv, tmask));
    CHK(sitrk_set_params(h, 3600.0, 1, 0.1));
    CHK(sitrk_alloc_records(h, 3, SITRK_F32));
    CHK(sitrk_push_record(h, 0, u, v, sic));
    for (int k = 0; k < N * N; k++) { u[k] = 1e30f; v[k] = -1e30f; sic[k] = 0.0f; }   /* poison, then free: the copy is the library's */
    free(u); free(v); free(sic);
    /* a second record read straight into the library's pinned staging, no intermediate copy */
    {
        void *pu, *pv, *ps;
        CHK(sitrk_stage_acquire(h, N, &pu, &pv, &ps));
        if (sitrk_stage_acquire(h, N, &pu, &pv, &ps) != SITRK_EINVAL) { printf("a second acquire without submit must be refused\n"); return 1; }
        /* a reader that fails half-way gives the buffer back; the next acquire hands out the same one */
        { void *qu = pu, *qv, *qs; CHK(sitrk_stage_release(h)); CHK(sitrk_stage_release(h)); CHK(sitrk_stage_acquire(h, N, &pu, &qv, &qs));
          if (pu != qu) { printf("release must hand the same buffer out again\n"); return 1; } pv = qv; ps = qs; }
        for (int k = 0; k < N * N; k++) { ((float *)pu)[k] = 0.5f; ((float *)pv)[k] = -0.25f; ((float *)ps)[k] = 1.0f; }
        CHK(sitrk_stage_submit(h, 1, 0, N));
        /* rows 2..N-3 only into the third slot: stepping with it needs the buoys' band */
        CHK(sitrk_stage_acquire(h, N - 4, &pu, &pv, &ps));
        for (int k = 0; k < (N - 4) * N; k++) { ((float *)pu)[k] = 0.5f; ((float *)pv)[k] = -0.25f; ((float *)ps)[k] = 1.0f; }
        CHK(sitrk_stage_submit(h, 2, 2, N - 2));
    }
    double yx[6] = {20.0, 20.0, 30.5, 33.0, 41.0, 27.9};     /* cells (5,5) (8,8) (10,7) */
    int32_t ji[6] = {5, 5, 8, 8, 10, 7};
    CHK(sitrk_set_buoys(h, 3, yx, ji, NULL, NULL));
    CHK(sitrk_step(h, 0, 0));
    double out[6]; int32_t jo[6]; int8_t alive[3]; int32_t kr[3];
    CHK(sitrk_fetch(h, out, jo, alive, kr));
    {
        /* same start, record from the staged slot: identical; the partly uploaded slot is refused until the band is known */
        double o2[6];
        CHK(sitrk_set_buoys(h, 3, yx, ji, NULL, NULL));
        if (sitrk_step(h, 2, 0) != SITRK_EINVAL) { printf("a partly uploaded slot must be refused before sitrk_buoy_rows\n"); return 1; }
        CHK(sitrk_step(h, 1, 0));
        CHK(sitrk_fetch(h, o2, NULL, NULL, NULL));
        if (memcmp(o2, out, sizeof(out))) { printf("staged record differs from pushed record\n"); return 1; }
        int32_t jmin, jmax;
        CHK(sitrk_set_buoys(h, 3, yx, ji, NULL, NULL));
        CHK(sitrk_buoy_rows(h, &jmin, &jmax));
        if (jmin != 5 || jmax != 10) { printf("buoy rows %d..%d\n", jmin, jmax); return 1; }
        CHK(sitrk_step(h, 2, 0));                                  /* rows 2..13 cover [5-2, 10+3) */
        CHK(sitrk_fetch(h, o2, NULL, NULL, NULL));
        if (memcmp(o2, out, sizeof(out))) { printf("row-band record differs from whole record\n"); return 1; }
        /* round 4: the same record as a BOX (rows x columns the buoys can touch), out of whole fields (ld = N), into the third slot */
        {
            int32_t imin, imax, age = -1;
            float *fu = (float *)malloc(sizeof(float) * N * N), *fv = (float *)malloc(sizeof(float) * N * N), *fs = (float *)malloc(sizeof(float) * N * N);
            for (int k = 0; k < N * N; k++) { fu[k] = 0.5f; fv[k] = -0.25f; fs[k] = 1.0f; }
            CHK(sitrk_set_buoys(h, 3, yx, ji, NULL, NULL));
            CHK(sitrk_buoy_box(h, &jmin, &jmax, &imin, &imax));
            if (jmin != 5 || jmax != 10 || imin != 5 || imax != 8) { printf("buoy box %d..%d x %d..%d\n", jmin, jmax, imin, imax); return 1; }
            const int j0 = jmin - 2, j1 = jmax + 3, i0 = imin - 2, i1 = imax + 3;
            CHK(sitrk_push_record_box(h, 2, j0, j1, i0, i1, fu + j0 * N + i0, fv + j0 * N + i0, fs + j0 * N + i0, N));
            memset(fu, 0xff, sizeof(float) * N * N); free(fu); free(fv); free(fs);          /* the caller's arrays are its own again */
            CHK(sitrk_buoy_box_begin(h));                              /* (the asynchronous form: queued, collected later) */
            CHK(sitrk_step(h, 2, 0));
            CHK(sitrk_buoy_box_end(h, &jmin, &jmax, &imin, &imax, &age));
            if (age != 1 || jmin != 5 || imax != 8) { printf("async box: age %d rows from %d columns to %d\n", age, jmin, imax); return 1; }
            CHK(sitrk_fetch(h, o2, NULL, NULL, NULL));
            if (memcmp(o2, out, sizeof(out))) { printf("box record differs from whole record\n"); return 1; }
            if (sitrk_step(h, 2, 1) != SITRK_EINVAL) { printf("a box one cell too narrow must be refused\n"); return 1; }
            CHK(sitrk_commit_records_box(h, 0, 2, 2, N - 2, 2, N - 2));   /* slots 0 and 1 hold whole records: commit a box of them, one launch */
            CHK(sitrk_commit_records_box_async(h, 0, 2, 2, N - 2, 2, N - 2));
            CHK(sitrk_sync(h));
        }
        int64_t nf = -1, nr = -1, ns = -1;
        CHK(sitrk_launch_stats(h, 1, &nf, &nr, &ns));
        if (nf != 0 || nr != 0 || ns != 4) { printf("launch stats %lld %lld %lld\n", (long long)nf, (long long)nr, (long long)ns); return 1; }
    }
    for (int p = 0; p < 3; p++) {
        /* dx = 0.5*3600/1000 = 1.8 km, dy = -0.9 km, exactly as the reference computes them */
        double ey = yx[2 * p] + (-0.25 * 3600.0) / 1000.0, ex = yx[2 * p + 1] + (0.5 * 3600.0) / 1000.0;
        printf("buoy %d: (%.17g, %.17g) cell (%d,%d) alive %d\n", p, out[2 * p], out[2 * p + 1], jo[2 * p], jo[2 * p + 1], alive[p]);
        if (out[2 * p] != ey || out[2 * p + 1] != ex) { printf("unexpected position\n"); return 1; }
    }
    int64_t nalive = -1;
    CHK(sitrk_count_alive(h, &nalive));
    CHK(sitrk_destroy(h));
    printf("ok (%lld alive)\n", (long long)nalive);
    return 0;
}
