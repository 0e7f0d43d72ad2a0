/* The 7-point solver of single samples outside the chunk pipeline: one problem per lane in registers (events of the commit), and the last real
 * root's model for the legacy drivers' symmetric check (exp_ranF.c:1196-1203).
 * Part of the fundamental-matrix kernel: included by dg_kernel_f_main.h, in this order, after dg_kernel_f.h and dg_score_tiles.h. */
#ifndef DG_F_SOLVE7_H
#define DG_F_SOLVE7_H

/* One 7-point problem per lane, registers only (own register allocation: not inlined into the driver).
 * ids: the 7 drawn ids in draw order.  Writes up to 3 models (9 doubles each) to out[0..27), packs their
 * root indices (2 bits each) into *rix and returns the number of valid models, or -1 when the null space
 * of the 7x9 system is not 2-dimensional (exp_ranF.c:1355-1358). */
__device__ __noinline__ int dg_solve7_lane(const dg_pt *P, const int *ids, double *out, unsigned *rix, double *wscr /* LDS, this wave's, >= 81 doubles */)
{
    dg_pt sp[7];
    double m[7][9];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        sp[i] = P[ids[i]];
        double a[3] = {sp[i].x1, sp[i].y1, 1.0}, b[3] = {sp[i].x2, sp[i].y2, 1.0};
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int l = 0; l < 3; l++) m[i][3*k+l] = b[k] * a[l];
    }
    double f1[9], f2[9];
    int ok = dg_gj7(m, f1, f2);
    /* degenerate samples only: a column without a usable pivot.  Those lanes take turns on the wave's LDS scratch
     * with the general elimination (no per-lane copy of the system in scratch memory) */
    for (unsigned long long need = __ballot(!ok); need; need &= need - 1) {
        if ((int)(threadIdx.x & 63) != __ffsll((long long)need) - 1) continue;
        for (int i = 0; i < 7; i++) {
            const double a[3] = {sp[i].x1, sp[i].y1, 1.0}, b[3] = {sp[i].x2, sp[i].y2, 1.0};
            for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++) wscr[9*i + 3*k + l] = b[k] * a[l];
        }
        if (dg_null9<7, 2>(wscr, wscr + 63) == 2) { for (int i = 0; i < 9; i++) { f1[i] = wscr[63 + i]; f2[i] = wscr[72 + i]; } ok = 1; }
        else ok = -1;
    }
    if (ok < 0) return -1;
    double poly[4], roots[3];
    dg_slcm(f1, f2, poly);
    int nsol = dg_rroots3(poly, roots);
    int nvalid = 0; unsigned rx = 0;
    for (int i = 0; i < nsol; i++) {
        double f[9];
#pragma unroll
        for (int j = 0; j < 9; j++) f[j] = f1[j] * roots[i] + f2[j] * (1 - roots[i]);
        if (!dg_ori_valid7(f, sp)) continue;
#pragma unroll
        for (int j = 0; j < 9; j++) out[9*nvalid + j] = f[j];
        rx |= (unsigned)i << (2*nvalid); nvalid++;
    }
    *rix = rx | ((unsigned)nsol << 8);       /* bits 8-9: the number of real roots */
    return nvalid;
}

/* The model the reference's driver holds in its local `f` after a sample whose roots were all computed: the LAST real root's
 * model, valid or not (exp_ranF.c:1365-1368 forms it before the orientation test).  One lane, for the legacy drivers' final
 * symmetric filter (exp_ranF.c:1196-1203).  Returns 0 when the null space is not two-dimensional. */
__device__ __noinline__ int dg_solve7_lastroot(const dg_pt *P, const int *ids, double *f /* 9 */, double *wscr /* LDS, >= 81 doubles */)
{
    dg_pt sp[7];
    double m[7][9];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        sp[i] = P[ids[i]];
        double a[3] = {sp[i].x1, sp[i].y1, 1.0}, b[3] = {sp[i].x2, sp[i].y2, 1.0};
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int l = 0; l < 3; l++) m[i][3*k+l] = b[k] * a[l];
    }
    double f1[9], f2[9];
    int ok = dg_gj7(m, f1, f2);
    if (!ok) {
        for (int i = 0; i < 7; i++) {
            const double a[3] = {sp[i].x1, sp[i].y1, 1.0}, b[3] = {sp[i].x2, sp[i].y2, 1.0};
            for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++) wscr[9*i + 3*k + l] = b[k] * a[l];
        }
        if (dg_null9<7, 2>(wscr, wscr + 63) != 2) return 0;
        for (int i = 0; i < 9; i++) { f1[i] = wscr[63 + i]; f2[i] = wscr[72 + i]; }
    }
    double poly[4], roots[3];
    dg_slcm(f1, f2, poly);
    const int nsol = dg_rroots3(poly, roots);
    if (nsol < 1) return 0;
    const double r = roots[nsol - 1];
#pragma unroll
    for (int j = 0; j < 9; j++) f[j] = f1[j] * r + f2[j] * (1 - r);
    return 1;
}

#endif /* DG_F_SOLVE7_H */
