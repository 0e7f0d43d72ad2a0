/* Per-point error metrics and the per-lane 7-point solver (device, gfx950, fp64, no contraction).
 * Operation order follows the reference exactly (file:line cited), so residuals are bit-identical
 * to the reference's for the same model bits. */
#ifndef DG_GEOM_H
#define DG_GEOM_H
#include "dg_dev_small.h"
#include "dg_wg.h"

/* ---- Ftools.c:83-101 (FDs), :124-146 (exFDs), :147-168 (FDsSym), :228-250 (exFDsSym) --------- */
#define DG_F_COMMON(F, x1, y1, x2, y2) \
    double rxc = F[0]*x2 + F[3]*y2 + F[6]; \
    double ryc = F[1]*x2 + F[4]*y2 + F[7]; \
    double rwc = F[2]*x2 + F[5]*y2 + F[8]; \
    double r   = (x1*rxc + y1*ryc + rwc); \
    double rx  = F[0]*x1 + F[1]*y1 + F[2]; \
    double ry  = F[3]*x1 + F[4]*y1 + F[5];

__device__ __forceinline__ double dg_FDs(const double *F, double x1, double y1, double x2, double y2)
{
    DG_F_COMMON(F, x1, y1, x2, y2);
    return r*r / (rxc*rxc + ryc*ryc + rx*rx + ry*ry);
}
__device__ __forceinline__ double dg_FDsSym(const double *F, double x1, double y1, double x2, double y2)
{
    DG_F_COMMON(F, x1, y1, x2, y2);
    double a = rxc*rxc + ryc*ryc, b = rx*rx + ry*ry;
    return r*r * (a + b) / (a*b);
}
/* exFDsSym computes r*r / ((a*b)/(a+b)): different rounding from FDsSym (Ftools.c:244-245) */
__device__ __forceinline__ double dg_exFDsSym(const double *F, double x1, double y1, double x2, double y2, double *w)
{
    DG_F_COMMON(F, x1, y1, x2, y2);
    double a = rxc*rxc + ryc*ryc, b = rx*rx + ry*ry;
    *w = (a*b) / (a + b);
    return r*r / *w;
}
/* exFDs weight: 1/sqrt(den) (Ftools.c:137-139); the residual equals FDs bit for bit */
__device__ __forceinline__ double dg_exFDs_w(const double *F, double x1, double y1, double x2, double y2)
{
    DG_F_COMMON(F, x1, y1, x2, y2);
    (void)r; (void)rwc;
    double w = rxc*rxc + ryc*ryc + rx*rx + ry*ry;
    return 1 / sqrt(w);
}

/* metric kinds */
enum { DG_K_FDS = 0, DG_K_FSYM = 1, DG_K_EXFSYM = 2 };

__device__ __forceinline__ double dg_Ferr(int kind, const double *F, const dg_pt &p)
{
    if (kind == DG_K_FDS) return dg_FDs(F, p.x1, p.y1, p.x2, p.y2);
    if (kind == DG_K_FSYM) return dg_FDsSym(F, p.x1, p.y1, p.x2, p.y2);
    double w; return dg_exFDsSym(F, p.x1, p.y1, p.x2, p.y2, &w);
}

/* ---- Htools.c:161-200 HDs (Sampson error of a homography), DLT rows of lin_hg (:20-58) -------- */
__device__ __forceinline__ double dg_HDs(const double *H, double u0, double u1, double u3, double u4)
{
    /* z0 = [s3,0,-s0*s3, s4,0,-s0*s4, s5,0,-s0*s5], z1 = [0,s3,-s1*s3, ...], s2 = s5 = 1 */
    double r1 = 0, r2 = 0;
    r1 += H[0] * u3;  r2 += H[0] * 0.0;
    r1 += H[1] * 0.0; r2 += H[1] * u3;
    r1 += H[2] * (-u0 * u3); r2 += H[2] * (-u1 * u3);
    r1 += H[3] * u4;  r2 += H[3] * 0.0;
    r1 += H[4] * 0.0; r2 += H[4] * u4;
    r1 += H[5] * (-u0 * u4); r2 += H[5] * (-u1 * u4);
    r1 += H[6] * 1.0; r2 += H[6] * 0.0;
    r1 += H[7] * 0.0; r2 += H[7] * 1.0;
    r1 += H[8] * (-u0 * 1.0); r2 += H[8] * (-u1 * 1.0);
    double a = H[0] - H[2] * u0;
    double b = H[3] - H[5] * u0;
    double c = -H[8] - H[2] * u3 - H[5] * u4;
    double d = H[1] - H[2] * u1;
    double e = H[4] - H[5] * u1;
    double pJ[8];
    dg_pinvJ(a, b, c, d, e, pJ);
    double p = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { double t = pJ[j] * r1 + pJ[j+4] * r2; p += t * t; }
    return p;
}

/* ---- the 7-point solver, one sample per lane, everything in registers ------------------------- */
/* Row i of the 7x9 system is the i-th DRAWN correspondence, entries u2_k*u1_l (lin_fm Ftools.c:15-37
 * + rsampleT rtools.c:74-92).  Gauss-Jordan with partial pivoting exactly as utools.c:97-167 for the
 * generic case (7 pivots in columns 0..6; rows 7,8 are zero so columns 7,8 are the free ones).
 * Returns 0 if some column j<7 has no pivot >= 1e-12 (the caller then runs the general routine). */
__device__ __forceinline__ int dg_gj7(double (&m)[7][9], double *f1, double *f2)
{
#pragma unroll
    for (int j = 0; j < 7; j++) {
        double pivot = fabs(m[j][j]); int mx = j;
#pragma unroll
        for (int k = j + 1; k < 7; k++) { double t = fabs(m[k][j]); if (pivot < t) { pivot = t; mx = k; } }
        if (pivot < 1e-12) return 0;
#pragma unroll
        for (int k = j + 1; k < 7; k++) {
            if (mx == k) {
#pragma unroll
                for (int l = j; l < 9; l++) { double t = m[j][l]; m[j][l] = m[k][l]; m[k][l] = t; }
            }
        }
        double pv = m[j][j];
#pragma unroll
        for (int l = j; l < 9; l++) m[j][l] /= pv;
#pragma unroll
        for (int k = 0; k < 7; k++) {
            if (k == j) continue;
            double pk = m[k][j];
#pragma unroll
            for (int l = j; l < 9; l++) m[k][l] -= pk * m[j][l];     /* == += (-pk)*.. for k<j (utools.c:140-152) */
        }
    }
#pragma unroll
    for (int l = 0; l < 7; l++) { f1[l] = -m[l][7]; f2[l] = -m[l][8]; }
    f1[7] = 1; f1[8] = 0; f2[7] = 0; f2[8] = 1;
    return 1;
}

/* Ftools.c:481-494 with the sample's own coordinates; s[.] are the 7 points in samidx order
 * (= reverse draw order: samidx[m] is draw 6-m, rtools.c:17-20) */
__device__ __forceinline__ int dg_ori_valid7(const double *F, const dg_pt *s /* draw order */)
{
    double ec[3];
    dg_epipole(ec, F);
    double sig1;
    {
        const dg_pt &p = s[6];
        double s1 = F[0]*p.x2 + F[3]*p.y2 + F[6]*1.0;
        double s2 = ec[1]*1.0 - ec[2]*p.y1;
        sig1 = s1 * s2;
    }
    int ok = 1;
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        const dg_pt &p = s[i];
        double s1 = F[0]*p.x2 + F[3]*p.y2 + F[6]*1.0;
        double s2 = ec[1]*1.0 - ec[2]*p.y1;
        double sig = s1 * s2;
        if (sig1 * sig < 0) ok = 0;
    }
    return ok;
}

#endif /* DG_GEOM_H */
