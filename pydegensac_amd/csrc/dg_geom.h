/* Per-point error metrics and the per-lane 7-point solver (device, gfx950, fp64, no contraction).
 * Operation order follows the reference exactly (file:line cited), so residuals are bit-identical
 * to the reference's for the same model bits. */
#ifndef DG_GEOM_H
#define DG_GEOM_H
#include "dg_dev_small.h"
#include "dg_mat3.h"
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
enum { DG_K_FDS = 0, DG_K_FSYM = 1, DG_K_EXFSYM = 2 };

/* Conservative, division-free test "could the residual of this point be below t?" used to bound a model's MSAC
 * gain from above before it is scored exactly (tb = t inflated by 1e-6, FMA allowed: nothing here reaches an
 * output).  Returns 1 for every point whose exact residual is < t, and for non-finite cases. */
__device__ __forceinline__ unsigned dg_Fbound(int kind, const double *F, const dg_pt &p, double tb)
{
    const double x1 = p.x1, y1 = p.y1, x2 = p.x2, y2 = p.y2;
    double rxc = __builtin_fma(F[0], x2, __builtin_fma(F[3], y2, F[6]));
    double ryc = __builtin_fma(F[1], x2, __builtin_fma(F[4], y2, F[7]));
    double rwc = __builtin_fma(F[2], x2, __builtin_fma(F[5], y2, F[8]));
    double r   = __builtin_fma(x1, rxc, __builtin_fma(y1, ryc, rwc));
    double rx  = __builtin_fma(F[0], x1, __builtin_fma(F[1], y1, F[2]));
    double ry  = __builtin_fma(F[3], x1, __builtin_fma(F[4], y1, F[5]));
    double a = __builtin_fma(rxc, rxc, ryc*ryc), b = __builtin_fma(rx, rx, ry*ry);
    double lhs, rhs;
    if (kind == DG_K_FDS) { lhs = r*r; rhs = tb * (a + b); }
    else                  { lhs = r*r * (a + b); rhs = tb * (a * b); }
    return !(lhs >= rhs) ? 1u : 0u;
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

/* Screening form of dg_HDs (main loop of the homography kernel): could this point's Sampson error be below tb?  The error is
 * r' (J J')^-1 r with J = [a b c 0; d e 0 c] (pinvJ is J' (J J')^-1 in closed form, Htools.c:135-159), i.e. q / det with
 * M = J J' = [m11 m12; m12 m22], q = m22 r1^2 - 2 m12 r1 r2 + m11 r2^2, det = m11 m22 - m12^2: no division, no pinvJ.
 * Conservative: tb carries a relative margin of 1e-6 (the two evaluations agree to 1e-9 while det > 1e-7 m11 m22), and a point
 * whose M is worse conditioned than that, or that yields a NaN, counts as a candidate.  So #(candidates) >= #(dg_HDs < t) >= J. */
__device__ __forceinline__ bool dg_HDs_maybe_below(const double *H, double u0, double u1, double u3, double u4, double tb)
{
    const double w = H[2] * u3 + H[5] * u4 + H[8];
    const double r1 = (H[0] * u3 + H[3] * u4 + H[6]) - u0 * w, r2 = (H[1] * u3 + H[4] * u4 + H[7]) - u1 * w;
    const double a = H[0] - H[2] * u0, b = H[3] - H[5] * u0, d = H[1] - H[2] * u1, e = H[4] - H[5] * u1;
    const double cc = w * w;
    const double m11 = a * a + b * b + cc, m22 = d * d + e * e + cc, m12 = a * d + b * e;
    const double det = m11 * m22 - m12 * m12, q = m22 * r1 * r1 - 2 * m12 * r1 * r2 + m11 * r2 * r2;
    const bool well = det > 1e-7 * (m11 * m22);
    return !(well && q > tb * det);
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


/* ---- Htools.c:202-370 / :428-605 / :649-816: the four symmetric transfer errors --------------------
 * Hinv = transpose of the stored H, H1 = minv(Hinv) (prepared once per model).  kind: 1 SymMaxSq,
 * 2 SymMax, 3 SymSumSq, 4 SymSum.  eps = the +1e-10 on the denominators: always for the Sum metrics and
 * for every i/idx variant, never for the plain HDsSymMax / HDsSymMaxSq (Htools.c:310-311, :352-353). */
struct dg_hsym { double Hinv[9], H1[9]; };
#define DG_HMAX(i,j) ( (i)<(j) ? (j):(i) )
__device__ __forceinline__ double dg_Hsym(const double *Hinv, const double *H1, double u0, double u1, double u3, double u4, int kind, int eps)
{
    double a = H1[6]*u0 + H1[7]*u1 + H1[8];
    double b = Hinv[6]*u3 + Hinv[7]*u4 + Hinv[8];
    if (eps) { a = a + 1e-10; b = b + 1e-10; }
    double xa = (H1[0]*u0 + H1[1]*u1 + H1[2]) / a;
    double ya = (H1[3]*u0 + H1[4]*u1 + H1[5]) / a;
    double xdiff = u3 - xa, ydiff = u4 - ya;
    double d1 = xdiff*xdiff + ydiff*ydiff;
    xa = (Hinv[0]*u3 + Hinv[1]*u4 + Hinv[2]) / b;
    ya = (Hinv[3]*u3 + Hinv[4]*u4 + Hinv[5]) / b;
    xdiff = u0 - xa; ydiff = u1 - ya;
    double d2 = xdiff*xdiff + ydiff*ydiff;
    if (kind == 1) return DG_HMAX(d1, d2);
    if (kind == 2) return sqrt(DG_HMAX(d1, d2));
    if (kind == 3) return d1 + d2;
    return sqrt(d1) + sqrt(d2);
}
/* HDsi / HDsidx (Htools.c:372-410, :607-647): residual rows from the ORIGINAL point's DLT rows, Jacobian
 * terms from the point set passed as u6 (the LAF-shifted points in the LAF checks) */
__device__ __forceinline__ double dg_HDs_mixed(const double *H, double o0, double o1, double o3, double o4,
                                               double u0, double u1, double u3, double u4)
{
    double r1 = 0, r2 = 0;
    r1 += H[0] * o3;  r2 += H[0] * 0.0;
    r1 += H[1] * 0.0; r2 += H[1] * o3;
    r1 += H[2] * (-o0 * o3); r2 += H[2] * (-o1 * o3);
    r1 += H[3] * o4;  r2 += H[3] * 0.0;
    r1 += H[4] * 0.0; r2 += H[4] * o4;
    r1 += H[5] * (-o0 * o4); r2 += H[5] * (-o1 * o4);
    r1 += H[6] * 1.0; r2 += H[6] * 0.0;
    r1 += H[7] * 0.0; r2 += H[7] * 1.0;
    r1 += H[8] * (-o0 * 1.0); r2 += H[8] * (-o1 * 1.0);
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
/* the metric selected by bindings.cpp:64-107 as a full pass (HDS1): kind 0 Sampson, 1..4 symmetric */
__device__ __forceinline__ double dg_Herr(int kind, const double *H, const double *Hinv, const double *H1, const dg_pt &p)
{
    if (kind == 0) return dg_HDs(H, p.x1, p.y1, p.x2, p.y2);
    return dg_Hsym(Hinv, H1, p.x1, p.y1, p.x2, p.y2, kind, kind >= 3);
}

/* exp_ranH.c:29-44 */
__device__ __forceinline__ int dg_HcloseToSingular(const double *h)
{
    double v = dg_det3(h), tol = h[8];
    if (tol == 0) { for (int i = 0; i < 9; ++i) tol += h[i]*h[i]; tol = sqrt(tol); tol *= 0.001; }
    tol = tol*tol*tol;
    return (fabs(v/tol) < 1e-2);
}

/* Htools.c:821-848 all_Hori_valid on the 4 sample points; s[] in DRAW order, samidx = reverse draw order */
__device__ __forceinline__ int dg_Hori_valid4(const dg_pt *s)
{
    const dg_pt &A = s[3], &B = s[2], &Cc = s[1], &D = s[0];
    double a[6] = {A.x1, A.y1, 1.0, A.x2, A.y2, 1.0}, b[6] = {B.x1, B.y1, 1.0, B.x2, B.y2, 1.0};
    double c[6] = {Cc.x1, Cc.y1, 1.0, Cc.x2, Cc.y2, 1.0}, d[6] = {D.x1, D.y1, 1.0, D.x2, D.y2, 1.0};
    double p[3], q[3];
    dg_crossprod_st(p, a, b, 1); dg_crossprod_st(q, a+3, b+3, 1);
    if ((p[0]*c[0]+p[1]*c[1]+p[2]*c[2])*(q[0]*c[3]+q[1]*c[4]+q[2]*c[5]) < 0) return 0;
    if ((p[0]*d[0]+p[1]*d[1]+p[2]*d[2])*(q[0]*d[3]+q[1]*d[4]+q[2]*d[5]) < 0) return 0;
    dg_crossprod_st(p, c, d, 1); dg_crossprod_st(q, c+3, d+3, 1);
    if ((p[0]*a[0]+p[1]*a[1]+p[2]*a[2])*(q[0]*a[3]+q[1]*a[4]+q[2]*a[5]) < 0) return 0;
    if ((p[0]*b[0]+p[1]*b[1]+p[2]*b[2])*(q[0]*b[3]+q[1]*b[4]+q[2]*b[5]) < 0) return 0;
    return 1;
}

/* 4-point DLT null vector, one sample per lane (utools.c:97-167 on the 8x9 system of multirsampleT,
 * rtools.c:136-156: rows 2i, 2i+1 = DLT rows of the i-th DRAWN point).  Generic case: pivots in
 * columns 0..7, column 8 free.  Returns 0 when some column j<8 has no pivot >= 1e-12. */
__device__ __forceinline__ int dg_gj8(double (&m)[8][9], double *h)
{
#pragma unroll
    for (int j = 0; j < 8; j++) {
        double pivot = fabs(m[j][j]); int mx = j;
#pragma unroll
        for (int k = j + 1; k < 8; k++) { double t = fabs(m[k][j]); if (pivot < t) { pivot = t; mx = k; } }
        if (pivot < 1e-12) return 0;
#pragma unroll
        for (int k = j + 1; k < 8; k++) {
            if (mx == k) {
#pragma unroll
                for (int l = j; l < 9; l++) { double t = m[j][l]; m[j][l] = m[k][l]; m[k][l] = t; }
            }
        }
        double pv = m[j][j];
#pragma unroll
        for (int l = j; l < 9; l++) m[j][l] /= pv;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (k == j) continue;
            double pk = m[k][j];
#pragma unroll
            for (int l = j; l < 9; l++) m[k][l] -= pk * m[j][l];
        }
    }
#pragma unroll
    for (int l = 0; l < 8; l++) h[l] = -m[l][8];
    h[8] = 1;
    return 1;
}

__device__ __forceinline__ void dg_hsym_prepare(const double *H, double *Hinv, double *H1)
{
    Hinv[0] = H[0]; Hinv[1] = H[3]; Hinv[2] = H[6];
    Hinv[3] = H[1]; Hinv[4] = H[4]; Hinv[5] = H[7];
    Hinv[6] = H[2]; Hinv[7] = H[5]; Hinv[8] = H[8];
    for (int i = 0; i < 9; i++) H1[i] = Hinv[i];
    dg_inv3(H1);
}

#endif /* DG_GEOM_H */
