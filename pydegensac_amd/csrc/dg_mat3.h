/* Small dense kernels of the DEGENSAC branch, the symmetric homography metrics and the degenerate-sample fallback of
 * the minimal solvers (3x3 inverse, 3x3 right singular vectors, Hdetect, 9-column null space), written for one GPU lane:
 * every matrix lives in named scalars / statically indexed locals (registers), every loop has a compile-time trip
 * count, nothing is addressed through a run-time index.
 *
 * The reference obtains these numbers from CCMATH (matutls/minv.c, matutls/svduv.c with ldvmat.c and qrbdv.c) and the
 * masks downstream must match bit for bit, so the SEQUENCE OF FLOATING-POINT OPERATIONS per output number is the one
 * those routines perform for n = 3 (cited where it matters: pivot rule, accumulation order, the implicit-shift QR
 * sweep); the code itself is a specialisation: fixed size, no work arrays, and only the factor the caller consumes
 * (DegUtils.c:84-161 uses one column of the right singular vectors, so the left factor is never formed).
 * Compile with -ffp-contract=off.  Host-compilable (tests/test_mat3_cpu.py checks it against oracle/_ref).
 */
#ifndef DG_MAT3_H
#define DG_MAT3_H
#include <math.h>

#if defined(__HIPCC__)
#define DG_HD __host__ __device__ __forceinline__
#else
#define DG_HD static inline
#endif

/* rows r and s of a row-major 3x3 change places */
DG_HD void dg3_swap_rows(double *a, int r, int s)
{
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double x0 = a[c], x1 = a[3 + c], x2 = a[6 + c];
        const double vr = r == 0 ? x0 : (r == 1 ? x1 : x2), vs = s == 0 ? x0 : (s == 1 ? x1 : x2);
        a[c]     = r == 0 ? vs : (s == 0 ? vr : x0);
        a[3 + c] = r == 1 ? vs : (s == 1 ? vr : x1);
        a[6 + c] = r == 2 ? vs : (s == 2 ? vr : x2);
    }
}
DG_HD void dg3_swap_cols(double *a, int r, int s)
{
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double x0 = a[3*c], x1 = a[3*c + 1], x2 = a[3*c + 2];
        const double vr = r == 0 ? x0 : (r == 1 ? x1 : x2), vs = s == 0 ? x0 : (s == 1 ? x1 : x2);
        a[3*c]     = r == 0 ? vs : (s == 0 ? vr : x0);
        a[3*c + 1] = r == 1 ? vs : (s == 1 ? vr : x1);
        a[3*c + 2] = r == 2 ? vs : (s == 2 ? vr : x2);
    }
}

/* In-place inverse of a row-major 3x3 matrix: column-wise LU with row pivoting on the largest magnitude of the
 * remaining column (ties keep the upper row), singular when a pivot is below 1e-15 of the largest pivot so far,
 * then inverse of the unit-lower and upper factors, their product, and the row exchanges undone as column exchanges
 * in reverse (operation order of matutls/minv.c).  Returns -1 when singular (the matrix is then partly factored,
 * like the reference leaves it), else 0. */
DG_HD int dg_inv3(double *a)
{
    const double zr = 1.e-15;
    double tq = 0., s, t, c0, c1, c2;
    int p0, p1;
    /* column 0 */
    s = fabs(a[0]); p0 = 0;
    t = fabs(a[3]); if (t > s) { s = t; p0 = 1; }
    t = fabs(a[6]); if (t > s) { s = t; p0 = 2; }
    tq = tq > s ? tq : s;
    if (s < zr * tq) return -1;
    if (p0 != 0) dg3_swap_rows(a, 0, p0);
    t = 1. / a[0]; a[3] *= t; a[6] *= t; a[0] = t;
    /* column 1: eliminate with the rows above, pivot among rows 1..2 */
    c0 = a[1]; c1 = a[4]; c2 = a[7];
    t = 0.; t += a[3] * c0; c1 -= t;
    t = 0.; t += a[6] * c0; c2 -= t;
    a[1] = c0; a[4] = c1; a[7] = c2;
    s = fabs(a[4]); p1 = 1;
    t = fabs(a[7]); if (t > s) { s = t; p1 = 2; }
    tq = tq > s ? tq : s;
    if (s < zr * tq) return -1;
    if (p1 != 1) dg3_swap_rows(a, 1, 2);
    t = 1. / a[4]; a[7] *= t; a[4] = t;
    /* column 2 */
    c0 = a[2]; c1 = a[5]; c2 = a[8];
    t = 0.; t += a[3] * c0; c1 -= t;
    t = 0.; t += a[6] * c0; t += a[7] * c1; c2 -= t;
    a[2] = c0; a[5] = c1; a[8] = c2;
    s = fabs(a[8]);
    tq = tq > s ? tq : s;
    if (s < zr * tq) return -1;
    t = 1. / a[8]; a[8] = t;
    /* the diagonal now holds reciprocal pivots: scale the strict upper triangle by its column's */
    a[1] *= a[4];
    a[2] *= a[8]; a[5] *= a[8];
    /* inverse of the upper factor, column by column */
    c0 = a[1];
    t = 0.; t -= a[0] * c0; c0 = t;
    a[1] = c0;
    c0 = a[2]; c1 = a[5];
    t = 0.; t -= a[0] * c0; t -= a[1] * c1; c0 = t;
    t = 0.; t -= a[4] * c1; c1 = t;
    a[2] = c0; a[5] = c1;
    /* inverse of the unit lower factor, bottom up */
    a[7] = -a[7];
    c0 = a[3]; c1 = a[6];
    t = -c1; t -= a[7] * c0; c1 = t;
    t = -c0; c0 = t;
    a[3] = c0; a[6] = c1;
    /* product of the two inverses, columns 0 and 1 */
    c0 = a[0]; c1 = a[3]; c2 = a[6];
    { double r0, r1, r2;
      t = c0; t += a[1] * c1; t += a[2] * c2; r0 = t;
      t = 0.; t += a[4] * c1; t += a[5] * c2; r1 = t;
      t = 0.; t += a[8] * c2; r2 = t;
      a[0] = r0; a[3] = r1; a[6] = r2; }
    c0 = a[1]; c1 = a[4]; c2 = a[7];
    { double r0, r1, r2;
      t = c0; t += a[2] * c2; r0 = t;
      t = c1; t += a[5] * c2; r1 = t;
      t = 0.; t += a[8] * c2; r2 = t;
      a[1] = r0; a[4] = r1; a[7] = r2; }
    /* undo the row exchanges (as column exchanges, last first) */
    if (p1 != 1) dg3_swap_cols(a, 1, p1);
    if (p0 != 0) dg3_swap_cols(a, 0, p0);
    return 0;
}

/* Right singular vectors of a row-major 3x3 matrix the way matutls/svduv.c produces them: Householder
 * bidiagonalisation (left reflector on column 0, right reflector on row 0, left reflector on column 1), the right
 * reflector accumulated into V (ldvmat.c), then implicit-shift QR sweeps on the bidiagonal (qrbdv.c) whose right
 * rotations act on the columns of V, and the sign of every column with a negative singular value flipped.  The
 * singular values are NOT sorted.  v: row-major V; d: the three singular values.  `a` is destroyed. */
DG_HD void dg_svd3_right(double *a, double *v, double *d)
{
    double e0 = 0., e1 = 0., w0, w1, w2, s, h, r, t, sv;
    /* left reflector on column 0 */
    sv = h = 0.;
    w0 = a[0]; w1 = a[3]; w2 = a[6];
    s = 0.; s += w0 * w0; s += w1 * w1; s += w2 * w2;
    if (s > 0.) {
        h = sqrt(s); if (a[0] < 0.) h = -h;
        s += a[0] * h; s = 1. / s; w0 += h; t = 1. / w0;
        sv = 1. + fabs(a[0] / h);
#pragma unroll
        for (int k = 1; k < 3; k++) {
            r = 0.; r += w0 * a[k]; r += w1 * a[3 + k]; r += w2 * a[6 + k];
            r *= s;
            a[k] -= r * w0; a[3 + k] -= r * w1; a[6 + k] -= r * w2;
        }
        a[3] = t * w1; a[6] = t * w2;
    }
    a[0] = sv; d[0] = -h;
    /* right reflector on row 0 (entries 1..2) */
    sv = h = 0.;
    s = 0.; s += a[1] * a[1]; s += a[2] * a[2];
    if (s > 0.) {
        h = sqrt(s); if (a[1] < 0.) h = -h;
        sv = 1. + fabs(a[1] / h);
        s += a[1] * h; s = 1. / s; a[1] += h; t = 1. / a[1];
#pragma unroll
        for (int row = 1; row < 3; row++) {
            r = 0.; r += a[1] * a[3*row + 1]; r += a[2] * a[3*row + 2];
            r *= s;
            a[3*row + 1] -= r * a[1]; a[3*row + 2] -= r * a[2];
        }
        a[2] *= t;
    }
    a[1] = sv; e0 = -h;
    /* left reflector on column 1 (rows 1..2) */
    sv = h = 0.;
    w0 = a[4]; w1 = a[7];
    s = 0.; s += w0 * w0; s += w1 * w1;
    if (s > 0.) {
        h = sqrt(s); if (a[4] < 0.) h = -h;
        s += a[4] * h; s = 1. / s; w0 += h; t = 1. / w0;
        sv = 1. + fabs(a[4] / h);
        r = 0.; r += w0 * a[5]; r += w1 * a[8];
        r *= s;
        a[5] -= r * w0; a[8] -= r * w1;
        a[7] = t * w1;
    }
    a[4] = sv; d[1] = -h;
    e1 = a[5];
    d[2] = a[8];
    /* V from the one right reflector */
    double v00 = 1., v01 = 0., v02 = 0., v10 = 0., v11, v12, v20 = 0., v21, v22 = 1.;
    if (a[1] != 0.) {
        h = a[1]; v11 = 1. - h;
        v21 = -h * a[2];
        s = 0.; s += v22 * a[2];
        s *= h;
        v22 -= s * a[2];
        v12 = -s;
    } else { v11 = 1.; v12 = 0.; v21 = 0.; }
    /* implicit-shift QR on the bidiagonal (d0 e0; d1 e1; d2), rotations applied to the columns of V */
    double d0 = d[0], d1 = d[1], d2 = d[2], e2 = 0.;
    double tol = fabs(d0);
    s = fabs(d1) + fabs(e0); if (s > tol) tol = s;
    s = fabs(d2) + fabs(e1); if (s > tol) tol = s;
    tol *= 1.e-15;
    int m = 3;
    for (int it = 0; m > 1 && it < 300; ++it) {
        /* split search from the bottom: k = first row of the active block */
        int k = 0; bool found = false;
#pragma unroll
        for (int kk = 2; kk >= 1; --kk) {
            if (found || kk > m - 1) continue;
            const double ek = kk == 2 ? e1 : e0, dk = kk == 2 ? d1 : d0;     /* em[kk-1], dm[kk-1] */
            if (fabs(ek) < tol) { k = kk; found = true; continue; }
            if (fabs(dk) < tol) {
                /* a negligible diagonal entry: chase its super-diagonal neighbour out of the block */
                double sn = 1., cs = 0., aa, bb, uu;
#pragma unroll
                for (int i = 1; i < 3; ++i) {
                    if (i < kk || i >= m) continue;
                    aa = sn * (i == 1 ? e0 : e1); bb = i == 1 ? d1 : d2;
                    if (i == 1) e0 *= cs; else e1 *= cs;
                    uu = sqrt(aa * aa + bb * bb);
                    if (i == 1) d1 = uu; else d2 = uu;
                    sn = -aa / uu; cs = bb / uu;
                }
                k = kk; found = true;
            }
        }
        double y = k == 0 ? d0 : (k == 1 ? d1 : d2), x = m == 3 ? d2 : d1, u = m == 3 ? e1 : e0;
        const double ekk = k == 0 ? e0 : (k == 1 ? e1 : e2);
        double aa = (y + x) * (y - x) - u * u, sn = y * ekk, bb = sn + sn, cs;
        u = sqrt(aa * aa + bb * bb);
        if (u != 0.) {
            cs = sqrt((u + aa) / (u + u));
            if (cs != 0.) sn /= (cs * u); else sn = 1.;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i < k || i >= m - 1) continue;
                bb = i == 0 ? e0 : e1;
                if (i > k) {                                   /* only i == 1, k == 0 */
                    aa = sn * e1; bb *= cs;
                    u = sqrt(x * x + aa * aa); e0 = u;
                    cs = x / u; sn = aa / u;
                }
                aa = cs * y + sn * bb; bb = cs * bb - sn * y;
                if (i == 0) {
                    double wv;
                    wv = cs * v00 + sn * v01; v01 = cs * v01 - sn * v00; v00 = wv;
                    wv = cs * v10 + sn * v11; v11 = cs * v11 - sn * v10; v10 = wv;
                    wv = cs * v20 + sn * v21; v21 = cs * v21 - sn * v20; v20 = wv;
                } else {
                    double wv;
                    wv = cs * v01 + sn * v02; v02 = cs * v02 - sn * v01; v01 = wv;
                    wv = cs * v11 + sn * v12; v12 = cs * v12 - sn * v11; v11 = wv;
                    wv = cs * v21 + sn * v22; v22 = cs * v22 - sn * v21; v21 = wv;
                }
                const double dn = i == 0 ? d1 : d2;
                sn *= dn; u = sqrt(aa * aa + sn * sn);
                if (i == 0) d0 = u; else d1 = u;
                y = cs * dn; cs = aa / u; sn /= u;
                x = cs * bb + sn * y; y = cs * y - sn * bb;
            }
        }
        if (m == 3) { e1 = x; d2 = y; } else { e0 = x; d1 = y; }
        if (fabs(x) < tol) --m;
        if (m == k + 1) --m;
    }
    if (d0 < 0.) { d0 = -d0; v00 = -v00; v10 = -v10; v20 = -v20; }
    if (d1 < 0.) { d1 = -d1; v01 = -v01; v11 = -v11; v21 = -v21; }
    if (d2 < 0.) { d2 = -d2; v02 = -v02; v12 = -v12; v22 = -v22; }
    d[0] = d0; d[1] = d1; d[2] = d2;
    v[0] = v00; v[1] = v01; v[2] = v02; v[3] = v10; v[4] = v11; v[5] = v12; v[6] = v20; v[7] = v21; v[8] = v22;
}

/* ---- DegUtils.c:84-161 Hdetect, one lane, registers only ---------------------------------------------
 * The homography induced by the plane through three of the seven sample correspondences and compatible with F:
 * H = A - e (M^-1 b)^T with A = [e]x F^T, e = the right singular vector svduv returns third, b from the three
 * correspondences (Hartley & Zisserman 13.6).  Every 3x3 product is spelled out: sums run over the inner index
 * in ascending order starting from 0.0, which is what the reference's mmul / rmmult calls do.  H is stored as the
 * reference stores it (H[i + 3 j] = (A - e m^T)[i][j]). */
DG_HD void dg_Hdetect(const double *F, const double (*u7)[4], const unsigned char *IDXS, double *H)
{
    double Ft[9], Fw[9], V[9], D[3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) { Ft[3*i + j] = F[3*j + i]; Fw[3*i + j] = F[3*i + j]; }
    dg_svd3_right(Fw, V, D);
    const double ec[3] = {V[2], V[5], V[8]};
    double Ex[9] = {0, -ec[2], ec[1], ec[2], 0, -ec[0], -ec[1], ec[0], 0};
    double A[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) { double z = 0.; z += Ex[3*i] * Ft[j]; z += Ex[3*i + 1] * Ft[3 + j]; z += Ex[3*i + 2] * Ft[6 + j]; A[3*i + j] = z; }
    double ua[3][3], ub[3][3], Aub[3][3], pa[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double *q = u7[IDXS[i]];
        ua[i][0] = q[0]; ua[i][1] = q[1]; ua[i][2] = 1.0; ub[i][0] = q[2]; ub[i][1] = q[3]; ub[i][2] = 1.0;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {                        /* A x2_i, then x1_i cross (A x2_i) */
#pragma unroll
        for (int r = 0; r < 3; r++) { double z = 0.; z += A[3*r] * ub[i][0]; z += A[3*r + 1] * ub[i][1]; z += A[3*r + 2] * ub[i][2]; Aub[i][r] = z; }
        const double *u = ua[i], *v = Aub[i];
        pa[i][0] = u[1]*v[2] - u[2]*v[1]; pa[i][1] = u[2]*v[0] - u[0]*v[2]; pa[i][2] = u[0]*v[1] - u[1]*v[0];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) Ex[i] *= -1;
    double pb[3][3], b[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {                        /* (-[e]x) x1_i */
#pragma unroll
        for (int r = 0; r < 3; r++) { double z = 0.; z += Ex[3*r] * ua[i][0]; z += Ex[3*r + 1] * ua[i][1]; z += Ex[3*r + 2] * ua[i][2]; pb[i][r] = z; }
        b[i] = (pa[i][0]*pb[i][0] + pa[i][1]*pb[i][1] + pa[i][2]*pb[i][2]) / (pb[i][0]*pb[i][0] + pb[i][1]*pb[i][1] + pb[i][2]*pb[i][2]);
    }
    double M[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) M[3*i + j] = ub[i][j];          /* rows = x2_i^T */
    const int sing = dg_inv3(M);
    double mv[3];
#pragma unroll
    for (int r = 0; r < 3; r++) { double z = 0.; z += M[3*r] * b[0]; z += M[3*r + 1] * b[1]; z += M[3*r + 2] * b[2]; mv[r] = z; }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) { double z = 0.; z += ec[i] * mv[j]; H[i + j*3] = A[i*3 + j] - z; }
    if (isnan(*H) || isinf(*H) || sing) { H[1] = H[2] = H[3] = H[5] = H[6] = H[7] = 0; H[0] = H[4] = H[8] = 1; }
}


/* Null space of a ROWS x 9 system (row-major M, ROWS <= 9, destroyed) by Gauss-Jordan elimination with row
 * pivoting, for the samples the register-unrolled solvers (dg_gj7 / dg_gj8) hand back because some column has no
 * usable pivot.  The elimination order is the reference's (utools.c:97-167: columns left to right, pivot = largest
 * magnitude of the column among the rows not yet used, columns below 1e-12 are free, pivot row scaled, then the
 * rows above, then the rows below), so the basis vectors are bit for bit the reference's; the bookkeeping is not:
 * pivot columns and free columns are both visited in ascending order, so a 9-bit mask replaces the index buffers,
 * and the 9 - ROWS zero rows the reference pads with (they can never hold a pivot) are not stored.  Writes the first
 * KMAX basis vectors to ns[k*9 + 0..8] and returns the nullity.  One lane at a time per M. */
template <int ROWS, int KMAX>
DG_HD int dg_null9(double *M, double *ns)
{
    const double tol = 1e-12;
    unsigned freemask = 0; int rank = 0;
    for (int j = 0; j < 9; j++) {
        double best = rank < ROWS ? fabs(M[9*rank + j]) : 0.; int at = rank;
        for (int k = rank + 1; k < ROWS; k++) { const double t = fabs(M[9*k + j]); if (best < t) { best = t; at = k; } }
        if (best < tol) {
            freemask |= 1u << j;
            for (int k = rank; k < ROWS; k++) M[9*k + j] = 0;
            continue;
        }
        for (int c = j; c < 9; c++) { const double t = M[9*rank + c]; M[9*rank + c] = M[9*at + c]; M[9*at + c] = t; }
        const double piv = M[9*rank + j];
        for (int c = j; c < 9; c++) M[9*rank + c] /= piv;
        for (int k = 0; k < rank; k++) { const double f = -M[9*k + j]; for (int c = j; c < 9; c++) M[9*k + c] += f * M[9*rank + c]; }
        for (int k = rank + 1; k < ROWS; k++) { const double f = M[9*k + j]; for (int c = j; c < 9; c++) M[9*k + c] -= f * M[9*rank + c]; }
        rank++;
    }
    /* basis vector of the k-th free column jf: 1 at jf, 0 at the other free columns, minus column jf of the reduced
     * rows at the pivot columns (row l belongs to the l-th pivot column) */
    int k = 0;
    for (int jf = 0; jf < 9 && k < KMAX; jf++) {
        if (!((freemask >> jf) & 1u)) continue;
        int l = 0;
        for (int c = 0; c < 9; c++) {
            if ((freemask >> c) & 1u) ns[9*k + c] = (c == jf) ? 1 : 0;
            else { ns[9*k + c] = -M[9*l + jf]; l++; }
        }
        k++;
    }
    return __builtin_popcount(freemask);
}

#endif /* DG_MAT3_H */
