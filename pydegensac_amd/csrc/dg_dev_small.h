/* Device-side small dense fp64 routines of the LO-RANSAC / DEGENSAC hot path (gfx950).
 *
 * Same arithmetic, statement for statement, as the reference's scalar code (file:line cited per
 * function, paths relative to /root/reference/src/pydegensac) so that a wave-uniform "lane 0"
 * section reproduces the reference's rounding exactly (compile with -ffp-contract=off).  LAPACK
 * dsyev is restated from the published netlib 3.12 algorithm (dsytd2/dorg2l/dsteqr), dgesvd 3x3 by a
 * one-sided Jacobi SVD, glibc rand() by its TYPE_3 additive-feedback generator.
 *
 * Execution model: these run on ONE lane of a workgroup (all other lanes wait at a barrier), so the
 * larger temporaries are function-local __shared__ arrays (LDS, statically indexed by the compiler
 * as ds_read/ds_write) instead of per-lane scratch memory.  They must therefore never be entered by
 * two lanes of the same workgroup at once.
 */
#ifndef DG_DEV_SMALL_H
#define DG_DEV_SMALL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dg_mat3.h"
#include "dg_crmath.h"

#define DG_FN static __device__ __forceinline__
#define DG_BIG static __device__ __noinline__
#define DG_LDS __shared__

/* ------------------------------------------------------------------------------------------------
 * glibc TYPE_3 random()/srandom() (r[i] = r[i-31] + r[i-3], 310 outputs discarded after seeding).
 * rand() == random(), srand() == srandom() share this one state.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int32_t r[34]; int f, b; } dg_rng;   /* ring of 31 words: r[0..30] */

DG_FN void dg_srand(dg_rng *g, unsigned seed)
{
    int i; int32_t word; long hi, lo;
    if (seed == 0) seed = 1;
    g->r[0] = (int32_t)seed;
    for (i = 1; i < 31; i++) {
        word = g->r[i - 1];
        hi = word / 127773; lo = word % 127773;
        word = (int32_t)(16807 * lo - 2836 * hi);
        if (word < 0) word += 2147483647;
        g->r[i] = word;
    }
    g->f = 3; g->b = 0;
    for (i = 0; i < 310; i++) {
        g->r[g->f] = (int32_t)((uint32_t)g->r[g->f] + (uint32_t)g->r[g->b]);
        if (++g->f >= 31) g->f = 0;
        if (++g->b >= 31) g->b = 0;
    }
}

DG_FN int dg_rand(dg_rng *g)
{
    uint32_t v = (uint32_t)g->r[g->f] + (uint32_t)g->r[g->b];
    g->r[g->f] = (int32_t)v;
    if (++g->f >= 31) g->f = 0;
    if (++g->b >= 31) g->b = 0;
    return (int)(v >> 1);
}

/* ------------------------------------------------------------------------------------------------
 * Symmetric 9x9 eigensolver = LAPACK dsyev("V","U") as called by lap_eig (degensac/lapwrap.c:67-96).
 * a: n x n symmetric, column-major (== row-major); on exit the columns (column-major: a[j*n+i] is
 * component i of eigenvector j) are the eigenvectors, w ascending.  n <= 9.  Returns 0 on success.
 * ---------------------------------------------------------------------------------------------- */
#define DG_EPS    1.1102230246251565e-16      /* dlamch('E') */
#define DG_SAFMIN 2.2250738585072014e-308     /* dlamch('S') */

DG_FN double dg_sign(double a, double b) { a = fabs(a); return (b >= 0. && !(b == 0. && signbit(b))) ? a : -a; }

DG_FN double dg_lapy2(double x, double y)
{
    double xa = fabs(x), ya = fabs(y), w = xa > ya ? xa : ya, z = xa > ya ? ya : xa, q;
    if (z == 0.) return w;
    q = z / w; return w * sqrt(1. + q*q);
}

DG_FN void dg_lartg(double f, double g, double *c, double *s, double *r)   /* LAPACK 3.10+ dlartg (unscaled range) */
{
    double f1 = fabs(f), g1 = fabs(g), d;
    if (g == 0.) { *c = 1.; *s = 0.; *r = f; }
    else if (f == 0.) { *c = 0.; *s = dg_sign(1., g); *r = g1; }
    else { d = sqrt(f*f + g*g); *c = f1 / d; *r = dg_sign(d, f); *s = g / *r; }
}

/* dlartg for the wave eigen-solver's rotation chain (the latency-critical scalar recurrence of dsteqr).
 * Same values as dg_lartg bit for bit: the compiler's IEEE fp64 sqrt / divide are v_rsq_f64 / v_rcp_f64 + Newton
 * steps + a Markstein correction wrapped in scaling and fix-up code for denormal or overflowing operands; with
 * 1e-140 < |f|,|g| < 1e140 none of that can trigger, and the two quotients share one refined reciprocal of d.
 * Outside that range (never seen in practice) the plain path runs. */
__device__ __forceinline__ void dg_lartg_fast(double f, double g, double *c, double *s, double *r)
{
    const double f1 = fabs(f), g1 = fabs(g);
    if (g == 0.) { *c = 1.; *s = 0.; *r = f; return; }
    if (f == 0.) { *c = 0.; *s = dg_sign(1., g); *r = g1; return; }
    const bool ok = f1 > 1e-140 && f1 < 1e140 && g1 > 1e-140 && g1 < 1e140;
    if (!ok) { dg_lartg(f, g, c, s, r); return; }
    const double x = f*f + g*g;
    /* sqrt(x) */
    double y = __builtin_amdgcn_rsq(x);
    double sg = x * y, sh = y * 0.5;
    double sr = __builtin_fma(-sh, sg, 0.5);
    sg = __builtin_fma(sg, sr, sg); sh = __builtin_fma(sh, sr, sh);
    double sd = __builtin_fma(-sg, sg, x); sg = __builtin_fma(sd, sh, sg);
    sd = __builtin_fma(-sg, sg, x);        sg = __builtin_fma(sd, sh, sg);
    const double d = sg;
    /* 1/d refined, then f1/d and g/d */
    double ry = __builtin_amdgcn_rcp(d);
    double re = __builtin_fma(-d, ry, 1.0); ry = __builtin_fma(ry, re, ry);
    re = __builtin_fma(-d, ry, 1.0);        ry = __builtin_fma(ry, re, ry);
    double q0 = f1 * ry, rr = __builtin_fma(-d, q0, f1);
    *c = __builtin_fma(rr, ry, q0);
    q0 = g * ry; rr = __builtin_fma(-d, q0, g);
    const double sq = __builtin_fma(rr, ry, q0);
    *r = dg_sign(d, f);
    *s = f < 0. ? -sq : sq;
}

DG_FN void dg_laev2(double a, double b, double c, double *rt1, double *rt2, double *cs1, double *sn1)
{
    double sm = a + c, df = a - c, adf = fabs(df), tb = b + b, ab = fabs(tb);
    double acmx, acmn, rt, cs, ct, tn, acs; int sgn1, sgn2;
    if (fabs(a) > fabs(c)) { acmx = a; acmn = c; } else { acmx = c; acmn = a; }
    if (adf > ab) { double q = ab/adf; rt = adf * sqrt(1. + q*q); }
    else if (adf < ab) { double q = adf/ab; rt = ab * sqrt(1. + q*q); }
    else rt = ab * sqrt(2.);
    if (sm < 0.) { *rt1 = .5*(sm - rt); sgn1 = -1; *rt2 = (acmx / *rt1)*acmn - (b / *rt1)*b; }
    else if (sm > 0.) { *rt1 = .5*(sm + rt); sgn1 = 1; *rt2 = (acmx / *rt1)*acmn - (b / *rt1)*b; }
    else { *rt1 = .5*rt; *rt2 = -.5*rt; sgn1 = 1; }
    if (df >= 0.) { cs = df + rt; sgn2 = 1; } else { cs = df - rt; sgn2 = -1; }
    acs = fabs(cs);
    if (acs > ab) { ct = -tb/cs; *sn1 = 1./sqrt(1. + ct*ct); *cs1 = ct * *sn1; }
    else {
        if (ab == 0.) { *cs1 = 1.; *sn1 = 0.; }
        else { tn = -cs/tb; *cs1 = 1./sqrt(1. + tn*tn); *sn1 = tn * *cs1; }
    }
    if (sgn1 == sgn2) { tn = *cs1; *cs1 = -*sn1; *sn1 = tn; }
}

/* z is n x n column-major (z[j*n+i] = Z(i,j)); applies rotations to columns j0..j0+cnt-1 */
/* ------------------------------------------------------------------------------------------------
 * Wave-cooperative dsyev: the same arithmetic as dg_eig_sym (same per-element operations, same
 * summation order inside every dot product), but the independent loops run in different lanes of ONE
 * wave: rows of dsymv, elements of the rank-2 update, columns of the reflector application, rows of the
 * plane rotations.  Must be called by all 64 lanes of a single wave with identical arguments; the matrix
 * lives in LDS.  n = 9.
 * ---------------------------------------------------------------------------------------------- */
/* broadcast of a double from lane l (l must be wave-uniform) */
static __device__ __forceinline__ double dg_rdl_d(double v, int l)
{
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
/* The fences are compiler-level only (at wavefront scope the back end emits no wait: it relies on a wave's accesses to one memory reaching
 * it in issue order).  That order does NOT hold between the two paths to LDS — a FLAT store through a generic pointer and a ds_read of the
 * same address: the ds_read can overtake the store (round 6: dg_u2f_small_w read the null vector its callee had just written through a
 * generic pointer, right behind the return) — so the barrier waits for the wave's outstanding memory operations explicitly. */
#define DG_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
                        __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
/* The same for data that travels through LDS only: the fences name the local address space, so the compiler waits for the wave's LDS
 * operations and leaves its global loads and stores in flight.  DG_WSYNC's fences cover every address space, i.e. s_waitcnt vmcnt(0): inside
 * a loop that keeps the NEXT step's points (or terms) in flight that wait is a full L2 round trip per step (round 6: the passes of the
 * local optimisations, the streamed least squares, the LDS-fed sums).  Use only where nothing written to global memory before the
 * barrier is read behind it before a full DG_WSYNC. */
#define DG_WSYNC_LDS() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local"); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
                            __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local"); } while (0)

#ifdef DG_EIG_TIMING
static __device__ long long dg_eig_ticks[4];
#define DG_ET(i) do { long long t_ = wall_clock64(); if (lane == 0) dg_eig_ticks[i] += t_ - t_et; t_et = t_; } while (0)
#else
#define DG_ET(i)
#endif
/* dg_steqr9.h on the device: the reciprocal-sharing dlartg, wave-uniform conditions through a ballot, and the three
 * searches for a negligible subdiagonal entry with one candidate per lane (lanes 8..63 repeat lanes 0..7) */
#define DG_STEQR_FN static __device__ __forceinline__
#define DG_STEQR_PTR __attribute__((address_space(3))) double *
#define DG_STEQR_LARTG(f, g, c, s, r) dg_lartg_fast((f), (g), (c), (s), (r))
#define DG_STEQR_ANY(cond) (__ballot(cond) != 0ull)
#define DG_STEQR_FIND_SPLIT(d, e, l1, m) do { const int i_ = lane & 7; const double ae_ = fabs((e)[i_]); \
        const bool c_ = i_ >= (l1) && (ae_ == 0. || ae_ <= (sqrt(fabs((d)[i_])) * sqrt(fabs((d)[i_+1]))) * DG_EPS); \
        const unsigned long long bm_ = __ballot(c_) & 0xffull; (m) = bm_ ? __ffsll((long long)bm_) - 1 : 8; } while (0)
#define DG_STEQR_FIND_QL(d, e, l, lend, m) do { const int i_ = lane & 7; double t2_ = fabs((e)[i_]); t2_ *= t2_; \
        const bool c_ = i_ >= (l) && i_ < (lend) && t2_ <= ((DG_EPS*DG_EPS) * fabs((d)[i_])) * fabs((d)[i_+1]) + DG_SAFMIN; \
        const unsigned long long bm_ = __ballot(c_) & 0xffull; (m) = bm_ ? __ffsll((long long)bm_) - 1 : (lend); } while (0)
#define DG_STEQR_FIND_QR(d, e, l, lend, m) do { const int i_ = lane & 7; double t2_ = fabs((e)[i_]); t2_ *= t2_; \
        const bool c_ = i_ >= (lend) && i_ < (l) && t2_ <= ((DG_EPS*DG_EPS) * fabs((d)[i_+1])) * fabs((d)[i_]) + DG_SAFMIN; \
        const unsigned long long bm_ = __ballot(c_) & 0xffull; (m) = bm_ ? 64 - __clzll((long long)bm_) : (lend); } while (0)
#ifdef DG_EIG_TIMING
static __device__ long long dg_steqr_ticks[2], dg_steqr_tq;
#define DG_STEQR_T(i) do { long long t__ = wall_clock64(); if (lane == 0) { dg_steqr_ticks[i] += t__ - dg_steqr_tq; dg_steqr_tq = t__; } } while (0)
#endif
#include "dg_steqr9.h"
struct dg_eig_ws { double d[9], e[9], tau[9], work[18]; };
static __device__ __noinline__ int dg_eig_sym_wave(double *a, double *w, int lane, dg_eig_ws *ews)
{
    const int n = 9;
#ifdef DG_EIG_TIMING
    long long t_et = wall_clock64();
#endif
    double *d = ews->d, *e = ews->e, *tau = ews->tau;
    int i, j, k, ii;
#define A_(r,c) a[(c)*n + (r)]
    /* ---- dsytd2, UPLO='U' ----
     * Lane r (< 9) keeps row r of the symmetric matrix in nine registers (both triangles, kept in step), every index
     * below is a compile-time constant after unrolling, and values cross lanes through v_readlane: no LDS round trips
     * inside the eight Householder steps.  Sums run in the reference's element order; the rank-2 update evaluates
     * A(kr,jc) - v_kr*tau_jc - tau_kr*v_jc with (kr, jc) = (min, max) of (row, column) for both triangles. */
    {
        double R[9];
#pragma unroll
        for (int cc = 0; cc < 9; cc++) R[cc] = lane < n ? A_(lane, cc) : 0.;
#pragma unroll
        for (int ih = n - 2; ih >= 0; ih--) {
            const int c1 = ih + 1;                                     /* column holding the reflector */
            double alpha = dg_rdl_d(R[c1], ih), xnorm = 0., taui, beta, sc = 0.;
            { const double sq = R[c1] * R[c1];
#pragma unroll
              for (int kk = 0; kk < ih; kk++) xnorm += dg_rdl_d(sq, kk); }
            xnorm = sqrt(xnorm);
            if (xnorm == 0.) taui = 0.;
            else { beta = -dg_sign(dg_lapy2(alpha, xnorm), alpha); taui = (beta - alpha) / beta; sc = 1. / (alpha - beta); alpha = beta; }
            if (xnorm != 0. && lane < ih) R[c1] *= sc;
            if (lane == 0) e[ih] = alpha;
            if (taui != 0.) {
                if (lane == ih) R[c1] = 1.;
                double tl = 0.;                                        /* tau_lane */
                { double sum = 0.;
#pragma unroll
                  for (int jj = 0; jj <= ih; jj++) sum += R[jj] * dg_rdl_d(R[c1], jj);
                  if (lane <= ih) tl = taui * sum; }
                double dot = 0.;
                { const double pr = tl * R[c1];
#pragma unroll
                  for (int kk = 0; kk <= ih; kk++) dot += dg_rdl_d(pr, kk); }
                const double al = -.5 * taui * dot;
                if (lane <= ih) tl += al * R[c1];
                const double vl = R[c1];
#pragma unroll
                for (int cc = 0; cc <= ih; cc++) {
                    const double vc = dg_rdl_d(vl, cc), tc = dg_rdl_d(tl, cc);
                    const double up = R[cc] - vl * tc - tl * vc;       /* row <= column: kr = lane, jc = cc */
                    const double lo = R[cc] - vc * tl - tc * vl;       /* row >  column: kr = cc,   jc = lane */
                    if (lane <= ih) R[cc] = lane <= cc ? up : lo;
                }
                if (lane == ih) R[c1] = alpha;
            }
            if (lane == c1) d[c1] = R[c1];
            if (lane == 0) tau[ih] = taui;
        }
        if (lane == 0) d[0] = R[0];
        if (lane < n) {
#pragma unroll
            for (int cc = 0; cc < 9; cc++) A_(lane, cc) = R[cc];
        }
        DG_WSYNC();
    }
    DG_ET(0);
    /* ---- dorgtr 'U' + dorg2l(n-1, n-1, n-1) ----
     * Lane c (< 9) takes column c of the shifted matrix into nine registers (reflector vectors one column left, unit
     * last row/column), applies H(0..7) with the reflector column broadcast by v_readlane from lane ii, and writes the
     * finished Q back once. */
    {
        double Cq[9];
#pragma unroll
        for (int r = 0; r < 9; r++) {
            double v = 0.;
            if (lane < n) {
                if (r == n - 1 || lane == n - 1) v = (r == n - 1 && lane == n - 1) ? 1. : 0.;
                else v = r < lane ? A_(r, lane + 1) : A_(r, lane);
            }
            Cq[r] = v;
        }
        double tq[8];
#pragma unroll
        for (int q = 0; q < 8; q++) tq[q] = tau[q];
        DG_WSYNC();
#pragma unroll
        for (int iq = 0; iq < n - 1; iq++) {
            if (lane == iq) Cq[iq] = 1.;
            double wv[9];
#pragma unroll
            for (int kk = 0; kk <= iq; kk++) wv[kk] = dg_rdl_d(Cq[kk], iq);
            if (lane < iq) {
                double sum = 0.;
#pragma unroll
                for (int kk = 0; kk <= iq; kk++) sum += Cq[kk] * wv[kk];
                sum *= tq[iq];
#pragma unroll
                for (int kk = 0; kk <= iq; kk++) Cq[kk] -= sum * wv[kk];
            }
            if (lane == iq) {
#pragma unroll
                for (int kk = 0; kk < iq; kk++) Cq[kk] *= -tq[iq];
                Cq[iq] = 1. - tq[iq];
#pragma unroll
                for (int kk = iq + 1; kk < n - 1; kk++) Cq[kk] = 0.;
            }
        }
        if (lane < n) {
#pragma unroll
            for (int r = 0; r < 9; r++) A_(r, lane) = Cq[r];
        }
        DG_WSYNC();
    }
    DG_ET(1);
    /* ---- dsteqr 'V' ----
     * dg_steqr9.h: d / e stay in LDS (every lane runs the scalar recurrence and stores the same values), lane r < 9
     * rotates row r of Z in place (a[c*9 + r]); lanes 9..63 repeat rows 0..8, same values to the same addresses. */
    {
        double p;
        DG_WSYNC();
#ifdef DG_EIG_TIMING
        if (lane == 0) dg_steqr_tq = wall_clock64();
#endif
        const int info = dg_steqr9((DG_STEQR_PTR)d, (DG_STEQR_PTR)e, (DG_STEQR_PTR)(a + (lane % 9)), 9, lane);
        const int jtot = info ? n * 30 : 0, nmaxit = n * 30;
        DG_STEQR_T(0);
        DG_WSYNC();
        DG_ET(2);
        /* dsteqr ends with an ascending selection sort; every caller only consumes the smallest pair (column 0,
         * or the first minimum of w[]), which the sort's first pass already puts in place: run that pass only */
        for (ii = 1; ii < 2; ii++) {
            i = ii - 1; k = i; p = d[i];
            for (j = ii; j < n; j++) if (d[j] < p) { k = j; p = d[j]; }
            DG_WSYNC();
            if (k != i) {
                double dk = d[i];
                if (lane == 0) { d[k] = dk; d[i] = p; }
                if (lane < n) { double t = A_(lane, i); A_(lane, i) = A_(lane, k); A_(lane, k) = t; }
            }
            DG_WSYNC();
        }
        if (lane < n) w[lane] = d[lane];
        DG_WSYNC();
        DG_ET(3);
        return jtot >= nmaxit ? 1 : 0;
    }
#undef A_
}

/* wave-cooperative form of dg_svd_lastcol_9x8: identical per-element arithmetic, no LDS inside the sweep.
 * The 9x8 matrix is held twice in registers — lane r (< 9) owns row r (Rw[0..7]), lane c (< 8) owns column c
 * (Cl[0..8]) — so the column reflector finds its column locally in lane i and the row reflector its row locally
 * in lane i; scalars and reflector entries travel by v_readlane with compile-time lane numbers, and every update is
 * applied to both copies with the same operands (bitwise equal).  All 64 lanes of one wave. */
static __device__ __noinline__ void dg_svd_lastcol_9x8_wave(double *a /* LDS 9x8 row-major */, double *col /* LDS 9 */, int lane)
{
    const int m = 9, n = 8;
    double Rw[8], Cl[9];
#pragma unroll
    for (int c = 0; c < 8; c++) Rw[c] = lane < m ? a[lane * n + c] : 0.;
#pragma unroll
    for (int r = 0; r < 9; r++) Cl[r] = lane < n ? a[r * n + lane] : 0.;
#pragma unroll
    for (int i = 0; i < n; i++) {
        const int mm = m - i, nm = n - 1 - i;
        /* ---- column reflector on a[i..8][i] ---- */
        {
            double sl = 0.;
#pragma unroll
            for (int j = 0; j < mm; j++) sl += Cl[i + j] * Cl[i + j];
            double s = dg_rdl_d(sl, i), sv = 0.;
            if (s > 0.) {
                const double p0 = dg_rdl_d(Cl[i], i);
                double h = sqrt(s); if (p0 < 0.) h = -h;
                s += p0 * h; s = 1. / s; const double t = 1. / (p0 + h);
                sv = 1. + fabs(p0 / h);
                double wv[9];
                wv[0] = p0 + h;
#pragma unroll
                for (int j = 1; j < mm; j++) wv[j] = dg_rdl_d(Cl[i + j], i);
                /* column copy: lanes k > i */
                double rl = 0.;
#pragma unroll
                for (int j = 0; j < mm; j++) rl += wv[j] * Cl[i + j];
                rl *= s;
                if (lane > i && lane < n) {
#pragma unroll
                    for (int j = 0; j < mm; j++) Cl[i + j] -= rl * wv[j];
                }
                /* row copy: lanes i..8, columns k > i, with r_k from column lane k */
                const double wl = lane == i ? p0 + h : Rw[i];
#pragma unroll
                for (int k = i + 1; k < n; k++) {
                    const double rk = dg_rdl_d(rl, k);
                    if (lane >= i && lane < m) Rw[k] -= rk * wl;
                }
                /* scaled reflector below the diagonal */
                if (lane == i) {
#pragma unroll
                    for (int j = 1; j < mm; j++) Cl[i + j] = t * wv[j];
                }
                if (lane > i && lane < m) Rw[i] = t * wl;
            }
            if (lane == i) { Cl[i] = sv; Rw[i] = sv; }
        }
        /* ---- row reflector on a[i][i+1..7] ---- */
        if (nm > 1) {
            double sl = 0.;
#pragma unroll
            for (int j = 0; j < nm; j++) sl += Rw[i + 1 + j] * Rw[i + 1 + j];
            double s = dg_rdl_d(sl, i), sv = 0.;
            if (s > 0.) {
                const double q0 = dg_rdl_d(Rw[i + 1], i);
                double h = sqrt(s); if (q0 < 0.) h = -h;
                sv = 1. + fabs(q0 / h);
                s += q0 * h; s = 1. / s; const double t = 1. / (q0 + h);
                double pv[8];
                pv[0] = q0 + h;
#pragma unroll
                for (int j = 1; j < nm; j++) pv[j] = dg_rdl_d(Rw[i + 1 + j], i);
                /* row copy: lanes R > i */
                double rl = 0.;
#pragma unroll
                for (int j = 0; j < nm; j++) rl += pv[j] * Rw[i + 1 + j];
                rl *= s;
                if (lane > i && lane < m) {
#pragma unroll
                    for (int j = 0; j < nm; j++) Rw[i + 1 + j] -= rl * pv[j];
                }
                /* column copy: lanes i+1..7, rows R > i, with r_R from row lane R */
                const double pl = lane == i + 1 ? q0 + h : Cl[i];
#pragma unroll
                for (int R = i + 1; R < m; R++) {
                    const double rR = dg_rdl_d(rl, R);
                    if (lane > i && lane < n) Cl[R] -= rR * pl;
                }
                if (lane == i) {
#pragma unroll
                    for (int j = 1; j < nm; j++) Rw[i + 1 + j] *= t;
                }
                if (lane > i + 1 && lane < n) Cl[i] *= t;
            }
            if (lane == i) Rw[i + 1] = sv;
            if (lane == i + 1) Cl[i] = sv;
        }
    }
    /* ldumat restricted to column 8 (sequential recurrence, every lane computes it; lane 0 stores) */
    {
        double c[9];
#pragma unroll
        for (int i = 0; i < 9; i++) c[i] = 0.;
        c[8] = 1.;
#pragma unroll
        for (int i = n - 1; i >= 0; --i) {
            const int mm2 = n - i;             /* rows below row i: 9 - 1 - i */
            const double p0 = dg_rdl_d(Cl[i], i);
            if (p0 != 0.) {
                double av[9];
#pragma unroll
                for (int j = 0; j < mm2; j++) av[j] = dg_rdl_d(Cl[i + 1 + j], i);
                double s = 0.;
#pragma unroll
                for (int j = 0; j < mm2; j++) s += av[j] * c[i + 1 + j];
                s *= p0;
#pragma unroll
                for (int j = 0; j < mm2; j++) c[i + 1 + j] -= s * av[j];
                c[i] = -s;
            } else c[i] = 0.;
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 9; i++) col[i] = c[i];
        }
    }
    DG_WSYNC();
}

/* ------------------------------------------------------------------------------------------------
 * degensac/utools.c
 * ---------------------------------------------------------------------------------------------- */
DG_FN double dg_det3(const double *A)                      /* utools.c:196-202 */
{
    double r;
    r = (A[0]*A[4]*A[8] + A[2]*A[3]*A[7] + A[1]*A[5]*A[6]);
    r -= (A[2]*A[4]*A[6] + A[0]*A[5]*A[7] + A[1]*A[3]*A[8]);
    return r;
}

/* (pointer types are template parameters: a caller that knows its buffers are LDS passes address-space-qualified pointers and gets ds_
 * instructions instead of FLAT ones) */
template <class PF, class PA>
DG_FN void dg_denormF(PF F, PA A1, PA A2)   /* utools.c:53-70 */
{
    double r, x, y;
    r = A2[0]; x = A2[1]; y = A2[2];
    F[6] += x * F[0] + y * F[3];
    F[7] += x * F[1] + y * F[4];
    F[8] += x * F[2] + y * F[5];
    F[0] *= r; F[1] *= r; F[2] *= r;
    F[3] *= r; F[4] *= r; F[5] *= r;
    r = A1[0]; x = A1[1]; y = A1[2];
    F[2] += x * F[0] + y * F[1];
    F[5] += x * F[3] + y * F[4];
    F[8] += x * F[6] + y * F[7];
    F[0] *= r; F[3] *= r; F[6] *= r;
    F[1] *= r; F[4] *= r; F[7] *= r;
}

DG_FN void dg_denormH(double *F, const double *A1, const double *A2)   /* utools.c:72-92 */
{
    double r, x, y; int i;
    r = A2[0]; x = A2[1]; y = A2[2];
    F[6] += x * F[0] + y * F[3];
    F[7] += x * F[1] + y * F[4];
    F[8] += x * F[2] + y * F[5];
    F[0] *= r; F[1] *= r; F[2] *= r;
    F[3] *= r; F[4] *= r; F[5] *= r;
    r = 1 / A1[0]; x = -A1[1] * r; y = -A1[2] * r;
    for (i = 0; i < 9; i += 3) {
        F[i]   = r * F[i]   + x * F[i+2];
        F[i+1] = r * F[i+1] + y * F[i+2];
    }
}

/* ------------------------------------------------------------------------------------------------
 * degensac/Ftools.c (7-point solver pieces, rank-2 projection, orientation test)
 * ---------------------------------------------------------------------------------------------- */
/* Ftools.c:39-81  cubic det(x A + (1-x) B); B is overwritten with A-B.  The expression trees are
 * written out exactly as in the reference so every rounding matches. */
DG_FN void dg_slcm(const double *A, double *B, double *p)
{
#define a11 A[0]
#define a12 A[1]
#define a13 A[2]
#define a21 A[3]
#define a22 A[4]
#define a23 A[5]
#define a31 A[6]
#define a32 A[7]
#define a33 A[8]
#define b11 B[0]
#define b12 B[1]
#define b13 B[2]
#define b21 B[3]
#define b22 B[4]
#define b23 B[5]
#define b31 B[6]
#define b32 B[7]
#define b33 B[8]
    int i;
    p[0] = -(b13*b22*b31) + b12*b23*b31 + b13*b21*b32 -
            b11*b23*b32 - b12*b21*b33 + b11*b22*b33;

    p[1] = -(a33*b12*b21) + a32*b13*b21 + a33*b11*b22 -
            a31*b13*b22 - a32*b11*b23 + a31*b12*b23 +
            a23*b12*b31 - a22*b13*b31 - a13*b22*b31 +
            3*b13*b22*b31 + a12*b23*b31 - 3*b12*b23*b31 -
            a23*b11*b32 + a21*b13*b32 + a13*b21*b32 -
            3*b13*b21*b32 - a11*b23*b32 + 3*b11*b23*b32 +
            (a22*b11 - a21*b12 - a12*b21 + 3*b12*b21 + a11*b22 -
             3*b11*b22)*b33;

    p[2] = -(a21*a33*b12) + a21*a32*b13 +
            a13*a32*b21 - a12*a33*b21 + 2*a33*b12*b21 -
            2*a32*b13*b21 - a13*a31*b22 + a11*a33*b22 -
            2*a33*b11*b22 + 2*a31*b13*b22 + a12*a31*b23 -
            a11*a32*b23 + 2*a32*b11*b23 - 2*a31*b12*b23 +
            2*a13*b22*b31 - 3*b13*b22*b31 - 2*a12*b23*b31 +
            3*b12*b23*b31 + a13*a21*b32 - 2*a21*b13*b32 -
            2*a13*b21*b32 + 3*b13*b21*b32 + 2*a11*b23*b32 -
            3*b11*b23*b32 + a23*
            (-(a32*b11) + a31*b12 + a12*b31 - 2*b12*b31 -
             a11*b32 + 2*b11*b32) +
            (-(a12*a21) + 2*a21*b12 + 2*a12*b21 - 3*b12*b21 -
             2*a11*b22 + 3*b11*b22)*b33 +
            a22*(a33*b11 - a31*b13 - a13*b31 + 2*b13*b31 +
                 a11*b33 - 2*b11*b33);

    for (i = 0; i < 9; i++) B[i] = A[i] - B[i];

    p[3] = -(b13*b22*b31) + b12*b23*b31 + b13*b21*b32 -
            b11*b23*b32 - b12*b21*b33 + b11*b22*b33;
#undef a11
#undef a12
#undef a13
#undef a21
#undef a22
#undef a23
#undef a31
#undef a32
#undef a33
#undef b11
#undef b12
#undef b13
#undef b21
#undef b22
#undef b23
#undef b31
#undef b32
#undef b33
}

#define DG_PIT 1.0471975511965967             /* Ftools.c:14 */

/* pow / acos / cos: the correctly rounded values (dg_crmath.h), which is what the host's libm returns in 99.8-99.9 % of its
 * calls; the device library's own (1-2 ulp) put one root in 29 off the host's bits */
DG_FN int dg_rroots3(const double *po, double *r)         /* Ftools.c:251-298 */
{
    double b, c, b2, bt, v, e;
    double p, q, D, A, cosphi, phit, R, _2R;
    b = po[1] / po[0];
    c = po[2] / po[0];
    b2 = b*b;
    bt = b/3;
    p = (3*c - b2) / 9;
    q = ((2 * b2 * b)/27 - b*c/3 + po[3]/po[0]) / 2;
    D = q*q + p*p*p;
    if (D > 0) {
        A = sqrt(D) - q;
        if (A > 0) { v = dg_cr_pow13(A); *r = v - p/v - bt; }
        else       { v = dg_cr_pow13(-A); *r = p/v - v - bt; }
        return 1;
    } else {
        if (q > 0) e = 1; else e = -1;
        R = e * sqrt(-p);
        _2R = R * 2;
        cosphi = q / (R*R*R);
        if (cosphi > 1) cosphi = 1; else if (cosphi < -1) cosphi = -1;
        phit = dg_cr_acos(cosphi) / 3;
        r[0] = -_2R * dg_cr_cos(phit) - bt;
        r[1] =  _2R * dg_cr_cos(DG_PIT - phit) - bt;
        r[2] =  _2R * dg_cr_cos(DG_PIT + phit) - bt;
        return 3;
    }
}

/* Ftools.c:330-348 singulF: project F onto rank 2 (zero the smallest singular value).  The
 * reference calls LAPACK dgesvd on the 3x3; U*diag(s1,s2,0)*V^T does not depend on any SVD sign or
 * ordering convention, so any accurate SVD gives the same matrix to rounding.  Here: one-sided
 * (Hestenes) Jacobi on the columns of F, which is accurate also for the tiny singular values
 * (CCMATH svduv is not: absolute 1e-15 deflation threshold).  F = sum_k a_k v_k^T after rotation;
 * drop the term with the smallest |a_k|. */
template <class PF>
static __device__ __forceinline__ void dg_singulF(PF F)
{
    /* fully unrolled: A, V live in registers (static indices only) */
    double A[9], V[9] = {1,0,0, 0,1,0, 0,0,1};
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 9; i++) { A[i] = F[i]; bad = bad || isnan(A[i]) || isinf(A[i]); }
    if (bad) { for (int k = 0; k < 9; k++) F[k] = (k % 4 == 0) ? 1. : 0.; return; }
    for (int sweep = 0; sweep < 40; sweep++) {
        int rotated = 0;
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int q = p + 1; q < 3; q++) {
                double alpha = 0., beta = 0., gamma = 0.;
#pragma unroll
                for (int i = 0; i < 3; i++) { alpha += A[3*i+p]*A[3*i+p]; beta += A[3*i+q]*A[3*i+q]; gamma += A[3*i+p]*A[3*i+q]; }
                if (!(gamma == 0. || fabs(gamma) <= 1e-15 * sqrt(alpha * beta))) {
                    rotated = 1;
                    double zeta = (beta - alpha) / (2. * gamma);
                    double t = (zeta >= 0. ? 1. : -1.) / (fabs(zeta) + sqrt(1. + zeta*zeta));
                    double c = 1. / sqrt(1. + t*t), sn = c * t;
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        double ap = A[3*i+p], aq = A[3*i+q], vp = V[3*i+p], vq = V[3*i+q];
                        A[3*i+p] = c*ap - sn*aq; A[3*i+q] = sn*ap + c*aq;
                        V[3*i+p] = c*vp - sn*vq; V[3*i+q] = sn*vp + c*vq;
                    }
                }
            }
        if (!rotated) break;
    }
    double n0 = A[0]*A[0] + A[3]*A[3] + A[6]*A[6], n1 = A[1]*A[1] + A[4]*A[4] + A[7]*A[7], n2 = A[2]*A[2] + A[5]*A[5] + A[8]*A[8];
    int k = 0; double nk = n0;
    if (n1 < nk) { k = 1; nk = n1; }
    if (n2 < nk) { k = 2; }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int p = 0; p < 3; p++) {
            double acc = 0.;
#pragma unroll
            for (int q = 0; q < 3; q++) { double term = A[3*i+q] * V[3*p+q]; acc = (q != k) ? acc + term : acc; }
            F[3*i+p] = acc;
        }
}

#define DG_XEPS 1.9984e-15                    /* Ftools.c:461 */

DG_FN void dg_crossprod_st(double *out, const double *a, const double *b, int st)   /* utools.c:187-193 */
{
    int st2 = 2 * st;
    out[0] = a[st]*b[st2] - a[st2]*b[st];
    out[1] = a[st2]*b[0]  - a[0]*b[st2];
    out[2] = a[0]*b[st]   - a[st]*b[0];
}

DG_FN void dg_epipole(double *ec, const double *F)        /* Ftools.c:463-470; crossprod = crossprod_st(...,1) */
{
    int i;
    dg_crossprod_st(ec, F, F + 6, 1);
    for (i = 0; i < 3; i++)
        if ((ec[i] > DG_XEPS) || (ec[i] < -DG_XEPS)) return;
    dg_crossprod_st(ec, F + 3, F + 6, 1);
}

DG_FN double dg_getorisig(const double *F, const double *ec, const double *u)   /* Ftools.c:472-479 */
{
    double s1, s2;
    s1 = F[0]*u[3] + F[3]*u[4] + F[6]*u[5];
    s2 = ec[1]*u[2] - ec[2]*u[1];
    return s1 * s2;
}

/* ------------------------------------------------------------------------------------------------
 * degensac/Htools.c pinvJ (:135-159)
 * ---------------------------------------------------------------------------------------------- */
DG_FN void dg_pinvJ(double a, double b, double c, double d, double e, double *pJ)
{
    double a2 = a*a, b2 = b*b, c2 = c*c, d2 = d*d, e2 = e*e;
    double c2pd2 = c2 + d2, ab = a*b, de = d*e;
    double Q = c * (c2pd2 + e2);
    double N; int i;
    pJ[0] = -b * de + a * (c2 + e2);
    pJ[1] = b * c2pd2 - a * de;
    pJ[2] = Q;
    pJ[3] = -c * (a*d + b*e);
    pJ[4] = d * (b2 + c2) - ab * e;
    pJ[5] = -ab * d + e * (a2 + c2);
    pJ[6] = pJ[3];
    pJ[7] = c * (a2 + b2 + c2);
    N = a * pJ[0] + b * pJ[1] + c * pJ[2];
    for (i = 0; i < 8; i++) pJ[i] /= N;
}

/* degensac/rtools.c:202-225 */
DG_FN int dg_nsamples(int ninl, int ptNum, int samsiz, double conf)
{
    double a = 1, b = 1; int i;
    for (i = 0; i < samsiz; i++) { a *= ninl - i; b *= ptNum - i; }
    a = a / b;
    if (a < 2.2204e-16) return 1000000;
    a = 1 - a;
    if (a < 2.2204e-16) return 1;
    b = log(1 - conf) / log(a);
    if (b > 1000000) return 1000000;
    return (int)ceil(b);
}

/* degensac/rtools.c:228-236 */
DG_FN double dg_truncQuad(double epsilon, double thr)
{
    if (thr == 0) return 0;
    if (epsilon >= thr*9/4) return 0;
    return 1 - (epsilon / (thr*9/4));
}

#include "dg_eig2.h"

#endif /* DG_DEV_SMALL_H */
