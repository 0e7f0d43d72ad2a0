/* TEST INFRASTRUCTURE ONLY — CPU oracle, never linked into or called by the product library.
 *
 * Sequential CPU restatement of the reference's LO-RANSAC / DEGENSAC hot path, bug-compatible
 * (SURVEY.md 3.4/3.5), fp64, no FMA contraction.  Every function cites the reference file:line it
 * follows (paths relative to /root/reference/src/pydegensac/degensac).  Points are kept in the
 * reference's u[N][6] = (x1,y1,1,x2,y2,1) layout so the index arithmetic reads like the original.
 *
 * Parity pin: this file is checked in tests/ against (a) oracle/_ref = the unmodified reference
 * compiled from /root/reference (same seed => same sample count, LO count, scored-model count,
 * mask; model to ~1e-12, up to ~1e-6 in plane-dominated scenes whose final eigenproblem is ill-conditioned and
 * depends on the LAPACK build — tools/cpu_port_vs_ref.py) and (b) the golden fixtures under tests/golden generated
 * from that build.  The reference itself ships no golden vectors.
 *
 * Deliberate deviations from the reference (all in places where the reference has undefined
 * behaviour or depends on an external library):
 *   - RNG: explicit glibc TYPE_3 restatement instead of the process-global rand().
 *   - dsyev / dgesvd: restated algorithms (dg_small.h) instead of a linked LAPACK.
 *   - uninitialised memory (u2h 4-point path Htools.c:108-114, u2f with <8 points Ftools.c:371,
 *     model buffers bindings.cpp:110,321, `do_update` exp_ranF.c:1254): zero / explicit flag.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "dg_small.h"
#include "dg_oracle.h"

/* rtools.h:4-15 */
#define ITER_SAM 50
#define RAN_REP 10
#define ILSQ_ITERS 4
#define TC 4
#define MWM (9/4)          /* integer 2: rtools.h:38 */

typedef void (*fds_fn)(const double *, const double *, double *, int);
typedef void (*exfds_fn)(const double *, const double *, double *, double *, int);
typedef void (*fdsidx_fn)(const double *, const double *, double *, int, const int *, int);

typedef struct { uint32_t hash; int length; int iterID; } ht_entry;

/* optional trace hook for the tests: kind 0 = FDS1 pass, 1 = EXFDS1 pass (model passed in f) */
typedef void (*dg_trace_fn)(int kind, const double *f);
static dg_trace_fn g_trace = 0;
void dg_oracle_set_trace(dg_trace_fn fn) { g_trace = fn; }
#define TRACE(kind, f) do { if (g_trace) g_trace(kind, f); } while (0)
/* second hook: (tag, I, J) checkpoints inside the local optimisation, mirrored by the device's debug trace */
typedef void (*dg_trace2_fn)(int tag, int I, double J);
static dg_trace2_fn g_trace2 = 0;
void dg_oracle_set_trace2(dg_trace2_fn fn) { g_trace2 = fn; }
#define TRACE2(tag, I, J) do { if (g_trace2) g_trace2(tag, (int)(I), (double)(J)); } while (0)

typedef struct {
    dg_rng rng;
    ht_entry *ht; int ht_n, ht_cap;
    long long n_fds, n_exfds, n_hds, n_fds_direct;
} dg_ctx;

/* ------------------------------------------------------------------ rtools.c */
static int *randsubset(dg_ctx *c, int *pool, int max_sz, int siz)        /* rtools.c:25-39 */
{
    int i, j, q, s;
    for (i = 0; i < siz; i++) {
        s = dg_rand(&c->rng) % (max_sz - i);
        j = max_sz - i - 1;
        q = pool[s]; pool[s] = pool[j]; pool[j] = q;
    }
    return pool + max_sz - siz;
}

static dg_score inlidxs(const double *err, int len, double th, int *inl)   /* rtools.c:160-171 */
{
    int i; dg_score s = {0, 0, 0, 0};
    for (i = 0; i < len; ++i) {
        s.J += dg_truncQuad(err[i], th);
        if (err[i] <= th) { inl[s.I] = i; ++(s.I); }
    }
    return s;
}
static int scoreLess(dg_score s1, dg_score s2) { return s1.J < s2.J; }     /* rtools.c:238-249, SC_M */

/* ------------------------------------------------------------------ hash.c:49-96 */
static void htInsert(dg_ctx *c, uint32_t hash, int length, int iterID)
{
    if (c->ht_n == c->ht_cap) {
        c->ht_cap = c->ht_cap ? 2 * c->ht_cap : 256;
        c->ht = (ht_entry *)realloc(c->ht, sizeof(ht_entry) * c->ht_cap);
    }
    c->ht[c->ht_n].hash = hash; c->ht[c->ht_n].length = length; c->ht[c->ht_n].iterID = iterID;
    c->ht_n++;
}
/* chains are LIFO per bucket and equal hashes share a bucket, so "first match in the chain" is
 * the most recently inserted match */
static int htContains(dg_ctx *c, uint32_t hash, int length, int iterID)
{
    int i;
    for (i = c->ht_n - 1; i >= 0; i--)
        if (c->ht[i].hash == hash && c->ht[i].length == length && c->ht[i].iterID == iterID) return iterID;
    for (i = c->ht_n - 1; i >= 0; i--)
        if (c->ht[i].hash == hash && c->ht[i].length == length) return c->ht[i].iterID;
    return -1;
}

/* ------------------------------------------------------------------ Ftools.c error metrics */
#define F_COMMON \
    double rxc = F[0]*u[3] + F[3]*u[4] + F[6]; \
    double ryc = F[1]*u[3] + F[4]*u[4] + F[7]; \
    double rwc = F[2]*u[3] + F[5]*u[4] + F[8]; \
    double r = (u[0]*rxc + u[1]*ryc + rwc); \
    double rx = F[0]*u[0] + F[1]*u[1] + F[2]; \
    double ry = F[3]*u[0] + F[4]*u[1] + F[5];

static void FDs(const double *u, const double *F, double *p, int len)     /* Ftools.c:83-101 */
{
    int i;
    for (i = 0; i < len; i++, u += 6) { F_COMMON; p[i] = r*r / (rxc*rxc + ryc*ryc + rx*rx + ry*ry); }
}
static void FDsidx(const double *mu, const double *F, double *p, int len, const int *idx, int siz)  /* :103-122 */
{
    int mi; (void)len;
    for (mi = 0; mi < siz; mi++) { int i = idx[mi]; const double *u = mu + 6*i; F_COMMON; p[i] = r*r / (rxc*rxc + ryc*ryc + rx*rx + ry*ry); }
}
static void exFDs(const double *u, const double *F, double *p, double *w, int len)   /* :124-146 */
{
    int i;
    for (i = 0; i < len; i++, u += 6) {
        F_COMMON;
        w[i] = rxc*rxc + ryc*ryc + rx*rx + ry*ry;
        p[i] = r*r / w[i];
        w[i] = 1 / sqrt(w[i]);
    }
}
static void FDsSym(const double *u, const double *F, double *p, int len)  /* :147-168 */
{
    int i;
    for (i = 0; i < len; i++, u += 6) { F_COMMON; double a = rxc*rxc + ryc*ryc, b = rx*rx + ry*ry; p[i] = r*r * (a+b)/(a*b); }
}
static void FDsSymidx(const double *mu, const double *F, double *p, int len, const int *idx, int siz)  /* :170-198 */
{
    int mi; (void)len;
    for (mi = 0; mi < siz; mi++) { int i = idx[mi]; const double *u = mu + 6*i; F_COMMON; double a = rxc*rxc + ryc*ryc, b = rx*rx + ry*ry;
        p[i] = r*r * (a+b)/(a*b); }
}
static void exFDsSym(const double *u, const double *F, double *p, double *w, int len)   /* :228-250 */
{
    int i;
    for (i = 0; i < len; i++, u += 6) {
        F_COMMON; double a = rxc*rxc + ryc*ryc, b = rx*rx + ry*ry;
        w[i] = (a*b)/(a+b);
        p[i] = r*r / w[i];
    }
}

/* ------------------------------------------------------------------ utools.c:7-51 normu */
static void normu(const double *p, const int *inl, int len, double *A1, double *A2)
{
    int i, j; double a, b; const double *u;
    for (j = 0; j < 3; j++) { A1[j] = 0; A2[j] = 0; }
    for (j = 0; j < len; j++) {
        u = p + 6*inl[j];
        A1[1] += u[0]; A1[2] += u[1];
        A2[1] += u[3]; A2[2] += u[4];
    }
    if (len > 0) for (i = 1; i < 3; i++) { A1[i] /= len; A2[i] /= len; }
    for (j = 0; j < len; j++) {
        u = p + 6*inl[j];
        a = u[0] - A1[1]; b = u[1] - A1[2]; A1[0] += sqrt(a*a + b*b);
        a = u[3] - A2[1]; b = u[4] - A2[2]; A2[0] += sqrt(a*a + b*b);
    }
    if (A1[0] != 0) A1[0] = len * sqrt(2) / A1[0];
    if (A2[0] != 0) A2[0] = len * sqrt(2) / A2[0];
    A1[1] *= -A1[0]; A1[2] *= -A1[0];
    A2[1] *= -A2[0]; A2[2] *= -A2[0];
}

/* utools.c:170-184 cov_mat for siz = 9 : Cv = Z^T Z, Z is len x 9 row-major */
static void cov_mat9(double *Cv, const double *Z, int len)
{
    int i, j, k, lenM = len * 9; double val;
    for (i = 0; i < 9; i++)
        for (j = 0; j <= i; j++) {
            val = 0;
            for (k = 0; k < lenM; k += 9) val += Z[k+i] * Z[k+j];
            Cv[9*i + j] = val; Cv[i + 9*j] = val;
        }
}

/* ------------------------------------------------------------------ Ftools.c:300-458 u2f / u2fw */
static void u2f_core(const double *u, const int *inl, const double *w, int len, double *F)
{
    double A1[3], A2[3], V[81], D[9];
    int i, j, k, l;
    if (len > 8) {
        double *Z = (double *)malloc(sizeof(double) * 9 * (size_t)len), *p = Z;
        normu(u, inl, len, A1, A2);
        for (i = 0; i < len; i++) {                       /* lin_fmN, Ftools.c:300-328 */
            const double *s = u + 6*inl[i]; double a[3], b[3];
            a[2] = 1; b[2] = 1;
            a[0] = s[0] * A1[0] + A1[1]; a[1] = s[1] * A1[0] + A1[2];
            b[0] = s[3] * A2[0] + A2[1]; b[1] = s[4] * A2[0] + A2[2];
            for (k = 0; k < 3; k++) for (l = 0; l < 3; l++) *p++ = a[l] * b[k];
        }
        if (w) for (i = 0; i < len; i++) { double m = w[inl[i]]; for (k = 0; k < 9; k++) Z[9*i+k] *= m; }   /* scalmul, Ftools.c:414-418 */
        cov_mat9(V, Z, len);
        dg_eig_sym(V, D, 9);                              /* lap_eig + trnm, Ftools.c:368-369 */
        j = 0; for (i = 1; i < 9; i++) if (D[i] < D[j]) j = i;
        for (i = 0; i < 9; i++) F[i] = V[j*9 + i];
        free(Z);
    } else {
        /* lin_fm (Ftools.c:15-37) into a 9 x 8 row-major Z, then CCMATH svduv(D,Z,V,9,U,8); the
         * model is the last column of the 9x9 left factor (Ftools.c:372-384). */
        double Z[72], U9[81], V8[64], D8[8];
        for (i = 0; i < 72; i++) Z[i] = 0.;
        for (i = 0; i < len && i < 8; i++) {
            const double *s = u + 6*inl[i];
            for (k = 0; k < 3; k++) for (l = 0; l < 3; l++) Z[(k*3+l)*8 + i] = s[k+3] * s[l];
        }
        /* Ftools.c:427-432: scalmul(Z+i, w[j], 9, 9) strides by 9 over a matrix whose row stride is
         * len(=8): the weights land on a skewed diagonal pattern (and past the 72 used entries).
         * Reproduced as is. */
        if (w) for (i = 0; i < len && i < 8; i++) { double m = w[inl[i]]; for (k = 0; k < 9; k++) if (i + 9*k < 72) Z[i + 9*k] *= m; }
        dg_svduv(D8, Z, U9, 9, V8, 8);
        for (i = 0; i < 9; i++) F[i] = U9[i*9 + 8];
    }
    dg_singulF(F);
    if (len > 8) dg_denormF(F, A1, A2);
}
static void u2f(const double *u, const int *inl, int len, double *F) { u2f_core(u, inl, 0, len, F); }
static void u2fw(const double *u, const int *inl, const double *w, int len, double *F) { u2f_core(u, inl, w, len, F); }

/* Ftools.c:481-494 */
static int all_ori_valid(const double *F, const double *us, const int *idx, int N)
{
    double sig, sig1, ec[3]; int i;
    dg_epipole(ec, F);
    sig1 = dg_getorisig(F, ec, us + 6*idx[0]);
    for (i = 1; i < N; i++) {
        sig = dg_getorisig(F, ec, us + 6*idx[i]);
        if (sig1 * sig < 0) return 0;
    }
    return 1;
}

/* ------------------------------------------------------------------ Htools.c */
/* the two DLT rows of lin_hg (Htools.c:20-58) for one correspondence */
static void dlt_rows(const double *s, double *z0, double *z1)
{
    int j;
    for (j = 0; j < 3; j++) {
        z0[3*j+0] = s[3+j]; z0[3*j+1] = 0;      z0[3*j+2] = -s[0] * s[3+j];
        z1[3*j+0] = 0;      z1[3*j+1] = s[3+j]; z1[3*j+2] = -s[1] * s[3+j];
    }
}

static double HDs_one(const double *u, const double *H)                  /* Htools.c:161-200 body */
{
    double z0[9], z1[9], pJ[8], r1 = 0, r2 = 0, a, b, c, d, e, p; int j;
    dlt_rows(u, z0, z1);
    for (j = 0; j < 9; j++) { r1 += H[j] * z0[j]; r2 += H[j] * z1[j]; }
    a = H[0] - H[2] * u[0];
    b = H[3] - H[5] * u[0];
    c = -H[8] - H[2] * u[3] - H[5] * u[4];
    d = H[1] - H[2] * u[1];
    e = H[4] - H[5] * u[1];
    dg_pinvJ(a, b, c, d, e, pJ);
    p = 0;
    for (j = 0; j < 4; j++) { a = pJ[j] * r1 + pJ[j+4] * r2; p += a * a; }
    return p;
}
static void HDs(const double *u, const double *H, double *p, int len)
{
    int i; for (i = 0; i < len; i++) p[i] = HDs_one(u + 6*i, H);
}

/* Htools.c:101-133 */
static void u2h(const double *u, const int *inl, int len, double *H)
{
    double A1[3], A2[3], V[81], D[9]; int i, j, nb[18];
    if (len < 4) return;
    if (len == 4) {
        /* Htools.c:106-114: lin_hg gives a 9(col) x 8(row) block which the reference transposes as
         * if it were 9x9 before nullspace(); the 9 never-written entries are zeroed here */
        double Z2[81], z0[9], z1[9];
        for (i = 0; i < 81; i++) Z2[i] = 0.;
        for (i = 0; i < 4; i++) {
            dlt_rows(u + 6*inl[i], z0, z1);
            for (j = 0; j < 9; j++) { Z2[j*8 + 2*i] = z0[j]; Z2[j*8 + 2*i + 1] = z1[j]; }
        }
        dg_trnm(Z2, 9);
        for (i = 72; i < 81; i++) Z2[i] = 0.;
        for (i = 0; i < 81; i++) V[i] = 0.;
        dg_nullspace(Z2, V, 9, nb);
        memcpy(H, V, 9 * sizeof(double));
    } else {
        double *Z = (double *)malloc(sizeof(double) * 18 * (size_t)len), *p = Z;
        normu(u, inl, len, A1, A2);
        for (i = 0; i < len; i++) {                       /* lin_hgN, Htools.c:60-99 (rows of 9) */
            const double *s = u + 6*inl[i]; double a[3], b[3];
            a[2] = 1; b[2] = 1;
            a[0] = s[0] * A1[0] + A1[1]; a[1] = s[1] * A1[0] + A1[2];
            b[0] = s[3] * A2[0] + A2[1]; b[1] = s[4] * A2[0] + A2[2];
            for (j = 0; j < 3; j++) { p[3*j] = b[j]; p[3*j+1] = 0; p[3*j+2] = -a[0] * b[j]; }
            p += 9;
            for (j = 0; j < 3; j++) { p[3*j] = 0; p[3*j+1] = b[j]; p[3*j+2] = -a[1] * b[j]; }
            p += 9;
        }
        cov_mat9(V, Z, 2*len);
        dg_eig_sym(V, D, 9);
        memcpy(H, V, 9 * sizeof(double));                 /* first (smallest) eigenvector, Htools.c:127 */
        dg_denormH(H, A1, A2);
        free(Z);
    }
}

/* ------------------------------------------------------------------ ranH.c:18-135 (LO of the plane homography inside DEGENSAC) */
static dg_score iterH(dg_ctx *c, const double *u, int len, int *inliers, double th, double ths,
                      double *H, double **errs, unsigned inlLimit)
{
    double *d = errs[1], h[9], dth; int it, *inlSubset;
    dg_score S = {0, 0, 0, 0}, Ss, maxS;
    dth = (ths - th) / ILSQ_ITERS;
    maxS = inlidxs(errs[4], len, th, inliers);
    if (maxS.I < 4) return S;
    if (maxS.I <= inlLimit) u2h(u, inliers, maxS.I, h);
    else { inlSubset = randsubset(c, inliers, maxS.I, inlLimit); u2h(u, inlSubset, inlLimit, h); }
    for (it = 0; it < ILSQ_ITERS; ++it) {
        HDs(u, h, d, len); c->n_hds++;
        S = inlidxs(d, len, th, inliers);
        Ss = inlidxs(d, len, ths, inliers);
        if (scoreLess(maxS, S)) {
            maxS = S; errs[1] = errs[0]; errs[0] = d; d = errs[1];
            memcpy(H, h, 9 * sizeof(double));
        }
        if (Ss.I < 4) return maxS;
        if (Ss.I <= inlLimit) u2h(u, inliers, Ss.I, h);
        else { inlSubset = randsubset(c, inliers, Ss.I, inlLimit); u2h(u, inlSubset, inlLimit, h); }
        ths -= dth;
    }
    HDs(u, h, d, len); c->n_hds++;
    S = inlidxs(d, len, th, inliers);
    if (scoreLess(maxS, S)) {
        maxS = S; errs[1] = errs[0]; errs[0] = d;
        memcpy(H, h, 9 * sizeof(double));
    }
    return maxS;
}

static dg_score inHrani(dg_ctx *c, const double *u, int len, int *inliers, int ninl, double th,
                        double **errs, double *H, unsigned inlLimit)
{
    int ssiz, i; dg_score S, maxS = {0, 0, 0, 0};
    double *d, h[9]; int *sample, *intbuff;
    if (ninl < 8) return maxS;
    intbuff = (int *)malloc(len * sizeof(int));
    ssiz = ninl / 2; if (ssiz > 12) ssiz = 12;
    d = errs[2]; errs[2] = errs[0]; errs[0] = d;
    for (i = 0; i < RAN_REP; ++i) {
        sample = randsubset(c, inliers, ninl, ssiz);
        u2h(u, sample, ssiz, h);
        HDs(u, h, errs[0], len); c->n_hds++;
        errs[4] = errs[0];
        S = iterH(c, u, len, intbuff, th, TC*th, h, errs, inlLimit);
        if (scoreLess(maxS, S)) {
            maxS = S; d = errs[2]; errs[2] = errs[0]; errs[0] = d;
            memcpy(H, h, 9 * sizeof(double));
        }
    }
    d = errs[2]; errs[2] = errs[0]; errs[0] = d;
    free(intbuff);
    return maxS;
}

/* ------------------------------------------------------------------ DegUtils.c */
static void skew_sym(const double *a, double *ax)                        /* DegUtils.c:209-220 */
{
    ax[0] = 0; ax[1] = -a[2]; ax[2] = a[1];
    ax[3] = a[2]; ax[4] = 0; ax[5] = -a[0];
    ax[6] = -a[1]; ax[7] = a[0]; ax[8] = 0;
}
static void crossp(double *h, const double *u, const double *v)          /* DegUtils.c:244-249 */
{
    h[0] = u[1]*v[2] - u[2]*v[1];
    h[1] = u[2]*v[0] - u[0]*v[2];
    h[2] = u[0]*v[1] - u[1]*v[0];
}

/* DegUtils.c:93-161 */
static void Hdetect(const double *F, const double *u7, const unsigned char *IDXS, double *H)
{
    double D[3], U[9], V[9], ec[3], Ex[9], A[9], u3a[9], u3b[9], u3aT[9], u3bT[9], Au3b[9],
           Ft[9], F1[9], p1[9], p1T[9], p2[9], b[3];
    int i, j, sing;
    dg_mattr(Ft, F, 3, 3);
    memcpy(F1, F, sizeof F1);
    dg_svduv(D, F1, U, 3, V, 3);
    ec[0] = V[2]; ec[1] = V[5]; ec[2] = V[8];
    skew_sym(ec, Ex);
    dg_mmul(A, Ex, Ft, 3);
    for (i = 0; i < 3; ++i)                                /* fillu3, DegUtils.c:225-233 */
        for (j = 0; j < 3; ++j) { u3a[i+j*3] = u7[IDXS[i]*6+j]; u3b[i+j*3] = u7[IDXS[i]*6+j+3]; }
    dg_mmul(Au3b, A, u3b, 3);
    dg_mattr(u3aT, u3a, 3, 3);
    dg_mattr(u3bT, Au3b, 3, 3);
    crossp(p1T, u3aT, u3bT);
    crossp(p1T+3, u3aT+3, u3bT+3);
    crossp(p1T+6, u3aT+6, u3bT+6);
    dg_mattr(p1, p1T, 3, 3);
    for (i = 0; i < 9; ++i) Ex[i] *= -1;
    dg_mmul(p2, Ex, u3a, 3);
    b[0] = (p1[0]*p2[0] + p1[3]*p2[3] + p1[6]*p2[6]) / (p2[0]*p2[0] + p2[3]*p2[3] + p2[6]*p2[6]);
    b[1] = (p1[1]*p2[1] + p1[4]*p2[4] + p1[7]*p2[7]) / (p2[1]*p2[1] + p2[4]*p2[4] + p2[7]*p2[7]);
    b[2] = (p1[2]*p2[2] + p1[5]*p2[5] + p1[8]*p2[8]) / (p2[2]*p2[2] + p2[5]*p2[5] + p2[8]*p2[8]);
    dg_mattr(u3bT, u3b, 3, 3);
    sing = dg_minv(u3bT, 3);
    dg_rmmult(u3b, u3bT, b, 3, 3, 1);
    dg_mattr(u3bT, u3b, 3, 1);
    dg_rmmult(u3b, ec, u3bT, 3, 1, 3);
    for (i = 0; i < 3; ++i) for (j = 0; j < 3; ++j) H[i+j*3] = A[i*3+j] - u3b[i*3+j];
    if (isnan(*H) || isinf(*H) || sing) {
        H[1] = H[2] = H[3] = H[5] = H[6] = H[7] = 0;
        H[0] = H[4] = H[8] = 1;
    }
}

/* DegUtils.c:164-183 */
static void sortDs(const double *Ds, double *sDs, unsigned char *idx)
{
    int i, j; unsigned char auxI; double auxD;
    memcpy(sDs, Ds, 7 * sizeof(double));
    for (i = 0; i < 7; ++i) idx[i] = (unsigned char)i;
    for (i = 0; i < 7; ++i)
        for (j = i + 1; j < 7; ++j)
            if (sDs[j] < sDs[i]) {
                auxD = sDs[j]; sDs[j] = sDs[i]; sDs[i] = auxD;
                auxI = idx[j]; idx[j] = idx[i]; idx[i] = auxI;
            }
}

/* DegUtils.c:42-82 */
static int checksample(const double *F, const double *u7, double th, double *H)
{
    static const unsigned char IDXS[5][3] = {{0,1,2}, {3,4,5}, {0,1,6}, {3,4,6}, {2,5,6}};
    int i, j, inl[7], inlCount; double Ds[7], sDs[7]; unsigned char idx[7];
    for (i = 0; i < 5; ++i) {
        Hdetect(F, u7, IDXS[i], H);
        HDs(u7, H, Ds, 7);
        sortDs(Ds, sDs, idx);
        for (j = 0; j < 5; ++j) inl[j] = idx[j];
        u2h(u7, inl, 5, H);
        HDs(u7, H, Ds, 7);
        inlCount = 0;
        for (j = 0; j < 7; ++j) if (Ds[j] < th) ++inlCount;
        if (inlCount > 4) return 1;
    }
    return 0;
}

/* DegUtils.c:693-731 */
static unsigned innerH(dg_ctx *c, double *H, const double *u, int len, double th, unsigned iters, unsigned char *inl)
{
    double *err, *d, *errs[5]; int i, j, I, *inliers; dg_score S;
    err = (double *)malloc((size_t)len * 4 * sizeof(double));
    for (i = 0; i < 4; i++) errs[i] = err + (size_t)i * len;
    errs[4] = errs[3];
    inliers = (int *)malloc(sizeof(int) * len);
    d = errs[0];
    HDs(u, H, d, len); c->n_hds++;
    S = inlidxs(d, len, th, inliers);
    S = inHrani(c, u, len, inliers, S.I, th, errs, H, iters);
    d = errs[0];
    I = 0;
    for (j = 0; j < len; j++) { if (d[j] <= th) { ++I; inl[j] = 1; } else inl[j] = 0; }
    free(err); free(inliers);
    return (unsigned)I;
}

/* DegUtils.c:635-690 */
static unsigned u2Fit(dg_ctx *c, const double *u, unsigned len, double *F, unsigned char *inl, double th, double ths, unsigned iters)
{
    double dth = (ths - th) / (iters - 1); unsigned iter, i, no_i;
    int *inlI = (int *)malloc(len * sizeof(int));
    double *Ds = (double *)malloc(len * sizeof(double));
    for (iter = 0; iter < iters; ++iter) {
        FDs(u, F, Ds, len); c->n_fds_direct++;
        no_i = 0;
        for (i = 0; i < len; ++i) { if (Ds[i] < ths) { inl[i] = 1; ++no_i; } else inl[i] = 0; }
        if (no_i < 8) { free(inlI); free(Ds); return no_i; }
        no_i = 0;
        for (i = 0; i < len; ++i) if (inl[i]) inlI[no_i++] = i;
        u2f(u, inlI, no_i, F);
        ths -= dth;
    }
    FDs(u, F, Ds, len); c->n_fds_direct++;
    no_i = 0;
    for (i = 0; i < len; ++i) { if (Ds[i] < th) { inl[i] = 1; ++no_i; } else inl[i] = 0; }
    free(inlI); free(Ds);
    return no_i;
}

/* DegUtils.c:596-632 */
static void dual_sample(dg_ctx *c, const double *uA, unsigned lenA, unsigned sA, const double *uB, unsigned lenB, unsigned sB, double *usam)
{
    unsigned idx, pos, i;
    unsigned *ptrA = (unsigned *)malloc(lenA * sizeof(unsigned));
    unsigned *ptrB = (unsigned *)malloc(lenB * sizeof(unsigned));
    for (i = 0; i < lenA; ++i) ptrA[i] = i;
    for (i = 0; i < lenB; ++i) ptrB[i] = i;
    for (pos = 0; pos < sA; ++pos) { idx = dg_rand(&c->rng) % lenA; i = ptrA[pos]; ptrA[pos] = ptrA[idx]; ptrA[idx] = i; }
    for (pos = 0; pos < sB; ++pos) { idx = dg_rand(&c->rng) % lenB; i = ptrB[pos]; ptrB[pos] = ptrB[idx]; ptrB[idx] = i; }
    for (i = 0; i < sA; ++i) memcpy(usam + 6*i, uA + 6*ptrA[i], 6 * sizeof(double));
    for (i = 0; i < sB; ++i) memcpy(usam + 6*(i+sA), uB + 6*ptrB[i], 6 * sizeof(double));
    free(ptrA); free(ptrB);
}

/* DegUtils.c:488-593 */
static void innerFH(dg_ctx *c, const double *uH, unsigned lenH, const double *uO, unsigned lenO,
                    const double *u, unsigned len, double th, unsigned repCount, unsigned sam_sizH, unsigned sam_sizO,
                    double *F, unsigned char *inl)
{
    unsigned i, rep, max_i, max_s, no_i; double aF[9];
    unsigned char *v = (unsigned char *)malloc(len);
    double *usam = (double *)malloc(6 * (sam_sizH + sam_sizO) * sizeof(double));
    double *Ds = (double *)malloc(len * sizeof(double));
    int *allInl = (int *)malloc((sam_sizH + sam_sizO) * sizeof(int));
    for (i = 0; i < sam_sizH + sam_sizO; ++i) allInl[i] = i;
    for (i = 0; i < 9; ++i) F[i] = 1;
    for (i = 0; i < len; ++i) inl[i] = 0;
    max_i = 0; max_s = 0;
    for (rep = 0; rep < repCount; ++rep) {
        dual_sample(c, uH, lenH, sam_sizH, uO, lenO, sam_sizO, usam);
        u2f(usam, allInl, sam_sizH + sam_sizO, aF);
        FDs(u, aF, Ds, len); c->n_fds_direct++;
        no_i = 0;
        for (i = 0; i < len; ++i) { if (Ds[i] < th) { v[i] = 1; ++no_i; } else v[i] = 0; }
        if (max_i < no_i) { memcpy(inl, v, len); memcpy(F, aF, sizeof aF); max_i = no_i; }
        if (no_i > max_s) {
            max_s = no_i;
            no_i = u2Fit(c, u, len, aF, v, th, th*3, 4);
            if (max_i < no_i) { memcpy(inl, v, len); memcpy(F, aF, sizeof aF); max_i = no_i; }
        }
    }
    free(usam); free(Ds); free(allInl); free(v);
}

/* DegUtils.c:254-444 */
static unsigned rFtH(dg_ctx *c, const double *u, const unsigned char *hinl, double th, const double *H, unsigned len, double *F)
{
    unsigned char *nhinl, *v, *inl; double *Ds, *uN, *us, *uH, *uV;
    unsigned i, nhinlCount = 0, hinlCount = 0, ninl, maxni;
    unsigned *ptr, max_i, m_i, max_sam, s_size, pos, idx, auxI, no_i;
    double ec[3], ecNorm, c1[3], c2[3], aFt[9], aFtH[9], aF[9], Ht[9];
    unsigned MAX_SAM = 10000, no_sam; double conf = .999;
    unsigned char sam_sizH = 6, sam_sizO = 4;

    inl = (unsigned char *)malloc(len);
    Ds = (double *)malloc(len * sizeof(double));
    HDs(u, H, Ds, len); c->n_hds++;
    nhinl = (unsigned char *)malloc(len);
    for (i = 0; i < len; ++i) {
        if (Ds[i] > 100*th) { nhinl[i] = 1; ++nhinlCount; } else nhinl[i] = 0;
        if (hinl[i]) ++hinlCount;
    }
    free(Ds);
    Ds = (double *)malloc((nhinlCount + 1) * sizeof(double));
    v = (unsigned char *)malloc(nhinlCount + 1);
    uN = (double *)malloc(6 * (nhinlCount + 1) * sizeof(double));
    us = (double *)malloc(6 * (nhinlCount + 1) * sizeof(double));
    uV = (double *)malloc(6 * (nhinlCount + 1) * sizeof(double));
    uH = (double *)malloc(6 * (hinlCount + 1) * sizeof(double));
    nhinlCount = 0; hinlCount = 0;
    for (i = 0; i < len; ++i) {
        if (nhinl[i]) {
            memcpy(uN + 6*nhinlCount, u + 6*i, 6 * sizeof(double));
            memcpy(us + 6*nhinlCount, u + 6*i, 3 * sizeof(double));
            us[6*nhinlCount+3] = H[0]*u[6*i+3] + H[3]*u[6*i+4] + H[6]*u[6*i+5];
            us[6*nhinlCount+4] = H[1]*u[6*i+3] + H[4]*u[6*i+4] + H[7]*u[6*i+5];
            us[6*nhinlCount+5] = H[2]*u[6*i+3] + H[5]*u[6*i+4] + H[8]*u[6*i+5];
            ++nhinlCount;
        }
        if (hinl[i]) { memcpy(uH + 6*hinlCount, u + 6*i, 6 * sizeof(double)); ++hinlCount; }
    }
    ptr = (unsigned *)malloc((nhinlCount + 1) * sizeof(unsigned));
    for (i = 0; i < nhinlCount; ++i) ptr[i] = i;
    max_i = 3; m_i = sam_sizO; max_sam = MAX_SAM; s_size = 2;

    if (nhinlCount < 4 || hinlCount < 6) {
        max_i = 0;
    } else {
        for (no_sam = 1; no_sam < 2*max_sam; ++no_sam) {
            for (pos = 0; pos < s_size; ++pos) {
                idx = pos + 1 + dg_rand(&c->rng) % (nhinlCount - pos - 1);
                auxI = ptr[pos]; ptr[pos] = ptr[idx]; ptr[idx] = auxI;
            }
            crossp(c1, us + 6*ptr[0], us + 6*ptr[0] + 3);
            crossp(c2, us + 6*ptr[1], us + 6*ptr[1] + 3);
            crossp(ec, c1, c2);
            ecNorm = sqrt(ec[0]*ec[0] + ec[1]*ec[1] + ec[2]*ec[2]);
            ec[0] = ec[0]/ecNorm; ec[1] = ec[1]/ecNorm; ec[2] = ec[2]/ecNorm;
            skew_sym(ec, aFt);
            dg_mattr(Ht, H, 3, 3);
            dg_mmul(aFtH, aFt, Ht, 3);
            dg_mattr(aFt, aFtH, 3, 3);
            FDs(uN, aFt, Ds, nhinlCount); c->n_fds_direct++;
            no_i = 0;
            for (i = 0; i < nhinlCount; ++i) { if (Ds[i] < th*2) { ++no_i; v[i] = 1; } else v[i] = 0; }
            if (no_i > m_i) {
                no_i = 0;
                for (i = 0; i < nhinlCount; ++i) if (v[i]) { memcpy(uV + 6*no_i, uN + 6*i, 6 * sizeof(double)); ++no_i; }
                m_i = no_i;
                innerFH(c, uH, hinlCount, uV, no_i, u, len, th, 15, sam_sizH, sam_sizO, aF, inl);
                ninl = 0;
                for (i = 0; i < len; ++i) if (inl[i]) ++ninl;
                if (ninl > max_i) {
                    unsigned ns;
                    max_i = ninl;
                    memcpy(F, aF, sizeof aF);
                    maxni = 0;
                    for (i = 0; i < len; ++i) if (inl[i] && nhinl[i]) ++maxni;
                    ns = (unsigned)dg_nsamples((int)maxni, (int)nhinlCount, 2, conf);
                    max_sam = max_sam > ns ? ns : max_sam;
                }
            }
        }
    }
    free(inl); free(Ds); free(nhinl); free(uN); free(us); free(uV); free(uH); free(ptr); free(v);
    return max_i;
}

/* ------------------------------------------------------------------ exp_ranF.c:621-743 */
static dg_score exp_iterFcustom(dg_ctx *c, const double *u, int len, int *inliers, double th, double ths, int iters,
                                double *F, double **errs, int iterID, unsigned inlLimit, exfds_fn EXFDS1, fds_fn FDS1)
{
    double *d = errs[1], *w, f[9], dth; int it;
    dg_score S = {0, 0, 0, 0}, Ss, maxS; int *detachedInl; unsigned detachedCount;
    int iterIDret; uint32_t hash;
    w = (double *)malloc(len * sizeof(double));
    dth = (ths - th) / ILSQ_ITERS;
    maxS = inlidxs(errs[4], len, th, inliers);
    TRACE2(10, maxS.I, maxS.J);
    if (maxS.I < 8) { free(w); return S; }
    S = inlidxs(errs[4], len, th*MWM, inliers);
    TRACE2(15, S.I, 0);
    detachedCount = (unsigned)(int)(S.I * 1);              /* D3_F_RATIO=1, D3_F_MIN=0: exp_ranF.h:15-16 */
    if (detachedCount > inlLimit) detachedCount = inlLimit;
    if (detachedCount < 8) detachedCount = 8;
    if (detachedCount >= S.I) u2f(u, inliers, S.I, f);
    else { detachedInl = randsubset(c, inliers, S.I, detachedCount); u2f(u, detachedInl, detachedCount, f); }
    for (it = 0; it < iters; it++) {
        EXFDS1(u, f, d, w, len); c->n_exfds++; TRACE(1, f);
        S = inlidxs(d, len, th, inliers);
        TRACE2(11, S.I, S.J);
        hash = dg_superfasthash((const unsigned char *)inliers, (int)(S.I * sizeof(*inliers)));
        iterIDret = htContains(c, hash, S.I, iterID);
        if (iterIDret != -1 && iterIDret != iterID) { TRACE2(13, iterIDret, 0); S.I = 0; S.J = 0; free(w); return S; }
        if (iterIDret == -1) htInsert(c, hash, S.I, iterID);
        if (scoreLess(maxS, S)) {
            maxS = S; errs[1] = errs[0]; errs[0] = d; d = errs[1];
            memcpy(F, f, 9 * sizeof(double));
        }
        Ss = inlidxs(d, len, ths*MWM, inliers);
        TRACE2(14, Ss.I, 0);
        if (Ss.I < 8) { free(w); return maxS; }
        detachedCount = (unsigned)(int)(Ss.I * 1);
        if (detachedCount > inlLimit) detachedCount = inlLimit;
        if (detachedCount < 8) detachedCount = 8;
        if (detachedCount >= Ss.I) u2fw(u, inliers, w, Ss.I, f);
        else { detachedInl = randsubset(c, inliers, Ss.I, detachedCount); u2fw(u, detachedInl, w, detachedCount, f); }
        ths -= dth;
    }
    FDS1(u, f, d, len); c->n_fds++; TRACE(0, f);
    S = inlidxs(d, len, th, inliers);
    TRACE2(12, S.I, S.J);
    if (scoreLess(maxS, S)) {
        maxS = S; errs[1] = errs[0]; errs[0] = d;
        memcpy(F, f, 9 * sizeof(double));
    }
    free(w);
    return maxS;
}

/* exp_ranF.c:745-806 */
static dg_score exp_inFranicustom(dg_ctx *c, const double *u, int len, int *inliers, int ninl, double th, double **errs,
                                  double *F, int *iterID, unsigned inlLimit, exfds_fn EXFDS1, fds_fn FDS1)
{
    unsigned ssiz, i; dg_score S = {0, 0, 0, 0}, maxS = {0, 0, 0, 0}; int jj;
    double *d, f[9]; int *sample, *intbuff, *intbuff_best;
    if (ninl < 16) return maxS;
    intbuff = (int *)malloc(sizeof(int) * len);
    intbuff_best = (int *)malloc(sizeof(int) * len);
    ssiz = ninl / 2; if (ssiz > 14) ssiz = 14;
    d = errs[2]; errs[2] = errs[0]; errs[0] = d;
    for (i = 0; i < RAN_REP; i++) {
        sample = randsubset(c, inliers, ninl, ssiz);
        u2f(u, sample, ssiz, f);
        FDS1(u, f, errs[0], len); c->n_fds++; TRACE(0, f);
        errs[4] = errs[0];
        S = exp_iterFcustom(c, u, len, intbuff, th, TC*th, ILSQ_ITERS, f, errs, ++*iterID, inlLimit, EXFDS1, FDS1);
        if (scoreLess(maxS, S)) {
            maxS = S; d = errs[2]; errs[2] = errs[0]; errs[0] = d;
            memcpy(F, f, 9 * sizeof(double));
            for (jj = 0; jj < (int)maxS.I; jj++) intbuff_best[jj] = intbuff[jj];
        }
    }
    d = errs[2]; errs[2] = errs[0]; errs[0] = d;
    for (jj = 0; jj < (int)maxS.I; jj++) inliers[jj] = intbuff_best[jj];
    free(intbuff); free(intbuff_best);
    return maxS;
}

/* sym + LAF consistency of a candidate, shared shape of exp_ranF.c:1383-1411 / :1526-1556 / :1654-1682.
 * Returns 0 if the candidate must be rejected. */
static int g_legacy_sym = 0;      /* exp_ransacFcustom's symmetric check: all points instead of the inliers (exp_ranF.c:943-953) */
static __thread int g_laf_rej = 0;         /* candidates the LAF check turned down (`S.Ilafs < maxS.Ilafs`, exp_ranF.c:1410 / :1553 / :1681):
   stats[DG_ST_REJECTED] of the F driver */
static int f_checks(const double *u, const double *u_1, const double *u_2, int len, const double *f, const int *inliers,
                    dg_score *S, const dg_score *maxS, int doSymCheck, double SymCheck_th, int DO_LAF_CHECK,
                    double th_laf_check, fdsidx_fn FDS1idx, double *d_check, double *err_laf)
{
    int j, p1_inliers;
    if (doSymCheck && g_legacy_sym) {
        FDsSym(u, f, d_check, len);
        S->Is = 0;
        for (j = 0; j < len; j++) if (d_check[j] <= SymCheck_th) S->Is++;
        if (S->Is < maxS->Is) return 0;
    } else if (doSymCheck) {
        FDsSymidx(u, f, d_check, len, inliers, S->I);
        S->Is = 0;
        for (j = 0; j < (int)S->I; j++) if (d_check[inliers[j]] <= SymCheck_th) S->Is++;
        if (S->Is < maxS->Is) return 0;
    }
    if (DO_LAF_CHECK) {
        FDS1idx(u_1, f, err_laf, len, inliers, S->I);
        p1_inliers = 0; S->Ilafs = 0;
        for (j = 0; j < (int)S->I; j++) if (err_laf[inliers[j]] <= th_laf_check) p1_inliers++;
        FDS1idx(u_2, f, err_laf, len, inliers, S->I);
        for (j = 0; j < (int)S->I; j++) if (err_laf[inliers[j]] <= th_laf_check) S->Ilafs++;
        S->Ilafs = S->Ilafs < (unsigned)p1_inliers ? S->Ilafs : (unsigned)p1_inliers;
        if (S->Ilafs < maxS->Ilafs) { g_laf_rej++; return 0; }
    }
    return 1;
}

/* exp_ranF.c:1244-1767 */
static int exp_ransacFcustomLAF(dg_ctx *c, const double *u, const double *u_1, const double *u_2, int len, double th,
                                double laf_coef, double conf, int max_sam, double *F, unsigned char *inl,
                                int do_lo, unsigned inlLimit, exfds_fn EXFDS1, fds_fn FDS1, fdsidx_fn FDS1idx,
                                double SymCheck_th_in, int enable_degen_check, unsigned seed0, int final_flags, int *stats)
{
    const int final_laf_filter = final_flags & 1, legacy = (final_flags >> 1) & 1;   /* bit 1: the legacy drivers exp_ransacF / exp_ransacFcustom */
    unsigned seed; int *pool, no_sam, new_sam; double u7[42], H[9], FBest[9];
    double *f1, *f2, poly[4], roots[3], f[9], *err, *d, *d_check, *errs[5];
    int nsol, i = 0, j, *inliers, new_max = 0, do_iterate; unsigned I;
    dg_score maxS = {0,0,0,0}, maxSs = {0,0,0,0}, S = {0,0,0,0};
    int *samidx, samidxBest[7]; double *errorsBest;
    int degen_cnt = 0, iter_cnt = 0, iterID = 0; unsigned non_degen_samples_count = 0;
    double jj, *HDsv = (double *)malloc(len * sizeof(double));
    /* the legacy drivers' symmetric check has its own constant, CHECK_COEF * th (exp_ranF.c:19, :837) */
    const double SymCheck_th = (legacy && SymCheck_th_in > 0) ? 16.0 * th : SymCheck_th_in;
    int Ihmax = 0; const int doSymCheck = SymCheck_th > 0; const int DO_LAF_CHECK = laf_coef > 0;
    const double th_laf_check = laf_coef * th; double *err_laf;
    double A[81], sol[81]; int nullspace_buff[18], nullsize, best_sample = 0;

    g_legacy_sym = legacy; g_laf_rej = 0;
    dg_srand(&c->rng, seed0);                               /* srand(time(NULL)), :1277 */
    c->ht_n = 0;                                            /* htInit, :1290 */
    pool = (int *)malloc(len * sizeof(int));
    for (i = 0; i < len; i++) pool[i] = i;
    samidx = pool + len - 7;
    errorsBest = (double *)malloc(len * sizeof(double));
    err = (double *)malloc((size_t)len * 4 * sizeof(double));
    err_laf = (double *)malloc(len * sizeof(double));
    d_check = (double *)malloc(len * sizeof(double));
    for (i = 0; i < 4; i++) errs[i] = err + (size_t)i * len;
    errs[4] = errs[3];
    memset(err, 0, (size_t)len * 4 * sizeof(double));
    memset(errorsBest, 0, len * sizeof(double));
    inliers = (int *)malloc(sizeof(int) * len);
    maxS.I = 8; maxSs.I = 8;
    no_sam = 0;
    f1 = sol; f2 = sol + 9;
    seed = (unsigned)dg_rand(&c->rng);

    while (no_sam < max_sam) {
        no_sam++;
        dg_srand(&c->rng, seed);
        /* rsampleT(Z,9,pool,7,len,A) (rtools.c:74-92) with Z = lin_fm(u) (Ftools.c:15-37): row i of A
         * is the i-th DRAWN point, entries u2_k*u1_l */
        for (i = 0; i < 7; i++) {
            int s = dg_rand(&c->rng) % (len - i), jx = len - i - 1, q = pool[s], k, l;
            const double *pt;
            pool[s] = pool[jx]; pool[jx] = q;
            pt = u + 6*q;
            for (k = 0; k < 3; k++) for (l = 0; l < 3; l++) A[i*9 + k*3 + l] = pt[k+3] * pt[l];
        }
        for (i = 0; i < 7; i++) memcpy(u7 + 6*i, u + 6*samidx[i], 6 * sizeof(double));   /* loadSample, :1340 */
        seed = (unsigned)dg_rand(&c->rng);

        for (i = 7*9; i < 9*9; ++i) A[i] = 0.0;
        nullsize = dg_nullspace(A, f1, 9, nullspace_buff);
        if (nullsize != 2) continue;
        dg_slcm(f1, f2, poly);
        nsol = dg_rroots3(poly, roots);

        new_max = 0; do_iterate = 0;
        for (i = 0; i < nsol; i++) {
            for (j = 0; j < 9; j++) f[j] = f1[j] * roots[i] + f2[j] * (1 - roots[i]);
            if (!all_ori_valid(f, u, samidx, 7)) continue;
            d = errs[i];
            FDS1(u, f, d, len); c->n_fds++; TRACE(0, f);
            S = inlidxs(d, len, th, inliers);

            if (scoreLess(maxS, S)) {
                if (!f_checks(u, u_1, u_2, len, f, inliers, &S, &maxS, doSymCheck, SymCheck_th, DO_LAF_CHECK,
                              th_laf_check, FDS1idx, d_check, err_laf)) continue;
                errs[i] = errs[3]; errs[3] = d;
                maxS = S;
                memcpy(F, f, 9 * sizeof(double));
                new_max = 1; best_sample = no_sam;
            }

            if (scoreLess(maxSs, S)) {
                maxSs = S;
                TRACE2(30, S.I, S.J);
                if (enable_degen_check && checksample(f, u7, 3*th, H)) {
                    HDs(u, H, HDsv, len); c->n_hds++;            /* dHDs, :1430 */
                    I = 0;
                    for (j = 0; j < len; ++j) if (HDsv[j] < th*3) ++I;
                    TRACE2(31, I, no_sam);
                    if (I < 8) break;
                    I = innerH(c, H, u, len, 16*th, 10, inl);
                    TRACE2(32, I, 0);
                    if ((int)I > Ihmax) Ihmax = (int)I;
                    if (I > 6) {
                        I = rFtH(c, u, inl, th, H, len, f);
                        TRACE2(33, I, maxS.I);
                        if (I > maxS.I) {
                            FDS1(u, f, errs[3], len); c->n_fds++; TRACE(0, f);
                            maxS.I = I;
                            memcpy(F, f, 9 * sizeof(double));
                            new_max = 1; best_sample = no_sam;
                            d = errs[3];
                        } else {
                            FDS1(u, f, errs[i], len); c->n_fds++; TRACE(0, f);
                            d = errs[i];
                        }
                        I = 0; jj = 0;
                        for (j = 0; j < len; j++) { if (d[j] <= th) I++; jj += dg_truncQuad(d[j], th); }
                        if (new_max) maxS.J = jj;
                        ++degen_cnt;
                    }
                } else {
                    do_iterate = (do_lo > 0 && (no_sam > ITER_SAM));
                    errs[4] = d;
                    non_degen_samples_count++;
                    memcpy(samidxBest, samidx, 7 * sizeof(int));
                    memcpy(errorsBest, d, len * sizeof(double));
                    memcpy(FBest, f, 9 * sizeof(double));
                }
            }
        }

        if (do_lo > 0 && (no_sam == ITER_SAM) && non_degen_samples_count) do_iterate = 1;

        if (do_iterate) {
            iter_cnt++;
            d = errs[0];
            S = inlidxs(errs[4], len, TC*th*MWM, inliers);
            TRACE2(1, S.I, no_sam);
            u2f(u, inliers, S.I, f);
            FDS1(u, f, d, len); c->n_fds++; TRACE(0, f);
            S = inlidxs(d, len, th, inliers);
            TRACE2(2, S.I, S.J);
            S = exp_inFranicustom(c, u, len, inliers, S.I, th, errs, f, &iterID, inlLimit, EXFDS1, FDS1);
            if (scoreLess(maxS, S)) {
                if (f_checks(u, u_1, u_2, len, f, inliers, &S, &maxS, doSymCheck, SymCheck_th, DO_LAF_CHECK,
                             th_laf_check, FDS1idx, d_check, err_laf)) {
                    d = errs[0]; errs[0] = errs[3]; errs[3] = d;
                    maxS = S;
                    memcpy(F, f, 9 * sizeof(double));
                    new_max = 1; best_sample = no_sam;
                }
            }
            if (new_max && !legacy) {
                new_sam = dg_nsamples(maxS.I + 1, len, 7, conf);
                if (new_sam < max_sam) max_sam = new_sam;
            }
        }
        /* the legacy drivers exp_ransacF / exp_ransacFcustom (exp_ranF.c:242, :811) update the sample budget after EVERY
         * sample that produced a new best model, whoever found it (:1085-1090 sits outside the do_iterate block there) */
        if (new_max && legacy) {
            new_sam = dg_nsamples(maxS.I + 1, len, 7, conf);
            if (new_sam < max_sam) max_sam = new_sam;
        }
    }

    /* "If there were no LOs, do at least one NOW!"  :1580-1697 */
    if (do_lo && (!iter_cnt && !degen_cnt) && non_degen_samples_count) {
        for (j = 0; j < 7; j++) memcpy(u7 + 6*j, u + 6*samidxBest[j], 6 * sizeof(double));
        if (enable_degen_check && checksample(FBest, u7, 3*th, H)) {
            HDs(u, H, HDsv, len); c->n_hds++;
            I = 0;
            for (j = 0; j < len; ++j) if (HDsv[j] < th*3) ++I;
            if (I >= 8) I = innerH(c, H, u, len, 16*th, 10, inl);
            if ((int)I > Ihmax) Ihmax = (int)I;
            if (I > 6) {
                /* NOTE: `inl` is whatever innerH left (or untouched memory when I in 7 and innerH skipped) */
                I = rFtH(c, u, inl, th, H, len, f);
                if (I > maxS.I) {
                    FDS1(u, f, errs[3], len); c->n_fds++; TRACE(0, f);
                    maxS.I = I;
                    memcpy(F, f, 9 * sizeof(double));
                    new_max = 1;
                    d = errs[3];
                } else {
                    int ii = i > 3 ? 3 : i;                 /* stale loop variable i == nsol, :1614 */
                    FDS1(u, f, errs[ii], len); c->n_fds++; TRACE(0, f);
                    d = errs[ii];
                }
                I = 0; jj = 0;
                for (j = 0; j < len; j++) { if (d[j] <= th) I++; jj += dg_truncQuad(d[j], th); }
                if (new_max) maxS.J = jj;
                ++degen_cnt;
            }
        } else {
            iter_cnt++;
            d = errs[0];
            S = inlidxs(errorsBest, len, TC*th*MWM, inliers);
            u2f(u, inliers, S.I, f);
            FDS1(u, f, d, len); c->n_fds++; TRACE(0, f);
            S = inlidxs(d, len, th, inliers);
            S = exp_inFranicustom(c, u, len, inliers, S.I, th, errs, f, &iterID, inlLimit, EXFDS1, FDS1);
            if (scoreLess(maxS, S)) {
                if (f_checks(u, u_1, u_2, len, f, inliers, &S, &maxS, doSymCheck, SymCheck_th, DO_LAF_CHECK,
                             th_laf_check, FDS1idx, d_check, err_laf)) {
                    d = errs[0]; errs[0] = errs[3]; errs[3] = d;
                    maxS = S;
                    memcpy(F, f, 9 * sizeof(double));
                    new_max = 1; best_sample = no_sam;
                }
            }
        }
    }

    d = errs[3];
    for (j = 0; j < len; j++) inl[j] = (d[j] <= th) ? 1 : 0;
    if (doSymCheck && legacy) {
        /* exp_ranF.c:1196-1203: on all points, and on `f` — whatever model the driver computed last — not on F */
        S = inlidxs(d, len, th, inliers);
        FDsSym(u, f, d_check, len);
        for (j = 0; j < len; j++) if (d_check[j] > SymCheck_th) inl[j] = 0;
    } else if (doSymCheck) {
        S = inlidxs(d, len, th, inliers);
        FDsSymidx(u, F, d_check, len, inliers, S.I);
        for (j = 0; j < (int)S.I; j++)
            if (d_check[inliers[j]] > SymCheck_th) inl[j] = 0;      /* list POSITION j, not inliers[j]: :1719-1721 */
    }
    if (DO_LAF_CHECK && final_laf_filter) {                          /* guarded by an uninitialised int in the reference (:1254,:1725) */
        S = inlidxs(d, len, th, inliers);
        FDS1idx(u_1, F, err_laf, len, inliers, S.I);
        for (j = 0; j < (int)S.I; j++) if (err_laf[inliers[j]] > th_laf_check) inl[j] = 0;
        FDS1idx(u_2, F, err_laf, len, inliers, S.I);
        for (j = 0; j < (int)S.I; j++) if (err_laf[inliers[j]] > th_laf_check) inl[j] = 0;
    }

    free(d_check); free(err_laf); free(pool); free(err); free(errorsBest); free(inliers); free(HDsv);
    g_legacy_sym = 0;
    if (stats) {
        stats[DG_ST_SAMPLES] = no_sam; stats[DG_ST_LO_RUNS] = iter_cnt; stats[DG_ST_REJECTED] = g_laf_rej;
        stats[DG_ST_I] = (int)maxS.I; stats[DG_ST_MODELS] = (int)(c->n_fds + c->n_exfds);
        stats[DG_ST_DEGEN] = degen_cnt; stats[DG_ST_IH] = Ihmax; stats[DG_ST_BEST_SAMPLE] = best_sample;
        stats[8] = (int)c->n_fds; stats[9] = (int)c->n_exfds; stats[10] = (int)c->n_hds; stats[11] = (int)c->n_fds_direct;
    }
    return (int)maxS.I;
}

/* ------------------------------------------------------------------ marshalling: bindings.cpp:126-198 / :337-409 */
static void build_u(const double *x1, const double *x2, int n, int dim, int laf, double *u, double *ua, double *ub)
{
    int i;
    for (i = 0; i < n; i++) {
        const double *a = x1 + (size_t)dim * i, *b = x2 + (size_t)dim * i;
        u[6*i+0] = a[0]; u[6*i+1] = a[1]; u[6*i+2] = 1.;
        u[6*i+3] = b[0]; u[6*i+4] = b[1]; u[6*i+5] = 1.;
        if (laf) {
            ua[6*i+0] = a[0] + a[3]; ua[6*i+1] = a[1] + a[5]; ua[6*i+2] = 1.;
            ua[6*i+3] = b[0] + b[3]; ua[6*i+4] = b[1] + b[5]; ua[6*i+5] = 1.;
            ub[6*i+0] = a[0] + a[2]; ub[6*i+1] = a[1] + a[4]; ub[6*i+2] = 1.;
            ub[6*i+3] = b[0] + b[2]; ub[6*i+4] = b[1] + b[4]; ub[6*i+5] = 1.;
        }
    }
}

int dg_oracle_find_fundamental(const double *x1, const double *x2, int n, int dim,
                               double px_th, double conf, int max_iters, int error_type,
                               int sym_check, double laf_coef, int degen, unsigned seed,
                               int final_laf_filter,
                               double *F, unsigned char *mask, int *stats)
{
    dg_ctx c; int laf = laf_coef > 0, ret, i;
    double th = px_th * px_th, sym_th = px_th * px_th * (3.0 * (sym_check ? 1 : 0));   /* bindings.cpp:297-318 */
    double *u, *ua, *ub;
    fds_fn FDS1 = error_type == 1 ? FDsSym : FDs;
    exfds_fn EXFDS1 = error_type == 1 ? exFDsSym : exFDs;
    fdsidx_fn FDSidx1 = error_type == 1 ? FDsSymidx : FDsidx;
    if ((dim != 2 && dim != 6) || n < 8) return -1;        /* bindings.cpp:267-272 */
    memset(&c, 0, sizeof c);
    u = (double *)malloc(sizeof(double) * 6 * (size_t)n);
    ua = (double *)malloc(sizeof(double) * 6 * (size_t)(laf ? n : 1));
    ub = (double *)malloc(sizeof(double) * 6 * (size_t)(laf ? n : 1));
    build_u(x1, x2, n, dim, laf && dim == 6, u, ua, ub);
    for (i = 0; i < 9; i++) F[i] = 0;
    ret = exp_ransacFcustomLAF(&c, u, ua, ub, n, th, laf_coef, conf, max_iters, F, mask, 1, 0,
                               EXFDS1, FDS1, FDSidx1, sym_th, degen, seed, final_laf_filter, stats);
    free(u); free(ua); free(ub); free(c.ht);
    return ret;
}

/* ------------------------------------------------------------------ unit-level exports */
void dg_oracle_rand_stream(unsigned seed, int count, int *out)
{ dg_rng g; int i; dg_srand(&g, seed); for (i = 0; i < count; i++) out[i] = dg_rand(&g); }

/* the main-loop sample stream of exp_ranF.c:1331-1342 / exp_ranH.c:545-552: per iteration srand(seed),
 * `sample_size` draws on the persistent pool, seed = rand().  samidx_out gets pool[n-size..n) per iteration. */
int dg_oracle_sample_stream(unsigned seed0, int n, int sample_size, int iters, int *samidx_out, unsigned *seeds_out)
{
    dg_rng g; int *pool = (int *)malloc(n * sizeof(int)), i, it; unsigned seed;
    for (i = 0; i < n; i++) pool[i] = i;
    dg_srand(&g, seed0); seed = (unsigned)dg_rand(&g);
    for (it = 0; it < iters; it++) {
        if (seeds_out) seeds_out[it] = seed;
        dg_srand(&g, seed);
        for (i = 0; i < sample_size; i++) {
            int s = dg_rand(&g) % (n - i), j = n - i - 1, q = pool[s];
            pool[s] = pool[j]; pool[j] = q;
        }
        seed = (unsigned)dg_rand(&g);
        memcpy(samidx_out + (size_t)it * sample_size, pool + n - sample_size, sample_size * sizeof(int));
    }
    free(pool);
    return 0;
}
int  dg_oracle_nullspace(double *A, double *ns, int n) { int b[18]; return dg_nullspace(A, ns, n, b); }
void dg_oracle_slcm(const double *A, double *B, double *p) { dg_slcm(A, B, p); }
int  dg_oracle_rroots3(const double *po, double *r) { return dg_rroots3(po, r); }
int  dg_oracle_eig_sym(double *a, double *w, int n) { return dg_eig_sym(a, w, n); }
int  dg_oracle_svduv(double *d, double *a, double *u, int m, double *v, int n) { return dg_svduv(d, a, u, m, v, n); }
int  dg_oracle_minv(double *a, int n) { return dg_minv(a, n); }
void dg_oracle_singulF(double *F) { dg_singulF(F); }
void dg_oracle_FDs(const double *u, const double *F, double *p, int len) { FDs(u, F, p, len); }
void dg_oracle_exFDs(const double *u, const double *F, double *p, double *w, int len) { exFDs(u, F, p, w, len); }
void dg_oracle_FDsSym(const double *u, const double *F, double *p, int len) { FDsSym(u, F, p, len); }
void dg_oracle_HDs(const double *u, const double *H, double *p, int len) { HDs(u, H, p, len); }
void dg_oracle_u2f(const double *u, const int *inl, int len, double *F) { u2f(u, inl, len, F); }
void dg_oracle_u2fw(const double *u, const int *inl, const double *w, int len, double *F) { u2fw(u, inl, w, len, F); }
void dg_oracle_u2h(const double *u, const int *inl, int len, double *H) { u2h(u, inl, len, H); }
dg_score dg_oracle_inlidxs(const double *err, int len, double th, int *inl) { return inlidxs(err, len, th, inl); }
int  dg_oracle_nsamples(int ninl, int ptNum, int samsiz, double conf) { return dg_nsamples(ninl, ptNum, samsiz, conf); }
uint32_t dg_oracle_hash(const int *list, int count) { return dg_superfasthash((const unsigned char *)list, count * 4); }
int  dg_oracle_checksample(const double *F, const double *u7, double th, double *H) { return checksample(F, u7, th, H); }
int  dg_oracle_all_ori_valid(const double *F, const double *u, const int *idx, int N) { return all_ori_valid(F, u, idx, N); }


/* ================================================================================================
 * Homography path: exp_ranH.c:29-44, :291-467, :470-930 ; Htools.c:202-370, :428-605, :649-848
 * ============================================================================================== */
/* the four symmetric transfer errors.  Hinv = transpose of the stored H, H1 = minv(Hinv).
 * kind: 1 SymMaxSq, 2 SymMax, 3 SymSumSq, 4 SymSum.  eps: the +1e-10 on the denominators, present in the
 * Sum* metrics and in every i/idx variant but NOT in the plain HDsSymMax / HDsSymMaxSq (Htools.c:310-311,352-353). */
typedef struct { double Hinv[9], H1[9]; } hsym_t;
static void hsym_prepare(const double *H, hsym_t *t)
{
    int i;
    t->Hinv[0] = H[0]; t->Hinv[1] = H[3]; t->Hinv[2] = H[6];
    t->Hinv[3] = H[1]; t->Hinv[4] = H[4]; t->Hinv[5] = H[7];
    t->Hinv[6] = H[2]; t->Hinv[7] = H[5]; t->Hinv[8] = H[8];
    for (i = 0; i < 9; i++) t->H1[i] = t->Hinv[i];
    dg_minv(t->H1, 3);
}
#define HMAX(i,j) ( (i)<(j) ? (j):(i) )
static double hsym_one(const hsym_t *t, const double *u, int kind, int eps)
{
    const double *H1 = t->H1, *Hinv = t->Hinv;
    double a, b, xa, ya, d1, d2, xdiff, ydiff;
    a = H1[6]*u[0] + H1[7]*u[1] + H1[8];
    b = Hinv[6]*u[3] + Hinv[7]*u[4] + Hinv[8];
    if (eps) { a = a + 1e-10; b = b + 1e-10; }
    xa = (H1[0]*u[0] + H1[1]*u[1] + H1[2]) / a;
    ya = (H1[3]*u[0] + H1[4]*u[1] + H1[5]) / a;
    xdiff = u[3] - xa; ydiff = u[4] - ya;
    d1 = xdiff*xdiff + ydiff*ydiff;
    xa = (Hinv[0]*u[3] + Hinv[1]*u[4] + Hinv[2]) / b;
    ya = (Hinv[3]*u[3] + Hinv[4]*u[4] + Hinv[5]) / b;
    xdiff = u[0] - xa; ydiff = u[1] - ya;
    d2 = xdiff*xdiff + ydiff*ydiff;
    switch (kind) {
    case 1: return HMAX(d1, d2);
    case 2: return sqrt(HMAX(d1, d2));
    case 3: return d1 + d2;
    default: return sqrt(d1) + sqrt(d2);
    }
}
/* HDsi / HDsidx (Htools.c:372-410, :607-647): residual rows from the DLT of the ORIGINAL points (lin),
 * Jacobian terms from the point set passed as u6 (the LAF-shifted points in the LAF checks) */
static double HDs_mixed(const double *uo, const double *ul, const double *H)
{
    double z0[9], z1[9], pJ[8], r1 = 0, r2 = 0, a, b, c, d, e, p; int j;
    dlt_rows(uo, z0, z1);
    for (j = 0; j < 9; j++) { r1 += H[j] * z0[j]; r2 += H[j] * z1[j]; }
    a = H[0] - H[2] * ul[0];
    b = H[3] - H[5] * ul[0];
    c = -H[8] - H[2] * ul[3] - H[5] * ul[4];
    d = H[1] - H[2] * ul[1];
    e = H[4] - H[5] * ul[1];
    dg_pinvJ(a, b, c, d, e, pJ);
    p = 0;
    for (j = 0; j < 4; j++) { a = pJ[j] * r1 + pJ[j+4] * r2; p += a * a; }
    return p;
}
/* metric selected by bindings.cpp:64-107: full pass over all points (HDS1) */
static void HDS_full(int kind, const double *u, const double *H, double *p, int len)
{
    int i;
    if (kind == 0) { HDs(u, H, p, len); return; }
    { hsym_t t; hsym_prepare(H, &t); for (i = 0; i < len; i++) p[i] = hsym_one(&t, u + 6*i, kind, kind >= 3); }
}
/* unit-level wrapper: kind 0 Sampson, 1 SymMaxSq, 2 SymMax, 3 SymSumSq, 4 SymSum (bindings.cpp:64-107 order) */
void dg_oracle_HDS_full(int kind, const double *u, const double *H, double *p, int len) { HDS_full(kind, u, H, p, len); }
/* HDSi1 / HDSidx1 value for point id on point set ul (original points uo) */
static double HDS_sub(int kind, const hsym_t *t, const double *uo, const double *ul, const double *H)
{
    if (kind == 0) return HDs_mixed(uo, ul, H);
    return hsym_one(t, ul, kind, 1);
}

static int HcloseToSingular(const double *h)               /* exp_ranH.c:29-44 */
{
    double v, tol; int i;
    v = dg_det3(h);
    tol = h[8];
    if (tol == 0) { for (i = 0; i < 9; ++i) tol += h[i]*h[i]; tol = sqrt(tol); tol *= 0.001; }
    tol = tol*tol*tol;
    return (fabs(v/tol) < 1e-2);
}

static int all_Hori_valid(const double *us, const int *idx)   /* Htools.c:821-848; crossprod = crossprod_st(..,1) */
{
    double p[3], q[3]; const double *a, *b, *c, *d;
    a = us + 6*idx[0]; b = us + 6*idx[1]; c = us + 6*idx[2]; d = us + 6*idx[3];
    dg_crossprod_st(p, a, b, 1); dg_crossprod_st(q, a+3, b+3, 1);
    if ((p[0]*c[0]+p[1]*c[1]+p[2]*c[2])*(q[0]*c[3]+q[1]*c[4]+q[2]*c[5]) < 0) return 0;
    if ((p[0]*d[0]+p[1]*d[1]+p[2]*d[2])*(q[0]*d[3]+q[1]*d[4]+q[2]*d[5]) < 0) return 0;
    dg_crossprod_st(p, c, d, 1); dg_crossprod_st(q, c+3, d+3, 1);
    if ((p[0]*a[0]+p[1]*a[1]+p[2]*a[2])*(q[0]*a[3]+q[1]*a[4]+q[2]*a[5]) < 0) return 0;
    if ((p[0]*b[0]+p[1]*b[1]+p[2]*b[2])*(q[0]*b[3]+q[1]*b[4]+q[2]*b[5]) < 0) return 0;
    return 1;
}

/* exp_ranH.c:291-412 */
static dg_score exp_iterHcustom(dg_ctx *c, const double *u, int len, int *inliers, double th, double ths, int steps,
                                double *H, double **errs, int iterID, unsigned inlLimit, int kind)
{
    double *d = errs[1], h[9], dth; int it;
    dg_score maxS = {0,0,0,0}, S = {0,0,0,0}, Ss; int *detachedInl; unsigned detachedCount;
    int iterIDret; uint32_t hash;
    dth = (ths - th) / (steps);
    maxS = inlidxs(errs[4], len, th, inliers);
    TRACE2(20, maxS.I, maxS.J);
    if (maxS.I < 4) return S;
    S = inlidxs(errs[4], len, th*MWM, inliers);
    detachedCount = (unsigned)(int)(S.I * 1);
    if (detachedCount > inlLimit) detachedCount = inlLimit;
    if (detachedCount < 4) detachedCount = 4;
    if (detachedCount >= S.I) u2h(u, inliers, S.I, h);
    else { detachedInl = randsubset(c, inliers, S.I, detachedCount); u2h(u, detachedInl, detachedCount, h); }
    for (it = 0; it < steps; it++) {
        HDS_full(kind, u, h, d, len); c->n_hds++; TRACE(2, h);
        Ss = inlidxs(d, len, th, inliers);
        TRACE2(21, Ss.I, Ss.J);
        hash = dg_superfasthash((const unsigned char *)inliers, (int)(Ss.I * sizeof(*inliers)));
        iterIDret = htContains(c, hash, Ss.I, iterID);
        if (iterIDret != -1 && iterIDret != iterID) { TRACE2(23, iterIDret, 0); S.I = 0; S.J = 0; return S; }
        if (iterIDret == -1) htInsert(c, hash, Ss.I, iterID);
        S = inlidxs(d, len, ths*MWM, inliers);
        TRACE2(24, S.I, 0);
        if (scoreLess(maxS, Ss)) {
            maxS = Ss; errs[1] = errs[0]; errs[0] = d; d = errs[1];
            memcpy(H, h, 9 * sizeof(double));
        }
        if (S.I < 4) return maxS;
        detachedCount = (unsigned)(int)(S.I * 1);
        if (detachedCount > inlLimit) detachedCount = inlLimit;
        if (detachedCount < 4) detachedCount = 4;
        if (detachedCount >= S.I) u2h(u, inliers, S.I, h);
        else { detachedInl = randsubset(c, inliers, S.I, detachedCount); u2h(u, detachedInl, detachedCount, h); }
        ths -= dth;
    }
    HDS_full(kind, u, h, d, len); c->n_hds++; TRACE(2, h);
    S = inlidxs(d, len, th, inliers);
    TRACE2(22, S.I, S.J);
    if (scoreLess(maxS, S)) {
        maxS = S; errs[1] = errs[0]; errs[0] = d;
        memcpy(H, h, 9 * sizeof(double));
    }
    return maxS;
}

/* exp_ranH.c:415-467 */
static dg_score exp_inHranicustom(dg_ctx *c, const double *u, int len, int *inliers, int ninl, double th, double **errs,
                                  double *H, int rep, int *iterID, unsigned inlLimit, int kind)
{
    int ssiz, i; dg_score S, maxS = {0,0,0,0}; double *d, h[9]; int *sample, *intbuff;
    if (ninl < 8) return maxS;
    intbuff = (int *)malloc(sizeof(int) * len);
    ssiz = ninl / 2; if (ssiz > 12) ssiz = 12;
    d = errs[2]; errs[2] = errs[0]; errs[0] = d;
    for (i = 0; i < rep; i++) {
        sample = randsubset(c, inliers, ninl, ssiz);
        u2h(u, sample, ssiz, h);
        HDS_full(kind, u, h, errs[0], len); c->n_hds++; TRACE(2, h);
        errs[4] = errs[0];
        S = exp_iterHcustom(c, u, len, intbuff, th, TC*th, ILSQ_ITERS, h, errs, ++*iterID, inlLimit, kind);
        if (scoreLess(maxS, S)) {
            maxS = S; d = errs[2]; errs[2] = errs[0]; errs[0] = d;
            memcpy(H, h, 9 * sizeof(double));
        }
    }
    d = errs[2]; errs[2] = errs[0]; errs[0] = d;
    free(intbuff);
    return maxS;
}

/* exp_ranH.c:470-930 with iter_type = 4, oriented_constraint = 1, inlLimit = 0 (bindings.cpp:206-222) */
static dg_score exp_ransacHcustomLAF(dg_ctx *c, const double *u, const double *u_1, const double *u_2, int len, double th,
                                     double laf_coef, double conf, int max_sam, double *H, unsigned char *inl,
                                     unsigned inlLimit, int kind, double SymCheck_th, unsigned seed0, int *stats)
{
    int *pool, no_sam, new_sam, *samidx, bestsamidx[4];
    double M[81], sol[81], *h, *err, *d, *d_check, *errs[5];
    int i, j, *inliers, *inliersS; char do_update = 0;
    dg_score maxS = {0,0,0,0}, maxSs = {0,0,0,0}, S = {0,0,0,0}, Scheck = {0,0,0,0};
    unsigned seed; int do_iterate, iter_cnt = 0, no_rej = 0, iterID = 0; char new_max = 0;
    const int doSymCheck = SymCheck_th > 0, DO_LAF_CHECK = laf_coef > 0; const double th_laf_check = laf_coef * th;
    int p1_inliers = 0, nullspace_buff[18], nullsize, best_sample = 0, accepted = 0;
    hsym_t hs;
    if (inlLimit == 0) inlLimit = 1000000;
    h = sol;
    dg_srand(&c->rng, seed0);
    c->ht_n = 0;
    pool = (int *)malloc(len * sizeof(int));
    for (i = 0; i < len; i++) pool[i] = i;
    err = (double *)calloc((size_t)len * 4, sizeof(double));
    d_check = (double *)malloc(len * sizeof(double));
    for (i = 0; i < 4; i++) errs[i] = err + (size_t)i * len;
    errs[4] = errs[3];
    inliers = (int *)malloc(sizeof(int) * len);
    inliersS = (int *)malloc(sizeof(int) * len);
    no_sam = 0;
    seed = (unsigned)dg_rand(&c->rng);
    samidx = pool + len - 4;
    for (i = 0; i < 81; i++) sol[i] = 0;

    while (no_sam < max_sam) {
        no_sam++;
        dg_srand(&c->rng, seed);
        /* multirsampleT(Z,9,2,pool,4,len,M) (rtools.c:136-156): rows 2i,2i+1 = DLT rows of the i-th drawn point */
        for (i = 0; i < 4; i++) {
            int s = dg_rand(&c->rng) % (len - i), jx = len - i - 1, q = pool[s];
            pool[s] = pool[jx]; pool[jx] = q;
            dlt_rows(u + 6*q, M + 18*i, M + 18*i + 9);
        }
        seed = (unsigned)dg_rand(&c->rng);
        if (!all_Hori_valid(u, samidx)) { no_rej++; continue; }
        for (i = 72; i < 81; ++i) M[i] = 0.0;
        nullsize = dg_nullspace(M, sol, 9, nullspace_buff);
        if (nullsize != 1) { no_rej++; continue; }
        if (HcloseToSingular(h)) { no_rej++; continue; }

        d = errs[0];
        HDS_full(kind, u, h, d, len); c->n_hds++; TRACE(2, h);
        S = inlidxs(d, len, th, inliersS);
        if (scoreLess(maxS, S)) {
            if (doSymCheck) {
                hsym_prepare(h, &hs);
                S.Is = 0;
                for (j = 0; j < (int)S.I; j++) if (hsym_one(&hs, u + 6*inliersS[j], 2, 1) <= SymCheck_th) S.Is++;   /* HDsSymMaxidx */
                if (S.Is < maxS.Is) continue;
            }
            if (DO_LAF_CHECK) {
                Scheck = inlidxs(d, len, th, inliersS);
                hsym_prepare(h, &hs);
                for (j = 0; j < (int)Scheck.I; j++)
                    if (HDS_sub(kind, &hs, u + 6*inliersS[j], u_1 + 6*inliersS[j], h) <= th_laf_check) p1_inliers++;   /* never reset: exp_ranH.c:501,605 */
                if (p1_inliers < (int)maxS.Ilafs) continue;
                S.Ilafs = 0;
                for (j = 0; j < (int)Scheck.I; j++)
                    if (HDS_sub(kind, &hs, u + 6*inliersS[j], u_2 + 6*inliersS[j], h) <= th_laf_check) S.Ilafs++;
                S.Ilafs = (int)S.Ilafs < p1_inliers ? S.Ilafs : (unsigned)p1_inliers;
                if (S.Ilafs < maxS.Ilafs) continue;
            }
            errs[0] = errs[3]; errs[3] = d;
            maxS = S; new_max = 1; accepted = 1; best_sample = no_sam;
            memcpy(H, h, 9 * sizeof(double));
        }
        if (scoreLess(maxSs, S)) {
            do_iterate = no_sam > ITER_SAM;
            maxSs = S; errs[4] = d;
            memcpy(bestsamidx, samidx, 4 * sizeof(int));
        } else do_iterate = 0;
        if ((no_sam >= ITER_SAM) && (iter_cnt == 0) && (maxSs.I > 4)) do_iterate = 1;

        if (do_iterate) {
            iter_cnt++;
            d = errs[0];
            S = inlidxs(errs[4], len, TC*th*MWM, inliers);
            TRACE2(1, S.I, no_sam);
            u2h(u, inliers, S.I, h);
            HDS_full(kind, u, h, d, len); c->n_hds++; TRACE(2, h);
            S = inlidxs(d, len, th, inliers);
            TRACE2(2, S.I, S.J);
            S = exp_inHranicustom(c, u, len, inliers, S.I, th, errs, h, RAN_REP, &iterID, inlLimit, kind);
            TRACE2(3, S.I, S.J);
            if (scoreLess(maxS, S) && !HcloseToSingular(h)) {
                do_update = 1;
                if (doSymCheck) {
                    /* `d` still is the buffer that was errs[0] before the LO rotated the pointers: whatever
                     * residuals were written there last define the "current inliers" (exp_ranH.c:708) */
                    Scheck = inlidxs(d, len, th, inliersS);
                    TRACE2(4, Scheck.I, Scheck.J);
                    hsym_prepare(h, &hs);
                    S.Is = 0;
                    for (j = 0; j < (int)Scheck.I; j++) if (hsym_one(&hs, u + 6*inliersS[j], 2, 1) <= SymCheck_th) S.Is++;
                    if (S.Is < maxS.Is) do_update = 0;
                }
                if (do_update && DO_LAF_CHECK) {
                    Scheck = inlidxs(d, len, th, inliersS);
                    hsym_prepare(h, &hs);
                    for (j = 0; j < (int)Scheck.I; j++)
                        if (HDS_sub(kind, &hs, u + 6*inliersS[j], u_1 + 6*inliersS[j], h) <= th_laf_check) p1_inliers++;
                    S.Ilafs = 0;
                    for (j = 0; j < (int)Scheck.I; j++)
                        if (HDS_sub(kind, &hs, u + 6*inliersS[j], u_2 + 6*inliersS[j], h) <= th_laf_check) S.Ilafs++;
                    S.Ilafs = (int)S.Ilafs < p1_inliers ? S.Ilafs : (unsigned)p1_inliers;
                    if (S.Ilafs < maxS.Ilafs) do_update = 0;
                }
                if (do_update) {
                    d = errs[0]; errs[0] = errs[3]; errs[3] = d;
                    maxS = S; new_max = 1; accepted = 1; best_sample = no_sam;
                    memcpy(H, h, 9 * sizeof(double));
                }
            }
        }
        if (new_max) {
            new_sam = dg_nsamples(maxS.I + 1, len, 4, conf);
            if (new_sam < max_sam) max_sam = new_sam;
            new_max = 0;
        }
    }
    if (iter_cnt == 0) {                                     /* exp_ranH.c:759-862 */
        iter_cnt++;
        d = errs[0];
        S = inlidxs(errs[4], len, TC*th*MWM, inliers);
        u2h(u, inliers, S.I, h);
        HDS_full(kind, u, h, d, len); c->n_hds++; TRACE(2, h);
        S = inlidxs(d, len, th, inliers);
        S = exp_inHranicustom(c, u, len, inliers, S.I, th, errs, h, RAN_REP, &iterID, inlLimit, kind);
        if (scoreLess(maxS, S) && !HcloseToSingular(h)) {
            do_update = 1;
            if (doSymCheck) {
                Scheck = inlidxs(d, len, th, inliersS);
                hsym_prepare(h, &hs);
                S.Is = 0;
                for (j = 0; j < (int)Scheck.I; j++) if (hsym_one(&hs, u + 6*inliersS[j], 2, 1) <= SymCheck_th) S.Is++;   /* HDsiSymMax */
                if (S.Is < maxS.Is) do_update = 0;
            }
            if (do_update && DO_LAF_CHECK) {
                Scheck = inlidxs(d, len, th, inliersS);
                hsym_prepare(h, &hs);
                for (j = 0; j < (int)Scheck.I; j++)
                    if (HDS_sub(kind, &hs, u + 6*inliersS[j], u_1 + 6*inliersS[j], h) <= th_laf_check) p1_inliers++;
                S.Ilafs = 0;
                for (j = 0; j < (int)Scheck.I; j++)
                    if (HDS_sub(kind, &hs, u + 6*inliersS[j], u_2 + 6*inliersS[j], h) <= th_laf_check) S.Ilafs++;
                S.Ilafs = (int)S.Ilafs < p1_inliers ? S.Ilafs : (unsigned)p1_inliers;
                if (S.Ilafs < maxS.Ilafs) do_update = 0;
            }
            if (do_update) {
                d = errs[0]; errs[0] = errs[3]; errs[3] = d;
                maxS = S; accepted = 1; best_sample = no_sam;
                memcpy(H, h, 9 * sizeof(double));
            }
        }
    }
    d = errs[3];
    if (!accepted) { for (j = 0; j < len; j++) inl[j] = 0; }
    else {
        for (j = 0; j < len; j++) inl[j] = (d[j] <= th) ? 1 : 0;
        if (doSymCheck) {
            Scheck = inlidxs(d, len, th, inliersS);
            hsym_prepare(H, &hs);
            for (j = 0; j < (int)Scheck.I; j++) if (hsym_one(&hs, u + 6*inliersS[j], 2, 1) > SymCheck_th) inl[inliersS[j]] = 0;
        }
        if (DO_LAF_CHECK) {
            Scheck = inlidxs(d, len, th, inliersS);
            hsym_prepare(H, &hs);
            for (j = 0; j < (int)Scheck.I; j++) if (HDS_sub(kind, &hs, u + 6*inliersS[j], u_1 + 6*inliersS[j], H) > th_laf_check) inl[inliersS[j]] = 0;
            for (j = 0; j < (int)Scheck.I; j++) if (HDS_sub(kind, &hs, u + 6*inliersS[j], u_2 + 6*inliersS[j], H) > th_laf_check) inl[inliersS[j]] = 0;
        }
    }
    free(pool); free(err); free(inliers); free(inliersS); free(d_check);
    (void)bestsamidx;
    if (stats) {
        stats[DG_ST_SAMPLES] = no_sam; stats[DG_ST_LO_RUNS] = iter_cnt; stats[DG_ST_REJECTED] = no_rej;
        stats[DG_ST_I] = (int)maxS.I; stats[DG_ST_MODELS] = (int)c->n_hds; stats[DG_ST_DEGEN] = 0; stats[DG_ST_IH] = 0;
        stats[DG_ST_BEST_SAMPLE] = best_sample; stats[8] = (int)c->n_hds; stats[9] = 0; stats[10] = 0; stats[11] = 0;
    }
    return maxS;
}

int dg_oracle_find_homography(const double *x1, const double *x2, int n, int dim,
                              double px_th, double conf, int max_iters, int error_type,
                              int sym_check, double laf_coef, unsigned seed,
                              double *H, unsigned char *mask, int *stats)
{
    dg_ctx c; int laf = laf_coef > 0 && dim == 6, i; double th, sym_th, coef = 3.0 * (sym_check ? 1 : 0);
    double *u, *ua, *ub; dg_score S;
    if ((dim != 2 && dim != 6) || n < 4) return -1;          /* bindings.cpp:32-37 */
    switch (error_type) {                                    /* bindings.cpp:64-107 */
    case 1:  th = px_th*px_th; sym_th = 0; break;
    case 2:  th = px_th;       sym_th = 0; break;
    case 3:  th = px_th*px_th; sym_th = px_th*coef; break;
    case 4:  th = px_th;       sym_th = px_th*coef; break;
    default: th = px_th*px_th; sym_th = px_th*coef; error_type = 0; break;
    }
    memset(&c, 0, sizeof c);
    u = (double *)malloc(sizeof(double) * 6 * (size_t)n);
    ua = (double *)malloc(sizeof(double) * 6 * (size_t)(laf ? n : 1));
    ub = (double *)malloc(sizeof(double) * 6 * (size_t)(laf ? n : 1));
    build_u(x1, x2, n, dim, laf, u, ua, ub);
    for (i = 0; i < 9; i++) H[i] = 0;
    S = exp_ransacHcustomLAF(&c, u, ua, ub, n, th, laf ? laf_coef : 0.0, conf, max_iters, H, mask, 0, error_type, sym_th, seed, stats);
    free(u); free(ua); free(ub); free(c.ht);
    return (int)S.I;
}

/* ------------------------------------------------------------------ ranH2el.c: RANSAC on ellipse-to-ellipse (LAF) correspondences
 * (SURVEY.md 8f #4).  Minimal sample = 2 correspondences, model from the 14 x 15 system of Chum & Matas (ICPR 2012) through
 * the reference's own Gauss-Jordan null space; scoring and the LO are those of ranH.c with a second, wider threshold.
 * u10 per correspondence: x1 y1 a1 b1 c1 | x2 y2 a2 b2 c2 (LAF = [a 0; b c]); the model maps image 2 to image 1 like every
 * homography of the reference's C layer (ranH2el.c:38-45 builds u6 = x1 y1 1 x2 y2 1 for HDs / u2h). */
#define H2_TAU (18.0*18.0/7.0/7.0)                                 /* ranH2el.h:31 */
static void getTransf(const double *u10, double *N, double *D)      /* ranH2el.c:209-230, column-wise */
{
    D[0] = u10[2]; D[1] = u10[3]; D[2] = 0; D[3] = 0; D[4] = u10[4]; D[5] = 0; D[6] = u10[0]; D[7] = u10[1]; D[8] = 1;
    N[0] = 1 / u10[7];
    N[1] = - u10[8] / u10[7] / u10[9];
    N[2] = 0; N[3] = 0;
    N[4] = 1 / u10[9];
    N[5] = 0;
    N[6] = - u10[5] / u10[7];
    N[7] = (u10[8]*u10[5] - u10[7]*u10[6]) / u10[7] / u10[9];
    N[8] = 1;
}
/* ranH2el.c:342-360 Zu and :382-399 Znd for len = 2: Z is 14 x 15 column-major (leading dimension 14) */
static void Zu2(double *Z, const double *u)
{
    const int ld = 14; const double u1 = u[0], u2 = u[1], u4 = u[5], u5 = u[6];
    Z[0 + 0*ld] = -1; Z[0 + 6*ld] = u1;
    Z[1 + 1*ld] = -1; Z[1 + 7*ld] = u1;
    Z[2 + 2*ld] = -1; Z[2 + 6*ld] = - u1 * u4; Z[2 + 7*ld] = - u1 * u5;
    Z[3 + 3*ld] = -1; Z[3 + 6*ld] = u2;
    Z[4 + 4*ld] = -1; Z[4 + 7*ld] = u2;
    Z[5 + 5*ld] = -1; Z[5 + 6*ld] = - u2 * u4; Z[5 + 7*ld] = - u2 * u5;
    Z[6 + 8*ld] = -1; Z[6 + 6*ld] = - u4; Z[6 + 7*ld] = - u5;
}
static void Znd2(double *Z, const double *A, const double *B)
{
    const int ld = 14;
    /* ranH2el.h:9-27: _aK / _bK read A and B transposed */
    const double a1 = A[0], a2 = A[3], a3 = A[6], a4 = A[1], a5 = A[4], a6 = A[7];
    const double b1 = B[0], b2 = B[3], b3 = B[6], b4 = B[1], b5 = B[4], b6 = B[7];
    Z[2 + 2*ld] = a3; Z[5 + 2*ld] = a6; Z[6 + 2*ld] = 1;
    Z[0 + 0*ld] = a2*b1 - a1*b4; Z[1 + 0*ld] = a2*b2 - a1*b5; Z[2 + 0*ld] = a2*b3 - a1*b6;
    Z[3 + 0*ld] = a5*b1 - a4*b4; Z[4 + 0*ld] = a5*b2 - a4*b5; Z[5 + 0*ld] = a5*b3 - a4*b6;
    Z[0 + 1*ld] = a1*b1 + a2*b4; Z[1 + 1*ld] = a1*b2 + a2*b5; Z[2 + 1*ld] = a1*b3 + a2*b6;
    Z[3 + 1*ld] = a4*b1 + a5*b4; Z[4 + 1*ld] = a4*b2 + a5*b5; Z[5 + 1*ld] = a4*b3 + a5*b6;
}
/* ranH2el.c:232-283 with do_norm = 0.  U (15 x 15) persists across calls like nothing in the reference does: there it is an
 * uninitialised stack array of which nullspace() writes the first `nullsize` rows; h = its first 9 entries, which are
 * written whenever nullsize >= 1.  With nullsize == 0 the reference copies stack garbage and then discards the sample
 * (return value != 0), so zeroes serve. */
static int A2toRH(const double *u10, const int *samidx, double *h)
{
    double Z[15*15], ZT[15*15], U[15*15], N1[9], D1[9], N2[9], D2[9], t; int i, j, nullsize, nb[30];
    getTransf(u10 + 10*samidx[0], N1, D1);
    getTransf(u10 + 10*samidx[1], N2, D2);
    for (i = 0; i < 15*15; i++) { Z[i] = 0.0; U[i] = 0.0; }
    Zu2(Z, u10 + 10*samidx[0]);
    Zu2(Z + 7, u10 + 10*samidx[1]);
    Znd2(Z + 2*7*9, D1, N1);
    Znd2(Z + 2*7*9 + 2*7*3 + 7, D2, N2);
    for (i = 0; i < 14; i++) for (j = 0; j < 15; j++) ZT[i*15 + j] = Z[j*14 + i];   /* mattr(ZT, Z, 15, 14) */
    for (i = 14*15; i < 15*15; i++) ZT[i] = 0;
    nullsize = dg_nullspace(ZT, U, 15, nb);
    memcpy(h, U, 9 * sizeof(double));
    t = h[1]; h[1] = h[3]; h[3] = t; t = h[2]; h[2] = h[6]; h[6] = t; t = h[5]; h[5] = h[7]; h[7] = t;   /* trnm(h, 3) */
    return nullsize != 1;
}
static double det3(const double *A)                                  /* utools.c:196-202 */
{
    double r = (A[0]*A[4]*A[8] + A[2]*A[3]*A[7] + A[1]*A[5]*A[6]);
    r -= (A[2]*A[4]*A[6] + A[0]*A[5]*A[7] + A[1]*A[3]*A[8]);
    return r;
}

/* ranH2el.c:19-206.  inHraniEl (ranH2el.c:493-543) is ranH.c's inHrani plus a minimal-sample branch (ssiz < 4) that its
 * loLimit = 8 never reaches, so inHrani above stands in for it.  seed0: the caller's srand() value; the driver itself only
 * draws `seed = rand()` (ranH2el.c:71). */
static dg_score ransacH2el(dg_ctx *c, const double *u10, int len, double th, double conf, int max_sam, double *H, unsigned char *inl,
                           int do_lo, int inlLimit, unsigned seed0, int *stats)
{
    int *pool, no_sam, new_sam, *samidx, i, j, *inliers, new_max, do_iterate, iter_cnt = 0, rej_cnt = 0, lo;
    double *u6, *err, *d, h[9], *errs[5], tol, v;
    dg_score maxS = {0,0,0,0}, maxSs = {0,0,0,0}, S;
    unsigned seed;
    if (inlLimit == 0) inlLimit = 0x7fffffff;
    u6 = (double *)malloc(6 * (size_t)len * sizeof(double));
    for (i = 0; i < len; ++i) { u6[i*6+0] = u10[i*10+0]; u6[i*6+1] = u10[i*10+1]; u6[i*6+2] = 1; u6[i*6+3] = u10[i*10+5]; u6[i*6+4] = u10[i*10+6];
        u6[i*6+5] = 1; }
    pool = (int *)malloc(len * sizeof(int));
    for (i = 0; i < len; i++) pool[i] = i;
    samidx = pool + len - 2;
    err = (double *)calloc((size_t)len * 4, sizeof(double));
    for (i = 0; i < 4; i++) errs[i] = err + (size_t)i * len;
    errs[4] = errs[3];
    inliers = (int *)malloc(len * sizeof(int));
    for (i = 0; i < 9; i++) h[i] = 0;
    no_sam = 0;
    dg_srand(&c->rng, seed0);
    seed = (unsigned)dg_rand(&c->rng);
    for (lo = 0; ; ) {
        if (!lo) {
            if (!(no_sam < max_sam)) { if (do_lo && !iter_cnt) lo = 2; else break; }   /* :163 "no LO's so far: make one now" */
        }
        if (!lo) {
            no_sam++; new_max = 0; do_iterate = 0;
            dg_srand(&c->rng, seed);
            randsubset(c, pool, len, 2);
            seed = (unsigned)dg_rand(&c->rng);
            if (A2toRH(u10, samidx, h)) continue;
            v = det3(h); tol = h[8]; tol = tol*tol*tol;
            if (fabs(v/tol) < 10e-2) continue;
            d = errs[0];
            HDs(u6, h, d, len); c->n_hds++;
            S = inlidxs(d, len, th, inliers);
            if (scoreLess(maxS, S)) { maxS = S; errs[0] = errs[3]; errs[3] = d; memcpy(H, h, 9*sizeof(double)); new_max = 1; }
            S = inlidxs(d, len, th*H2_TAU, inliers);
            if (scoreLess(maxSs, S)) {
                maxSs = S; do_iterate = no_sam > ITER_SAM;
                if (!new_max) { errs[0] = errs[2]; errs[2] = d; }
                errs[4] = d;
            }
            if (no_sam >= ITER_SAM && iter_cnt == 0 && maxSs.I > 4) do_iterate = 1;
            if (do_iterate && do_lo) lo = 1;
        }
        if (lo) {                                                   /* :128-151 and :165-187, the same block twice */
            iter_cnt++;
            d = errs[0];
            S = inlidxs(errs[4], len, TC*th*H2_TAU, inliers);
            u2h(u6, inliers, S.I, h);                               /* fewer than 4 ids: h keeps the last sample's model */
            HDs(u6, h, d, len); c->n_hds++;
            S = inlidxs(d, len, th, inliers);
            S = inHrani(c, u6, len, inliers, S.I, th, errs, h, (unsigned)inlLimit);
            tol = h[8]; tol = tol*tol*tol;
            if (scoreLess(maxS, S) && (fabs(det3(h)/tol) > 10e-2)) {
                maxS = S; d = errs[0]; errs[0] = errs[3]; errs[3] = d; memcpy(H, h, 9*sizeof(double)); new_max = 1;
            }
            if (lo == 2) break;
            lo = 0;
        }
        if (new_max) { new_sam = dg_nsamples(maxS.I + 1, len, 2, conf); if (new_sam < max_sam) max_sam = new_sam; }
    }
    if (inl) { d = errs[3]; for (j = 0; j < len; j++) inl[j] = d[j] <= th ? 1 : 0; }
    if (stats) {
        memset(stats, 0, sizeof(int) * DG_ST_COUNT);
        stats[DG_ST_SAMPLES] = no_sam; stats[DG_ST_LO_RUNS] = iter_cnt; stats[DG_ST_REJECTED] = rej_cnt;
        stats[DG_ST_I] = (int)maxS.I; stats[DG_ST_MODELS] = (int)c->n_hds;
    }
    free(pool); free(err); free(inliers); free(u6);
    return maxS;
}

/* u10: [n, 10] row-major.  th is the threshold on the (squared) transfer error HDs returns, as ranH2el.h:35 takes it. */
int dg_oracle_ransacH2el(const double *u10, int n, double th, double conf, int max_iters, int do_lo, int inl_limit, unsigned seed,
                         double *H, unsigned char *mask, int *stats)
{
    dg_ctx c; dg_score S; int i;
    if (n < 2) return -1;
    memset(&c, 0, sizeof c);
    for (i = 0; i < 9; i++) H[i] = 0;
    S = ransacH2el(&c, u10, n, th, conf, max_iters, H, mask, do_lo, inl_limit, seed, stats);
    free(c.ht);
    return (int)S.I;
}
