/* Non-minimal solvers (normalised least squares) of the LO / DEGENSAC steps.
 *   - "small" variants (<= 16 correspondences): one wave, the reference's operation order per output number,
 *     bit-compatible with the scalar reference up to the restated dsyev / SVD.
 *   - "big" variants (an inlier list of any length): the whole workgroup accumulates the Hartley
 *     normalisation and the 9x9 normal matrix in parallel, lane 0 finishes (eig, rank-2, denorm).
 */
#ifndef DG_LSQ_H
#define DG_LSQ_H
#include "dg_kernel_common.h"

struct dg_lsq_scratch {
    double Z[24 * 9];          /* design matrix of a small problem (<= 14 F rows or 2*12 H rows) */
    double V[81], D[9];
    double A1[3], A2[3];
    double Z8[72], U9[81], V8[64], D8[8];
    double part[DG_NW][48];    /* per-wave partial sums of the big variants */
    double out[9];
    double px[16 * 4];         /* gathered coordinates of a small problem */
    dg_eig_ws ews;             /* dsyev work vectors */
    double svw[32];            /* svduv work vectors */
};

/* per-wave scratch for the wave-parallel sections (checksample's 5 triplets, innerFH's repetitions): the
 * members the templated small solvers touch have the same names as in dg_lsq_scratch */
struct dg_wave_ws {
    double Z[14 * 9], V[81], D[9], A1[3], A2[3], px[14 * 4];     /* Z .. px also serve as the MSAC-term tile of a wave's pass (dg_wpass_impl): idle then */
    dg_eig_ws ews;
    double H[9], F[9], Ds[7], sDs[7], cpx[20];
    int idx[8], res, cnt;
};

/* utools.c:7-51 normu over k gathered points p[4*i + {0,1,2,3}] = x1,y1,x2,y2 */
template <class PC>
__device__ __forceinline__ void dg_normu_small(PC p, int len, double *A1, double *A2)
{
    int i, j; double a, b;
    for (j = 0; j < 3; j++) { A1[j] = 0; A2[j] = 0; }
    for (j = 0; j < len; j++) { A1[1] += p[4*j]; A1[2] += p[4*j+1]; A2[1] += p[4*j+2]; A2[2] += p[4*j+3]; }
    if (len > 0) for (i = 1; i < 3; i++) { A1[i] /= len; A2[i] /= len; }
    for (j = 0; j < len; j++) {
        a = p[4*j] - A1[1]; b = p[4*j+1] - A1[2]; A1[0] += sqrt(a*a + b*b);
        a = p[4*j+2] - A2[1]; b = p[4*j+3] - A2[2]; A2[0] += sqrt(a*a + b*b);
    }
    if (A1[0] != 0) A1[0] = len * sqrt(2.0) / A1[0];
    if (A2[0] != 0) A2[0] = len * sqrt(2.0) / A2[0];
    A1[1] *= -A1[0]; A1[2] *= -A1[0];
    A2[1] *= -A2[0]; A2[2] *= -A2[0];
}

/* Htools.c:101-114 u2h on exactly 4 gathered points (one lane): the 8 x 9 DLT system laid out as the reference
 * lays it out (entry (j, r) of the 9 x 8 array lin_hg fills, read back as a 9 x 9 array after an in-place 9 x 9
 * transpose; entries the reference never writes are zero here), then its null vector. */
__device__ __noinline__ void dg_u2h_4pt_mv(double *M /* 81 */, double *V /* 81 */, const double *p, double *H)
{
    for (int i = 0; i < 81; i++) { M[i] = 0.; V[i] = 0.; }
    for (int i = 0; i < 4; i++) {
        const double s0 = p[4*i], s1 = p[4*i+1], s3 = p[4*i+2], s4 = p[4*i+3];
        const double z0[9] = {s3, 0, -s0*s3, s4, 0, -s0*s4, 1.0, 0, -s0*1.0};
        const double z1[9] = {0, s3, -s1*s3, 0, s4, -s1*s4, 0, 1.0, -s1*1.0};
        /* flat index e = 8 j + r of the 9 x 8 array is element (e / 9, e % 9) of the 9 x 9 view; transposed */
        for (int j = 0; j < 9; j++) {
            const int e0 = 8*j + 2*i, e1 = e0 + 1;
            M[9*(e0 % 9) + e0 / 9] = z0[j]; M[9*(e1 % 9) + e1 / 9] = z1[j];
        }
    }
    for (int i = 72; i < 81; i++) M[i] = 0.;
    dg_null9<9, 1>(M, V);
    for (int i = 0; i < 9; i++) H[i] = V[i];
}
__device__ __forceinline__ void dg_u2h_4pt(dg_lsq_scratch *s, const double *p, double *H) { dg_u2h_4pt_mv(s->U9, s->V, p, H); }

/* ---- wave-cooperative variants of the small solvers (all 64 lanes of wave 0 call these) ----------------
 * Same arithmetic as the reference's u2f / u2h (Ftools.c:350-458, Htools.c:101-133); the design-matrix rows, the 45 normal-matrix entries and
 * the eigen-solver's inner loops are spread over lanes.  p (gathered coordinates) is in LDS. */
template <class PV, class PZ>
__device__ __forceinline__ void dg_cov9_wave(PV Cv, PZ Z, int rows, int lane)
{
    if (lane < 45) {
        int i = 0; while ((i+1)*(i+2)/2 <= lane) i++;
        int j = lane - i*(i+1)/2;
        double val = 0;
        for (int k = 0; k < rows; k++) val += Z[9*k+i] * Z[9*k+j];
        Cv[9*i+j] = val; Cv[i+9*j] = val;
    }
}

/* long form (> 8 points, normalised LSQ) for any scratch type with members Z, V, D, A1, A2, ews */
template <class SC, class PP, class PF>
__device__ __forceinline__ void dg_u2f_norm_w(SC *s, PP p, const double *wts, int len, PF F, int lane)
{
    double A1[3], A2[3];
    dg_normu_small(p, len, A1, A2);
    if (lane < len) {
        int i = lane; double a[3], b[3];
        a[2] = 1; b[2] = 1;
        a[0] = p[4*i]   * A1[0] + A1[1]; a[1] = p[4*i+1] * A1[0] + A1[2];
        b[0] = p[4*i+2] * A2[0] + A2[1]; b[1] = p[4*i+3] * A2[0] + A2[2];
        for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++) { double z = a[l] * b[k]; if (wts) z *= wts[i]; s->Z[9*i + 3*k + l] = z; }
    }
    if (lane == 0) for (int i = 0; i < 3; i++) { s->A1[i] = A1[i]; s->A2[i] = A2[i]; }
    DG_WSYNC();
    dg_cov9_wave(&s->V[0], &s->Z[0], len, lane);
    DG_WSYNC();
    dg_eig_sym_wave((double *)&s->V[0], (double *)&s->D[0], lane, (dg_eig_ws *)&s->ews);
    if (lane == 0) {
        int j = 0; for (int i = 1; i < 9; i++) if (s->D[i] < s->D[j]) j = i;
        for (int i = 0; i < 9; i++) F[i] = s->V[j*9 + i];
        dg_singulF(F);
        dg_denormF(F, &s->A1[0], &s->A2[0]);
    }
    DG_WSYNC();
}

__device__ __noinline__ void dg_u2f_small_w(dg_lsq_scratch *s, const double *p, const double *wts /* LDS or 0 */, int len, double *F, int lane)
{
    if (len > 8) {
        dg_u2f_norm_w(s, p, wts, len, F, lane);
    } else {
        /* <= 8 points: lin_fm rows + the stride-9 weight pattern (Ftools.c:427-432), left null vector, rank 2 */
        /* entry e = 8*row + point; the weight pattern of Ftools.c:427-432 hits entry e with wts[e % 9] */
        for (int e = lane; e < 72; e += 64) {
            const int r = e >> 3, i = e & 7, k = r / 3, l = r - 3*k;
            double z = 0.;
            if (i < len) {
                const double a = l == 2 ? 1.0 : p[4*i + l], b = k == 2 ? 1.0 : p[4*i + 2 + k];
                z = b * a;
            }
            const int wi = e % 9;
            if (wts && wi < len && wi < 8) z *= wts[wi];
            s->Z8[e] = z;
        }
        DG_WSYNC();
        dg_svd_lastcol_9x8_wave(s->Z8, s->U9, lane);
        if (lane == 0) { for (int i = 0; i < 9; i++) F[i] = s->U9[i]; dg_singulF(F); }
        DG_WSYNC();
    }
}

/* dg_u2f_small_w on a wave's own scratch (dg_wave_ws: Z doubles as the 9 x 8 system, V as the left factor's column) */
__device__ __noinline__ void dg_u2f_small_wave(dg_wave_ws *s_, const double *p_, const double *wts /* LDS or 0 */, int len, double *F_, int lane)
{
    /* the wave's scratch, the gathered points and the model are LDS on every call: address-space-qualified views, so that every access of
     * this function and of the helpers inlined into it is a ds_ instruction (through the generic parameters they were all FLAT: 62 stores and
     * 54 loads).  The callees below take generic pointers again; they begin (prologue) and end (DG_WSYNC) with a wait for the wave's memory
     * operations, so the two paths never meet an unordered pair of accesses. */
    __attribute__((address_space(3))) dg_wave_ws *s = (__attribute__((address_space(3))) dg_wave_ws *)s_;
    const __attribute__((address_space(3))) double *p = (const __attribute__((address_space(3))) double *)p_;
    __attribute__((address_space(3))) double *F = (__attribute__((address_space(3))) double *)F_;
    if (len > 8) {
        dg_u2f_norm_w(s, p, wts, len, F, lane);
    } else {
        for (int e = lane; e < 72; e += 64) {
            const int r = e >> 3, i = e & 7, k = r / 3, l = r - 3*k;
            double z = 0.;
            if (i < len) {
                const double a = l == 2 ? 1.0 : p[4*i + l], b = k == 2 ? 1.0 : p[4*i + 2 + k];
                z = b * a;
            }
            const int wi = e % 9;
            if (wts && wi < len && wi < 8) z *= wts[wi];
            s->Z[e] = z;
        }
        DG_WSYNC();
        dg_svd_lastcol_9x8_wave((double *)&s->Z[0], (double *)&s->V[0], lane);
        if (lane == 0) { for (int i = 0; i < 9; i++) F[i] = s->V[i]; dg_singulF(F); }
        DG_WSYNC();
    }
}

/* > 4 points: normalised DLT for any scratch type with members Z (>= 18*len), V, D, A1, A2, ews */
template <class SC>
__device__ __forceinline__ void dg_u2h_norm_w(SC *s, const double *p, int len, double *H, int lane)
{
    double A1[3], A2[3];
    dg_normu_small(p, len, A1, A2);
    if (lane < len) {
        int i = lane;
        double a0 = p[4*i] * A1[0] + A1[1], a1 = p[4*i+1] * A1[0] + A1[2];
        double b[3] = {p[4*i+2] * A2[0] + A2[1], p[4*i+3] * A2[0] + A2[2], 1.0};
        double *z = s->Z + 18*i;
        for (int j = 0; j < 3; j++) { z[3*j] = b[j]; z[3*j+1] = 0; z[3*j+2] = -a0 * b[j]; }
        for (int j = 0; j < 3; j++) { z[9+3*j] = 0; z[9+3*j+1] = b[j]; z[9+3*j+2] = -a1 * b[j]; }
    }
    if (lane == 0) for (int i = 0; i < 3; i++) { s->A1[i] = A1[i]; s->A2[i] = A2[i]; }
    DG_WSYNC();
    dg_cov9_wave(s->V, s->Z, 2*len, lane);
    DG_WSYNC();
    dg_eig_sym_wave(s->V, s->D, lane, &s->ews);
    if (lane == 0) { for (int i = 0; i < 9; i++) H[i] = s->V[i]; dg_denormH(H, s->A1, s->A2); }
    DG_WSYNC();
}

/* dg_u2h_norm_w without the 2 len x 9 design matrix in LDS (a wave's own scratch holds 14 rows): the 45 lanes of the normal
 * matrix form the entries of the DLT rows on the fly from the normalised coordinates — the same products and the same
 * sums in the same order as lin_hgN + cov_mat (Htools.c:60-99, utools.c:170-184), so the same bits.  len <= 14. */
__device__ __forceinline__ void dg_u2h_norm_wave_noz(dg_wave_ws *s, const double *p, int len, double *H, int lane)
{
    double A1[3], A2[3];
    dg_normu_small(p, len, A1, A2);
    /* normalised coordinates of point i: Z[4 i] = a0, a1, b0, b1 */
    if (lane < len) {
        const int i = lane;
        s->Z[4*i] = p[4*i] * A1[0] + A1[1]; s->Z[4*i+1] = p[4*i+1] * A1[0] + A1[2];
        s->Z[4*i+2] = p[4*i+2] * A2[0] + A2[1]; s->Z[4*i+3] = p[4*i+3] * A2[0] + A2[2];
    }
    if (lane == 0) for (int i = 0; i < 3; i++) { s->A1[i] = A1[i]; s->A2[i] = A2[i]; }
    DG_WSYNC();
    if (lane < 45) {
        int ie = 0; while ((ie+1)*(ie+2)/2 <= lane) ie++;
        const int je = lane - ie*(ie+1)/2;
        /* entry c of DLT row r (0: the x row, 1: the y row) of a point: (b_j, 0, -a0 b_j) resp. (0, b_j, -a1 b_j), j = c / 3 */
        auto zent = [](int r, int c, double a0, double a1, double b0, double b1) {
            const int j = c / 3, m = c - 3 * j;
            const double bj = j == 0 ? b0 : (j == 1 ? b1 : 1.0);
            if (m == 2) return -(r == 0 ? a0 : a1) * bj;
            return m == r ? bj : 0.0;
        };
        double val = 0;
        for (int k = 0; k < len; k++) {
            const double a0 = s->Z[4*k], a1 = s->Z[4*k+1], b0 = s->Z[4*k+2], b1 = s->Z[4*k+3];
            val += zent(0, ie, a0, a1, b0, b1) * zent(0, je, a0, a1, b0, b1);
            val += zent(1, ie, a0, a1, b0, b1) * zent(1, je, a0, a1, b0, b1);
        }
        s->V[9*ie+je] = val; s->V[ie+9*je] = val;
    }
    DG_WSYNC();
    dg_eig_sym_wave(s->V, s->D, lane, &s->ews);
    if (lane == 0) { for (int i = 0; i < 9; i++) H[i] = s->V[i]; dg_denormH(H, s->A1, s->A2); }
    DG_WSYNC();
}

/* ---- two small normalised fits side by side in one wave (dg_eig2.h) --------------------------------------------------------------
 * Problem A in lanes 0..31 with the wave's own scratch (w->V, D, A1, A2, ews), problem B in lanes 32..63 with the block x2
 * (DG_X2_* doubles: V, D, A1, A2, the eigen-solver's work vectors, then room for the caller's per-problem scalars).  The design matrix
 * is not stored: the 45 normal-matrix entries (two passes of 32 lanes per problem) form its entries on the fly from the normalised
 * coordinates (w->Z: A at 0, B at 40) — the products and sums of lin_fmN / lin_hgN + cov_mat in their order (Ftools.c:300-328, Htools.c:60-99,
 * utools.c:170-184), i.e. the bits of dg_u2f_norm_w / dg_u2h_norm_w.  outB = null: one problem (the upper half repeats it in x2).
 * len <= 10, no weights.  All 64 lanes. */
#define DG_X2_V 0
#define DG_X2_D 81
#define DG_X2_A1 90
#define DG_X2_A2 93
#define DG_X2_EWS 96
#define DG_X2_USER 141               /* first double a caller may use for its own second-problem data */
#define DG_X2_DOUBLES 188            /* per wave (checksample: + H, Ds, sDs, idx, five gathered points) */
static_assert(sizeof(dg_eig_ws) == (DG_X2_USER - DG_X2_EWS) * sizeof(double), "x2 layout");
template <bool HOMOG>
__device__ __noinline__ void dg_fit_norm_w2(dg_wave_ws *w, double *x2, const double *pA, const double *pB, int len, double *outA, double *outB, int lane_w)
{
    const bool uph = lane_w >= 32, two = outB != (double *)0;
    const int hl = lane_w & 31;
    const double *p = (uph && two) ? pB : pA;
    double *V = uph ? x2 + DG_X2_V : w->V, *D = uph ? x2 + DG_X2_D : w->D, *A1s = uph ? x2 + DG_X2_A1 : w->A1, *A2s = uph ? x2 + DG_X2_A2 : w->A2;
    double *nz = w->Z + (uph ? 40 : 0);
    double A1[3], A2[3];
    dg_normu_small(p, len, A1, A2);
    if (hl < len) {
        nz[4*hl] = p[4*hl] * A1[0] + A1[1]; nz[4*hl+1] = p[4*hl+1] * A1[0] + A1[2];
        nz[4*hl+2] = p[4*hl+2] * A2[0] + A2[1]; nz[4*hl+3] = p[4*hl+3] * A2[0] + A2[2];
    }
    if (hl == 0) for (int i = 0; i < 3; i++) { A1s[i] = A1[i]; A2s[i] = A2[i]; }
    DG_WSYNC();
    for (int en = hl; en < 45; en += 32) {
        int ie = 0; while ((ie+1)*(ie+2)/2 <= en) ie++;
        const int je = en - ie*(ie+1)/2;
        double val = 0;
        if (HOMOG) {
            /* entry c of DLT row r (0: the x row, 1: the y row) of a point: (b_j, 0, -a0 b_j) resp. (0, b_j, -a1 b_j), j = c / 3 */
            auto zent = [](int r, int c, double a0, double a1, double b0, double b1) {
                const int j = c / 3, m = c - 3 * j;
                const double bj = j == 0 ? b0 : (j == 1 ? b1 : 1.0);
                if (m == 2) return -(r == 0 ? a0 : a1) * bj;
                return m == r ? bj : 0.0;
            };
            for (int k = 0; k < len; k++) {
                const double a0 = nz[4*k], a1 = nz[4*k+1], b0 = nz[4*k+2], b1 = nz[4*k+3];
                val += zent(0, ie, a0, a1, b0, b1) * zent(0, je, a0, a1, b0, b1);
                val += zent(1, ie, a0, a1, b0, b1) * zent(1, je, a0, a1, b0, b1);
            }
        } else {
            /* entry c = 3 k + l of a point's row: a[l] * b[k] with a = (a0, a1, 1), b = (b0, b1, 1) */
            const int li = ie % 3, ki = ie / 3, lj = je % 3, kj = je / 3;
            for (int k = 0; k < len; k++) {
                const double a0 = nz[4*k], a1 = nz[4*k+1], b0 = nz[4*k+2], b1 = nz[4*k+3];
                const double zi = (li == 0 ? a0 : (li == 1 ? a1 : 1.0)) * (ki == 0 ? b0 : (ki == 1 ? b1 : 1.0));
                const double zj = (lj == 0 ? a0 : (lj == 1 ? a1 : 1.0)) * (kj == 0 ? b0 : (kj == 1 ? b1 : 1.0));
                val += zi * zj;
            }
        }
        V[9*ie+je] = val; V[ie+9*je] = val;
    }
    DG_WSYNC();
    dg_eig_sym_wave2(w->V, w->D, &w->ews, x2 + DG_X2_V, x2 + DG_X2_D, (dg_eig_ws *)(x2 + DG_X2_EWS), lane_w);
    if (hl == 0 && (!uph || two)) {
        double *out = uph ? outB : outA;
        if (HOMOG) { for (int i = 0; i < 9; i++) out[i] = V[i]; dg_denormH(out, A1s, A2s); }
        else {
            int j = 0; for (int i = 1; i < 9; i++) if (D[i] < D[j]) j = i;
            for (int i = 0; i < 9; i++) out[i] = V[j*9 + i];
            dg_singulF(out);
            dg_denormF(out, A1s, A2s);
        }
    }
    DG_WSYNC();
}

__device__ __noinline__ void dg_u2h_small_w(dg_lsq_scratch *s, const double *p, int len, double *H, int lane)
{
    if (len < 4) return;
    if (len == 4) { if (lane == 0) dg_u2h_4pt(s, p, H); DG_WSYNC(); return; }
    dg_u2h_norm_w(s, p, len, H, lane);
}

/* ---- normalised LSQ over an id list of any length, in the REFERENCE'S summation order -----------------
 * Hartley-normalised coordinates have zero mean, so several entries of the 9x9 normal matrix are "zero
 * up to summation noise"; the sign LAPACK's dsyev gives the eigenvector depends on that noise.  To return
 * the same model (sign included) as the scalar reference, every sum below is accumulated sequentially
 * over the list exactly like normu (utools.c:7-51) and cov_mat (utools.c:170-184) do — but the 4 + 2 + 45
 * independent sums run in different lanes of wave 0 (one lane per output entry).  `rows2`: 0 = fundamental
 * (lin_fmN, Ftools.c:300-328, one row per point), 1 = homography (lin_hgN, Htools.c:60-99, two rows). */
/* the sequential sums, run by ONE wave on already staged (gathered) correspondences; V, A1o, A2o in LDS */
template <class SC>
__device__ __forceinline__ void dg_lsq_seq_core(SC *s, const dg_pt *stage, int len, int lane, int rows2, double *A1o, double *A2o)
{
    {
        /* centroids: lane l in 0..3 sums coordinate l, in list order */
        double acc = 0;
        {
            const double *sp = (const double *)stage + (lane & 3);
            int j = 0;
            for (; j + 4 <= len; j += 4) { double v0 = sp[4*j], v1 = sp[4*j+4], v2 = sp[4*j+8], v3 = sp[4*j+12]; acc += v0; acc += v1; acc += v2; acc += v3; }
            for (; j < len; j++) acc += sp[4*j];
        }
        if (len > 0) acc /= len;
        double m1x = __shfl(acc, 0, 64), m1y = __shfl(acc, 1, 64), m2x = __shfl(acc, 2, 64), m2y = __shfl(acc, 3, 64);
        /* mean distances: lane 0 image 1, lane 1 image 2 */
        double dsum = 0;
        {
            const double *sp = (const double *)stage + ((lane & 1) ? 2 : 0);
            const double mx = (lane & 1) ? m2x : m1x, my = (lane & 1) ? m2y : m1y;
            /* four points per step so that their loads are in flight together (the stage can be HBM/L2); same order */
            int j = 0;
            for (; j + 4 <= len; j += 4) {
                double a0 = sp[4*j], b0 = sp[4*j+1], a1 = sp[4*j+4], b1 = sp[4*j+5], a2 = sp[4*j+8], b2 = sp[4*j+9], a3 = sp[4*j+12], b3 = sp[4*j+13];
                a0 -= mx; b0 -= my; a1 -= mx; b1 -= my; a2 -= mx; b2 -= my; a3 -= mx; b3 -= my;
                dsum += sqrt(a0*a0 + b0*b0); dsum += sqrt(a1*a1 + b1*b1); dsum += sqrt(a2*a2 + b2*b2); dsum += sqrt(a3*a3 + b3*b3);
            }
            for (; j < len; j++) { double a = sp[4*j] - mx, b = sp[4*j+1] - my; dsum += sqrt(a*a + b*b); }
        }
        double A1[3], A2[3];
        A1[0] = __shfl(dsum, 0, 64); A2[0] = __shfl(dsum, 1, 64);
        if (A1[0] != 0) A1[0] = len * sqrt(2.0) / A1[0];
        if (A2[0] != 0) A2[0] = len * sqrt(2.0) / A2[0];
        A1[1] = m1x * -A1[0]; A1[2] = m1y * -A1[0];
        A2[1] = m2x * -A2[0]; A2[2] = m2y * -A2[0];
        /* normal matrix: lane e < 45 owns entry (ie, je), je <= ie, in cov_mat's enumeration order.  Each lane
         * forms only its own two design-matrix entries: z[3k+l] = a[l]*b[k] (F) or the lin_hgN pattern (H). */
        int ie = 0, je = 0;
        { int e = 0; for (int i = 0; i < 9; i++) for (int q = 0; q <= i; q++) { if (e == lane) { ie = i; je = q; } e++; } }
        const int ki = ie / 3, li = ie % 3, kj = je / 3, lj = je % 3;
        double val = 0;
#define DG_NM_TERM(p) do { \
            double a0 = (p).x1 * A1[0] + A1[1], a1 = (p).y1 * A1[0] + A1[2]; \
            double b0 = (p).x2 * A2[0] + A2[1], b1 = (p).y2 * A2[0] + A2[2]; \
            if (!rows2) { \
                double ai = li == 0 ? a0 : li == 1 ? a1 : 1.0, bi = ki == 0 ? b0 : ki == 1 ? b1 : 1.0; \
                double aj = lj == 0 ? a0 : lj == 1 ? a1 : 1.0, bj = kj == 0 ? b0 : kj == 1 ? b1 : 1.0; \
                val += (ai * bi) * (aj * bj); \
            } else { \
                /* row 0: z[3q] = b[q], z[3q+1] = 0, z[3q+2] = -a0*b[q];  row 1: z[3q] = 0, z[3q+1] = b[q], z[3q+2] = -a1*b[q] */ \
                double bqi = ki == 0 ? b0 : ki == 1 ? b1 : 1.0, bqj = kj == 0 ? b0 : kj == 1 ? b1 : 1.0; \
                double z0i = li == 0 ? bqi : li == 1 ? 0.0 : -a0 * bqi, z0j = lj == 0 ? bqj : lj == 1 ? 0.0 : -a0 * bqj; \
                double z1i = li == 0 ? 0.0 : li == 1 ? bqi : -a1 * bqi, z1j = lj == 0 ? 0.0 : lj == 1 ? bqj : -a1 * bqj; \
                val += z0i * z0j; \
                val += z1i * z1j; \
            } } while (0)
        if (lane < 45) {
            int j = 0;
            for (; j + 8 <= len; j += 8) {                   /* eight loads in flight, terms added in list order */
                const dg_pt p0 = stage[j], p1 = stage[j+1], p2 = stage[j+2], p3 = stage[j+3];
                const dg_pt p4 = stage[j+4], p5 = stage[j+5], p6 = stage[j+6], p7 = stage[j+7];
                DG_NM_TERM(p0); DG_NM_TERM(p1); DG_NM_TERM(p2); DG_NM_TERM(p3);
                DG_NM_TERM(p4); DG_NM_TERM(p5); DG_NM_TERM(p6); DG_NM_TERM(p7);
            }
            for (; j + 4 <= len; j += 4) {
                const dg_pt p0 = stage[j], p1 = stage[j+1], p2 = stage[j+2], p3 = stage[j+3];
                DG_NM_TERM(p0); DG_NM_TERM(p1); DG_NM_TERM(p2); DG_NM_TERM(p3);
            }
            for (; j < len; j++) { const dg_pt p = stage[j]; DG_NM_TERM(p); }
        }
#undef DG_NM_TERM
        if (lane < 45) { s->V[9*ie + je] = val; s->V[ie + 9*je] = val; }
        if (lane == 0) { for (int i = 0; i < 3; i++) { A1o[i] = A1[i]; A2o[i] = A2[i]; } }
    }
}

/* The same sums with the per-point work taken out of the serial loops.  `nthr` threads (a whole workgroup, or one
 * wave) compute in parallel, per listed point, the two centroid distances and then the Hartley-normalised
 * coordinates (identical operations to the serial form, so identical bits); the sequential, reference-order
 * accumulations that remain are plain adds (coordinates, distances) or two to four multiplies and an add (normal
 * matrix) per term.  The plain sums run over CONTIGUOUS arrays (one per coordinate / per image, written by the parallel
 * phases behind the staged points) through dg_seq_sum_from, whose loads run ahead of the add chain: these sums used to
 * pay one cache round trip per four or eight terms, which made them the larger part of a long-list fit.
 * Needs 32 B of scratch per point behind the staged points: stage must hold 2 * len entries.
 * `sync` separates the phases (__syncthreads for a workgroup, a wave barrier for one wave); tid < 64 is the wave that
 * runs the serial parts. */
/* dg_lsq_seq_par by ONE wave with every sequential sum fed from LDS (ltab: 10 * ltab_pts doubles).
 * The staged points are read from global memory once per phase, 64 at a time with the next block's load in flight; a phase's terms —
 * the four coordinates (centroids), the two centroid distances, the design-matrix entries — go through the LDS table in sub-blocks and are
 * added from there, in list order, by the lanes that own the sums.  (Round 6: the sums used to read their terms back from global memory
 * sixteen at a time, two batches in flight — one L2 round trip per sixteen dependent adds: 63 of the 140 us of a u2Fit iteration.)
 * Same terms, same adds, same order as dg_lsq_seq_core / the reference (utools.c:7-51, :170-184); `stage` is left as it was. */
template <class SC>
__device__ __forceinline__ void dg_lsq_seq_wave(SC *s, const dg_pt *stage, int len, int lane, int rows2, double *A1o, double *A2o, double *ltab,
                                                const int ltab_pts)
{
    len = __builtin_amdgcn_readfirstlane(len);
    const int tabd = 10 * ltab_pts;
    /* sub-block sizes: multiples of eight (the sums take their terms eight at a time), at most one load block */
    const int BA = ((tabd / 4) & ~7) < 64 ? ((tabd / 4) & ~7) : 64, BB = ((tabd / 2) & ~7) < 64 ? ((tabd / 2) & ~7) : 64;
    const int fpts = ltab_pts >= 8 ? (ltab_pts & ~7) : ltab_pts;
    /* the terms of all three sweeps are READ with ds_read (dg_seq_sum_impl<3>, the table of the normal matrix): they are written with ds_write
     * too — a FLAT store through the generic pointer and a ds_read of the same address are not ordered against each other.  (A FLAT read of
     * the table also counts as a vector-memory operation: waiting for it waits for the next block's points, which are meant to stay in flight.) */
    typedef __attribute__((address_space(3))) double dg_ldsd;
    dg_ldsd *lt3 = (dg_ldsd *)ltab;
    /* ---- centroids: lane l in 0..3 sums coordinate l */
    double acc = 0;
    {
        dg_pt qn = stage[lane < len ? lane : 0];
        for (int blk = 0; blk < len; blk += 64) {
            const dg_pt q = qn;
            if (blk + 64 < len) qn = stage[blk + 64 + lane < len ? blk + 64 + lane : blk];
            const int bc = len - blk < 64 ? len - blk : 64;
            for (int f0 = 0; f0 < bc; f0 += BA) {
                const int cnt = bc - f0 < BA ? bc - f0 : BA;
                if (lane >= f0 && lane < f0 + cnt) { const int r = lane - f0; lt3[r] = q.x1; lt3[BA + r] = q.y1; lt3[2*BA + r] = q.x2;
                    lt3[3*BA + r] = q.y2; }
                DG_WSYNC_LDS();
                if (lane < 4) acc = dg_seq_sum_impl<3>(ltab + BA * lane, cnt, acc);
                DG_WSYNC_LDS();
            }
        }
    }
    if (len > 0) acc /= len;
    if (lane < 4) s->D[lane] = acc;
    DG_WSYNC_LDS();
    const double m1x = s->D[0], m1y = s->D[1], m2x = s->D[2], m2y = s->D[3];
    /* ---- mean distances to the centroids: lane 0 image 1, lane 1 image 2 */
    double dsum = 0;
    {
        dg_pt qn = stage[lane < len ? lane : 0];
        for (int blk = 0; blk < len; blk += 64) {
            const dg_pt q = qn;
            if (blk + 64 < len) qn = stage[blk + 64 + lane < len ? blk + 64 + lane : blk];
            double a = q.x1 - m1x, b = q.y1 - m1y; const double d1 = sqrt(a*a + b*b);
            a = q.x2 - m2x; b = q.y2 - m2y; const double d2 = sqrt(a*a + b*b);
            const int bc = len - blk < 64 ? len - blk : 64;
            for (int f0 = 0; f0 < bc; f0 += BB) {
                const int cnt = bc - f0 < BB ? bc - f0 : BB;
                if (lane >= f0 && lane < f0 + cnt) { const int r = lane - f0; lt3[r] = d1; lt3[BB + r] = d2; }
                DG_WSYNC_LDS();
                if (lane < 2) dsum = dg_seq_sum_impl<3>(ltab + BB * lane, cnt, dsum);
                DG_WSYNC_LDS();
            }
        }
    }
    double A1[3], A2[3];
    A1[0] = __shfl(dsum, 0, 64); A2[0] = __shfl(dsum, 1, 64);
    if (A1[0] != 0) A1[0] = len * sqrt(2.0) / A1[0];
    if (A2[0] != 0) A2[0] = len * sqrt(2.0) / A2[0];
    A1[1] = m1x * -A1[0]; A1[2] = m1y * -A1[0];
    A2[1] = m2x * -A2[0]; A2[2] = m2y * -A2[0];
    if (lane == 0) { for (int i = 0; i < 3; i++) { A1o[i] = A1[i]; A2o[i] = A2[i]; } }
    DG_WSYNC_LDS();
    /* ---- normal matrix with the per-point work shared: every lane forms the nine (F: z[3k+l] = a_l b_k, lin_fmN) or ten (H: b_q, -a0 b_q,
     * -a1 b_q and a structural zero, lin_hgN) design-matrix entries of its own point, normalised on the fly, in registers; a fill is
     * ltab_pts consecutive lanes writing theirs to the table, from which every accumulating lane reads its two (F) or four (H) factors per
     * point: same factors, same multiplies, same adds in list order. */
    {
        const double s1 = A1[0], t1x = A1[1], t1y = A1[2], s2 = A2[0], t2x = A2[1], t2y = A2[2];
        int ie = 0, je = 0;
        { int e = 0; for (int i = 0; i < 9; i++) for (int q = 0; q <= i; q++) { if (e == lane) { ie = i; je = q; } e++; } }
        const int ki = ie / 3, li = ie % 3, kj = je / 3, lj = je % 3;
        const int x0 = !rows2 ? ie : (li == 0 ? ki : li == 1 ? 9 : 3 + ki), y0 = !rows2 ? je : (lj == 0 ? kj : lj == 1 ? 9 : 3 + kj);
        const int x1 = li == 0 ? 9 : li == 1 ? ki : 6 + ki, y1 = lj == 0 ? 9 : lj == 1 ? kj : 6 + kj;
        double val = 0;
        dg_pt qn = stage[lane < len ? lane : 0];
        for (int blk = 0; blk < len; blk += 64) {
            const dg_pt p = qn; dg_pt q;
            if (blk + 64 < len) qn = stage[blk + 64 + lane < len ? blk + 64 + lane : blk];
            q.x1 = p.x1 * s1 + t1x; q.y1 = p.y1 * s1 + t1y; q.x2 = p.x2 * s2 + t2x; q.y2 = p.y2 * s2 + t2y;
            double z[9];
            if (!rows2) {
                const double a[3] = {q.x1, q.y1, 1.0}, b[3] = {q.x2, q.y2, 1.0};
#pragma unroll
                for (int k = 0; k < 3; k++)
#pragma unroll
                    for (int l = 0; l < 3; l++) z[3*k + l] = a[l] * b[k];
            } else {
                const double b[3] = {q.x2, q.y2, 1.0};
#pragma unroll
                for (int k = 0; k < 3; k++) { z[k] = b[k]; z[3 + k] = -q.x1 * b[k]; z[6 + k] = -q.y1 * b[k]; }
            }
            const int bc = len - blk < 64 ? len - blk : 64;
            for (int f0 = 0; f0 < bc; f0 += fpts) {
                const int cnt = bc - f0 < fpts ? bc - f0 : fpts;
                if (lane >= f0 && lane < f0 + cnt) {
                    dg_ldsd *t = lt3 + 10 * (lane - f0);
#pragma unroll
                    for (int k = 0; k < 9; k++) t[k] = z[k];
                    t[9] = 0.0;
                }
                DG_WSYNC_LDS();
                if (lane < 45) {
                    int p = 0;
                    if (!rows2) {
                        for (; p + 8 <= cnt; p += 8) {
                            const dg_ldsd *t = lt3 + 10 * p;
                            const double u0 = t[x0], v0 = t[y0], u1 = t[10 + x0], v1 = t[10 + y0], u2 = t[20 + x0], v2 = t[20 + y0], u3 = t[30 + x0],
                                v3 = t[30 + y0];
                            const double u4 = t[40 + x0], v4 = t[40 + y0], u5 = t[50 + x0], v5 = t[50 + y0], u6 = t[60 + x0], v6 = t[60 + y0], u7 = t[70 + x0],
                                v7 = t[70 + y0];
                            val += u0 * v0; val += u1 * v1; val += u2 * v2; val += u3 * v3; val += u4 * v4; val += u5 * v5; val += u6 * v6; val += u7 * v7;
                        }
                        for (; p < cnt; p++) { const dg_ldsd *t = lt3 + 10 * p; val += t[x0] * t[y0]; }
                    } else {
                        for (; p + 4 <= cnt; p += 4) {
                            const dg_ldsd *t = lt3 + 10 * p;
                            const double u0 = t[x0], v0 = t[y0], p0 = t[x1], q0 = t[y1], u1 = t[10 + x0], v1 = t[10 + y0], p1 = t[10 + x1], q1 = t[10 + y1];
                            const double u2 = t[20 + x0], v2 = t[20 + y0], p2 = t[20 + x1], q2 = t[20 + y1], u3 = t[30 + x0], v3 = t[30 + y0], p3 = t[30 + x1],
                                q3 = t[30 + y1];
                            val += u0 * v0; val += p0 * q0; val += u1 * v1; val += p1 * q1; val += u2 * v2; val += p2 * q2; val += u3 * v3; val += p3 * q3;
                        }
                        for (; p < cnt; p++) { const dg_ldsd *t = lt3 + 10 * p; val += t[x0] * t[y0]; val += t[x1] * t[y1]; }
                    }
                }
                DG_WSYNC_LDS();
            }
        }
        if (lane < 45) { s->V[9*ie + je] = val; s->V[ie + 9*je] = val; }
    }
}

template <class SC, class Sync>
__device__ __forceinline__ void dg_lsq_seq_par(SC *s, dg_pt *stage, int len, int tid, int nthr, int rows2, double *A1o, double *A2o, Sync sync,
                                               double *ltab = (double *)0 /* optional LDS scratch, 10 doubles per point of a fill, used by wave 0 only */,
                                               const int ltab_pts = 64 /* points per fill of ltab (<= 64) */)
{
    if (ltab) {          /* an LDS table: the whole fit by the first wave, every sum fed from LDS (dg_lsq_seq_wave) */
        if (tid < 64) dg_lsq_seq_wave(s, stage, len, tid & 63, rows2, A1o, A2o, ltab, ltab_pts);
        sync();
        return;
    }
    /* [4][len] coordinates, then [2][len] centroid distances; stored and loaded as global memory (not flat) */
    __attribute__((address_space(1))) double *aux = (__attribute__((address_space(1))) double *)(double *)(stage + len);
    const int lane = tid & 63; const bool w0 = tid < 64;
    for (int j = tid; j < len; j += nthr) {                       /* one array per coordinate */
        const dg_pt p = stage[j];
        aux[j] = p.x1; aux[len + j] = p.y1; aux[2*len + j] = p.x2; aux[3*len + j] = p.y2;
    }
    sync();
    if (w0) {                                                     /* centroids: lane l in 0..3 sums coordinate l, in list order */
        double acc = 0;
        if (lane < 4) acc = dg_seq_sum_from<1>((const double *)(aux + (size_t)lane * len), len, 0.0);
        if (len > 0) acc /= len;
        if (lane < 4) s->D[lane] = acc;
    }
    sync();
    const double m1x = s->D[0], m1y = s->D[1], m2x = s->D[2], m2y = s->D[3];
    for (int j = tid; j < len; j += nthr) {                       /* distances to the centroids, one point per thread */
        const dg_pt p = stage[j];
        double a = p.x1 - m1x, b = p.y1 - m1y; aux[j] = sqrt(a*a + b*b);
        a = p.x2 - m2x; b = p.y2 - m2y; aux[len + j] = sqrt(a*a + b*b);
    }
    sync();
    if (w0) {                                                     /* mean distances: lane 0 image 1, lane 1 image 2 */
        double dsum = 0;
        if (lane < 2) dsum = dg_seq_sum_from<1>((const double *)(aux + (size_t)lane * len), len, 0.0);
        double A1[3], A2[3];
        A1[0] = __shfl(dsum, 0, 64); A2[0] = __shfl(dsum, 1, 64);
        if (A1[0] != 0) A1[0] = len * sqrt(2.0) / A1[0];
        if (A2[0] != 0) A2[0] = len * sqrt(2.0) / A2[0];
        A1[1] = m1x * -A1[0]; A1[2] = m1y * -A1[0];
        A2[1] = m2x * -A2[0]; A2[2] = m2y * -A2[0];
        if (lane == 0) { for (int i = 0; i < 3; i++) { A1o[i] = A1[i]; A2o[i] = A2[i]; } }
    }
    sync();
    {
        const double s1 = A1o[0], t1x = A1o[1], t1y = A1o[2], s2 = A2o[0], t2x = A2o[1], t2y = A2o[2];
        for (int j = tid; j < len; j += nthr) {                   /* normalised coordinates in place */
            const dg_pt p = stage[j]; dg_pt q;
            q.x1 = p.x1 * s1 + t1x; q.y1 = p.y1 * s1 + t1y; q.x2 = p.x2 * s2 + t2x; q.y2 = p.y2 * s2 + t2y;
            stage[j] = q;
        }
    }
    sync();
    if (w0 && lane < 45) {                                        /* normal matrix: lane e owns entry (ie, je), je <= ie */
        int ie = 0, je = 0;
        { int e = 0; for (int i = 0; i < 9; i++) for (int q = 0; q <= i; q++) { if (e == lane) { ie = i; je = q; } e++; } }
        const int ki = ie / 3, li = ie % 3, kj = je / 3, lj = je % 3;
        double val = 0;
#define DG_NM_TERM(p) do { \
            const double a0 = (p).x1, a1 = (p).y1, b0 = (p).x2, b1 = (p).y2; \
            if (!rows2) { \
                double ai = li == 0 ? a0 : li == 1 ? a1 : 1.0, bi = ki == 0 ? b0 : ki == 1 ? b1 : 1.0; \
                double aj = lj == 0 ? a0 : lj == 1 ? a1 : 1.0, bj = kj == 0 ? b0 : kj == 1 ? b1 : 1.0; \
                val += (ai * bi) * (aj * bj); \
            } else { \
                double bqi = ki == 0 ? b0 : ki == 1 ? b1 : 1.0, bqj = kj == 0 ? b0 : kj == 1 ? b1 : 1.0; \
                double z0i = li == 0 ? bqi : li == 1 ? 0.0 : -a0 * bqi, z0j = lj == 0 ? bqj : lj == 1 ? 0.0 : -a0 * bqj; \
                double z1i = li == 0 ? 0.0 : li == 1 ? bqi : -a1 * bqi, z1j = lj == 0 ? 0.0 : lj == 1 ? bqj : -a1 * bqj; \
                val += z0i * z0j; \
                val += z1i * z1j; \
            } } while (0)
        int j = 0;
        for (; j + 8 <= len; j += 8) {
            const dg_pt p0 = stage[j], p1 = stage[j+1], p2 = stage[j+2], p3 = stage[j+3];
            const dg_pt p4 = stage[j+4], p5 = stage[j+5], p6 = stage[j+6], p7 = stage[j+7];
            DG_NM_TERM(p0); DG_NM_TERM(p1); DG_NM_TERM(p2); DG_NM_TERM(p3);
            DG_NM_TERM(p4); DG_NM_TERM(p5); DG_NM_TERM(p6); DG_NM_TERM(p7);
        }
        for (; j < len; j++) { const dg_pt p = stage[j]; DG_NM_TERM(p); }
#undef DG_NM_TERM
        s->V[9*ie + je] = val; s->V[ie + 9*je] = val;
    }
}

template <class PtFn>
__device__ __forceinline__ void dg_lsq_seq(dg_lsq_scratch *s, PtFn pt, const int *list, int len, int tid, int rows2, double *A1o, double *A2o, dg_pt *stage,
    int stage_cap,
                                           double *ltab = (double *)0)
{
    /* gather the listed correspondences into a contiguous staging array (all lanes), so that the sequential
     * sums stream uniform addresses */
    __syncthreads();
    for (int j = tid; j < len; j += DG_T) stage[j] = pt(list[j]);
    __syncthreads();
    if (ltab || stage_cap >= 2 * len) dg_lsq_seq_par(s, stage, len, tid, DG_T, rows2, A1o, A2o, [] { __syncthreads(); }, ltab);
    else if (tid < 64) dg_lsq_seq_core(s, stage, len, tid, rows2, A1o, A2o);
    __syncthreads();
}

template <class PtFn>
__device__ __forceinline__ void dg_u2f_big(dg_red *r, dg_lsq_scratch *s, PtFn pt, const int *list, int len, int tid, double *Fout /* LDS */, dg_pt *stage,
    int stage_cap,
                                           double *ltab = (double *)0)
{
    (void)r;
    dg_lsq_seq(s, pt, list, len, tid, 0, s->A1, s->A2, stage, stage_cap, ltab);
    if (tid < 64) dg_eig_sym_wave(s->V, s->D, tid, &s->ews);
    if (tid == 0) {
        int jm = 0; for (int i = 1; i < 9; i++) if (s->D[i] < s->D[jm]) jm = i;
        for (int i = 0; i < 9; i++) Fout[i] = s->V[jm*9 + i];
        dg_singulF(Fout);
        dg_denormF(Fout, s->A1, s->A2);
    }
    __syncthreads();
}

#endif /* DG_LSQ_H */
