/* Two symmetric 9 x 9 eigen-problems side by side in ONE wave: problem 0 in lanes 0..31, problem 1 in lanes 32..63.
 *
 * The same arithmetic as dg_eig_sym_wave (dg_dev_small.h) per problem — this file is that function with three changes:
 *   - a lane's number is its number inside its half, and every pointer (matrix, d / e / tau) is the half's own;
 *   - a broadcast from lane l (v_readlane with a compile-time lane) becomes two broadcasts, from lanes l and 32 + l, and a select;
 *   - dsteqr (dg_steqr9.h, included a second time under another name) takes its decisions per half: the three searches for a negligible
 *     subdiagonal entry ballot over the half's own eight lanes, the QL / QR choice is a per-lane condition.  Where the two problems take
 *     different paths (QL against QR, another number of sweeps) the wave executes both, each under its half's execution mask.
 * Used where two independent small fits wait for the same wave: checksample's fifth triplet at four waves, innerFH's fifteen
 * repetitions (dg_kernel_f.h).  Unit entry: mi_degensac_mat3 op 5 (tests/test_gpu_units.py: bit for bit against the one-problem solver
 * and the oracle's dsyev). */
#ifndef DG_EIG2_H
#define DG_EIG2_H

static __device__ __forceinline__ double dg_rdl2_d(double v, int l, bool up)
{
    const double lo = dg_rdl_d(v, l), hi = dg_rdl_d(v, 32 + l);
    return up ? hi : lo;
}
#define DG_RDL2(v, l) dg_rdl2_d((v), (l), uph)

/* dg_steqr9.h once more, as dg_steqr9_x2: decisions per half-wave */
#undef DG_STEQR9_H
#undef DG_STEQR_ANY
#undef DG_STEQR_FIND_SPLIT
#undef DG_STEQR_FIND_QL
#undef DG_STEQR_FIND_QR
#undef DG_STEQR_T
#define DG_STEQR_T(i)
#define DG_STEQR_ANY(cond) (cond)
#define DG_STEQR_HALF_BALLOT(c_) ((__ballot(c_) >> (lane & 32)) & 0xffull)
#define DG_STEQR_FIND_SPLIT(d, e, l1, m) do { const int i_ = lane & 7; const double ae_ = fabs((e)[i_]); \
        const bool c_ = i_ >= (l1) && (ae_ == 0. || ae_ <= (sqrt(fabs((d)[i_])) * sqrt(fabs((d)[i_+1]))) * DG_EPS); \
        const unsigned long long bm_ = DG_STEQR_HALF_BALLOT(c_); (m) = bm_ ? __ffsll((long long)bm_) - 1 : 8; } while (0)
#define DG_STEQR_FIND_QL(d, e, l, lend, m) do { const int i_ = lane & 7; double t2_ = fabs((e)[i_]); t2_ *= t2_; \
        const bool c_ = i_ >= (l) && i_ < (lend) && t2_ <= ((DG_EPS*DG_EPS) * fabs((d)[i_])) * fabs((d)[i_+1]) + DG_SAFMIN; \
        const unsigned long long bm_ = DG_STEQR_HALF_BALLOT(c_); (m) = bm_ ? __ffsll((long long)bm_) - 1 : (lend); } while (0)
#define DG_STEQR_FIND_QR(d, e, l, lend, m) do { const int i_ = lane & 7; double t2_ = fabs((e)[i_]); t2_ *= t2_; \
        const bool c_ = i_ >= (lend) && i_ < (l) && t2_ <= ((DG_EPS*DG_EPS) * fabs((d)[i_+1])) * fabs((d)[i_]) + DG_SAFMIN; \
        const unsigned long long bm_ = DG_STEQR_HALF_BALLOT(c_); (m) = bm_ ? 64 - __clzll((long long)bm_) : (lend); } while (0)
#define dg_steqr9 dg_steqr9_x2
#include "dg_steqr9.h"
#undef dg_steqr9

static __device__ __noinline__ int dg_eig_sym_wave2(double *a0, double *w0, dg_eig_ws *ews0, double *a1, double *w1, dg_eig_ws *ews1, const int lane_w)
{
    const int n = 9;
    const bool uph = lane_w >= 32;                   /* this lane belongs to the second problem */
    const int lane = lane_w & 31;                    /* its number inside its half */
    double *a = uph ? a1 : a0, *w = uph ? w1 : w0; dg_eig_ws *ews = uph ? ews1 : ews0;
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
            double alpha = DG_RDL2(R[c1], ih), xnorm = 0., taui, beta, sc = 0.;
            { const double sq = R[c1] * R[c1];
#pragma unroll
              for (int kk = 0; kk < ih; kk++) xnorm += DG_RDL2(sq, kk); }
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
                  for (int jj = 0; jj <= ih; jj++) sum += R[jj] * DG_RDL2(R[c1], jj);
                  if (lane <= ih) tl = taui * sum; }
                double dot = 0.;
                { const double pr = tl * R[c1];
#pragma unroll
                  for (int kk = 0; kk <= ih; kk++) dot += DG_RDL2(pr, kk); }
                const double al = -.5 * taui * dot;
                if (lane <= ih) tl += al * R[c1];
                const double vl = R[c1];
#pragma unroll
                for (int cc = 0; cc <= ih; cc++) {
                    const double vc = DG_RDL2(vl, cc), tc = DG_RDL2(tl, cc);
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
            for (int kk = 0; kk <= iq; kk++) wv[kk] = DG_RDL2(Cq[kk], iq);
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
    /* ---- dsteqr 'V' ----
     * dg_steqr9.h: d / e stay in LDS (every lane runs the scalar recurrence and stores the same values), lane r < 9
     * rotates row r of Z in place (a[c*9 + r]); lanes 9..63 repeat rows 0..8, same values to the same addresses. */
    {
        double p;
        DG_WSYNC();
        const int info = dg_steqr9_x2((DG_STEQR_PTR)d, (DG_STEQR_PTR)e, (DG_STEQR_PTR)(a + (lane % 9)), 9, lane_w);
        const int jtot = info ? n * 30 : 0, nmaxit = n * 30;
        DG_WSYNC();
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
            return jtot >= nmaxit ? 1 : 0;
    }
#undef A_
}

#undef DG_RDL2
#endif /* DG_EIG2_H */
