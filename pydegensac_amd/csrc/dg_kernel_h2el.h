/* RANSAC on ellipse-to-ellipse (local affine frame) correspondences: the reference's ransacH2el (degensac/ranH2el.c:19-206,
 * declared in ranH2el.h:35).  Minimal sample = 2 correspondences; the model comes from the 14 x 15 system
 * of Chum & Matas (ICPR 2012) through the reference's Gauss-Jordan null space (utools.c:97-167); scoring is HDs with two
 * thresholds (th and th * TAU), the local optimisation is ranH.c's inHrani / iterH on the point coordinates.
 *
 * The driver needs a few dozen to a few hundred samples (the sample size is 2), so there is nothing to speculate on: one
 * workgroup runs the reference's loop as it stands — thread 0 draws, wave 0 solves (the 15 x 15 elimination in LDS, one
 * element group per lane), the workgroup scores with the passes of the homography kernel (dg_h_pass), and the least squares
 * are the homography kernel's (dg_u2h_list).  What each errs[] buffer holds is tracked as "the model whose residuals it
 * contains" (S->bufF), like in the other drivers.
 *
 * Input: u10 [n, 10] = x1 y1 a1 b1 c1 | x2 y2 a2 b2 c2 per correspondence (frame = [a 0; b c]); the model maps image 2 to
 * image 1 (ranH2el.c:38-45 builds u6 = x1 y1 1 x2 y2 1 for HDs / u2h).  Included by the 512-thread translation unit only. */
#ifndef DG_KERNEL_H2EL_H
#define DG_KERNEL_H2EL_H
#include "dg_kernel_h.h"

#define DG_H2_TAU (18.0*18.0/7.0/7.0)            /* ranH2el.h:31 */

/* utools.c:97-167 nullspace(): Gauss-Jordan null space of an n x n row-major matrix (n <= 16), tol 1e-12, by ONE WAVE on a
 * matrix in LDS.  Lane e handles the elements e, e + 64, ... of the matrix; within one pivot step the updates of different
 * elements are independent of each other (each reads its row's entry of the pivot column, the pivot row and itself), so every
 * element sees the reference's sequence of operations.  buffer: 2n ints (LDS).  All 64 lanes call it; returns the number of
 * null vectors (rows of `nullspace`, n doubles each). */
__device__ __noinline__ int dg_nullspace_wave(double *matrix, double *nullspace, int n, int *buffer, int lane)
{
    int nonpivot = 0, npiv = 0, i = 0;
    const double tol = 1e-12;
    for (int j = 0; j < n; j++) {
        DG_WSYNC();
        /* pivot search, the reference's order: first row of the largest magnitude from the diagonal down */
        const double mine = lane < n ? fabs(matrix[n*lane + j]) : 0.0;
        double pivot = dg_readlane_d(mine, i); int max = i;
        for (int k = i + 1; k < n; k++) { const double t = dg_readlane_d(mine, k); if (pivot < t) { pivot = t; max = k; } }
        if (pivot < tol) {
            if (lane == 0) buffer[nonpivot] = j;
            nonpivot++;
            if (lane >= i && lane < n) matrix[n*lane + j] = 0;
        } else {
            if (lane == 0) buffer[n + npiv] = j;
            npiv++;
            /* swap rows i <-> max (columns j..n-1), then divide the pivot row by the pivot element */
            if (lane >= j && lane < n) {
                const double a = matrix[i*n + lane], b = matrix[max*n + lane];
                matrix[i*n + lane] = b; matrix[max*n + lane] = a;
            }
            DG_WSYNC();
            const double pv = matrix[i*n + j];
            DG_WSYNC();
            if (lane >= j && lane < n) matrix[i*n + lane] /= pv;
            DG_WSYNC();
            /* subtract multiples of the pivot row from all the other rows: reads first, then the stores */
            double f[4], r[4], v[4]; bool act[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int e = lane + 64 * u, k = e / n, l = e - k * n;
                act[u] = e < n * n && k != i && l >= j;
                f[u] = act[u] ? matrix[k*n + j] : 0.0; r[u] = act[u] ? matrix[i*n + l] : 0.0; v[u] = act[u] ? matrix[e] : 0.0;
            }
            DG_WSYNC();
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int e = lane + 64 * u, k = e / n;
                if (act[u]) {
                    if (k < i) { const double pm = -f[u]; matrix[e] = v[u] + pm * r[u]; }
                    else matrix[e] = v[u] - f[u] * r[u];
                }
            }
            i++;
        }
    }
    DG_WSYNC();
    if (lane == 0) {
        for (int k = 0; k < nonpivot; k++) {
            const int j = buffer[k];
            for (int l = 0; l < n - nonpivot; l++) nullspace[k*n + buffer[n + l]] = -matrix[l*n + j];
            for (int l = 0; l < nonpivot; l++) nullspace[k*n + buffer[l]] = (j == buffer[l]) ? 1 : 0;
        }
    }
    DG_WSYNC();
    return nonpivot;
}

/* ranH2el.c:209-230 getTransf (column-wise 3 x 3) */
__device__ __forceinline__ void dg_h2_transf(const double *u10, double *N, double *D)
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
/* ranH2el.c:342-360 Zu and :382-399 Znd for two correspondences: Z is 14 x 15, column-major with leading dimension 14 */
__device__ __forceinline__ void dg_h2_Zu(double *Z, const double *u)
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
__device__ __forceinline__ void dg_h2_Znd(double *Z, const double *A, const double *B)
{
    const int ld = 14;
    /* ranH2el.h:9-27: A and B are read transposed */
    const double a1 = A[0], a2 = A[3], a3 = A[6], a4 = A[1], a5 = A[4], a6 = A[7];
    const double b1 = B[0], b2 = B[3], b3 = B[6], b4 = B[1], b5 = B[4], b6 = B[7];
    Z[2 + 2*ld] = a3; Z[5 + 2*ld] = a6; Z[6 + 2*ld] = 1;
    Z[0 + 0*ld] = a2*b1 - a1*b4; Z[1 + 0*ld] = a2*b2 - a1*b5; Z[2 + 0*ld] = a2*b3 - a1*b6;
    Z[3 + 0*ld] = a5*b1 - a4*b4; Z[4 + 0*ld] = a5*b2 - a4*b5; Z[5 + 0*ld] = a5*b3 - a4*b6;
    Z[0 + 1*ld] = a1*b1 + a2*b4; Z[1 + 1*ld] = a1*b2 + a2*b5; Z[2 + 1*ld] = a1*b3 + a2*b6;
    Z[3 + 1*ld] = a4*b1 + a5*b4; Z[4 + 1*ld] = a4*b2 + a5*b5; Z[5 + 1*ld] = a4*b3 + a5*b6;
}
/* ranH2el.c:232-283 A2toRH without normalisation (do_norm = 0 there), by one wave (all 64 lanes call it).  scr: 3 * 225 doubles +
 * 30 ints of LDS.  Returns 1 when the null space is not one-dimensional (sample rejected); h (LDS) is written either way. */
__device__ __noinline__ int dg_h2_A2toRH(const double *ua, const double *ub, double *scr, double *h, int lane)
{
    double *Z = scr, *ZT = scr + 225, *U = scr + 450; int *nb = (int *)(scr + 675);
    for (int i = lane; i < 225; i += 64) { Z[i] = 0.0; U[i] = 0.0; }
    DG_WSYNC();
    if (lane == 0) {
        double N1[9], D1[9], N2[9], D2[9];
        dg_h2_transf(ua, N1, D1);
        dg_h2_transf(ub, N2, D2);
        dg_h2_Zu(Z, ua);
        dg_h2_Zu(Z + 7, ub);
        dg_h2_Znd(Z + 2*7*9, D1, N1);
        dg_h2_Znd(Z + 2*7*9 + 2*7*3 + 7, D2, N2);
    }
    DG_WSYNC();
    /* mattr(ZT, Z, 15, 14), last row zero */
    for (int e = lane; e < 225; e += 64) { const int i = e / 15, j = e - 15 * i; ZT[e] = i < 14 ? Z[j*14 + i] : 0.0; }
    DG_WSYNC();
    const int nullsize = dg_nullspace_wave(ZT, U, 15, nb, lane);
    if (lane == 0) {
        for (int i = 0; i < 9; i++) h[i] = U[i];
        double t = h[1]; h[1] = h[3]; h[3] = t; t = h[2]; h[2] = h[6]; h[6] = t; t = h[5]; h[5] = h[7]; h[7] = t;   /* trnm(h, 3) */
    }
    DG_WSYNC();
    return nullsize != 1;
}

/* errs[] bookkeeping of the driver: pe[i] = physical buffer behind errs[i] (i = 0..4), S->bufF[b] = the model whose
 * residuals buffer b holds */
struct dg_h2bufs { int pe[5]; unsigned wr; /* bit b: buffer b has been written (a model stands behind it) */ };
#define DG_H2SET(S, b, src) do { __syncthreads(); if (tid < 9) (S)->bufF[(b)][tid] = (src)[tid]; B.wr |= 1u << (b); __syncthreads(); } while (0)

/* ranH.c:18-86 iterH.  h (LDS) = in/out parameter H; errs[4]'s model is bufF[B.pe[4]] */
template <int LDSPTS>
__device__ __forceinline__ dg_score dg_h2_iterH(CTX &c, int *inliers, double th, double ths, double *h, unsigned inlLimit, dg_h2bufs &B)
{
    dg_f_shared *S = c.S; const int n = c.n, tid = c.tid;
    double *hl = S->fLO;
    dg_score zero = {0, 0, 0, 0}, maxS = zero;
    const double dth = (ths - th) / DG_ILSQ_ITERS;
    int pd = B.pe[1];
    dg_pass_cfg c0 = dg_cfg0(n); c0.wantJ = 1; c0.thJ = th; c0.list = inliers; c0.thL = th;
    dg_pass_res r0 = dg_h_pass(c, S->bufF[B.pe[4]], c0);
    maxS.I = r0.I; maxS.J = r0.J;
    if (maxS.I < 4) return zero;
    {
        int cnt = (int)maxS.I, o = 0, use = cnt;
        __syncthreads();
        if ((unsigned)cnt > inlLimit) { if (tid == 0) dg_randsubset(&S->rng, inliers, cnt, (int)inlLimit); use = (int)inlLimit; o = cnt - use; }
        __syncthreads();
        dg_u2h_list(c, inliers + o, use, hl);
    }
    for (int it = 0; it < DG_ILSQ_ITERS; it++) {
        dg_pass_cfg c1 = dg_cfg0(n); c1.wantJ = 1; c1.thJ = th; c1.list = inliers; c1.thL = ths;
        dg_pass_res r1 = dg_h_pass(c, hl, c1); c.n_hds++;
        DG_H2SET(S, pd, hl);
        if (maxS.J < r1.J) {
            maxS = zero; maxS.I = r1.I; maxS.J = r1.J;
            { const int t = B.pe[0]; B.pe[1] = t; B.pe[0] = pd; pd = B.pe[1]; }
            __syncthreads();
            if (tid < 9) h[tid] = hl[tid];
            __syncthreads();
        }
        if (r1.nL < 4) return maxS;
        {
            int cnt = (int)r1.nL, o = 0, use = cnt;
            __syncthreads();
            if ((unsigned)cnt > inlLimit) { if (tid == 0) dg_randsubset(&S->rng, inliers, cnt, (int)inlLimit); use = (int)inlLimit; o = cnt - use; }
            __syncthreads();
            dg_u2h_list(c, inliers + o, use, hl);
        }
        ths -= dth;
    }
    dg_pass_cfg c3 = dg_cfg0(n); c3.wantJ = 1; c3.thJ = th;
    dg_pass_res r3 = dg_h_pass(c, hl, c3); c.n_hds++;
    DG_H2SET(S, pd, hl);
    if (maxS.J < r3.J) {
        maxS = zero; maxS.I = r3.I; maxS.J = r3.J;
        B.pe[1] = B.pe[0]; B.pe[0] = pd;
        __syncthreads();
        if (tid < 9) h[tid] = hl[tid];
        __syncthreads();
    }
    return maxS;
}

/* ranH2el.c:493-543 inHraniEl = ranH.c:88-135 inHrani (its minimal-sample branch needs ssiz < 4, which loLimit = 8 excludes).
 * inliers = L[0] (ninl entries), Hout (LDS) = in/out parameter H */
template <int LDSPTS>
__device__ __forceinline__ dg_score dg_h2_inHrani(CTX &c, int ninl, double th, double *Hout, unsigned inlLimit, dg_h2bufs &B)
{
    dg_f_shared *S = c.S; const int tid = c.tid;
    int *inliers = c.K->L[0], *intbuff = c.K->L[1];
    dg_score maxS = {0, 0, 0, 0};
    if (ninl < 8) return maxS;
    int ssiz = ninl / 2; if (ssiz > 12) ssiz = 12;
    { const int t = B.pe[2]; B.pe[2] = B.pe[0]; B.pe[0] = t; }
    for (int i = 0; i < DG_RAN_REP; i++) {
        __syncthreads();
        if (tid < 64) {
            if (tid == 0) { const int o = dg_randsubset(&S->rng, inliers, ninl, ssiz); dg_gather(c, inliers + o, ssiz, S->lsq.px); }
            DG_WSYNC();
            dg_u2h_small_w(&S->lsq, S->lsq.px, ssiz, S->f, tid);
        }
        __syncthreads();
        DG_H2SET(S, B.pe[0], S->f); c.n_hds++;               /* HDs(h) -> errs[0] */
        B.pe[4] = B.pe[0];                                    /* errs[4] = errs[0] */
        dg_score Sc = dg_h2_iterH(c, intbuff, th, DG_TC * th, S->f, inlLimit, B);
        if (maxS.J < Sc.J) {
            maxS = Sc;
            { const int t = B.pe[2]; B.pe[2] = B.pe[0]; B.pe[0] = t; }
            __syncthreads();
            if (tid < 9) Hout[tid] = S->f[tid];
            __syncthreads();
        }
    }
    { const int t = B.pe[2]; B.pe[2] = B.pe[0]; B.pe[0] = t; }
    return maxS;
}

/* the local-optimisation block of the driver (ranH2el.c:128-151 and :165-187): h = S->Hx (the driver's `h`, in/out).
 * Returns 1 when it set a new maximum. */
template <int LDSPTS>
__device__ __noinline__ int dg_h2_lo(CTX &c, double th, unsigned inlLimit, dg_score &maxS, dg_h2bufs &B, const bool e4_written = true)
{
    dg_f_shared *S = c.S; const int n = c.n, tid = c.tid;
    const int pd = B.pe[0];                                                       /* d = errs[0] */
    dg_pass_cfg ca = dg_cfg0(n); ca.list = c.K->L[0]; ca.thL = DG_TC * th * DG_H2_TAU;
    dg_pass_res ra;
    if (e4_written) ra = dg_h_pass(c, S->bufF[B.pe[4]], ca);
    else {
        /* errs[4] never written (no sample beat the running best before the run after the loop, ranH2el.c:163-171): the reference
         * reads its uninitialised allocation; as the oracle, a zero-filled buffer = every point within the threshold */
        for (int j = tid; j < n; j += DG_T) c.K->L[0][j] = j;
        __syncthreads();
        ra = dg_pass_res(); ra.nL = (unsigned)n;
    }
    if (ra.nL >= 4) dg_u2h_list(c, c.K->L[0], (int)ra.nL, S->Hx);                 /* u2h leaves h alone below 4 ids (Htools.c:106) */
    DG_H2SET(S, pd, S->Hx); c.n_hds++;                                            /* HDs(h) -> d */
    dg_pass_cfg cb = dg_cfg0(n); cb.list = c.K->L[0]; cb.thL = th;
    dg_pass_res rb = dg_h_pass(c, S->Hx, cb);
    dg_score Sl = dg_h2_inHrani(c, (int)rb.nL, th, S->Hx, inlLimit, B);
    __syncthreads();
    double tol = S->Hx[8]; tol = tol*tol*tol;
    if (maxS.J < Sl.J && fabs(dg_det3(S->Hx) / tol) > 10e-2) {
        maxS = Sl;
        { const int t = B.pe[0]; B.pe[0] = B.pe[3]; B.pe[3] = t; }
        __syncthreads();
        if (tid < 9) S->F[tid] = S->Hx[tid];
        __syncthreads();
        return 1;
    }
    return 0;
}

template <int LDSPTS>
__device__ __forceinline__ void dg_h2_pair(const dg_args &A, dg_f_shared *S, const int pair, const int slot)
{
    const int tid = threadIdx.x;
    const long long off = A.offsets[pair];
    const int n = (int)(A.offsets[pair + 1] - off);
    const dg_params &pr = A.prm;
    const double th = pr.th;
    const long long t_start = wall_clock64();
    char *ws = A.ws + (size_t)slot * A.wl.stride;
    CTX c;
    c.S = S; c.K = (const __attribute__((address_space(3))) dg_f_cshared *)&S->K; c.n = n; c.tid = tid; c.A = &A; c.off = off;
    __syncthreads();
    if (tid == 0) dg_fill_views(&S->K, ws, A.wl);
    __syncthreads();
    c.ht.heads = (int *)(ws + A.wl.off_ht); c.ht.count = c.ht.heads + 64; c.ht.ent = c.ht.heads + 80;
    c.seeds = S->seeds3[0]; c.draws = S->draws3[0];
    c.n_fds = c.n_exfds = c.n_hds = c.n_aux = 0; c.rrun = 0; c.hlt = (double *)0;
    c.cb = (dg_coop_cb *)0; c.coop_gen = (int *)0; c.coop_slot = 0;
    dg_pt *Pw = (dg_pt *)(ws + A.wl.off_pts); int *pool = (int *)(ws + A.wl.off_pool);
    c.P = Pw; c.pool = pool;
    const double *u10 = A.pts1 + (size_t)off * 10;
    for (int i = tid; i < n; i += DG_T) {
        const double *a = u10 + (size_t)i * 10;
        dg_pt p; p.x1 = a[0]; p.y1 = a[1]; p.x2 = a[5]; p.y2 = a[6];
        Pw[i] = p; pool[i] = i;
    }
    if (tid < 9) { S->F[tid] = 0; S->Hx[tid] = 0; }
    if (tid < 36) S->bufF[tid / 9][tid % 9] = 0;
    __syncthreads();
    /* 3 * 225 doubles + 30 ints of LDS for the elimination (the least-squares scratch is idle then) */
    double *scr = (double *)&S->lsq;
    static_assert(sizeof(dg_lsq_scratch) >= 3 * 225 * sizeof(double) + 30 * sizeof(int), "elimination scratch does not fit");
    const unsigned inlLimit = pr.h2_inl_limit == 0 ? 0x7fffffffu : (unsigned)pr.h2_inl_limit;
    const int do_lo = pr.h2_do_lo;
    dg_score maxS = {0, 0, 0, 0}, maxSs = {0, 0, 0, 0};
    dg_h2bufs B; B.pe[0] = 0; B.pe[1] = 1; B.pe[2] = 2; B.pe[3] = 3; B.pe[4] = 3; B.wr = 0;
    int no_sam = 0, max_sam = pr.max_iters, iter_cnt = 0, best_sample = 0; long long t_best = t_start;
    if (tid < 64) { dg_srand_wave(&S->rng, A.seeds[pair], tid); const int v_ = dg_rand_block(&S->rng, 1, tid); if (tid == 0) S->itmp[31] = v_; }
    __syncthreads();
    unsigned seed = (unsigned)S->itmp[31];
    while (no_sam < max_sam) {
        no_sam++;
        int new_max = 0, do_iterate = 0;
        __syncthreads();
        if (tid < 64) {
            /* srand(seed), the two draws of randsubset and the next seed: one wave step each (dg_srand_wave, dg_rand_block) */
            dg_srand_wave(&S->rng, seed, tid);
            const int d3_ = dg_rand_block(&S->rng, 3, tid);
            const int dr0 = __builtin_amdgcn_readlane(d3_, 0), dr1 = __builtin_amdgcn_readlane(d3_, 1), dr2 = __builtin_amdgcn_readlane(d3_, 2);
            if (tid == 0) {
                /* rtools.c:25-39 randsubset(pool, n, 2) on the two draws */
                { const int s = dr0 % n, j = n - 1; const int q = pool[s]; pool[s] = pool[j]; pool[j] = q; }
                { const int s = dr1 % (n - 1), j = n - 2; const int q = pool[s]; pool[s] = pool[j]; pool[j] = q; }
                S->itmp[31] = dr2;
                S->itmp[28] = pool[n - 2]; S->itmp[29] = pool[n - 1];
            }
            DG_WSYNC();
            /* the driver's `h` is S->Hx: A2toRH overwrites it even when it rejects the sample */
            const int rej = dg_h2_A2toRH(u10 + (size_t)S->itmp[28] * 10, u10 + (size_t)S->itmp[29] * 10, scr, S->Hx, tid);
            if (tid == 0) S->itmp[30] = rej;
        }
        __syncthreads();
        seed = (unsigned)S->itmp[31];
        if (S->itmp[30]) continue;
        {
            const double v = dg_det3(S->Hx); double tol = S->Hx[8]; tol = tol*tol*tol;
            if (fabs(v / tol) < 10e-2) continue;
        }
        const int pd = B.pe[0];                                                   /* d = errs[0] */
        DG_H2SET(S, pd, S->Hx); c.n_hds++;
        dg_pass_cfg c1 = dg_cfg0(n); c1.wantJ = 1; c1.thJ = th;
        const dg_pass_res r1 = dg_h_pass(c, S->Hx, c1);
        if (maxS.J < r1.J) {
            maxS.I = r1.I; maxS.J = r1.J;
            B.pe[0] = B.pe[3]; B.pe[3] = pd;
            __syncthreads();
            if (tid < 9) S->F[tid] = S->Hx[tid];
            __syncthreads();
            new_max = 1; best_sample = no_sam; t_best = wall_clock64();
        }
        dg_pass_cfg c2 = dg_cfg0(n); c2.wantJ = 1; c2.thJ = th * DG_H2_TAU;
        const dg_pass_res r2 = dg_h_pass(c, S->Hx, c2);
        if (maxSs.J < r2.J) {
            maxSs.I = r2.I; maxSs.J = r2.J;
            do_iterate = no_sam > DG_ITER_SAM;
            if (!new_max) { B.pe[0] = B.pe[2]; B.pe[2] = pd; }
            B.pe[4] = pd;
        }
        if (no_sam >= DG_ITER_SAM && iter_cnt == 0 && maxSs.I > 4) do_iterate = 1;
        if (do_iterate && do_lo) {
            iter_cnt++;
            if (dg_h2_lo(c, th, inlLimit, maxS, B)) { new_max = 1; best_sample = no_sam; t_best = wall_clock64(); }
        }
        if (new_max) {
            const int new_sam = dg_nsamples((int)maxS.I + 1, n, 2, pr.conf);
            if (new_sam < max_sam) max_sam = new_sam;
        }
    }
    if (do_lo && !iter_cnt) {
        iter_cnt++;
        if (dg_h2_lo(c, th, inlLimit, maxS, B, maxSs.J > 0)) { best_sample = no_sam; t_best = wall_clock64(); }
    }
    /* inl[j] = errs[3][j] <= th */
    __syncthreads();
    {
        double H[9];
#pragma unroll
        for (int i = 0; i < 9; i++) H[i] = S->bufF[B.pe[3]][i];
        unsigned char *mask = A.mask_out + off;
        if ((B.wr >> B.pe[3]) & 1u)
            for (int j = tid; j < n; j += DG_T) { const dg_pt p = Pw[j]; mask[j] = dg_HDs(H, p.x1, p.y1, p.x2, p.y2) <= th ? 1 : 0; }
        else {
            /* errs[3] was never written: no sample's model and no local optimisation ever became the best (budgets of a few samples,
             * thresholds nothing meets).  The reference reads the buffer as its malloc left it (ranH2el.c:189-198); oracle and device take the
             * zero-filled buffer of a fresh allocation, as for errs[4]: every residual 0, so inl[j] = (0 <= th).  The model stays
             * zero; pydegensac_amd.ransacH2el turns "zero H" into an all-false mask for its callers (raw=True returns this one). */
            const unsigned char v0 = 0.0 <= th ? 1 : 0;
            for (int j = tid; j < n; j += DG_T) mask[j] = v0;
        }
    }
    if (tid < 9) A.model_out[(size_t)pair * 9 + tid] = S->F[tid];
    if (A.stats_out && tid == 0) {
        int *st = A.stats_out + (size_t)pair * 16;
        const long long t_end = wall_clock64();
        st[0] = no_sam; st[1] = iter_cnt; st[2] = 0; st[3] = (int)maxS.I; st[4] = c.n_hds;
        st[5] = 0; st[6] = 0; st[7] = best_sample; st[8] = c.n_hds; st[9] = 0; st[10] = 0; st[11] = 0;
        st[12] = (int)(t_best - t_start); st[13] = (int)(t_end - t_start); st[14] = A.variant_threads; st[15] = A.mode;
    }
    __syncthreads();
}

__global__ __launch_bounds__(DG_T, DG_MINW) void dg_ransac_h2el_kernel(dg_args A)
{
    __shared__ dg_f_shared Sh;
    __shared__ int next_pair;
    __shared__ dg_args As;
    if (threadIdx.x == 0) As = A;
    __syncthreads();
    for (;;) {
        const int pair = dg_next_pair(As, &next_pair);
        if (pair < 0) break;
        dg_h2_pair<0>(As, &Sh, pair, (int)blockIdx.x);
    }
}

#endif /* DG_KERNEL_H2EL_H */
