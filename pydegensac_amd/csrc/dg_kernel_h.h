/* Persistent homography kernel: one workgroup runs the whole reference driver exp_ransacHcustomLAF
 * (degensac/exp_ranH.c:470-930, iter_type 4, oriented constraint on, unlimited LSQ) for one pair.
 * Same speculate-then-commit structure as the F kernel (dg_kernel_f_main.h); differences:
 *   - 4 draws per sample (next seed = 5th output), orientation / rank / near-singularity rejects
 *     happen in the per-lane solve, at most one model per sample;
 *   - five selectable error metrics (Htools.c) incl. the symmetric ones that need H^-1 per model;
 *   - LO re-fits on ALL band inliers (inlLimit = 1e6): normalised DLT over long id lists;
 *   - the ten repetitions of a local optimisation run one per WAVE and are replayed in repetition order
 *     (dg_inHranic_waves below); the workgroup-wide serial order (dg_inHranic) remains for the residual dump.
 */
#ifndef DG_KERNEL_H_H
#define DG_KERNEL_H_H
#include "dg_kernel_f_main.h"
#define DG_SW0 (DG_NW > 2 ? 2 : 0)          /* first scoring wave of the main loop */
/* development build only: time since the previous mark goes to dbg[i] (0 passes, 1 least squares + eigen-solve over a long
 * list, 2 inlier-set hash, 3 small fits, 4 consistency checks); dbg[7] holds the previous mark */
#define DG_HT(i) DG_DEVT(do { __syncthreads(); if (c.tid == 0) { long long t_ = DG_CLK(); c.S->dbg[i] += t_ - c.S->dbg[7]; c.S->dbg[7] = t_; } } while (0))

/* u2h (Htools.c:115-131) over a list of ids, reference summation order (see dg_lsq_seq) */
template <class PtFn>
__device__ __forceinline__ void dg_u2h_big(dg_red *r, dg_lsq_scratch *s, PtFn pt, const int *list, int len, int tid, double *Hout /* LDS */, dg_pt *stage,
    int stage_cap,
                                           double *ltab)
{
    (void)r;
    dg_lsq_seq(s, pt, list, len, tid, 1, s->A1, s->A2, stage, stage_cap, ltab);
    if (tid < 64) dg_eig_sym_wave(s->V, s->D, tid, &s->ews);
    if (tid == 0) {
        for (int i = 0; i < 9; i++) Hout[i] = s->V[i];
        dg_denormH(Hout, s->A1, s->A2);
    }
    __syncthreads();
}

/* u2h over a global id list of any length (lane 0 for <= 12 points, cooperative otherwise) */
template <int LDSPTS>
__device__ __forceinline__ void dg_u2h_list(CTX &c, const int *list, int len, double *Hout)
{
    dg_f_shared *S = c.S;
    if (len <= 12) {
        __syncthreads();
        if (c.tid < 64) { if (c.tid == 0) dg_gather(c, list, len, S->lsq.px); DG_WSYNC(); dg_u2h_small_w(&S->lsq, S->lsq.px, len, Hout, c.tid); }
        __syncthreads();
    } else {
        const dg_pt *P = c.P;
        dg_u2h_big(&S->red, &S->lsq, [&](int i) { return dg_ldpt<LDSPTS>(P, i); }, list, len, c.tid, Hout, c.K->stage, 2 * c.K->n_max,
                   DG_LSQ_LTAB(S));
    }
}

/* a full pass of the selected H metric; counts as one HDS1 call when `count` */
template <int LDSPTS>
__device__ __forceinline__ dg_pass_res dg_hm_pass(CTX &c, int kind, const double *Hm /* LDS */, dg_pass_cfg cfg)
{
    dg_f_shared *S = c.S;
    double H[9], Hinv[9], H1[9];
    if (kind != 0) {
        __syncthreads();
        if (c.tid == 0) dg_hsym_prepare(Hm, S->lsq.Z8, S->lsq.Z8 + 9);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 9; i++) { H[i] = Hm[i]; Hinv[i] = kind ? S->lsq.Z8[i] : 0; H1[i] = kind ? S->lsq.Z8[9+i] : 0; }
    const dg_pt *P = c.P;
    /* ordered MSAC terms: the per-wave solver scratch is idle during a workgroup pass (LDS); what does not fit goes to the HBM staging area */
    cfg.jbuf = (double *)c.K->stage; cfg.jl = (double *)c.S->ww; cfg.jl_cap = (int)(DG_JBUF_LDS_BYTES / sizeof(double));
    return dg_pass(&S->red, cfg, [&](int pid, int) { return dg_Herr(kind, H, Hinv, H1, dg_ldpt<LDSPTS>(P, pid)); }, c.tid);
}

/* symmetric (HDsSymMaxidx, eps variant) and LAF (HDSi1 on u_1, u_2) consistency of model h over the ids
 * list[0..cnt): exp_ranH.c:587-618, :706-735, :828-853.  p1_inliers is the driver's never-reset counter
 * (exp_ranH.c:501).  early_p1: the main loop bails out after the first LAF set when p1 < maxS.Ilafs. */
template <int LDSPTS>
__device__ __forceinline__ int dg_h_checks(CTX &c, int kind, const double *h /* LDS */, const int *list, int cnt, dg_score &S,
                                           const dg_score &maxS, int *p1_inliers, int early_p1)
{
    const dg_params &pr = c.A->prm;
    dg_f_shared *Sh = c.S;
    double H[9], Hinv[9], H1[9];
    __syncthreads();
    if (c.tid == 0) dg_hsym_prepare(h, Sh->lsq.Z8, Sh->lsq.Z8 + 9);
    __syncthreads();
    for (int i = 0; i < 9; i++) { H[i] = h[i]; Hinv[i] = Sh->lsq.Z8[i]; H1[i] = Sh->lsq.Z8[9+i]; }
    const dg_pt *P = c.P;
    if (pr.sym_th > 0) {
        dg_pass_cfg cfg = dg_cfg0(cnt); cfg.src = list; cfg.wantC = 1; cfg.thC = pr.sym_th;
        dg_pass_res r = dg_pass(&Sh->red, cfg, [&](int pid, int) { dg_pt p = dg_ldpt<LDSPTS>(P, pid); return dg_Hsym(Hinv, H1, p.x1, p.y1, p.x2, p.y2, 2, 1); },
            c.tid);
        S.Is = r.C;
        if (S.Is < maxS.Is) return 0;
    }
    if (pr.laf_coef > 0) {
        double thl = pr.laf_coef * pr.th;
        dg_pass_cfg cfg = dg_cfg0(cnt); cfg.src = list; cfg.wantC = 1; cfg.thC = thl;
        dg_pass_res r1 = dg_pass(&Sh->red, cfg, [&](int pid, int) {
            dg_pt o = dg_ldpt<LDSPTS>(P, pid), l = c.laf_pt(pid, 1);
            return kind == 0 ? dg_HDs_mixed(H, o.x1, o.y1, o.x2, o.y2, l.x1, l.y1, l.x2, l.y2) : dg_Hsym(Hinv, H1, l.x1, l.y1, l.x2, l.y2, kind, 1); }, c.tid);
        *p1_inliers += (int)r1.C;
        if (early_p1 && *p1_inliers < (int)maxS.Ilafs) return 0;
        dg_pass_res r2 = dg_pass(&Sh->red, cfg, [&](int pid, int) {
            dg_pt o = dg_ldpt<LDSPTS>(P, pid), l = c.laf_pt(pid, 2);
            return kind == 0 ? dg_HDs_mixed(H, o.x1, o.y1, o.x2, o.y2, l.x1, l.y1, l.x2, l.y2) : dg_Hsym(Hinv, H1, l.x1, l.y1, l.x2, l.y2, kind, 1); }, c.tid);
        S.Ilafs = (int)r2.C < *p1_inliers ? r2.C : (unsigned)*p1_inliers;
        if (S.Ilafs < maxS.Ilafs) return 0;
    }
    return 1;
}

/* physical residual-buffer bookkeeping of the LO: pe[k] = physical id behind errs[k], k=0..2;
 * bufM[b] = model whose residuals physical buffer b holds */
struct dg_hbufs { int pe[3]; };
#define DG_BUFSET(S, b, src) do { __syncthreads(); if (tid < 9) (S)->bufF[(b)][tid] = (src)[tid]; __syncthreads(); } while (0)

/* exp_ranH.c:291-412 exp_iterHcustom; h (LDS) = in/out parameter H */
template <int LDSPTS>
__device__ __forceinline__ dg_score dg_iterHc(CTX &c, int kind, int *inliers, double th, double ths, double *h, int iterID, unsigned inlLimit, dg_hbufs &B,
                                              int rrow /* first diagnostics row of this repetition's iterations */)
{
    dg_f_shared *S = c.S; const int n = c.n, tid = c.tid;
    double *hl = S->fLO;
    dg_score zero = {0, 0, 0, 0}, maxS = zero;
    double dth = (ths - th) / DG_ILSQ_ITERS;
    int pd = B.pe[1];                                         /* d = errs[1] */
    /* errs[4] = errs[0] holds HDS1(h): inlidxs(th) and the list at th*MWM in one pass */
    dg_pass_cfg c0 = dg_cfg0(n); c0.wantJ = 1; c0.thJ = th; c0.list = inliers; c0.thL = th * DG_MWM;
    dg_pass_res r0 = dg_hm_pass(c, kind, h, c0); c.n_hds++;
    DG_HT(0);
    maxS.I = r0.I; maxS.J = r0.J;
    DG_TRACE(c, 20, maxS.I, maxS.J);
    if (maxS.I < 4) {
        dg_pass_cfg c1 = dg_cfg0(n); c1.list = inliers; c1.thL = th;
        dg_hm_pass(c, kind, h, c1);
        return zero;
    }
    {
        int cnt = (int)r0.nL, o = 0, use = cnt;
        unsigned dc = (unsigned)cnt; if (dc > inlLimit) dc = inlLimit; if (dc < 4) dc = 4;
        __syncthreads();
        if (dc < (unsigned)cnt) { if (tid == 0) dg_randsubset(&S->rng, inliers, cnt, (int)dc); use = (int)dc; o = cnt - (int)dc; }
        __syncthreads();
        dg_u2h_list(c, inliers + o, use, hl);
        DG_HT(1);
    }
    for (int it = 0; it < DG_ILSQ_ITERS; it++) {
        dg_pass_cfg c1 = dg_cfg0(n); c1.wantJ = 1; c1.thJ = th; c1.list = inliers; c1.thL = th;
        dg_pass_res r1 = dg_hm_pass(c, kind, hl, c1); c.n_hds++;
        DG_HT(0);
        dg_dump_resid(c, rrow + it, hl, 10 + kind);                       /* exp_ranH.c:344 */
        DG_BUFSET(S, pd, hl);
        dg_score Ss = zero; Ss.I = r1.I; Ss.J = r1.J;
        DG_TRACE(c, 21, Ss.I, Ss.J);
        __syncthreads();
        if (tid < 64) {
            unsigned hash = dg_hash_list(inliers, (int)Ss.I, n < 65536);
            if (tid == 0) {
                int ret = dg_ht_contains(c.ht, hash, (int)Ss.I, iterID);
                if (ret == -1) dg_ht_insert(c.ht, hash, (int)Ss.I, iterID);
                S->itmp[0] = (ret != -1 && ret != iterID) ? 1 : 0;
            }
        }
        __syncthreads();
        DG_HT(2);
        if (S->itmp[0]) { DG_TRACE(c, 23, 0, 0); return zero; }
        dg_pass_cfg c2 = dg_cfg0(n); c2.list = inliers; c2.thL = ths * DG_MWM;
        dg_pass_res r2 = dg_hm_pass(c, kind, hl, c2);
        DG_HT(0);
        DG_TRACE(c, 24, r2.nL, 0);
        if (maxS.J < Ss.J) {
            maxS = Ss;
            { int t = B.pe[0]; B.pe[1] = t; B.pe[0] = pd; pd = B.pe[1]; }
            __syncthreads();
            if (tid < 9) h[tid] = hl[tid];
            __syncthreads();
        }
        if (r2.nL < 4) return maxS;
        {
            int cnt = (int)r2.nL, o = 0, use = cnt;
            unsigned dc = (unsigned)cnt; if (dc > inlLimit) dc = inlLimit; if (dc < 4) dc = 4;
            __syncthreads();
            if (dc < (unsigned)cnt) { if (tid == 0) dg_randsubset(&S->rng, inliers, cnt, (int)dc); use = (int)dc; o = cnt - (int)dc; }
            __syncthreads();
            dg_u2h_list(c, inliers + o, use, hl);
            DG_HT(1);
        }
        ths -= dth;
    }
    dg_pass_cfg c3 = dg_cfg0(n); c3.wantJ = 1; c3.thJ = th; c3.list = inliers; c3.thL = th;
    dg_pass_res r3 = dg_hm_pass(c, kind, hl, c3); c.n_hds++;
    DG_HT(0);
    dg_dump_resid(c, rrow + 4, hl, 10 + kind);                            /* exp_ranH.c:400 */
    DG_BUFSET(S, pd, hl);
    DG_TRACE(c, 22, r3.I, r3.J);
    if (maxS.J < r3.J) {
        maxS = zero; maxS.I = r3.I; maxS.J = r3.J;
        B.pe[1] = B.pe[0]; B.pe[0] = pd;
        __syncthreads();
        if (tid < 9) h[tid] = hl[tid];
        __syncthreads();
    }
    return maxS;
}

/* exp_ranH.c:415-467 exp_inHranicustom; inliers = L[0]; result model -> Hout (LDS) */
template <int LDSPTS>
__device__ __forceinline__ dg_score dg_inHranic(CTX &c, int kind, int ninl, double th, double *Hout, int *iterID, unsigned inlLimit, dg_hbufs &B)
{
    dg_f_shared *S = c.S; const int tid = c.tid;
    int *inliers = c.K->L[0], *intbuff = c.K->L[1];
    dg_score maxS = {0, 0, 0, 0};
    if (ninl < 8) {
        if (c.rrun) {                                                    /* exp_ranH.c:429: memset(.., 0xFF, ..) */
            const double ff = __longlong_as_double(-1ll);
            for (size_t j = tid; j < (size_t)(DG_RESIDS_M - 2) * c.n; j += DG_T) c.rrun[2 * (size_t)c.n + j] = ff;
            __syncthreads();
        }
        return maxS;
    }
    int ssiz = ninl / 2; if (ssiz > 12) ssiz = 12;
    { int t = B.pe[2]; B.pe[2] = B.pe[0]; B.pe[0] = t; }
    for (int i = 0; i < DG_RAN_REP; i++) {
        __syncthreads();
        if (tid < 64) {
            if (tid == 0) { int o = dg_randsubset(&S->rng, inliers, ninl, ssiz); dg_gather(c, inliers + o, ssiz, S->lsq.px); }
            DG_WSYNC();
            dg_u2h_small_w(&S->lsq, S->lsq.px, ssiz, S->f, tid);
        }
        __syncthreads();
        DG_HT(3);
        DG_BUFSET(S, B.pe[0], S->f);                          /* HDS1(h) -> errs[0] (scored inside dg_iterHc) */
        dg_dump_resid(c, 2 + 6 * i, S->f, 10 + kind);         /* exp_ranH.c:445-446 */
        ++*iterID;
        dg_score Sc = dg_iterHc(c, kind, intbuff, th, DG_TC * th, S->f, *iterID, inlLimit, B, 2 + 6 * i + 1);
        if (maxS.J < Sc.J) {
            maxS = Sc;
            { int t = B.pe[2]; B.pe[2] = B.pe[0]; B.pe[0] = t; }
            __syncthreads();
            if (tid < 9) Hout[tid] = S->f[tid];
            __syncthreads();
        }
    }
    { int t = B.pe[2]; B.pe[2] = B.pe[0]; B.pe[0] = t; }
    return maxS;
}

/* ---- the repetitions of exp_inHranicustom, one wave each (exp_ranH.c:415-467, :291-412) ---------------------------------
 * A repetition = a 12-point fit, its score, then up to four rounds of (score the model, hash its inlier set, refit on the
 * ids under the shrinking threshold) and a last score.  On one workgroup that is a string of workgroup-wide passes
 * separated by stretches where one wave runs a reference-order sum or the 9 x 9 eigen-solve and the others wait.  But the
 * repetitions barely depend on each other:
 *   - the samples: randsubset permutes `inliers` in place and nothing else draws (inlLimit is never reached), so all ten
 *     samples can be drawn up front;
 *   - the inlier-set hash table (a repetition that meets a set an EARLIER repetition inserted stops and counts as empty),
 *     the best-so-far comparison and the rotation of the errs[] buffers: all three are decided by a handful of numbers
 *     per repetition.
 * So every wave runs whole repetitions on its own (its own id lists, MSAC terms and staging area in the workspace, its own
 * solver scratch in LDS, wave barriers only) and records those numbers (dg_hrep_log); thread 0 then replays the hash
 * table, the comparisons and the buffer rotation in repetition order.  Same arithmetic per repetition, same decisions in
 * the same order; a repetition the reference would have cut short is simply computed further than needed. */
#define DG_HLT 640                /* doubles of LDS per wave: the long-list fit's table (dg_lsq_seq_par), the 12-point fit's design matrix */
/* This kernel's static LDS is dg_f_shared + the argument block + DG_HLT doubles per wave.  At 256 threads that is 48 bytes more than half
 * a CU's LDS, so there the LAST wave's table lives in members of dg_f_shared that only the fundamental-matrix kernel uses (wpad, the
 * checksample tables and the union behind them are contiguous): two workgroups per CU. */
#define DG_HLT_STATIC_WAVES (DG_T == 256 ? DG_NW - 1 : DG_NW)
static_assert(DG_T != 256 || offsetof(dg_f_shared, n_ahead) - offsetof(dg_f_shared, wpad) >= DG_HLT * sizeof(double), "no room for the last wave's table");
template <class C_>
__device__ __forceinline__ double *dg_hlt_wave(C_ &c, int wave)
{
    if (DG_T == 256 && wave == DG_NW - 1) return c.S->wpad;
    return c.hlt + (size_t)DG_HLT * wave;
}
struct dg_hrep_sc { double *Z, *V, *D, *A1, *A2; dg_eig_ws &ews; };     /* scratch view with the member names the solvers use */
/* behind the ten records: the published inlier sets of a local optimisation's repetitions (dg_hpub_*): per repetition eight 8-byte words */
#define DG_HPUB_OFF   (((size_t)DG_RAN_REP * sizeof(dg_hrep_log) + 127) & ~(size_t)127)
#define DG_HPUB_BYTES ((size_t)DG_RAN_REP * 8 * sizeof(unsigned long long))
#define DG_HREP_HDR_BYTES ((DG_HPUB_OFF + DG_HPUB_BYTES + 255) & ~(size_t)255)
__device__ __forceinline__ size_t dg_hrep_logs_bytes() { return DG_HREP_HDR_BYTES; }
__device__ __forceinline__ size_t dg_hrep_wave_bytes(int n_max) { return ((size_t)n_max * (2 * sizeof(int) + sizeof(double) + 2 * sizeof(dg_pt)) + 255) &
    ~(size_t)255;
    }

#define DG_AS1(T) __attribute__((address_space(1))) T
#define DG_AS3(T) __attribute__((address_space(3))) T
/* acc += t[STRIDE * k], k = 0..cnt-1, in that order (LDS): eight loads, then their eight adds */
template <int STRIDE>
__device__ __forceinline__ double dg_chain_lds(const DG_AS3(double) *t, int cnt, double acc)
{
    int k = 0;
    for (; k + 8 <= cnt; k += 8) {
        const double v0 = t[STRIDE*k], v1 = t[STRIDE*(k+1)], v2 = t[STRIDE*(k+2)], v3 = t[STRIDE*(k+3)];
        const double v4 = t[STRIDE*(k+4)], v5 = t[STRIDE*(k+5)], v6 = t[STRIDE*(k+6)], v7 = t[STRIDE*(k+7)];
        acc += v0; acc += v1; acc += v2; acc += v3; acc += v4; acc += v5; acc += v6; acc += v7;
    }
    for (; k < cnt; k++) acc += t[STRIDE*k];
    return acc;
}

/* The sums of dg_lsq_seq_par (normu, utools.c:7-51, and cov_mat, utools.c:170-184, in the reference's order per output
 * number) by ONE wave without a memory round trip per point: the list is taken 256 ids at a time, the next 256 being
 * loaded while the current ones are worked on; each group of points goes through this wave's LDS block and the lanes
 * that own an output number add its terms from there in list order.  Three sweeps: gather + centroids (the gathered
 * points are kept contiguously in `stage` for the other two), distances to the centroids, Hartley-normalised
 * design-matrix entries + the 45 normal-matrix sums.
 * ROWS2: 0 = lin_fmN (one row per point), 1 = lin_hgN (two).  V (9 x 9, both triangles), A1o, A2o: LDS. */
__device__ __forceinline__ void dg_stage_ld4(const DG_AS1(double) *stage, int base, int len, int lane, dg_pt *q)
{
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int j = base + 64 * u + lane; q[u].x1 = q[u].y1 = q[u].x2 = q[u].y2 = 0.0;
        if (j < len) { const DG_AS1(double) *o = stage + 4 * (size_t)j; q[u].x1 = o[0]; q[u].y1 = o[1]; q[u].x2 = o[2]; q[u].y2 = o[3]; }
    }
}
template <int LDSPTS, int ROWS2>
__device__ __forceinline__ void dg_lsq_wave_stream(const dg_pt *P, const int *list_, int len, dg_pt *stage_, double *lt_, double *V, double *A1o, double *A2o,
    int lane)
{
    const DG_AS1(int) *list = (const DG_AS1(int) *)list_;
    DG_AS1(double) *stage = (DG_AS1(double) *)(double *)stage_;
    DG_AS3(double) *t = (DG_AS3(double) *)lt_;
    double acc = 0;
    {
        int idn[4]; dg_pt qn[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int j = 64 * u + lane; idn[u] = j < len ? list[j] : -1; }
#pragma unroll
        for (int u = 0; u < 4; u++) { qn[u].x1 = qn[u].y1 = qn[u].x2 = qn[u].y2 = 0.0; if (idn[u] >= 0) qn[u] = dg_ldpt<LDSPTS>(P, idn[u]); }
#pragma unroll
        for (int u = 0; u < 4; u++) { const int j = 256 + 64 * u + lane; idn[u] = j < len ? list[j] : -1; }
        for (int base = 0; base < len; base += 4 * 64) {
            dg_pt q[4];
#pragma unroll
            for (int u = 0; u < 4; u++) q[u] = qn[u];
            if (base + 256 < len) {
#pragma unroll
                for (int u = 0; u < 4; u++) { qn[u].x1 = qn[u].y1 = qn[u].x2 = qn[u].y2 = 0.0; if (idn[u] >= 0) qn[u] = dg_ldpt<LDSPTS>(P, idn[u]); }
#pragma unroll
                for (int u = 0; u < 4; u++) { const int j = base + 512 + 64 * u + lane; idn[u] = j < len ? list[j] : -1; }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) if (base + 64 * u + lane < len) { DG_AS1(double) *o = stage + 4 * (size_t)(base + 64 * u + lane); o[0] = q[u].x1;
                o[1] = q[u].y1; o[2] = q[u].x2; o[3] = q[u].y2; }
#pragma unroll
            for (int f = 0; f < 2; f++) {            /* 128 points per fill, one array per coordinate */
                const int cnt = len - base - 128 * f < 128 ? len - base - 128 * f : 128;
                if (cnt <= 0) break;
                t[lane] = q[2*f].x1; t[128 + lane] = q[2*f].y1; t[256 + lane] = q[2*f].x2; t[384 + lane] = q[2*f].y2;
                t[64 + lane] = q[2*f+1].x1; t[192 + lane] = q[2*f+1].y1; t[320 + lane] = q[2*f+1].x2; t[448 + lane] = q[2*f+1].y2;
                DG_WSYNC_LDS();
                if (lane < 4) acc = dg_seq_sum_impl<3>(lt_ + 128 * lane, cnt, acc);
                DG_WSYNC_LDS();
            }
        }
    }
    DG_WSYNC();                      /* `stage` (global) is complete: the two sweeps below read it */
    if (len > 0) acc /= len;
    const double m1x = dg_readlane_d(acc, 0), m1y = dg_readlane_d(acc, 1), m2x = dg_readlane_d(acc, 2), m2y = dg_readlane_d(acc, 3);
    double dsum = 0;
    {
        dg_pt qn[4];
        dg_stage_ld4(stage, 0, len, lane, qn);
        for (int base = 0; base < len; base += 4 * 64) {
            dg_pt q[4];
#pragma unroll
            for (int u = 0; u < 4; u++) q[u] = qn[u];
            if (base + 256 < len) dg_stage_ld4(stage, base + 256, len, lane, qn);
            /* 256 points per fill: image 1 distances in t[0..256), image 2 in t[256..512) */
            const int cnt = len - base < 256 ? len - base : 256;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                double a = q[u].x1 - m1x, b = q[u].y1 - m1y; t[64 * u + lane] = sqrt(a*a + b*b);
                a = q[u].x2 - m2x; b = q[u].y2 - m2y; t[256 + 64 * u + lane] = sqrt(a*a + b*b);
            }
            DG_WSYNC_LDS();
            if (lane < 2) dsum = dg_seq_sum_impl<3>(lt_ + 256 * lane, cnt, dsum);
            DG_WSYNC_LDS();
        }
    }
    double A1[3], A2[3];
    A1[0] = dg_readlane_d(dsum, 0); A2[0] = dg_readlane_d(dsum, 1);
    if (A1[0] != 0) A1[0] = len * sqrt(2.0) / A1[0];
    if (A2[0] != 0) A2[0] = len * sqrt(2.0) / A2[0];
    A1[1] = m1x * -A1[0]; A1[2] = m1y * -A1[0];
    A2[1] = m2x * -A2[0]; A2[2] = m2y * -A2[0];
    if (lane == 0) { for (int i = 0; i < 3; i++) { A1o[i] = A1[i]; A2o[i] = A2[i]; } }
    /* normal matrix: lane e < 45 owns entry (ie, je), je <= ie, in cov_mat's enumeration order; per 64-point tile the wave
     * forms, one point per lane, the nine (lin_fmN: z[3k+l] = a_l b_k) or ten (lin_hgN: b_q, -a0 b_q, -a1 b_q and a
     * structural zero) design-matrix entries of each point, and every accumulating lane reads its two (F) or four (H)
     * factors per point from there: same factors, same multiplies, same adds in list order as dg_lsq_seq_par */
    int ie = 0, je = 0;
    { int e = 0; for (int i = 0; i < 9; i++) for (int q_ = 0; q_ <= i; q_++) { if (e == lane) { ie = i; je = q_; } e++; } }
    const int ki = ie / 3, li = ie % 3, kj = je / 3, lj = je % 3;
    const int x0 = !ROWS2 ? ie : (li == 0 ? ki : li == 1 ? 9 : 3 + ki), y0 = !ROWS2 ? je : (lj == 0 ? kj : lj == 1 ? 9 : 3 + kj);
    const int x1 = li == 0 ? 9 : li == 1 ? ki : 6 + ki, y1 = lj == 0 ? 9 : lj == 1 ? kj : 6 + kj;
    double val = 0;
    {
        dg_pt qn[4];
        dg_stage_ld4(stage, 0, len, lane, qn);
        for (int base = 0; base < len; base += 4 * 64) {
            dg_pt q[4];
#pragma unroll
            for (int u = 0; u < 4; u++) q[u] = qn[u];
            if (base + 256 < len) dg_stage_ld4(stage, base + 256, len, lane, qn);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int cnt = len - base - 64 * u < 64 ? len - base - 64 * u : 64;
                if (cnt <= 0) break;
                {
                    const double nx1 = q[u].x1 * A1[0] + A1[1], ny1 = q[u].y1 * A1[0] + A1[2], nx2 = q[u].x2 * A2[0] + A2[1], ny2 = q[u].y2 * A2[0] + A2[2];
                    DG_AS3(double) *w = t + 10 * lane;
                    if (!ROWS2) {
                        const double a[3] = {nx1, ny1, 1.0}, b[3] = {nx2, ny2, 1.0};
#pragma unroll
                        for (int k = 0; k < 3; k++)
#pragma unroll
                            for (int l = 0; l < 3; l++) w[3*k + l] = a[l] * b[k];
                    } else {
                        const double b[3] = {nx2, ny2, 1.0};
#pragma unroll
                        for (int k = 0; k < 3; k++) { w[k] = b[k]; w[3 + k] = -nx1 * b[k]; w[6 + k] = -ny1 * b[k]; }
                    }
                    w[9] = 0.0;
                }
                DG_WSYNC_LDS();
                if (lane < 45) {
                    int p = 0;
                    if (!ROWS2) {
                        for (; p + 8 <= cnt; p += 8) {
                            const DG_AS3(double) *w = t + 10 * p;
                            const double u0 = w[x0], v0 = w[y0], u1 = w[10 + x0], v1 = w[10 + y0], u2 = w[20 + x0], v2 = w[20 + y0], u3 = w[30 + x0],
                                v3 = w[30 + y0];
                            const double u4 = w[40 + x0], v4 = w[40 + y0], u5 = w[50 + x0], v5 = w[50 + y0], u6 = w[60 + x0], v6 = w[60 + y0], u7 = w[70 + x0],
                                v7 = w[70 + y0];
                            val += u0 * v0; val += u1 * v1; val += u2 * v2; val += u3 * v3; val += u4 * v4; val += u5 * v5; val += u6 * v6; val += u7 * v7;
                        }
                        for (; p < cnt; p++) { const DG_AS3(double) *w = t + 10 * p; val += w[x0] * w[y0]; }
                    } else {
                        for (; p + 4 <= cnt; p += 4) {
                            const DG_AS3(double) *w = t + 10 * p;
                            const double u0 = w[x0], v0 = w[y0], p0 = w[x1], q0 = w[y1], u1 = w[10 + x0], v1 = w[10 + y0], p1 = w[10 + x1], q1 = w[10 + y1];
                            const double u2 = w[20 + x0], v2 = w[20 + y0], p2 = w[20 + x1], q2 = w[20 + y1], u3 = w[30 + x0], v3 = w[30 + y0], p3 = w[30 + x1],
                                q3 = w[30 + y1];
                            val += u0 * v0; val += p0 * q0; val += u1 * v1; val += p1 * q1; val += u2 * v2; val += p2 * q2; val += u3 * v3; val += p3 * q3;
                        }
                        for (; p < cnt; p++) { const DG_AS3(double) *w = t + 10 * p; val += w[x0] * w[y0]; val += w[x1] * w[y1]; }
                    }
                }
                DG_WSYNC_LDS();
            }
        }
    }
    if (lane < 45) { V[9*ie + je] = val; V[ie + 9*je] = val; }
}

/* u2h on an id list of any length >= 4, by one wave (Htools.c:101-133) */
template <int LDSPTS>
__device__ __noinline__ void dg_u2h_wave(CTX &c, dg_wave_ws *w, double *lt, const int *list, int len, dg_pt *stage, double *Hout /* LDS */, int lane)
{
    const dg_pt *P = c.P;
    dg_hrep_sc sc = {lt, w->V, w->D, w->A1, w->A2, w->ews};
    DG_WSYNC();
    if (len <= 12) {
        if (lane < len) { const dg_pt q = dg_ldpt<LDSPTS>(P, list[lane]); w->px[4*lane] = q.x1; w->px[4*lane+1] = q.y1; w->px[4*lane+2] = q.x2;
            w->px[4*lane+3] = q.y2; }
        DG_WSYNC();
        if (len == 4) { if (lane == 0) dg_u2h_4pt_mv(lt, lt + 81, w->px, Hout); DG_WSYNC(); }
        else if (len > 4) dg_u2h_norm_w(&sc, w->px, len, Hout, lane);
    } else {
        dg_lsq_wave_stream<LDSPTS, 1>(P, list, len, stage, lt, w->V, w->A1, w->A2, lane);
        DG_WSYNC();
        dg_eig_sym_wave(w->V, w->D, lane, &w->ews);
        if (lane == 0) { for (int i = 0; i < 9; i++) Hout[i] = w->V[i]; dg_denormH(Hout, w->A1, w->A2); }
        DG_WSYNC();
    }
}

/* one wave's pass of model Hm over all n points: I = #(d <= thJ), J = the reference-order MSAC sum, the ordered id list
 * at thL (la, when given) and a second one at thL2 (lb, when given): what dg_pass does for a workgroup.  The nonzero MSAC
 * terms of a step (256 points) go through this wave's LDS block and are added, in point order, before the next step;
 * the next step's points are loaded before that. */
template <int LDSPTS>
__device__ __noinline__ dg_pass_res dg_hm_wpass(const dg_pt *P, int n, int kind, const double *Hm /* LDS */, double *z18 /* LDS */, double thJ,
                                                int *la_, double thL, int *lb_, double thL2, double *lt_, int lane)
{
    double H[9], Hinv[9], H1[9];
    DG_WSYNC();
    if (kind != 0) { if (lane == 0) dg_hsym_prepare(Hm, z18, z18 + 9); DG_WSYNC(); }
#pragma unroll
    for (int i = 0; i < 9; i++) { H[i] = Hm[i]; Hinv[i] = kind ? z18[i] : 0; H1[i] = kind ? z18[9+i] : 0; }
    DG_AS3(double) *t = (DG_AS3(double) *)lt_;
    DG_AS1(int) *la = (DG_AS1(int) *)la_, *lb = (DG_AS1(int) *)lb_;
    dg_pass_res out; out.I = 0; out.J = 0; out.C = 0; out.nL = 0; out.nF = 0; out.nL2 = 0; out.nJ = 0;
    const double t94 = thJ * 9 / 4;
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned cI = 0, nJ = 0, nA = 0, nB = 0;
    double J = 0;
    dg_pt qn[DG_PU];
#pragma unroll
    for (int u = 0; u < DG_PU; u++) { qn[u].x1 = qn[u].y1 = qn[u].x2 = qn[u].y2 = 0.0; const int j = u * 64 + lane; if (j < n) qn[u] = dg_ldpt<LDSPTS>(P, j); }
    for (int base = 0; base < n; base += DG_PU * 64) {
        dg_pt q[DG_PU];
#pragma unroll
        for (int u = 0; u < DG_PU; u++) q[u] = qn[u];
        if (LDSPTS != 1) {
#pragma unroll
            for (int u = 0; u < DG_PU; u++) { const int j = base + DG_PU * 64 + u * 64 + lane; if (j < n) qn[u] = dg_ldpt<LDSPTS>(P, j); }
        }
        unsigned sJ = 0;
#pragma unroll
        for (int u = 0; u < DG_PU; u++) {
            const int j = base + u * 64 + lane; const bool act = j < n;
            const double d = act ? dg_Herr(kind, H, Hinv, H1, q[u]) : 0.0;
            double term = 0.0;
            if (act && thJ != 0 && !(d >= t94)) term = 1 - (d / t94);
            const bool nz = !(term == 0.0);
            cI += (act && d <= thJ) ? 1u : 0u;
            const bool inA = la_ && act && d <= thL, inB = lb_ && act && d <= thL2;
            const unsigned long long mJ = __ballot(nz), mA = __ballot(inA), mB = __ballot(inB);
            if (nz) t[sJ + (unsigned)__popcll(mJ & below)] = term;
            if (inA) la[nA + (unsigned)__popcll(mA & below)] = j;
            if (inB) lb[nB + (unsigned)__popcll(mB & below)] = j;
            sJ += (unsigned)__popcll(mJ); nA += (unsigned)__popcll(mA); nB += (unsigned)__popcll(mB);
        }
        if (LDSPTS == 1) {
#pragma unroll
            for (int u = 0; u < DG_PU; u++) { const int j = base + DG_PU * 64 + u * 64 + lane; if (j < n) qn[u] = dg_ldpt<LDSPTS>(P, j); }
        }
        DG_WSYNC_LDS();                  /* (the terms are in LDS; the id lists, global, are read after the pass) */
        J = dg_seq_sum_impl<3>((const double *)lt_, (int)sJ, J);
        nJ += sJ;
        DG_WSYNC_LDS();
    }
    out.I = dg_wave_sum_u(cI);
    out.J = J;
    out.nL = nA; out.nL2 = nB; out.nJ = nJ;
    DG_WSYNC();
    return out;
}

/* ---- what the repetitions of ONE local optimisation tell each other while they run ------------------------------------
 * In the reference a repetition stops at the first inlier set that an earlier repetition (or an earlier local optimisation)
 * has put into the hash table: with ten repetitions running side by side nobody sees the others' sets, and every one of
 * them runs its four long-list fits where the reference runs 1.2 on average (C3 data: 25.6 passes per local optimisation
 * against 60).  So every repetition PUBLISHES the (hash, I) of each iteration as one 8-byte word in the owner's workspace
 * (one self-contained agent-scope store: no fence, readable from any XCD), and a status word when it ends.  A repetition j
 * may stop at a set X as soon as X is certainly in the reference's table at that point: the sets INSERTED by repetitions
 * k < j.  What a repetition inserts depends on the lower ones, so that is only known for the leading run of FINISHED
 * repetitions 0 .. p - 1 — their words are final, a replay of them in order gives their inserts exactly — and for
 * repetition p itself as far as it has got (everything below it is final).  Repetition 0 is always in that set, and the
 * others converge to the sets it visits.  Stopping early never changes a result: the owner's replay finds the same set in
 * the real table at that iteration (or cuts the repetition earlier still); a repetition that misses a word that is
 * published a moment later just runs one iteration more, as before.
 * word: bit 63 valid, bit 62 "this set was already in the table of the earlier local optimisations" (not an insert; the
 * repetition ended there), bits 32-61 I, bits 0-31 hash; word 7 of a repetition: bit 63 finished, low bits = its iterations */
#define DG_HPUB_VALID (1ull << 63)
#define DG_HPUB_KNOWN (1ull << 62)
__device__ __forceinline__ void dg_hpub_store(unsigned long long *w, unsigned long long v)
{
    __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
/* all 64 lanes of one wave, j >= 1: is the set (hash, I) among the inserts of the repetitions below j that are certain? */
__device__ __forceinline__ bool dg_hpub_seen(unsigned long long *pub, int j, unsigned hash, int I, int lane)
{
    /* lane 6 r + e holds word e of repetition r (e = 0..3: iterations, e = 5: status) */
    const int r = lane / 6, e = lane - 6 * r;
    unsigned long long v = 0;
    /* ONE consistent snapshot per repetition: its status word FIRST, then (behind an agent-scope acquire, which also waits for the
     * status loads) its iteration words.  The writer stores the iteration words, drains its stores and then stores the status word
     * (`finish` in dg_hrep_wave), so a repetition that is seen finished here is seen with all of its iteration words; an unfinished
     * one counts as far as its words go, and any prefix of them is a valid state.  (Loading all six words of a repetition at once,
     * unordered, could pair "finished" with a stale iteration word and undercount that repetition's inserts.) */
    const bool mine_r = r < j && r < DG_RAN_REP;
    if (mine_r && e == 5) v = __hip_atomic_load(pub + 8 * r + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (mine_r && e < DG_ILSQ_ITERS) v = __hip_atomic_load(pub + 8 * r + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    /* p = the leading finished repetitions */
    const unsigned long long finm = __ballot(e == 5 && (v & DG_HPUB_VALID) != 0ull);
    int p = 0; while (p < j && ((finm >> (6 * p + 5)) & 1ull)) p++;
    const int top = p < j ? p : j - 1;                      /* repetitions 0 .. top count */
    const unsigned long long key = v & ~(DG_HPUB_VALID | DG_HPUB_KNOWN);
    bool inserted = false;
    for (int k = 0; k <= top; k++) {
        /* a finished repetition's iteration count is in its status word; the unfinished one (k == p) counts as far as its words go */
        for (int i = 0; i < DG_ILSQ_ITERS; i++) {
            const int src = 6 * k + i;
            const unsigned long long w = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32),
                src) << 32) | (unsigned)__builtin_amdgcn_readlane((int)v, src);
            if (!(w & DG_HPUB_VALID)) break;                                    /* not (yet) reached */
            const unsigned long long kk = w & ~(DG_HPUB_VALID | DG_HPUB_KNOWN);
            if (w & DG_HPUB_KNOWN) break;                                       /* ended at a set of an earlier local optimisation */
            if (__ballot(inserted && r != k && key == kk) != 0ull) break;       /* cut by a lower repetition's set: it inserts nothing from here on */
            /* (a set it visited itself before is in the table already: marking it twice is harmless) */
            if (lane == src) inserted = true;
        }
    }
    const unsigned long long mine = ((unsigned long long)(unsigned)I << 32) | hash;
    return __ballot(inserted && key == mine) != 0ull;
}

/* one repetition (sample lg->ids), by one wave; writes the rest of *lg.  pub: the published sets of this local optimisation
 * (dg_hpub_seen), rep: this repetition's number */
template <int LDSPTS>
__device__ __noinline__ void dg_hrep_wave(CTX &c, int kind, dg_hrep_log *lg, int ssiz, double th, double *lt, char *wb, int lane, int wave,
                                         unsigned long long *pub, int rep)
{
    dg_f_shared *S = c.S; const int n = c.n, nm = c.K->n_max; const dg_pt *P = c.P;
    dg_wave_ws *w = &S->ww[wave];
    int *la = (int *)wb, *lb = la + nm; dg_pt *stage = (dg_pt *)((double *)(lb + nm) + nm);
    double *h = w->H, *hl = w->F, *z18 = w->Z;
#define DG_HW(i) DG_DEVT(do { if (wave == 0 && lane == 0) { long long t_ = DG_CLK(); S->dbg[i] += t_ - tw_; tw_ = t_; } } while (0))
    long long tw_ = DG_CLK(); (void)tw_;
    dg_u2h_wave<LDSPTS>(c, w, lt, lg->ids, ssiz, stage, h, lane);
    DG_HW(3);
    if (lane < 9) lg->h0[lane] = h[lane];
    const dg_pass_res r0 = dg_hm_wpass<LDSPTS>(P, n, kind, h, z18, th, lb, th * DG_MWM, (int *)0, 0.0, lt, lane);
    DG_HW(0);
    if (lane == 0) { lg->I0 = (int)r0.I; lg->J0 = r0.J; lg->nit = 0; lg->last_short = 0; lg->has_fin = 0; }
    /* the status word goes out when the repetition ends, behind its iteration words (the wave's stores are drained first) */
    auto finish = [&](int nit_) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) dg_hpub_store(pub + 8 * rep + 7, DG_HPUB_VALID | (unsigned long long)nit_); };
    /* the real table takes at most DG_HT_CAP entries and drops the rest: near that limit "inserted" is no longer certain */
    const bool share = *c.ht.count + 4 * DG_RAN_REP * DG_ILSQ_ITERS < DG_HT_CAP;
    if (r0.I < 4) { finish(0); return; }
    dg_u2h_wave<LDSPTS>(c, w, lt, lb, (int)r0.nL, stage, hl, lane);
    DG_HW(1);
    double ths = DG_TC * th; const double dth = (ths - th) / DG_ILSQ_ITERS;
    for (int it = 0; it < DG_ILSQ_ITERS; it++) {
        const dg_pass_res r1 = dg_hm_wpass<LDSPTS>(P, n, kind, hl, z18, th, la, th, lb, ths * DG_MWM, lt, lane);
        DG_WSYNC();
        DG_HW(0);
        const unsigned hash = dg_hash_list(la, (int)r1.I, n < 65536);
        DG_HW(2);
        if (lane < 9) lg->it[it].hl[lane] = hl[lane];
        if (lane == 0) { lg->it[it].J = r1.J; lg->it[it].hash = hash; lg->it[it].I = (int)r1.I; lg->nit = it + 1; }
        /* a set some EARLIER local optimisation already inserted ends the repetition here at the latest, whatever the other
         * repetitions of this one do (the table is not written before the replay) */
        { const bool known = dg_ht_known_wave(c.ht, hash, (int)r1.I, lane);
          if (lane == 0) dg_hpub_store(pub + 8 * rep + it, DG_HPUB_VALID | (known ? DG_HPUB_KNOWN : 0ull) | ((unsigned long long)(unsigned)r1.I << 32) | hash);
          if (known) { finish(it + 1); return; } }
        /* ... and so does a set that a repetition below this one has certainly inserted by now */
        if (share && rep > 0 && dg_hpub_seen(pub, rep, hash, (int)r1.I, lane)) { finish(it + 1); return; }
        if (r1.nL2 < 4) { if (lane == 0) lg->last_short = 1; finish(it + 1); return; }
        dg_u2h_wave<LDSPTS>(c, w, lt, lb, (int)r1.nL2, stage, hl, lane);
        DG_HW(1);
        ths -= dth;
    }
    const dg_pass_res rf = dg_hm_wpass<LDSPTS>(P, n, kind, hl, z18, th, (int *)0, 0.0, (int *)0, 0.0, lt, lane);
    DG_HW(0);
    if (lane < 9) lg->hf[lane] = hl[lane];
    if (lane == 0) { lg->If = (int)rf.I; lg->Jf = rf.J; lg->has_fin = 1; }
    finish(DG_ILSQ_ITERS);
}

/* ---- repetitions as claimable jobs (dg_hjob_cb) -------------------------------------------------------------------- */
/* one wave (all 64 lanes, uniform control flow: the retry loop runs on the whole wave behind scalar branches and only the
 * atomic itself sits under `lane == 0` — a compare-and-swap loop under a per-lane branch, inside code with workgroup barriers
 * around it, was seen to make the compiler split the wave around those barriers): claim a repetition of job generation g
 * (cb = null: from the workgroup's own counter in LDS); -1 = none left */
__device__ __forceinline__ int dg_hjob_claim(dg_hjob_cb *cb, int g, int *lds_next, int lane)
{
    if (!cb) {
        int q = 0;
        if (lane == 0) q = atomicAdd(lds_next, 1);
        q = __builtin_amdgcn_readfirstlane(q);
        return q < DG_RAN_REP ? q : -1;
    }
    for (;;) {
        const int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&cb->next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if ((v >> 8) != g || (v & 255) >= DG_RAN_REP) return -1;
        int ok = 0;
        if (lane == 0) { int e = v;
            ok = __hip_atomic_compare_exchange_strong(&cb->next, &e, v + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0; }
        if (__builtin_amdgcn_readfirstlane(ok)) return v & 255;
    }
}
/* one wave: work on job g until no repetition is left to claim */
template <int LDSPTS>
__device__ __forceinline__ void dg_hjob_work(CTX &c, dg_hjob_cb *cb, int g, int *lds_next, int kind, dg_hrep_log *logs, int ssiz, double th, char *wb, int lane,
    int wave)
{
    for (;;) {
        const int r = dg_hjob_claim(cb, g, lds_next, lane);
        if (r < 0) break;
        dg_hrep_wave<LDSPTS>(c, kind, &logs[r], ssiz, th, dg_hlt_wave(c, wave), wb, lane, wave, (unsigned long long *)((char *)logs + DG_HPUB_OFF), r);
        if (cb) {
            /* the record is in the owner's workspace: visible before the count goes up */
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&cb->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

/* exp_inHranicustom with the repetitions as jobs claimed by waves (this workgroup's, and helper workgroups' when the launch has
 * any); ninl >= 8; same results as dg_inHranic */
template <int LDSPTS>
__device__ __forceinline__ dg_score dg_inHranic_waves(CTX &c, int kind, int ninl, double th, double *Hout, int *iterID, dg_hbufs &B)
{
    dg_f_shared *S = c.S; const int tid = c.tid, lane = tid & 63, wave = tid >> 6;
    int *inliers = c.K->L[0];
    dg_score maxS = {0, 0, 0, 0};
    int ssiz = ninl / 2; if (ssiz > 12) ssiz = 12;
    { int t = B.pe[2]; B.pe[2] = B.pe[0]; B.pe[0] = t; }
    dg_hrep_log *logs = (dg_hrep_log *)c.K->hrep;
    char *wb = c.K->hrep + dg_hrep_logs_bytes() + (size_t)wave * dg_hrep_wave_bytes(c.K->n_max);
    /* helpers can only read the points where the owner keeps them in its workspace */
    dg_hjob_cb *cb = (LDSPTS != 1 && c.A->hjob) ? c.A->hjob + c.coop_slot : (dg_hjob_cb *)0;
    __syncthreads();
    if (tid < 64) {
        for (int r = 0; r < DG_RAN_REP; r++) {
            int id = 0;
            dg_randsubset_wave(&S->rng, inliers, ninl, ssiz, lane, &id);
            if (lane < ssiz) logs[r].ids[lane] = id;
            DG_WSYNC();
        }
    }
    if (tid == 0) S->itmp[20] = 0;
    /* no word of the previous local optimisation survives (plain stores: the job's release below, or this barrier for the
     * workgroup's own waves, puts them in front of every reader) */
    if (tid < (int)(DG_HPUB_BYTES / sizeof(unsigned long long))) ((unsigned long long *)((char *)logs + DG_HPUB_OFF))[tid] = 0ull;
    __syncthreads();
    int g = 0;
    if (cb) {
        /* open the job: parameters, then one release (points, samples and the hash table of this pair are plain memory), then the generation */
        g = (c.hjob_gen = (c.hjob_gen % 0x7fffff) + 1);
        if (tid == 0) {
            cb->n = c.n; cb->kind = kind; cb->ssiz = ssiz; cb->wsid = c.coop_slot; cb->th = th;
            __hip_atomic_store(&cb->done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&cb->next, g << 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(wave) == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) { __hip_atomic_store(&cb->gen, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(c.A->done_pairs + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
        __syncthreads();
    }
    dg_hjob_work<LDSPTS>(c, cb, g, &S->itmp[20], kind, logs, ssiz, th, wb, lane, wave);
    __syncthreads();
    if (cb) {
        /* repetitions that helpers claimed may still run */
        if (__builtin_amdgcn_readfirstlane(wave) == 0) {
            dg_wait_count(*c.A, &cb->done, DG_RAN_REP, 2, 4);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (lane == 0) { __hip_atomic_store(&cb->gen, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(c.A->done_pairs + 2, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
        __syncthreads();
    }
    /* thread 0 replays, in repetition order, what the repetitions decide (hash table, best-so-far, buffer rotation) */
    if (tid == 0) {
        int pe0 = B.pe[0], pe1 = B.pe[1], pe2 = B.pe[2], nh = 0, id = *iterID;
        unsigned mI = 0; double mJ = 0;
        for (int r = 0; r < DG_RAN_REP; r++) {
            const dg_hrep_log *g_ = &logs[r];
            for (int i = 0; i < 9; i++) S->bufF[pe0][i] = g_->h0[i];
            ++id; nh++;
            int pd = pe1; unsigned ScI = 0; double ScJ = 0; const double *hres = g_->h0;
            if (g_->I0 >= 4) {
                unsigned mlI = (unsigned)g_->I0; double mlJ = g_->J0; bool dead = false, fin = false;
                for (int it = 0; it < g_->nit; it++) {
                    const dg_hrep_it *q = &g_->it[it];
                    nh++;
                    for (int i = 0; i < 9; i++) S->bufF[pd][i] = q->hl[i];
                    const int ret = dg_ht_contains(c.ht, q->hash, q->I, id);
                    if (ret == -1) dg_ht_insert(c.ht, q->hash, q->I, id);
                    if (ret != -1 && ret != id) { dead = true; break; }
                    if (mlJ < q->J) { mlI = (unsigned)q->I; mlJ = q->J; const int t = pe0; pe1 = t; pe0 = pd; pd = pe1; hres = q->hl; }
                    if (it == g_->nit - 1 && g_->last_short) fin = true;
                }
                if (!dead && !fin && !g_->has_fin) {
                    /* The repetition stopped early because it took a set for certainly inserted (dg_hpub_seen) — and the real table
                     * does not hold it here.  Its final pass was never made (hf / If / Jf are not this repetition's): never consume
                     * them.  Raise the launch's error word: the pair's results are discarded and the host-pointer entry points run
                     * it again (include/mi_degensac.h, MI_ST_PLACEMENT bit 10 / 11). */
                    __hip_atomic_store(c.A->err_flag, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    dead = true;
                }
                if (!dead && !fin) {
                    nh++;
                    for (int i = 0; i < 9; i++) S->bufF[pd][i] = g_->hf[i];
                    if (mlJ < g_->Jf) { mlI = (unsigned)g_->If; mlJ = g_->Jf; pe1 = pe0; pe0 = pd; hres = g_->hf; }
                }
                if (!dead) { ScI = mlI; ScJ = mlJ; }
            }
            if (mJ < ScJ) {
                mI = ScI; mJ = ScJ;
                { const int t = pe2; pe2 = pe0; pe0 = t; }
                for (int i = 0; i < 9; i++) Hout[i] = hres[i];
            }
        }
        { const int t = pe2; pe2 = pe0; pe0 = t; }
        S->red.bi[0] = pe0; S->red.bi[1] = pe1; S->red.bi[2] = pe2; S->red.bi[3] = nh; S->red.bi[4] = (int)mI; S->red.bc[0] = mJ;
    }
    __syncthreads();
    B.pe[0] = S->red.bi[0]; B.pe[1] = S->red.bi[1]; B.pe[2] = S->red.bi[2];
    c.n_hds += S->red.bi[3]; *iterID += DG_RAN_REP;
    maxS.I = (unsigned)S->red.bi[4]; maxS.J = S->red.bc[0];
    __syncthreads();
    return maxS;
}

/* A workgroup that has run out of pairs: the owner slot of an open job, or -1 once every pair of the launch is finished */
__device__ __forceinline__ int dg_hjob_find(const dg_args &A, int *bc /* LDS */)
{
    const int lane = (int)(threadIdx.x & 63);
    for (;;) {
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0) {
            int res = -2;
            if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(A.done_pairs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= A.n_pairs) res = -1;
            else if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(A.done_pairs + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) <= 0) {
                for (int q = 0; q < 4; q++) __builtin_amdgcn_s_sleep(127);
            } else {
                int found = -1;
                /* (uniform trip count: every lane looks at its slots of every round) */
                for (int q = 0; q < A.n_res; q += 64) {
                    const int j = (int)((blockIdx.x + (unsigned)(q + lane)) % (unsigned)A.n_res);
                    if (q + lane < A.n_res && found < 0) {
                        const int g = __hip_atomic_load(&A.hjob[j].gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (g > 0) { const int v = __hip_atomic_load(&A.hjob[j].next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((v >> 8) == g && (v & 255) < DG_RAN_REP) found = j; }
                    }
                }
                const unsigned long long m = __ballot(found >= 0);
                if (m) { res = __builtin_amdgcn_readlane(found, __ffsll((long long)m) - 1); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
                else __builtin_amdgcn_s_sleep(32);
            }
            *bc = res;
        }
        __syncthreads();
        const int r = *bc;
        if (r != -2) return r;
    }
}

/* helper workgroup: repetitions of the job that is open at owner slot `oslot` (this workgroup's own workspace `slot` is free) */
template <int T, int LDSPTS>
__device__ __forceinline__ void dg_h_help(const dg_args &A, dg_f_shared *S, double *hlt, const int oslot, const int slot)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    dg_hjob_cb *cb = A.hjob + oslot;
    char *ws = A.ws + (size_t)slot * A.wl.stride, *wso = A.ws + (size_t)oslot * A.wl.stride;
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(wave) == 0) {
        /* the generation first, then (behind an acquire) the parameters: they are at least as new as that generation, and a claim
         * of that generation only succeeds while they are still its own (the owner opens the next job after the last repetition) */
        int g_ = 0;
        if (tid == 0) { dg_fill_views(&S->K, ws, A.wl); g_ = __hip_atomic_load(&cb->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (tid == 0) { S->itmp[24] = g_; S->itmp[25] = cb->n; S->itmp[26] = cb->kind; S->itmp[27] = cb->ssiz; S->dtmp[31] = cb->th; }
    }
    __syncthreads();
    const int g = S->itmp[24], n = S->itmp[25], kind = S->itmp[26], ssiz = S->itmp[27]; const double th = S->dtmp[31];
    __syncthreads();
    if (g <= 0) return;
    /* never the case for an open job: refuse instead of reading wild memory */
    if (n < 8 || n > A.wl.n_max || ssiz < 4 || ssiz > 12 || kind < 0 || kind > 4) {
        if (tid == 0) __hip_atomic_store(A.err_flag, 16 + (n < 8 || n > A.wl.n_max ? 1 : 0) + (ssiz < 4 || ssiz > 12 ? 2 : 0) + (kind < 0 || kind > 4 ? 4 : 0),
            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    /* (the parameters belong to generation g as long as a claim of generation g succeeds: the owner does not open the next job
     * before every repetition of this one is done) */
    dg_f_ctx<0> c;                  /* the owner's points are read through global loads whatever this kernel's own placement is */
    c.S = S; c.K = (const __attribute__((address_space(3))) dg_f_cshared *)&S->K; c.n = n; c.tid = tid; c.A = &A; c.off = 0;
    c.ht.heads = (int *)(wso + A.wl.off_ht); c.ht.count = c.ht.heads + 64; c.ht.ent = c.ht.heads + 80;     /* the owner's table: lookups only */
    c.seeds = S->seeds3[0]; c.draws = S->draws3[0];
    c.n_fds = c.n_exfds = c.n_hds = c.n_aux = 0; c.rrun = 0; c.hlt = hlt;
    c.cb = (dg_coop_cb *)0; c.coop_gen = (int *)0; c.coop_slot = slot; c.hjob_gen = 0;
    c.P = (const dg_pt *)(wso + A.wl.off_pts); c.pool = (int *)0;
    dg_hrep_log *logs = (dg_hrep_log *)(wso + A.wl.off_hrep);
    char *wb = c.K->hrep + dg_hrep_logs_bytes() + (size_t)wave * dg_hrep_wave_bytes(c.K->n_max);
    dg_hjob_work<0>(c, cb, g, (int *)0, kind, logs, ssiz, th, wb, lane, wave);
    __syncthreads();
}

/* one LO run of the driver (exp_ranH.c:678-747 / :795-861).  e4 = model behind errs[4].  Returns 1 if accepted. */
template <int LDSPTS>
__device__ __noinline__ int dg_h_lo(CTX &c, int kind, const double *e4, double th, dg_score &maxS, int *iterID, int *p1_inliers, int no_sam,
    int lo_run /* 0-based */)
{
    dg_f_shared *S = c.S; const int n = c.n, tid = c.tid;
    dg_hbufs B; B.pe[0] = 0; B.pe[1] = 1; B.pe[2] = 2;
    const int B0 = B.pe[0];                                   /* d = errs[0] */
    dg_resid_begin(c, lo_run); __syncthreads();
    DG_DEVT(if (tid == 0) S->dbg[7] = DG_CLK());
    dg_pass_cfg ca = dg_cfg0(n); ca.list = c.K->L[0]; ca.thL = DG_TC * th * DG_MWM;
    dg_pass_res ra;
    if (e4) {
        dg_dump_resid(c, 0, e4, 10 + kind);                               /* errs[4], exp_ranH.c:679 / :794 */
        ra = dg_hm_pass(c, kind, e4, ca);
    } else {
        /* the run after the loop with errs[4] still where it started (exp_ranH.c:531: errs[4] = errs[3], never written because no
         * sample ever beat the running best -- all rejected, or no sample with an inlier): the reference reads its uninitialised
         * allocation there.  As the oracle: a zero-filled buffer (what a fresh allocation of this size holds), i.e. every point is
         * within the threshold and the least squares runs over ALL points. */
        if (c.rrun) for (int j = tid; j < n; j += DG_T) c.rrun[j] = 0.0;
        for (int j = tid; j < n; j += DG_T) c.K->L[0][j] = j;
        __syncthreads();
        ra = dg_pass_res(); ra.nL = (unsigned)n;
    }
    DG_HT(0);
    DG_TRACE(c, 1, ra.nL, no_sam);
    if (c.hlt && c.K->hrep && (int)ra.nL > 12) {
        /* the long-list fit on ONE wave with the streaming least squares of the per-wave repetitions (every reference-order sum from
         * an LDS tile): the workgroup form spends a barrier round trip per block of the list, and at 128 threads has only two waves
         * to spread its sums over */
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(tid >> 6) == 0) {
            char *wb = c.K->hrep + dg_hrep_logs_bytes();
            const int nm = c.K->n_max;
            dg_pt *stage = (dg_pt *)((double *)((int *)wb + 2 * (size_t)nm) + nm);
            dg_u2h_wave<LDSPTS>(c, &S->ww[0], c.hlt, c.K->L[0], (int)ra.nL, stage, S->f, tid & 63);
        }
        __syncthreads();
    } else dg_u2h_list(c, c.K->L[0], (int)ra.nL, S->f);
    DG_HT(1);
    DG_BUFSET(S, B0, S->f);
    dg_pass_cfg cb = dg_cfg0(n); cb.wantJ = 1; cb.thJ = th; cb.list = c.K->L[0]; cb.thL = th;
    dg_pass_res rb = dg_hm_pass(c, kind, S->f, cb); c.n_hds++;
    DG_HT(0);
    dg_dump_resid(c, 1, S->f, 10 + kind);                                 /* d after the LSQ, exp_ranH.c:694 / :810 */
    DG_TRACE(c, 2, rb.nL, rb.J);
    /* h (the driver's `sol`) = S->Hx: u2h wrote the LSQ model there; inHrani overwrites it on improvement */
    __syncthreads();
    if (tid < 9) S->Hx[tid] = S->f[tid];
    __syncthreads();
    /* one repetition per wave unless a diagnostic needs the reference's order of events (residual dump, trace) */
    const bool waves = c.hlt && c.K->hrep && !c.rrun && !c.A->trace && !c.A->lo_serial && (int)rb.nL >= 8 && n <= 1000000;
    dg_score Sl = waves ? dg_inHranic_waves(c, kind, (int)rb.nL, th, S->Hx, iterID, B) : dg_inHranic(c, kind, (int)rb.nL, th, S->Hx, iterID, 1000000u, B);
    DG_TRACE(c, 3, Sl.I, Sl.J);
    if (!(maxS.J < Sl.J)) return 0;
    if (dg_HcloseToSingular(S->Hx)) return 0;
    const dg_params &pr = c.A->prm;
    if (pr.sym_th > 0 || pr.laf_coef > 0) {
        /* Scheck = inlidxs(d, th): `d` is the physical buffer that was errs[0] before the LO (exp_ranH.c:708) */
        dg_pass_cfg cl = dg_cfg0(n); cl.wantJ = 1; cl.thJ = th; cl.list = c.K->L[2]; cl.thL = th;
        dg_pass_res rl = dg_hm_pass(c, kind, S->bufF[B0], cl);
        DG_TRACE(c, 4, rl.nL, rl.J);
        const int ok_ = dg_h_checks(c, kind, S->Hx, c.K->L[2], (int)rl.nL, Sl, maxS, p1_inliers, 0);
        DG_HT(4);
        if (!ok_) return 0;
    }
    maxS = Sl;
    __syncthreads();
    if (tid < 9) S->F[tid] = S->Hx[tid];
    __syncthreads();
    return 1;
}

/* One 4-point problem per lane (own register allocation): orientation test, 8x9 null vector, near-singularity
 * test and, for the symmetric metrics, H1 = inverse of the transposed H.  Returns 1 when the sample yields a model. */
__device__ __noinline__ int dg_solve4_lane(const dg_pt *P, const int *ids, int kind, double *hm, double *H1m,
    double *wscr /* LDS, this wave's, >= 81 doubles */)
{
    dg_pt sp[4];
#pragma unroll
    for (int i = 0; i < 4; i++) sp[i] = P[ids[i]];
    if (!dg_Hori_valid4(sp)) return 0;
    double m[8][9];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        double s0 = sp[i].x1, s1 = sp[i].y1, s3 = sp[i].x2, s4 = sp[i].y2;
        double z0[9] = {s3, 0, -s0*s3, s4, 0, -s0*s4, 1.0, 0, -s0*1.0};
        double z1[9] = {0, s3, -s1*s3, 0, s4, -s1*s4, 0, 1.0, -s1*1.0};
#pragma unroll
        for (int j = 0; j < 9; j++) { m[2*i][j] = z0[j]; m[2*i+1][j] = z1[j]; }
    }
    int ok = dg_gj8(m, hm);
    /* degenerate samples only: those lanes take turns on the wave's LDS scratch with the general elimination */
    for (unsigned long long need = __ballot(!ok); need; need &= need - 1) {
        if ((int)(threadIdx.x & 63) != __ffsll((long long)need) - 1) continue;
        for (int i = 0; i < 4; i++) {
            const double s0 = sp[i].x1, s1 = sp[i].y1, s3 = sp[i].x2, s4 = sp[i].y2;
            const double z0[9] = {s3, 0, -s0*s3, s4, 0, -s0*s4, 1.0, 0, -s0*1.0};
            const double z1[9] = {0, s3, -s1*s3, 0, s4, -s1*s4, 0, 1.0, -s1*1.0};
            for (int j = 0; j < 9; j++) { wscr[18*i + j] = z0[j]; wscr[18*i + 9 + j] = z1[j]; }
        }
        if (dg_null9<8, 1>(wscr, wscr + 72) == 1) { for (int i = 0; i < 9; i++) hm[i] = wscr[72 + i]; ok = 1; }
    }
    if (!ok || dg_HcloseToSingular(hm)) return 0;
    if (kind != 0) { double Hi[9]; dg_hsym_prepare(hm, Hi, H1m); }
    return 1;
}

/* ---------------------------------------------------------------------------------------------- */
/* Main-loop screen, one wave: the candidate counts (dg_HDs_maybe_below at tb) of the four models g0, g0 + stride, ... of a chunk's model
 * table in ONE sweep over the points, the next step's points in flight (a sweep is bound by the latency of its point loads, not by
 * the 45 flops of a bound).  Own register allocation: four models + two steps of points do not fit next to the driver's state. */
struct dg_u4 { unsigned v[4]; };
template <int LDSPTS>
__device__ __noinline__ dg_u4 dg_h_screen4(const dg_pt *P, int n, const double *gmodels, int g0, int stride, int Mtot, double tb, int lane)
{
    n = __builtin_amdgcn_readfirstlane(n);
    double Hg[4][9];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int mj = g0 + j * stride;
        const double *gs = gmodels + (size_t)(mj < Mtot ? mj : g0) * 18;
#pragma unroll
        for (int q = 0; q < 9; q++) Hg[j][q] = gs[q];
    }
    unsigned cb_[4] = {0u, 0u, 0u, 0u};
    dg_pt qn[DG_PU];
#pragma unroll
    for (int u = 0; u < DG_PU; u++) { const int p = 64 * u + lane; qn[u] = dg_ldpt<LDSPTS>(P, p < n ? p : 0); }
    for (int base = 0; base < n; base += 64 * DG_PU) {
        dg_pt qq[DG_PU];
#pragma unroll
        for (int u = 0; u < DG_PU; u++) qq[u] = qn[u];
#pragma unroll
        for (int u = 0; u < DG_PU; u++) { const int p = base + 64 * DG_PU + 64 * u + lane; if (p < n) qn[u] = dg_ldpt<LDSPTS>(P, p); }
#pragma unroll
        for (int u = 0; u < DG_PU; u++) {
            const bool act = base + 64 * u + lane < n;
#pragma unroll
            for (int j = 0; j < 4; j++) cb_[j] += (act && dg_HDs_maybe_below(Hg[j], qq[u].x1, qq[u].y1, qq[u].x2, qq[u].y2, tb)) ? 1u : 0u;
        }
    }
    dg_u4 out;
#pragma unroll
    for (int j = 0; j < 4; j++) out.v[j] = dg_wave_sum_u(cb_[j]);
    return out;
}

template <int T, int LDSPTS>
__device__ __forceinline__ void dg_h_pair(const dg_args &A, dg_f_shared *S, unsigned char *dyn_smem, double *hlt, const int pair, const int slot, int &hjob_gen)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long off = A.offsets[pair];
    const int n = (int)(A.offsets[pair + 1] - off);
    const dg_params &pr = A.prm;
    const double th = pr.th;
    const int kind = pr.error_type;
    long long t_start = wall_clock64();

    char *ws = A.ws + (size_t)slot * A.wl.stride;
    CTX c;
    c.S = S; c.K = (const __attribute__((address_space(3))) dg_f_cshared *)&S->K; c.n = n; c.tid = tid; c.A = &A; c.off = off;
    __syncthreads();                                       /* nobody still reads the previous pair's views */
    if (tid == 0) dg_fill_views(&S->K, ws, A.wl);
    __syncthreads();
    c.ht.heads = (int *)(ws + A.wl.off_ht); c.ht.count = c.ht.heads + 64; c.ht.ent = c.ht.heads + 80;
    c.seeds = S->seeds3[0]; c.draws = S->draws3[0];
    c.n_fds = c.n_exfds = c.n_hds = c.n_aux = 0; c.rrun = 0; c.hlt = hlt;
    c.cb = (dg_coop_cb *)0; c.coop_gen = (int *)0; c.coop_slot = slot; c.hjob_gen = hjob_gen;
    dg_pt *Pw; int *pool;
    /* LDSPTS: 1 = point set and sampler pool in LDS, 2 = pool in LDS / points in the HBM workspace (L2), 0 = both in HBM */
    if (LDSPTS == 1) { Pw = (dg_pt *)dyn_smem; pool = (int *)(dyn_smem + (size_t)n * sizeof(dg_pt)); }
    else             { Pw = (dg_pt *)(ws + A.wl.off_pts); pool = LDSPTS == 2 ? (int *)dyn_smem : (int *)(ws + A.wl.off_pool); }
    c.P = Pw; c.pool = pool;
    const dg_pt *P = Pw;
    int *const pscr = A.pool_seq ? (int *)0 : (int *)S->ww;     /* LDS scratch of the parallel pool stage; null selects the sequential one */
    for (int i = tid; i < n; i += DG_T) {
        const double *a = A.pts1 + (size_t)(off + i) * A.dim, *b = A.pts2 + (size_t)(off + i) * A.dim;
        dg_pt p; p.x1 = a[0]; p.y1 = a[1]; p.x2 = b[0]; p.y2 = b[1];
        Pw[i] = p; pool[i] = i;
    }
    dg_ht_init(c.ht, tid);
    if (tid < 9) { S->F[tid] = 0; S->FBest[tid] = 0; }
    __syncthreads();

    dg_score maxS = {0, 0, 0, 0}, maxSs = {0, 0, 0, 0};
    int no_sam = 0, max_sam = pr.max_iters, iter_cnt = 0, no_rej = 0, iterID = 0, p1_inliers = 0;
    int best_sample = 0, accepted = 0, done = 0; long long t_best = t_start;
    double *e4 = S->FBest;                                   /* model behind errs[4] (last so-far-best sample) */

    if (__builtin_amdgcn_readfirstlane(wave) == 0) { dg_srand_wave(&S->rng, A.seeds[pair], lane); const int v_ = dg_rand_block(&S->rng, 1, lane);
        if (lane == 0) S->itmp[31] = v_; }
    __syncthreads();
    unsigned seed = (unsigned)S->itmp[31];
    __syncthreads();

    /* software pipeline: chunk c is scored while chunk c+1 gets its pool swaps and chunk c+2 its seeds and draws */
    int cur = 0, chunk_s[3] = {0, 0, 0}, chunk_base = 0;
    {
        /* the first chunk is 64 samples whatever the variant's chunk size: the first local optimisation runs at sample 50 and leaves the
         * bound that lets the chunks behind it be screened (below) instead of scored */
        int cn0 = max_sam - no_sam; if (cn0 > 64) cn0 = 64; if (cn0 < 0) cn0 = 0;
        int cn1 = max_sam - no_sam - cn0; if (cn1 > DG_CHUNK) cn1 = DG_CHUNK; if (cn1 < 0) cn1 = 0;
        chunk_s[0] = cn0; chunk_s[1] = cn1;
        if (wave == 0) {
            unsigned sd = seed;
            if (cn0 > 0) sd = dg_sample_chunk<4, LDSPTS>(sd, cn0, n, pool, S->seeds3[0], S->draws3[0], S->alm3[0], pscr, lane);
            if (cn1 > 0) sd = dg_sample_draws<4>(sd, cn1, n, S->seeds3[1], S->draws3[1], S->alm3[1], lane);
            if (lane == 0) S->itmp[31] = (int)sd;
        }
        __syncthreads();
        seed = (unsigned)S->itmp[31];
    }
    DG_DEVT(if (tid == 0) { for (int i = 0; i < 8; i++) { S->ph[i] = 0; S->dbg[i] = 0; } S->tq = DG_CLK(); });
#define DG_PHH(i) DG_DEVT(if (tid == 0) { long long tq2_ = DG_CLK(); S->ph[i] += tq2_ - S->tq; S->tq = tq2_; })
    while (!done && no_sam < max_sam) {
        int chunk = chunk_s[cur]; if (chunk > max_sam - no_sam) chunk = max_sam - no_sam;
        c.seeds = S->seeds3[cur]; c.draws = S->draws3[cur]; chunk_base = no_sam;
        DG_PHH(2);
        /* ---- solve: orientation test, 8x9 null vector, near-singularity test; <= 1 model per lane ---- */
        double hm[9], H1m[9]; int valid = 0;
        if (tid < chunk) valid = dg_solve4_lane(P, c.draws[tid], kind, hm, H1m, (double *)&S->ww[wave]);
        {
            unsigned v = (unsigned)valid, incl = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
            if (lane == 63) S->wave_cnt[wave] = incl;
            __syncthreads();
            unsigned wbase = 0;
            for (int w = 0; w < wave; w++) wbase += S->wave_cnt[w];
            unsigned excl = wbase + incl - v;
            S->moff[tid] = (unsigned short)excl;             /* all lanes: moff[k] = scored samples before k, for any k <= DG_CHUNK */
            if (tid < chunk) {
                S->nv[tid] = (unsigned char)valid;
                if (valid) {
                    double *g = c.K->gmodels + (size_t)excl * 18;
#pragma unroll
                    for (int j = 0; j < 9; j++) { g[j] = hm[j]; g[9+j] = kind ? H1m[j] : 0.0; }
                }
            }
            if (tid == DG_T - 1) S->moff[DG_CHUNK] = (unsigned short)(excl + v);
            __syncthreads();
        }
        const int Mtot = __builtin_amdgcn_readfirstlane((int)S->moff[DG_CHUNK]);
        DG_PHH(0);

        /* ---- score chunk c (waves 2.., one wave per model)  ||  pool swaps of chunk c+1 (wave 0)  ||  seeds + draws of chunk c+2 (wave 1) ---- */
        const int nxt = cur == 2 ? 0 : cur + 1, nx2 = nxt == 2 ? 0 : nxt + 1;
        int cn2;
        {
            cn2 = max_sam - (no_sam + chunk_s[cur] + chunk_s[nxt]); if (cn2 > DG_CHUNK) cn2 = DG_CHUNK; if (cn2 < 0) cn2 = 0;
            chunk_s[nx2] = cn2;
            if (wave == 0) {
                if (chunk_s[nxt] > 0) dg_sample_pool<4, LDSPTS>(chunk_s[nxt], n, pool, S->draws3[nxt], S->alm3[nxt], pscr, lane, S->dbg);
            } else if (wave == 1) {
                /* its draws: after the barrier */
                if (cn2 > 0) { unsigned sd = dg_sample_chain<4>(seed, cn2, S->seeds3[nx2], lane, S->dbg); if (lane == 0) S->itmp[31] = (int)sd; }
            }
            /* static round-robin over the scoring waves: waves 2.. when there are more than two, else both waves after their sampler stage */
            /* Screen (Sampson metric): once a local optimisation has run, a sample past the 50th only matters if its J beats
             * tau = min(maxS.J, maxSs.J) (the commit below), and J <= #(d < 9/4 th) <= the division-free candidate count of
             * dg_HDs_maybe_below: a model whose count does not exceed tau gets J = 0 without the exact pass (pinvJ: eight divisions
             * per point).  Every model still counts as scored (n_hds), as in the fundamental-matrix kernel. */
            const double tau_s = (kind == 0 && iter_cnt > 0 && no_sam >= DG_ITER_SAM && th != 0 && !c.rrun && !A.trace) ? (maxS.J < maxSs.J ? maxS.J : maxSs.J)
                : 0.0;
            if (wave >= DG_SW0) {
                constexpr int STR = DG_NW - DG_SW0;
                /* exact score of model mi: I, and J as the reference's sequential sum over the nonzero terms in point order */
                auto exact = [&](const int mi) {
                    /* this wave's pass (dg_hm_wpass: the next step's points in flight, J added tile by tile in point order) from this
                     * wave's block of the local optimisation's LDS tables, which are idle in the main loop */
                    double *lt = dg_hlt_wave(c, wave), *Hm = lt + 64 * DG_PU, *z18 = Hm + 16;
                    const double *g = c.K->gmodels + (size_t)mi * 18;
                    DG_WSYNC();
                    if (lane < 9) Hm[lane] = g[lane];
                    DG_WSYNC();
                    const dg_pass_res r = dg_hm_wpass<LDSPTS>(P, n, kind, Hm, z18, th, (int *)0, 0.0, (int *)0, 0.0, lt, lane);
                    if (lane == 0) { c.K->res_I[mi] = r.I; c.K->res_J[mi] = r.J; }
                };
                if (tau_s > 0) {
                    /* four models of this wave per sweep over the points (a sweep is bound by the latency of its point loads, not by
                     * the 45 flops of a bound), the next step's points in flight */
                    const double tb = (th * 9 / 4) * (1.0 + 1e-6);
                    for (int g0 = wave - DG_SW0; g0 < Mtot; g0 += 4 * STR) {
                        const dg_u4 cb_ = dg_h_screen4<LDSPTS>(P, n, c.K->gmodels, g0, STR, Mtot, tb, lane);
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int mj = g0 + j * STR;
                            const unsigned tot_ = cb_.v[j];
                            if (mj < Mtot) {
                                if (!((double)tot_ > tau_s)) { if (lane == 0) { c.K->res_I[mj] = 0; c.K->res_J[mj] = 0; } }
                                else exact(mj);
                            }
                        }
                    }
                } else {
                    for (int mi = wave - DG_SW0; mi < Mtot; mi += STR) exact(mi);
                }
            }
        }
        c.n_hds += Mtot;
        __syncthreads();
        if (cn2 > 0) seed = (unsigned)S->itmp[31];
        if (cn2 > 0 && wave < DG_CHUNK / 64) {               /* the draws of chunk c+2, one wave per 64 samples (see the F kernel) */
            for (int rd = wave; rd < DG_CHUNK / 64; rd += DG_NW) dg_sample_draws_round<4>(rd, cn2, n, S->seeds3[nx2], S->draws3[nx2], S->alm3[nx2], lane,
                S->dbg);
        }
        DG_PHH(1);

        /* ---- commit: replay exp_ranH.c:547-757 in order ---- */
        int k;
        for (k = 0; k < chunk; k++) {
            if (no_sam >= max_sam) break;
            if (no_sam >= DG_ITER_SAM) {
                /* jump to the next sample that can change state: a scored model beating maxS/maxSs, or (while no LO
                 * has run yet) any scored sample, which fires the first LO (exp_ranH.c:639-640) */
                const double tau = maxS.J < maxSs.J ? maxS.J : maxSs.J;
                const bool first_lo = iter_cnt == 0 && maxSs.I > 4;
                bool ev = tid >= k && tid < chunk && S->nv[tid] && (first_lo || tau < c.K->res_J[S->moff[tid]]);
                unsigned long long bal = __ballot(ev);
                __syncthreads();
                if (lane == 0) S->wave_cnt[wave] = bal ? (unsigned)(wave * 64 + __ffsll((long long)bal) - 1) : 0xffffffffu;
                __syncthreads();
                unsigned kE = S->wave_cnt[0];
                for (int w = 1; w < DG_NW; w++) kE = S->wave_cnt[w] < kE ? S->wave_cnt[w] : kE;
                int stop = kE == 0xffffffffu ? chunk : (int)kE;
                if (stop - k > max_sam - no_sam) stop = k + (max_sam - no_sam);
                no_rej += (stop - k) - ((int)S->moff[stop] - (int)S->moff[k]);
                no_sam += stop - k; k = stop;
                if (k >= chunk || no_sam >= max_sam) break;
            }
            no_sam++;
            if (!S->nv[k]) { no_rej++; continue; }
            const int mi = S->moff[k];
            dg_score Sc = {c.K->res_I[mi], c.K->res_J[mi], 0, 0};
            int new_max = 0, do_iterate = 0;
            const bool ev1 = maxS.J < Sc.J;
            if (ev1 || maxSs.J < Sc.J) {
                __syncthreads();
                if (tid < 9) S->f[tid] = c.K->gmodels[(size_t)mi*18 + tid];
                __syncthreads();
            }
            if (ev1) {
                int pass = 1;
                if (pr.sym_th > 0 || pr.laf_coef > 0) {
                    dg_pass_cfg cl = dg_cfg0(n); cl.list = c.K->L[2]; cl.thL = th;
                    dg_pass_res rl = dg_hm_pass(c, kind, S->f, cl);
                    pass = dg_h_checks(c, kind, S->f, c.K->L[2], (int)rl.nL, Sc, maxS, &p1_inliers, 1);
                }
                if (!pass) continue;
                maxS = Sc; new_max = 1; accepted = 1; best_sample = no_sam; t_best = wall_clock64();
                __syncthreads();
                if (tid < 9) S->F[tid] = S->f[tid];
                __syncthreads();
            }
            if (maxSs.J < Sc.J) {
                do_iterate = no_sam > DG_ITER_SAM;
                maxSs = Sc;
                __syncthreads();
                if (tid < 9) e4[tid] = S->f[tid];
                __syncthreads();
            } else do_iterate = 0;
            if (no_sam >= DG_ITER_SAM && iter_cnt == 0 && maxSs.I > 4) do_iterate = 1;
            if (do_iterate) {
                __syncthreads();
                if (__builtin_amdgcn_readfirstlane(wave) == 0) { dg_srand_wave(&S->rng, c.seeds[k], lane); dg_rand_skip(&S->rng, 5, lane); }
                __syncthreads();
                iter_cnt++;
                DG_PHH(2);
                if (dg_h_lo(c, kind, e4, th, maxS, &iterID, &p1_inliers, no_sam, iter_cnt - 1)) { new_max = 1; accepted = 1; best_sample = no_sam;
                    t_best = wall_clock64(); }
                DG_PHH(3);
            }
            if (new_max) {
                int new_sam = dg_nsamples((int)maxS.I + 1, n, 4, pr.conf);
                if (new_sam < max_sam) max_sam = new_sam;
            }
        }
        if (k < chunk) { c.n_hds -= (Mtot - (int)S->moff[k]); done = 1; }
        else if (no_sam >= max_sam) done = 1;
        __syncthreads();
        if (!done) cur = nxt;
    }

    /* ---- "If there were no LOs, do at least one NOW!"  exp_ranH.c:759-862 ---- */
    if (iter_cnt == 0) {
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(wave) == 0 && no_sam > 0) { int li = no_sam - 1 - chunk_base; if (li < 0) li = 0;
            dg_srand_wave(&S->rng, c.seeds[li], lane); dg_rand_skip(&S->rng, 5, lane); }
        __syncthreads();
        iter_cnt++;
        DG_PHH(2);
        if (dg_h_lo(c, kind, maxSs.J > 0 ? e4 : (const double *)0 /* errs[4] never written */, th, maxS, &iterID, &p1_inliers, no_sam,
            iter_cnt - 1)) { accepted = 1;
            best_sample = no_sam; t_best = wall_clock64(); }
        DG_PHH(3);
    }

    /* ---- final mask: exp_ranH.c:864-907 (this driver indexes the filters correctly) ---- */
    unsigned char *mask = A.mask_out + off;
    if (!accepted) {
        for (int j = tid; j < n; j += DG_T) mask[j] = 0;
    } else {
        __syncthreads();
        if (tid == 0) dg_hsym_prepare(S->F, S->lsq.Z8, S->lsq.Z8 + 9);
        __syncthreads();
        double H[9], Hinv[9], H1[9];
        for (int i = 0; i < 9; i++) { H[i] = S->F[i]; Hinv[i] = S->lsq.Z8[i]; H1[i] = S->lsq.Z8[9+i]; }
        const double thl = pr.laf_coef * th;
        for (int j = tid; j < n; j += DG_T) {
            dg_pt p = dg_ldpt<LDSPTS>(P, j);
            int in = dg_Herr(kind, H, Hinv, H1, p) <= th;
            if (in && pr.sym_th > 0 && dg_Hsym(Hinv, H1, p.x1, p.y1, p.x2, p.y2, 2, 1) > pr.sym_th) in = 0;
            if (in && pr.laf_coef > 0) {
                for (int wch = 1; wch <= 2; wch++) {
                    dg_pt l = c.laf_pt(j, wch);
                    double e = kind == 0 ? dg_HDs_mixed(H, p.x1, p.y1, p.x2, p.y2, l.x1, l.y1, l.x2, l.y2) : dg_Hsym(Hinv, H1, l.x1, l.y1, l.x2, l.y2, kind, 1);
                    if (e > thl) in = 0;
                }
            }
            mask[j] = (unsigned char)in;
        }
    }
    if (tid < 9) A.model_out[(size_t)pair * 9 + tid] = accepted ? S->F[tid] : 0.0;
    if (A.stats_out && tid == 0) {
        int *st = A.stats_out + (size_t)pair * 16;
        long long t_end = wall_clock64();
        st[0] = no_sam; st[1] = iter_cnt; st[2] = no_rej; st[3] = (int)maxS.I; st[4] = c.n_hds;
        st[5] = 0; st[6] = 0; st[7] = best_sample; st[8] = c.n_hds; st[9] = 0; st[10] = 0; st[11] = 0;
        st[12] = (int)(t_best - t_start); st[13] = (int)(t_end - t_start); st[14] = A.variant_threads; st[15] = A.mode;
    }
    hjob_gen = c.hjob_gen;
    if (A.done_pairs && tid == 0) __hip_atomic_fetch_add(A.done_pairs, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    DG_PHH(6);
    DG_DEVT(if (A.phase_out && tid == 0) { S->ph[7] = DG_CLK() - t_start; for (int i = 0; i < 8; i++) A.phase_out[(size_t)pair * 16 + i] = S->ph[i];
        for (int i = 0; i < 8; i++) A.phase_out[(size_t)pair * 16 + 8 + i] = S->dbg[i]; });
#undef DG_PHH
}

template <int T, int LDSPTS>
#define DG_MINW_H DG_MINW
__global__ __launch_bounds__(DG_T, DG_MINW_H) void dg_find_homography_kernel(dg_args A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
    __shared__ dg_f_shared Sh;
    __shared__ int next_pair;
    __shared__ double Hlt[DG_HLT_STATIC_WAVES][DG_HLT];       /* per-wave solver tables of the one-repetition-per-wave local optimisation (dg_hlt_wave) */
    /* the device code reads the arguments through a pointer (dg_f_ctx::A, also inside non-inlined functions): give it an
     * LDS copy, so the by-value kernel argument's address is never taken (that would make the compiler keep a private
     * per-lane copy of the whole block in scratch memory and turn every uniform argument into a vector value) */
    __shared__ dg_args As;
    if (threadIdx.x == 0) As = A;
    __syncthreads();
    int hjob_gen = 0;
    for (;;) {
        const int pair = dg_next_pair(As, &next_pair);
        if (pair < 0) break;
        dg_h_pair<T, LDSPTS>(As, &Sh, dyn_smem, &Hlt[0][0], pair, (int)blockIdx.x, hjob_gen);
        dg_discard_if_failed(As, pair, &next_pair);
    }
    /* out of pairs: help the local optimisations of the pairs that still run, until every pair of the launch is finished */
    if (As.hjob) {
        for (;;) {
            const int j = dg_hjob_find(As, &next_pair);
            if (j < 0) break;
            dg_h_help<T, LDSPTS>(As, &Sh, &Hlt[0][0], j, (int)blockIdx.x);
        }
    }
}

#endif /* DG_KERNEL_H_H */
