/* Local optimisation (exp_ranF.c:621-806) and the persistent F kernel body (exp_ranF.c:1244-1767). */
#ifndef DG_KERNEL_F_MAIN_H
#define DG_KERNEL_F_MAIN_H
#include "dg_kernel_f.h"
#include "dg_score_tiles.h"

/* Wave 0 (all 64 lanes): hash of an id list in global memory (hash.c:4-47 over the ints' bytes).  The list is
 * fetched 64 ids per load (one per lane) and the serial state chain runs on the scalar unit over readlane'd
 * operands, so the cost is ~13 load latencies + 5-9 SALU ops per id instead of one load latency per 8 ids. */
__device__ __forceinline__ unsigned dg_hash_list(const int *list, int count, bool small_ids = false)
{
    if (count <= 0) return 0;
    const int lane = (int)(threadIdx.x & 63);
    unsigned hash = (unsigned)__builtin_amdgcn_readfirstlane(count * 4), tmp;
#define DG_HSTEP(v_) { unsigned v = (unsigned)(v_); \
        hash += v & 0xffffu; tmp = ((v >> 16) << 11) ^ hash; hash = (hash << 16) ^ tmp; hash += hash >> 11; }
    /* ids below 65536 (every LDS-resident pair): the high half-word of each int is 0, so tmp == hash */
#define DG_HSTEP16(v_) { unsigned v = (unsigned)(v_); \
        hash += v; hash = (hash << 16) ^ hash; hash += hash >> 11; }
    int k = 0;
    /* the ids of four 64-id blocks are in flight while one block's chain runs (the list is in global memory) */
    int cur = (lane < count) ? list[lane] : 0;
    int n1 = (64 + lane < count) ? list[64 + lane] : 0, n2 = (128 + lane < count) ? list[128 + lane] : 0, n3 = (192 + lane < count) ? list[192 + lane] : 0;
    for (; k + 64 <= count; k += 64) {
        const int n4 = (k + 256 + lane < count) ? list[k + 256 + lane] : 0;
        if (small_ids) {
#pragma unroll
            for (int i = 0; i < 64; i++) DG_HSTEP16(__builtin_amdgcn_readlane(cur, i))
        } else {
#pragma unroll
            for (int i = 0; i < 64; i++) DG_HSTEP(__builtin_amdgcn_readlane(cur, i))
        }
        cur = n1; n1 = n2; n2 = n3; n3 = n4;
    }
    const int rem = count - k;
    for (int i = 0; i < rem; i++) DG_HSTEP(__builtin_amdgcn_readlane(cur, i))
#undef DG_HSTEP16
#undef DG_HSTEP
    hash ^= hash << 3;  hash += hash >> 5;
    hash ^= hash << 4;  hash += hash >> 17;
    hash ^= hash << 25; hash += hash >> 6;
    return hash;
}

#ifndef DG_AHEAD_ON
#define DG_AHEAD_ON 1
#endif
#ifdef DG_LO_PROF
#define DG_LT(i) do { __syncthreads(); if (c.tid == 0) { long long t_ = wall_clock64(); c.S->lt[i] += t_ - c.S->ltq; c.S->ltq = t_; } } while (0)
#else
#define DG_LT(i) do {} while (0)
#endif

/* exp_ranF.c:621-743 exp_iterFcustom.  f (LDS) is the in/out model parameter `F`; on return *kind0 is the
 * metric variant (FDS1 / EXFDS1) whose residuals the reference would hold in errs[0] for that model. */
template <int LDSPTS>
__device__ __forceinline__ dg_score dg_iterF(CTX &c, int *inliers, double th, double ths, double *f, int iterID,
                                             int mk_full, int mk_ex, int *kind0, int rrow /* first diagnostics row of this repetition's iterations */)
{
    dg_f_shared *S = c.S; const int n = c.n, tid = c.tid;
    double *fl = S->fLO;
    dg_score zero = {0, 0, 0, 0}, maxS = zero, Sc = zero;
    double dth = (ths - th) / DG_ILSQ_ITERS;
    /* errs[4] = errs[0] = FDS1(f): one pass gives inlidxs(.., th) and the list at th*MWM */
    dg_pass_cfg c0 = dg_cfg0(n); c0.wantJ = 1; c0.thJ = th; c0.list = inliers; c0.thL = th * DG_MWM;
    DG_LT(0);
    dg_pass_res r0 = dg_f_pass(c, f, mk_full, c0); c.n_fds++;
    DG_LT(1);
    maxS.I = r0.I; maxS.J = r0.J;
    *kind0 = mk_full;
    DG_TRACE(c, 10, maxS.I, maxS.J);
    if (maxS.I < 8) {
        dg_pass_cfg c1 = dg_cfg0(n); c1.list = inliers; c1.thL = th;      /* the list the reference leaves behind */
        dg_f_pass(c, f, mk_full, c1);
        return zero;
    }
    {
        int cnt = (int)r0.nL;                                              /* S.I at th*MWM */
        DG_TRACE(c, 15, cnt, 0);
        int o = 0, use = cnt;
        __syncthreads();
        if (8 < cnt) { if (tid < 64) { int id; dg_randsubset_wave(&S->rng, inliers, cnt, 8, tid, &id); } use = 8; o = cnt - 8; }
        __syncthreads();
        dg_u2f_list(c, inliers + o, use, 0, 0, fl);
    }
    DG_LT(2);
    for (int it = 0; it < DG_ILSQ_ITERS; it++) {
        /* the same residuals also give the list at ths*MWM that the re-fit uses when this model does not improve */
        int *alt = c.K->L[9];
        dg_pass_cfg c1 = dg_cfg0(n); c1.wantJ = 1; c1.thJ = th; c1.list = inliers; c1.thL = th; c1.list2 = alt; c1.thL2 = ths * DG_MWM;
        dg_pass_res r1 = dg_f_pass(c, fl, mk_ex, c1); c.n_exfds++;
        dg_dump_resid(c, rrow + it, fl, mk_ex);
        DG_LT(3);
        Sc = zero; Sc.I = r1.I; Sc.J = r1.J;
        DG_TRACE(c, 11, Sc.I, Sc.J);
        /* Reference order: hash lookup ("seen" -> return 0), then on improvement rotate the buffers, then the
         * inlidxs(d, ths*MWM) list, then the weighted 8-point re-fit.  exp_ranF.c:687-696: after a rotation `d`
         * is the OLD errs[0], so that list is taken on the residuals of the previous best model of this chain
         * (= the current value of the out-parameter F), not on the new one.  Reproduced.
         * Here the serial hash (wave 1) runs concurrently with the serial re-fit (wave 0): the list goes to a
         * second buffer so the hashed list stays intact, and nothing is committed before the lookup is known. */
        const int improve = maxS.J < Sc.J;
        dg_pass_res r2; r2.nL = r1.nL2;
        if (improve) { dg_pass_cfg c2 = dg_cfg0(n); c2.list = alt; c2.thL = ths * DG_MWM; r2 = dg_f_pass(c, f, *kind0, c2); }
        const int fit = r2.nL >= 8;
        const int wv = tid >> 6;
        __syncthreads();
        DG_LT(4);
        if (wv == 1) {
            unsigned hash = dg_hash_list(inliers, (int)Sc.I, n < 65536);
            if ((tid & 63) == 0) {
                int ret = dg_ht_contains(c.ht, hash, (int)Sc.I, iterID);
                if (ret == -1) dg_ht_insert(c.ht, hash, (int)Sc.I, iterID);
                S->itmp[0] = (ret != -1 && ret != iterID) ? 1 : 0;
            }
        } else if (wv == 0 && fit) {
            const int cnt = (int)r2.nL; int use = cnt, id;
            if (tid == 0) S->rng_save = S->rng;
            DG_WSYNC();
            if (8 < cnt) { dg_randsubset_wave(&S->rng, alt, cnt, 8, tid, &id); use = 8; }
            else id = tid < cnt ? alt[tid] : 0;
            /* u2fw: weights are exFDs' w of the current model at the subset points */
            if (tid < use) {
                dg_pt q = dg_ldpt<LDSPTS>(c.P, id);
                double *px = S->lsq.px + 4*tid; px[0] = q.x1; px[1] = q.y1; px[2] = q.x2; px[3] = q.y2;
                if (mk_ex == DG_K_FDS) S->lsq.part[0][tid] = dg_exFDs_w(fl, q.x1, q.y1, q.x2, q.y2);
                else { double w; dg_exFDsSym(fl, q.x1, q.y1, q.x2, q.y2, &w); S->lsq.part[0][tid] = w; }
            }
            DG_WSYNC();
            dg_u2f_small_w(&S->lsq, S->lsq.px, S->lsq.part[0], use, S->ftmp, tid);
        }
        __syncthreads();
        DG_LT(5);
        if (S->itmp[0]) {
            if (fit && tid == 0) S->rng = S->rng_save;
            __syncthreads();
            DG_TRACE(c, 13, 0, 0); return zero;
        }
        if (improve) {
            maxS = Sc; *kind0 = mk_ex;
            if (tid < 9) f[tid] = fl[tid];
        }
        DG_TRACE(c, 14, r2.nL, 0);
        /* the reference builds this list (and shuffles it) in `inliers` itself, and callers later read stale
         * entries of that buffer (exp_ranF.c:776-779 copies maxS.I ids whatever the list length): keep it identical */
        for (int j = tid; j < (int)r2.nL; j += DG_T) inliers[j] = alt[j];
        if (tid < 9 && fit) fl[tid] = S->ftmp[tid];
        __syncthreads();
        DG_LT(6);
        if (!fit) return maxS;
        ths -= dth;
    }
    dg_pass_cfg c3 = dg_cfg0(n); c3.wantJ = 1; c3.thJ = th; c3.list = inliers; c3.thL = th;
    DG_LT(0);
    dg_pass_res r3 = dg_f_pass(c, fl, mk_full, c3); c.n_fds++;
    dg_dump_resid(c, rrow + 4, fl, mk_full);
    DG_LT(7);
    DG_TRACE(c, 12, r3.I, r3.J);
    if (maxS.J < r3.J) {
        maxS = zero; maxS.I = r3.I; maxS.J = r3.J; *kind0 = mk_full;
        __syncthreads();
        if (tid < 9) f[tid] = fl[tid];
        __syncthreads();
    }
    return maxS;
}

/* exp_ranF.c:745-806 exp_inFranicustom.  inliers = L[0] (in/out), result model -> Fout (LDS). */
template <int LDSPTS>
__device__ __noinline__ dg_score dg_inFrani_serial(CTX &c, int ninl, double th, double *Fout, int *iterID,
                                               int mk_full, int mk_ex, int *kindBest)
{
    dg_f_shared *S = c.S; const int tid = c.tid;
    int *inliers = c.K->L[0], *intbuff = c.K->L[1], *intbuff_best = c.K->L[2];
    dg_score maxS = {0, 0, 0, 0};
    *kindBest = mk_full;
    if (ninl < 16) {
        if (c.rrun) { for (size_t j = tid; j < (size_t)(DG_RESIDS_M - 2) * c.n; j += DG_T) c.rrun[2 * (size_t)c.n + j] = 0.; __syncthreads(); }   /* exp_ranF.c:761 */
        return maxS;
    }
    int ssiz = ninl / 2; if (ssiz > 14) ssiz = 14;
    /* The ten repetitions are chained through the generator and the list order only: repetition i+1 draws its sample from
     * the state repetition i's iterF leaves, and iterF advances the generator by 8 draws per re-fit subset — 2 subsets in
     * 62 % of the repetitions, 3 in 26 %, 4 in 8 % (C2 data).  While wave 0 fits this repetition's sample (one 9x9
     * eigen-problem, the other waves would idle), the other waves each prepare the NEXT repetition's sample and model
     * for one of those counts on a private copy of the generator, without touching the list.  The next repetition
     * compares its generator state with the prepared ones and, on a match, stores the prepared list slots and takes the
     * model instead of drawing and fitting; otherwise it proceeds as if nothing had been prepared. */
    const int wave = tid >> 6, lane = tid & 63;
    if (tid == 0) S->n_ahead = 0;
    for (int i = 0; i < DG_RAN_REP; i++) {
        DG_LT(0);
        __syncthreads();
        int taken = 0;
        if (S->n_ahead > 0) {
            if (tid < 64) {
                int hit = -1;
                for (int k = 0; k < S->n_ahead; k++) {
                    const int *a = (const int *)&S->rng, *b = (const int *)&S->ahead[k].before;
                    const bool same = lane < 31 ? a[lane] == b[lane] : (lane == 31 ? S->rng.f == S->ahead[k].before.f : (lane == 32 ? S->rng.b == S->ahead[k].before.b : true));
                    if (hit < 0 && __ballot(!same) == 0ull) hit = k;
                }
                if (hit >= 0) {
                    const dg_lo_ahead *h = &S->ahead[hit];
                    if (lane < 2 * ssiz && h->pos[lane] >= 0) inliers[h->pos[lane]] = h->val[lane];
                    if (lane < 9) S->f[lane] = h->F[lane];
                    DG_WSYNC();
                    if (lane == 0) S->rng = h->after;
                }
                if (lane == 0) S->itmp[29] = hit;
            }
            __syncthreads();
            taken = S->itmp[29] >= 0;
        }
        if (!taken) {
            if (tid < 64) { int id; dg_randsubset_wave(&S->rng, inliers, ninl, ssiz, tid, &id); dg_gather_wave(c, id, ssiz, S->lsq.px, tid); }
            __syncthreads();
            if (wave == 0) {
                dg_u2f_small_w(&S->lsq, S->lsq.px, 0, ssiz, S->f, tid);
            } else if (DG_AHEAD_ON && wave <= DG_LO_AHEAD && ssiz > 8 && i + 1 < DG_RAN_REP) {
                /* subsets assumed for this repetition's iterF, most frequent first */
                const int sub = wave == 1 ? 2 : wave == 2 ? 3 : wave == 3 ? 4 : wave == 4 ? 1 : 5;
                dg_lo_ahead *h = &S->ahead[wave - 1];
                dg_wave_ws *w = &S->ww[wave];
                if (lane == 0) { h->before = S->rng; for (int q = 0; q < 8 * sub; q++) dg_rand(&h->before); h->after = h->before; }
                DG_WSYNC();
                int id;
                dg_randsubset_wave_ahead(&h->after, inliers, ninl, ssiz, lane, &id, h->pos, h->val);
                dg_gather_wave(c, id, ssiz, w->px, lane);
                DG_WSYNC();
                dg_u2f_norm_w(w, w->px, (const double *)0, ssiz, h->F, lane);
            }
            if (tid == 0) S->n_ahead = (DG_AHEAD_ON && ssiz > 8 && i + 1 < DG_RAN_REP) ? DG_LO_AHEAD : 0;
        } else if (tid == 0) S->n_ahead = 0;
        DG_LT(8);
#ifdef DG_LO_PROF
        if (c.tid == 0) { if (taken) c.S->lt[10]++; else c.S->lt[11]++; }
#endif
        __syncthreads();
        int k0;
        ++*iterID;
        dg_dump_resid(c, 2 + 6 * i, S->f, mk_full);                       /* errs[0] = FDS1(f): exp_ranF.c:776-779 */
        dg_score Sc = dg_iterF(c, intbuff, th, DG_TC * th, S->f, *iterID, mk_full, mk_ex, &k0, 2 + 6 * i + 1);
        if (maxS.J < Sc.J) {
            maxS = Sc; *kindBest = k0;
            __syncthreads();
            if (tid < 9) Fout[tid] = S->f[tid];
            for (int j = tid; j < (int)maxS.I; j += DG_T) intbuff_best[j] = intbuff[j];
            __syncthreads();
        }
    }
    __syncthreads();
    for (int j = tid; j < (int)maxS.I; j += DG_T) inliers[j] = intbuff_best[j];
    __syncthreads();
    return maxS;
}

/* ---- the local optimisation with one repetition per wave --------------------------------------------------------------
 * The ten repetitions of exp_inFranicustom (exp_ranF.c:745-806) are chained through the generator (14 draws for the sample,
 * then one 8-subset per re-fit of exp_iterFcustom: two of them in 62 % of the repetitions, three in 26 %), the order of
 * `inliers`, the inlier-set hash table ("seen by an earlier repetition" ends a repetition) and the best-so-far comparison.
 * A round runs DG_NW repetitions concurrently, one per wave, each on its own lists / MSAC-term buffer in the workspace and
 * its own solver scratch: wave 0 draws the round's samples one after the other from generator states that ASSUME two
 * 8-subsets per earlier repetition of the round; every wave then runs its whole repetition (same fits, same passes, J as the
 * reference's sequential sum, the hash of every iteration's inlier set) without touching the hash table, stopping only at a
 * set that an EARLIER round or local optimisation inserted; afterwards thread 0 replays the repetitions in order — hash
 * lookups / inserts with the repetition's own iterID, "seen by another repetition -> empty result", the draws it really
 * consumed — and commits them as long as the assumption behind their start state held (the first one always does).  The
 * generator is set to the exact state behind the last committed repetition and `inliers` put back into the order its sample
 * left; the next round starts there.  A repetition that the replay cuts short has only computed further than needed.
 * Results and counters equal the serial order (dg_inFrani_serial, kept for the residual dump, the cooperative large-n
 * mode and behind MI_DEGENSAC_TUNE_F_SERIAL_REPS for the equality test). */
/* one wave's pass of model Fm under metric `kind` over all n points: I = #(d <= thJ), J = the reference-order MSAC sum, the
 * ordered id lists at thL (la) and thL2 (lb, optional); `tile` = the wave's LDS tile for the MSAC terms (dg_wpass_impl) */
template <int LDSPTS>
__device__ __noinline__ dg_pass_res dg_f_wpass(const dg_pt *P, int n, int kind, const double *Fm /* LDS */, double thJ, int *la_, double thL, int *lb_, double thL2,
                                               double *tile, int lane)
{
    n = __builtin_amdgcn_readfirstlane(n); kind = __builtin_amdgcn_readfirstlane(kind);
    double F[9];
#pragma unroll
    for (int i = 0; i < 9; i++) F[i] = Fm[i];
    return dg_wpass_impl<LDSPTS>(P, n, [&](const dg_pt &q) { return dg_Ferr(kind, F, q); }, thJ, la_, thL, lb_, thL2, tile, lane);
}

#define DG_LO_ASSUMED_DRAWS 16          /* two 8-subsets per repetition: the most frequent count (62 % on C2 data) */
/* one repetition (sample lg->ids, generator lg->g right behind the sample's draws, iterID for the table lookups) by one wave:
 * the 14-point fit and exp_iterFcustom (exp_ranF.c:621-743) */
template <int LDSPTS>
__device__ __noinline__ void dg_lo_rep_wave(CTX &c, dg_lo_log *lg, int ssiz, double th, int mk_full, int mk_ex, int lane, int wave)
{
    dg_f_shared *S = c.S; const int n = c.n, nm = c.K->n_max; const dg_pt *P = c.P;
    dg_wave_ws *w = &S->ww[wave];
    int *ib = c.K->wlist + (size_t)wave * nm;                                    /* this repetition's `inliers` (intbuff) */
    int *alt = (int *)(c.K->wstage + (size_t)wave * nm);
    double *jb = w->Z;                                                           /* the passes' MSAC-term tile: Z .. px, idle during a pass */
    double *f = w->F, *fl = w->H, *wts = w->cpx, *ftmp = w->cpx + 8;
    const bool small_ids = n < 65536;
    /* an earlier repetition of this round has finished with another number of draws than this one's start state assumes:
     * this repetition will not be committed, stop it */
#ifdef DG_LO_PROF
    long long tw_ = wall_clock64(); const long long tw0_ = tw_;
#define DG_RW(i) do { if (wave == 0 && lane == 0) { long long t_ = wall_clock64(); S->lt[i] += t_ - tw_; tw_ = t_; } } while (0)
#else
#define DG_RW(i) do {} while (0)
#endif
    auto stale = [&]() {
        int bad = 0;
        if (lane < wave) { const int d = __hip_atomic_load(&S->lo[lane].pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); bad = d >= 0 && d != DG_LO_ASSUMED_DRAWS; }
        return __ballot(bad) != 0ull;
    };
    int drawn = 0;
    /* the sample's model */
    DG_WSYNC();
    dg_gather_wave(c, lane < ssiz ? lg->ids[lane] : 0, ssiz, w->px, lane);
    DG_WSYNC();
    dg_u2f_small_wave(w, w->px, (const double *)0, ssiz, f, lane);
    DG_RW(8);
    /* errs[4] = errs[0] = FDS1(f): inlidxs(.., th) and the list at th * MWM */
    const dg_pass_res r0 = dg_f_wpass<LDSPTS>(P, n, mk_full, f, th, ib, th * DG_MWM, (int *)0, 0.0, jb, lane);
    DG_RW(9);
    unsigned mI = r0.I; double mJ = r0.J; int kind0 = mk_full;
    if (lane == 0) { lg->I0 = (int)r0.I; lg->drew0 = 0; lg->nit = 0; lg->has_fin = 0; }
    if (mI < 8) { if (lane == 0) { lg->I = 0; lg->J = 0; lg->kind0 = mk_full; __hip_atomic_store(&lg->pub, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } DG_WSYNC(); return; }
    /* the first 8-point model */
    {
        const int cnt = (int)r0.nL; int id;
        if (8 < cnt) { dg_randsubset_wave(&lg->g, ib, cnt, 8, lane, &id); if (lane == 0) lg->drew0 = 8; drawn += 8; }
        else id = lane < cnt ? ib[lane] : 0;
        const int use = 8 < cnt ? 8 : cnt;
        DG_WSYNC();
        dg_gather_wave(c, id, use, w->px, lane);
        DG_WSYNC();
        dg_u2f_small_wave(w, w->px, (const double *)0, use, fl, lane);
    }
    DG_RW(10);
    double ths = DG_TC * th; const double dth = (ths - th) / DG_ILSQ_ITERS;
    int it = 0, ended = 0;
    for (; it < DG_ILSQ_ITERS; it++) {
        if (stale()) { if (lane == 0) lg->aborted = 1; DG_WSYNC(); return; }
        const dg_pass_res r1 = dg_f_wpass<LDSPTS>(P, n, mk_ex, fl, th, ib, th, alt, ths * DG_MWM, jb, lane);
        const int improve = mJ < r1.J;
        unsigned nL2 = r1.nL2;
        /* exp_ranF.c:687-696: after a rotation `d` is the OLD errs[0]: that list is taken on the residuals of the previous best */
        if (improve) { const dg_pass_res r2 = dg_f_wpass<LDSPTS>(P, n, kind0, f, 0.0, alt, ths * DG_MWM, (int *)0, 0.0, jb, lane); nL2 = r2.nL; }
        const int fit = nL2 >= 8;
        DG_WSYNC();
        DG_RW(9);
        const unsigned hash = dg_hash_list(ib, (int)r1.I, small_ids);
        if (lane == 0) { lg->it[it].hash = hash; lg->it[it].I = (int)r1.I; lg->it[it].drew = 0; lg->nit = it + 1; }
        /* a set an EARLIER round or local optimisation inserted ends the repetition here whatever the others of this round do
         * (the table is not written before the replay) */
        { int known = 0; if (lane == 0) known = dg_ht_contains(c.ht, hash, (int)r1.I, -1) != -1; DG_RW(11); if (__builtin_amdgcn_readfirstlane(known)) { ended = 2; break; } }
        if (fit) {
            const int cnt = (int)nL2; int id;
            if (8 < cnt) {
                dg_randsubset_wave(&lg->g, alt, cnt, 8, lane, &id); if (lane == 0) lg->it[it].drew = 8; drawn += 8;
                /* more draws than the later repetitions of this round assume: they will not be committed whatever follows — tell them now */
                if (drawn > DG_LO_ASSUMED_DRAWS && lane == 0) __hip_atomic_store(&lg->pub, drawn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            else id = lane < cnt ? alt[lane] : 0;
            const int use = 8 < cnt ? 8 : cnt;
            DG_WSYNC();
            if (lane < use) {
                const dg_pt q = dg_ldpt<LDSPTS>(P, id);
                double *px = w->px + 4 * lane; px[0] = q.x1; px[1] = q.y1; px[2] = q.x2; px[3] = q.y2;
                if (mk_ex == DG_K_FDS) wts[lane] = dg_exFDs_w(fl, q.x1, q.y1, q.x2, q.y2);
                else { double ww_; dg_exFDsSym(fl, q.x1, q.y1, q.x2, q.y2, &ww_); wts[lane] = ww_; }
            }
            DG_WSYNC();
            dg_u2f_small_wave(w, w->px, wts, use, ftmp, lane);
        }
        if (improve) { mI = r1.I; mJ = r1.J; kind0 = mk_ex; DG_WSYNC(); if (lane < 9) f[lane] = fl[lane]; DG_WSYNC(); }
        /* the reference builds this list (and shuffles it) in `inliers` itself */
        for (int j = lane; j < (int)nL2; j += 64) ib[j] = alt[j];
        DG_WSYNC();
        if (lane < 9 && fit) fl[lane] = ftmp[lane];
        DG_WSYNC();
        DG_RW(10);
        if (!fit) { ended = 1; break; }
        ths -= dth;
    }
    if (!ended) {
        const dg_pass_res r3 = dg_f_wpass<LDSPTS>(P, n, mk_full, fl, th, ib, th, (int *)0, 0.0, jb, lane);
        if (lane == 0) lg->has_fin = 1;
        if (mJ < r3.J) { mI = r3.I; mJ = r3.J; kind0 = mk_full; DG_WSYNC(); if (lane < 9) f[lane] = fl[lane]; DG_WSYNC(); }
    }
    DG_WSYNC();
    if (lane < 9) lg->f[lane] = f[lane];
#ifdef DG_LO_PROF
    DG_RW(9);
    if (wave == 0 && lane == 0) { S->lt[12] += wall_clock64() - tw0_; S->lt[13] += 100000; }
#endif
    if (lane == 0) { lg->I = (int)mI; lg->J = mJ; lg->kind0 = kind0; __hip_atomic_store(&lg->pub, drawn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    DG_WSYNC();
}

template <int LDSPTS> __device__ __forceinline__ void dg_lo_round_coop(CTX &c, int nr, int ssiz, double th, int mk_full, int mk_ex);
__device__ __forceinline__ char *dg_coop_lojob(const dg_args &A, int slot);
__device__ __forceinline__ int *dg_coop_lo_list(const dg_args &A, int slot, int k);
/* COOP (cooperative large-n mode): a round is all the repetitions that are left, each run by one claiming workgroup of the
 * pair (stage 4; dg_lo_rep_wg) on records and lists in the owner's workspace, and the draws assumed per repetition are the
 * ones the last committed repetition consumed (long lists: 8 + 4 x 8, where 2000-point pairs mostly stop after 16) */
template <int LDSPTS, bool COOP>
__device__ __noinline__ dg_score dg_inFrani_waves(CTX &c, int ninl, double th, double *Fout, int *iterID, int mk_full, int mk_ex, int *kindBest)
{
    dg_f_shared *S = c.S; const int tid = c.tid, lane = tid & 63, wave = tid >> 6;
    int *inliers = c.K->L[0], *intbuff_best = c.K->L[2];
    dg_score maxS = {0, 0, 0, 0};
    *kindBest = mk_full;
    if (ninl < 16) return maxS;
    int ssiz = ninl / 2; if (ssiz > 14) ssiz = 14;
    int next = 0;
    /* repetitions per round: one per wave, or fewer (dg_args::lo_width): every repetition behind the first rests on an assumption that
     * holds 63 % of the time, so a wide round buys latency with wasted wave time */
    const int NRMAX = COOP ? DG_RAN_REP : ((c.A->lo_width > 0 && c.A->lo_width < DG_NW) ? c.A->lo_width : DG_NW);
    char *glog = (char *)0;
    if constexpr (COOP) glog = dg_coop_lojob(*c.A, c.coop_slot) + 128;
    auto LG = [&](int q) -> dg_lo_log * { if constexpr (COOP) return (dg_lo_log *)(glog + (size_t)DG_LOJOB_STRIDE * q); else return &S->lo[q]; };
    int assumed = COOP ? c.lo_assumed : DG_LO_ASSUMED_DRAWS;
#ifdef DG_LO_PROF
#define DG_LW(i) do { if (tid == 0) { long long t_ = wall_clock64(); S->lt[i] += t_ - S->ltq; S->ltq = t_; } } while (0)
#else
#define DG_LW(i) do {} while (0)
#endif
    while (next < DG_RAN_REP) {
        const int nr = DG_RAN_REP - next < NRMAX ? DG_RAN_REP - next : NRMAX;
        __syncthreads();
        DG_LW(7);
        if (__builtin_amdgcn_readfirstlane(wave) == 0) {
            if (lane == 0) S->lo_work = S->rng;
            DG_WSYNC();
            for (int q = 0; q < nr; q++) {
                /* the sample of repetition next + q: the draws, the slots they store (kept with the values they replace) */
                dg_lo_log *g = LG(q);
                int id = 0;
                dg_randsubset_wave_ahead(&S->lo_work, inliers, ninl, ssiz, lane, &id, g->upos, g->uval);
                if (lane < ssiz) g->ids[lane] = id;
                if (lane < 2 * ssiz && g->upos[lane] >= 0) { const int old = inliers[g->upos[lane]]; inliers[g->upos[lane]] = g->uval[lane]; g->uval[lane] = old; }
                if (lane == 0) { g->g = S->lo_work; g->g0 = S->lo_work; g->pub = -1; g->aborted = 0; for (int k = 0; k < assumed; k++) dg_rand(&S->lo_work); }
                DG_WSYNC();
            }
        }
        __syncthreads();
        DG_LW(0);
        if constexpr (COOP) dg_lo_round_coop<LDSPTS>(c, nr, ssiz, th, mk_full, mk_ex);
        else { if (wave < nr) dg_lo_rep_wave<LDSPTS>(c, &S->lo[wave], ssiz, th, mk_full, mk_ex, lane, wave); }
        __syncthreads();
        DG_LW(1);
        /* replay in repetition order (thread 0): the hash table with each repetition's own iterID, what it really drew */
        if (tid == 0) {
            int v = 0;
            for (int q = 0; q < nr; q++) {
                dg_lo_log *g = LG(q);
                if (g->aborted) break;                            /* stopped as stale: it runs again in the next round (q >= 1 here) */
                const int id = *iterID + next + q + 1;
                int draws = 0, cut = 0, n_ex = 0;
                if (g->I0 >= 8) {
                    draws = g->drew0;
                    for (int i = 0; i < g->nit; i++) {
                        n_ex++;
                        const int ret = dg_ht_contains(c.ht, g->it[i].hash, g->it[i].I, id);
                        if (ret == -1) dg_ht_insert(c.ht, g->it[i].hash, g->it[i].I, id);
                        if (ret != -1 && ret != id) { cut = 1; break; }
                        draws += g->it[i].drew;
                    }
                }
                g->cut = cut; g->draws = draws; g->n_ex = n_ex; g->n_fd = (!cut && g->has_fin) ? 2 : 1;
                v++;
                if (draws != assumed) break;
            }
            S->red.bi[0] = v;
        }
        __syncthreads();
        DG_LW(2);
        const int v = S->red.bi[0];
#ifdef DG_LO_PROF
        if (tid == 0) { S->lt[5] += 100000; S->lt[6] += 100000 * v; }
#endif
        for (int q = 0; q < v; q++) {
            const dg_lo_log *g = LG(q);
            c.n_exfds += g->n_ex; c.n_fds += g->n_fd;
            const int cut = g->cut;
            if (!cut && maxS.J < g->J) {
                maxS.I = (unsigned)g->I; maxS.J = g->J; maxS.Is = 0; maxS.Ilafs = 0; *kindBest = g->kind0;
                const int *ibq = COOP ? dg_coop_lo_list(*c.A, c.coop_slot, 2 * q) : c.K->wlist + (size_t)q * c.K->n_max;
                __syncthreads();
                if (tid < 9) Fout[tid] = g->f[tid];
                for (int j = tid; j < g->I; j += DG_T) intbuff_best[j] = ibq[j];
                __syncthreads();
            }
        }
        __syncthreads();
        DG_LW(3);
        if (__builtin_amdgcn_readfirstlane(wave) == 0) {
            /* the list order behind repetition next + v - 1: undo the samples of the repetitions that were not committed, last first;
             * the exact generator state behind it: the state behind its sample, then the draws it really consumed */
            for (int q = nr - 1; q >= v; q--) {
                if (lane < 2 * ssiz && LG(q)->upos[lane] >= 0) inliers[LG(q)->upos[lane]] = LG(q)->uval[lane];
                DG_WSYNC();
            }
            if (lane == 0) { S->rng = LG(v - 1)->g0; for (int k = 0; k < LG(v - 1)->draws; k++) dg_rand(&S->rng); }
            DG_WSYNC();
        }
        next += v;
        if constexpr (COOP) {
            /* the assumption follows the committed repetitions, but one odd count (a repetition cut short) does not change it */
            for (int q = 0; q < v; q++) { const int d = LG(q)->draws; if (d == c.lo_prev || c.lo_prev < 0) assumed = d; c.lo_prev = d; }
        }
#ifdef DG_LO_PROF
        __syncthreads();
#endif
        DG_LW(4);
    }
    if constexpr (COOP) c.lo_assumed = assumed;
    *iterID += DG_RAN_REP;
    __syncthreads();
    for (int j = tid; j < (int)maxS.I; j += DG_T) inliers[j] = intbuff_best[j];
    __syncthreads();
    return maxS;
}

template <int LDSPTS>
__device__ __forceinline__ dg_score dg_inFrani(CTX &c, int ninl, double th, double *Fout, int *iterID, int mk_full, int mk_ex, int *kindBest)
{
    /* the serial order for the residual dump (its rows are written in repetition order) and on request */
    if (c.rrun || c.A->innerh_serial || c.A->trace) return dg_inFrani_serial<LDSPTS>(c, ninl, th, Fout, iterID, mk_full, mk_ex, kindBest);
    if (c.cb) {
        /* cooperative large-n mode: whole repetitions go to the claiming workgroups (the serial order distributes every pass instead) */
        if constexpr (LDSPTS == 0) return dg_inFrani_waves<LDSPTS, true>(c, ninl, th, Fout, iterID, mk_full, mk_ex, kindBest);
        else return dg_inFrani_serial<LDSPTS>(c, ninl, th, Fout, iterID, mk_full, mk_ex, kindBest);
    }
    return dg_inFrani_waves<LDSPTS, false>(c, ninl, th, Fout, iterID, mk_full, mk_ex, kindBest);
}

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

/* One chunk of the reference's sample stream, executed by ONE wave (all 64 lanes): the seed chain
 * (seed_{k+1} = output #NDRAW after srand(seed_k)), the NDRAW draws of every sample and the Fisher-Yates
 * pool swaps (rtools.c:12-23).  Fills seeds[0..cn) and draws[k][0..NDRAW) (drawn ids in draw order) and
 * returns the seed of the sample after the chunk.  NDRAW = 7 (F) or 4 (H). */
/* Sampler stage 1 (one wave): the seed chain of a chunk and the raw draws of every sample.
 * Returns the seed that follows the chunk; almask[] (LDS) receives the per-sample alias flags. */
template <int NDRAW>
__device__ __noinline__ unsigned dg_sample_chain(unsigned seed, int cn, unsigned *seeds, int lane, long long *dbg = 0)
{
    long long ts0 = DG_CLK();
    __builtin_amdgcn_s_setprio(3);                        /* the serial waves must not queue behind the scoring waves */
    /* seed chain: lane j carries the term C[NDRAW][j] * r_j, r_j = seed * 16807^j mod (2^31-1) */
    const unsigned gk = lane < 31 ? dg_rng_G[lane] : 0u, ck = lane < 31 ? dg_rng_C[NDRAW][lane] : 0u;
    unsigned sd = seed;
    for (int k = 0; k < cn; k++) {
        if (lane == 0) seeds[k] = sd;
        unsigned s1 = sd ? sd : 1u;                              /* rand() outputs are < 2^31: Schrage == exact mulmod */
        unsigned rj = lane == 0 ? s1 : dg_mulmod31(s1, gk);
        sd = dg_wave_sum_u(ck * rj) >> 1;
    }
    DG_WSYNC();
    __builtin_amdgcn_s_setprio(0);
    DG_DEVT(if (dbg && lane == 0) { dbg[4] += DG_CLK() - ts0; });
    return sd;
}
/* Sampler stage 1b: the draws of the samples 64 rd .. 64 rd + 63 of a chunk whose seeds are known (lane = sample) + the
 * per-sample alias flag: two draws on the same position, or a draw inside the tail block, make the swaps of that sample
 * order-dependent -> replayed sequentially in stage 2.  The rounds of a chunk are independent: one wave each. */
template <int NDRAW>
__device__ __noinline__ void dg_sample_draws_round(int rd, int cn, int n, const unsigned *seeds, int (*draws)[8], unsigned long long *almask, int lane, long long *dbg = 0)
{
    long long ts1 = DG_CLK();
    const int k = rd * 64 + lane;
    bool al = false;
    if (k < cn) {
        unsigned o[8]; int dr[NDRAW];
        dg_rng_outputs(seeds[k], o);
#pragma unroll
        for (int i = 0; i < NDRAW; i++) { dr[i] = (int)(o[i] % (unsigned)(n - i)); draws[k][i] = dr[i]; al = al || dr[i] >= n - NDRAW; }
#pragma unroll
        for (int i = 0; i < NDRAW; i++)
#pragma unroll
            for (int j = i + 1; j < NDRAW; j++) al = al || dr[i] == dr[j];
    }
    unsigned long long b = __ballot(al);
    if (lane == 0) almask[rd] = b;
    DG_WSYNC();
    DG_DEVT(if (dbg && lane == 0 && rd == 0) { dbg[5] += DG_CLK() - ts1; });
}
/* Sampler stage 1 on ONE wave (prologue of the kernels, unit-test kernel): the seed chain of a chunk, then its draws.
 * Returns the seed that follows the chunk; almask[] (LDS) receives the per-sample alias flags. */
template <int NDRAW>
__device__ __forceinline__ unsigned dg_sample_draws(unsigned seed, int cn, int n, unsigned *seeds, int (*draws)[8],
                                                  unsigned long long *almask, int lane, long long *dbg = 0)
{
    const unsigned sd = dg_sample_chain<NDRAW>(seed, cn, seeds, lane, dbg);
    for (int rd = 0; rd < DG_CHUNK / 64; rd++) dg_sample_draws_round<NDRAW>(rd, cn, n, seeds, draws, almask, lane, dbg);
    return sd;
}

/* Sampler stage 2 (one wave): the pool swaps of a chunk (rtools.c:12-23) turn the raw draws into drawn ids.
 * Lanes 0..NDRAW-1 own one draw each, the NDRAW tail slots live in registers.  Software-pipelined: LDS
 * operations of one wave execute in issue order (read_k, write_k, read_{k+1}, ...), so read_{k+1} is issued
 * before read_k's result is consumed; draw positions are prefetched two ahead. */
template <int NDRAW, int LDSPTS>
__device__ __noinline__ void dg_sample_pool_seq(int cn, int n, int *pool, int (*draws)[8], const unsigned long long *almask_in,
                                                int lane, long long *dbg = 0)
{
    long long ts2 = DG_CLK();
    __builtin_amdgcn_s_setprio(3);
    unsigned long long almask[DG_CHUNK / 64];
#pragma unroll
    for (int rd = 0; rd < DG_CHUNK / 64; rd++) almask[rd] = almask_in[rd];
    int *vp = pool;
    const bool act = lane < NDRAW;
    int t = act ? vp[n - 1 - lane] : 0;
#define DG_AL(k_) ((int)((almask[(k_) >> 6] >> ((k_) & 63)) & 1ull))
    int s0 = act ? draws[0][lane] : 0, s1 = (act && cn > 1) ? draws[1][lane] : 0;
    int al0 = DG_AL(0), al1 = cn > 1 ? DG_AL(1) : 1;
    int r0 = (act && !al0) ? vp[s0] : 0;                  /* read_0 */
    for (int k = 0; k < cn; k++) {
        int s2 = (act && k + 2 < cn) ? draws[k + 2][lane] : 0;
        int al2 = k + 2 < cn ? DG_AL(k + 2) : 1;
        int r1 = 0;
        if (al0) {
            /* order-dependent sample: replay it sequentially on lane 0 */
            if (act) vp[n - 1 - lane] = t;
            if (LDSPTS == 0) __threadfence_block();
            DG_WSYNC();
            if (lane == 0) {
                for (int i = 0; i < NDRAW; i++) { int si = draws[k][i], j = n - 1 - i, q = vp[si]; vp[si] = vp[j]; vp[j] = q; draws[k][i] = q; }
            }
            if (LDSPTS == 0) __threadfence_block();
            DG_WSYNC();
            if (act) t = vp[n - 1 - lane];
            if (act && !al1 && k + 1 < cn) r1 = vp[s1];
        } else {
            if (act) vp[s0] = t;                                          /* write_k  (t = result of read_{k-1}) */
            if (LDSPTS == 0) __threadfence_block();
            if (act && !al1 && k + 1 < cn) r1 = vp[s1];                   /* read_{k+1} */
            if (act) { t = r0; draws[k][lane] = r0; }                     /* consume read_k */
        }
        s0 = s1; s1 = s2; al0 = al1; al1 = al2; r0 = r1;
    }
#undef DG_AL
    if (act) vp[n - 1 - lane] = t;
    DG_WSYNC();
    __builtin_amdgcn_s_setprio(0);
    DG_DEVT(if (dbg && lane == 0) { long long ts3 = DG_CLK(); dbg[6] += ts3 - ts2; });
}

/* ---------------------------------------------------------------------------------------------- */
/* Scoring phase of one wave (own register allocation).  The chunk's models are dealt round-robin to the NS scoring
 * waves (wave ws takes mi = ws, ws + NS, ...); lane j of the wave owns the wave's j-th model of the current batch of up
 * to 64.  Screens (dg_score_tiles.h): level 1 when tau >= 64, level 2 when tau >= 4, each tile-major over the whole
 * point set with the models' coefficients in this wave's LDS table `tab`; models whose count does not exceed tau get
 * J = 0 (never an event in the commit, so decisions are unchanged); the survivors are scored exactly, one wave per
 * model: I, and J as the reference's sequential sum (dg_seq_sum): the wave stores the nonzero terms in point order, lane 0
 * adds them one after the other. */
template <int LDSPTS>
__device__ __noinline__ void dg_score_chunk_F(const dg_pt *P, int n, const double *gmodels, const unsigned short *mslot,
                                             int Mtot, int ws, int NS, int kind, double th, double tauJ, const double *ext /* LDS[4] */,
                                             char *tab /* LDS, this wave's */, int tab_bytes,
                                             double *jbuf /* this wave's scratch, >= n doubles */, unsigned *res_I, double *res_J, int lane,
                                             unsigned *scnt /* LDS[4]: dg_f_shared::scnt */)
{
    /* workgroup-uniform arguments arrive in vector registers (separate function): make the loop control scalar again */
    n = __builtin_amdgcn_readfirstlane(n); Mtot = __builtin_amdgcn_readfirstlane(Mtot); ws = __builtin_amdgcn_readfirstlane(ws);
    NS = __builtin_amdgcn_readfirstlane(NS); kind = __builtin_amdgcn_readfirstlane(kind); tab_bytes = __builtin_amdgcn_readfirstlane(tab_bytes);
    const double t94 = th * 9 / 4, t94b = t94 * (1.0 + 1e-6);
    const bool use_bound = th != 0 && kind != DG_K_EXFSYM && tauJ >= 4.0;
    const bool use_l1 = use_bound && tauJ >= 64.0;
    const int nm = Mtot > ws ? (Mtot - ws + NS - 1) / NS : 0;
    int B1 = tab_bytes / (DG_L1_ENTRY_FLOATS * (int)sizeof(float)), B2 = tab_bytes / (DG_L2_ENTRY_DOUBLES * (int)sizeof(double));
    B1 = B1 > 64 ? 64 : B1; B2 = B2 > 64 ? 64 : B2;
    float *tab_f = (float *)tab; double *tab_d = (double *)tab;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int j0 = 0; j0 < nm; j0 += 64) {
        const int nb = nm - j0 < 64 ? nm - j0 : 64;
        const bool have = lane < nb;
        const int mi = ws + (j0 + lane) * NS;                       /* this lane's model (have) */
        double F[9];
        {
            const double *gp = gmodels + (size_t)mslot[have ? mi : ws] * 9;
#pragma unroll
            for (int j = 0; j < 9; j++) F[j] = gp[j];
        }
        unsigned long long surv = __ballot(have);
        unsigned n_all = (unsigned)nb, n_l1 = 0, n_l2 = 0;
        if (use_l1) {
            n_l1 = (unsigned)nb;
            unsigned C1 = 0;
            for (int s0 = 0; s0 < nb; s0 += B1) {
                const int sb = nb - s0 < B1 ? nb - s0 : B1;
                const bool in = lane >= s0 && lane < s0 + sb;
                if (in) {
                    float Ff[9]; const float thr = dg_l1_setup(kind, F, ext, t94b, Ff);
                    float *e = tab_f + (lane - s0) * DG_L1_ENTRY_FLOATS;
#pragma unroll
                    for (int j = 0; j < 9; j++) e[j] = Ff[j];
                    e[9] = thr; e[10] = 0.f; e[11] = 0.f;
                }
                DG_WSYNC();
                const unsigned cq = dg_l1_tile_counts<LDSPTS>(P, 0, n, tab_f, sb, lane);      /* lane r < sb: model s0 + r */
                const unsigned cs = (unsigned)__shfl((int)cq, (lane - s0) & 63, 64);
                if (in) C1 = cs;
                DG_WSYNC();
            }
            const bool keep = have && ((double)C1 > tauJ);
            if (have && !keep) { res_I[mi] = 0; res_J[mi] = 0; }
            surv = __ballot(keep);
        }
        if (use_bound && surv) {
            const bool mine = (surv >> lane) & 1ull;
            const int myrank = __popcll(surv & lt_mask), ns = __popcll(surv);
            n_l2 = (unsigned)ns;
            unsigned C2 = 0;
            for (int s0 = 0; s0 < ns; s0 += B2) {
                const int sb = ns - s0 < B2 ? ns - s0 : B2;
                const bool in = mine && myrank >= s0 && myrank < s0 + sb;
                if (in) {
                    double *e = tab_d + (myrank - s0) * DG_L2_ENTRY_DOUBLES;
#pragma unroll
                    for (int j = 0; j < 9; j++) e[j] = F[j];
                    e[9] = 0.;
                }
                DG_WSYNC();
                const unsigned cq = dg_l2_tile_counts<LDSPTS>(P, 0, n, tab_d, sb, kind, t94b, lane);   /* lane r < sb: survivor s0 + r */
                const unsigned cs = (unsigned)__shfl((int)cq, (myrank - s0) & 63, 64);
                if (in) C2 = cs;
                DG_WSYNC();
            }
            const bool keep = mine && ((double)C2 > tauJ);
            if (mine && !keep) { res_I[mi] = 0; res_J[mi] = 0; }
            surv = __ballot(keep);
        }
        if (lane == 0) {                 /* four LDS adds per batch of up to 64 models */
            __hip_atomic_fetch_add(&scnt[0], n_l1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&scnt[1], n_l2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&scnt[2], (unsigned)__popcll(surv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&scnt[3], n_all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        for (unsigned long long m = surv; m; m &= m - 1ull) {
            const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
            const int mie = ws + (j0 + l) * NS;
            double Fe[9];
#pragma unroll
            for (int j = 0; j < 9; j++) Fe[j] = dg_readlane_d(F[j], l);
            unsigned cI = 0, cnt = 0;
            for (int base = 0; base < n; base += 64 * DG_PU) {
                dg_pt qq[DG_PU]; double dd[DG_PU];
#pragma unroll
                for (int u = 0; u < DG_PU; u++) { const int p = base + 64 * u + lane; qq[u] = dg_ldpt<LDSPTS>(P, p < n ? p : 0); }
#pragma unroll
                for (int u = 0; u < DG_PU; u++) dd[u] = dg_Ferr(kind, Fe, qq[u]);
#pragma unroll
                for (int u = 0; u < DG_PU; u++) {
                    const bool act = base + 64 * u + lane < n; const double d = dd[u];
                    double term = 0.0; if (act && th != 0 && !(d >= t94)) term = 1 - (d / t94);
                    cI += (act && d <= th) ? 1u : 0u;
                    const bool nz = !(term == 0.0);
                    const unsigned long long bJ = __ballot(nz);
                    if (nz) ((__attribute__((address_space(1))) double *)jbuf)[cnt + (unsigned)__popcll(bJ & lt_mask)] = term;
                    cnt += (unsigned)__popcll(bJ);
                }
            }
            DG_WSYNC();
            double J = 0.0; if (lane == 0) J = dg_seq_sum(jbuf, (int)cnt);
            J = __shfl(J, 0, 64);
            const unsigned I = dg_wave_sum_u(cI);
            DG_WSYNC();
            if (lane == 0) { res_I[mie] = I; res_J[mie] = J; }
        }
    }
}

/* Sampler stage 2, parallel form (pool in LDS, n < 65536).  The chunk's cn * NDRAW swaps vp[s] <-> vp[n-1-i] are a
 * chain only through the positions they share.  Every swap touches two positions; touch u = 2 tau + side (side 0: the
 * drawn slot s, side 1: the tail slot).  R(u) = value of that position before its swap.  Phase A walks the touches in
 * order, 64 per LDS atomic exchange, leaving "last toucher + 1" in the upper half-word of the pool entry (ids < 2^16):
 * lanes of one ds_wrxchg that hit the same address are served in ascending lane order on gfx950 (tools/
 * gpu_atomic_order.py: 0 violations in 1.4 M), so the returned marker IS the predecessor touch v, and R(u) = R(v ^ 1)
 * (the other side of the predecessor's swap; a swap with s == tail slot hands its own value over); a zero marker
 * means first touch: R(u) = the id stored there.  Phase B resolves the pointers by jumping (chains are a few hops:
 * tail slot -> previous sample's draw -> ...), phase C emits id(tau) = R(2 tau), phase D lets the last toucher of every
 * position store the value its swap left there.  ~12 us per 256-sample chunk instead of ~59 us sequential. */
template <int NDRAW>
__device__ __noinline__ void dg_sample_pool_par(int cn, int n, int *vp_generic, int (*draws)[8], int *ptr /* LDS, 2*cn*NDRAW ints */, int lane, long long *dbg)
{
    long long ts2 = DG_CLK();
    __builtin_amdgcn_s_setprio(3);
    __attribute__((address_space(3))) int *vp = (__attribute__((address_space(3))) int *)vp_generic;
    const int M2 = 2 * cn * NDRAW;
    /* A: predecessor of every touch */
    for (int u0 = 0; u0 < M2; u0 += 64 * 4) {
        int oldv[4], posv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int u = u0 + 64 * q + lane;
            if (u0 + 64 * q >= M2) { oldv[q] = 0; posv[q] = 0; continue; }
            const bool on = u < M2;
            const int tau = u >> 1, k = tau / NDRAW, i = tau - k * NDRAW;
            const int pos = on ? ((u & 1) ? n - 1 - i : draws[k][i]) : 0;
            posv[q] = pos;
            oldv[q] = on ? __hip_atomic_exchange(vp + pos, (u + 1) << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int u = u0 + 64 * q + lane;
            if (u >= M2) continue;
            const int old = oldv[q], m = (int)((unsigned)old >> 16);
            int pv;
            if (m == 0) pv = -1 - (old & 0xffff);
            else { const int v = m - 1; pv = ((v >> 1) == (u >> 1)) ? u - 1 : (v ^ 1); }
            ptr[u] = pv;
        }
    }
    DG_WSYNC();
    /* B: pointer jumping until every touch holds a value (negative = -1 - id); two hops per round, 8 touches per lane in
     * flight */
    for (;;) {
        bool any = false;
        for (int u0 = lane; u0 < M2; u0 += 64 * 8) {
            int pv[8], qv[8];
#pragma unroll
            for (int q = 0; q < 8; q++) { const int u = u0 + 64 * q; pv[q] = u < M2 ? ptr[u] : -1; }
#pragma unroll
            for (int q = 0; q < 8; q++) qv[q] = pv[q] >= 0 ? ptr[pv[q]] : -1;
#pragma unroll
            for (int q = 0; q < 8; q++) if (pv[q] >= 0 && qv[q] >= 0) qv[q] = ptr[qv[q]];
#pragma unroll
            for (int q = 0; q < 8; q++) { const int u = u0 + 64 * q; if (pv[q] >= 0) { ptr[u] = qv[q]; any = any || qv[q] >= 0; } }
        }
        DG_WSYNC();
        if (!__ballot(any)) break;
    }
    /* D1: which touches are the last on their position (reads the markers; writes come after a wave barrier) */
    unsigned long long lastm = 0;
    for (int u0 = lane, s0 = 0; u0 < M2; u0 += 64 * 8, s0 += 8) {
        int pos[8], mk[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int u = u0 + 64 * q; const int tau = u >> 1, k = tau / NDRAW, i = tau - k * NDRAW;
            pos[q] = u < M2 ? ((u & 1) ? n - 1 - i : draws[k][i]) : 0;
        }
#pragma unroll
        for (int q = 0; q < 8; q++) mk[q] = vp[pos[q]];
#pragma unroll
        for (int q = 0; q < 8; q++) { const int u = u0 + 64 * q; if (u < M2 && (int)((unsigned)mk[q] >> 16) == u + 1) lastm |= 1ull << (s0 + q); }
    }
    DG_WSYNC();
    /* D2: they store what their swap left there = the value the other side held before it */
    for (int u0 = lane, s0 = 0; u0 < M2; u0 += 64 * 8, s0 += 8) {
        int pos[8], val[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int u = u0 + 64 * q; const int tau = u >> 1, k = tau / NDRAW, i = tau - k * NDRAW;
            const bool on = u < M2 && ((lastm >> (s0 + q)) & 1ull);
            pos[q] = on ? ((u & 1) ? n - 1 - i : draws[k][i]) : -1;
            val[q] = on ? ptr[u ^ 1] : 0;
        }
#pragma unroll
        for (int q = 0; q < 8; q++) if (pos[q] >= 0) vp[pos[q]] = -1 - val[q];
    }
    DG_WSYNC();
    /* C: the drawn ids replace the raw draws */
    for (int t0 = lane; t0 < cn * NDRAW; t0 += 64 * 8) {
        int val[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { const int tau = t0 + 64 * q; val[q] = tau < cn * NDRAW ? ptr[2 * tau] : 0; }
#pragma unroll
        for (int q = 0; q < 8; q++) { const int tau = t0 + 64 * q; if (tau < cn * NDRAW) { const int k = tau / NDRAW, i = tau - k * NDRAW; draws[k][i] = -1 - val[q]; } }
    }
    DG_WSYNC();
    __builtin_amdgcn_s_setprio(0);
    DG_DEVT(if (dbg && lane == 0) { long long ts3 = DG_CLK(); dbg[6] += ts3 - ts2; });
}

/* Sampler stage 2 for a pool in the HBM workspace (placement HBM: n too large for LDS, or many small workgroups per CU).
 * The sequential form above pays one memory round trip per SAMPLE (each swap reads a drawn slot and writes it back; only
 * the tail slots live in registers).  Here the samples are taken in GROUPS of G = LANES / NDRAW, lane = (sample, draw):
 * when no sample of the group is alias-flagged and no two lanes of the group hold the same drawn position, the swaps of
 * the group touch pairwise distinct slots besides the tail slots, which form NDRAW independent chains
 *     id(k, i) = pool[s(k, i)]        pool[s(k, i)] <- tail_i before sample k = id(k - 1, i)        tail_i <- id(k, i)
 * so the group is ONE gather, lane shuffles and ONE scatter: one round trip per G samples.  A group with a collision
 * (LANES is chosen so that LANES^2 / 2n is at most ~0.1; found with two small LDS hash tables) or an alias-flagged sample is run by the sequential form
 * (dg_sample_pool_seq_range), whose result is the reference's by construction.  Same pool contents and drawn ids either way. */
#define DG_AS1(T) __attribute__((address_space(1))) T
#define DG_AS3(T) __attribute__((address_space(3))) T
/* draws and the alias mask live in LDS, the pool in global memory: qualified pointers, so that the accesses are ds_ /
 * global_ instructions and not flat ones (a flat access to LDS waits on both memory counters) */
template <int NDRAW>
__device__ __forceinline__ int dg_sample_pool_seq_range(int k_lo, int k_hi, int n, DG_AS1(int) *vp, DG_AS3(int) *draws /* [.][8] */,
                                                        const DG_AS3(unsigned long long) *almask, int t, int lane)
{
    const bool act = lane < NDRAW;
    for (int k = k_lo; k < k_hi; k++) {
        if ((almask[k >> 6] >> (k & 63)) & 1ull) {
            /* order-dependent sample: replay it sequentially on lane 0 */
            if (act) vp[n - 1 - lane] = t;
            __threadfence_block();
            DG_WSYNC();
            if (lane == 0) {
                for (int i = 0; i < NDRAW; i++) { int si = draws[8 * k + i], j = n - 1 - i, q = vp[si]; vp[si] = vp[j]; vp[j] = q; draws[8 * k + i] = q; }
            }
            __threadfence_block();
            DG_WSYNC();
            if (act) t = vp[n - 1 - lane];
        } else if (act) {
            const int s0 = draws[8 * k + lane];
            const int r0 = vp[s0];
            vp[s0] = t; t = r0; draws[8 * k + lane] = r0;
        }
        __threadfence_block();
    }
    return t;
}

#define DG_PGT (DG_JBUF_LDS_BYTES >= 8192 ? 1024 : 512)   /* slots per collision table; the two tables live in the pool stage's LDS scratch */
static_assert(2 * DG_PGT * sizeof(int) <= DG_JBUF_LDS_BYTES, "collision tables do not fit the pool-stage scratch");
template <int NDRAW>
__device__ __noinline__ void dg_sample_pool_grp(int cn, int n, int *vp_, int (*draws_)[8], const unsigned long long *almask_, int *pscratch /* LDS, 2 * DG_PGT ints */,
                                                int lane, long long *dbg)
{
    long long ts2 = DG_CLK();
    __builtin_amdgcn_s_setprio(3);
    DG_AS3(unsigned) *tab = (DG_AS3(unsigned) *)(unsigned *)pscratch;
    for (int q = lane; q < 2 * DG_PGT; q += 64) tab[q] = 0u;
    DG_WSYNC();
    DG_AS1(int) *vp = (DG_AS1(int) *)vp_;
    DG_AS3(int) *draws = (DG_AS3(int) *)(int *)draws_;
    const DG_AS3(unsigned long long) *almask_in = (const DG_AS3(unsigned long long) *)almask_;
    cn = __builtin_amdgcn_readfirstlane(cn); n = __builtin_amdgcn_readfirstlane(n);
    /* lanes per group: 64, 32 or 16, the largest with LANES^2 <= n / 5 (collision probability ~ LANES^2 / 2n <= 0.1) */
    const int LANES = (long long)64 * 64 * 5 <= n ? 64 : ((long long)32 * 32 * 5 <= n ? 32 : 16);
    const int G = LANES / NDRAW;
    int t = lane < NDRAW ? vp[n - 1 - lane] : 0;                 /* tail slot i lives in lane i */
    const int j = lane / NDRAW, i = lane - j * NDRAW;
    for (int k0 = 0; k0 < cn; k0 += G) {
        const int g = cn - k0 < G ? cn - k0 : G;
        const bool active = j < g;
        /* any alias-flagged sample in [k0, k0 + g)?  (g <= 16 flag bits starting at bit k0: at most two words of the mask) */
        bool al;
        {
            const int w = k0 >> 6, b = k0 & 63;
            unsigned long long win = almask_in[w] >> b;
            if (b && (w + 1) * 64 < cn) win |= almask_in[w + 1] << (64 - b);
            al = (win & ((1ull << g) - 1ull)) != 0ull;
        }
        const int s = active ? draws[8 * (k0 + j) + i] : -1 - lane;
        const int r = active ? vp[s] : 0;                         /* one gather for the whole group (used when nothing collides) */
        /* Do two lanes hold the same position?  Two LDS tables of DG_PGT slots, each slot = max over the lanes that hash to
         * it of (position << 6 | lane) (LDS atomic max).  A lane that finds its own position in its slot knows the answer
         * exactly (a duplicate iff the lane part is not its own: the lower lane of a duplicate pair always sees the higher
         * one); a lane whose slot shows a larger foreign position in both tables cannot tell and reports a collision
         * (conservative: the group then takes the sequential form; ~1 group in 50).  The slots are cleared afterwards. */
        bool coll = false;
        if (!al) {
            const unsigned key = ((unsigned)s << 6) | (unsigned)lane;
            const unsigned h1 = (unsigned)s % DG_PGT, h2 = ((unsigned)s * 40503u >> 7) % DG_PGT;
            if (active) { __hip_atomic_fetch_max(tab + h1, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_fetch_max(tab + DG_PGT + h2, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            DG_WSYNC();
            if (active) {
                const unsigned e1 = tab[h1], e2 = tab[DG_PGT + h2];
                if ((e1 >> 6) == (unsigned)s) coll = (e1 & 63u) != (unsigned)lane;
                else if ((e2 >> 6) == (unsigned)s) coll = (e2 & 63u) != (unsigned)lane;
                else coll = true;
            }
            DG_WSYNC();
            if (active) { tab[h1] = 0u; tab[DG_PGT + h2] = 0u; }
            DG_WSYNC();
        }
        if (al || __ballot(active && coll) != 0ull) {
            t = dg_sample_pool_seq_range<NDRAW>(k0, k0 + g, n, vp, draws, almask_in, t, lane);
            continue;
        }
        const int prev = __shfl(r, lane >= NDRAW ? lane - NDRAW : 0, 64), carry = __shfl(t, i, 64);
        if (active) { vp[s] = j == 0 ? carry : prev; draws[8 * (k0 + j) + i] = r; }      /* one scatter */
        const int tn = __shfl(r, (g - 1) * NDRAW + (lane < NDRAW ? lane : 0), 64);
        if (lane < NDRAW) t = tn;
        __threadfence_block();
    }
    if (lane < NDRAW) vp[n - 1 - lane] = t;
    __threadfence_block();
    DG_WSYNC();
    __builtin_amdgcn_s_setprio(0);
    DG_DEVT(if (dbg && lane == 0) { long long ts3 = DG_CLK(); dbg[6] += ts3 - ts2; });
}

/* stage 2 dispatch: the parallel form needs the pool in LDS with 16-bit ids and 2*cn*NDRAW ints of LDS scratch */
template <int NDRAW, int LDSPTS>
__device__ __forceinline__ void dg_sample_pool(int cn, int n, int *pool, int (*draws)[8], const unsigned long long *almask, int *pscratch /* LDS or 0 */,
                                               int lane, long long *dbg = 0)
{
    if (LDSPTS != 0 && pscratch && n < 65536) dg_sample_pool_par<NDRAW>(cn, n, pool, draws, pscratch, lane, dbg);
    else if (LDSPTS == 0 && pscratch) dg_sample_pool_grp<NDRAW>(cn, n, pool, draws, almask, pscratch, lane, dbg);
    else dg_sample_pool_seq<NDRAW, LDSPTS>(cn, n, pool, draws, almask, lane, dbg);
}

/* both stages back to back on one wave (prologue of the main kernels, unit-test kernel) */
template <int NDRAW, int LDSPTS>
__device__ __forceinline__ unsigned dg_sample_chunk(unsigned seed, int cn, int n, int *pool, unsigned *seeds, int (*draws)[8],
                                                    unsigned long long *almask, int *pscratch, int lane)
{
    unsigned sd = dg_sample_draws<NDRAW>(seed, cn, n, seeds, draws, almask, lane);
    dg_sample_pool<NDRAW, LDSPTS>(cn, n, pool, draws, almask, pscratch, lane);
    return sd;
}

/* ---- setting a long pair aside (dg_args::park_sam) -------------------------------------------------------------
 * Cross-workgroup hand-off as in the cooperative mode: plain payload, then ONE agent-scope release by wave 0 after the
 * workgroup barrier, then the flag (the queue entry) with a relaxed agent-scope atomic; the taker polls the entry with
 * its whole first wave behind a scalar branch, then acquires. */
/* legacy drivers' symmetric check: remember that the reference's local `f` holds model M (LDS) after sample no_sam */
#define DG_FLAST(M) do { if (legacy_sym) { __syncthreads(); if (tid < 9) S->flast[tid] = (M)[tid]; D.flast_k = no_sam; __syncthreads(); } } while (0)
#define DG_PARK_SPARE   0
#define DG_PARK_CLAIMED 32                   /* queue q: claimed count at DG_PARK_CLAIMED + 64 q, taken count at DG_PARK_HEAD + 64 q */
#define DG_PARK_HEAD    64                   /* (one 128-byte line each; q = 0: pairs with few samples left, q = 1: many) */
#define DG_PARK_DYN_OFF ((sizeof(dg_f_shared) + 255) & ~(size_t)255)   /* the dynamic LDS follows the dg_f_shared image */

/* copies between LDS and the workspace, 16 bytes per thread and step (both sides 16-byte aligned) */
__device__ __forceinline__ void dg_copy16(void *dst, const void *src, size_t bytes, int tid)
{
    const size_t nv = bytes / 16;
    const uint4 *s4 = (const uint4 *)src; uint4 *d4 = (uint4 *)dst;
    for (size_t i = tid; i < nv; i += DG_T) d4[i] = s4[i];
    const size_t done = nv * 16;
    for (size_t i = done + tid; i < bytes; i += DG_T) ((unsigned char *)dst)[i] = ((const unsigned char *)src)[i];
}

/* the next pair of queue `q` (pair << 32 | workspace), or -1 when that queue is empty.
 * Entries are taken with compare-and-swap, never past the claimed count: a workgroup that sets a pair aside goes round
 * its loop again (long queue, tickets, short queue), so it finds its own entry if nobody else has taken it, and no entry
 * is left behind. */
__device__ __forceinline__ long long dg_park_take(const dg_args &A, long long *bc /* LDS */, const int q)
{
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0) {
        int *const p_head = A.park_ctl + DG_PARK_HEAD + 64 * q, *const p_cl = A.park_ctl + DG_PARK_CLAIMED + 64 * q;
        const long long *const pq = A.park_q + (size_t)q * A.park_cap;
        int h;
        for (;;) {
            h = __builtin_amdgcn_readfirstlane(__hip_atomic_load(p_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const int cl = __builtin_amdgcn_readfirstlane(__hip_atomic_load(p_cl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (h >= cl) { h = -1; break; }
            int ok = 0;
            if (threadIdx.x == 0) {
                int expect = h;
                ok = __hip_atomic_compare_exchange_strong(p_head, &expect, h + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
            }
            if (__builtin_amdgcn_readfirstlane(ok)) break;
        }
        long long e = -1;
        if (h >= 0) {
            for (;;) {
                const long long v = __hip_atomic_load(pq + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
                e = ((long long)hi << 32) | (unsigned)lo;
                if (e >= 0) break;
                __builtin_amdgcn_s_sleep(4);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        *bc = e;                                                             /* every lane stores the same value */
    }
    __syncthreads();
    return *bc;
}

/* ---- cooperative large-n mode (dg_coop_cb, dg_kernel_common.h) -----------------------------------------------------
 * Views of the owner's workspace that every claiming workgroup (the owner itself or one of its helpers) needs. */
struct dg_coop_ws {
    const dg_pt *P; const double *gmodels; const unsigned short *gms;
    unsigned *cnt; unsigned short *surv; unsigned *res_I; double *res_J;
    dg_coop_job *job; dg_coop_rec *rec; int *stg_list, *stg_list2; double *stg_j;
};
__device__ __forceinline__ dg_coop_ws dg_coop_views(const dg_args &A, int slot)
{
    char *ws = A.ws + (size_t)slot * A.wl.stride;
    dg_coop_ws v;
    v.P = (const dg_pt *)(ws + A.wl.off_pts);
    v.gmodels = (const double *)(ws + A.wl.off_models);
    v.gms = (const unsigned short *)(ws + A.wl.off_mslot);
    v.cnt = (unsigned *)(ws + A.wl.off_mslot + (size_t)3 * DG_CHUNK * sizeof(unsigned short));
    v.surv = (unsigned short *)(v.cnt + 3 * DG_CHUNK);
    v.res_J = (double *)(ws + A.wl.off_res); v.res_I = (unsigned *)(v.res_J + 3 * DG_CHUNK);
    v.job = (dg_coop_job *)(ws + A.wl.off_job);
    v.rec = (dg_coop_rec *)(ws + A.wl.off_job + ((sizeof(dg_coop_job) + 255) & ~(size_t)255));
    v.stg_list = (int *)((char *)v.rec + ((DG_COOP_MAX_SLICES * sizeof(dg_coop_rec) + 255) & ~(size_t)255));
    v.stg_list2 = v.stg_list + A.wl.n_max; v.stg_j = (double *)(v.stg_list2 + A.wl.n_max);
    return v;
}
#define DG_COOP_GEN_MASK 0xfffff
#ifndef DG_COOP_SPW
#define DG_COOP_SPW 1            /* stage 1: point slices per claiming workgroup (C5: 85.4 ms with 3, 82.8 with 2, 80.4 with 1: a unit's claim and its release cost ~3 us) */
#endif
/* stage 4 (repetitions of a local optimisation as units): the job header + records (DG_LOJOB_BYTES behind the stage-3 staging),
 * and list k (0 .. 4 DG_RAN_REP - 1; repetition q: `inliers` = list 2q, the second list = 2q + 1, the slice-local staging of its
 * passes = lists 2 DG_RAN_REP + 2q and + 2q + 1) from the per-wave area */
__device__ __forceinline__ char *dg_coop_lojob(const dg_args &A, int slot)
{
    char *ws = A.ws + (size_t)slot * A.wl.stride;
    return ws + A.wl.off_job + ((sizeof(dg_coop_job) + 255) & ~(size_t)255) + ((DG_COOP_MAX_SLICES * sizeof(dg_coop_rec) + 255) & ~(size_t)255)
              + (((size_t)A.wl.n_max * (2 * sizeof(int) + sizeof(double)) + 255) & ~(size_t)255);
}
__device__ __forceinline__ int *dg_coop_lo_list(const dg_args &A, int slot, int k)
{
    return (int *)(A.ws + (size_t)slot * A.wl.stride + A.wl.off_wave) + (size_t)k * A.wl.n_max;
}

/* Owner, whole workgroup: publish a stage of `n_units` units (<= 4095).  Parameters first (plain), the claim counter and
 * the unit count with agent-scope atomics, one release, then the generation. */
__device__ __forceinline__ void dg_coop_publish(dg_coop_cb *cb, int &coop_gen, int stage, int n_units, int Mtot, int n, int kind, int slice, int use_l1,
                                                double th, const double *ext /* LDS */, double tau)
{
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0) {
        if (threadIdx.x == 0) {
            cb->stage = stage; cb->n_units = n_units; cb->Mtot = Mtot; cb->n = n; cb->kind = kind; cb->slice = slice; cb->use_l1 = use_l1; cb->th = th;
            for (int i = 0; i < 4; i++) cb->ext[i] = ext[i];
            /* the device-wide best-score bound only rises while a pair runs (atomic max on the ordered bits of a double >= 0) */
            __hip_atomic_fetch_max(&cb->tau_bits, (unsigned long long)__double_as_longlong(tau < 0 ? 0.0 : tau), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&cb->done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&cb->next, (int)((((unsigned)(coop_gen + 1) & DG_COOP_GEN_MASK) << 12) | (unsigned)n_units), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) __hip_atomic_store(&cb->gen, coop_gen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    coop_gen++;
    __syncthreads();
}

/* Whole workgroup: claim a unit of generation G.  Returns its index, or -1 when that stage has no unclaimed unit left
 * (or is already over).  The first wave does the compare-and-swap behind a scalar branch and broadcasts through LDS. */
__device__ __forceinline__ int dg_coop_claim(dg_coop_cb *cb, int G, int *bc /* LDS */)
{
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0) {
        int res = -1;
        for (;;) {
            const int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&cb->next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const unsigned d = (((unsigned)v >> 12) - (unsigned)G) & DG_COOP_GEN_MASK;
            if (d == 0) {
                const int rem = v & 0xfff;
                if (rem == 0) break;
                int ok = 0;
                if (threadIdx.x == 0) { int e = v; ok = __hip_atomic_compare_exchange_strong(&cb->next, &e, v - 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0; }
                if (__builtin_amdgcn_readfirstlane(ok)) { res = rem - 1; break; }
            } else if (d < (DG_COOP_GEN_MASK + 1) / 2) break;             /* a newer stage is up: this one is over */
            else __builtin_amdgcn_s_sleep(1);                            /* the counter still carries an older tag: not visible yet */
        }
        *bc = res;                                                       /* every lane stores the same value */
    }
    __syncthreads();
    return *bc;
}

/* Whole workgroup: one unit is finished (its results are in global memory) */
__device__ __forceinline__ void dg_coop_unit_done(dg_coop_cb *cb)
{
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&cb->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

/* Stage 1, one unit = the point slice [lo, hi): every wave of the workgroup takes the models w, w + DG_NW, ... in batches
 * that fit its LDS table, counts them tile-major over the slice (level 1 when the owner asked for it and tau >= 64, else
 * level 2) and adds the counts to the per-model device counters.  A model whose device counter already exceeds the
 * device-wide bound needs no more counting (it will be scored exactly whatever this slice adds). */
template <int T>
__device__ __forceinline__ void dg_coop_unit_screen(dg_f_shared *S, const dg_coop_ws &v, const dg_coop_cb *cb, int lo, int hi, double tau, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    const int Mtot = cb->Mtot, kind = cb->kind;
    const double th = cb->th, t94b = th * 9 / 4 * (1.0 + 1e-6);
    const bool l1 = cb->use_l1 && tau >= 64.0;
    const int capw = (int)((sizeof(dg_lsq_scratch) / DG_NW) & ~(size_t)15);
    char *tab = (char *)&S->lsq + (size_t)wave * capw;
    int B = l1 ? capw / (DG_L1_ENTRY_FLOATS * (int)sizeof(float)) : capw / (DG_L2_ENTRY_DOUBLES * (int)sizeof(double));
    B = B > 64 ? 64 : B;
    const int nm = Mtot > wave ? (Mtot - wave + DG_NW - 1) / DG_NW : 0;
    for (int j0 = 0; j0 < nm; j0 += B) {
        const int nb = nm - j0 < B ? nm - j0 : B;
        const bool have = lane < nb;
        const int mi = wave + (j0 + lane) * DG_NW;
        bool need = have;
        if (have) need = !((double)__hip_atomic_load(v.cnt + mi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > tau);
        /* all lanes write an entry (the tile loop runs over nb entries); models that need no counting get a zero model */
        double F[9];
        {
            const double *gp = v.gmodels + (size_t)cb->mtab * (DG_MTAB_BYTES / sizeof(double)) + (size_t)v.gms[have ? mi : wave] * 9;
#pragma unroll
            for (int q = 0; q < 9; q++) F[q] = gp[q];
        }
        if (have) {
            if (l1) {
                float Ff[9]; const float thr = dg_l1_setup(kind, F, S->ext, t94b, Ff);
                float *e = (float *)tab + lane * DG_L1_ENTRY_FLOATS;
#pragma unroll
                for (int q = 0; q < 9; q++) e[q] = Ff[q];
                e[9] = thr; e[10] = 0.f; e[11] = 0.f;
            } else {
                double *e = (double *)tab + lane * DG_L2_ENTRY_DOUBLES;
#pragma unroll
                for (int q = 0; q < 9; q++) e[q] = F[q];
                e[9] = 0.;
            }
        }
        DG_WSYNC();
        unsigned cq = 0;
        if (__ballot(need) != 0ull)
            cq = l1 ? dg_l1_tile_counts<0>(v.P, lo, hi, (const float *)tab, nb, lane) : dg_l2_tile_counts<0>(v.P, lo, hi, (const double *)tab, nb, kind, t94b, lane);
        if (have && need && cq) __hip_atomic_fetch_add(v.cnt + mi, cq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        DG_WSYNC();
    }
}

/* Stage 2, one unit = one model, scored exactly by the whole workgroup: I and the reference-order J through dg_pass */
template <int T>
__device__ __forceinline__ void dg_coop_unit_exact(dg_f_shared *S, const dg_coop_ws &v, const dg_coop_cb *cb, int mi, double *jbuf, int tid)
{
    const int n = cb->n, kind = cb->kind; const double th = cb->th;
    double F[9];
    const double *gp = v.gmodels + (size_t)cb->mtab * (DG_MTAB_BYTES / sizeof(double)) + (size_t)v.gms[mi] * 9;
#pragma unroll
    for (int q = 0; q < 9; q++) F[q] = gp[q];
    dg_pass_cfg cfg = dg_cfg0(n); cfg.wantJ = 1; cfg.thJ = th; cfg.jbuf = jbuf;
    const dg_pt *P = v.P;
    dg_pass_res r = dg_pass(&S->red, cfg, [&](int pid, int) { return dg_Ferr(kind, F, dg_ldpt<0>(P, pid)); }, tid);
    if (tid == 0) { v.res_I[mi] = r.I; v.res_J[mi] = r.J; }
}

/* Stage 3, one unit = slice u of a distributed pass (dg_coop_job): the ordinary workgroup pass on the slice, outputs in
 * slice-local staging, counts in rec[u] */
template <int T>
__device__ __forceinline__ void dg_coop_unit_pass(dg_f_shared *S, const dg_coop_ws &v, int u, int tid)
{
    const dg_coop_job *jb = v.job;
    const int n = jb->n, slice = jb->slice, lo = u * slice, hi = lo + slice < n ? lo + slice : n, kind = jb->kind;
    double F[9];
#pragma unroll
    for (int q = 0; q < 9; q++) F[q] = jb->F[q];
    dg_pass_cfg cfg = dg_cfg0(hi - lo); cfg.p0 = lo;
    if (jb->wantJ) { cfg.wantJ = 2; cfg.thJ = jb->thJ; cfg.jbuf = v.stg_j + lo; }
    if (jb->has_list) { cfg.list = v.stg_list + lo; cfg.thL = jb->thL; cfg.listStrict = jb->listStrict; }
    if (jb->has_list2) { cfg.list2 = v.stg_list2 + lo; cfg.thL2 = jb->thL2; }
    const dg_pt *P = v.P;
    const dg_pass_res r = dg_pass(&S->red, cfg, [&](int pid, int) { return dg_Ferr(kind, F, dg_ldpt<0>(P, pid)); }, tid);
    if (tid == 0) { dg_coop_rec rc; rc.I = r.I; rc.nL = r.nL; rc.nL2 = r.nL2; rc.nJ = r.nJ; v.rec[u] = rc; }
}

/* Stage 4, one unit = repetition u of the current round of a local optimisation (exp_ranF.c:621-743 behind the sample of
 * exp_ranF.c:771), by the whole claiming workgroup: the same fits, passes, hashes and record as dg_lo_rep_wave, with
 * workgroup passes over all n points (ordered MSAC terms in LDS + this workgroup's HBM buffer), the hash of a set on wave 1
 * while wave 0 draws and fits the next 8-subset.  The table is only looked up. */
template <int T>
__device__ __noinline__ void dg_lo_rep_wg(dg_f_shared *S, const dg_pt *P, const int n, const dg_ht &ht, dg_lo_log *lg, int *ib, int *alt, int *sA, int *sB, double *jbuf,
                                          const int ssiz, const double th, const int mk_full, const int mk_ex, const int tid)
{
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *f = S->f, *fl = S->fLO, *ftmp = S->ftmp, *px = S->lsq.px, *wts = S->lsq.part[0];
    const bool small_ids = n < 65536;
    /* a pass = one point slice per wave (no workgroup barrier inside, lists and MSAC terms compacted into slice-local staging), then
     * the lists copied to their places in slice order while one lane adds the terms slice after slice: the lists and the J of
     * dg_pass over all points (a workgroup pass of 50 000 points: ~100 us; this: ~35) */
    auto pass = [&](const double *Fm, int kind, int wantJ, double thJ, int *la, double thL, int *lb, double thL2) -> dg_pass_res {
        double F[9];
#pragma unroll
        for (int i = 0; i < 9; i++) F[i] = Fm[i];
        constexpr int NWt = T / 64;
        const int sl = (((n + NWt - 1) / NWt) + 63) & ~63;
        const int lo = wv * sl < n ? wv * sl : n, hi = lo + sl < n ? lo + sl : n;
        unsigned *wc = (unsigned *)S->lsq.svw;
        __syncthreads();
        const dg_pass_res r = dg_wpass_slice(P, lo, hi, [&](const dg_pt &q) { return dg_Ferr(kind, F, q); }, thJ, la ? sA + lo : (int *)0, thL, lb ? sB + lo : (int *)0, thL2,
                                             wantJ ? jbuf + lo : (double *)0, lane);
        if (lane == 0) { wc[4 * wv] = r.I; wc[4 * wv + 1] = r.nL; wc[4 * wv + 2] = r.nL2; wc[4 * wv + 3] = r.nJ; }
        __syncthreads();
        dg_pass_res out; out.I = 0; out.J = 0; out.C = 0; out.nL = 0; out.nF = 0; out.nL2 = 0; out.nJ = 0;
        unsigned offA = 0, offB = 0;
#pragma unroll
        for (int w = 0; w < NWt; w++) {
            if (w < wv) { offA += wc[4 * w + 1]; offB += wc[4 * w + 2]; }
            out.I += wc[4 * w]; out.nL += wc[4 * w + 1]; out.nL2 += wc[4 * w + 2]; out.nJ += wc[4 * w + 3];
        }
        if (la) for (int k = lane; k < (int)r.nL; k += 64) la[offA + k] = sA[lo + k];
        if (lb) for (int k = lane; k < (int)r.nL2; k += 64) lb[offB + k] = sB[lo + k];
        if (wantJ && tid == T - 64) {
            double J = 0.0;
            for (int w = 0; w < NWt; w++) { const int l_ = w * sl < n ? w * sl : n; J = dg_seq_sum_from<1>(jbuf + l_, (int)wc[4 * w + 3], J); }
            S->red.bc[0] = J;
        }
        __syncthreads();
        if (wantJ) out.J = S->red.bc[0];
        return out;
    };
    auto gather = [&](int id, int len) {
        if (lane < len) { const dg_pt q = dg_ldpt<0>(P, id); double *o = px + 4 * lane; o[0] = q.x1; o[1] = q.y1; o[2] = q.x2; o[3] = q.y2; }
    };
    __syncthreads();
    if (wv == 0) {
        gather(lane < ssiz ? lg->ids[lane] : 0, ssiz);
        DG_WSYNC();
        dg_u2f_small_w(&S->lsq, px, (const double *)0, ssiz, f, lane);
    }
    __syncthreads();
    const dg_pass_res r0 = pass(f, mk_full, 1, th, ib, th * DG_MWM, (int *)0, 0.0);
    unsigned mI = r0.I; double mJ = r0.J; int kind0 = mk_full, drawn = 0;
    if (tid == 0) { lg->I0 = (int)r0.I; lg->drew0 = 0; lg->nit = 0; lg->has_fin = 0; }
    if (mI < 8) { if (tid == 0) { lg->I = 0; lg->J = 0; lg->kind0 = mk_full; lg->pub = 0; } __syncthreads(); return; }
    if (wv == 0) {
        const int cnt = (int)r0.nL; int id;
        if (8 < cnt) { dg_randsubset_wave(&lg->g, ib, cnt, 8, lane, &id); if (lane == 0) lg->drew0 = 8; }
        else id = lane < cnt ? ib[lane] : 0;
        const int use = 8 < cnt ? 8 : cnt;
        DG_WSYNC();
        gather(id, use);
        DG_WSYNC();
        dg_u2f_small_w(&S->lsq, px, (const double *)0, use, fl, lane);
    }
    if ((int)r0.nL > 8) drawn += 8;
    __syncthreads();
    double ths = DG_TC * th; const double dth = (ths - th) / DG_ILSQ_ITERS;
    int ended = 0;
    for (int it = 0; it < DG_ILSQ_ITERS; it++) {
        const dg_pass_res r1 = pass(fl, mk_ex, 1, th, ib, th, alt, ths * DG_MWM);
        const int improve = mJ < r1.J;
        unsigned nL2 = r1.nL2;
        /* exp_ranF.c:687-696: after a rotation `d` is the OLD errs[0]: that list is taken on the residuals of the previous best */
        if (improve) { const dg_pass_res r2 = pass(f, kind0, 0, 0.0, alt, ths * DG_MWM, (int *)0, 0.0); nL2 = r2.nL; }
        const int fit = nL2 >= 8;
        __syncthreads();
        if (wv == 1 || (T == 64 && wv == 0)) {
            const unsigned hash = dg_hash_list(ib, (int)r1.I, small_ids);
            if (lane == 0) {
                lg->it[it].hash = hash; lg->it[it].I = (int)r1.I; lg->nit = it + 1;
                /* a set an EARLIER round or local optimisation inserted ends the repetition here whatever the others of this round do */
                S->itmp[0] = dg_ht_contains(ht, hash, (int)r1.I, -1) != -1;
            }
        }
        if (wv == 0) {
            const int cnt = (int)nL2; int id = 0;
            if (lane == 0) lg->it[it].drew = (fit && 8 < cnt) ? 8 : 0;
            if (fit) {
                if (8 < cnt) dg_randsubset_wave(&lg->g, alt, cnt, 8, lane, &id);
                else id = lane < cnt ? alt[lane] : 0;
                const int use = 8 < cnt ? 8 : cnt;
                DG_WSYNC();
                if (lane < use) {
                    const dg_pt q = dg_ldpt<0>(P, id);
                    double *o = px + 4 * lane; o[0] = q.x1; o[1] = q.y1; o[2] = q.x2; o[3] = q.y2;
                    if (mk_ex == DG_K_FDS) wts[lane] = dg_exFDs_w(fl, q.x1, q.y1, q.x2, q.y2);
                    else { double ww_; dg_exFDsSym(fl, q.x1, q.y1, q.x2, q.y2, &ww_); wts[lane] = ww_; }
                }
                DG_WSYNC();
                dg_u2f_small_w(&S->lsq, px, wts, use, ftmp, lane);
            }
        }
        if (fit && nL2 > 8) drawn += 8;
        __syncthreads();
        if (S->itmp[0]) { ended = 2; break; }
        if (improve) { mI = r1.I; mJ = r1.J; kind0 = mk_ex; if (tid < 9) f[tid] = fl[tid]; }
        /* the reference builds this list (and shuffles it) in `inliers` itself */
        for (int j = tid; j < (int)nL2; j += T) ib[j] = alt[j];
        if (tid < 9 && fit) fl[tid] = ftmp[tid];
        __syncthreads();
        if (!fit) { ended = 1; break; }
        ths -= dth;
    }
    if (!ended) {
        const dg_pass_res r3 = pass(fl, mk_full, 1, th, ib, th, (int *)0, 0.0);
        if (tid == 0) lg->has_fin = 1;
        if (mJ < r3.J) { mI = r3.I; mJ = r3.J; kind0 = mk_full; __syncthreads(); if (tid < 9) f[tid] = fl[tid]; }
    }
    __syncthreads();
    if (tid < 9) lg->f[tid] = f[tid];
    if (tid == 0) { lg->I = (int)mI; lg->J = mJ; lg->kind0 = kind0; lg->pub = drawn; }
    __syncthreads();
}
template <int T>
__device__ __forceinline__ void dg_coop_unit_rep(const dg_args &A, dg_f_shared *S, const dg_coop_ws &v, int slot, int u, double *jbuf, int tid)
{
    char *lj = dg_coop_lojob(A, slot);
    const dg_lo_job *job = (const dg_lo_job *)lj;
    char *ws = A.ws + (size_t)slot * A.wl.stride;
    dg_ht ht; ht.heads = (int *)(ws + A.wl.off_ht); ht.count = ht.heads + 64; ht.ent = ht.heads + 80;
    dg_lo_rep_wg<T>(S, v.P, job->n, ht, (dg_lo_log *)(lj + 128 + (size_t)DG_LOJOB_STRIDE * u), dg_coop_lo_list(A, slot, 2 * u), dg_coop_lo_list(A, slot, 2 * u + 1),
                    dg_coop_lo_list(A, slot, 2 * DG_RAN_REP + 2 * u), dg_coop_lo_list(A, slot, 2 * DG_RAN_REP + 2 * u + 1), jbuf, job->ssiz, job->th, job->mk_full, job->mk_ex, tid);
}

/* Whole workgroup (owner or helper): work on generation G until it has no unclaimed unit left */
template <int T>
__device__ __forceinline__ void dg_coop_work(const dg_args &A, int slot, dg_f_shared *S, const dg_coop_ws &v, dg_coop_cb *cb, int G, double *jbuf, int *bc /* LDS */, int tid)
{
    for (;;) {
        const int u = dg_coop_claim(cb, G, bc);
        if (u < 0) break;
        /* the stage cannot end before this unit is done: its parameters are stable now */
        const int stage = cb->stage, n = cb->n, slice = cb->slice;
        if (tid < 4) S->ext[tid] = cb->ext[tid];
        __syncthreads();
        if (stage == 1) {
            const double tau = __longlong_as_double((long long)__hip_atomic_load(&cb->tau_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const int lo = u * slice, hi = lo + slice < n ? lo + slice : n;
            dg_coop_unit_screen<T>(S, v, cb, lo, hi, tau, tid);
        } else if (stage == 2) {
            dg_coop_unit_exact<T>(S, v, cb, (int)v.surv[u], jbuf, tid);
        } else if (stage == 3) {
            dg_coop_unit_pass<T>(S, v, u, tid);
        } else {
            dg_coop_unit_rep<T>(A, S, v, slot, u, jbuf, tid);
        }
        dg_coop_unit_done(cb);
    }
}

/* helper h (1..coop_k) of owner slot `slot`: follows the owner's stage generations until the owner retires the slot */
template <int T>
__device__ __forceinline__ void dg_f_helper(const dg_args &A, dg_f_shared *S, const int slot, const int h, int *bc /* LDS */)
{
    const int tid = threadIdx.x;
    const dg_coop_ws v = dg_coop_views(A, slot);
    double *jbuf = (double *)(A.ws + (size_t)slot * A.wl.stride + A.wl.off_hjbuf) + (size_t)h * A.wl.n_max;
    dg_coop_cb *cb = A.coop + slot;
    const bool wave0 = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;      /* scalar: a spin loop under a per-lane `if` inside a loop with
                                                                              workgroup barriers lets the compiler split the wave around them */
    int last = 0;
    for (;;) {
        if (wave0) {
            int g;
            while ((g = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&cb->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) == last) __builtin_amdgcn_s_sleep(4);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            *bc = g;                                                         /* every lane stores the same value */
        }
        __syncthreads();
        const int g = *bc;
        __syncthreads();
        if (g < 0) break;
        last = g;
        dg_coop_work<T>(A, slot, S, v, cb, g, jbuf, bc, tid);
    }
}

/* Owner, whole workgroup: one pass over the whole point set, distributed (stage 3).  Same result as dg_pass on all points:
 * the lists are the slices' lists in slice order, J the sequential sum of the slices' terms in slice order. */
template <int LDSPTS>
__device__ __noinline__ dg_pass_res dg_coop_pass(CTX &c, const double *Fm /* LDS */, int kind, const dg_pass_cfg &cfg)
{
    dg_f_shared *S = c.S; const dg_args &A = *c.A; dg_coop_cb *cb = c.cb;
    const int tid = c.tid, lane = tid & 63, wave = tid >> 6, n = cfg.n;
    const dg_coop_ws v = dg_coop_views(A, c.coop_slot);
    int slice = (n + A.coop_k) / (A.coop_k + 1);                               /* one slice per claiming workgroup ... */
    slice = (slice + DG_T * DG_PU - 1) / (DG_T * DG_PU) * (DG_T * DG_PU);      /* ... in whole steps of the workgroup pass */
    if ((n + slice - 1) / slice > DG_COOP_MAX_SLICES) slice = (n + DG_COOP_MAX_SLICES - 1) / DG_COOP_MAX_SLICES;
    const int n_units = (n + slice - 1) / slice;
    __syncthreads();
    if (tid == 0) {
        dg_coop_job *jb = v.job;
        for (int q = 0; q < 9; q++) jb->F[q] = Fm[q];
        jb->thJ = cfg.thJ; jb->thL = cfg.thL; jb->thL2 = cfg.thL2; jb->kind = kind; jb->wantJ = cfg.wantJ ? 1 : 0; jb->listStrict = cfg.listStrict;
        jb->has_list = cfg.list ? 1 : 0; jb->has_list2 = cfg.list2 ? 1 : 0; jb->slice = slice; jb->n = n; jb->pad = 0;
    }
    dg_coop_publish(cb, *c.coop_gen, 3, n_units, 0, n, kind, slice, 0, cfg.thJ, S->ext, 0.0);
    dg_coop_work<DG_T>(A, c.coop_slot, S, v, cb, *c.coop_gen, (double *)(A.ws + (size_t)c.coop_slot * A.wl.stride + A.wl.off_hjbuf), &S->itmp[28], tid);
    if (__builtin_amdgcn_readfirstlane(tid >> 6) == 0) {
        while (__hip_atomic_load(&cb->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_units) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    dg_pass_res out; out.I = 0; out.J = 0; out.C = 0; out.nL = 0; out.nF = 0; out.nL2 = 0; out.nJ = 0;
    /* concatenate: wave w copies the slices w, w + DG_NW, ...; the last wave's first lane meanwhile adds the terms in order */
    for (int u = 0; u < n_units; u++) {
        const dg_coop_rec rc = v.rec[u];
        if (u % DG_NW == wave) {
            const int lo = u * slice;
            if (cfg.list)  for (int k = lane; k < (int)rc.nL; k += 64)  cfg.list[out.nL + k] = v.stg_list[lo + k];
            if (cfg.list2) for (int k = lane; k < (int)rc.nL2; k += 64) cfg.list2[out.nL2 + k] = v.stg_list2[lo + k];
        }
        out.I += rc.I; out.nL += rc.nL; out.nL2 += rc.nL2; out.nJ += rc.nJ;
    }
    if (cfg.wantJ && tid == DG_T - 64) {
        double J = 0.0;
        for (int u = 0; u < n_units; u++) J = dg_seq_sum_from<1>(v.stg_j + (size_t)u * slice, (int)v.rec[u].nJ, J);
        S->red.bc[0] = J;
    }
    __syncthreads();
    if (cfg.wantJ) out.J = S->red.bc[0];
    __syncthreads();
    return out;
}

/* Owner, whole workgroup: one round of a local optimisation's repetitions as stage 4 (the records of the round's nr repetitions
 * are planned in the workspace); returns when all of them are finished and visible */
template <int LDSPTS>
__device__ __forceinline__ void dg_lo_round_coop(CTX &c, int nr, int ssiz, double th, int mk_full, int mk_ex)
{
    dg_f_shared *S = c.S; const dg_args &A = *c.A; dg_coop_cb *cb = c.cb; const int tid = c.tid;
    const dg_coop_ws v = dg_coop_views(A, c.coop_slot);
    if (tid == 0) { dg_lo_job *job = (dg_lo_job *)dg_coop_lojob(A, c.coop_slot); job->n = c.n; job->ssiz = ssiz; job->mk_full = mk_full; job->mk_ex = mk_ex; job->th = th; }
    dg_coop_publish(cb, *c.coop_gen, 4, nr, 0, c.n, mk_full, 0, 0, th, S->ext, 0.0);
    dg_coop_work<DG_T>(A, c.coop_slot, S, v, cb, *c.coop_gen, (double *)(A.ws + (size_t)c.coop_slot * A.wl.stride + A.wl.off_hjbuf), &S->itmp[28], tid);
    if (__builtin_amdgcn_readfirstlane(tid >> 6) == 0) {
        while (__hip_atomic_load(&cb->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nr) __builtin_amdgcn_s_sleep(8);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

/* ---- stream mode (dg_stream_cb, dg_stream_ent) -------------------------------------------------------------------- */
#define DG_STREAM_TIMEOUT 400000000ll         /* 4 s of the 100 MHz clock: a wait that long is a bug; flag it and go on instead of hanging */
__device__ __forceinline__ dg_stream_ent *dg_stream_entry(const dg_args &A, int oslot, int seq)
{
    return (dg_stream_ent *)(A.ring + ((size_t)oslot * A.stream_depth + (size_t)(seq % A.stream_depth)) * A.stream_ent_bytes);
}
/* whole workgroup: wait until *flag (agent-scope) satisfies `pred` or the owner's stop flag is up (stop = null: ignore); returns the
 * value seen (workgroup-uniform), -1 on timeout / stop.  The whole first wave polls behind a scalar branch. */
template <class Pred>
__device__ __forceinline__ int dg_stream_wait(const dg_args &A, int *flag, int *stop, Pred pred, int *bc /* LDS */, const long long limit = DG_STREAM_TIMEOUT)
{
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0) {
        const long long t0 = wall_clock64();
        int v;
        for (;;) {
            v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (pred(v) && limit != 0) break;                        /* limit 0 = fault injection (tests): every data wait fails at once */
            if (stop && __builtin_amdgcn_readfirstlane(__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { v = -1; break; }
            if (wall_clock64() - t0 > limit) { if (threadIdx.x == 0) __hip_atomic_store(A.err_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v = -1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *bc = v;                                                             /* every lane stores the same value */
    }
    __syncthreads();
    return *bc;
}
/* whole workgroup: the payload written so far becomes visible device-wide, then *flag = v */
__device__ __forceinline__ void dg_stream_publish(int *flag, int v)
{
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
/* A workgroup without a pair: the owner slot of a pair that asks for a producer (its request is taken), or -1 once every
 * pair of the launch is finished. */
__device__ __forceinline__ int dg_stream_find(const dg_args &A, int *bc /* LDS */)
{
    const int lane = (int)(threadIdx.x & 63);
    for (;;) {
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0) {
            int res = -2;
            if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(A.done_pairs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= A.n_pairs) res = -1;
            else if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(A.done_pairs + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) <= 0) {
                /* no open request (done_pairs[1] counts them): nothing to scan; hundreds of idle workgroups poll these two words only */
                for (int q = 0; q < 8; q++) __builtin_amdgcn_s_sleep(127);
            } else {
                /* the open request with the most samples left (pairs that have cut their budget end soon by themselves; the ones that
                 * keep all of it are the ones that end the launch): key = (samples left, slot); uniform trip counts throughout */
                long long key = -1;
                for (int q = 0; q < A.n_res; q += 64) {
                    const int j = (int)((blockIdx.x + (unsigned)(q + lane)) % (unsigned)A.n_res);
                    if (q + lane < A.n_res && __hip_atomic_load(&A.scb[j].state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == DG_ST_REQ) {
                        int left = __hip_atomic_load(&A.scb[j].max_sam, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                 - DG_CHUNK * __hip_atomic_load(&A.scb[j].tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (left < 0) left = 0;
                        /* ... and among those the pair that has been running longest: a pair that has just started also has its whole budget */
                        int age = (int)(wall_clock64() >> 10) - __hip_atomic_load(&A.scb[j].owner_sam, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        age = age < 0 ? 0 : (age > 0xfffff ? 0xfffff : age);
                        const long long k_ = ((long long)(left >> 12) << 40) | ((long long)age << 16) | (long long)(unsigned)(j & 0xffff);
                        key = k_ > key ? k_ : key;
                    }
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) { const long long t = __shfl_xor(key, o, 64); key = t > key ? t : key; }
                if (key >= 0) {
                    const int j = __builtin_amdgcn_readfirstlane((int)(key & 0xffffll));
                    int ok = 0;
                    if (threadIdx.x == 0) {
                        int e = DG_ST_REQ; ok = __hip_atomic_compare_exchange_strong(&A.scb[j].state, &e, DG_ST_ATTACHED, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
                        if (ok) __hip_atomic_fetch_add(A.done_pairs + 1, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (__builtin_amdgcn_readfirstlane(ok)) { res = j; __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
                } else __builtin_amdgcn_s_sleep(64);
            }
            *bc = res;
        }
        __syncthreads();
        const int r = *bc;
        if (r != -2) return r;
    }
}

/* the whole driver for ONE pair, run by one workgroup on workspace `wsid` (a resident workgroup starts on the workspace
 * of its own index).  resume != 0: `wsid` holds the image of a pair that was set aside, continue it.
 * Returns -1 when the pair is finished, else the pair was set aside and the return value is the spare workspace the
 * workgroup continues on. */
template <int T, int LDSPTS>
/* resume == 2: PRODUCER of the stream mode: continue the image in `wsid` (the owner's workspace) on this workgroup's own
 * workspace `own_wsid`, for the owner at slot `oslot`: sample, solve and score chunk after chunk into the owner's ring, never commit. */
__device__ __forceinline__ int dg_f_pair(const dg_args &A, dg_f_shared *S, unsigned char *dyn_smem, const int pair, const int slot, const int wsid,
                                         const int resume, int &coop_gen, const int own_wsid, const int oslot)
{
    const int producer = resume == 2;
    dg_stream_cb *const scb = A.stream_on ? A.scb + oslot : (dg_stream_cb *)0;
    int head_seen = 0;               /* owner: ring entries below this sequence number are known to be visible */
    int gpar = 0, pend_draws = 0;    /* deep pipeline: the seed buffer this iteration's chain writes; the chunk in slot nx2 still needs its draws (its seeds are in the other buffer) */
    int mtab = 0, presolved = 0;     /* cooperative mode: the model table of the current chunk; samples of the current chunk that were solved during the previous chunk's scoring */
    int strm = 0, img_sam = 0;       /* owner: 0 = own sample stream, 1 = asked for a producer (image written), 2 = takes its chunks from the ring */
    const int coopK = LDSPTS == 0 ? A.coop_k : 0;
    dg_coop_cb *const cb = coopK > 0 ? A.coop + slot : (dg_coop_cb *)0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    /* cooperative mode with eight waves: the sampler runs one chunk further ahead (pool swaps of chunk c + 2, seed chain of chunk c + 3),
     * so that the 7-point problems of chunk c + 1 can be solved from the start of the phase in which the helpers score chunk c */
    const bool deep = DG_NW >= 8 && LDSPTS == 0 && coopK > 0 && !A.hist_out;
    const long long off = A.offsets[pair];
    const int n = (int)(A.offsets[pair + 1] - off);
    const dg_params &pr = A.prm;
    const double th = pr.th;

    char *ws = A.ws + (size_t)wsid * A.wl.stride;
    char *const wsown = A.ws + (size_t)own_wsid * A.wl.stride;          /* = ws unless this workgroup is another pair's producer */
    CTX c;
    c.S = S; c.K = (const __attribute__((address_space(3))) dg_f_cshared *)&S->K; c.n = n; c.tid = tid; c.A = &A; c.off = off;
    __syncthreads();                                       /* nobody still reads the previous pair's views */
    if (tid == 0) dg_fill_views(&S->K, wsown, A.wl);
    __syncthreads();
    c.ht.heads = (int *)(ws + A.wl.off_ht); c.ht.count = c.ht.heads + 64; c.ht.ent = c.ht.heads + 80;
    c.seeds = S->seeds3[0]; c.draws = S->draws3[0];
    c.n_fds = c.n_exfds = c.n_hds = c.n_aux = 0; c.rrun = 0; c.hlt = (double *)0;
    c.cb = cb; c.coop_gen = &coop_gen; c.coop_slot = slot; c.lo_assumed = DG_LO_ASSUMED_DRAWS; c.lo_prev = -1;
    dg_pt *Pw; int *pool;
    /* LDSPTS: 1 = point set and sampler pool in LDS, 2 = pool in LDS / points in the HBM workspace (L2), 0 = both in HBM */
    if (LDSPTS == 1) { Pw = (dg_pt *)dyn_smem; pool = (int *)(dyn_smem + (size_t)n * sizeof(dg_pt)); }
    else             { Pw = (dg_pt *)(ws + A.wl.off_pts); pool = LDSPTS == 2 ? (int *)dyn_smem : (int *)(ws + A.wl.off_pool); }
    c.P = Pw; c.pool = pool;
    const dg_pt *P = Pw;
    int *const pscr = A.pool_seq ? (int *)0 : (int *)S->ww;     /* LDS scratch of the parallel pool stage; null selects the sequential one */

    const int mk_full = pr.error_type == 1 ? DG_K_FSYM : DG_K_FDS;
    const int mk_ex   = pr.error_type == 1 ? DG_K_EXFSYM : DG_K_FDS;
    const int doSym = pr.sym_th > 0, doLaf = pr.laf_coef > 0;
    const int legacy_sym = pr.legacy && doSym;         /* exp_ransacFcustom's symmetric check: its final filter needs the driver's last `f` */

    /* ---- driver state (workgroup-uniform, replicated in every lane) ---- */
    dg_f_drv D;
    dg_score &maxS = D.maxS, &maxSs = D.maxSs;
    int &no_sam = D.no_sam, &max_sam = D.max_sam, &iter_cnt = D.iter_cnt, &degen_cnt = D.degen_cnt, &iterID = D.iterID, &Ihmax = D.Ihmax;
    unsigned &non_degen = D.non_degen;
    int &best_sample = D.best_sample; long long &t_best = D.t_best, &t_start = D.t_start;
    int &finKind = D.finKind, &accepted = D.accepted;          /* errs[3] = residuals of S->F under finKind */
    int (&perm)[4] = D.perm, &p4 = D.p4, &e4kind = D.e4kind, &track = D.track;   /* errs[] pointer bookkeeping */
    double *e4F = S->bufF[0];                         /* model whose residuals errs[4] points at */
    int &done = D.done;
    unsigned &seed = D.seed;
    int &cur = D.cur, (&chunk_s)[3] = D.chunk_s, &chunk_base = D.chunk_base;
    int park_on = (A.park_sam > 0 && !resume) ? 1 : 0;

  if (!resume) {
    t_start = wall_clock64();
    /* ---- stage the correspondences (bindings.cpp:337-409: only x,y of each row are geometry) ---- */
    double ex0 = 0., ex1 = 0., ex2 = 0., ex3 = 0.;
    for (int i = tid; i < n; i += DG_T) {
        const double *a = A.pts1 + (size_t)(off + i) * A.dim, *b = A.pts2 + (size_t)(off + i) * A.dim;
        dg_pt p; p.x1 = a[0]; p.y1 = a[1]; p.x2 = b[0]; p.y2 = b[1];
        Pw[i] = p; pool[i] = i;
        ex0 = fmax(ex0, fabs(p.x1)); ex1 = fmax(ex1, fabs(p.y1)); ex2 = fmax(ex2, fabs(p.x2)); ex3 = fmax(ex3, fabs(p.y2));
    }
    /* coordinate extents of the pair (used only by the conservative screening bound of the scoring phase) */
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        ex0 = fmax(ex0, __shfl_xor(ex0, o, 64)); ex1 = fmax(ex1, __shfl_xor(ex1, o, 64));
        ex2 = fmax(ex2, __shfl_xor(ex2, o, 64)); ex3 = fmax(ex3, __shfl_xor(ex3, o, 64));
    }
    if (lane == 0) { S->extw[wave][0] = ex0; S->extw[wave][1] = ex1; S->extw[wave][2] = ex2; S->extw[wave][3] = ex3; }
    dg_ht_init(c.ht, tid);
    if (A.hist_out) for (int j = tid; j < n + 3; j += DG_T) A.hist_out[(size_t)off + 3 * (size_t)pair + j] = 0;
    if (tid < 9) { S->F[tid] = 0; S->FBest[tid] = 0; }
    if (tid == 0) { S->n_lafrej = 0; S->scnt[0] = S->scnt[1] = S->scnt[2] = S->scnt[3] = 0; }
    __syncthreads();
    if (tid < 4) { double e = 0.; for (int w = 0; w < DG_NW; w++) e = fmax(e, S->extw[w][tid]); S->ext[tid] = e; }
    __syncthreads();

    maxS.I = 8; maxS.J = 0; maxS.Is = 0; maxS.Ilafs = 0; maxSs = maxS;
    no_sam = 0; max_sam = pr.max_iters; iter_cnt = 0; degen_cnt = 0; iterID = 0; Ihmax = 0;
    non_degen = 0;
    best_sample = 0; t_best = t_start;
    finKind = mk_full; accepted = 0;
    perm[0] = 0; perm[1] = 1; perm[2] = 2; perm[3] = 3; p4 = 3; e4kind = mk_full; track = 1;
    done = 0;
    D.flast_k = 0; D.has_last = 0;
    /* a new pair on this slot: its best-score bound starts from zero (no stage of this pair has been published yet) */
    if (coopK > 0 && tid == 0) __hip_atomic_store(&cb->tau_bits, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    /* srand(seed0); seed = rand() */
    if (tid == 0) { dg_srand(&S->rng, A.seeds[pair]); S->itmp[31] = dg_rand(&S->rng); }
    __syncthreads();
    seed = (unsigned)S->itmp[31];
    __syncthreads();

    DG_DEVT(if (tid == 0) { for (int i = 0; i < 8; i++) { S->ph[i] = 0; S->dbg[i] = 0; } S->tq = DG_CLK(); });
#ifdef DG_LO_PROF
    if (tid == 0) { for (int i = 0; i < 16; i++) S->lt[i] = 0; S->ltq = wall_clock64(); }
#endif
#define DG_PH(i) DG_DEVT(if (tid == 0) { long long tq2_ = DG_CLK(); S->ph[i] += tq2_ - S->tq; S->tq = tq2_; })
    /* software pipeline: chunk c is scored while chunk c+1 gets its pool swaps and chunk c+2 its seeds and draws */
    cur = 0; chunk_s[0] = chunk_s[1] = chunk_s[2] = 0; chunk_base = 0;
    if (tid == 0) S->itmp[23] = 0;
    {
        /* the first chunk is 64 samples when nothing counts chunks of DG_CHUNK (the stream mode's ring does): the first local optimisation
         * runs at sample 50 and leaves the bound with which the samples behind it are screened instead of scored */
        const int first = (A.stream_on || coopK > 0) ? DG_CHUNK : (DG_CHUNK < 64 ? DG_CHUNK : 64);
        int cn0 = max_sam - no_sam; if (cn0 > first) cn0 = first; if (cn0 < 0) cn0 = 0;
        int cn1 = max_sam - no_sam - cn0; if (cn1 > DG_CHUNK) cn1 = DG_CHUNK; if (cn1 < 0) cn1 = 0;
        chunk_s[0] = cn0; chunk_s[1] = cn1;
        if (wave == 0) {
            unsigned sd = seed;
            if (cn0 > 0) sd = dg_sample_chunk<7, LDSPTS>(sd, cn0, n, pool, S->seeds3[0], S->draws3[0], S->alm3[0], pscr, lane);
            if (cn1 > 0) sd = dg_sample_draws<7>(sd, cn1, n, S->seeds3[1], S->draws3[1], S->alm3[1], lane);
            if (deep) {
                int cn2_ = max_sam - no_sam - cn0 - cn1; if (cn2_ > DG_CHUNK) cn2_ = DG_CHUNK; if (cn2_ < 0) cn2_ = 0;
                if (cn1 > 0) dg_sample_pool<7, LDSPTS>(cn1, n, pool, S->draws3[1], S->alm3[1], pscr, lane, S->dbg);
                if (cn2_ > 0) sd = dg_sample_draws<7>(sd, cn2_, n, S->seeds3[2], S->draws3[2], S->alm3[2], lane);
            }
            if (lane == 0) S->itmp[31] = (int)sd;
        }
        if (deep) { int cn2_ = max_sam - no_sam - cn0 - cn1; if (cn2_ > DG_CHUNK) cn2_ = DG_CHUNK; if (cn2_ < 0) cn2_ = 0; chunk_s[2] = cn2_; }
        __syncthreads();
        seed = (unsigned)S->itmp[31];
    }
  } else {
    /* continue a pair that was set aside: its LDS image, then the driver state the image carries */
    const char *pk = ws + A.wl.off_park;
    dg_copy16(S, pk, sizeof(dg_f_shared), tid);
    dg_copy16(dyn_smem, pk + DG_PARK_DYN_OFF, (size_t)A.dyn_bytes, tid);
    __syncthreads();
    if (tid == 0) dg_fill_views(&S->K, wsown, A.wl);      /* the image carries the views of the workspace it was written from */
    D = S->park;
    { const long long waited = wall_clock64() - D.t_parked; D.t_start += waited; D.t_best += waited; }   /* reported times = time the pair was being worked on */
    c.n_fds = D.n_fds; c.n_exfds = D.n_exfds; c.n_hds = D.n_hds; c.n_aux = D.n_aux;
    DG_DEVT(if (tid == 0) S->tq = DG_CLK());
    __syncthreads();
  }
    /* the image of the pair for a producer = the image of a pair that is set aside */
    auto write_image = [&]() {
        D.n_fds = c.n_fds; D.n_exfds = c.n_exfds; D.n_hds = c.n_hds; D.n_aux = c.n_aux; D.t_parked = wall_clock64();
        __syncthreads();
        if (tid == 0) S->park = D;
        __syncthreads();
        char *pk = ws + A.wl.off_park;
        dg_copy16(pk, S, sizeof(dg_f_shared), tid);
        dg_copy16(pk + DG_PARK_DYN_OFF, dyn_smem, (size_t)A.dyn_bytes, tid);
        if (tid == 0) { scb->pair = pair; scb->wsid = wsid; scb->img_sam = no_sam; }
    };
    while (!done && no_sam < max_sam) {
        int coop_no_ev = 0;          /* cooperative mode: the screen left no survivor: no model of this chunk can be an event of the commit */
        int pre_cnt = 0, cn3 = 0;    /* cooperative mode: samples of the next chunk solved during this one's scoring; size of the chunk whose seed chain runs in this iteration (deep pipeline) */
        int ff = 0, tail_p = 0;      /* producer: the owner is already past this chunk: sampler stages only; the owner's position */
        const int seq = no_sam / DG_CHUNK;
        dg_stream_ent *ent = (dg_stream_ent *)0;
        if (producer) {
            /* what the owner says: done?  its budget, its position, its bound */
            __syncthreads();
            if (tid == 0) {
                S->itmp[24] = __hip_atomic_load(&scb->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                S->itmp[25] = __hip_atomic_load(&scb->tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                S->itmp[26] = __hip_atomic_load(&scb->max_sam, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            const int stop_ = S->itmp[24], tail_ = S->itmp[25], omax_ = S->itmp[26];
            tail_p = tail_;
            __syncthreads();
            if (stop_) break;
            if (omax_ < max_sam) max_sam = omax_;
            if (no_sam >= max_sam) break;
            ff = seq < tail_;
            if (!ff) {
                /* room in the ring: the slot of this chunk is free once the owner is done with chunk seq - depth */
                if (seq - tail_ >= A.stream_depth) {
                    const int depth_ = A.stream_depth;
                    if (dg_stream_wait(A, &scb->tail, &scb->stop, [=](int t) { return seq - t < depth_; }, &S->itmp[28], A.wait_ticks) < 0) break;
                }
                ent = dg_stream_entry(A, oslot, seq);
            }
        } else if (scb) {
            if (strm >= 1 && tid == 0) {
                /* the owner's position, budget and bound for the producer */
                const double tau_ = maxS.J < maxSs.J ? maxS.J : maxSs.J;
                __hip_atomic_store(&scb->tau_bits, (unsigned long long)__double_as_longlong(tau_ < 0 ? 0.0 : tau_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&scb->max_sam, max_sam, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&scb->tail, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (strm < 2 && !legacy_sym && no_sam >= A.stream_min_sam && max_sam - no_sam >= A.stream_min_left && (strm == 1 || (seq & 3) == 0 || no_sam < 1024)) {
                /* ask for a producer (or renew an unanswered request with a fresh image; or see whether the producer has caught up) */
                __syncthreads();
                if (tid == 0) {
                    int act = 0;                        /* 1 = write the image (the block is mine), 2 = switch to the ring */
                    if (strm == 0) {
                        if ((A.stream_test & 2) || A.stream_early || __hip_atomic_load(A.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= A.n_pairs) {
                            int e = DG_ST_IDLE;
                            if (__hip_atomic_compare_exchange_strong(&scb->state, &e, DG_ST_BUSY, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) act = 1;
                        }
                    } else {
                        const int st_ = __hip_atomic_load(&scb->state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (st_ == DG_ST_REQ) {
                            if (no_sam - img_sam >= 8192) {
                                int e = DG_ST_REQ;
                                if (__hip_atomic_compare_exchange_strong(&scb->state, &e, DG_ST_BUSY, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                                    act = 1; __hip_atomic_fetch_add(A.done_pairs + 1, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                }
                            }
                        } else if (__hip_atomic_load(&scb->head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > seq) act = 2;
                    }
                    S->itmp[30] = act;
                }
                __syncthreads();
                const int act = S->itmp[30];
                __syncthreads();
                if (act == 1) {
                    if (tid == 0) {
                        const double tau_ = maxS.J < maxSs.J ? maxS.J : maxSs.J;
                        __hip_atomic_store(&scb->tau_bits, (unsigned long long)__double_as_longlong(tau_ < 0 ? 0.0 : tau_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&scb->max_sam, max_sam, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&scb->tail, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&scb->head, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&scb->stop, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&scb->owner_sam, (int)(t_start >> 10), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      /* since when the pair runs (10 us units) */
                    }
                    write_image();
                    dg_stream_publish(&scb->state, DG_ST_REQ);
                    if (tid == 0) __hip_atomic_fetch_add(A.done_pairs + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    strm = 1; img_sam = no_sam; park_on = 0;
                } else if (act == 2) strm = 2;
            }
        }
        if (park_on && no_sam >= A.park_sam) {
            /* still running after park_sam samples: set the pair aside if unstarted pairs remain and a spare workspace is left */
            park_on = 0;
            __syncthreads();
            if (tid == 0) {
                int nw = -1;
                if (__hip_atomic_load(A.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < A.n_pairs) {
                    nw = A.n_res + __hip_atomic_fetch_add(A.park_ctl + DG_PARK_SPARE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (nw >= A.n_ws) nw = -1;
                }
                S->itmp[30] = nw;
            }
            __syncthreads();
            const int nw = S->itmp[30];
            __syncthreads();
            if (nw >= 0) {
                D.n_fds = c.n_fds; D.n_exfds = c.n_exfds; D.n_hds = c.n_hds; D.n_aux = c.n_aux; D.t_parked = wall_clock64();
                /* development build (tools/gpu_tail.py): what is known about the pair when it is set aside */
                DG_DEVT(if (A.phase_out && tid == 0) { long long *o_ = A.phase_out + ((size_t)A.n_pairs + 4096 + pair) * 16; o_[0] = max_sam - no_sam; o_[1] = iter_cnt; o_[2] = degen_cnt; o_[3] = D.t_parked - t_start; o_[4] = (long long)maxS.I; });
                if (tid == 0) S->park = D;
                __syncthreads();
                char *pk = ws + A.wl.off_park;
                dg_copy16(pk, S, sizeof(dg_f_shared), tid);
                dg_copy16(pk + DG_PARK_DYN_OFF, dyn_smem, (size_t)A.dyn_bytes, tid);
                __syncthreads();
                /* queue 1 = many samples left (resumed first: the pairs that will run longest must start early), queue 0 = few */
                const int q = (max_sam - no_sam >= A.park_long) ? 1 : 0;
                if (__builtin_amdgcn_readfirstlane(tid >> 6) == 0) {
                    int idx = 0;
                    if (tid == 0) idx = __hip_atomic_fetch_add(A.park_ctl + DG_PARK_CLAIMED + 64 * q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    idx = __builtin_amdgcn_readfirstlane(idx);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (tid == 0) __hip_atomic_store(A.park_q + (size_t)q * A.park_cap + idx, ((long long)pair << 32) | (long long)(unsigned)wsid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return nw;
            }
        }
        int chunk = chunk_s[cur]; if (chunk > max_sam - no_sam) chunk = max_sam - no_sam;
        c.seeds = S->seeds3[cur]; c.draws = S->draws3[cur]; chunk_base = no_sam;
        DG_PH(3);
        DG_PH(0);
        int Mtot = 0, nxt = cur, cn2 = 0;
      int full = 1;                   /* the chunk's 7-point solves and scoring run in this workgroup */
      if (strm == 2) {
        /* ================= the chunk comes from the producer's ring ================= */
        if (seq >= head_seen) {
            const int h_ = dg_stream_wait(A, &scb->head, (int *)0, [=](int h) { return h > seq; }, &S->itmp[28], A.wait_ticks);
            if (h_ < 0) { done = 1; break; }
            head_seen = h_;          /* everything below it is visible after this one acquire */
        }
        const dg_stream_ent *e_ = dg_stream_entry(A, oslot, seq);
        for (int i = tid; i < DG_CHUNK; i += DG_T) {
            S->seeds3[cur][i] = e_->seeds[i];
#pragma unroll
            for (int q = 0; q < 8; q++) S->draws3[cur][i][q] = e_->draws[i][q];
        }
        if (tid == 0) { S->itmp[24] = e_->cn; S->itmp[25] = e_->Mtot; S->itmp[26] = e_->n_ev; S->itmp[27] = e_->overflow; S->dtmp[31] = e_->tau_used; }
        __syncthreads();
        const int cn_ = S->itmp[24], n_ev = S->itmp[26], ovf = S->itmp[27];
        const double tau_used = S->dtmp[31], tau_now = A.hist_out ? 0.0 : (maxS.J < maxSs.J ? maxS.J : maxSs.J);
        chunk = cn_ < max_sam - no_sam ? cn_ : max_sam - no_sam;
        /* more models above the producer's bound than an entry holds (the first chunks of a pair), or the bound has fallen since
         * the producer screened this chunk (a DEGENSAC completion can lower maxS.J: its screens are no superset any more):
         * solve and score the chunk here, from the entry's drawn ids */
        full = ovf || tau_used > tau_now || (A.stream_test & 1);
        __syncthreads();
        if (!full) {
            /* the commit's tables from the entry: models per sample -> slots, every score 0 except the entry's models */
            const unsigned char nvb = tid < DG_CHUNK ? e_->nv[tid] : (unsigned char)0;
            const unsigned v = (tid < chunk && nvb != 255) ? (unsigned)nvb : 0u;     /* only the samples this pair still draws count (its budget may end inside the chunk) */
            unsigned incl = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
            if (lane == 63) S->wave_cnt[wave] = incl;
            __syncthreads();
            unsigned wbase = 0;
            for (int w = 0; w < wave; w++) wbase += S->wave_cnt[w];
            const unsigned excl = wbase + incl - v;
            if (tid < DG_CHUNK) {
                S->moff[tid] = (unsigned short)excl; S->nv[tid] = nvb; S->nsolv[tid] = 0;
                for (unsigned r = 0; r < v; r++) S->mslot[excl + r] = (unsigned short)(tid * 3 + r);
            }
            if (tid == DG_T - 1) S->moff[DG_CHUNK] = (unsigned short)(excl + v);
            __syncthreads();
            Mtot = (int)S->moff[DG_CHUNK];
            for (int i = tid; i < Mtot; i += DG_T) { c.K->res_I[i] = 0; c.K->res_J[i] = 0; }
            __syncthreads();
            for (int e = tid; e < n_ev; e += DG_T) {
                const dg_stream_ev *ev = &e_->ev[e];
                const int k_ = ev->k, r_ = ev->r, mi = (int)S->moff[k_] + r_;
                c.K->res_I[mi] = ev->I; c.K->res_J[mi] = ev->J;
#pragma unroll
                for (int j = 0; j < 9; j++) c.K->gmodels[(size_t)(k_ * 3 + r_) * 9 + j] = ev->model[j];
#pragma unroll
                for (int q = 0; q < 4; q++) S->ridx[k_][q] = ev->ridx[q];
            }
            __syncthreads();
            c.n_fds += Mtot;
        }
      }
      if (full) {
        /* ================= solve: one 7-point problem per lane ================= */
        /* With eight waves the solves run on the scoring waves INSIDE the phase below, next to the sampler stages of waves 0 and 1
         * (the seed chain of chunk c + 2 is the longest of the three): see solve_fused. */
        const bool fuse = DG_NW >= 8 && !(LDSPTS == 0 && coopK > 0) && strm != 2 && !ff;
        if (fuse) { __syncthreads(); if (tid == 0) { S->itmp[21] = 0; S->itmp[22] = 0; } __syncthreads(); }
        int nvalid = 0, nullbad = 0; unsigned rixp = 0;
        if (presolved >= chunk && chunk > 0) {
            /* cooperative mode: this chunk's 7-point problems were solved while the helpers scored the previous chunk */
            if (tid < chunk) { const int *pre = (const int *)(ws + A.wl.off_models + 2 * DG_MTAB_BYTES) + 2 * tid; const int a_ = pre[0]; nvalid = a_ & 0xff; nullbad = (a_ >> 8) & 1; rixp = (unsigned)pre[1]; }
        } else
        if (!fuse && !ff && tid < chunk) {
            int r_ = dg_solve7_lane(P, c.draws[tid], c.K->gmodels + (size_t)tid * 27, &rixp, (double *)&S->ww[wave]);
            if (r_ < 0) nullbad = 1; else nvalid = r_;
        }
        /* ordered slots: exclusive scan of nvalid over the lanes of the chunk */
        if (!fuse) {
            unsigned v = (unsigned)nvalid, incl = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
            if (lane == 63) S->wave_cnt[wave] = incl;
            __syncthreads();
            unsigned wbase = 0;
            for (int w = 0; w < wave; w++) wbase += S->wave_cnt[w];
            unsigned excl = wbase + incl - v;
            if (tid < chunk) {
                S->moff[tid] = (unsigned short)excl;
                S->nv[tid] = nullbad ? 255 : (unsigned char)nvalid;
                S->nsolv[tid] = (unsigned char)((rixp >> 8) & 3u);
                for (int r = 0; r < nvalid; r++) {
                    S->ridx[tid][r] = (unsigned char)((rixp >> (2*r)) & 3u);
                    S->mslot[excl + r] = (unsigned short)(tid * 3 + r);      /* compact model index -> fixed slot */
                }
            }
            if (tid == DG_T - 1) S->moff[DG_CHUNK] = (unsigned short)(excl + v);
            __syncthreads();
        }
        Mtot = (ff || fuse) ? 0 : __builtin_amdgcn_readfirstlane((int)S->moff[DG_CHUNK]);
        /* cooperative mode: the chunk's models are scored by whoever claims the units: the helpers of this slot at once,
         * this workgroup after its sampler stages (dg_coop_cb) */
        double tau_c = A.hist_out ? 0.0 : (maxS.J < maxSs.J ? maxS.J : maxSs.J);
        if (producer) {     /* the owner's bound as it stands now (it may fall later: the owner checks tau_used when it takes the chunk) */
            __syncthreads();
            if (tid == 0) S->dtmp[31] = __longlong_as_double((long long)__hip_atomic_load(&scb->tau_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            __syncthreads();
            tau_c = S->dtmp[31];
            __syncthreads();
        }
        const bool coop_screen = th != 0 && tau_c >= 4.0;
        int coop_units = 0;
        if (LDSPTS == 0 && coopK > 0) {
            const dg_coop_ws cv = dg_coop_views(A, slot);
            unsigned short *gms = (unsigned short *)(ws + A.wl.off_mslot);
            for (int i = tid; i < Mtot; i += DG_T) { gms[i] = S->mslot[i]; cv.cnt[i] = 0u; if (!coop_screen) cv.surv[i] = (unsigned short)i; }
            /* stage 1: screening counts over slices of the point set (DG_COOP_SPW slices per claiming workgroup, at least
             * four tiles each); without a bound to beat, straight to stage 2 with every model */
            int slice = (n + DG_COOP_SPW * (coopK + 1) - 1) / (DG_COOP_SPW * (coopK + 1)); if (slice < 64 * DG_PU * 4) slice = 64 * DG_PU * 4;
            slice = (slice + 64 * DG_PU - 1) / (64 * DG_PU) * (64 * DG_PU);
            coop_units = coop_screen ? (n + slice - 1) / slice : Mtot;
            /* level 2 only: the cooperative mode is for large point sets with few inliers, where random models have far
             * more points inside the looser level-1 band than the bound to beat (C5: thousands against a few hundred), so
             * level 1 would pass nearly every model on to exact scoring */
            if (tid == 0) cb->mtab = mtab;                 /* (published by the release of dg_coop_publish: same wave) */
            if (coop_units > 0) dg_coop_publish(cb, coop_gen, coop_screen ? 1 : 2, coop_units, Mtot, n, mk_full, slice, 0, th, S->ext, tau_c);
        }

        DG_PH(1);
        /* ====== score chunk c (waves 2.., one wave per model, points streamed from LDS)  ||  pool swaps of chunk c+1 (wave 0)  ||  seeds + draws of chunk c+2 (wave 1) ====== */
        nxt = cur == 2 ? 0 : cur + 1; const int nx2 = nxt == 2 ? 0 : nxt + 1;
        {
            if (deep) {
                /* chunk c + 1 has its drawn ids, chunk c + 2 its draws (slot nx2); the chunk behind them gets its seeds now (into the
                 * workspace: all three LDS slots are live) and its draws after the commit, in the slot this chunk leaves */
                cn3 = max_sam - (no_sam + chunk_s[cur] + chunk_s[nxt] + chunk_s[nx2]); if (cn3 > DG_CHUNK) cn3 = DG_CHUNK; if (cn3 < 0) cn3 = 0;
            } else {
                cn2 = max_sam - (no_sam + chunk_s[cur] + chunk_s[nxt]); if (cn2 > DG_CHUNK) cn2 = DG_CHUNK; if (cn2 < 0) cn2 = 0;
                if (strm == 2) cn2 = 0;                        /* the sample stream comes from the producer */
                chunk_s[nx2] = cn2;
            }
            /* deep pipeline: while the helpers score this chunk, waves 2-5 solve the NEXT chunk's 7-point problems into the other model table */
            const bool presolve = deep && chunk_s[nxt] > 0;
            if (presolve) pre_cnt = chunk_s[nxt];
            unsigned *const gseedT = (unsigned *)(ws + A.wl.off_models + 2 * DG_MTAB_BYTES + DG_PRE_BYTES) + (size_t)gpar * DG_CHUNK;
            const unsigned *const gseedP = (const unsigned *)(ws + A.wl.off_models + 2 * DG_MTAB_BYTES + DG_PRE_BYTES) + (size_t)(gpar ^ 1) * DG_CHUNK;
            if (strm == 2) { /* no sampler stages */ }
            else if (wave == 0) {
                const int ps = deep ? nx2 : nxt;
                if (deep && pend_draws) {          /* the draws of that chunk come from waves 6 and 7 (below) */
                    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&S->itmp[23], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < 2) __builtin_amdgcn_s_sleep(1);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                }
                if (chunk_s[ps] > 0) dg_sample_pool<7, LDSPTS>(chunk_s[ps], n, pool, S->draws3[ps], S->alm3[ps], pscr, lane, S->dbg);
            } else if (presolve && wave >= 2 && wave < 2 + DG_CHUNK / 64) {
                const int k_ = (wave - 2) * 64 + lane;
                if (k_ < chunk_s[nxt]) {
                    unsigned rx = 0;
                    double *tab = (double *)(ws + A.wl.off_models + (size_t)(mtab ^ 1) * DG_MTAB_BYTES);
                    const int r_ = dg_solve7_lane(P, S->draws3[nxt][k_], tab + (size_t)k_ * 27, &rx, (double *)((char *)&S->lsq + (size_t)(wave - 2) * 1024));
                    int *pre = (int *)(ws + A.wl.off_models + 2 * DG_MTAB_BYTES) + 2 * k_;
                    pre[0] = r_ < 0 ? 0x100 : r_; pre[1] = (int)rx;
                }
            } else if (deep && pend_draws && wave >= 6 && wave < 8) {
                /* the draws of the chunk whose seeds the previous iteration chained (two rounds of 64 samples per wave), into the slot
                 * the previous chunk left; wave 0 waits for them before that chunk's pool swaps */
                const int cnp = chunk_s[nx2];
                for (int rd = 2 * (wave - 6); rd < 2 * (wave - 6) + 2 && rd < DG_CHUNK / 64; rd++) {
                    dg_sample_draws_round<7>(rd, cnp, n, gseedP, S->draws3[nx2], S->alm3[nx2], lane, S->dbg);
                    const int i_ = rd * 64 + lane; if (i_ < cnp) S->seeds3[nx2][i_] = gseedP[i_];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_fetch_add(&S->itmp[23], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (wave == 1) {
                if (cn3 > 0) { unsigned sd = dg_sample_chain<7>(seed, cn3, gseedT, lane, S->dbg); if (lane == 0) S->itmp[31] = (int)sd; }
                else if (cn2 > 0) { unsigned sd = dg_sample_chain<7>(seed, cn2, S->seeds3[nx2], lane, S->dbg); if (lane == 0) S->itmp[31] = (int)sd; }   /* its draws: after the barrier, one wave per 64 samples */
            }
            /* cooperative mode: the helpers score every group (a whole workgroup per group); the owner's waves only sample.
             * Otherwise: waves 2.. score while waves 0 and 1 run their sampler stages (the critical path); with two
             * waves both score once their stage is done.  Each scoring wave's table of model coefficients lives in its
             * share of the least-squares scratch, which is idle during the main loop. */
            {
                const int NS = DG_NW >= 4 ? DG_NW - 2 : DG_NW, wsi = DG_NW >= 4 ? wave - 2 : wave;
                const int capw = (int)((sizeof(dg_lsq_scratch) / NS) & ~(size_t)15);
                int Ms = Mtot;
                if (fuse && wsi >= 0) {
                    /* solve_fused: scoring wave b < 4 solves the samples 64 b .. 64 b + 63 of the chunk (its scratch: its own scoring
                     * table, not yet in use), the ordered model slots come from the four block totals, and the scoring waves meet at two
                     * LDS counters instead of workgroup barriers (waves 0 and 1 are inside their sampler stages) */
                    constexpr int NB = DG_CHUNK / 64;
                    int nv_ = 0, nb_ = 0; unsigned rx = 0, incl = 0; const int k_ = wsi * 64 + lane;
                    if (wsi < NB) {
                        if (k_ < chunk) {
                            const int r_ = dg_solve7_lane(P, c.draws[k_], c.K->gmodels + (size_t)k_ * 27, &rx, (double *)((char *)&S->lsq + (size_t)wsi * capw));
                            if (r_ < 0) nb_ = 1; else nv_ = r_;
                        }
                        incl = (unsigned)nv_;
#pragma unroll
                        for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
                        if (lane == 63) S->wave_cnt[wsi] = incl;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) __hip_atomic_fetch_add(&S->itmp[21], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&S->itmp[21], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < NB) __builtin_amdgcn_s_sleep(1);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    unsigned tot = 0, wbase = 0;
#pragma unroll
                    for (int b = 0; b < NB; b++) { const unsigned t = S->wave_cnt[b]; if (b < wsi) wbase += t; tot += t; }
                    if (wsi < NB) {
                        const unsigned excl = wbase + incl - (unsigned)nv_;
                        if (k_ < chunk) {
                            S->moff[k_] = (unsigned short)excl;
                            S->nv[k_] = nb_ ? 255 : (unsigned char)nv_;
                            S->nsolv[k_] = (unsigned char)((rx >> 8) & 3u);
                            for (int r = 0; r < nv_; r++) {
                                S->ridx[k_][r] = (unsigned char)((rx >> (2*r)) & 3u);
                                S->mslot[excl + r] = (unsigned short)(k_ * 3 + r);
                            }
                        }
                        if (wsi == 0 && lane == 0) S->moff[DG_CHUNK] = (unsigned short)tot;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) __hip_atomic_fetch_add(&S->itmp[22], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&S->itmp[22], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < NB) __builtin_amdgcn_s_sleep(1);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    Ms = (int)__builtin_amdgcn_readfirstlane(tot);
                }
                if (!(LDSPTS == 0 && coopK > 0) && wsi >= 0 && !ff)
                    dg_score_chunk_F<LDSPTS>(P, n, c.K->gmodels, S->mslot, Ms, wsi, NS, mk_full, th,
                                             tau_c, S->ext, (char *)&S->lsq + (size_t)wsi * capw, capw,
                                             (double *)(c.K->wstage + (size_t)wave * c.K->n_max), c.K->res_I, c.K->res_J, lane, S->scnt);
            }
        }
        c.n_fds += Mtot;   /* provisional: models past the termination point are subtracted below */
        if (LDSPTS == 0 && coopK > 0 && coop_units > 0) {
            /* the sampler stages of this workgroup are done: claim units like a helper, then wait for the units others
             * claimed (each claimed unit belongs to a running workgroup, so the wait ends).  The WHOLE first wave polls,
             * behind a scalar branch: a spin loop under a per-lane `if` would let the compiler re-order the lanes of that wave
             * around the barriers of the enclosing loop (a wave then arrives at s_barrier twice). */
            const dg_coop_ws cv = dg_coop_views(A, slot);
            double *jb0 = (double *)(ws + A.wl.off_hjbuf);
            int units = coop_units;
            for (int st = coop_screen ? 1 : 2; st <= 2; st++) {
                dg_coop_work<T>(A, slot, S, cv, cb, coop_gen, jb0, &S->itmp[28], tid);          /* (starts with a workgroup barrier) */
                if (__builtin_amdgcn_readfirstlane(tid >> 6) == 0) {
                    while (__hip_atomic_load(&cb->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < units) __builtin_amdgcn_s_sleep(2);
                    /* stage 1 leaves device counters that are read with agent-scope atomic loads below; only stage 2 leaves plain data */
                    if (st == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                if (st == 2) break;
                /* survivors of the screen, in model order; the others get J = 0 (never an event in the commit) */
                unsigned ns = 0;
                for (int base = 0; base < Mtot; base += DG_T) {
                    const int mi = base + tid; const bool have = mi < Mtot;
                    const bool keep = have && ((double)__hip_atomic_load(cv.cnt + (have ? mi : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > tau_c);
                    if (have && !keep) { c.K->res_I[mi] = 0; c.K->res_J[mi] = 0; }
                    const unsigned long long bk = __ballot(keep);
                    if (lane == 0) S->wave_cnt[wave] = (unsigned)__popcll(bk);
                    __syncthreads();
                    unsigned off = ns;
                    for (int w = 0; w < DG_NW; w++) { if (w < wave) off += S->wave_cnt[w]; ns += S->wave_cnt[w]; }
                    if (keep) cv.surv[off + (unsigned)__popcll(bk & ((1ull << lane) - 1ull))] = (unsigned short)mi;
                    __syncthreads();
                }
                units = (int)ns;
                if (units == 0) { coop_no_ev = 1; break; }
                dg_coop_publish(cb, coop_gen, 2, units, Mtot, n, mk_full, 0, 0, th, S->ext, tau_c);
            }
        }
        __syncthreads();
        if (tid == 0) S->itmp[23] = 0;                 /* deep pipeline: the "draws are ready" count of waves 6 and 7 */
        if (fuse) { Mtot = __builtin_amdgcn_readfirstlane((int)S->moff[DG_CHUNK]); c.n_fds += Mtot; }
        if (cn2 > 0 || cn3 > 0) seed = (unsigned)S->itmp[31];
        /* the draws of chunk c+2 (its seeds are complete now): the rounds of 64 samples are independent, one wave each; nothing
         * reads them before the pool stage of the next iteration, which sits behind the barriers of the commit */
        if (cn2 > 0 && wave < DG_CHUNK / 64) {
            for (int rd = wave; rd < DG_CHUNK / 64; rd += DG_NW) dg_sample_draws_round<7>(rd, cn2, n, S->seeds3[nx2], S->draws3[nx2], S->alm3[nx2], lane, S->dbg);
        }
        DG_PH(2);
        if (producer) {
            /* leave the chunk in the owner's ring (or just count it when the owner is past it) and go on: a producer never commits */
            if (!ff) {
                __syncthreads();
                if (tid == 0) S->itmp[24] = 0;
                __syncthreads();
                for (int i = tid; i < DG_CHUNK; i += DG_T) {
                    ent->seeds[i] = S->seeds3[cur][i];
#pragma unroll
                    for (int q = 0; q < 8; q++) ent->draws[i][q] = S->draws3[cur][i][q];
                }
                if (tid < DG_CHUNK) {
                    const unsigned char nvb = tid < chunk ? S->nv[tid] : (unsigned char)0;
                    ent->nv[tid] = nvb;
                    if (tid < chunk && nvb != 255)
                        for (int r = 0; r < (int)nvb; r++) {
                            const int mi = (int)S->moff[tid] + r;
                            const double J_ = c.K->res_J[mi];
                            if (J_ > tau_c) {            /* only such a model can be an event of the commit */
                                const int sl = atomicAdd(&S->itmp[24], 1);
                                if (sl < DG_STREAM_EV_MAX) {
                                    dg_stream_ev *ev = &ent->ev[sl];
                                    ev->J = J_; ev->I = c.K->res_I[mi]; ev->k = (short)tid; ev->r = (unsigned char)r; ev->pad = 0;
#pragma unroll
                                    for (int j = 0; j < 9; j++) ev->model[j] = c.K->gmodels[(size_t)S->mslot[mi] * 9 + j];
#pragma unroll
                                    for (int q = 0; q < 4; q++) ev->ridx[q] = S->ridx[tid][q];
                                }
                            }
                        }
                }
                __syncthreads();
                if (tid == 0) {
                    const int ne = S->itmp[24];
                    ent->cn = chunk; ent->Mtot = Mtot; ent->n_ev = ne < DG_STREAM_EV_MAX ? ne : DG_STREAM_EV_MAX; ent->overflow = ne > DG_STREAM_EV_MAX ? 1 : 0; ent->tau_used = tau_c;
                }
            }
            /* one release (a write-back of this XCD's L2) per batch of chunks while the producer is far ahead of the owner */
            if ((seq & 7) == 7 || seq - tail_p < 16 || no_sam + chunk >= max_sam) dg_stream_publish(&scb->head, seq + 1);
            no_sam += chunk;
            __syncthreads();
            cur = nxt;
            continue;
        }
      }
        /* ================= commit: replay exp_ranF.c:1334-1577 in order ================= */
        int k;
        for (k = 0; k < chunk; k++) {
            if (no_sam >= max_sam) break;
            if (no_sam >= DG_ITER_SAM) {
                /* past sample 50 a sample without an "event" (a model beating maxS or maxSs) has no side effect
                 * but no_sam++: jump to the next event sample, found by all lanes in parallel */
                track = 0;
                int stop = chunk;
                if (!coop_no_ev) {             /* (no survivor of the screen: every score of the chunk is 0, nothing to look for) */
                    const double tau = maxS.J < maxSs.J ? maxS.J : maxSs.J;
                    bool ev = false;
                    if (tid >= k && tid < chunk && S->nv[tid] != 255)
                        for (int r = 0; r < S->nv[tid]; r++) ev = ev || (tau < c.K->res_J[S->moff[tid] + r]);
                    unsigned long long bal = __ballot(ev);
                    __syncthreads();
                    if (lane == 0) S->wave_cnt[wave] = bal ? (unsigned)(wave * 64 + __ffsll((long long)bal) - 1) : 0xffffffffu;
                    __syncthreads();
                    unsigned kE = S->wave_cnt[0];
                    for (int w = 1; w < DG_NW; w++) kE = S->wave_cnt[w] < kE ? S->wave_cnt[w] : kE;
                    stop = kE == 0xffffffffu ? chunk : (int)kE;
                }
                int skip = stop - k; if (skip > max_sam - no_sam) skip = max_sam - no_sam;
                no_sam += skip; k += skip;
                if (k >= chunk || no_sam >= max_sam) break;
            }
            no_sam++;
            const int nvk = S->nv[k];
            if (nvk == 255) continue;                              /* nullsize != 2 */
            int new_max = 0, do_iterate = 0, rng_ready = 0, brk = 0;
            for (int r = 0; r < nvk && !brk; r++) {
                const int mi = S->moff[k] + r, ri = S->ridx[k][r];
                dg_score Sc = {c.K->res_I[mi], c.K->res_J[mi], 0, 0};
                const int phys = perm[ri];
                const bool ev1 = maxS.J < Sc.J, ev2 = maxSs.J < Sc.J;
                if (!(ev1 || ev2)) {
                    if (track && phys == p4) { __syncthreads(); if (tid < 9) e4F[tid] = c.K->gmodels[(size_t)S->mslot[mi]*9 + tid]; e4kind = mk_full; __syncthreads(); }
                    continue;
                }
                __syncthreads();
                if (tid < 9) S->f[tid] = c.K->gmodels[(size_t)S->mslot[mi]*9 + tid];
                __syncthreads();
                if (track && phys == p4) { if (tid < 9) e4F[tid] = S->f[tid]; e4kind = mk_full; __syncthreads(); }
                if (ev1) {
                    int pass = 1;
                    if (doSym || doLaf) {
                        /* `inliers` = exact th-list of this model */
                        dg_pass_cfg cl = dg_cfg0(n); cl.list = c.K->L[0]; cl.thL = th;
                        dg_pass_res rl = dg_f_pass(c, S->f, mk_full, cl);
                        pass = dg_f_checks(c, S->f, c.K->L[0], (int)rl.nL, Sc, maxS, mk_full);
                    }
                    if (!pass) continue;
                    { int t = perm[ri]; perm[ri] = perm[3]; perm[3] = t; }
                    maxS = Sc;
                    __syncthreads();
                    if (tid < 9) S->F[tid] = S->f[tid];
                    __syncthreads();
                    new_max = 1; accepted = 1; finKind = mk_full; best_sample = no_sam; t_best = wall_clock64();
                }
                if (maxSs.J < Sc.J) {
                    maxSs = Sc;
                    int degenerate = 0;
                    if (pr.degen) {
                        __syncthreads();
                        if (tid < 7) {                                 /* u7 in samidx order = reverse draw order */
                            dg_pt q = dg_ldpt<LDSPTS>(P, c.draws[k][6 - tid]);
                            S->u7[tid][0] = q.x1; S->u7[tid][1] = q.y1; S->u7[tid][2] = q.x2; S->u7[tid][3] = q.y2;
                        }
                        __syncthreads();
                        { int dgn = dg_checksample(c, S->f, S->u7, 3*th, S->H); if (tid == 0) S->itmp[1] = dgn; }
                        __syncthreads();
                        degenerate = S->itmp[1];
                    }
                    if (degenerate) {
                        DG_PH(3);
                        if (!rng_ready) {
                            __syncthreads();
                            if (tid == 0) { dg_srand(&S->rng, c.seeds[k]); for (int i = 0; i < 8; i++) dg_rand(&S->rng); }
                            __syncthreads();
                            rng_ready = 1;
                        }
                        dg_pass_cfg ch = dg_cfg0(n); ch.flags = c.K->Fl[1]; ch.thF = th*3;
                        dg_pass_res rh = dg_h_pass(c, S->H, ch); c.n_hds++;
                        unsigned I = rh.nF;
                        if (I < 8) { DG_FLAST(S->f); brk = 1; c.n_fds -= (nvk - 1 - r); if (A.hist_out) { __syncthreads(); if (tid == 0) S->nv[k] = (unsigned char)(r + 1); __syncthreads(); } break; }   /* exp_ranF.c:1437-1439: later roots are never scored */
                        { long long ti0 = DG_CLK(); I = dg_innerH(c, S->H, 16*th, 10, c.K->Fl[0]); DG_DEVT(if (tid == 0) S->dbg[0] += DG_CLK() - ti0); (void)ti0; }
                        if ((int)I > Ihmax) Ihmax = (int)I;
                        if (I > 6) {
                            I = dg_rFtH(c, c.K->Fl[0], th, S->H, S->f);
                            if (ri == (int)S->nsolv[k] - 1) DG_FLAST(S->f);      /* no later root overwrites f (exp_ranF.c:1365-1368) */
                            int dphys;
                            if (I > maxS.I) {
                                maxS.I = I;
                                __syncthreads();
                                if (tid < 9) S->F[tid] = S->f[tid];
                                __syncthreads();
                                new_max = 1; accepted = 1; finKind = mk_full; best_sample = no_sam; t_best = wall_clock64();
                                dphys = perm[3];
                            } else dphys = perm[ri];
                            /* FDS1(u, f, errs[..]) + the I/J recount (exp_ranF.c:1456-1480) */
                            dg_pass_cfg cj = dg_cfg0(n); cj.wantJ = 1; cj.thJ = th;
                            dg_pass_res rj = dg_f_pass(c, S->f, mk_full, cj); c.n_fds++;
                            if (new_max) maxS.J = rj.J;
                            if (track && dphys == p4) { __syncthreads(); if (tid < 9) e4F[tid] = S->f[tid]; e4kind = mk_full; __syncthreads(); }
                            ++degen_cnt;
                        }
                        DG_PH(5);
                    } else {
                        do_iterate = (no_sam > DG_ITER_SAM);
                        p4 = phys;                                  /* errs[4] = d */
                        __syncthreads();
                        if (tid < 9) { e4F[tid] = S->f[tid]; S->FBest[tid] = S->f[tid]; }
                        if (tid < 7) S->samidxBest[tid] = c.draws[k][6 - tid];
                        e4kind = mk_full;
                        __syncthreads();
                        non_degen++;
                    }
                }
            }
            if (no_sam == DG_ITER_SAM && non_degen) do_iterate = 1;
            /* a producer that runs ahead screens with the owner's bound: tell it as soon as the bound or the budget moves (an event
             * can keep this workgroup busy for milliseconds before the next chunk starts) */
            if (scb && strm >= 1 && tid == 0) {
                const double tau_ = maxS.J < maxSs.J ? maxS.J : maxSs.J;
                __hip_atomic_store(&scb->tau_bits, (unsigned long long)__double_as_longlong(tau_ < 0 ? 0.0 : tau_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&scb->max_sam, max_sam, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }

            if (do_iterate) {
                if (!rng_ready) {
                    __syncthreads();
                    if (tid == 0) { dg_srand(&S->rng, c.seeds[k]); for (int i = 0; i < 8; i++) dg_rand(&S->rng); }
                    __syncthreads();
                    rng_ready = 1;
                }
                iter_cnt++; track = 0;
                DG_PH(3);
                dg_resid_begin(c, iter_cnt - 1); __syncthreads();
                dg_dump_resid(c, 0, e4F, e4kind);                              /* errs[4], exp_ranF.c:1503-1504 */
                /* LSQ before LO: S = inlidxs(errs[4], TC*th*MWM); u2f; FDS1; inlidxs(th)  (:1506-1511) */
                dg_pass_cfg ca = dg_cfg0(n); ca.list = c.K->L[0]; ca.thL = DG_TC * th * DG_MWM;
                dg_pass_res ra = dg_f_pass(c, e4F, e4kind, ca);
                DG_TRACE(c, 1, ra.nL, no_sam);
                dg_u2f_list(c, c.K->L[0], (int)ra.nL, 0, 0, S->f);
                dg_pass_cfg cb = dg_cfg0(n); cb.wantJ = 1; cb.thJ = th; cb.list = c.K->L[0]; cb.thL = th;
                dg_pass_res rb = dg_f_pass(c, S->f, mk_full, cb); c.n_fds++;
                dg_dump_resid(c, 1, S->f, mk_full);                            /* d after the LSQ, :1511 */
                DG_TRACE(c, 2, rb.nL, rb.J);
                int kb;
                DG_FLAST(S->f);                                                /* u2f wrote f, :1506-1511 */
                dg_score Sl = dg_inFrani(c, (int)rb.nL, th, S->Hx /* LO result model */, &iterID, mk_full, mk_ex, &kb);
                if (Sl.J > 0) DG_FLAST(S->Hx);                                 /* exp_inFranicustom copies its best model into f (:793) */
                if (maxS.J < Sl.J) {
                    if (dg_f_checks(c, S->Hx, c.K->L[0], (int)Sl.I, Sl, maxS, mk_full)) {
                        maxS = Sl;
                        __syncthreads();
                        if (tid < 9) S->F[tid] = S->Hx[tid];
                        __syncthreads();
                        new_max = 1; accepted = 1; finKind = kb; best_sample = no_sam; t_best = wall_clock64();
                    }
                }
                if (new_max && !pr.legacy) {
                    int new_sam = dg_nsamples((int)maxS.I + 1, n, 7, pr.conf);
                    if (new_sam < max_sam) max_sam = new_sam;
                }
                if (scb && strm >= 1 && tid == 0) {
                    const double tau_ = maxS.J < maxSs.J ? maxS.J : maxSs.J;
                    __hip_atomic_store(&scb->tau_bits, (unsigned long long)__double_as_longlong(tau_ < 0 ? 0.0 : tau_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&scb->max_sam, max_sam, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                DG_PH(4);
            }
            if (new_max && pr.legacy) {                 /* exp_ransacF / exp_ransacFcustom: after every sample with a new best model (exp_ranF.c:1085-1090) */
                int new_sam = dg_nsamples((int)maxS.I + 1, n, 7, pr.conf);
                if (new_sam < max_sam) max_sam = new_sam;
            }
        }
        /* models of samples that were never committed do not count as scored */
        if (A.hist_out) {
            /* data_out[LmaxI + 2]++ for every sample of the chunk that was reached (exp_ranF.c:1495; samples whose null
             * space is not 2-dimensional `continue` before it, :1355-1358) */
            int *hist = A.hist_out + (size_t)off + 3 * (size_t)pair;
            const int reached = no_sam - chunk_base;
            if (tid < reached && tid < chunk && S->nv[tid] != 255) {
                unsigned best = 0;
                for (int r = 0; r < S->nv[tid]; r++) { const unsigned I_ = c.K->res_I[S->moff[tid] + r]; best = I_ > best ? I_ : best; }
                atomicAdd(&hist[2 + best], 1);
            }
        }
        if (k < chunk) { c.n_fds -= (Mtot - (int)S->moff[k]); done = 1; }
        else if (no_sam >= max_sam) done = 1;
        __syncthreads();
        if (legacy_sym && !done) {
            /* the chunk was processed to its end: keep the ids of its last sample that reached the cubic (the final filter
             * may need that sample's last root when the run ends inside samples of a later chunk that never get there) */
            const bool cand = tid < chunk && S->nv[tid] != 255;
            const unsigned long long bal = __ballot(cand);
            if (lane == 0) S->wave_cnt[wave] = bal ? (unsigned)(wave * 64 + 63 - __clzll((long long)bal)) : 0xffffffffu;
            __syncthreads();
            int kf = -1;
            for (int w = 0; w < DG_NW; w++) if (S->wave_cnt[w] != 0xffffffffu) kf = (int)S->wave_cnt[w];
            __syncthreads();
            if (kf >= 0) { if (tid < 7) S->lastIds[tid] = c.draws[kf][tid]; D.has_last = 1; }
            __syncthreads();
        }
        if (!done) {
            if (deep) { chunk_s[cur] = cn3; pend_draws = cn3 > 0; gpar ^= 1; }     /* its draws: next iteration, waves 6 and 7, into the slot this chunk leaves */
            cur = nxt;
            presolved = pre_cnt;
            if (pre_cnt > 0) { mtab ^= 1; if (tid == 0) S->K.gmodels = (double *)(ws + A.wl.off_models + (size_t)mtab * DG_MTAB_BYTES); }
        }
    }
    if (producer) {
        /* the producer leaves (everything it wrote is published): the owner may reuse its control block */
        dg_stream_publish(&scb->head, no_sam / DG_CHUNK + (no_sam % DG_CHUNK ? 1 : 0));
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&scb->state, DG_ST_RELEASED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return -1;
    }

    /* ---- "If there were no LOs, do at least one NOW!"  exp_ranF.c:1580-1697 ---- */
    if (!iter_cnt && !degen_cnt && non_degen) {
        int degenerate = 0;
        /* the libc stream continues from wherever the last iteration left it: after its seed draw */
        __syncthreads();
        {
            int dgn = 0;
            if (pr.degen) {
                if (tid < 7) { dg_pt q = dg_ldpt<LDSPTS>(P, S->samidxBest[tid]); S->u7[tid][0] = q.x1; S->u7[tid][1] = q.y1; S->u7[tid][2] = q.x2; S->u7[tid][3] = q.y2; }
                __syncthreads();
                dgn = dg_checksample(c, S->FBest, S->u7, 3*th, S->H);
            }
            if (tid == 0) S->itmp[1] = dgn;
        }
        __syncthreads();
        degenerate = S->itmp[1];
        /* RNG: re-create the state the reference has here (last iteration's srand + 7 draws + seed draw) */
        __syncthreads();
        if (tid == 0) {
            /* S->seeds[] of the last chunk still holds the per-iteration seeds; the last executed
             * iteration is no_sam (1-based) => index (no_sam-1) % DG_CHUNK within its chunk */
            int li = no_sam - 1 - chunk_base; if (li < 0) li = 0;
            dg_srand(&S->rng, c.seeds[li]); for (int i = 0; i < 8; i++) dg_rand(&S->rng);
        }
        __syncthreads();
        if (degenerate) {
            dg_pass_cfg ch = dg_cfg0(n); ch.flags = c.K->Fl[1]; ch.thF = th*3;
            dg_pass_res rh = dg_h_pass(c, S->H, ch); c.n_hds++;
            unsigned I = rh.nF;
            if (I >= 8) I = dg_innerH(c, S->H, 16*th, 10, c.K->Fl[0]);
            else { for (int j = tid; j < n; j += DG_T) c.K->Fl[0][j] = 0; __syncthreads(); }
            if ((int)I > Ihmax) Ihmax = (int)I;
            if (I > 6) {
                __syncthreads();
                if (tid < 9) S->f[tid] = S->FBest[tid];
                __syncthreads();
                I = dg_rFtH(c, c.K->Fl[0], th, S->H, S->f);
                DG_FLAST(S->f);
                int nm = 0;
                if (I > maxS.I) {
                    maxS.I = I;
                    __syncthreads();
                    if (tid < 9) S->F[tid] = S->f[tid];
                    __syncthreads();
                    nm = 1; accepted = 1; finKind = mk_full; best_sample = no_sam; t_best = wall_clock64();
                }
                dg_pass_cfg cj = dg_cfg0(n); cj.wantJ = 1; cj.thJ = th;
                dg_pass_res rj = dg_f_pass(c, S->f, mk_full, cj); c.n_fds++;
                if (nm) maxS.J = rj.J;
                ++degen_cnt;
            }
        } else {
            iter_cnt++;
            dg_resid_begin(c, iter_cnt - 1); __syncthreads();
            /* row 0 (errs[4], exp_ranF.c:1634) stays NaN here: which sample's residuals that physical buffer holds after the
             * loop is not tracked past sample 50 */
            dg_pass_cfg ca = dg_cfg0(n); ca.list = c.K->L[0]; ca.thL = DG_TC * th * DG_MWM;
            dg_pass_res ra = dg_f_pass(c, S->FBest, mk_full, ca);
            dg_u2f_list(c, c.K->L[0], (int)ra.nL, 0, 0, S->f);
            dg_pass_cfg cb = dg_cfg0(n); cb.list = c.K->L[0]; cb.thL = th;
            dg_pass_res rb = dg_f_pass(c, S->f, mk_full, cb); c.n_fds++;
            dg_dump_resid(c, 1, S->f, mk_full);
            int kb;
            DG_FLAST(S->f);
            dg_score Sl = dg_inFrani(c, (int)rb.nL, th, S->Hx, &iterID, mk_full, mk_ex, &kb);
            if (Sl.J > 0) DG_FLAST(S->Hx);
            if (maxS.J < Sl.J) {
                if (dg_f_checks(c, S->Hx, c.K->L[0], (int)Sl.I, Sl, maxS, mk_full)) {
                    maxS = Sl;
                    __syncthreads();
                    if (tid < 9) S->F[tid] = S->Hx[tid];
                    __syncthreads();
                    accepted = 1; finKind = kb; best_sample = no_sam; t_best = wall_clock64();
                }
            }
        }
    }

    DG_PH(3);
    /* ---- final mask: exp_ranF.c:1699-1740 ---- */
    unsigned char *mask = A.mask_out + off;
    if (!accepted) {
        for (int j = tid; j < n; j += DG_T) mask[j] = 0;
    } else {
        double F[9];
        for (int i = 0; i < 9; i++) F[i] = S->F[i];
        for (int j = tid; j < n; j += DG_T) mask[j] = dg_Ferr(finKind, F, dg_ldpt<LDSPTS>(P, j)) <= th ? 1 : 0;
        __syncthreads();
        if (legacy_sym) {
            /* exp_ranF.c:1196-1203: all points against the symmetric error of `f`, the model the driver computed LAST (not
             * the best one F): the recorded event model of the last sample that reached the cubic, else that sample's last root */
            const int li = no_sam - 1 - chunk_base;
            const bool cand = tid <= li && tid < DG_CHUNK && S->nv[tid] != 255;
            const unsigned long long bal = __ballot(cand);
            if (lane == 0) S->wave_cnt[wave] = bal ? (unsigned)(wave * 64 + 63 - __clzll((long long)bal)) : 0xffffffffu;
            __syncthreads();
            int kf = -1;
            for (int w = 0; w < DG_NW; w++) if (S->wave_cnt[w] != 0xffffffffu) kf = (int)S->wave_cnt[w];
            __syncthreads();
            int have = 0;
            if (kf >= 0 && D.flast_k == chunk_base + kf + 1) have = 1;                  /* an event of that very sample wrote f last */
            else if (kf >= 0 || D.has_last) {
                if (tid == 0) { int ids[7]; for (int i = 0; i < 7; i++) ids[i] = kf >= 0 ? c.draws[kf][i] : S->lastIds[i];
                                S->itmp[2] = dg_solve7_lastroot(P, ids, S->flast, (double *)&S->ww[0]); }
                __syncthreads();
                have = S->itmp[2];
            }
            __syncthreads();
            if (have) {
                double Fl[9]; for (int i = 0; i < 9; i++) Fl[i] = S->flast[i];
                for (int j = tid; j < n; j += DG_T) if (dg_Ferr(DG_K_FSYM, Fl, dg_ldpt<LDSPTS>(P, j)) > pr.sym_th) mask[j] = 0;
            }
        } else if (doSym || (doLaf && pr.final_laf_filter)) {
            dg_pass_cfg cl = dg_cfg0(n); cl.list = c.K->L[0]; cl.thL = th;
            dg_pass_res rl = dg_f_pass(c, S->F, finKind, cl);
            const int cnt = (int)rl.nL; const int *lst = c.K->L[0];
            /* clears list POSITION j, not lst[j]: exp_ranF.c:1719-1721 */
            if (doSym)
                for (int j = tid; j < cnt; j += DG_T) if (dg_Ferr(DG_K_FSYM, F, dg_ldpt<LDSPTS>(P, lst[j])) > pr.sym_th) mask[j] = 0;
            if (doLaf && pr.final_laf_filter) {
                double thl = pr.laf_coef * th;
                for (int j = tid; j < cnt; j += DG_T) {
                    if (dg_Ferr(mk_full, F, c.laf_pt(lst[j], 1)) > thl) mask[j] = 0;
                    if (dg_Ferr(mk_full, F, c.laf_pt(lst[j], 2)) > thl) mask[j] = 0;
                }
            }
        }
    }
    if (tid < 9) A.model_out[(size_t)pair * 9 + tid] = accepted ? S->F[tid] : 0.0;
    if (A.hist_out && tid == 0) { int *hist = A.hist_out + (size_t)off + 3 * (size_t)pair; hist[0] = no_sam; hist[1] = iter_cnt; }
    if (A.screen_out && tid == 0) { int *so = A.screen_out + (size_t)pair * 4; for (int i = 0; i < 4; i++) so[i] = (int)S->scnt[i]; }
    if (A.stats_out && tid == 0) {
        int *st = A.stats_out + (size_t)pair * 16;
        long long t_end = wall_clock64();
        st[0] = no_sam; st[1] = iter_cnt; st[2] = S->n_lafrej; st[3] = (int)maxS.I; st[4] = c.n_fds + c.n_exfds;
        st[5] = degen_cnt; st[6] = Ihmax; st[7] = best_sample; st[8] = c.n_fds; st[9] = c.n_exfds;
        st[10] = c.n_hds; st[11] = c.n_aux; st[12] = (int)(t_best - t_start); st[13] = (int)(t_end - t_start);
        st[14] = A.variant_threads; st[15] = A.mode | (resume ? 256 : 0) | (strm == 2 ? 512 : 0);      /* bit 8: the pair was set aside and resumed; bit 9: its chunks came from a producer workgroup */
    }
    if (scb && strm >= 1) {
        /* take the request back, or tell the producer to stop and wait until it has left */
        __syncthreads();
        int gone = 0;
        if (tid == 0) {
            int e = DG_ST_REQ; gone = __hip_atomic_compare_exchange_strong(&scb->state, &e, DG_ST_IDLE, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0; S->itmp[30] = gone;
            if (gone) __hip_atomic_fetch_add(A.done_pairs + 1, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        gone = S->itmp[30];
        __syncthreads();
        if (!gone) {
            if (tid == 0) __hip_atomic_store(&scb->stop, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dg_stream_wait(A, &scb->state, (int *)0, [](int st_) { return st_ == DG_ST_RELEASED; }, &S->itmp[28]);
            if (tid == 0) {
                __hip_atomic_store(&scb->head, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&scb->stop, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&scb->state, DG_ST_IDLE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (A.done_pairs && tid == 0) __hip_atomic_fetch_add(A.done_pairs, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    DG_PH(6);
#ifdef DG_LO_PROF
    if (A.phase_out && tid == 0) { for (int i = 0; i < 16; i++) A.phase_out[(size_t)pair * 16 + i] = S->lt[i]; }
    if (0)
#endif
    DG_DEVT(if (A.phase_out && tid == 0) { S->ph[7] = DG_CLK() - t_start; for (int i = 0; i < 8; i++) A.phase_out[(size_t)pair * 16 + i] = S->ph[i]; for (int i = 0; i < 8; i++) A.phase_out[(size_t)pair * 16 + 8 + i] = S->dbg[i]; A.phase_out[(size_t)pair * 16 + 15] = DG_CLK(); });
#undef DG_PH
    return -1;
}

/* Persistent workgroups: the grid is at most the number of workgroups the device keeps resident, every workgroup owns
 * one scratch slot and pulls pairs from a device-wide ticket counter until the batch is exhausted (optionally in a
 * caller-given order, e.g. expected-cost descending). */
__device__ __forceinline__ int dg_next_pair(const dg_args &A, int *bc /* LDS */)
{
    __syncthreads();
    if (threadIdx.x == 0) *bc = atomicAdd(A.ticket, 1);
    __syncthreads();
    const int t = *bc;
    if (t >= A.n_pairs) return -1;
    return A.order ? A.order[t] : t;
}

/* whole workgroup, after a pair has ended: when a hand-over wait of this launch has timed out (dg_args::err_flag), the pair's
 * results may rest on incomplete data, so they are discarded where every caller sees it — zero model, zero mask, bit 10 of
 * stats[15] — instead of being returned as a success (the asynchronous entry points have no other way to report it; the
 * host-pointer entry points run the flagged pairs again without producers / helpers). */
__device__ __noinline__ void dg_discard_if_failed(const dg_args &A, const int pair, int *bc /* LDS */)
{
    __syncthreads();
    if (threadIdx.x == 0) *bc = __hip_atomic_load(A.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int failed = *bc;
    __syncthreads();
    if (!failed) return;
    const long long off = A.offsets[pair];
    const int n = (int)(A.offsets[pair + 1] - off);
    for (int j = (int)threadIdx.x; j < n; j += (int)blockDim.x) A.mask_out[(size_t)off + j] = 0;
    if (threadIdx.x < 9) A.model_out[(size_t)pair * 9 + threadIdx.x] = 0.0;
    if (A.stats_out && threadIdx.x == 0) { A.stats_out[(size_t)pair * 16 + 3] = 0; A.stats_out[(size_t)pair * 16 + 15] |= 1024; }
}

template <int T, int LDSPTS>
__global__ __launch_bounds__(DG_T, DG_MINW) void dg_find_fundamental_kernel(dg_args A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
    __shared__ __attribute__((aligned(16))) dg_f_shared Sh;
    __shared__ int next_pair;
    __shared__ long long next_parked;
    /* the device code reads the arguments through a pointer (dg_f_ctx::A, also inside non-inlined functions): give it an
     * LDS copy, so the by-value kernel argument's address is never taken (that would make the compiler keep a private
     * per-lane copy of the whole block in scratch memory and turn every uniform argument into a vector value) */
    __shared__ dg_args As;
    if (threadIdx.x == 0) As = A;
    __syncthreads();
    int slot = (int)blockIdx.x, coop_gen = 0;
    if (LDSPTS == 0 && As.coop_k > 0) {
        /* cooperative large-n mode: block b = owner of slot b / (k+1) when b % (k+1) == 0, else one of its helpers */
        slot = (int)blockIdx.x / (As.coop_k + 1);
        const int h = (int)blockIdx.x % (As.coop_k + 1);
        if (h != 0) { dg_f_helper<T>(As, &Sh, slot, h, &next_pair); return; }
    }
    int wsid = slot;
    for (;;) {
        /* 1. pairs that have not been started (with pairs being set aside after park_sam samples this is the discovery
         *    round: it is short, and at its end every pair with a lot of work left is known);
         * 2. the pairs that were set aside with many samples left: they run longest, so they restart first;
         * 3. the pairs that were set aside with few samples left fill the end of the batch.
         * (The queues can look empty one after the other while another workgroup queues a pair in between: that workgroup
         * takes it itself on its next round.) */
        int resume = 0;
        int pair = dg_next_pair(As, &next_pair);
        if (pair < 0 && As.park_sam > 0) {
            long long e = dg_park_take(As, &next_parked, 1);
            if (e < 0) e = dg_park_take(As, &next_parked, 0);
            if (e < 0) e = dg_park_take(As, &next_parked, 1);
            if (e >= 0) { pair = (int)(e >> 32); wsid = (int)(e & 0xffffffffll); resume = 1; }   /* the image lives in the pair's own workspace */
        }
        int img_wsid = wsid, oslot = slot;
        if (pair < 0 && As.stream_on) {
            /* nothing left to own: produce for a pair that asks for it, until every pair of the launch is finished */
            const int j = dg_stream_find(As, &next_pair);
            if (j >= 0) { pair = As.scb[j].pair; img_wsid = As.scb[j].wsid; oslot = j; resume = 2; }
        }
        if (pair < 0) break;
        const int spare = dg_f_pair<T, LDSPTS>(As, &Sh, dyn_smem, pair, slot, img_wsid, resume, coop_gen, wsid, oslot);
        if (spare >= 0) wsid = spare;                /* the pair was set aside with its workspace */
        else if (resume != 2) dg_discard_if_failed(As, pair, &next_pair);
    }
    if (LDSPTS == 0 && As.coop_k > 0 && threadIdx.x == 0)          /* retire the slot: its helpers leave */
        __hip_atomic_store(&As.coop[slot].gen, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    /* development build: when this workgroup ran out of work (tools/gpu_sched.py: the tail of a launch) */
    DG_DEVT(if (As.phase_out && threadIdx.x == 0) As.phase_out[((size_t)As.n_pairs + blockIdx.x) * 16] = DG_CLK());

}

#endif /* DG_KERNEL_F_MAIN_H */
