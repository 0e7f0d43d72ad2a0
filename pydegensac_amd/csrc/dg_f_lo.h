/* Local optimisation of the fundamental-matrix kernel (exp_ranF.c:621-806): the inlier-set hash, exp_iterFcustom / exp_inFranicustom in the serial order
 * (whole workgroup; residual dump, tests) and with one repetition per wave from speculated generator states.
 * Part of the fundamental-matrix kernel: included by dg_kernel_f_main.h, in this order, after dg_kernel_f.h and dg_score_tiles.h. */
#ifndef DG_F_LO_H
#define DG_F_LO_H

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
        /* exp_ranF.c:761 */
        if (c.rrun) { for (size_t j = tid; j < (size_t)(DG_RESIDS_M - 2) * c.n; j += DG_T) c.rrun[2 * (size_t)c.n + j] = 0.; __syncthreads(); }
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
                    const bool same = lane < 31 ? a[lane] == b[lane] : (lane == 31 ? S->rng.f == S->ahead[k].before.f : (lane == 32
                        ? S->rng.b == S->ahead[k].before.b : true));
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
__device__ __noinline__ dg_pass_res dg_f_wpass(const dg_pt *P, int n, int kind, const double *Fm /* LDS */, double thJ, int *la_, double thL, int *lb_,
    double thL2,
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
        if (lane < wave) { const int d = __hip_atomic_load(&S->lo[lane].pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            bad = d >= 0 && d != DG_LO_ASSUMED_DRAWS; }
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
    if (mI < 8) { if (lane == 0) { lg->I = 0; lg->J = 0; lg->kind0 = mk_full; __hip_atomic_store(&lg->pub, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } DG_WSYNC(); return; }
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
        { const bool known = dg_ht_known_wave(c.ht, hash, (int)r1.I, lane); DG_RW(11);
            if (known) { ended = 2; break; } }
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
                if (lane < 2 * ssiz && g->upos[lane] >= 0) { const int old = inliers[g->upos[lane]]; inliers[g->upos[lane]] = g->uval[lane];
                    g->uval[lane] = old; }
                if (lane == 0) { g->g = S->lo_work; g->g0 = S->lo_work; g->pub = -1; g->aborted = 0; }
                DG_WSYNC();
                dg_rand_skip(&S->lo_work, assumed, lane);
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
            if (lane == 0) S->rng = LG(v - 1)->g0;
            DG_WSYNC();
            dg_rand_skip(&S->rng, LG(v - 1)->draws, lane);
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

#endif /* DG_F_LO_H */
