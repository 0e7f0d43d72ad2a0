/* Cooperative large-n mode (dg_coop_cb; C5): claimable units of screening, exact scoring, distributed passes and whole repetitions of the local
 * optimisation, the helper workgroups' loop and the owner's side of every stage.
 * Part of the fundamental-matrix kernel: included by dg_kernel_f_main.h, in this order, after dg_kernel_f.h and dg_score_tiles.h. */
#ifndef DG_F_COOP_H
#define DG_F_COOP_H

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
#define DG_COOP_SPW 1            /* stage 1: point slices per claiming workgroup (C5: 85.4 ms with 3, 82.8 with 2, 80.4 with 1: a unit's claim and its release \
   cost ~3 us) */
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
            __hip_atomic_store(&cb->next, (int)((((unsigned)(coop_gen + 1) & DG_COOP_GEN_MASK) << 12) | (unsigned)n_units), __ATOMIC_RELAXED,
                __HIP_MEMORY_SCOPE_AGENT);
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
                if (threadIdx.x == 0) { int e = v;
                    ok = __hip_atomic_compare_exchange_strong(&cb->next, &e, v - 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0; }
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
            cq = l1 ? dg_l1_tile_counts<0>(v.P, lo, hi, (const float *)tab, nb, lane) : dg_l2_tile_counts<0>(v.P, lo, hi, (const double *)tab, nb, kind, t94b,
                lane);
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
__device__ __noinline__ void dg_lo_rep_wg(dg_f_shared *S, const dg_pt *P, const int n, const dg_ht &ht, dg_lo_log *lg, int *ib, int *alt, int *sA, int *sB,
    double *jbuf,
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
        const dg_pass_res r = dg_wpass_slice(P, lo, hi, [&](const dg_pt &q) { return dg_Ferr(kind, F, q); }, thJ, la ? sA + lo : (int *)0, thL,
            lb ? sB + lo : (int *)0, thL2,
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
        if (wantJ && tid >= T - 64) {       /* the last wave: the slices' terms one slice after the other, fed through LDS (the fits' design-matrix block) */
            double J = 0.0;
            for (int w = 0; w < NWt; w++) { const int l_ = w * sl < n ? w * sl : n; J = dg_seq_sum_wave_g(jbuf + l_, (int)wc[4 * w + 3], J, S->lsq.Z, 192,
                lane); }
            if (lane == 0) S->red.bc[0] = J;
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
    dg_lo_rep_wg<T>(S, v.P, job->n, ht, (dg_lo_log *)(lj + 128 + (size_t)DG_LOJOB_STRIDE * u), dg_coop_lo_list(A, slot, 2 * u), dg_coop_lo_list(A, slot,
        2 * u + 1),
                    dg_coop_lo_list(A, slot, 2 * DG_RAN_REP + 2 * u), dg_coop_lo_list(A, slot, 2 * DG_RAN_REP + 2 * u + 1), jbuf, job->ssiz, job->th,
                        job->mk_full, job->mk_ex, tid);
}

/* Whole workgroup (owner or helper): work on generation G until it has no unclaimed unit left */
template <int T>
__device__ __forceinline__ void dg_coop_work(const dg_args &A, int slot, dg_f_shared *S, const dg_coop_ws &v, dg_coop_cb *cb, int G, double *jbuf,
    int *bc /* LDS */, int tid)
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
            while ((g = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&cb->gen, __ATOMIC_RELAXED,
                __HIP_MEMORY_SCOPE_AGENT))) == last) __builtin_amdgcn_s_sleep(4);
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
        dg_wait_count(A, &cb->done, n_units, 4, 2);
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
    if (cfg.wantJ && tid >= DG_T - 64) {    /* the last wave, the units' terms fed through LDS (dg_seq_sum_wave_g) */
        double J = 0.0;
        for (int u = 0; u < n_units; u++) J = dg_seq_sum_wave_g(v.stg_j + (size_t)u * slice, (int)v.rec[u].nJ, J, S->lsq.Z, 192, tid & 63);
        if ((tid & 63) == 0) S->red.bc[0] = J;
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
    if (tid == 0) { dg_lo_job *job = (dg_lo_job *)dg_coop_lojob(A, c.coop_slot); job->n = c.n; job->ssiz = ssiz; job->mk_full = mk_full; job->mk_ex = mk_ex;
        job->th = th; }
    dg_coop_publish(cb, *c.coop_gen, 4, nr, 0, c.n, mk_full, 0, 0, th, S->ext, 0.0);
    dg_coop_work<DG_T>(A, c.coop_slot, S, v, cb, *c.coop_gen, (double *)(A.ws + (size_t)c.coop_slot * A.wl.stride + A.wl.off_hjbuf), &S->itmp[28], tid);
    if (__builtin_amdgcn_readfirstlane(tid >> 6) == 0) {
        dg_wait_count(A, &cb->done, nr, 4, 8);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

#endif /* DG_F_COOP_H */
