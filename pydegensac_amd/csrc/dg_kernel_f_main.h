/* The persistent fundamental-matrix kernel (exp_ranF.c:1244-1767): the driver of ONE pair (dg_f_pair) and the kernel's ticket loop.
 * Its parts: dg_f_lo.h (local optimisation), dg_f_solve7.h, dg_f_sampler.h, dg_f_score.h, dg_f_sched.h (set-aside queues, stream mode),
 * dg_f_coop.h (cooperative large-n mode). */
#ifndef DG_KERNEL_F_MAIN_H
#define DG_KERNEL_F_MAIN_H
#include "dg_kernel_f.h"
#include "dg_score_tiles.h"

#include "dg_f_lo.h"
#include "dg_f_solve7.h"
#include "dg_f_sampler.h"
#include "dg_f_score.h"
#include "dg_f_sched.h"
#include "dg_f_fan.h"
#include "dg_f_coop.h"

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
    dg_stream_cb *const scb = (A.stream_on || A.fan_k > 0) ? A.scb + oslot : (dg_stream_cb *)0;
    int head_seen = 0;               /* owner: ring entries below this sequence number are known to be visible */
    /* deep pipeline: the seed buffer this iteration's chain writes; the chunk in slot nx2 still needs its draws (its seeds are in the other buffer) */
    int gpar = 0, pend_draws = 0;
    /* cooperative mode: the model table of the current chunk; samples of the current chunk that were solved during the previous chunk's scoring */
    int mtab = 0, presolved = 0;
    int strm = 0, img_sam = 0;       /* owner: 0 = own sample stream, 1 = asked for a producer (image written), 2 = takes its chunks from the ring */
    const int coopK = LDSPTS == 0 ? A.coop_k : 0;
    dg_coop_cb *const cb = coopK > 0 ? A.coop + slot : (dg_coop_cb *)0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    /* cooperative mode with eight waves: the sampler runs one chunk further ahead (pool swaps of chunk c + 2, seed chain of chunk c + 3),
     * so that the 7-point problems of chunk c + 1 can be solved from the start of the phase in which the helpers score chunk c */
    /* fan mode (dg_f_fan.h): this workgroup draws the sample stream into the ring, worker workgroups solve and score the chunks, the
     * commit below takes the completed entries in order (the stream mode's consumer path, strm == 2, from the first chunk on) */
    const bool fan = LDSPTS == 0 && A.fan_k > 0 && coopK > 0 && !resume && !(A.prm.legacy && A.prm.sym_th > 0);
    const bool deep = DG_NW >= 8 && LDSPTS == 0 && coopK > 0 && !A.hist_out && !fan;
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
    if (__builtin_amdgcn_readfirstlane(wave) == 0) { dg_srand_wave(&S->rng, A.seeds[pair], lane); const int v_ = dg_rand_block(&S->rng, 1, lane);
        if (lane == 0) S->itmp[31] = v_; }
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
        if (fan) {
            /* open the pair for this slot's workers: parameters, counters, one release, then the state word */
            if (tid == 0) {
                scb->pair = pair; scb->wsid = wsid; scb->img_sam = 0; scb->fan_kind = mk_full; scb->fan_th = th;
                for (int i = 0; i < 4; i++) scb->fan_ext[i] = S->ext[i];
                __hip_atomic_store(&scb->tau_bits, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&scb->max_sam, max_sam, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&scb->tail, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&scb->head, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&scb->claim, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&scb->stop, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            dg_stream_publish(&scb->state, DG_ST_ATTACHED);
            strm = 2;
        }
    }
  } else {
    /* continue a pair that was set aside: its LDS image, then the driver state the image carries */
    const char *pk = ws + A.wl.off_park;
    dg_copy16(S, pk, sizeof(dg_f_shared), tid);
    dg_copy16(dyn_smem, pk + DG_PARK_DYN_OFF, (size_t)A.dyn_bytes, tid);
    __syncthreads();
    if (tid == 0) dg_fill_views(&S->K, wsown, A.wl);      /* the image carries the views of the workspace it was written from */
    D = S->park;
    /* reported times = time the pair was being worked on */
    { const long long waited = wall_clock64() - D.t_parked; D.t_start += waited; D.t_best += waited; }
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
        /* cooperative mode: samples of the next chunk solved during this one's scoring; size of the chunk whose seed chain runs in this iteration (deep
           pipeline) */
        int pre_cnt = 0, cn3 = 0;
        int ff = 0, tail_p = 0;      /* producer: the owner is already past this chunk: sampler stages only; the owner's position */
        int seq = no_sam / DG_CHUNK;
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
            if (strm < 2 && !legacy_sym && no_sam >= A.stream_min_sam && max_sam - no_sam >= A.stream_min_left && (strm == 1 || (seq & 3) == 0 ||
                no_sam < 1024)) {
                /* ask for a producer (or renew an unanswered request with a fresh image; or see whether the producer has caught up) */
                __syncthreads();
                if (tid == 0) {
                    int act = 0;                        /* 1 = write the image (the block is mine), 2 = switch to the ring */
                    if (strm == 0) {
                        if ((A.stream_test & 2) || A.stream_early || __hip_atomic_load(A.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= A.n_pairs) {
                            int e = DG_ST_IDLE;
                            if (__hip_atomic_compare_exchange_strong(&scb->state, &e, DG_ST_BUSY, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                __HIP_MEMORY_SCOPE_AGENT)) act = 1;
                        }
                    } else {
                        const int st_ = __hip_atomic_load(&scb->state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (st_ == DG_ST_REQ) {
                            if (no_sam - img_sam >= 8192) {
                                int e = DG_ST_REQ;
                                if (__hip_atomic_compare_exchange_strong(&scb->state, &e, DG_ST_BUSY, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT)) {
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
                        __hip_atomic_store(&scb->tau_bits, (unsigned long long)__double_as_longlong(tau_ < 0 ? 0.0 : tau_), __ATOMIC_RELAXED,
                            __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&scb->max_sam, max_sam, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&scb->tail, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&scb->head, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&scb->stop, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        /* since when the pair runs (10 us units) */
                        __hip_atomic_store(&scb->owner_sam, (int)(t_start >> 10), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    write_image();
                    dg_stream_publish(&scb->state, DG_ST_REQ);
                    if (tid == 0) __hip_atomic_fetch_add(A.done_pairs + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    strm = 1; img_sam = no_sam; park_on = 0;
                } else if (act == 2) strm = 2;
            }
        }
        if (fan) {
            /* fan mode: the slot's sampler workgroup draws the stream, the workers complete the entries; here only the commit.  First the
             * scout (one wave): how many of the next entries are complete and uneventful ("uneventful chunk" below) — they are taken in one
             * step; then entry `seq` itself, which the ordinary ring path below commits: wait for it if it is not complete yet. */
            DG_PH(3);
            {
                const double tau_now = maxS.J < maxSs.J ? maxS.J : maxSs.J;
                __syncthreads();
                if (__builtin_amdgcn_readfirstlane(wave) == 0) {
                    int cnt = 0, ns = no_sam, msum = 0;
                    /* lane l looks at the flag of entry seq + l: the leading run of complete entries, then ONE acquire for all of them */
                    const bool fl_ = __hip_atomic_load(dg_fan_flag(A, oslot, seq + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == seq + lane + 1;
                    const unsigned long long nb_ = ~__ballot(fl_);
                    const int nready = nb_ ? __ffsll((long long)nb_) - 1 : 64;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    for (int s_ = seq; cnt < nready; s_++) {
                        const dg_stream_ent *q_ = dg_stream_entry(A, oslot, s_);
                        const int cn_ = __builtin_amdgcn_readfirstlane(q_->cn), mt_ = __builtin_amdgcn_readfirstlane(q_->Mtot);
                        const int ne_ = __builtin_amdgcn_readfirstlane(q_->n_ev), ov_ = __builtin_amdgcn_readfirstlane(q_->overflow);
                        const double tu_ = q_->tau_used;
                        const bool quiet = ne_ == 0 && !ov_ && !(tu_ > tau_now) && !(A.stream_test & 1) && ns >= DG_ITER_SAM && cn_ > 0 && ns + cn_ < max_sam;
                        if (!__builtin_amdgcn_readfirstlane(quiet ? 1 : 0)) break;
                        ns += cn_; msum += mt_; cnt++;
                    }
                    if (lane == 0) { S->itmp[20] = cnt; S->itmp[21] = ns; S->itmp[22] = msum; }
                }
                __syncthreads();
                if (S->itmp[20] > 0) {
                    DG_DEVT(if (tid == 0) S->dbg[5] += 100000ll * S->itmp[20]);
                    no_sam = S->itmp[21]; c.n_fds += S->itmp[22]; track = 0;
                    seq = no_sam / DG_CHUNK;
                    if (tid == 0) __hip_atomic_store(&scb->tail, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
            }
            DG_PH(2);
            {
                const int want = seq + 1;
                if (dg_stream_wait(A, dg_fan_flag(A, oslot, seq), (int *)0, [=](int v) { return v == want; }, &S->itmp[28], A.wait_ticks) < 0) { done = 1;
                    break; }
            }
            DG_PH(0);                    /* development build: phase 2 = the scout, phase 0 = waiting for a worker's entry */
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
                DG_DEVT(if (A.phase_out && tid == 0) { long long *o_ = A.phase_out + ((size_t)A.n_pairs + 4096 + pair) * 16; o_[0] = max_sam - no_sam;
                    o_[1] = iter_cnt; o_[2] = degen_cnt; o_[3] = D.t_parked - t_start; o_[4] = (long long)maxS.I; });
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
                    if (tid == 0) __hip_atomic_store(A.park_q + (size_t)q * A.park_cap + idx, ((long long)pair << 32) | (long long)(unsigned)wsid,
                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return nw;
            }
        }
        int chunk = chunk_s[cur]; if (chunk > max_sam - no_sam) chunk = max_sam - no_sam;
        c.seeds = S->seeds3[cur]; c.draws = S->draws3[cur]; chunk_base = no_sam;
        DG_PH(3);
        DG_PH(0);
        int Mtot = 0, nxt = cur, cn2 = 0;
        double tau_scored = 0.0;      /* the bound this chunk's models were screened against: a model whose candidate count did not exceed it has J = 0 */
      int full = 1;                   /* the chunk's 7-point solves and scoring run in this workgroup */
      if (strm == 2) {
        /* ================= the chunk comes from the producer's ring ================= */
        if (!fan && seq >= head_seen) {
            const int h_ = dg_stream_wait(A, &scb->head, (int *)0, [=](int h) { return h > seq; }, &S->itmp[28], A.wait_ticks);
            if (h_ < 0) { done = 1; break; }
            head_seen = h_;          /* everything below it is visible after this one acquire */
        }
        DG_PH(0);                    /* development build: phase 0 = waiting for the producer's ring */
        const dg_stream_ent *e_ = dg_stream_entry(A, oslot, seq);
        __syncthreads();
        if (tid == 0) { S->itmp[24] = e_->cn; S->itmp[25] = e_->Mtot; S->itmp[26] = e_->n_ev; S->itmp[27] = e_->overflow; S->dtmp[31] = e_->tau_used; }
        __syncthreads();
        const int cn_ = S->itmp[24], n_ev = S->itmp[26], ovf = S->itmp[27];
        const double tau_used = S->dtmp[31], tau_now = A.hist_out ? 0.0 : (maxS.J < maxSs.J ? maxS.J : maxSs.J);
        /* An uneventful chunk: the producer found no model above a bound that is still at most the pair's (so none is above the pair's
         * bound now), the samples before it are past the point where every sample has side effects (DG_ITER_SAM) and the budget does
         * not end inside it.  The commit would skip all of its samples in one step (the event search finds nothing): do exactly that,
         * without staging its seeds, ids and tables — a long pair's owner spent ~64 us per such chunk, 25 of its 40-50 ms (round 6).  The last
         * chunk of the budget always takes the ordinary path: what runs after the loop reads its seeds. */
        if (n_ev == 0 && !ovf && !(tau_used > tau_now) && !(A.stream_test & 1) && no_sam >= DG_ITER_SAM && cn_ > 0 && no_sam + cn_ < max_sam) {
            const int Mt_ = S->itmp[25];
            __syncthreads();                                /* (the header words are read; the next iteration rewrites them) */
            track = 0;
            no_sam += cn_; c.n_fds += Mt_;
            /* development build: chunks taken this way (tools/gpu_phases.py: "draws" of a streamed pair, x 1e5) */
            DG_DEVT(if (tid == 0) S->dbg[5] += 100000);
            continue;
        }
        if (fan) {
            /* the three LDS slots belong to this workgroup's own sampler: the commit reads the chunk's seeds and ids where they are */
            c.seeds = (unsigned *)e_->seeds; c.draws = (int (*)[8])e_->draws;
        } else {
            for (int i = tid; i < DG_CHUNK; i += DG_T) {
                S->seeds3[cur][i] = e_->seeds[i];
#pragma unroll
                for (int q = 0; q < 8; q++) S->draws3[cur][i][q] = e_->draws[i][q];
            }
        }
        __syncthreads();
        chunk = cn_ < max_sam - no_sam ? cn_ : max_sam - no_sam;
        /* more models above the producer's bound than an entry holds (the first chunks of a pair), or the bound has fallen since
         * the producer screened this chunk (a DEGENSAC completion can lower maxS.J: its screens are no superset any more):
         * solve and score the chunk here, from the entry's drawn ids */
        full = ovf || tau_used > tau_now || (A.stream_test & 1);
        __syncthreads();
        if (!full) {
            /* the commit's tables from the entry: models per sample -> slots, every score 0 except the entry's models */
            const unsigned char nvb = tid < DG_CHUNK ? e_->nv[tid] : (unsigned char)0;
            /* only the samples this pair still draws count (its budget may end inside the chunk) */
            const unsigned v = (tid < chunk && nvb != 255) ? (unsigned)nvb : 0u;
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
            tau_scored = tau_used;
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
            if (tid < chunk) { const int *pre = (const int *)(ws + A.wl.off_models + 2 * DG_MTAB_BYTES) + 2 * tid; const int a_ = pre[0]; nvalid = a_ & 0xff;
                nullbad = (a_ >> 8) & 1; rixp = (unsigned)pre[1]; }
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
        tau_scored = tau_c;
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
        /* ====== score chunk c (waves 2.., one wave per model, points streamed from LDS)  ||  pool swaps of chunk c+1 (wave 0)  ||  seeds + draws of chunk c+2
           (wave 1) ====== */
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
                    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&S->itmp[23], __ATOMIC_RELAXED,
                        __HIP_MEMORY_SCOPE_WORKGROUP)) < 2) __builtin_amdgcn_s_sleep(1);
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
                /* its draws: after the barrier, one wave per 64 samples */
                else if (cn2 > 0) { unsigned sd = dg_sample_chain<7>(seed, cn2, S->seeds3[nx2], lane, S->dbg); if (lane == 0) S->itmp[31] = (int)sd; }
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
                            const int r_ = dg_solve7_lane(P, c.draws[k_], c.K->gmodels + (size_t)k_ * 27, &rx,
                                (double *)((char *)&S->lsq + (size_t)wsi * capw));
                            if (r_ < 0) nb_ = 1; else nv_ = r_;
                        }
                        incl = (unsigned)nv_;
#pragma unroll
                        for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
                        if (lane == 63) S->wave_cnt[wsi] = incl;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) __hip_atomic_fetch_add(&S->itmp[21], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&S->itmp[21], __ATOMIC_RELAXED,
                        __HIP_MEMORY_SCOPE_WORKGROUP)) < NB) __builtin_amdgcn_s_sleep(1);
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
                    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&S->itmp[22], __ATOMIC_RELAXED,
                        __HIP_MEMORY_SCOPE_WORKGROUP)) < NB) __builtin_amdgcn_s_sleep(1);
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
                    dg_wait_count(A, &cb->done, units, 4, 2);
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
            for (int rd = wave; rd < DG_CHUNK / 64; rd += DG_NW) dg_sample_draws_round<7>(rd, cn2, n, S->seeds3[nx2], S->draws3[nx2], S->alm3[nx2], lane,
                S->dbg);
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
                    ent->cn = chunk; ent->Mtot = Mtot; ent->n_ev = ne < DG_STREAM_EV_MAX ? ne : DG_STREAM_EV_MAX; ent->overflow = ne > DG_STREAM_EV_MAX ? 1 : 0;
                        ent->tau_used = tau_c;
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
                    if (track && phys == p4) { __syncthreads(); if (tid < 9) e4F[tid] = c.K->gmodels[(size_t)S->mslot[mi]*9 + tid]; e4kind = mk_full;
                        __syncthreads(); }
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
                    DG_TRACE(c, 30, Sc.I, Sc.J);
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
                            if (__builtin_amdgcn_readfirstlane(wave) == 0) { dg_srand_wave(&S->rng, c.seeds[k], lane); dg_rand_skip(&S->rng, 8, lane); }
                            __syncthreads();
                            rng_ready = 1;
                        }
                        dg_pass_cfg ch = dg_cfg0(n); ch.flags = c.K->Fl[1]; ch.thF = th*3;
                        dg_pass_res rh = dg_h_pass(c, S->H, ch); c.n_hds++;
                        unsigned I = rh.nF;
                        DG_TRACE(c, 31, I, no_sam);
                        /* exp_ranF.c:1437-1439: later roots are never scored */
                        if (I < 8) { DG_FLAST(S->f); brk = 1; c.n_fds -= (nvk - 1 - r); if (A.hist_out) { __syncthreads();
                            if (tid == 0) S->nv[k] = (unsigned char)(r + 1); __syncthreads(); } break; }
                        { long long ti0 = DG_CLK(); I = dg_innerH(c, S->H, 16*th, 10, c.K->Fl[0]); DG_DEVT(if (tid == 0) S->dbg[0] += DG_CLK() - ti0);
                            (void)ti0; }
                        DG_TRACE(c, 32, I, 0);
                        if ((int)I > Ihmax) Ihmax = (int)I;
                        if (I > 6) {
                            I = dg_rFtH(c, c.K->Fl[0], th, S->H, S->f);
                            DG_TRACE(c, 33, I, maxS.I);
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
                            {
                                /* exp_ranF.c:1478-1480 can LOWER maxS.J (the recount is the MSAC sum of the plane-and-parallax model, whatever
                                 * the sample's own model scored).  The rest of this chunk was screened against the bound of the chunk's start: a
                                 * later model whose J lies between the new bound and that one may carry J = 0 although the reference would
                                 * now look at it (found by the LAF-rejection counter of the fuzz sweep: same results, 3 rejections against the
                                 * reference's 10).  Rare (a DEGENSAC completion inside the chunk that also made a new best): score the chunk's
                                 * models again against the new bound, all waves; a ring chunk that only carried the producer's few candidates
                                 * gets its models solved again first. */
                                const double tau_new = maxS.J < maxSs.J ? maxS.J : maxSs.J;
                                if (tau_new < tau_scored && !A.hist_out && th != 0) {
                                    __syncthreads();
                                    if (strm == 2 && !full) {
                                        if (tid >= k && tid < chunk) {
                                            unsigned rixp = 0;
                                            const int r_ = dg_solve7_lane(P, c.draws[tid], c.K->gmodels + (size_t)tid * 27, &rixp, (double *)&S->ww[wave]);
                                            const int nvv = r_ < 0 ? 0 : r_;
                                            S->nsolv[tid] = (unsigned char)((rixp >> 8) & 3u);
                                            for (int q = 0; q < nvv; q++) S->ridx[tid][q] = (unsigned char)((rixp >> (2*q)) & 3u);
                                        }
                                        __syncthreads();
                                    }
                                    /* the helpers' bound of this pair follows (only this workgroup writes it) */
                                    if (LDSPTS == 0 && coopK > 0 && tid == 0)
                                        __hip_atomic_store(&cb->tau_bits, (unsigned long long)__double_as_longlong(tau_new < 0 ? 0.0 : tau_new),
                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    {
                                        const int capr = (int)((sizeof(dg_lsq_scratch) / DG_NW) & ~(size_t)15);
                                        /* only what the commit has not consumed yet: this sample's remaining roots and the later samples (the slots of
                                         * earlier samples may hold event models or leftovers of earlier chunks, and their scores are final); the second
                                         * look at these models is not counted in scnt */
                                        dg_score_chunk_F<LDSPTS>(P, n, c.K->gmodels, S->mslot, Mtot, wave, DG_NW, mk_full, th, tau_new, S->ext,
                                                                 (char *)&S->lsq + (size_t)wave * capr, capr,
                                                                 (double *)(c.K->wstage + (size_t)wave * c.K->n_max), c.K->res_I, c.K->res_J, lane,
                                                                     (unsigned *)0,
                                                                 (int)S->moff[k] + r + 1);
                                    }
                                    __syncthreads();
                                    tau_scored = tau_new;
                                }
                            }
                        }
                        DG_PH(5);
                    } else {
                        do_iterate = (no_sam > DG_ITER_SAM);
                        p4 = phys;                                  /* errs[4] = d */
                        /* ... and d's buffer can be written again BEFORE this sample's local optimisation reads errs[4]: when this root was
                         * also accepted as the best model, d is errs[3] now; a later root of the same sample that is accepted takes that
                         * buffer as its errs[i] (exp_ranF.c:1409-1410) and, if it turns out degenerate, the plane-and-parallax model's
                         * residuals land in it (:1463-1466).  Past sample 50 the bookkeeping of errs[4]'s buffer is otherwise off (every
                         * use is preceded by an assignment): switch it on for the rest of this sample.  (`tools/gpu_fuzz.py 6000 202`,
                         * case 3465: three roots of one sample, the first sets errs[4], the third is accepted and degenerate.) */
                        track = 1;
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
                    if (__builtin_amdgcn_readfirstlane(wave) == 0) { dg_srand_wave(&S->rng, c.seeds[k], lane); dg_rand_skip(&S->rng, 8, lane); }
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
                    __hip_atomic_store(&scb->tau_bits, (unsigned long long)__double_as_longlong(tau_ < 0 ? 0.0 : tau_), __ATOMIC_RELAXED,
                        __HIP_MEMORY_SCOPE_AGENT);
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
            /* its draws: next iteration, waves 6 and 7, into the slot this chunk leaves */
            if (deep) { chunk_s[cur] = cn3; pend_draws = cn3 > 0; gpar ^= 1; }
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
                if (tid < 7) { dg_pt q = dg_ldpt<LDSPTS>(P, S->samidxBest[tid]); S->u7[tid][0] = q.x1; S->u7[tid][1] = q.y1; S->u7[tid][2] = q.x2;
                    S->u7[tid][3] = q.y2; }
                __syncthreads();
                dgn = dg_checksample(c, S->FBest, S->u7, 3*th, S->H);
            }
            if (tid == 0) S->itmp[1] = dgn;
        }
        __syncthreads();
        degenerate = S->itmp[1];
        /* RNG: re-create the state the reference has here (last iteration's srand + 7 draws + seed draw) */
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(wave) == 0) {
            /* S->seeds[] of the last chunk still holds the per-iteration seeds; the last executed
             * iteration is no_sam (1-based) => index (no_sam-1) % DG_CHUNK within its chunk */
            int li = no_sam - 1 - chunk_base; if (li < 0) li = 0;
            dg_srand_wave(&S->rng, c.seeds[li], lane); dg_rand_skip(&S->rng, 8, lane);
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
        /* bit 8: the pair was set aside and resumed; bit 9: its chunks came from a producer workgroup */
        st[14] = A.variant_threads; st[15] = A.mode | (resume ? 256 : 0) | (strm == 2 ? 512 : 0);
    }
    if (fan) {
        /* the workers leave: nothing of theirs is read any more */
        if (tid == 0) __hip_atomic_store(&scb->stop, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (scb && strm >= 1) {
        /* take the request back, or tell the producer to stop and wait until it has left */
        __syncthreads();
        int gone = 0;
        if (tid == 0) {
            int e = DG_ST_REQ;
                gone = __hip_atomic_compare_exchange_strong(&scb->state, &e, DG_ST_IDLE, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
                S->itmp[30] = gone;
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
    DG_DEVT(if (A.phase_out && tid == 0) { S->ph[7] = DG_CLK() - t_start; for (int i = 0; i < 8; i++) A.phase_out[(size_t)pair * 16 + i] = S->ph[i];
        for (int i = 0; i < 8; i++) A.phase_out[(size_t)pair * 16 + 8 + i] = S->dbg[i]; A.phase_out[(size_t)pair * 16 + 15] = DG_CLK(); });
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
    if (LDSPTS == 0 && As.fan_k > 0) {
        /* fan mode: the blocks behind the owners and their helpers are the owners' workers */
        const int base = As.n_res * (As.coop_k + 1);
        if ((int)blockIdx.x >= base) {
            const int w = (int)blockIdx.x - base;
            if (w % As.fan_k == 0) dg_f_fan_sampler<T>(As, &Sh, w / As.fan_k, As.fan_ws0 + w, &next_pair);       /* the slot's sampler ... */
            else dg_f_fan_worker<T>(As, &Sh, w / As.fan_k, As.fan_ws0 + w, &next_pair);                         /* ... and its workers */
            return;
        }
    }
    int slot = (int)blockIdx.x, coop_gen = 0;
    /* development build: when this workgroup started */
    DG_DEVT(if (As.phase_out && threadIdx.x == 0) { long long *o_ = As.phase_out + ((size_t)As.n_pairs + blockIdx.x) * 16; o_[10] = DG_CLK(); o_[11] = 0; });
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
    /* ... and where its waves sit: HW_ID | XCC_ID << 32 per wave (tools/gpu_simd.py: which SIMDs the serial waves of co-resident workgroups share) */
    DG_DEVT(if (As.phase_out && (threadIdx.x & 63) == 0) As.phase_out[((size_t)As.n_pairs + blockIdx.x) * 16 + 1 + (threadIdx.x >> 6)] =
                (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32));

}

#endif /* DG_KERNEL_F_MAIN_H */
