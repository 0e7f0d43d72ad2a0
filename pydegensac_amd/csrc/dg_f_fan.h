/* Fan mode of the fundamental-matrix kernel (dg_args::fan_k; one pair of many thousand correspondences on an otherwise idle device: C5).
 * The cooperative large-n mode hands every chunk of 256 samples to its helper workgroups in two stages (screening counts over point
 * slices, then exact scoring of the survivors) and waits for each: two hand-overs per chunk, 80 us per chunk against the 25-34 us the
 * seed chain needs.  Here nothing waits per chunk.  Three kinds of workgroup serve one pair:
 *   - the SAMPLER (dg_f_fan_sampler) draws the pair's sample stream chunk after chunk (seed chain, draws, pool swaps: outcome-independent,
 *     exp_ranF.c:1337-1342) and writes each chunk's seeds and drawn ids into an entry of the stream mode's ring;
 *   - WORKERS (dg_f_fan_worker, fan_k - 1 per pair) claim entries as they appear, solve the chunk's 7-point problems, screen and score
 *     its models against the owner's points with the owner's current bound, and complete the entry exactly as a stream-mode producer
 *     would (models per sample, the few models above the bound);
 *   - the OWNER commits the completed entries in order (dg_f_pair, strm == 2 from the first chunk): an entry without a model above the
 *     bound is one addition ("uneventful chunk"; a scout wave takes up to 64 of them per step), the others go through the ordinary ring
 *     path; its local optimisations run while the sampler keeps drawing.
 * The cooperative helpers stay for what does need many workgroups at once: the passes and repetitions of the local optimisations, and
 * the chunks the owner has to score itself (the first one; a chunk whose bound has fallen).
 * Hand-over: plain payload, one agent-scope release, then a relaxed agent-scope word on its own 128-byte line (scb->head for "ids
 * published", fan_flags[] for "entry complete", scb->tail for "entry committed: its slot is free").  Nobody waits for a workgroup that
 * may not be running: sampler and workers wait only once their owner has opened the pair (state ATTACHED), with a limit; the owner
 * waits (wait_ticks) for an entry that a running worker has claimed or will claim — a time-out raises the launch's error word:
 * discard and re-run without the mode.  C5: 61.3 -> 30.1 ms (profiles/r6_phases_c5.log).
 * Part of the fundamental-matrix kernel: included by dg_kernel_f_main.h after dg_f_sched.h. */
#ifndef DG_F_FAN_H
#define DG_F_FAN_H

__device__ __forceinline__ int *dg_fan_flag(const dg_args &A, int oslot, int seq)
{
    return A.fan_flags + ((size_t)oslot * A.stream_depth + (size_t)(seq % A.stream_depth)) * 32;
}

/* worker `widx` of owner slot `oslot`, on workspace `wsid`: serves the one pair its owner opens, then leaves */
template <int T>
__device__ __noinline__ void dg_f_fan_worker(const dg_args &A, dg_f_shared *S, const int oslot, const int wsid, int *bc /* LDS */)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    dg_stream_cb *const scb = A.scb + oslot;
    /* the owner opens its pair (state ATTACHED) or the launch ends without one for this slot */
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(wave) == 0) {
        int res = 0;
        const long long t0 = wall_clock64();
        for (;;) {
            if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&scb->state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == DG_ST_ATTACHED) { res = 1; break;
                }
            if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&scb->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) break;
            if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(A.done_pairs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= A.n_pairs) break;
            if (wall_clock64() - t0 > 4 * (long long)A.wait_ticks + 1000000ll) break;             /* (an owner that never starts: leave) */
            __builtin_amdgcn_s_sleep(32);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *bc = res;
    }
    __syncthreads();
    if (!*bc) return;
    __syncthreads();
    const int pair = scb->pair, kind = scb->fan_kind;
    const double th = scb->fan_th;
    const long long off = A.offsets[pair];
    const int n = (int)(A.offsets[pair + 1] - off);
    char *const ws = A.ws + (size_t)wsid * A.wl.stride;
    const dg_pt *P = (const dg_pt *)(A.ws + (size_t)scb->wsid * A.wl.stride + A.wl.off_pts);          /* the owner's staged correspondences */
    if (tid == 0) dg_fill_views(&S->K, ws, A.wl);
    if (tid < 4) S->ext[tid] = scb->fan_ext[tid];
    __syncthreads();
    const __attribute__((address_space(3))) dg_f_cshared *K = (const __attribute__((address_space(3))) dg_f_cshared *)&S->K;
    for (;;) {
        /* claim the next chunk whose ids are in the ring */
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(wave) == 0) {
            int seq = -1;
            const long long t0 = wall_clock64();
            for (;;) {
                if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&scb->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) break;
                const int cl = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&scb->claim, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                const int ms = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&scb->max_sam, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if ((long long)cl * DG_CHUNK >= (long long)ms) break;                              /* the owner's budget ends before that chunk */
                const int hd = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&scb->head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (cl < hd) {
                    int ok = 0;
                    if (lane == 0) { int e = cl;
                        ok = __hip_atomic_compare_exchange_strong(&scb->claim, &e, cl + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                            __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0; }
                    if (__builtin_amdgcn_readfirstlane(ok)) { seq = cl; break; }
                    continue;
                }
                /* (the owner is a running workgroup: it publishes or stops) */
                if (wall_clock64() - t0 > 4 * (long long)A.wait_ticks + 1000000ll) break;
                __builtin_amdgcn_s_sleep(8);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            *bc = seq;
        }
        __syncthreads();
        const int seq = *bc;
        __syncthreads();
        if (seq < 0) break;
        dg_stream_ent *ent = dg_stream_entry(A, oslot, seq);
        const int chunk = ent->cn;
        const double tau_c = __longlong_as_double((long long)__hip_atomic_load(&scb->tau_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        /* solve: one 7-point problem per lane; ordered model slots by an exclusive scan (as the owner's own solve phase) */
        int nvalid = 0, nullbad = 0; unsigned rixp = 0;
        if (tid < chunk) {
            const int r_ = dg_solve7_lane(P, ent->draws[tid], K->gmodels + (size_t)tid * 27, &rixp, (double *)&S->ww[wave]);
            if (r_ < 0) nullbad = 1; else nvalid = r_;
        }
        {
            unsigned v = (unsigned)nvalid, incl = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
            if (lane == 63) S->wave_cnt[wave] = incl;
            __syncthreads();
            unsigned wbase = 0;
            for (int w = 0; w < wave; w++) wbase += S->wave_cnt[w];
            const unsigned excl = wbase + incl - v;
            if (tid < chunk) {
                S->moff[tid] = (unsigned short)excl;
                S->nv[tid] = nullbad ? 255 : (unsigned char)nvalid;
                S->nsolv[tid] = (unsigned char)((rixp >> 8) & 3u);
                for (int r = 0; r < nvalid; r++) { S->ridx[tid][r] = (unsigned char)((rixp >> (2*r)) & 3u); S->mslot[excl + r] = (unsigned short)(tid * 3 + r);
                    }
            }
            if (tid == T - 1) S->moff[DG_CHUNK] = (unsigned short)(excl + v);
            __syncthreads();
        }
        const int Mtot = __builtin_amdgcn_readfirstlane((int)S->moff[DG_CHUNK]);
        {
            const int capr = (int)((sizeof(dg_lsq_scratch) / DG_NW) & ~(size_t)15);
            dg_score_chunk_F<0>(P, n, K->gmodels, S->mslot, Mtot, wave, DG_NW, kind, th, tau_c, S->ext, (char *)&S->lsq + (size_t)wave * capr, capr,
                                (double *)(K->wstage + (size_t)wave * K->n_max), K->res_I, K->res_J, lane, (unsigned *)0, 0, n < 8192);
        }
        __syncthreads();
        /* complete the entry: models per sample, the models above the bound (only such a model can be an event of the owner's commit) */
        if (tid == 0) S->itmp[24] = 0;
        __syncthreads();
        if (tid < DG_CHUNK) {
            const unsigned char nvb = tid < chunk ? S->nv[tid] : (unsigned char)0;
            ent->nv[tid] = nvb;
            if (tid < chunk && nvb != 255)
                for (int r = 0; r < (int)nvb; r++) {
                    const int mi = (int)S->moff[tid] + r;
                    const double J_ = K->res_J[mi];
                    if (J_ > tau_c) {
                        const int sl = atomicAdd(&S->itmp[24], 1);
                        if (sl < DG_STREAM_EV_MAX) {
                            dg_stream_ev *ev = &ent->ev[sl];
                            ev->J = J_; ev->I = K->res_I[mi]; ev->k = (short)tid; ev->r = (unsigned char)r; ev->pad = 0;
#pragma unroll
                            for (int j = 0; j < 9; j++) ev->model[j] = K->gmodels[(size_t)S->mslot[mi] * 9 + j];
#pragma unroll
                            for (int q = 0; q < 4; q++) ev->ridx[q] = S->ridx[tid][q];
                        }
                    }
                }
        }
        __syncthreads();
        if (tid == 0) {
            const int ne = S->itmp[24];
            ent->Mtot = Mtot; ent->n_ev = ne < DG_STREAM_EV_MAX ? ne : DG_STREAM_EV_MAX; ent->overflow = ne > DG_STREAM_EV_MAX ? 1 : 0; ent->tau_used = tau_c;
        }
        dg_stream_publish(dg_fan_flag(A, oslot, seq), seq + 1);
    }
}

/* The SAMPLER of owner slot `oslot` (worker 0 of the slot), on workspace `wsid`: the pair's whole sample stream — what the owner's own
 * sampler stages would draw (same seed, same pool, same chunking) — chunk after chunk into the ring, nothing else; the owner only
 * commits, so its local optimisations run while this workgroup keeps drawing.  The stream depends on nothing but the pair's seed and
 * its size (exp_ranF.c:1277, :1337-1342), so the sampler starts from scratch: no state crosses over.  It follows the owner's budget
 * (scb->max_sam only shrinks; what is drawn past the final budget is dropped) and its position (scb->tail: an entry is reused once
 * the owner has committed it). */
template <int T>
__device__ __noinline__ void dg_f_fan_sampler(const dg_args &A, dg_f_shared *S, const int oslot, const int wsid, int *bc /* LDS */)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    dg_stream_cb *const scb = A.scb + oslot;
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(wave) == 0) {
        int res = 0;
        const long long t0 = wall_clock64();
        for (;;) {
            if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&scb->state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == DG_ST_ATTACHED) { res = 1; break;
                }
            if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&scb->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) break;
            if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(A.done_pairs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= A.n_pairs) break;
            if (wall_clock64() - t0 > 4 * (long long)A.wait_ticks + 1000000ll) break;
            __builtin_amdgcn_s_sleep(32);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *bc = res;
    }
    __syncthreads();
    if (!*bc) return;
    __syncthreads();
    const int pair = scb->pair;
    const int n = (int)(A.offsets[pair + 1] - A.offsets[pair]);
    int *pool = (int *)(A.ws + (size_t)wsid * A.wl.stride + A.wl.off_pool);
    int *const pscr = A.pool_seq ? (int *)0 : (int *)S->ww;
    for (int i = tid; i < n; i += T) pool[i] = i;
    int max_sam = __hip_atomic_load(&scb->max_sam, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    /* srand(seed0); seed = rand(); then the first two chunks as the owner's prologue draws them */
    if (__builtin_amdgcn_readfirstlane(wave) == 0) { dg_srand_wave(&S->rng, A.seeds[pair], lane); const int v_ = dg_rand_block(&S->rng, 1, lane);
        if (lane == 0) S->itmp[31] = v_; }
    __syncthreads();
    unsigned seed = (unsigned)S->itmp[31];
    /* Pipeline, four chunks in flight: chunk i (slot f_cur: drawn ids) goes into the ring, chunk i + 1 gets its pool swaps (wave 0), chunk
     * i + 2 — whose seeds the previous step chained into sx[par] — its draws (four waves, one block of 64 samples each), chunk i + 3 its
     * seed chain (wave 1, into sx[par ^ 1]).  The chain is the long stage (one dependent step per sample): nobody polls it, and the wave
     * that shares its SIMD (wave 5: waves w and w + 4 sit on one SIMD) stays idle. */
    static_assert(sizeof(dg_lsq_scratch) >= 2 * DG_CHUNK * sizeof(unsigned), "the seed buffers of the fan sampler do not fit the least-squares scratch");
    unsigned (*sx)[DG_CHUNK] = (unsigned (*)[DG_CHUNK])&S->lsq;
    int f_ch[3] = {0, 0, 0}, f_cur = 0, f_pub = 0, f_sam = 0, cX = 0, par = 0;
    {
        int c0 = max_sam < DG_CHUNK ? max_sam : DG_CHUNK; if (c0 < 0) c0 = 0;
        int c1 = max_sam - c0; if (c1 > DG_CHUNK) c1 = DG_CHUNK; if (c1 < 0) c1 = 0;
        int c2 = max_sam - c0 - c1; if (c2 > DG_CHUNK) c2 = DG_CHUNK; if (c2 < 0) c2 = 0;
        f_ch[0] = c0; f_ch[1] = c1; cX = c2;
    }
    __syncthreads();
    if (wave == 0) {
        unsigned sd = seed;
        if (f_ch[0] > 0) sd = dg_sample_chunk<7, 0>(sd, f_ch[0], n, pool, S->seeds3[0], S->draws3[0], S->alm3[0], pscr, lane);
        if (f_ch[1] > 0) sd = dg_sample_draws<7>(sd, f_ch[1], n, S->seeds3[1], S->draws3[1], S->alm3[1], lane);
        if (cX > 0) sd = dg_sample_chain<7>(sd, cX, sx[0], lane);
        if (lane == 0) S->itmp[31] = (int)sd;
    }
    __syncthreads();
    seed = (unsigned)S->itmp[31];
    while (f_ch[f_cur] > 0 && f_sam < max_sam) {
        /* the owner's word: stop?  its budget, its position (the ring's free room): every fourth chunk (four chunks of slack in the window test) */
        __syncthreads();
        if ((f_pub & 3) != 0) { if (tid == 0) { S->itmp[24] = 1; S->itmp[26] = max_sam; } }
        else if (__builtin_amdgcn_readfirstlane(wave) == 0) {
            int go = 1;
            const long long t0 = wall_clock64();
            for (;;) {
                if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&scb->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { go = 0; break; }
                const int tl = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&scb->tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (f_pub - tl < A.stream_depth - 5) break;
                if (wall_clock64() - t0 > 4 * (long long)A.wait_ticks + 1000000ll) { go = 0; break; }
                __builtin_amdgcn_s_sleep(16);
            }
            if (lane == 0) { S->itmp[24] = go; S->itmp[26] = __hip_atomic_load(&scb->max_sam, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
        __syncthreads();
        if (!S->itmp[24]) break;
        if (S->itmp[26] < max_sam) max_sam = S->itmp[26];
        if (f_sam >= max_sam) break;
        const int fnxt = f_cur == 2 ? 0 : f_cur + 1, fnx2 = fnxt == 2 ? 0 : fnxt + 1;
        int cY = max_sam - (f_sam + f_ch[f_cur] + f_ch[fnxt] + cX); if (cY > DG_CHUNK) cY = DG_CHUNK; if (cY < 0) cY = 0;
        dg_stream_ent *fe = dg_stream_entry(A, oslot, f_pub);
        __syncthreads();
        if (wave == 0) { if (f_ch[fnxt] > 0) dg_sample_pool<7, 0>(f_ch[fnxt], n, pool, S->draws3[fnxt], S->alm3[fnxt], pscr, lane); }
        else if (wave == 1) { if (cY > 0) { const unsigned sd = dg_sample_chain<7>(seed, cY, sx[par ^ 1], lane); if (lane == 0) S->itmp[31] = (int)sd; } }
        else if (wave == 4 % DG_NW) {
            for (int i = lane; i < DG_CHUNK; i += 64) {
                fe->seeds[i] = S->seeds3[f_cur][i];
#pragma unroll
                for (int q = 0; q < 8; q++) fe->draws[i][q] = S->draws3[f_cur][i][q];
            }
            if (lane == 0) fe->cn = f_ch[f_cur];
        } else if (wave != 5) {
            const int rd = wave == 2 ? 0 : wave == 3 ? 1 : wave == 6 ? 2 : 3;                 /* (DG_NW == 8: the host only takes this mode at 512 threads) */
            if (rd * 64 < cX) {
                dg_sample_draws_round<7>(rd, cX, n, sx[par], S->draws3[fnx2], S->alm3[fnx2], lane);
                const int k_ = rd * 64 + lane; if (k_ < cX) S->seeds3[fnx2][k_] = sx[par][k_];
            }
        }
        __syncthreads();
        if (cY > 0) seed = (unsigned)S->itmp[31];
        f_sam += f_ch[f_cur]; f_pub++; f_ch[f_cur] = 0; f_ch[fnx2] = cX; cX = cY; par ^= 1; f_cur = fnxt;
        /* one release (a write-back of this XCD's L2) per four chunks, and for the first ones and the last one at once */
        if ((f_pub & 3) == 0 || f_pub <= 8 || !(f_ch[f_cur] > 0 && f_sam < max_sam)) dg_stream_publish(&scb->head, f_pub);
    }
    dg_stream_publish(&scb->head, f_pub);
}

#endif /* DG_F_FAN_H */
