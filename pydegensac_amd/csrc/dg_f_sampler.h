/* The reference's sample stream (exp_ranF.c:1337-1342, rtools.c:12-23) as pipeline stages of one wave each: seed chain, draws, and the Fisher-Yates
 * pool swaps in their sequential, parallel (pool in LDS) and grouped (pool in the HBM workspace) forms.
 * Part of the fundamental-matrix kernel: included by dg_kernel_f_main.h, in this order, after dg_kernel_f.h and dg_score_tiles.h. */
#ifndef DG_F_SAMPLER_H
#define DG_F_SAMPLER_H

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
    /* seed chain: lane j < 31 carries the term C[NDRAW][j] * r_j, r_j = seed * 16807^j mod (2^31-1).  One step is one dependent chain
     * (121 ns in round 5: the floor of a pair that draws its whole budget): the seed stays on the scalar unit, the loop control is
     * scalar, the modular product is 32-bit arithmetic behind ONE 64-bit multiply, lane 0's own term is a select (no divergent
     * branch), only the two rows that carry terms are read back, and the seeds leave in blocks of 64 (lane j of a register keeps the
     * seed of step j: one compare-and-select per step instead of a store under a one-lane branch). */
    cn = __builtin_amdgcn_readfirstlane(cn);
    const unsigned gk = lane < 31 ? dg_rng_G[lane] : 0u, ck = lane < 31 ? dg_rng_C[NDRAW][lane] : 0u;
    unsigned sd = (unsigned)__builtin_amdgcn_readfirstlane((int)seed);
    for (int k0 = 0; k0 < cn; k0 += 64) {
        const int m = cn - k0 < 64 ? cn - k0 : 64;
        int keep = 0;
        for (int j = 0; j < m; j++) {
            keep = lane == j ? (int)sd : keep;
            const unsigned s1 = sd ? sd : 1u;                        /* rand() outputs are < 2^31: Schrage == exact mulmod */
            const unsigned long long x = (unsigned long long)s1 * gk;                           /* < 2^62 */
            unsigned t = ((unsigned)x & 0x7fffffffu) + (unsigned)(x >> 31);                     /* < 2^32, congruent mod 2^31 - 1 */
            t = (t & 0x7fffffffu) + (t >> 31);                                                  /* <= 2^31 */
            /* t >= p: t - p (the difference of a smaller t wraps to a huge value) */
            { const unsigned t2 = t - 0x7fffffffu; t = t2 < t ? t2 : t; }
            const unsigned rj = lane == 0 ? s1 : t;
            unsigned v = ck * rj;
            v += (unsigned)dg_dpp<DG_DPP_ROR(8)>((int)v);
            v += (unsigned)dg_dpp<DG_DPP_ROR(4)>((int)v);
            v += (unsigned)dg_dpp<DG_DPP_ROR(2)>((int)v);
            v += (unsigned)dg_dpp<DG_DPP_ROR(1)>((int)v);
            sd = ((unsigned)__builtin_amdgcn_readlane((int)v, 0) + (unsigned)__builtin_amdgcn_readlane((int)v, 16)) >> 1;      /* lanes 32.. carry zeros */
        }
        if (lane < m) seeds[k0 + lane] = (unsigned)keep;
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
__device__ __noinline__ void dg_sample_draws_round(int rd, int cn, int n, const unsigned *seeds, int (*draws)[8], unsigned long long *almask, int lane,
    long long *dbg = 0)
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
        for (int q = 0; q < 8; q++) { const int tau = t0 + 64 * q; if (tau < cn * NDRAW) { const int k = tau / NDRAW, i = tau - k * NDRAW;
            draws[k][i] = -1 - val[q]; } }
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
__device__ __noinline__ void dg_sample_pool_grp(int cn, int n, int *vp_, int (*draws_)[8], const unsigned long long *almask_,
    int *pscratch /* LDS, 2 * DG_PGT ints */,
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
            if (active) { __hip_atomic_fetch_max(tab + h1, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_max(tab + DG_PGT + h2, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            DG_WSYNC_LDS();          /* (the tables are LDS, accessed through LDS-qualified pointers only; the gather above stays in flight) */
            if (active) {
                const unsigned e1 = tab[h1], e2 = tab[DG_PGT + h2];
                if ((e1 >> 6) == (unsigned)s) coll = (e1 & 63u) != (unsigned)lane;
                else if ((e2 >> 6) == (unsigned)s) coll = (e2 & 63u) != (unsigned)lane;
                else coll = true;
            }
            DG_WSYNC_LDS();
            if (active) { tab[h1] = 0u; tab[DG_PGT + h2] = 0u; }
            DG_WSYNC_LDS();
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

#endif /* DG_F_SAMPLER_H */
