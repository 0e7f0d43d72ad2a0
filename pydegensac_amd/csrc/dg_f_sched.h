/* Scheduling across workgroups: pairs that are set aside and resumed by priority (dg_args::park_sam) and the stream mode's hand-offs between an
 * owner and its producer workgroup (dg_stream_cb, dg_stream_ent).
 * Part of the fundamental-matrix kernel: included by dg_kernel_f_main.h, in this order, after dg_kernel_f.h and dg_score_tiles.h. */
#ifndef DG_F_SCHED_H
#define DG_F_SCHED_H

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
            if (wall_clock64() - t0 > limit) { if (threadIdx.x == 0) __hip_atomic_store(A.err_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v = -1;
                break; }
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
                        int e = DG_ST_REQ;
                            ok = __hip_atomic_compare_exchange_strong(&A.scb[j].state, &e, DG_ST_ATTACHED, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
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

#endif /* DG_F_SCHED_H */
