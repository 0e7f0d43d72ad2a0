/* Workgroup-cooperative primitives of the persistent LO-RANSAC kernels (gfx950, wave64).
 *
 * One workgroup (DG_T threads = DG_NW waves) owns one image pair.  All control flow is
 * workgroup-uniform: every thread executes the same branches on the same (LDS-broadcast or reduced)
 * values; scalar "reference-order" arithmetic is done by lane 0 between barriers.
 *
 * MSAC sum: see dg_seq_sum below — every scoring path reproduces the reference's sequential sum, so J of a
 * model does not depend on which path or kernel variant scored it.
 */
#ifndef DG_WG_H
#define DG_WG_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef DG_T
#define DG_T   512
#endif
#define DG_NW  (DG_T / 64)

struct dg_pt { double x1, y1, x2, y2; };           /* 32 B per correspondence */

/* Load correspondence i of a point set whose placement is a compile-time property of the kernel instantiation
 * (LDSPTS == 1: LDS, else the per-pair HBM workspace): address-space-qualified, so the access is a ds_read /
 * global_load instead of a flat load through a generic pointer. */
template <int LDSPTS>
__device__ __forceinline__ dg_pt dg_ldpt(const dg_pt *P, int i)
{
    dg_pt r;
    if (LDSPTS == 1) {
        const __attribute__((address_space(3))) double *q = (const __attribute__((address_space(3))) double *)(P + i);
        r.x1 = q[0]; r.y1 = q[1]; r.x2 = q[2]; r.y2 = q[3];
    } else {
        const __attribute__((address_space(1))) double *q = (const __attribute__((address_space(1))) double *)(P + i);
        r.x1 = q[0]; r.y1 = q[1]; r.x2 = q[2]; r.y2 = q[3];
    }
    return r;
}
/* points per lane and step of the point loops: the loads of one step are issued back to back, so a pass over n points
 * costs n / (64 * DG_PU) memory round trips per wave instead of n / 64 (the loops are latency-bound, not ALU-bound) */
#define DG_PU 4

/* ---- wave reductions on the VALU (DPP row rotates + 4 readlanes), no LDS traffic -------------------
 * Row step: rotate-add by 8,4,2,1 inside each 16-lane row leaves the row total in every lane of the row
 * (bitwise the same value in all 16 lanes: each step adds two values that are equal up to commutation).
 * Then the four row totals are combined in row order ((r0+r1)+r2)+r3.  This is THE association of every
 * 64-term sum in the kernels (canonical MSAC tile sum). */
#define DG_DPP_ROR(k) (0x120 + (k))            /* DPP ctrl: row_ror:k */
template <int CTRL> __device__ __forceinline__ int dg_dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ double dg_dpp_d(double v)
{
    long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    lo = dg_dpp<CTRL>(lo); hi = dg_dpp<CTRL>(hi);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double dg_readlane_d(double v, int l)
{
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double dg_tile_sum(double v)
{
    v += dg_dpp_d<DG_DPP_ROR(8)>(v);
    v += dg_dpp_d<DG_DPP_ROR(4)>(v);
    v += dg_dpp_d<DG_DPP_ROR(2)>(v);
    v += dg_dpp_d<DG_DPP_ROR(1)>(v);
    double r0 = dg_readlane_d(v, 0), r1 = dg_readlane_d(v, 16), r2 = dg_readlane_d(v, 32), r3 = dg_readlane_d(v, 48);
    return ((r0 + r1) + r2) + r3;
}
__device__ __forceinline__ unsigned dg_wave_sum_u(unsigned v)
{
    v += (unsigned)dg_dpp<DG_DPP_ROR(8)>((int)v);
    v += (unsigned)dg_dpp<DG_DPP_ROR(4)>((int)v);
    v += (unsigned)dg_dpp<DG_DPP_ROR(2)>((int)v);
    v += (unsigned)dg_dpp<DG_DPP_ROR(1)>((int)v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 0) + (unsigned)__builtin_amdgcn_readlane((int)v, 16) +
           (unsigned)__builtin_amdgcn_readlane((int)v, 32) + (unsigned)__builtin_amdgcn_readlane((int)v, 48);
}
__device__ __forceinline__ double dg_wave_sum_d(double v) { return dg_tile_sum(v); }

/* MSAC gain J of one model = the reference's sequential fp64 sum of truncQuad terms in point order (rtools.c:160-171,
 * 228-236).  Terms that are exactly zero do not change a running sum, so every scoring path stores the nonzero terms in
 * point order and one thread adds them one after the other: bit-identical to the reference for identical residuals,
 * whatever the kernel variant or path.  The adds are one dependent chain (~2.2 ns per term on one lane); what can be
 * taken off it is the memory latency: the loads run 16 terms ahead of the adds (address-space-qualified: ds_read /
 * global_load, not flat), so a batch's latency is covered by the previous batch's adds.  Own register allocation: the
 * callers are inlined into the drivers dozens of times. */
template <int AS>
__device__ __forceinline__ double dg_seq_sum_impl(const double *t_, int cnt, double J)
{
    const __attribute__((address_space(AS))) double *t = (const __attribute__((address_space(AS))) double *)t_;
    /* every active lane gets the same count (the callers' lanes differ in WHERE their terms are, not in how many): scalar loop control */
    cnt = __builtin_amdgcn_readfirstlane(cnt);
    int k = 0;
    /* DG_SEQ_AHEAD terms per batch, two batches in flight: at most 32 loads outstanding, well inside the 6-bit vmcnt
     * (64 outstanding global loads returned wrong sums on gfx950: the counter saturates at 63) */
#define DG_SEQ_AHEAD 16
    if (cnt >= 2 * DG_SEQ_AHEAD) {
        double a[DG_SEQ_AHEAD];
#pragma unroll
        for (int i = 0; i < DG_SEQ_AHEAD; i++) a[i] = t[i];
        for (; k + 2 * DG_SEQ_AHEAD <= cnt; k += DG_SEQ_AHEAD) {
            double b[DG_SEQ_AHEAD];
#pragma unroll
            for (int i = 0; i < DG_SEQ_AHEAD; i++) b[i] = t[k + DG_SEQ_AHEAD + i];
#pragma unroll
            for (int i = 0; i < DG_SEQ_AHEAD; i++) J += a[i];
#pragma unroll
            for (int i = 0; i < DG_SEQ_AHEAD; i++) a[i] = b[i];
        }
#pragma unroll
        for (int i = 0; i < DG_SEQ_AHEAD; i++) J += a[i];
        k += DG_SEQ_AHEAD;
    }
#undef DG_SEQ_AHEAD
    /* the last (fewer than 32) terms: every load first — one latency for all of them; they used to be taken eight at a time and then one
     * at a time, a memory round trip per batch and per term (a pass's step, a fit's fill end in such a tail) — then the adds, the last
     * partial group of eight under a select */
    const int r = cnt - k;
#define DG_SEQ_TAIL(NT) do { double v[NT]; \
        _Pragma("unroll") for (int i = 0; i < NT; i++) v[i] = t[k + (i < r ? i : r - 1)]; \
        _Pragma("unroll") for (int c = 0; c < NT / 8; c++) { const int rc = r - 8 * c; \
            if (rc >= 8) { J += v[8*c]; J += v[8*c+1]; J += v[8*c+2]; J += v[8*c+3]; J += v[8*c+4]; J += v[8*c+5]; J += v[8*c+6]; J += v[8*c+7]; } \
            else if (rc > 0) { _Pragma("unroll") for (int i = 0; i < 7; i++) J = i < rc ? J + v[8*c+i] : J; } } } while (0)
    if (r > 16) DG_SEQ_TAIL(32); else if (r > 8) DG_SEQ_TAIL(16); else if (r > 0) DG_SEQ_TAIL(8);
#undef DG_SEQ_TAIL
    return J;
}
/* the same as a function of its own (own register allocation: the callers are inlined into the drivers dozens of times).  A call is also a
 * wait for EVERY outstanding memory operation of the wave (the callee's prologue is s_waitcnt 0): where the caller keeps the next step's
 * points or terms in flight across the sum — the passes of the local optimisations, the streamed least squares — it inlines
 * dg_seq_sum_impl instead (round 6). */
template <int AS>
__device__ __noinline__ double dg_seq_sum_from(const double *t_, int cnt, double J) { return dg_seq_sum_impl<AS>(t_, cnt, J); }
/* The same sum over terms in GLOBAL memory by a whole wave: 64 terms per coalesced load (the next block's load in flight), through an LDS
 * tile of `cap` doubles (a multiple of 64, at most 256), from which every lane adds them in order (dg_seq_sum_from<3>).  One L2 round
 * trip per `cap` terms instead of one per sixteen.  All 64 lanes of one wave call it together; every lane returns the sum. */
__device__ __noinline__ double dg_seq_sum_wave_g(const double *g_, int cnt, double J, double *tile_, int cap, int lane)
{
    const __attribute__((address_space(1))) double *g = (const __attribute__((address_space(1))) double *)g_;
    __attribute__((address_space(3))) double *t = (__attribute__((address_space(3))) double *)tile_;
    cnt = __builtin_amdgcn_readfirstlane(cnt); cap = __builtin_amdgcn_readfirstlane(cap);
    if (cap > 256) cap = 256;
    const int nb = cap / 64;                                   /* terms per lane and block */
    double nx[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { const int k = 64 * i + lane; nx[i] = (i < nb && k < cnt) ? g[k] : 0.0; }
    for (int k0 = 0; k0 < cnt; k0 += cap) {
        const int m = cnt - k0 < cap ? cnt - k0 : cap;
#pragma unroll
        for (int i = 0; i < 4; i++) if (i < nb) t[64 * i + lane] = nx[i];
        if (k0 + cap < cnt) {
#pragma unroll
            for (int i = 0; i < 4; i++) { const int k = k0 + cap + 64 * i + lane; nx[i] = (i < nb && k < cnt) ? g[k] : 0.0; }
        }
        DG_WSYNC_LDS();
        J = dg_seq_sum_impl<3>((const double *)tile_, m, J);
        DG_WSYNC_LDS();
    }
    return J;
}
/* terms in global memory (HBM workspace) */
__device__ __forceinline__ double dg_seq_sum(const double *t, int cnt) { return dg_seq_sum_from<1>(t, cnt, 0.0); }
/* the first min(cnt, cap) terms in LDS (jl), the rest in global memory (jg[0..cnt - cap)) */
__device__ __forceinline__ double dg_seq_sum_split(const double *jl, int cap, const double *jg, int cnt)
{
    const int nl = cnt < cap ? cnt : cap;
    double J = dg_seq_sum_from<3>(jl, nl, 0.0);
    if (cnt > nl) J = dg_seq_sum_from<1>(jg, cnt - nl, J);
    return J;
}

/* LDS block used by the reductions below (declared once per kernel) */
struct dg_red {
    double   d[2][DG_NW][4];
    unsigned u[2][DG_NW][3 * DG_PU];   /* per-wave counts of one step of dg_pass (list, J, second list) x DG_PU tiles; [0][w][0..3] also final counts */
    double   bc[96];          /* broadcast slots */
    int      bi[16];
};

/* broadcast a double / int computed by thread 0 */
__device__ __forceinline__ double dg_bcast_d(dg_red *r, double v, int tid)
{
    __syncthreads();
    if (tid == 0) r->bc[0] = v;
    __syncthreads();
    return r->bc[0];
}
__device__ __forceinline__ int dg_bcast_i(dg_red *r, int v, int tid)
{
    __syncthreads();
    if (tid == 0) r->bi[0] = v;
    __syncthreads();
    return r->bi[0];
}

/* workgroup sum of unsigned (exact) */
__device__ __forceinline__ unsigned dg_block_sum_u(dg_red *r, unsigned v, int tid)
{
    unsigned w = dg_wave_sum_u(v);
    __syncthreads();
    if ((tid & 63) == 0) r->u[0][tid >> 6][0] = w;
    __syncthreads();
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < DG_NW; i++) s += r->u[0][i][0];
    return s;
}
/* workgroup sum of doubles, fixed association (wave butterflies, then waves in order) */
__device__ __forceinline__ double dg_block_sum_d(dg_red *r, double v, int tid)
{
    double w = dg_wave_sum_d(v);
    __syncthreads();
    if ((tid & 63) == 0) r->d[0][tid >> 6][0] = w;
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int i = 0; i < DG_NW; i++) s += r->d[0][i][0];
    return s;
}

/* ------------------------------------------------------------------------------------------------
 * Generic cooperative pass over `n` items (points 0..n-1, or the ids src[0..n) when src != 0).
 * err(pid, j) returns the residual of point pid (j = its position in the pass).  Everything the reference derives from one residual
 * vector in separate loops (inlidxs at several thresholds, flag vectors, counts) is fused here.
 * ---------------------------------------------------------------------------------------------- */
struct dg_pass_cfg {
    int         n;
    const int  *src;        /* optional indirection */
    int         p0;         /* without src: item j is point p0 + j (a slice of the point set) */
    /* (I, J): I = #(d <= thJ), J = sum truncQuad(d, thJ)  (rtools.c:160-171, 228-236) */
    int         wantJ;  double thJ;  double *jbuf;   /* jbuf: >= n doubles of scratch (HBM) for the ordered nonzero MSAC terms ...;
                                                        wantJ == 2: only store the terms and count them (res.nJ), the caller sums */
    double     *jl;     int jl_cap;                  /* ... behind the first jl_cap of them, which go to this LDS buffer (0 = none)   */
    /* second counter: #(d <= thC) */
    int         wantC;  double thC;
    /* ordered list of ids with d <= thL  (inlidxs' index list) */
    int        *list;   double thL;  int listStrict;   /* listStrict: ids with d < thL instead of d <= thL */
    /* optional second ordered list of the same residuals: ids with d <= thL2 */
    int        *list2;  double thL2;
    /* flags[item position] = d < thF (strict, DegUtils.c style) and their count */
    unsigned char *flags; double thF;
};
struct dg_pass_res { unsigned I; double J; unsigned C; unsigned nL; unsigned nF; unsigned nL2; unsigned nJ; };

template <class Err>
__device__ __forceinline__ dg_pass_res dg_pass(dg_red *r, const dg_pass_cfg &c, Err err, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    dg_pass_res out; out.I = 0; out.J = 0; out.C = 0; out.nL = 0; out.nF = 0; out.nL2 = 0; out.nJ = 0;
    const double t94 = c.thJ * 9 / 4;
    unsigned cI = 0, cC = 0, cF = 0, nJ = 0;
    int par = 0;
    /* one step = DG_PU consecutive tiles of DG_T items (item j of tile u belongs to thread j - u * DG_T): the id and
     * point loads of the whole step are in flight together, and the ordered compactions of its tiles share one barrier */
    for (int base = 0; base < c.n; base += DG_PU * DG_T) {
        int jj[DG_PU], pid[DG_PU]; bool act[DG_PU]; double d[DG_PU];
#pragma unroll
        for (int u = 0; u < DG_PU; u++) { jj[u] = base + u * DG_T + tid; act[u] = jj[u] < c.n; }
#pragma unroll
        for (int u = 0; u < DG_PU; u++) pid[u] = act[u] ? (c.src ? c.src[jj[u]] : c.p0 + jj[u]) : 0;
#pragma unroll
        for (int u = 0; u < DG_PU; u++) d[u] = act[u] ? err(pid[u], jj[u]) : 0.0;
        double term[DG_PU]; bool nz[DG_PU], in[DG_PU], in2[DG_PU];
#pragma unroll
        for (int u = 0; u < DG_PU; u++) {
            term[u] = 0.0; nz[u] = false;
            if (c.wantJ) {
                if (act[u] && c.thJ != 0 && !(d[u] >= t94)) term[u] = 1 - (d[u] / t94);
                nz[u] = !(term[u] == 0.0);                          /* also true for NaN */
                cI += (act[u] && d[u] <= c.thJ) ? 1u : 0u;
            }
            if (c.wantC) cC += (act[u] && d[u] <= c.thC) ? 1u : 0u;
            if (c.flags) { bool f = act[u] && d[u] < c.thF; cF += f ? 1u : 0u; if (act[u]) c.flags[jj[u]] = f ? 1 : 0; }
            in[u] = c.list && act[u] && (c.listStrict ? d[u] < c.thL : d[u] <= c.thL);
            in2[u] = c.list2 && act[u] && d[u] <= c.thL2;
        }
        if (c.list || c.wantJ) {
            /* ordered compaction (inlier ids, nonzero MSAC terms) needs the per-wave counts of every tile: one barrier per step */
            unsigned long long bL[DG_PU], bJ[DG_PU], bL2[DG_PU];
#pragma unroll
            for (int u = 0; u < DG_PU; u++) { bL[u] = __ballot(in[u]); bJ[u] = __ballot(nz[u]); bL2[u] = c.list2 ? __ballot(in2[u]) : 0ull; }
            if (lane == 0) {
#pragma unroll
                for (int u = 0; u < DG_PU; u++) { r->u[par][wave][3*u] = (unsigned)__popcll(bL[u]); r->u[par][wave][3*u+1] = (unsigned)__popcll(bJ[u]);
                    r->u[par][wave][3*u+2] = (unsigned)__popcll(bL2[u]); }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < DG_PU; u++) {
                unsigned lbase = out.nL, jbase = nJ, l2base = out.nL2;
#pragma unroll
                for (int w = 0; w < DG_NW; w++) {
                    const unsigned a = r->u[par][w][3*u], b = r->u[par][w][3*u+1], a2 = r->u[par][w][3*u+2];
                    if (w < wave) { lbase += a; jbase += b; l2base += a2; }
                    out.nL += a; nJ += b; out.nL2 += a2;
                }
                if (in[u]) c.list[lbase + (unsigned)__popcll(bL[u] & ((1ull << lane) - 1ull))] = pid[u];
                if (nz[u]) {
                    const unsigned jx = jbase + (unsigned)__popcll(bJ[u] & ((1ull << lane) - 1ull));
                    if (jx < (unsigned)c.jl_cap) ((__attribute__((address_space(3))) double *)c.jl)[jx] = term[u];
                    else ((__attribute__((address_space(1))) double *)c.jbuf)[jx - (unsigned)c.jl_cap] = term[u];
                }
                if (in2[u]) c.list2[l2base + (unsigned)__popcll(bL2[u] & ((1ull << lane) - 1ull))] = pid[u];
            }
            par ^= 1;
        }
    }
    /* final reductions: counts, and J = the reference's own sequential sum (rtools.c:160-171 adds truncQuad(d_i) in point
     * order): the terms that are exactly 0 leave the running sum unchanged, so thread 0 adds the nonzero terms, which
     * the loop above stored in point order, one after the other — the same roundings as the reference. */
    cI = dg_wave_sum_u(cI); cC = dg_wave_sum_u(cC); cF = dg_wave_sum_u(cF);
    __syncthreads();
    if (lane == 0) { r->u[0][wave][0] = cI; r->u[0][wave][1] = cC; r->u[0][wave][3] = cF; }
    if (tid == 0 && c.wantJ == 1) r->bc[0] = dg_seq_sum_split(c.jl, c.jl_cap, c.jbuf, (int)nJ);
    __syncthreads();
#pragma unroll
    for (int w = 0; w < DG_NW; w++) { out.I += r->u[0][w][0]; out.C += r->u[0][w][1]; out.nF += r->u[0][w][3]; }
    if (c.wantJ == 1) out.J = r->bc[0];
    out.nJ = nJ;
    __syncthreads();
    return out;
}

#endif /* DG_WG_H */
