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
 * whatever the kernel variant or path.  Eight independent loads are kept in flight; the adds stay strictly ordered. */
__device__ __forceinline__ double dg_seq_sum(const double *t, int cnt)
{
    double J = 0.0; int k = 0;
    for (; k + 8 <= cnt; k += 8) {
        const double v0 = t[k], v1 = t[k+1], v2 = t[k+2], v3 = t[k+3], v4 = t[k+4], v5 = t[k+5], v6 = t[k+6], v7 = t[k+7];
        J += v0; J += v1; J += v2; J += v3; J += v4; J += v5; J += v6; J += v7;
    }
    for (; k < cnt; k++) J += t[k];
    return J;
}

/* LDS block used by the reductions below (declared once per kernel) */
struct dg_red {
    double   d[2][DG_NW][4];
    unsigned u[2][DG_NW][4];
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
    /* (I, J): I = #(d <= thJ), J = sum truncQuad(d, thJ)  (rtools.c:160-171, 228-236) */
    int         wantJ;  double thJ;  double *jbuf;   /* jbuf: >= n doubles of scratch (HBM) for the ordered nonzero MSAC terms */
    /* second counter: #(d <= thC) */
    int         wantC;  double thC;
    /* ordered list of ids with d <= thL  (inlidxs' index list) */
    int        *list;   double thL;  int listStrict;   /* listStrict: ids with d < thL instead of d <= thL */
    /* flags[item position] = d < thF (strict, DegUtils.c style) and their count */
    unsigned char *flags; double thF;
};
struct dg_pass_res { unsigned I; double J; unsigned C; unsigned nL; unsigned nF; };

template <class Err>
__device__ __forceinline__ dg_pass_res dg_pass(dg_red *r, const dg_pass_cfg &c, Err err, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    dg_pass_res out; out.I = 0; out.J = 0; out.C = 0; out.nL = 0; out.nF = 0;
    const double t94 = c.thJ * 9 / 4;
    unsigned cI = 0, cC = 0, cF = 0, nJ = 0;
    int par = 0;
    for (int base = 0; base < c.n; base += DG_T) {
        int j = base + tid;
        bool act = j < c.n;
        int pid = act ? (c.src ? c.src[j] : j) : 0;
        double d = act ? err(pid, j) : 0.0;
        double term = 0.0; bool nz = false;
        if (c.wantJ) {
            if (act && c.thJ != 0 && !(d >= t94)) term = 1 - (d / t94);
            nz = !(term == 0.0);                                   /* also true for NaN */
            cI += (act && d <= c.thJ) ? 1u : 0u;
        }
        if (c.wantC) cC += (act && d <= c.thC) ? 1u : 0u;
        if (c.flags) { bool f = act && d < c.thF; cF += f ? 1u : 0u; if (act) c.flags[j] = f ? 1 : 0; }
        if (c.list || c.wantJ) {
            /* ordered compaction (inlier ids, nonzero MSAC terms) needs the block's per-wave counts: one barrier per DG_T items */
            bool in = c.list && act && (c.listStrict ? d < c.thL : d <= c.thL);
            unsigned long long bL = __ballot(in), bJ = __ballot(nz);
            if (lane == 0) { r->u[par][wave][2] = (unsigned)__popcll(bL); r->u[par][wave][3] = (unsigned)__popcll(bJ); }
            __syncthreads();
            unsigned lbase = out.nL, jbase = nJ;
#pragma unroll
            for (int w = 0; w < DG_NW; w++) {
                if (w < wave) { lbase += r->u[par][w][2]; jbase += r->u[par][w][3]; }
                out.nL += r->u[par][w][2]; nJ += r->u[par][w][3];
            }
            if (in) c.list[lbase + (unsigned)__popcll(bL & ((1ull << lane) - 1ull))] = pid;
            if (nz) c.jbuf[jbase + (unsigned)__popcll(bJ & ((1ull << lane) - 1ull))] = term;
            par ^= 1;
        }
    }
    /* final reductions: counts, and J = the reference's own sequential sum (rtools.c:160-171 adds truncQuad(d_i) in point
     * order): the terms that are exactly 0 leave the running sum unchanged, so thread 0 adds the nonzero terms, which
     * the loop above stored in point order, one after the other — the same roundings as the reference. */
    cI = dg_wave_sum_u(cI); cC = dg_wave_sum_u(cC); cF = dg_wave_sum_u(cF);
    __syncthreads();
    if (lane == 0) { r->u[0][wave][0] = cI; r->u[0][wave][1] = cC; r->u[0][wave][3] = cF; }
    if (tid == 0 && c.wantJ) r->bc[0] = dg_seq_sum(c.jbuf, (int)nJ);
    __syncthreads();
#pragma unroll
    for (int w = 0; w < DG_NW; w++) { out.I += r->u[0][w][0]; out.C += r->u[0][w][1]; out.nF += r->u[0][w][3]; }
    if (c.wantJ) out.J = r->bc[0];
    __syncthreads();
    return out;
}

#endif /* DG_WG_H */
