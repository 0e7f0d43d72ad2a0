/* Workgroup-cooperative primitives of the persistent LO-RANSAC kernels (gfx950, wave64).
 *
 * One workgroup (DG_T threads = DG_NW waves) owns one image pair.  All control flow is
 * workgroup-uniform: every thread executes the same branches on the same (LDS-broadcast or reduced)
 * values; scalar "reference-order" arithmetic is done by lane 0 between barriers.
 *
 * Canonical MSAC sum: J = sum over 64-point tiles, in tile order, of the xor-butterfly sum of the
 * tile (offsets 32,16,...,1).  Every scoring path (wave-per-model in the main loop, all-waves-per-
 * model in LO) uses this same association, so J of a model does not depend on which path scored it.
 */
#ifndef DG_WG_H
#define DG_WG_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DG_T   256
#define DG_NW  (DG_T / 64)

struct dg_pt { double x1, y1, x2, y2; };           /* 32 B per correspondence */

__device__ __forceinline__ double dg_tile_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ unsigned dg_wave_sum_u(unsigned v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double dg_wave_sum_d(double v) { return dg_tile_sum(v); }

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
    int         wantJ;  double thJ;
    /* second counter: #(d <= thC) */
    int         wantC;  double thC;
    /* ordered list of ids with d <= thL  (inlidxs' index list) */
    int        *list;   double thL;
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
    int par = 0;
    for (int base = 0; base < c.n; base += DG_T, par ^= 1) {
        int j = base + tid;
        bool act = j < c.n;
        int pid = act ? (c.src ? c.src[j] : j) : 0;
        double d = act ? err(pid, j) : 0.0;
        unsigned long long bI = 0, bC = 0, bL = 0, bF = 0;
        double tsum = 0;
        if (c.wantJ) {
            double term = 0.0;
            if (act && c.thJ != 0 && !(d >= t94)) term = 1 - (d / t94);
            tsum = dg_tile_sum(term);
            bI = __ballot(act && d <= c.thJ);
        }
        if (c.wantC) bC = __ballot(act && d <= c.thC);
        if (c.list)  bL = __ballot(act && d <= c.thL);
        if (c.flags) { bool f = act && d < c.thF; bF = __ballot(f); if (act) c.flags[j] = f ? 1 : 0; }
        if (lane == 0) {
            r->d[par][wave][0] = tsum;
            r->u[par][wave][0] = (unsigned)__popcll(bI);
            r->u[par][wave][1] = (unsigned)__popcll(bC);
            r->u[par][wave][2] = (unsigned)__popcll(bL);
            r->u[par][wave][3] = (unsigned)__popcll(bF);
        }
        __syncthreads();
        unsigned lbase = out.nL;
#pragma unroll
        for (int w = 0; w < DG_NW; w++) {
            out.J += r->d[par][w][0];
            out.I += r->u[par][w][0];
            out.C += r->u[par][w][1];
            if (w < wave) lbase += r->u[par][w][2];
            out.nL += r->u[par][w][2];
            out.nF += r->u[par][w][3];
        }
        if (c.list && act && d <= c.thL) {
            unsigned rank = (unsigned)__popcll(bL & ((1ull << lane) - 1ull));
            c.list[lbase + rank] = pid;
        }
    }
    __syncthreads();
    return out;
}

#endif /* DG_WG_H */
