/* Tile-major screening of a chunk's hypotheses (fundamental matrix).
 *
 * A model can only matter if its MSAC gain beats tau = min(maxS.J, maxSs.J) strictly, and J <= #points with residual
 * < 9/4 th.  So before any exact scoring a model gets conservative, division-free COUNTS of a superset of those points:
 *   level 1   |r32| < thr   single precision, only the epipolar residual r against the largest possible denominator
 *                            over the pair's coordinate extents, threshold widened by a rigorous rounding bound (dg_l1_setup)
 *   level 2   r^2 < t den    double precision, the point's own denominator (dg_Fbound), threshold inflated by 1e-6
 * and only models whose count exceeds tau are scored exactly; the others get J = 0, which is never an event in the
 * commit, so decisions are unchanged.
 *
 * The counts are computed TILE-MAJOR: a wave loads a tile of correspondences once (64 lanes x DG_PU points, coalesced,
 * the next tile's loads in flight while the current one is used) and runs ALL of its models over the tile, the
 * coefficients of a model coming from an LDS table as broadcast reads.  The point set is therefore streamed once per
 * wave and chunk, not once per group of four models (round 2: one pass of the point set per group, one memory round
 * trip per 256 points and group — the scoring phase was latency-bound).  Level 1 evaluates two correspondences per
 * instruction with packed single-precision FMAs (v_pk_fma_f32); the nesting of the FMAs is the one dg_l1_setup's error
 * bound is derived for.  Lane j of the wave carries the count of the wave's j-th model.
 */
#ifndef DG_SCORE_TILES_H
#define DG_SCORE_TILES_H
#include "dg_geom.h"

typedef float dg_f2 __attribute__((ext_vector_type(2)));
typedef float dg_f4 __attribute__((ext_vector_type(4)));
typedef double dg_d2 __attribute__((ext_vector_type(2)));

#define DG_L1_ENTRY_FLOATS 12          /* Ff[0..8], thr, 2 pad: three 16-byte LDS reads per model */
#define DG_L2_ENTRY_DOUBLES 10         /* F[0..8], pad: five 16-byte LDS reads per model */

__device__ __forceinline__ dg_f2 dg_fma2(dg_f2 a, dg_f2 b, dg_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ dg_f2 dg_splat2(float v) { dg_f2 r = {v, v}; return r; }

/* Level-1 screen of one model (see the file comment): fp32 copy of the coefficients and the threshold on |r32| below
 * which a point may still be inside the 9/4 th band; +inf (everything passes) when the bound is not a normal fp32 number.
 * The Sampson denominator is at most Dmax = sum of the squared bounds |F00| X + |F10| Y + |F20| ... over the pair's
 * coordinate extents, so {d < t} is inside {r^2 < t Dmax}; for the symmetric metric d = r^2 (a + b) / (a b) >= r^2 /
 * min(a, b).  With u = 2^-24, inputs rounded to fp32 and 4 nested FMAs, |r32 - r| <= 8 u M,
 * M = X1 u1 + Y1 u2 + (|F02| X2 + |F12| Y2 + |F22|) >= sum of |terms|; 32 u M is used.  So every point with
 * r^2 < t Dmax has |r32| < sqrt(t Dmax) + 32 u M. */
__device__ __forceinline__ float dg_l1_setup(int kind, const double *f, const double *ext, double t94b, float *Ff)
{
    const double X1 = ext[0], Y1 = ext[1], X2 = ext[2], Y2 = ext[3];
    const double u1 = fabs(f[0]) * X2 + fabs(f[3]) * Y2 + fabs(f[6]), u2 = fabs(f[1]) * X2 + fabs(f[4]) * Y2 + fabs(f[7]);
    const double u3 = fabs(f[0]) * X1 + fabs(f[1]) * Y1 + fabs(f[2]), u4 = fabs(f[3]) * X1 + fabs(f[4]) * Y1 + fabs(f[5]);
    const double uw = fabs(f[2]) * X2 + fabs(f[5]) * Y2 + fabs(f[8]);
    const double am = u1*u1 + u2*u2, bm = u3*u3 + u4*u4;
    const double lim = t94b * (1.0 + 1e-9) * (kind == DG_K_FDS ? am + bm : fmin(am, bm));
    const double M = X1 * u1 + Y1 * u2 + uw;
    const double tg = sqrt(lim) + M * (32.0 / 16777216.0);
#pragma unroll
    for (int j = 0; j < 9; j++) Ff[j] = (float)f[j];
    /* unusable bound (overflow / underflow / NaN): make the level pass everything for this model */
    return (tg > 1e-30 && tg < 1e30 && M < 1e30) ? (float)tg * (1.0f + 1.1920929e-7f) : __builtin_inff();
}

/* |epipolar residual| of (x1, y1, x2, y2) under the fp32 model f_ with 4 nested FMAs (the scalar form of the level-1 screen) */
#define DG_R32(f_) fabsf(__builtin_fmaf(x1, __builtin_fmaf((f_)[0], x2, __builtin_fmaf((f_)[3], y2, (f_)[6])), \
                         __builtin_fmaf(y1, __builtin_fmaf((f_)[1], x2, __builtin_fmaf((f_)[4], y2, (f_)[7])), \
                                        __builtin_fmaf((f_)[2], x2, __builtin_fmaf((f_)[5], y2, (f_)[8])))))

/* Level-1 counts of the `nb` models whose table entries are tab[0 .. nb) over the points [p_lo, p_hi).
 * Returns, in lane j < nb, #{p : !(|r32_j(p)| >= thr_j)}. */
template <int LDSPTS>
__device__ __forceinline__ unsigned dg_l1_tile_counts(const dg_pt *P, int p_lo, int p_hi, const float *tab /* LDS */, int nb, int lane)
{
    const __attribute__((address_space(3))) dg_f4 *t4 = (const __attribute__((address_space(3))) dg_f4 *)tab;
    unsigned cnt = 0;
    p_lo = __builtin_amdgcn_readfirstlane(p_lo); p_hi = __builtin_amdgcn_readfirstlane(p_hi); nb = __builtin_amdgcn_readfirstlane(nb);
    if (p_lo >= p_hi || nb <= 0) return 0;
    dg_pt nq[DG_PU];
#pragma unroll
    for (int u = 0; u < DG_PU; u++) { const int p = p_lo + 64 * u + lane; nq[u] = dg_ldpt<LDSPTS>(P, p < p_hi ? p : p_hi - 1); }
    for (int base = p_lo; base < p_hi; base += 64 * DG_PU) {
        static_assert(DG_PU == 4, "the packed level-1 screen is written for four points per lane and step");
        const dg_f2 X1a = {(float)nq[0].x1, (float)nq[1].x1}, X1b = {(float)nq[2].x1, (float)nq[3].x1};
        const dg_f2 Y1a = {(float)nq[0].y1, (float)nq[1].y1}, Y1b = {(float)nq[2].y1, (float)nq[3].y1};
        const dg_f2 X2a = {(float)nq[0].x2, (float)nq[1].x2}, X2b = {(float)nq[2].x2, (float)nq[3].x2};
        const dg_f2 Y2a = {(float)nq[0].y2, (float)nq[1].y2}, Y2b = {(float)nq[2].y2, (float)nq[3].y2};
        unsigned long long on[DG_PU];
#pragma unroll
        for (int u = 0; u < DG_PU; u++) on[u] = __ballot(base + 64 * u + lane < p_hi);
        /* the next tile's loads go out before this tile's arithmetic */
        if (base + 64 * DG_PU < p_hi) {
#pragma unroll
            for (int u = 0; u < DG_PU; u++) { const int p = base + 64 * DG_PU + 64 * u + lane; nq[u] = dg_ldpt<LDSPTS>(P, p < p_hi ? p : p_hi - 1); }
        }
        /* the next model's coefficients are read while the current model is evaluated */
        dg_f4 c0 = t4[0], c1 = t4[1], c2 = t4[2];                              /* f0 f1 f2 f3 | f4 f5 f6 f7 | f8 thr - - */
        for (int j = 0; j < nb; j++) {
            const int jn = j + 1 < nb ? j + 1 : j;
            const dg_f4 n0 = t4[3 * jn], n1 = t4[3 * jn + 1], n2 = t4[3 * jn + 2];
            const dg_f2 f0 = dg_splat2(c0.x), f1 = dg_splat2(c0.y), f2 = dg_splat2(c0.z), f3 = dg_splat2(c0.w);
            const dg_f2 f4 = dg_splat2(c1.x), f5 = dg_splat2(c1.y), f6 = dg_splat2(c1.z), f7 = dg_splat2(c1.w), f8 = dg_splat2(c2.x);
            const float thr = c2.y;
            /* r = x1 (f0 x2 + f3 y2 + f6) + y1 (f1 x2 + f4 y2 + f7) + (f2 x2 + f5 y2 + f8), 4 nested FMAs per element */
            const dg_f2 ra = dg_fma2(X1a, dg_fma2(f0, X2a, dg_fma2(f3, Y2a, f6)), dg_fma2(Y1a, dg_fma2(f1, X2a, dg_fma2(f4, Y2a, f7)), dg_fma2(f2, X2a,
                dg_fma2(f5, Y2a, f8))));
            const dg_f2 rb = dg_fma2(X1b, dg_fma2(f0, X2b, dg_fma2(f3, Y2b, f6)), dg_fma2(Y1b, dg_fma2(f1, X2b, dg_fma2(f4, Y2b, f7)), dg_fma2(f2, X2b,
                dg_fma2(f5, Y2b, f8))));
            const unsigned c = (unsigned)__popcll(__ballot(!(fabsf(ra.x) >= thr)) & on[0]) + (unsigned)__popcll(__ballot(!(fabsf(ra.y) >= thr)) & on[1]) +
                               (unsigned)__popcll(__ballot(!(fabsf(rb.x) >= thr)) & on[2]) + (unsigned)__popcll(__ballot(!(fabsf(rb.y) >= thr)) & on[3]);
            cnt += lane == j ? c : 0u;
            c0 = n0; c1 = n1; c2 = n2;
        }
    }
    return cnt;
}

/* Level-2 counts (dg_Fbound: double precision, division-free, the point's own denominator) of the `nb` models whose
 * coefficients are tab[0 .. nb) (DG_L2_ENTRY_DOUBLES doubles each) over the points [p_lo, p_hi); lane j < nb gets its count. */
template <int LDSPTS>
__device__ __forceinline__ unsigned dg_l2_tile_counts(const dg_pt *P, int p_lo, int p_hi, const double *tab /* LDS */, int nb, int kind, double t94b, int lane)
{
    const __attribute__((address_space(3))) dg_d2 *t2 = (const __attribute__((address_space(3))) dg_d2 *)tab;
    unsigned cnt = 0;
    p_lo = __builtin_amdgcn_readfirstlane(p_lo); p_hi = __builtin_amdgcn_readfirstlane(p_hi); nb = __builtin_amdgcn_readfirstlane(nb);
    if (p_lo >= p_hi || nb <= 0) return 0;
    dg_pt nq[DG_PU];
#pragma unroll
    for (int u = 0; u < DG_PU; u++) { const int p = p_lo + 64 * u + lane; nq[u] = dg_ldpt<LDSPTS>(P, p < p_hi ? p : p_hi - 1); }
    for (int base = p_lo; base < p_hi; base += 64 * DG_PU) {
        dg_pt q[DG_PU]; unsigned long long on[DG_PU];
#pragma unroll
        for (int u = 0; u < DG_PU; u++) { q[u] = nq[u]; on[u] = __ballot(base + 64 * u + lane < p_hi); }
        if (base + 64 * DG_PU < p_hi) {
#pragma unroll
            for (int u = 0; u < DG_PU; u++) { const int p = base + 64 * DG_PU + 64 * u + lane; nq[u] = dg_ldpt<LDSPTS>(P, p < p_hi ? p : p_hi - 1); }
        }
        for (int j = 0; j < nb; j++) {
            const dg_d2 a0 = t2[5 * j], a1 = t2[5 * j + 1], a2 = t2[5 * j + 2], a3 = t2[5 * j + 3], a4 = t2[5 * j + 4];
            const double F[9] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y, a3.x, a3.y, a4.x};
            unsigned c = 0;
#pragma unroll
            for (int u = 0; u < DG_PU; u++) c += (unsigned)__popcll(__ballot(dg_Fbound(kind, F, q[u], t94b) != 0u) & on[u]);
            cnt += lane == j ? c : 0u;
        }
    }
    return cnt;
}

#endif /* DG_SCORE_TILES_H */
