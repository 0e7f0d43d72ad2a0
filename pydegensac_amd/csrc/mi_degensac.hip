/* libmi_degensac.so — host side of the C-ABI (include/mi_degensac.h) and kernel instantiations.
 * gfx950 only.  No CPU fallback: every entry point needs a HIP device. */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>
#include <atomic>
#include <chrono>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>
#include "../../include/mi_degensac.h"
#define DG_T 512
#include "dg_kernel_f_main.h"
#include "dg_kernel_h.h"
#include "dg_kernel_h2el.h"
#include "dg_variant_impl.h"
#include "mi_degensac_host.inc"

/* device buffer of a unit-level (test) entry point */
template <class T> struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { if (hipMalloc((void **)&p, (n ? n : 1) * sizeof(T)) == hipSuccess) return 0; (void)hipGetLastError(); return MI_DEGENSAC_ENOMEM; }
};
#define DG_UNIT_ENTER(device) DevGuard g_; { int rc_ = (mi_degensac_device_count() == 0) ? (set_err("no HIP device: this library has no CPU path"), \
    MI_DEGENSAC_ENODEV) : g_.enter(device); if (rc_) return rc_; rc_ = dev_init(device); if (rc_) return rc_; }

/* ---- unit-level kernels --------------------------------------------------------------------------- */
__global__ void dg_score_models_kernel(const double *p1, const double *p2, int n, int dim, const double *models, int n_models,
                                       int kind, double th, unsigned *Iout, double *Jout, double *resid)
{
    /* one wave per model, the same residual code as the main kernels' scoring phase */
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mi = blockIdx.x * (blockDim.x >> 6) + wave;
    if (mi >= n_models) return;
    double M[9], Hinv[9], H1[9];
    for (int j = 0; j < 9; j++) { M[j] = models[(size_t)mi * 9 + j]; Hinv[j] = 0; H1[j] = 0; }
    if (kind > 10) dg_hsym_prepare(M, Hinv, H1);
    /* J = the reference's sequential sum: the wave walks the points in order; the nonzero terms of each tile are added
     * lane after lane (no scratch buffer in this unit kernel) */
    unsigned cI = 0; const double t94 = th * 9 / 4; double J = 0.0;
    for (int base = 0; base < n; base += 64) {
        int p = base + lane; bool act = p < n; double d = 0;
        if (act) {
            dg_pt q; q.x1 = p1[(size_t)p * dim]; q.y1 = p1[(size_t)p * dim + 1]; q.x2 = p2[(size_t)p * dim]; q.y2 = p2[(size_t)p * dim + 1];
            if (kind < 10) d = dg_Ferr(kind, M, q); else d = dg_Herr(kind - 10, M, Hinv, H1, q);
            if (resid) resid[(size_t)mi * n + p] = d;
        }
        double term = 0.0;
        if (act && th != 0 && !(d >= t94)) term = 1 - (d / t94);
        cI += (act && d <= th) ? 1u : 0u;
        unsigned long long m = __ballot(!(term == 0.0));
        while (m) { const int l = __ffsll((long long)m) - 1; J += dg_rdl_d(term, l); m &= m - 1; }
    }
    unsigned I = dg_wave_sum_u(cI);
    if (lane == 0) { Iout[mi] = I; Jout[mi] = J; }
}

extern "C" int mi_degensac_score_models(const double *pts1, const double *pts2, int n, int dim, const double *models, int n_models,
        int kind, double th, int device, uint32_t *I, double *J, double *resid)
{
    DG_UNIT_ENTER(device);
    if (!(kind == 0 || kind == 1 || kind == 2 || (kind >= 10 && kind <= 14))) { set_err("unsupported metric kind"); return MI_DEGENSAC_EINVAL; }
    DevBuf<double> d1, d2, dm, dJ, dr; DevBuf<uint32_t> dI;
    if (d1.alloc((size_t)n * dim) || d2.alloc((size_t)n * dim) || dm.alloc((size_t)n_models * 9) || dJ.alloc(n_models) || dI.alloc(n_models) ||
        (resid && dr.alloc((size_t)n_models * n))) { set_err("device allocation failed"); return MI_DEGENSAC_ENOMEM; }
    HIPCHK(hipMemcpy(d1.p, pts1, (size_t)n * dim * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d2.p, pts2, (size_t)n * dim * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dm.p, models, (size_t)n_models * 72, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(dg_score_models_kernel, dim3((n_models + 3) / 4), dim3(256), 0, 0, d1.p, d2.p, n, dim, dm.p, n_models, kind, th, dI.p, dJ.p,
        resid ? dr.p : nullptr);
    HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(I, dI.p, (size_t)n_models * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(J, dJ.p, (size_t)n_models * 8, hipMemcpyDeviceToHost));
    if (resid) HIPCHK(hipMemcpy(resid, dr.p, (size_t)n_models * n * 8, hipMemcpyDeviceToHost));
    return 0;
}

/* ---- unit level: screening counts ------------------------------------------- */
__global__ void dg_screen_counts_kernel(const dg_pt *P, int n, const double *models, int n_models, int kind, double th,
                                        double e0, double e1, double e2, double e3, unsigned *c1, unsigned *c2)
{
    __shared__ __attribute__((aligned(16))) float tab1[64 * DG_L1_ENTRY_FLOATS];
    __shared__ __attribute__((aligned(16))) double tab2[64 * DG_L2_ENTRY_DOUBLES];
    const int lane = threadIdx.x, m = blockIdx.x * 64 + lane;
    const int nb = n_models - blockIdx.x * 64 < 64 ? n_models - blockIdx.x * 64 : 64;
    const double t94b = th * 9 / 4 * (1.0 + 1e-6), ext[4] = {e0, e1, e2, e3};
    if (lane < nb) {
        double F[9]; float Ff[9];
        for (int j = 0; j < 9; j++) F[j] = models[(size_t)m * 9 + j];
        const float thr = dg_l1_setup(kind, F, ext, t94b, Ff);
        for (int j = 0; j < 9; j++) { tab1[lane * DG_L1_ENTRY_FLOATS + j] = Ff[j]; tab2[lane * DG_L2_ENTRY_DOUBLES + j] = F[j]; }
        tab1[lane * DG_L1_ENTRY_FLOATS + 9] = thr; tab1[lane * DG_L1_ENTRY_FLOATS + 10] = 0.f; tab1[lane * DG_L1_ENTRY_FLOATS + 11] = 0.f;
        tab2[lane * DG_L2_ENTRY_DOUBLES + 9] = 0.;
    }
    DG_WSYNC();
    const unsigned a = dg_l1_tile_counts<0>(P, 0, n, tab1, nb, lane);
    const unsigned b = dg_l2_tile_counts<0>(P, 0, n, tab2, nb, kind, t94b, lane);
    if (lane < nb) { c1[m] = a; c2[m] = b; }
}

extern "C" int mi_degensac_screen_counts(const double *pts1, const double *pts2, int n, int dim, const double *models, int n_models,
        int kind, double th, int device, uint32_t *c1, uint32_t *c2)
{
    DG_UNIT_ENTER(device);
    if (!(kind == 0 || kind == 1) || n <= 0 || n_models <= 0 || (dim != 2 && dim != 6)) { set_err("bad argument"); return MI_DEGENSAC_EINVAL; }
    std::vector<dg_pt> hp((size_t)n);
    double ext[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        dg_pt q; q.x1 = pts1[(size_t)i * dim]; q.y1 = pts1[(size_t)i * dim + 1]; q.x2 = pts2[(size_t)i * dim]; q.y2 = pts2[(size_t)i * dim + 1];
        hp[i] = q;
        ext[0] = fmax(ext[0], fabs(q.x1)); ext[1] = fmax(ext[1], fabs(q.y1)); ext[2] = fmax(ext[2], fabs(q.x2)); ext[3] = fmax(ext[3], fabs(q.y2));
    }
    DevBuf<dg_pt> dp; DevBuf<double> dm; DevBuf<uint32_t> d1, d2;
    if (dp.alloc(n) || dm.alloc((size_t)n_models * 9) || d1.alloc(n_models) || d2.alloc(n_models)) { set_err("device allocation failed");
        return MI_DEGENSAC_ENOMEM; }
    HIPCHK(hipMemcpy(dp.p, hp.data(), (size_t)n * sizeof(dg_pt), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dm.p, models, (size_t)n_models * 72, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(dg_screen_counts_kernel, dim3((n_models + 63) / 64), dim3(64), 0, 0, dp.p, n, dm.p, n_models, kind, th, ext[0], ext[1], ext[2], ext[3],
        d1.p, d2.p);
    HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(c1, d1.p, (size_t)n_models * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(c2, d2.p, (size_t)n_models * 4, hipMemcpyDeviceToHost));
    return 0;
}

/* ---- unit level: the homography main loop's screen (dg_HDs_maybe_below / dg_h_screen4, gfx950 build) -------------------- */
__global__ void dg_screen_counts_h_kernel(const dg_pt *P, int n, const double *gm /* 18 doubles per model */, int n_models, double tb,
                                          unsigned *cnt, unsigned char *cand)
{
    const int lane = threadIdx.x, g0 = blockIdx.x * 4;
    const dg_u4 c4 = dg_h_screen4<0>(P, n, gm, g0, 1, n_models, tb, lane);       /* the sweep the kernel runs: four models per wave */
    if (lane < 4 && g0 + lane < n_models) cnt[g0 + lane] = c4.v[lane];
    if (cand)                                                                     /* ... and the per-point verdicts of the same bound */
        for (int j = 0; j < 4 && g0 + j < n_models; j++)
            for (int i = lane; i < n; i += 64) {
                const dg_pt q = P[i];
                cand[(size_t)(g0 + j) * n + i] = dg_HDs_maybe_below(gm + (size_t)(g0 + j) * 18, q.x1, q.y1, q.x2, q.y2, tb) ? 1 : 0;
            }
}

extern "C" int mi_degensac_screen_counts_h(const double *pts1, const double *pts2, int n, int dim, const double *models, int n_models,
        double th, int device, uint32_t *cnt, uint8_t *cand)
{
    DG_UNIT_ENTER(device);
    if (n <= 0 || n_models <= 0 || (dim != 2 && dim != 6) || !cnt) { set_err("bad argument"); return MI_DEGENSAC_EINVAL; }
    std::vector<dg_pt> hp((size_t)n); std::vector<double> hm((size_t)n_models * 18, 0.0);
    for (int i = 0; i < n; i++) { dg_pt q; q.x1 = pts1[(size_t)i * dim]; q.y1 = pts1[(size_t)i * dim + 1]; q.x2 = pts2[(size_t)i * dim];
        q.y2 = pts2[(size_t)i * dim + 1]; hp[i] = q; }
    for (int m = 0; m < n_models; m++) for (int j = 0; j < 9; j++) hm[(size_t)m * 18 + j] = models[(size_t)m * 9 + j];
    DevBuf<dg_pt> dp; DevBuf<double> dm; DevBuf<uint32_t> dc; DevBuf<uint8_t> dk;
    if (dp.alloc(n) || dm.alloc((size_t)n_models * 18) || dc.alloc(n_models) || (cand && dk.alloc((size_t)n_models * n))) { set_err("device allocation failed");
        return MI_DEGENSAC_ENOMEM; }
    HIPCHK(hipMemcpy(dp.p, hp.data(), (size_t)n * sizeof(dg_pt), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dm.p, hm.data(), hm.size() * 8, hipMemcpyHostToDevice));
    const double tb = (th * 9 / 4) * (1.0 + 1e-6);            /* the kernel's own bound (dg_kernel_h.h, main loop) */
    hipLaunchKernelGGL(dg_screen_counts_h_kernel, dim3((n_models + 3) / 4), dim3(64), 0, 0, dp.p, n, dm.p, n_models, tb, dc.p, cand ? dk.p : nullptr);
    HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(cnt, dc.p, (size_t)n_models * 4, hipMemcpyDeviceToHost));
    if (cand) HIPCHK(hipMemcpy(cand, dk.p, (size_t)n_models * n, hipMemcpyDeviceToHost));
    return 0;
}

/* ---- unit level: the wave forms of srand / rand (dg_srand_wave, dg_rand_skip, dg_rand_block) -------------------------------- */
__global__ void dg_rng_wave_kernel(unsigned seed, int skip, int block, int count, int *out)
{
    __shared__ dg_rng g;
    const int lane = threadIdx.x;
    dg_srand_wave(&g, seed, lane);
    dg_rand_skip(&g, skip, lane);
    for (int q0 = 0; q0 < count; q0 += block) {
        const int m = count - q0 < block ? count - q0 : block;
        const int v = dg_rand_block(&g, m, lane);
        if (lane < m) out[q0 + lane] = v;
    }
}
extern "C" int mi_degensac_rng_wave(uint32_t seed, int skip, int block, int count, int device, int32_t *out)
{
    DG_UNIT_ENTER(device);
    if (block < 1 || block > 31 || count < 0 || skip < 0 || !out) { set_err("bad argument"); return MI_DEGENSAC_EINVAL; }
    DevBuf<int> d; if (d.alloc((size_t)count)) { set_err("device allocation failed"); return MI_DEGENSAC_ENOMEM; }
    hipLaunchKernelGGL(dg_rng_wave_kernel, dim3(1), dim3(64), 0, 0, seed, skip, block, count, d.p);
    HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, d.p, (size_t)count * 4, hipMemcpyDeviceToHost));
    return 0;
}

__global__ void dg_sample_stream_kernel(unsigned seed0, int n, int ssz, int iters, int seq_pool, int *pool_g, int *out)
{
    /* the main kernels' sampler (dg_sample_chunk), chunk by chunk, run by one wave: with the pool in LDS (n <= 4096:
     * the parallel pool stage) or in global memory (the sequential one) */
    __shared__ unsigned seeds[DG_CHUNK]; __shared__ int draws[DG_CHUNK][8]; __shared__ dg_rng g; __shared__ unsigned sd0;
        __shared__ unsigned long long alm[DG_CHUNK / 64];
    __shared__ int pool_l[4096]; __shared__ int scratch[2 * DG_CHUNK * 7];
    const int lane = threadIdx.x;
    const bool lds = n <= 4096;
    int *pool = lds ? pool_l : pool_g;
    int *const pscr = seq_pool ? (int *)0 : scratch;
    for (int i = lane; i < n; i += 64) pool[i] = i;
    if (lane == 0) { dg_srand(&g, seed0); sd0 = (unsigned)dg_rand(&g); }
    __syncthreads();
    unsigned seed = sd0;
    for (int base = 0; base < iters; base += DG_CHUNK) {
        int chunk = iters - base; if (chunk > DG_CHUNK) chunk = DG_CHUNK;
        if (lds) seed = ssz == 7 ? dg_sample_chunk<7, 2>(seed, chunk, n, pool, seeds, draws, alm, pscr, lane) : dg_sample_chunk<4, 2>(seed, chunk, n, pool,
            seeds, draws, alm, pscr, lane);
        else     seed = ssz == 7 ? dg_sample_chunk<7, 0>(seed, chunk, n, pool, seeds, draws, alm, 0, lane) : dg_sample_chunk<4, 0>(seed, chunk, n, pool, seeds,
            draws, alm, 0, lane);
        __syncthreads();
        for (int k = lane; k < chunk; k += 64) for (int i = 0; i < ssz; i++) out[(size_t)(base + k) * ssz + i] = draws[k][i];
        __syncthreads();
    }
}

extern "C" int mi_degensac_sample_stream_ex(uint32_t seed, int n, int sample_size, int iters, int device, int seq_pool, int32_t *samples)
{
    DG_UNIT_ENTER(device);
    if (!g_dev[device].pool_par_ok) seq_pool = 1;
    if ((sample_size != 4 && sample_size != 7) || n < sample_size + 1) { set_err("bad sample size"); return MI_DEGENSAC_EINVAL; }
    DevBuf<int> dpool, dout;
    if (dpool.alloc(n) || dout.alloc((size_t)iters * sample_size)) { set_err("device allocation failed"); return MI_DEGENSAC_ENOMEM; }
    hipLaunchKernelGGL(dg_sample_stream_kernel, dim3(1), dim3(64), 0, 0, seed, n, sample_size, iters, seq_pool, dpool.p, dout.p);
    HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(samples, dout.p, (size_t)iters * sample_size * 4, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int mi_degensac_sample_stream(uint32_t seed, int n, int sample_size, int iters, int device, int32_t *samples)
{ return mi_degensac_sample_stream_ex(seed, n, sample_size, iters, device, 0, samples); }

__global__ void dg_solve7_kernel(const double *p1, const double *p2, int dim, const int *samples, int n_samples, int *nsol, int *ridx, double *models)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_samples) return;
    dg_pt sp[7]; double m[7][9];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        int id = samples[(size_t)t * 7 + i];
        sp[i].x1 = p1[(size_t)id * dim]; sp[i].y1 = p1[(size_t)id * dim + 1]; sp[i].x2 = p2[(size_t)id * dim]; sp[i].y2 = p2[(size_t)id * dim + 1];
        double a[3] = {sp[i].x1, sp[i].y1, 1.0}, b[3] = {sp[i].x2, sp[i].y2, 1.0};
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int l = 0; l < 3; l++) m[i][3*k+l] = b[k] * a[l];
    }
    double f1[9], f2[9]; int nv = 0;
    int ok = dg_gj7(m, f1, f2);
    __shared__ double wscr[81];                  /* one wave per block: the general elimination, one lane at a time */
    for (unsigned long long need = __ballot(!ok); need; need &= need - 1) {
        if ((int)(threadIdx.x & 63) != __ffsll((long long)need) - 1) continue;
        for (int i = 0; i < 7; i++) { double a[3] = {sp[i].x1, sp[i].y1, 1.0}, b[3] = {sp[i].x2, sp[i].y2, 1.0};
            for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++) wscr[9*i+3*k+l] = b[k] * a[l]; }
        if (dg_null9<7, 2>(wscr, wscr + 63) == 2) { for (int i = 0; i < 9; i++) { f1[i] = wscr[63+i]; f2[i] = wscr[72+i]; } ok = 1; }
    }
    if (ok) {
        double poly[4], roots[3];
        dg_slcm(f1, f2, poly);
        int ns = dg_rroots3(poly, roots);
        for (int i = 0; i < ns; i++) {
            double f[9];
            for (int j = 0; j < 9; j++) f[j] = f1[j] * roots[i] + f2[j] * (1 - roots[i]);
            if (!dg_ori_valid7(f, sp)) continue;
            for (int j = 0; j < 9; j++) models[(size_t)t * 27 + nv * 9 + j] = f[j];
            ridx[(size_t)t * 3 + nv] = i; nv++;
        }
    } else nv = -1;
    nsol[t] = nv;
}

extern "C" int mi_degensac_solve7(const double *pts1, const double *pts2, int n, int dim, const int32_t *samples, int n_samples, int device,
        int32_t *nsol, int32_t *root_idx, double *models)
{
    DG_UNIT_ENTER(device);
    DevBuf<double> d1, d2, dm; DevBuf<int> ds, dn, dr;
    if (d1.alloc((size_t)n * dim) || d2.alloc((size_t)n * dim) || dm.alloc((size_t)n_samples * 27) || ds.alloc((size_t)n_samples * 7) || dn.alloc(n_samples) ||
        dr.alloc((size_t)n_samples * 3))
    { set_err("device allocation failed"); return MI_DEGENSAC_ENOMEM; }
    HIPCHK(hipMemcpy(d1.p, pts1, (size_t)n * dim * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d2.p, pts2, (size_t)n * dim * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ds.p, samples, (size_t)n_samples * 28, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dm.p, 0, (size_t)n_samples * 27 * 8)); HIPCHK(hipMemset(dr.p, 0, (size_t)n_samples * 12));
    hipLaunchKernelGGL(dg_solve7_kernel, dim3((n_samples + 63) / 64), dim3(64), 0, 0, d1.p, d2.p, dim, ds.p, n_samples, dn.p, dr.p, dm.p);
    HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(nsol, dn.p, (size_t)n_samples * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(root_idx, dr.p, (size_t)n_samples * 12, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(models, dm.p, (size_t)n_samples * 27 * 8, hipMemcpyDeviceToHost));
    return 0;
}


/* one problem per lane through the lane-level 3x3 routines of dg_mat3.h (op 0: inverse, 1: right singular vectors,
 * 2: Hdetect) */
__global__ void dg_mat3_kernel(int op, const double *in, int count, double *out, int *flag)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    if (op == 0) {
        double a[9]; for (int i = 0; i < 9; i++) a[i] = in[(size_t)t * 9 + i];
        flag[t] = dg_inv3(a);
        for (int i = 0; i < 9; i++) out[(size_t)t * 9 + i] = a[i];
    } else if (op == 1) {
        double a[9], v[9], d[3]; for (int i = 0; i < 9; i++) a[i] = in[(size_t)t * 9 + i];
        dg_svd3_right(a, v, d);
        for (int i = 0; i < 9; i++) out[(size_t)t * 12 + i] = v[i];
        for (int i = 0; i < 3; i++) out[(size_t)t * 12 + 9 + i] = d[i];
        flag[t] = 0;
    } else if (op == 4) {
        /* the cubic's real roots as the 7-point solver takes them (Ftools.c:251-298): the one place where the device's math
         * library (pow / acos / cos) stands in for the host's */
        double po[4], r[3] = {0, 0, 0}; for (int i = 0; i < 4; i++) po[i] = in[(size_t)t * 4 + i];
        flag[t] = dg_rroots3(po, r);
        for (int i = 0; i < 3; i++) out[(size_t)t * 3 + i] = r[i];
    } else {
        /* in: F (9), seven correspondences x1 y1 x2 y2 (28), triplet (3, as doubles) */
        double F[9], u7[7][4]; unsigned char ids[3];
        const double *q = in + (size_t)t * 40;
        for (int i = 0; i < 9; i++) F[i] = q[i];
        for (int i = 0; i < 7; i++) for (int j = 0; j < 4; j++) u7[i][j] = q[9 + 4*i + j];
        for (int i = 0; i < 3; i++) ids[i] = (unsigned char)q[37 + i];
        double H[9]; dg_Hdetect(F, u7, ids, H);
        for (int i = 0; i < 9; i++) out[(size_t)t * 9 + i] = H[i];
        flag[t] = 0;
    }
}

/* op 3: the wave eigen-solver (dsyev restated; dg_dev_small.h dg_eig_sym_wave), one symmetric 9x9 problem per wave.
 * out: the nine eigenvalues as the solver leaves them (smallest first, the rest unordered) + the 81 entries of the
 * matrix it leaves behind (column 0 = the eigenvector of the smallest eigenvalue, which is all the estimator reads) */
__global__ void dg_eig9_kernel(const double *in, int count, double *out, int *flag)
{
    __shared__ double a[81], w[9]; __shared__ dg_eig_ws ews;
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= count) return;
    for (int i = lane; i < 81; i += 64) a[i] = in[(size_t)t * 81 + i];
    DG_WSYNC();
    const int info = dg_eig_sym_wave(a, w, lane, &ews);
    DG_WSYNC();
    if (lane < 9) out[(size_t)t * 90 + lane] = w[lane];
    for (int i = lane; i < 81; i += 64) out[(size_t)t * 90 + 9 + i] = a[i];
    if (lane == 0) flag[t] = info;
}

/* op 5: two problems per wave (dg_eig2.h dg_eig_sym_wave2): block t solves problems 2t (lanes 0..31) and 2t + 1 (lanes 32..63; the last
 * block of an odd count solves its one problem twice).  Output layout as op 3.
 * op 6 / 7: timing of op 3 / op 5 — every block solves its problem(s) `reps` (= flag[0] on entry, via `count`'s high bits: see the host
 * side) times from fresh copies and writes the ticks of the 100 MHz clock it took to out[t] */
__global__ void dg_eig9x2_kernel(const double *in, int count, double *out, int *flag, int reps /* 0 = results, else timing */, int two)
{
    __shared__ double a[2][81], w[2][9]; __shared__ dg_eig_ws ews[2];
    const int t = blockIdx.x, lane = threadIdx.x;
    const int p0 = two ? 2 * t : t, p1 = two ? (2 * t + 1 < count ? 2 * t + 1 : 2 * t) : t;
    if (p0 >= count) return;
    int info = 0;
    const long long t0 = wall_clock64();
    for (int r = 0; r < (reps > 0 ? reps : 1); r++) {
        for (int i = lane; i < 81; i += 64) { a[0][i] = in[(size_t)p0 * 81 + i]; a[1][i] = in[(size_t)p1 * 81 + i]; }
        DG_WSYNC();
        if (two) info = dg_eig_sym_wave2(a[0], w[0], &ews[0], a[1], w[1], &ews[1], lane);
        else info = dg_eig_sym_wave(a[0], w[0], lane, &ews[0]);
        DG_WSYNC();
    }
    const long long t1 = wall_clock64();
    if (reps > 0) { if (lane == 0) out[t] = (double)(t1 - t0); return; }
    for (int h = 0; h < (two ? 2 : 1); h++) {
        const int p = h ? p1 : p0;
        if (lane < 9) out[(size_t)p * 90 + lane] = w[h][lane];
        for (int i = lane; i < 81; i += 64) out[(size_t)p * 90 + 9 + i] = a[h][i];
    }
    if (lane == 0) flag[p0] = info;
    if (lane == 32 && two) flag[p1] = info;
}

extern "C" int mi_degensac_mat3(int op, const double *in, int count, int device, double *out, int32_t *flag)
{
    DG_UNIT_ENTER(device);
    if (op < 0 || op > 7 || count < 0) { set_err("bad op"); return MI_DEGENSAC_EINVAL; }
    const int reps = op >= 6 ? (flag[0] > 0 ? flag[0] : 1) : 0;            /* ops 6 / 7: repetitions per block, passed in flag[0] */
    const size_t ni = op == 4 ? 4 : (op == 3 || op >= 5) ? 81 : op == 2 ? 40 : 9, no = op == 4 ? 3 : (op == 3 || op >= 5) ? 90 : op == 1 ? 12 : 9;
    DevBuf<double> di, dout; DevBuf<int> df;
    if (di.alloc(count * ni) || dout.alloc(count * no) || df.alloc(count)) { set_err("device allocation failed"); return MI_DEGENSAC_ENOMEM; }
    HIPCHK(hipMemcpy(di.p, in, count * ni * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(df.p, 0, (size_t)count * sizeof(int)));
    if (count && op == 3) hipLaunchKernelGGL(dg_eig9_kernel, dim3(count), dim3(64), 0, 0, di.p, count, dout.p, df.p);
    else if (count && op >= 5) {
        const int two = (op == 5 || op == 7) ? 1 : 0;
        hipLaunchKernelGGL(dg_eig9x2_kernel, dim3(two ? (count + 1) / 2 : count), dim3(64), 0, 0, di.p, count, dout.p, df.p, reps, two);
    }
    else if (count) hipLaunchKernelGGL(dg_mat3_kernel, dim3((count + 63) / 64), dim3(64), 0, 0, op, di.p, count, dout.p, df.p);
    HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, dout.p, count * no * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(flag, df.p, (size_t)count * 4, hipMemcpyDeviceToHost));
    return 0;
}

#ifdef MI_DEGENSAC_DEV
#include "mi_degensac_dev.inc"
#endif
