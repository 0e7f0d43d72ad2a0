/* libmi_degensac.so — host side of the C-ABI (include/mi_degensac.h) and kernel instantiations.
 * gfx950 only.  No CPU fallback: every entry point needs a HIP device. */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>
#include "../../include/mi_degensac.h"
#define DG_T 512
#include "dg_kernel_f_main.h"
#include "dg_kernel_h.h"
#include "dg_variant_impl.h"

static thread_local char g_err[512] = "";
static void set_err(const char *fmt, const char *a = "", const char *b = "") { snprintf(g_err, sizeof g_err, fmt, a, b); }
/* a failed call must not leave HIP's per-thread "last error" set: other users of the runtime in this process
 * (e.g. torch's lazy device initialisation) treat a stale error as their own */
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err("%s failed: %s", #x, hipGetErrorString(e_)); (void)hipGetLastError(); return MI_DEGENSAC_EHIP; } } while (0)

extern "C" const char *mi_degensac_last_error(void) { return g_err; }
/* debug trace (device buffer owned by the caller of mi_degensac_debug_trace); not part of the public header */
static int *g_trace_dev = nullptr; static int g_trace_cap = 0;
static long long *g_phase_dev = nullptr;
extern "C" void mi_degensac_debug_phases(void *dev_ptr) { g_phase_dev = (long long *)dev_ptr; }
extern "C" int mi_degensac_debug_trace(int cap, int *host_out)
{
    if (cap > 0 && !host_out) {           /* arm */
        if (g_trace_dev) hipFree(g_trace_dev);
        if (hipMalloc((void **)&g_trace_dev, (size_t)(1 + 4 * cap) * sizeof(int)) != hipSuccess) return -1;
        hipMemset(g_trace_dev, 0, sizeof(int)); g_trace_cap = cap; return 0;
    }
    if (host_out && g_trace_dev) {        /* fetch and disarm */
        hipDeviceSynchronize();
        hipMemcpy(host_out, g_trace_dev, (size_t)(1 + 4 * g_trace_cap) * sizeof(int), hipMemcpyDeviceToHost);
        hipFree(g_trace_dev); g_trace_dev = nullptr; g_trace_cap = 0; return 0;
    }
    return -1;
}
extern "C" const char *mi_degensac_version(void) { return "mi_degensac 0.1 (gfx950)"; }
extern "C" const char *mi_degensac_kernel_name(int homography) { return homography ? "dg_find_homography_kernel" : "dg_find_fundamental_kernel"; }
extern "C" int mi_degensac_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

/* ---- per-device state: RNG tables uploaded, cached workspace ---------------------------------- */
struct DevState { bool init = false; char *ws = nullptr; size_t ws_bytes = 0; int max_lds = 0, cus = 0;
                  int static_lds[2][2] = {{0, 0}, {0, 0}};   /* [variant: 0 = 512 threads, 1 = 256][F, H] */ };
static const int g_variant_threads[2] = {512, 256};
static DevState g_dev[64];
static int g_last_variant = 0, g_last_mode = 0;
/* which kernel variant (threads per workgroup) and placement mode the last launch of this process used */
extern "C" void mi_degensac_debug_last_launch(int *threads, int *mode) { *threads = g_last_variant; *mode = g_last_mode; }
static std::mutex g_mu;

static void rng_tables(unsigned C[8][32], unsigned G[32])
{
    /* C[k][j] = raw (pre-shift) k-th output after srandom's 310 discards when the initial state is
     * the unit vector e_j: the additive-feedback recurrence is linear over Z/2^32. */
    for (int j = 0; j < 31; j++) {
        unsigned r[31]; for (int i = 0; i < 31; i++) r[i] = (i == j);
        int f = 3, b = 0;
        for (int t = 0; t < 310 + 8; t++) {
            r[f] += r[b];
            if (t >= 310) C[t - 310][j] = r[f];
            if (++f >= 31) f = 0; if (++b >= 31) b = 0;
        }
    }
    for (int k = 0; k < 8; k++) C[k][31] = 0;
    unsigned long long g = 1;
    for (int j = 0; j < 32; j++) { G[j] = (unsigned)g; g = (g * 16807ull) % 2147483647ull; }
}

static int dev_init(int device)
{
    if (device < 0 || device >= 64) { set_err("bad device index"); return MI_DEGENSAC_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_err("no HIP device: this library has no CPU path"); return MI_DEGENSAC_ENODEV; }
    if (device >= ndev) { set_err("device index out of range"); return MI_DEGENSAC_ENODEV; }
    HIPCHK(hipSetDevice(device));
    std::lock_guard<std::mutex> lk(g_mu);
    DevState &d = g_dev[device];
    if (!d.init) {
        hipDeviceProp_t prop; HIPCHK(hipGetDeviceProperties(&prop, device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { set_err("device is %s, this build targets gfx950 only", prop.gcnArchName); return MI_DEGENSAC_ENODEV; }
        unsigned C[8][32], G[32], Ct[32][8]; rng_tables(C, G);
        for (int j = 0; j < 32; j++) for (int k = 0; k < 8; k++) Ct[j][k] = C[k][j];
        d.max_lds = (int)prop.sharedMemPerBlock; d.cus = prop.multiProcessorCount;
        HIPCHK(dg_variant_512_init(C, Ct, G, d.max_lds, d.static_lds[0]));
        HIPCHK(dg_variant_256_init(C, Ct, G, d.max_lds, d.static_lds[1]));
        d.init = true;
    }
    return 0;
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static dg_ws_layout make_layout(int n_max, bool pts_in_ws)
{
    dg_ws_layout w; memset(&w, 0, sizeof w);
    size_t o = 0;
    w.n_max = n_max;
    w.off_lists = o;  o += align_up((size_t)10 * n_max * sizeof(int), 256);
    w.off_flags = o;  o += align_up((size_t)5 * n_max, 256);
    w.off_ht = o;     o += align_up((size_t)(80 + 4 * DG_HT_CAP) * sizeof(int), 256);
    w.off_models = o; o += align_up((size_t)3 * DG_CHUNK * 9 * sizeof(double), 256);
    w.off_stage = o;  o += align_up((size_t)n_max * sizeof(dg_pt), 256);
    w.off_wave = o;   o += align_up((size_t)DG_NW * n_max * (sizeof(int) + sizeof(dg_pt)), 256);
    w.off_res = o;    o += align_up((size_t)3 * DG_CHUNK * 12 + (size_t)DG_CHUNK * 20, 256);
    w.off_pts = o;    if (pts_in_ws) o += align_up((size_t)n_max * sizeof(dg_pt), 256);
    w.off_pool = o;   if (pts_in_ws) o += align_up((size_t)n_max * sizeof(int), 256);
    w.stride = align_up(o, 4096);
    return w;
}

static int ensure_ws(int device, size_t bytes, char **out)
{
    std::lock_guard<std::mutex> lk(g_mu);
    DevState &d = g_dev[device];
    if (d.ws_bytes < bytes) {
        if (d.ws) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(d.ws)); d.ws = nullptr; d.ws_bytes = 0; }
        size_t want = bytes + bytes / 4;
        if (hipMalloc((void **)&d.ws, want) != hipSuccess) { set_err("workspace allocation failed"); return MI_DEGENSAC_ENOMEM; }
        d.ws_bytes = want;
    }
    *out = d.ws;
    return 0;
}

static int fill_params(const mi_degensac_params *p, int homography, int dim, dg_params *o)
{
    if (!p) { set_err("params is NULL"); return MI_DEGENSAC_EINVAL; }
    memset(o, 0, sizeof *o);
    const double coef = 3.0 * (p->symmetric_error_check ? 1 : 0);
    if (!homography) {
        /* bindings.cpp:297-318 */
        if (p->error_type != 0 && p->error_type != 1) { set_err("error_type must be 0 or 1 for the fundamental matrix"); return MI_DEGENSAC_EINVAL; }
        o->th = p->px_th * p->px_th; o->sym_th = p->px_th * p->px_th * coef;
    } else {
        /* bindings.cpp:64-107 */
        switch (p->error_type) {
        case 0: o->th = p->px_th * p->px_th; o->sym_th = p->px_th * coef; break;
        case 1: o->th = p->px_th * p->px_th; o->sym_th = 0; break;
        case 2: o->th = p->px_th;            o->sym_th = 0; break;
        case 3: o->th = p->px_th * p->px_th; o->sym_th = p->px_th * coef; break;
        case 4: o->th = p->px_th;            o->sym_th = p->px_th * coef; break;
        default: set_err("error_type must be 0..4 for the homography"); return MI_DEGENSAC_EINVAL;
        }
    }
    o->laf_coef = (p->laf_consistensy_coef > 0 && dim == 6) ? p->laf_consistensy_coef : 0.0;
    o->conf = p->conf; o->max_iters = p->max_iters; o->error_type = p->error_type;
    o->degen = p->enable_degeneracy_check ? 1 : 0;
    o->final_laf_filter = (p->flags & MI_DEGENSAC_FLAG_FINAL_LAF_FILTER) ? 1 : 0;
    return 0;
}

/* ---- batched device entry point ---------------------------------------------------------------- */
static int launch_batch(int homography, const double *d_p1, const double *d_p2, const int64_t *d_off, const int64_t *h_off,
                        int n_pairs, int dim, const mi_degensac_params *prm, const uint32_t *d_seeds, int device,
                        hipStream_t stream, double *d_model, uint8_t *d_mask, int32_t *d_stats)
{
    if (n_pairs <= 0) return 0;
    if (dim != 2 && dim != 6) { set_err("points must be [n,2] or [n,6]"); return MI_DEGENSAC_EINVAL; }
    int rc = dev_init(device); if (rc) return rc;
    dg_args A; memset(&A, 0, sizeof A);
    rc = fill_params(prm, homography, dim, &A.prm); if (rc) return rc;
    int n_max = 0, n_min = 1 << 30;
    for (int p = 0; p < n_pairs; p++) { long long n = h_off[p+1] - h_off[p]; if (n > n_max) n_max = (int)n; if (n < n_min) n_min = (int)n; }
    const int min_pts = homography ? 4 : 8;                      /* bindings.cpp:35,270 */
    if (n_min < min_pts) { set_err(homography ? "need n >= 4 correspondences" : "need n >= 8 correspondences"); return MI_DEGENSAC_EINVAL; }
    /* Variant and placement.  Latency: 512-thread workgroups, one pair per CU, point set + sampler pool in LDS when
     * they fit (36 B per correspondence next to the static LDS).  Throughput: once the batch holds several pairs per CU,
     * 256-thread workgroups with only the pool in LDS leave room for two resident pairs per CU, which hides the serial
     * small-solver chains of one pair behind the other.  MI_DEGENSAC_VARIANT / MI_DEGENSAC_MODE override. */
    const DevState &ds = g_dev[device];
    const size_t dyn_all = (size_t)n_max * (sizeof(dg_pt) + sizeof(int)), dyn_pool = (size_t)n_max * sizeof(int);
    int variant = 0, mode;
    if (!homography && n_pairs >= 3 * ds.cus && ds.static_lds[1][0] + dyn_pool + 512 <= (size_t)ds.max_lds / 2) variant = 1;
    if (const char *e = getenv("MI_DEGENSAC_VARIANT")) variant = atoi(e) == 256 ? 1 : 0;
    const size_t room = (size_t)(ds.max_lds - ds.static_lds[variant][homography] - 256);
    if (variant == 1)           mode = dyn_pool <= room ? DG_MODE_POOL_LDS : DG_MODE_HBM;
    else if (dyn_all <= room)   mode = DG_MODE_LDS;
    else                        mode = dyn_pool <= room ? DG_MODE_POOL_LDS : DG_MODE_HBM;
    if (const char *e = getenv("MI_DEGENSAC_MODE")) {
        const int m = atoi(e);
        if (m == DG_MODE_HBM || (m == DG_MODE_POOL_LDS && dyn_pool <= room) || (m == DG_MODE_LDS && dyn_all <= room)) mode = m;
    }
    if (getenv("MI_DEGENSAC_FORCE_GLOBAL")) mode = DG_MODE_HBM;
    const size_t dyn = mode == DG_MODE_LDS ? dyn_all : (mode == DG_MODE_POOL_LDS ? dyn_pool : 0);
    A.wl = make_layout(n_max, mode != DG_MODE_LDS);
    char *ws; rc = ensure_ws(device, A.wl.stride * (size_t)n_pairs, &ws); if (rc) return rc;
    A.ws = ws; A.pts1 = d_p1; A.pts2 = d_p2; A.offsets = (const long long *)d_off; A.seeds = d_seeds;
    A.trace = g_trace_dev; A.trace_cap = g_trace_cap; A.phase_out = g_phase_dev;
    A.model_out = d_model; A.mask_out = d_mask; A.stats_out = d_stats; A.dim = dim; A.n_pairs = n_pairs; A.pts_in_lds = mode == DG_MODE_LDS;
    if (variant == 1) HIPCHK(dg_variant_256_launch(homography, mode, n_pairs, dyn, stream, A));
    else              HIPCHK(dg_variant_512_launch(homography, mode, n_pairs, dyn, stream, A));
    g_last_variant = g_variant_threads[variant]; g_last_mode = mode;
    return 0;
}

extern "C" int mi_degensac_find_fundamental_batch_dev(const double *d_pts1, const double *d_pts2, const int64_t *d_offsets,
        const int64_t *offsets_host, int n_pairs, int dim, const mi_degensac_params *prm, const uint32_t *d_seeds, int device,
        void *stream, double *d_F, uint8_t *d_mask, int32_t *d_stats)
{
    return launch_batch(0, d_pts1, d_pts2, d_offsets, offsets_host, n_pairs, dim, prm, d_seeds, device, (hipStream_t)stream, d_F, d_mask, d_stats);
}
extern "C" int mi_degensac_find_homography_batch_dev(const double *d_pts1, const double *d_pts2, const int64_t *d_offsets,
        const int64_t *offsets_host, int n_pairs, int dim, const mi_degensac_params *prm, const uint32_t *d_seeds, int device,
        void *stream, double *d_H, uint8_t *d_mask, int32_t *d_stats)
{
    return launch_batch(1, d_pts1, d_pts2, d_offsets, offsets_host, n_pairs, dim, prm, d_seeds, device, (hipStream_t)stream, d_H, d_mask, d_stats);
}

/* ---- host-pointer entry points: stage through HBM ----------------------------------------------- */
template <class T> struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    int alloc(size_t n) { return hipMalloc((void **)&p, (n ? n : 1) * sizeof(T)) == hipSuccess ? 0 : MI_DEGENSAC_ENOMEM; }
};

static int host_batch(int homography, const double *p1, const double *p2, const int64_t *off, int n_pairs, int dim,
                      const mi_degensac_params *prm, const uint32_t *seeds, int device, double *model, uint8_t *mask, int32_t *stats)
{
    if (!p1 || !p2 || !off || !model || !mask || !seeds) { set_err("NULL argument"); return MI_DEGENSAC_EINVAL; }
    if (n_pairs <= 0) return 0;
    int rc = dev_init(device); if (rc) return rc;
    size_t total = (size_t)off[n_pairs];
    DevBuf<double> d1, d2, dm; DevBuf<int64_t> doff; DevBuf<uint32_t> ds; DevBuf<uint8_t> dmask; DevBuf<int32_t> dst;
    if (d1.alloc(total * dim) || d2.alloc(total * dim) || dm.alloc((size_t)n_pairs * 9) || doff.alloc(n_pairs + 1) || ds.alloc(n_pairs) ||
        dmask.alloc(total) || dst.alloc((size_t)n_pairs * 16)) { set_err("device allocation failed"); return MI_DEGENSAC_ENOMEM; }
    HIPCHK(hipMemcpy(d1.p, p1, total * dim * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d2.p, p2, total * dim * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(doff.p, off, (n_pairs + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ds.p, seeds, n_pairs * sizeof(uint32_t), hipMemcpyHostToDevice));
    rc = launch_batch(homography, d1.p, d2.p, doff.p, off, n_pairs, dim, prm, ds.p, device, 0, dm.p, dmask.p, dst.p);
    if (rc) return rc;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(model, dm.p, (size_t)n_pairs * 9 * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(mask, dmask.p, total, hipMemcpyDeviceToHost));
    if (stats) HIPCHK(hipMemcpy(stats, dst.p, (size_t)n_pairs * 16 * sizeof(int32_t), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int mi_degensac_find_fundamental_batch(const double *pts1, const double *pts2, const int64_t *offsets, int n_pairs, int dim,
        const mi_degensac_params *prm, const uint32_t *seeds, int device, double *F, uint8_t *mask, int32_t *stats)
{ return host_batch(0, pts1, pts2, offsets, n_pairs, dim, prm, seeds, device, F, mask, stats); }
extern "C" int mi_degensac_find_homography_batch(const double *pts1, const double *pts2, const int64_t *offsets, int n_pairs, int dim,
        const mi_degensac_params *prm, const uint32_t *seeds, int device, double *H, uint8_t *mask, int32_t *stats)
{ return host_batch(1, pts1, pts2, offsets, n_pairs, dim, prm, seeds, device, H, mask, stats); }

extern "C" int mi_degensac_find_fundamental(const double *pts1, const double *pts2, int n, int dim, const mi_degensac_params *prm,
        uint32_t seed, int device, double *F, uint8_t *mask, int32_t *stats)
{ int64_t off[2] = {0, n}; return host_batch(0, pts1, pts2, off, 1, dim, prm, &seed, device, F, mask, stats); }
extern "C" int mi_degensac_find_homography(const double *pts1, const double *pts2, int n, int dim, const mi_degensac_params *prm,
        uint32_t seed, int device, double *H, uint8_t *mask, int32_t *stats)
{ int64_t off[2] = {0, n}; return host_batch(1, pts1, pts2, off, 1, dim, prm, &seed, device, H, mask, stats); }

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
    int rc = dev_init(device); if (rc) return rc;
    if (!(kind == 0 || kind == 1 || kind == 2 || (kind >= 10 && kind <= 14))) { set_err("unsupported metric kind"); return MI_DEGENSAC_EINVAL; }
    DevBuf<double> d1, d2, dm, dJ, dr; DevBuf<uint32_t> dI;
    if (d1.alloc((size_t)n * dim) || d2.alloc((size_t)n * dim) || dm.alloc((size_t)n_models * 9) || dJ.alloc(n_models) || dI.alloc(n_models) ||
        (resid && dr.alloc((size_t)n_models * n))) { set_err("device allocation failed"); return MI_DEGENSAC_ENOMEM; }
    HIPCHK(hipMemcpy(d1.p, pts1, (size_t)n * dim * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d2.p, pts2, (size_t)n * dim * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dm.p, models, (size_t)n_models * 72, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(dg_score_models_kernel, dim3((n_models + 3) / 4), dim3(256), 0, 0, d1.p, d2.p, n, dim, dm.p, n_models, kind, th, dI.p, dJ.p, resid ? dr.p : nullptr);
    HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(I, dI.p, (size_t)n_models * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(J, dJ.p, (size_t)n_models * 8, hipMemcpyDeviceToHost));
    if (resid) HIPCHK(hipMemcpy(resid, dr.p, (size_t)n_models * n * 8, hipMemcpyDeviceToHost));
    return 0;
}

__global__ void dg_sample_stream_kernel(unsigned seed0, int n, int ssz, int iters, int *pool_g, int *out)
{
    /* the main kernels' sampler (dg_sample_chunk), chunk by chunk, run by one wave: with the pool in LDS (n <= 4096:
     * the parallel pool stage) or in global memory (the sequential one) */
    __shared__ unsigned seeds[DG_CHUNK]; __shared__ int draws[DG_CHUNK][8]; __shared__ dg_rng g; __shared__ unsigned sd0; __shared__ unsigned long long alm[DG_CHUNK / 64];
    __shared__ int pool_l[4096]; __shared__ int scratch[2 * DG_CHUNK * 7];
    const int lane = threadIdx.x;
    const bool lds = n <= 4096;
    int *pool = lds ? pool_l : pool_g;
    for (int i = lane; i < n; i += 64) pool[i] = i;
    if (lane == 0) { dg_srand(&g, seed0); sd0 = (unsigned)dg_rand(&g); }
    __syncthreads();
    unsigned seed = sd0;
    for (int base = 0; base < iters; base += DG_CHUNK) {
        int chunk = iters - base; if (chunk > DG_CHUNK) chunk = DG_CHUNK;
        if (lds) seed = ssz == 7 ? dg_sample_chunk<7, 2>(seed, chunk, n, pool, seeds, draws, alm, scratch, lane) : dg_sample_chunk<4, 2>(seed, chunk, n, pool, seeds, draws, alm, scratch, lane);
        else     seed = ssz == 7 ? dg_sample_chunk<7, 0>(seed, chunk, n, pool, seeds, draws, alm, 0, lane) : dg_sample_chunk<4, 0>(seed, chunk, n, pool, seeds, draws, alm, 0, lane);
        __syncthreads();
        for (int k = lane; k < chunk; k += 64) for (int i = 0; i < ssz; i++) out[(size_t)(base + k) * ssz + i] = draws[k][i];
        __syncthreads();
    }
}

extern "C" int mi_degensac_sample_stream(uint32_t seed, int n, int sample_size, int iters, int device, int32_t *samples)
{
    int rc = dev_init(device); if (rc) return rc;
    if ((sample_size != 4 && sample_size != 7) || n < sample_size + 1) { set_err("bad sample size"); return MI_DEGENSAC_EINVAL; }
    DevBuf<int> dpool, dout;
    if (dpool.alloc(n) || dout.alloc((size_t)iters * sample_size)) { set_err("device allocation failed"); return MI_DEGENSAC_ENOMEM; }
    hipLaunchKernelGGL(dg_sample_stream_kernel, dim3(1), dim3(64), 0, 0, seed, n, sample_size, iters, dpool.p, dout.p);
    HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(samples, dout.p, (size_t)iters * sample_size * 4, hipMemcpyDeviceToHost));
    return 0;
}

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
    if (!ok) {
        double Ag[81], sol[81]; int nb[18];
        for (int i = 0; i < 7; i++) { double a[3] = {sp[i].x1, sp[i].y1, 1.0}, b[3] = {sp[i].x2, sp[i].y2, 1.0}; for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++) Ag[9*i+3*k+l] = b[k] * a[l]; }
        for (int i = 63; i < 81; i++) Ag[i] = 0;
        for (int i = 0; i < 81; i++) sol[i] = 0;
        if (dg_nullspace(Ag, sol, 9, nb) == 2) { for (int i = 0; i < 9; i++) { f1[i] = sol[i]; f2[i] = sol[9+i]; } ok = 1; }
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
    int rc = dev_init(device); if (rc) return rc;
    DevBuf<double> d1, d2, dm; DevBuf<int> ds, dn, dr;
    if (d1.alloc((size_t)n * dim) || d2.alloc((size_t)n * dim) || dm.alloc((size_t)n_samples * 27) || ds.alloc((size_t)n_samples * 7) || dn.alloc(n_samples) || dr.alloc((size_t)n_samples * 3))
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


/* ---- micro-benchmark of the lane-0 small dense routines (development aid) ------------------------- */
__global__ void dg_microbench_kernel(const double *in, double *out, long long *ticks, int reps)
{
    __shared__ dg_lsq_scratch ls; __shared__ double F[9], H[9], u7[7][4]; __shared__ int list[800];
    const int tid = threadIdx.x;
    if (tid == 0) {
        for (int i = 0; i < 64; i++) ls.px[i] = in[i];
        for (int i = 0; i < 7; i++) for (int j = 0; j < 4; j++) u7[i][j] = in[4*i + j];
        for (int i = 0; i < 9; i++) F[i] = in[200 + i];
        dg_singulF(F);
        for (int i = 0; i < 800; i++) list[i] = (i * 37) % 2000;
    }
    DG_WSYNC();
    long long t0, t1;
    t0 = wall_clock64();
    for (int r = 0; r < reps; r++) { if (tid == 0) { for (int i = 0; i < 14; i++) for (int k = 0; k < 9; k++) ls.Z[9*i+k] = in[64 + 9*i + k] + 1e-9 * r; } DG_WSYNC(); dg_cov9_wave(ls.V, ls.Z, 14, tid); DG_WSYNC(); dg_eig_sym_wave(ls.V, ls.D, tid, &ls.ews); }
    t1 = wall_clock64(); if (tid == 0) ticks[0] = t1 - t0;
#ifdef DG_EIG_TIMING
    long long et_[4]; for (int i = 0; i < 4; i++) et_[i] = dg_eig_ticks[i];
#endif
    t0 = wall_clock64();
    for (int r = 0; r < reps; r++) { if (tid == 0) { for (int i = 0; i < 14; i++) for (int k = 0; k < 9; k++) ls.Z[9*i+k] = in[64 + 9*i + k] + 1e-9 * r; dg_cov9(ls.V, ls.Z, 14); dg_eig_sym(ls.V, ls.D, 9); } DG_WSYNC(); }
    t1 = wall_clock64(); if (tid == 0) ticks[1] = t1 - t0;
    t0 = wall_clock64();
    for (int r = 0; r < reps; r++) dg_u2f_small_w(&ls, ls.px, 0, 14, F, tid);
    t1 = wall_clock64(); if (tid == 0) ticks[2] = t1 - t0;
    t0 = wall_clock64();
    for (int r = 0; r < reps; r++) dg_u2f_small_w(&ls, ls.px, 0, 8, F, tid);
    t1 = wall_clock64(); if (tid == 0) ticks[3] = t1 - t0;
    t0 = wall_clock64();
    for (int r = 0; r < reps; r++) { if (tid == 0) { for (int i = 0; i < 9; i++) F[i] = in[200 + i] + 1e-9 * r; dg_singulF(F); } DG_WSYNC(); }
    t1 = wall_clock64(); if (tid == 0) ticks[4] = t1 - t0;
    t0 = wall_clock64();
    /* slot 5: number of (f, g) pairs (out of 64 * 20000, exponents spread over +-2^60) on which dg_lartg_fast differs
     * from dg_lartg in any bit */
    {
        unsigned long long st = 0x9E3779B97F4A7C15ull * (unsigned long long)(tid + 1); unsigned bad = 0;
        for (int it = 0; it < 20000; it++) {
            st = st * 6364136223846793005ull + 1442695040888963407ull; const unsigned long long a = st;
            st = st * 6364136223846793005ull + 1442695040888963407ull; const unsigned long long b = st;
            double f = (double)(long long)(a >> 11) * (1.0 / 9007199254740992.0) - 0.5, g = (double)(long long)(b >> 11) * (1.0 / 9007199254740992.0) - 0.5;
            f = ldexp(f, (int)(a & 127) - 64); g = ldexp(g, (int)(b & 127) - 64);
            double c1, s1, r1, c2, s2, r2;
            dg_lartg(f, g, &c1, &s1, &r1); dg_lartg_fast(f, g, &c2, &s2, &r2);
            bad += (__double_as_longlong(c1) != __double_as_longlong(c2) || __double_as_longlong(s1) != __double_as_longlong(s2) || __double_as_longlong(r1) != __double_as_longlong(r2)) ? 1u : 0u;
        }
        bad = dg_wave_sum_u(bad);
        if (tid == 0) ticks[5] = (long long)bad;
    }
    t0 = wall_clock64();
    for (int r = 0; r < reps; r++) dg_u2h_small_w(&ls, ls.px, 5, H, tid);
    t1 = wall_clock64(); if (tid == 0) ticks[6] = t1 - t0;
    t0 = wall_clock64();
    if (tid < 64) { unsigned hsum = 0; for (int r = 0; r < reps; r++) hsum += dg_hash_list(list, 800 - (r & 1), true); if (tid == 0) out[10] = hsum; }
    t1 = wall_clock64(); if (tid == 0) ticks[7] = t1 - t0;
    if (tid == 0) for (int i = 0; i < 9; i++) out[i] = F[i];
#ifdef DG_EIG_TIMING
    if (tid == 0) for (int i = 0; i < 4; i++) ticks[4 + i] = et_[i];
#endif
}

/* dev probe: dependent-issue latency of the fp64 building blocks, one wave.  out[k] = wall_clock64 ticks (10 ns) per 1000 ops */
__global__ void dg_latency_kernel(double *io, long long *out)
{
    const int lane = threadIdx.x;
    double x = io[0] + lane * 1e-9, y = io[1], z;
    long long t0, t1; const int N = 4000;
    t0 = wall_clock64(); for (int i = 0; i < N; i++) x = __builtin_fma(x, y, 1e-3); t1 = wall_clock64(); out[0] = (t1 - t0);
    t0 = wall_clock64(); for (int i = 0; i < N; i++) x = x * y; t1 = wall_clock64(); out[1] = (t1 - t0);
    t0 = wall_clock64(); for (int i = 0; i < N; i++) x = x + y; t1 = wall_clock64(); out[2] = (t1 - t0);
    x = io[0] + 3.0;
    t0 = wall_clock64(); for (int i = 0; i < N; i++) x = 1.7 / x + 1.0; t1 = wall_clock64(); out[3] = (t1 - t0);
    t0 = wall_clock64(); for (int i = 0; i < N; i++) x = sqrt(x) + 2.0; t1 = wall_clock64(); out[4] = (t1 - t0);
    z = x;
    t0 = wall_clock64(); for (int i = 0; i < N; i++) { double c, s, r; dg_lartg_fast(z, y, &c, &s, &r); z = r * 0.7 + s; } t1 = wall_clock64(); out[5] = (t1 - t0);
    t0 = wall_clock64(); for (int i = 0; i < N; i++) { double c, s, r; dg_lartg(z, y, &c, &s, &r); z = r * 0.7 + s; } t1 = wall_clock64(); out[6] = (t1 - t0);
    int k = lane & 7; double w = x;
    t0 = wall_clock64(); for (int i = 0; i < N; i++) { w = dg_rdl_d(w, (i + (int)io[2]) & 7) + 1.0; } t1 = wall_clock64(); out[7] = (t1 - t0);
    unsigned u = (unsigned)lane;
    t0 = wall_clock64(); for (int i = 0; i < N; i++) { u = u * 1664525u + 1013904223u; } t1 = wall_clock64(); out[8] = (t1 - t0);
    long long c0 = clock64(); for (int i = 0; i < N; i++) x = __builtin_fma(x, y, 1e-3); long long c1 = clock64(); out[9] = (c1 - c0);
    io[8 + lane] = x + z + w + (double)u + k;
}
extern "C" int mi_degensac_latency_probe(long long *out_host)
{
    int rc = dev_init(0); if (rc) return rc;
    DevBuf<double> io; DevBuf<long long> o;
    if (io.alloc(128) || o.alloc(16)) return MI_DEGENSAC_ENOMEM;
    double h[8] = {1.0000001, 0.9999999, 0.0, 0, 0, 0, 0, 0};
    HIPCHK(hipMemcpy(io.p, h, sizeof h, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(dg_latency_kernel, dim3(1), dim3(64), 0, 0, io.p, o.p);
    HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out_host, o.p, 16 * sizeof(long long), hipMemcpyDeviceToHost));
    return 0;
}


/* dev probe: when several lanes of ONE ds_wrxchg_rtn instruction hit the same LDS address, are they serialised in
 * ascending lane order?  out[0] = violations over all repetitions and patterns, out[1] = exchanges checked */
__global__ void dg_atomic_order_kernel(long long *out)
{
    __shared__ int tab[64];
    const int lane = threadIdx.x; long long bad = 0, tot = 0;
    for (int rep = 0; rep < 2000; rep++) {
        for (int K = 1; K <= 64; K = (K < 8 ? K + 1 : K * 2)) {
            tab[lane] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int addr = (lane * 7 + rep) % K;                         /* many lanes per address, scattered */
            const int old = atomicExch(&tab[addr], lane + 1);
            /* expected predecessor: the largest lane l' < lane with the same address, else 0 */
            int exp = 0;
            for (int l2 = 0; l2 < lane; l2++) if ((l2 * 7 + rep) % K == addr) exp = l2 + 1;
            bad += (old != exp); tot++;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    for (int o = 32; o >= 1; o >>= 1) { bad += __shfl_xor(bad, o, 64); tot += __shfl_xor(tot, o, 64); }
    if (lane == 0) { out[0] = bad; out[1] = tot; }
}
extern "C" int mi_degensac_atomic_order_probe(long long *out_host)
{
    int rc = dev_init(0); if (rc) return rc;
    DevBuf<long long> o; if (o.alloc(2)) return MI_DEGENSAC_ENOMEM;
    hipLaunchKernelGGL(dg_atomic_order_kernel, dim3(1), dim3(64), 0, 0, o.p);
    HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out_host, o.p, 16, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int mi_degensac_microbench(const double *in_host, int reps, long long *ticks_host)
{
    int rc = dev_init(0); if (rc) return rc;
    DevBuf<double> din, dout; DevBuf<long long> dt;
    if (din.alloc(512) || dout.alloc(16) || dt.alloc(8)) return MI_DEGENSAC_ENOMEM;
    HIPCHK(hipMemcpy(din.p, in_host, 512 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(dg_microbench_kernel, dim3(1), dim3(64), 0, 0, din.p, dout.p, dt.p, reps);
    HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(ticks_host, dt.p, 64, hipMemcpyDeviceToHost));
    return 0;
}
