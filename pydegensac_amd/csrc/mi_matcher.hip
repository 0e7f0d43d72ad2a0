/* libmi_degensac.so — tentative-correspondence stage in front of the estimators: brute-force 2-nearest-
 * neighbour search over descriptor matrices, the second-nearest-neighbour ratio test and the optional mutual check of
 * the reference's example pipeline (examples/simple-example.py:46-53: cv2.BFMatcher().knnMatch(descs1, descs2, k=2),
 * then `m.distance < 0.9 * n.distance`).  gfx950 only, no CPU path.
 *
 * Distances are formed the way the tests' oracle (oracle/matcher_np.py) forms them, so that ranks, ties and ratio
 * decisions are bit-reproducible: L2 = sqrt of the fp32 sum of squared differences accumulated over the descriptor
 * dimension in ascending order (no FMA contraction: compiled with -ffp-contract=off), Hamming = popcount over the
 * bytes; ties go to the lower train index.  That rules out the |a|^2 + |b|^2 - 2ab matrix-core form (different
 * roundings, cancellation near duplicates); the direct form is 3 n1 n2 dim flop, a few tens of microseconds for two
 * images' worth of descriptors, and is LDS-tiled instead: a workgroup owns 64 queries, streams the train set through
 * LDS 64 rows at a time (both tiles stored dimension-major, so a wave reads consecutive words / one broadcast word),
 * every thread keeps a 4 x 4 block of running sums in registers and its own running top-2 per query; the 16 threads
 * sharing a query merge their candidates at the end.  The train set is additionally split over blockIdx.y (see mt_knn2_kernel). */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/mi_degensac.h"

#define MT_Q   64            /* queries per workgroup */
#define MT_T   64            /* train rows per LDS tile */
#define MT_DC  64            /* descriptor words per LDS chunk */

struct mt_best { float d0, d1; int i0, i1; };     /* (d0, i0) <= (d1, i1) lexicographically */

/* candidate (d, i) into a running top-2; equal distances keep the lower index first */
__device__ __forceinline__ void mt_push(mt_best &b, float d, int i)
{
    const bool lt0 = d < b.d0 || (d == b.d0 && i < b.i0);
    const bool lt1 = d < b.d1 || (d == b.d1 && i < b.i1);
    if (lt0) { b.d1 = b.d0; b.i1 = b.i0; b.d0 = d; b.i0 = i; }
    else if (lt1) { b.d1 = d; b.i1 = i; }
}

/* NORM: 0 = L2 over float words, 1 = Hamming over 32-bit words of packed bytes.  q, t: [n, words] row-major. */
/* Launch shape: grid = (ceil(n1 / 64), train splits).  Two images' worth of descriptors give only a few dozen query
 * tiles (1 500 queries = 24), so the train set is split over blockIdx.y until the grid covers the CUs about twice; a split
 * writes its top-2 per query (squared distances) to part[split][query] and mt_merge_kernel merges the splits with the same
 * (distance, index) order, so the result does not depend on the split count.  With one split the kernel writes the final
 * answer itself. */
template <int NORM>
__global__ __launch_bounds__(256) void mt_knn2_kernel(const uint32_t *q, int n1, const uint32_t *t, int n2, int words,
    int t_chunk /* train rows per split, multiple of 64 */,
                                                      int32_t *idx /* [n1,2] */, float *dist /* [n1,2] */, mt_best *part /* [splits][n1] or null */)
{
    __shared__ uint32_t qs[MT_DC][MT_Q + 1], ts[MT_DC][MT_T + 1];
    __shared__ mt_best merge[MT_Q][16];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int q0 = blockIdx.x * MT_Q;
    mt_best best[4];
#pragma unroll
    for (int a = 0; a < 4; a++) { best[a].d0 = best[a].d1 = __builtin_inff(); best[a].i0 = best[a].i1 = -1; }
    const int t_lo = (int)blockIdx.y * t_chunk, t_hi = t_lo + t_chunk < n2 ? t_lo + t_chunk : n2;
    for (int t0 = t_lo; t0 < t_hi; t0 += MT_T) {
        float acc[4][4]; unsigned hacc[4][4];
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) { acc[a][b] = 0.f; hacc[a][b] = 0u; }
        for (int w0 = 0; w0 < words; w0 += MT_DC) {
            __syncthreads();
            /* stage both tiles dimension-major: thread r loads row (r / 4), a quarter of the chunk's words, coalesced per row */
            for (int e = tid; e < MT_Q * MT_DC; e += 256) {
                const int r = e / MT_DC, w = e - r * MT_DC;
                const bool okq = q0 + r < n1 && w0 + w < words, okt = t0 + r < t_hi && w0 + w < words;
                qs[w][r] = okq ? q[(size_t)(q0 + r) * words + w0 + w] : 0u;
                ts[w][r] = okt ? t[(size_t)(t0 + r) * words + w0 + w] : 0u;
            }
            __syncthreads();
            const int wn = words - w0 < MT_DC ? words - w0 : MT_DC;
            for (int w = 0; w < wn; w++) {
                uint32_t qa[4], tb[4];
#pragma unroll
                for (int a = 0; a < 4; a++) qa[a] = qs[w][4 * ty + a];
#pragma unroll
                for (int b = 0; b < 4; b++) tb[b] = ts[w][tx + 16 * b];
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        if (NORM == 0) { const float df = __uint_as_float(qa[a]) - __uint_as_float(tb[b]); acc[a][b] = acc[a][b] + df * df; }
                        else hacc[a][b] += (unsigned)__popc(qa[a] ^ tb[b]);
                    }
            }
        }
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int ti = t0 + tx + 16 * b;
                if (ti < t_hi) mt_push(best[a], NORM == 0 ? acc[a][b] : (float)hacc[a][b], ti);
            }
    }
    /* the 16 threads of a query row merge their candidates (lane order does not matter: mt_push orders by (d, i)) */
#pragma unroll
    for (int a = 0; a < 4; a++) merge[4 * ty + a][tx] = best[a];
    __syncthreads();
    if (tid < MT_Q && q0 + tid < n1) {
        mt_best m = merge[tid][0];
        for (int k = 1; k < 16; k++) { const mt_best c = merge[tid][k]; if (c.i0 >= 0) mt_push(m, c.d0, c.i0); if (c.i1 >= 0) mt_push(m, c.d1, c.i1); }
        if (part) { part[(size_t)blockIdx.y * n1 + q0 + tid] = m; return; }
        const size_t o = (size_t)(q0 + tid) * 2;
        idx[o] = m.i0; idx[o + 1] = m.i1;
        dist[o] = NORM == 0 ? sqrtf(m.d0) : m.d0; dist[o + 1] = NORM == 0 ? sqrtf(m.d1) : m.d1;
    }
}

/* merge the per-split top-2 of every query (any order gives the same result: mt_push orders by (distance, index)) */
__global__ void mt_merge_kernel(const mt_best *part, int splits, int n1, int l2, int32_t *idx, float *dist)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    mt_best m = part[i];
    for (int s = 1; s < splits; s++) { const mt_best c = part[(size_t)s * n1 + i]; if (c.i0 >= 0) mt_push(m, c.d0, c.i0); if (c.i1 >= 0) mt_push(m, c.d1, c.i1);
        }
    idx[2 * i] = m.i0; idx[2 * i + 1] = m.i1;
    dist[2 * i] = l2 ? sqrtf(m.d0) : m.d0; dist[2 * i + 1] = l2 ? sqrtf(m.d1) : m.d1;
}

/* keep[i] = second-nearest-neighbour ratio test (strict, as the example's `m.distance < ratio * n.distance`; a query with
 * fewer than two train rows never passes) and, when back != 0, the mutual check back[idx[i][0]][0] == i */
__global__ void mt_filter_kernel(const int32_t *idx, const float *dist, int n1, float ratio, const int32_t *back, uint8_t *keep)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    const int j = idx[2 * i], j2 = idx[2 * i + 1];
    bool ok = j >= 0 && j2 >= 0 && dist[2 * i] < ratio * dist[2 * i + 1];
    if (ok && back) ok = back[2 * j] == i;
    keep[i] = ok ? 1 : 0;
}

/* utils.py:24-41 convert_cv2_kpts_to_xyA on the device: (x, y, size, angle in degrees) -> (x, y, s cos a, s sin a,
 * -s sin a, s cos a) in float64, the [n, 6] rows the estimators take for the LAF consistency checks */
__global__ void mt_kpts_to_xyA_kernel(const float *kp, int n, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = kp[4 * i], y = kp[4 * i + 1], s = kp[4 * i + 2], a = kp[4 * i + 3];
    const double r = a * 3.141592653589793 / 180.0, cs = cos(r), sn = sin(r);
    double *o = out + (size_t)i * 6;
    o[0] = x; o[1] = y; o[2] = s * cs; o[3] = s * sn; o[4] = -s * sn; o[5] = s * cs;
}

static thread_local char mt_err[256] = "";
extern "C" const char *mi_degensac_match_last_error(void) { return mt_err; }
#define MTCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { snprintf(mt_err, sizeof mt_err, "%s failed: %s", #x, hipGetErrorString(e_)); \
    (void)hipGetLastError(); return MI_DEGENSAC_EHIP; } } while (0)

struct MtDevGuard {
    int prev = -1; bool armed = false;
    int enter(int device)
    {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { (void)hipGetLastError();
            snprintf(mt_err, sizeof mt_err, "no HIP device: this library has no CPU path"); return MI_DEGENSAC_ENODEV; }
        if (device < 0 || device >= n) { snprintf(mt_err, sizeof mt_err, "device index out of range"); return MI_DEGENSAC_ENODEV; }
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        if (prev != device) { MTCHK(hipSetDevice(device)); armed = prev >= 0; }
        return 0;
    }
    ~MtDevGuard() { if (armed) (void)hipSetDevice(prev); }
};

static int mt_words(int norm, int dim)
{
    if (norm == MI_DEGENSAC_NORM_L2) return dim;
    return dim % 4 == 0 ? dim / 4 : -1;           /* Hamming rows are passed padded to whole 32-bit words */
}

extern "C" int mi_degensac_match_knn2_dev(int norm, const void *d_desc1, int n1, const void *d_desc2, int n2, int dim, int device,
                                          void *stream, int32_t *d_idx, float *d_dist)
{
    if ((norm != MI_DEGENSAC_NORM_L2 && norm != MI_DEGENSAC_NORM_HAMMING) || n1 < 0 || n2 < 0 || dim <= 0) { snprintf(mt_err, sizeof mt_err, "bad argument");
        return MI_DEGENSAC_EINVAL; }
    const int words = mt_words(norm, dim);
    if (words < 0) { snprintf(mt_err, sizeof mt_err, "Hamming descriptors must be padded to a multiple of 4 bytes"); return MI_DEGENSAC_EINVAL; }
    MtDevGuard g; int rc = g.enter(device); if (rc) return rc;
    if (n1 == 0) return 0;
    /* train splits: enough workgroups to cover the device about twice, never less than one 64-row tile per split */
    const int qtiles = (n1 + MT_Q - 1) / MT_Q, ttiles = n2 > 0 ? (n2 + MT_T - 1) / MT_T : 1;
    int cus = 256; { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess) cus = pr.multiProcessorCount; else (void)hipGetLastError(); }
    int splits = (2 * cus + qtiles - 1) / qtiles; if (splits > ttiles) splits = ttiles; if (splits < 1) splits = 1; if (splits > 256) splits = 256;
    const int t_chunk = ((ttiles + splits - 1) / splits) * MT_T;
    splits = n2 > 0 ? (n2 + t_chunk - 1) / t_chunk : 1;
    mt_best *part = nullptr;
    if (splits > 1) MTCHK(hipMallocAsync((void **)&part, (size_t)splits * n1 * sizeof(mt_best), (hipStream_t)stream));
    const dim3 grid(qtiles, splits), block(256);
    if (norm == MI_DEGENSAC_NORM_L2) hipLaunchKernelGGL(mt_knn2_kernel<0>, grid, block, 0, (hipStream_t)stream, (const uint32_t *)d_desc1, n1,
        (const uint32_t *)d_desc2, n2, words, n2 > 0 ? t_chunk : MT_T, d_idx, d_dist, part);
    else                             hipLaunchKernelGGL(mt_knn2_kernel<1>, grid, block, 0, (hipStream_t)stream, (const uint32_t *)d_desc1, n1,
        (const uint32_t *)d_desc2, n2, words, n2 > 0 ? t_chunk : MT_T, d_idx, d_dist, part);
    hipError_t le = hipGetLastError();
    if (le == hipSuccess && part) {
        hipLaunchKernelGGL(mt_merge_kernel, dim3((n1 + 255) / 256), dim3(256), 0, (hipStream_t)stream, part, splits, n1, norm == MI_DEGENSAC_NORM_L2 ? 1 : 0,
            d_idx, d_dist);
        le = hipGetLastError();
    }
    if (part) (void)hipFreeAsync(part, (hipStream_t)stream);
    MTCHK(le);
    return 0;
}

extern "C" int mi_degensac_match_filter_dev(const int32_t *d_idx, const float *d_dist, int n1, float ratio, const int32_t *d_back_idx_or_null,
                                            int device, void *stream, uint8_t *d_keep)
{
    if (n1 < 0) { snprintf(mt_err, sizeof mt_err, "bad argument"); return MI_DEGENSAC_EINVAL; }
    MtDevGuard g; int rc = g.enter(device); if (rc) return rc;
    if (n1 == 0) return 0;
    hipLaunchKernelGGL(mt_filter_kernel, dim3((n1 + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_idx, d_dist, n1, ratio, d_back_idx_or_null, d_keep);
    MTCHK(hipGetLastError());
    return 0;
}

/* host pointers: stage, search both directions when the mutual check is asked for, filter, copy back */
extern "C" int mi_degensac_match(int norm, const void *desc1, int n1, const void *desc2, int n2, int dim, float ratio, int mutual, int device,
                                 int32_t *idx /* [n1,2] */, float *dist /* [n1,2] */, uint8_t *keep /* [n1] or NULL */)
{
    if (!desc1 || !desc2 || !idx || !dist || n1 < 0 || n2 < 0 || dim <= 0) { snprintf(mt_err, sizeof mt_err, "bad argument"); return MI_DEGENSAC_EINVAL; }
    MtDevGuard g; int rc = g.enter(device); if (rc) return rc;
    if (n1 == 0) return 0;
    const size_t esz = norm == MI_DEGENSAC_NORM_L2 ? 4 : 1, b1 = (size_t)n1 * dim * esz, b2 = (size_t)n2 * dim * esz;
    char *d1 = nullptr, *d2 = nullptr; int32_t *di = nullptr, *dbi = nullptr; float *dd = nullptr, *dbd = nullptr; uint8_t *dk = nullptr;
    struct Free { char *&a, *&b; int32_t *&c, *&d; float *&e, *&f; uint8_t *&g;
                  ~Free() { (void)hipFree(a); (void)hipFree(b); (void)hipFree(c); (void)hipFree(d); (void)hipFree(e); (void)hipFree(f); (void)hipFree(g);
                      } } fr{d1, d2, di, dbi, dd, dbd, dk};
    MTCHK(hipMalloc((void **)&d1, b1 ? b1 : 4)); MTCHK(hipMalloc((void **)&d2, b2 ? b2 : 4));
    MTCHK(hipMalloc((void **)&di, (size_t)n1 * 8)); MTCHK(hipMalloc((void **)&dd, (size_t)n1 * 8));
    MTCHK(hipMemcpy(d1, desc1, b1, hipMemcpyHostToDevice)); MTCHK(hipMemcpy(d2, desc2, b2, hipMemcpyHostToDevice));
    rc = mi_degensac_match_knn2_dev(norm, d1, n1, d2, n2, dim, device, nullptr, di, dd); if (rc) return rc;
    if (keep) {
        MTCHK(hipMalloc((void **)&dk, (size_t)n1));
        if (mutual && n2 > 0) {
            MTCHK(hipMalloc((void **)&dbi, (size_t)n2 * 8)); MTCHK(hipMalloc((void **)&dbd, (size_t)n2 * 8));
            rc = mi_degensac_match_knn2_dev(norm, d2, n2, d1, n1, dim, device, nullptr, dbi, dbd); if (rc) return rc;
        }
        rc = mi_degensac_match_filter_dev(di, dd, n1, ratio, dbi, device, nullptr, dk); if (rc) return rc;
    }
    MTCHK(hipDeviceSynchronize());
    MTCHK(hipMemcpy(idx, di, (size_t)n1 * 8, hipMemcpyDeviceToHost)); MTCHK(hipMemcpy(dist, dd, (size_t)n1 * 8, hipMemcpyDeviceToHost));
    if (keep) MTCHK(hipMemcpy(keep, dk, (size_t)n1, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int mi_degensac_kpts_to_xyA_dev(const float *d_kpts, int n, int device, void *stream, double *d_out)
{
    if (n < 0) { snprintf(mt_err, sizeof mt_err, "bad argument"); return MI_DEGENSAC_EINVAL; }
    MtDevGuard g; int rc = g.enter(device); if (rc) return rc;
    if (n == 0) return 0;
    hipLaunchKernelGGL(mt_kpts_to_xyA_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_kpts, n, d_out);
    MTCHK(hipGetLastError());
    return 0;
}
extern "C" int mi_degensac_kpts_to_xyA(const float *kpts, int n, int device, double *out)
{
    if (!kpts || !out || n < 0) { snprintf(mt_err, sizeof mt_err, "bad argument"); return MI_DEGENSAC_EINVAL; }
    MtDevGuard g; int rc = g.enter(device); if (rc) return rc;
    if (n == 0) return 0;
    float *dk = nullptr; double *dout = nullptr;
    struct Free { float *&a; double *&b; ~Free() { (void)hipFree(a); (void)hipFree(b); } } fr{dk, dout};
    MTCHK(hipMalloc((void **)&dk, (size_t)n * 16)); MTCHK(hipMalloc((void **)&dout, (size_t)n * 48));
    MTCHK(hipMemcpy(dk, kpts, (size_t)n * 16, hipMemcpyHostToDevice));
    rc = mi_degensac_kpts_to_xyA_dev(dk, n, device, nullptr, dout); if (rc) return rc;
    MTCHK(hipDeviceSynchronize());
    MTCHK(hipMemcpy(out, dout, (size_t)n * 48, hipMemcpyDeviceToHost));
    return 0;
}
