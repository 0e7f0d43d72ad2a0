/* Included once per translation unit after the device headers, with DG_T defined (512 or 256). */
#define DG_VCAT2(a, b, c) a##b##c
#define DG_VCAT(a, b, c) DG_VCAT2(a, b, c)
#include "dg_variant.h"

hipError_t DG_VCAT(dg_variant_, DG_T, _init)(const unsigned C[8][32], const unsigned Ct[32][8], const unsigned G[32], const unsigned T[31][32], int max_lds,
    int static_lds[2])
{
    hipError_t e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(dg_rng_T), T, sizeof(unsigned) * 31 * 32)) != hipSuccess) return e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(dg_rng_C), C, sizeof(unsigned) * 8 * 32)) != hipSuccess) return e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(dg_rng_Ct), Ct, sizeof(unsigned) * 32 * 8)) != hipSuccess) return e;
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(dg_rng_G), G, sizeof(unsigned) * 32)) != hipSuccess) return e;
    hipFuncAttributes fa;
    const void *kf[3] = {(const void *)dg_find_fundamental_kernel<DG_T, DG_MODE_LDS>, (const void *)dg_find_fundamental_kernel<DG_T, DG_MODE_POOL_LDS>,
                         (const void *)dg_find_fundamental_kernel<DG_T, DG_MODE_HBM>};
    const void *kh[3] = {(const void *)dg_find_homography_kernel<DG_T, DG_MODE_LDS>, (const void *)dg_find_homography_kernel<DG_T, DG_MODE_POOL_LDS>,
                         (const void *)dg_find_homography_kernel<DG_T, DG_MODE_HBM>};
    for (int h = 0; h < 2; h++) {
        const void **k = h ? kh : kf;
        /* static LDS of the largest of the three placement instantiations: it sizes the residency clamp and the image of a
         * pair that is set aside (which must hold the whole dg_f_shared) */
        static_lds[h] = (int)sizeof(dg_f_shared);
        for (int m = 0; m < 3; m++) {
            if ((e = hipFuncGetAttributes(&fa, k[m])) != hipSuccess) return e;
            if ((int)fa.sharedSizeBytes > static_lds[h]) static_lds[h] = (int)fa.sharedSizeBytes;
        }
        for (int m = 0; m < 2; m++)
            if ((e = hipFuncSetAttribute(k[m], hipFuncAttributeMaxDynamicSharedMemorySize, max_lds - static_lds[h] - 256)) != hipSuccess) return e;
    }
    return hipSuccess;
}

static const void *DG_VCAT(dg_variant_, DG_T, _fn)(int homography, int mode)
{
    if (!homography) {
        if (mode == DG_MODE_LDS)      return (const void *)dg_find_fundamental_kernel<DG_T, DG_MODE_LDS>;
        if (mode == DG_MODE_POOL_LDS) return (const void *)dg_find_fundamental_kernel<DG_T, DG_MODE_POOL_LDS>;
        return (const void *)dg_find_fundamental_kernel<DG_T, DG_MODE_HBM>;
    }
    if (mode == DG_MODE_LDS)      return (const void *)dg_find_homography_kernel<DG_T, DG_MODE_LDS>;
    if (mode == DG_MODE_POOL_LDS) return (const void *)dg_find_homography_kernel<DG_T, DG_MODE_POOL_LDS>;
    return (const void *)dg_find_homography_kernel<DG_T, DG_MODE_HBM>;
}

hipError_t DG_VCAT(dg_variant_, DG_T, _resident)(int homography, int mode, size_t dyn, int *blocks_per_cu)
{
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, DG_VCAT(dg_variant_, DG_T, _fn)(homography, mode), DG_T, dyn);
}

hipError_t DG_VCAT(dg_variant_, DG_T, _launch)(int homography, int mode, int grid, size_t dyn, hipStream_t stream, const dg_args &A)
{
    const dim3 g(grid), b(DG_T);
    if (!homography) {
        if (mode == DG_MODE_LDS)           hipLaunchKernelGGL((dg_find_fundamental_kernel<DG_T, DG_MODE_LDS>), g, b, dyn, stream, A);
        else if (mode == DG_MODE_POOL_LDS) hipLaunchKernelGGL((dg_find_fundamental_kernel<DG_T, DG_MODE_POOL_LDS>), g, b, dyn, stream, A);
        else                               hipLaunchKernelGGL((dg_find_fundamental_kernel<DG_T, DG_MODE_HBM>), g, b, 0, stream, A);
    } else {
        if (mode == DG_MODE_LDS)           hipLaunchKernelGGL((dg_find_homography_kernel<DG_T, DG_MODE_LDS>), g, b, dyn, stream, A);
        else if (mode == DG_MODE_POOL_LDS) hipLaunchKernelGGL((dg_find_homography_kernel<DG_T, DG_MODE_POOL_LDS>), g, b, dyn, stream, A);
        else                               hipLaunchKernelGGL((dg_find_homography_kernel<DG_T, DG_MODE_HBM>), g, b, 0, stream, A);
    }
    return hipGetLastError();
}
