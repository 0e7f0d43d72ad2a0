/* Host-visible interface of one compiled kernel variant (workgroup size is a compile-time constant of the device
 * code, so each variant is its own translation unit: mi_degensac.hip = 512 threads, mi_degensac_t256.hip = 256,
 * mi_degensac_t128.hip = 128). */
#ifndef DG_VARIANT_H
#define DG_VARIANT_H
#include <hip/hip_runtime.h>
#include "dg_kernel_common.h"

/* placement of the per-pair arrays: DG_MODE_HBM = point set and sampler pool in the HBM workspace (L2),
 * DG_MODE_LDS = both in LDS, DG_MODE_POOL_LDS = pool in LDS, points in the workspace */
enum { DG_MODE_HBM = 0, DG_MODE_LDS = 1, DG_MODE_POOL_LDS = 2 };

/* init: uploads the RNG tables of the variant's translation unit, raises the dynamic-LDS limit of its kernels and
 * reports the static LDS bytes of the F / H kernels.  resident: workgroups of that kernel one CU keeps resident with
 * `dyn` bytes of dynamic LDS.  launch: enqueues `grid` persistent workgroups (they pull pairs from A.ticket). */
#define DG_VARIANT_DECL(T_) \
    hipError_t dg_variant_##T_##_init(const unsigned C[8][32], const unsigned Ct[32][8], const unsigned G[32], const unsigned T[31][32], int max_lds, \
        int static_lds[2]); \
    hipError_t dg_variant_##T_##_resident(int homography, int mode, size_t dyn, int *blocks_per_cu); \
    hipError_t dg_variant_##T_##_launch(int homography, int mode, int grid, size_t dyn, hipStream_t stream, const dg_args &A);
DG_VARIANT_DECL(512)
DG_VARIANT_DECL(256)
DG_VARIANT_DECL(128)
#endif /* DG_VARIANT_H */
