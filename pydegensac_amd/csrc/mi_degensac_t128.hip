/* libmi_degensac.so — third instantiation of the device code: 128-thread workgroups (2 waves), 128-sample chunks.
 * At 256 VGPRs a CU holds 8 waves, so this variant keeps FOUR pairs resident per CU (when their LDS fits): a pair
 * is mostly a chain of small serial solves on one wave, and chains of different pairs interleave on the SIMDs.  The
 * host side (mi_degensac.hip) picks the variant per launch. */
#include <hip/hip_runtime.h>
#define DG_T 128
#define DG_CHUNK 128
#include "dg_dev_small.h"
#include "dg_wg.h"
#include "dg_geom.h"
#include "dg_kernel_common.h"
#include "dg_lsq.h"
#include "dg_kernel_f.h"
#include "dg_kernel_f_main.h"
#include "dg_kernel_h.h"
#include "dg_variant_impl.h"
