/* libmi_degensac.so — third instantiation of the device code: 128-thread workgroups (2 waves), 64-sample chunks
 * (27.9 KB of static LDS).  At 256 VGPRs a CU holds 8 waves, so this variant keeps FOUR pairs resident per CU: a pair
 * is mostly a chain of small serial solves on one wave, and chains of different pairs interleave on the SIMDs.  Its
 * steady state is the best of the three variants (sum of pair times / resident slots = 127 ms for 4096 C2 pairs against
 * 166 ms at 256 threads), its slowest pairs the slowest (170 vs 104 ms), so a 4096-pair batch takes the same 227 ms:
 * on request only (tuning = 3).  The host side (mi_degensac.hip) picks the variant per launch. */
#include <hip/hip_runtime.h>
#define DG_T 128
#define DG_CHUNK 64
#include "dg_dev_small.h"
#include "dg_wg.h"
#include "dg_geom.h"
#include "dg_kernel_common.h"
#include "dg_lsq.h"
#include "dg_kernel_f.h"
#include "dg_kernel_f_main.h"
#include "dg_kernel_h.h"
#include "dg_variant_impl.h"
