/* libmi_degensac.so — second instantiation of the device code: 256-thread workgroups (4 waves), two resident
 * workgroups per CU.  Large batches are latency-bound chains of small dense solves per pair, so twice the pairs
 * in flight per CU beats twice the waves per pair; the host side (mi_degensac.hip) picks the
 * variant per launch. */
#include <hip/hip_runtime.h>
#define DG_T 256
#include "dg_dev_small.h"
#include "dg_wg.h"
#include "dg_geom.h"
#include "dg_kernel_common.h"
#include "dg_lsq.h"
#include "dg_kernel_f.h"
#include "dg_kernel_f_main.h"
#include "dg_kernel_h.h"
#include "dg_variant_impl.h"
