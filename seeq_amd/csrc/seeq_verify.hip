/*
 * seeq_verify.hip -- translation unit of the round-4 post-pass kernels (k_verify: seeq_verify.h).  Device code for gfx950;
 * the host side is the launcher seeq_device.hip calls (seeq_post.h).
 */
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "seeq_amd.h"
#include "seeq_kernel_core.h"
#include "seeq_types.h"
#include "seeq_scan_common.h"
#include "seeq_direct.h"
#include "seeq_exact1.h"
#include "seeq_verify.h"
#include "seeq_post.h"

void seeq_launch_verify(int fw, int var, unsigned grid, hipStream_t st, const ScanArgs &a, const uint32_t *eq, const uint32_t *hit_col,
                        uint4 *cache)
{
#define SEEQ_VERIFY(WW, VV) hipLaunchKernelGGL((k_verify<WW, VV>), dim3(grid), dim3(256), 0, st, a, eq, hit_col, cache)
   if (fw == 1) {
      if (var == VERIFY_BEST) SEEQ_VERIFY(1, VERIFY_BEST); else if (var == VERIFY_ALL) SEEQ_VERIFY(1, VERIFY_ALL); else SEEQ_VERIFY(1, VERIFY_ANY);
   } else {
      if (var == VERIFY_BEST) SEEQ_VERIFY(2, VERIFY_BEST); else if (var == VERIFY_ALL) SEEQ_VERIFY(2, VERIFY_ALL); else SEEQ_VERIFY(2, VERIFY_ANY);
   }
#undef SEEQ_VERIFY
   hipLaunchKernelGGL(k_nh_top, dim3(1), dim3(256), 0, st, a);
}

void seeq_launch_emit1(unsigned grid, hipStream_t st, const ScanArgs &a, const uint4 *cache)
{
   hipLaunchKernelGGL(k_emit1, dim3(grid), dim3(256), 0, st, a, cache);
}
