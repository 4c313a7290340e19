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
#include "seeq_verify_packed.h"
#include "seeq_order.h"
#include "seeq_post.h"

void seeq_launch_verify(int fw, int var, unsigned grid, hipStream_t st, const ScanArgs &a, const uint32_t *eq, const uint32_t *hit_col,
                        uint4 *cache)
{
   /* (6 / 5 waves per SIMD: the instances built for 8 / 6 spill ten registers each and are 2 - 4 % slower -- profiles/r04) */
#define SEEQ_VERIFY(WW, VV) hipLaunchKernelGGL((k_verify<WW, VV, (WW == 1 ? 6 : 5)>), dim3(grid), dim3(256), 0, st, a, eq, hit_col, cache)
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

void seeq_launch_emit_all(int fw, unsigned grid, unsigned vgrid, hipStream_t st, const ScanArgs &a, const uint32_t *eq, const uint32_t *hit_col, uint4 *cache)
{
   if (fw == 1) hipLaunchKernelGGL(k_emit_all<1>, dim3(grid), dim3(256), 0, st, a, eq, hit_col, cache, vgrid);
   else hipLaunchKernelGGL(k_emit_all<2>, dim3(grid), dim3(256), 0, st, a, eq, hit_col, cache, vgrid);
}

static_assert(ORDER_MAX_BLOCKS == SEEQ_ORDER_MAX_BLOCKS && ORDER_SCAN_BLOCK == SEEQ_ORDER_BLOCK, "seeq_order.h / seeq_post.h");

void seeq_launch_tiles_post(hipStream_t st, const FusedArgs &f, uint32_t nslices, uint32_t *bsum, uint32_t nb)
{
   hipLaunchKernelGGL(k_tiles_post, dim3(nb ? nb : 1, 3), dim3(256), 0, st, f, nslices, bsum, nb);
}

void seeq_launch_order(unsigned grid, hipStream_t st, const FusedArgs &f, uint32_t nslices, const uint32_t *bsum, uint32_t nb, uint4 *ent)
{
   hipLaunchKernelGGL(k_order, dim3(grid), dim3(256), 0, st, f, nslices, bsum, nb, ent);
}

void seeq_launch_bounds2(unsigned grid, hipStream_t st, const ScanArgs &a, uint4 *ent, uint32_t *hit_col)
{
   hipLaunchKernelGGL(k_bounds2, dim3(grid), dim3(256), 0, st, a, ent, hit_col);
}

void seeq_launch_verify_packed(int fw, int var, unsigned grid, hipStream_t st, const ScanArgs &a, const void *bases, const void *nmask, uint32_t stride,
                               uint32_t nstride, uint32_t read_len, uint64_t total_bytes, uint64_t ntotal_bytes, const uint32_t *eq, const uint32_t *hit_col, uint4 *cache)
{
   VerifyPacked pk;
   pk.bases = (const uint8_t *)bases; pk.nmask = (const uint8_t *)nmask; pk.stride = stride; pk.nstride = nstride; pk.read_len = read_len;
   pk.total_bytes = total_bytes; pk.ntotal_bytes = ntotal_bytes;
#define SEEQ_VERIFYP(WW, VV) hipLaunchKernelGGL((k_verify_packed<WW, VV>), dim3(grid), dim3(256), 0, st, a, pk, eq, hit_col, cache)
   if (fw == 1) {
      if (var == VERIFY_BEST) SEEQ_VERIFYP(1, VERIFY_BEST); else if (var == VERIFY_ALL) SEEQ_VERIFYP(1, VERIFY_ALL); else SEEQ_VERIFYP(1, VERIFY_ANY);
   } else {
      if (var == VERIFY_BEST) SEEQ_VERIFYP(2, VERIFY_BEST); else if (var == VERIFY_ALL) SEEQ_VERIFYP(2, VERIFY_ALL); else SEEQ_VERIFYP(2, VERIFY_ANY);
   }
#undef SEEQ_VERIFYP
   hipLaunchKernelGGL(k_nh_top, dim3(1), dim3(256), 0, st, a);
}
