/*
 * seeq_fused_post.h -- the two small kernels behind every one-pass scan kernel (k_pair, k_stream, k_direct): the reduction
 * of the per-wave partial counts and the ordering of k_direct's hit slices.  Main translation unit only.
 */
#ifndef SEEQ_FUSED_POST_H_
#define SEEQ_FUSED_POST_H_

__global__ __launch_bounds__(256) void k_fused_post(FusedArgs a, uint32_t nslices) { fused_post_body(a, nslices); }

/* Slices -> ordered (hit_start, hit_line).  tile_hits / tile_cl hold exclusive prefixes by now.
   One wave per slice (k_direct). */
__global__ __launch_bounds__(256) void k_fused_reorder(FusedArgs a, uint32_t nslices, uint32_t *hit_start, uint32_t *hit_line)
{
   const Counters *c = a.cnt;
   if (c->overflow & 2u) return;
   const uint32_t lane = threadIdx.x & 63;
   for (uint32_t sl = blockIdx.x * 4 + (threadIdx.x >> 6); sl < nslices; sl += gridDim.x * 4) {      /* one wave per slice */
      const uint32_t n = a.wg_hits[sl];
      const uint4 *slice = a.tmp + (size_t)sl * a.slice_cap;
      for (uint32_t i = lane; i < n; i += 64) {
         const uint4 e = slice[i];
         const uint32_t dst = a.tile_hits[e.x] + e.y;
         hit_start[dst] = e.z;
         hit_line[dst] = (uint32_t)(c->lines + a.tile_cl[e.x] + e.w + 1);      /* 1-based, reference seeq.c:377 */
      }
   }
}

#endif
