/*
 * seeq_fused_post.h -- the two small kernels behind every one-pass scan kernel (k_pair, k_stream, k_direct): the reduction
 * of the per-wave partial counts and the ordering of k_direct's hit slices.  Main translation unit only.
 */
#ifndef SEEQ_FUSED_POST_H_
#define SEEQ_FUSED_POST_H_

/* After k_stream / k_direct: reduce the per-slice partial counts (no atomics in the hot kernels)
   and publish the hit-line count of the segment, or the overflow. */
__global__ __launch_bounds__(256) void k_fused_post(FusedArgs a, uint32_t nslices)
{
   __shared__ uint32_t s_red[4][4];
   uint32_t lines = 0, hdrs = 0, hits = 0, mx = 0, ovf = 0, lastnl = 0, flags = 0, busy = 0, crowded = 0;
   /* one 16-byte load per slice, four slices per thread in flight (this kernel is one workgroup on an idle chip: its time is
      the latency of its loads) */
   const uint4 *part = reinterpret_cast<const uint4 *>(a.wg_part);
   for (uint32_t i0 = threadIdx.x; i0 < nslices; i0 += 1024) {
      uint4 pv[4];
      uint32_t lv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
         const uint32_t i = i0 + 256u * u;
         pv[u] = i < nslices ? part[i] : make_uint4(0u, 0u, 0u, 0u);
         lv[u] = a.wg_lastnl && i < nslices ? a.wg_lastnl[i] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
         lastnl = lv[u] > lastnl ? lv[u] : lastnl;
         lines += pv[u].x;
         hdrs += pv[u].y;
         const uint32_t h = pv[u].z;
         flags |= pv[u].w;
         busy += pv[u].x != 0;                               /* waves that saw text / of them, those drowning in made-up candidates */
         crowded += (pv[u].w >> 3) & 1u;
         hits += h & 0x7FFFFFFFu;
         mx = (h & 0x7FFFFFFFu) > mx ? (h & 0x7FFFFFFFu) : mx;
         ovf |= h >> 31;
      }
   }
#pragma unroll
   for (int d = 32; d >= 1; d >>= 1) {
      lines += __shfl_xor(lines, d, 64);
      hdrs += __shfl_xor(hdrs, d, 64);
      hits += __shfl_xor(hits, d, 64);
      const uint32_t o = __shfl_xor(mx, d, 64);
      mx = o > mx ? o : mx;
      ovf |= __shfl_xor(ovf, d, 64);
      flags |= __shfl_xor(flags, d, 64);
      busy += __shfl_xor(busy, d, 64);
      crowded += __shfl_xor(crowded, d, 64);
      const uint32_t ol = __shfl_xor(lastnl, d, 64);
      lastnl = ol > lastnl ? ol : lastnl;
   }
   __shared__ uint32_t s_last[4], s_flags[4], s_busy[4], s_crowded[4];
   const int w = threadIdx.x >> 6;
   if ((threadIdx.x & 63) == 0) { s_last[w] = lastnl; s_flags[w] = flags; s_busy[w] = busy; s_crowded[w] = crowded; }
   if ((threadIdx.x & 63) == 0) { s_red[w][0] = lines; s_red[w][1] = hdrs; s_red[w][2] = hits; s_red[w][3] = mx | (ovf << 31); }
   __syncthreads();
   if (threadIdx.x == 0) {
      lines = hdrs = hits = mx = ovf = 0;
      for (int k = 0; k < 4; k++) {
         lines += s_red[k][0]; hdrs += s_red[k][1]; hits += s_red[k][2];
         const uint32_t m = s_red[k][3] & 0x7FFFFFFFu;
         mx = m > mx ? m : mx;
         ovf |= s_red[k][3] >> 31;
      }
      Counters *c = a.cnt;
      /* what the scan kernel noticed about the text (kept out of its own code path: the scan of the NEXT segment may
         be running while this segment's post-pass reads these) */
      flags = s_flags[0] | s_flags[1] | s_flags[2] | s_flags[3];
      if (flags & 1u) {
         c->dirty |= 1u;
         if ((a.options & MASK_NONDNA) && !a.pair) c->overflow |= 16u;     /* SQ_CONVERT / SQ_IGNORE: k_stream is only exact on clean text -> re-run (k_pair's candidates are verified anyway) */
      }
      if (flags & 4u) {
         c->dirty |= 1u;                                      /* skip bytes in a warm-up window / a NUL: the hit lines are candidates */
         /* SQ_IGNORE on text that is mostly skip bytes (FASTQ quality lines): nearly every line becomes a candidate and the
            exact pass scans them all -- the per-line kernel does that in one pass: re-run there, and stay.  (Decided by the
            waves: more than half of those that saw text made up more candidates than a quarter of their lines.) */
         busy = s_busy[0] + s_busy[1] + s_busy[2] + s_busy[3];
         crowded = s_crowded[0] + s_crowded[1] + s_crowded[2] + s_crowded[3];
         if ((a.options & MASK_NONDNA) == SQ_IGNORE && crowded * 2 > busy) c->overflow |= 16u;
      }
      if (flags & 2u) c->overflow |= 32u;                     /* re-run once with the long-line variant (then kept) */
      c->seg_nlines = lines;
      c->seg_nheaders = hdrs;
      /* capacity wanted next time: every slice as large as the fullest one, plus slack */
      const uint64_t need = (uint64_t)mx * nslices + (uint64_t)nslices * 64;
      if (need > c->need_hitlines) c->need_hitlines = need > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)need;
      if (ovf) { atomicOr(&c->overflow, 2u); hits = 0; }
      /* a hit list overflowed, in this segment or in an earlier one: the run is void (seeqdevScanFetch grows the workspace
         and runs it again) and k_stream_reorder / k_fused_reorder write nothing any more -- so no later kernel of this
         run may look at the (stale) hit arrays either: no hit lines from here on */
      if (c->overflow & 2u) hits = 0;
      c->seg_nhitlines = hits;
      c->seg_nrec = hits;                                   /* (k_seg_mid's job; the slices cannot hold more than cap_hitlines) */
      if (hits > c->need_hitlines) c->need_hitlines = hits;
      c->seg_novf = 0;
      lastnl = 0;
      for (int k = 0; k < 4; k++) lastnl = s_last[k] > lastnl ? s_last[k] : lastnl;
      c->seg_last_nl = lastnl;
   }
}

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
