/*
 * seeq_order.h -- behind the line-agnostic scan kernels on read-length lines (k_pair, k_stream's line mode): from the per-wave
 * hit slices to the ordered per-line arrays the exact pass reads, in three launches (round 4; seven before):
 *
 *   k_tiles_post  the reduction of the per-wave counts (fused_post_body) in one workgroup, and beside it the per-tile arrays
 *                 (hits, counted lines) scanned LOCALLY -- exclusive offsets inside blocks of 2 048 tiles written in place, the
 *                 blocks' sums to bsum[].  No second level: there are <= 1 024 blocks per segment, and
 *   k_order       every workgroup scans the block sums itself into LDS (one load per thread) before it moves its slices'
 *                 entries to their places: ONE 16-byte store per entry {position, line number, flags | column} -- the four
 *                 4-byte stores per entry of k_stream_reorder were 5 M scattered write transactions per segment, its whole time;
 *   k_bounds2     entry by entry (coalesced 16-byte loads): repeats of a line dropped, the 2 % of entries whose line starts
 *                 before their tile get its start from the 128 bytes before the candidate held in registers (eight loads in
 *                 flight, a second round for longer lines; k_stream_bounds walked 64 bytes per dependent step, tile by
 *                 tile), and the arrays hit_start / hit_line / hit_col written in order.
 *
 * Long-line input (k_stream's LL variant, the window walk, leaders) keeps k_fused_post / k_scanset_* / k_stream_reorder /
 * k_stream_bounds: there a line's start may lie thousands of tiles back, and the per-tile prefix is what finds it.
 */
#ifndef SEEQ_ORDER_H_
#define SEEQ_ORDER_H_

#define ORDER_SCAN_ITEMS 8
#define ORDER_SCAN_BLOCK (256 * ORDER_SCAN_ITEMS)          /* tiles per block of the local scan */
#define ORDER_MAX_BLOCKS 1024                              /* (a 3.75 GiB segment of 8 KB tiles: 240) */

__global__ __launch_bounds__(256) void k_tiles_post(FusedArgs f, uint32_t nslices, uint32_t *bsum, uint32_t nb)
{
   if (blockIdx.y == 2) {                                  /* one workgroup: the per-wave counts, the flags, the overflow report */
      if (blockIdx.x == 0) fused_post_body(f, nslices);
      return;
   }
   __shared__ uint32_t s_wave[4];
   uint32_t *arr = blockIdx.y == 0 ? f.tile_hits : f.tile_cl;
   const uint32_t n = f.ntiles, base = blockIdx.x * ORDER_SCAN_BLOCK;
   uint32_t item[ORDER_SCAN_ITEMS], v = 0;
#pragma unroll
   for (int k = 0; k < ORDER_SCAN_ITEMS; k++) {
      const uint32_t i = base + threadIdx.x * ORDER_SCAN_ITEMS + k;
      item[k] = i < n ? arr[i] : 0u;
      v += item[k];
   }
   uint32_t tot;
   uint32_t ex = block_excl_scan(v, &tot, s_wave);
#pragma unroll
   for (int k = 0; k < ORDER_SCAN_ITEMS; k++) {
      const uint32_t i = base + threadIdx.x * ORDER_SCAN_ITEMS + k;
      if (i < n) arr[i] = ex;
      ex += item[k];
   }
   if (threadIdx.x == 0) bsum[blockIdx.y * nb + blockIdx.x] = tot;
}

/* exclusive scan of bsum[0 .. nb) and bsum[nb .. 2 nb) into LDS (nb <= ORDER_MAX_BLOCKS) */
__device__ __forceinline__ void order_block_prefix(const uint32_t *bsum, uint32_t nb, uint32_t *s_ph, uint32_t *s_pc, uint32_t *s_wave)
{
   uint32_t run_h = 0, run_c = 0;
   for (uint32_t b0 = 0; b0 < nb; b0 += 256u) {
      const uint32_t i = b0 + threadIdx.x;
      const uint32_t vh = i < nb ? bsum[i] : 0u, vc = i < nb ? bsum[nb + i] : 0u;
      uint32_t th, tc;
      const uint32_t eh = block_excl_scan(vh, &th, s_wave), ec = block_excl_scan(vc, &tc, s_wave);
      if (i < nb) { s_ph[i] = run_h + eh; s_pc[i] = run_c + ec; }
      run_h += th; run_c += tc;
   }
   __syncthreads();
}

/* entries: {segment-relative (biased) position: the line's start, or the candidate itself when the line starts before its tile;
            1-based counted line number; bit 0: unresolved, bits 1..: the candidate's column in its line (resolved entries); 0} */
__global__ __launch_bounds__(256) void k_order(FusedArgs f, uint32_t nslices, const uint32_t *bsum, uint32_t nb, uint4 *ent)
{
   __shared__ uint32_t s_ph[ORDER_MAX_BLOCKS], s_pc[ORDER_MAX_BLOCKS], s_wave[4];
   order_block_prefix(bsum, nb, s_ph, s_pc, s_wave);
   const Counters *c = f.cnt;
   if (c->overflow & 2u) return;
   const uint32_t lane = threadIdx.x & 63;
   const uint32_t lines0 = (uint32_t)c->lines;
   for (uint32_t sl = blockIdx.x * 4 + (threadIdx.x >> 6); sl < nslices; sl += gridDim.x * 4) {      /* one wave per slice */
      const uint32_t n = f.wg_hits[sl];
      const uint4 *slice = f.tmp + (size_t)sl * f.slice_cap;
      for (uint32_t i0 = lane; i0 < n; i0 += 256) {          /* four entries per lane in flight */
         uint4 e[4];
         uint32_t th[4], tc[4];
#pragma unroll
         for (int u = 0; u < 4; u++) e[u] = i0 + 64u * u < n ? slice[i0 + 64u * u] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
         for (int u = 0; u < 4; u++) {
            const uint32_t tile = e[u].x & 0x7FFFFFFFu;
            const bool ok = i0 + 64u * u < n;
            th[u] = ok ? f.tile_hits[tile] + s_ph[tile / ORDER_SCAN_BLOCK] : 0u;
            tc[u] = ok ? f.tile_cl[tile] + s_pc[tile / ORDER_SCAN_BLOCK] : 0u;
         }
#pragma unroll
         for (int u = 0; u < 4; u++) {
            if (i0 + 64u * u >= n) continue;
            const uint32_t dst = th[u] + (e[u].y & 0x1FFFu);
            ent[dst] = make_uint4(e[u].z, lines0 + tc[u] + e[u].w + 1u /* reference seeq.c:377 */, (e[u].x >> 31) | (((e[u].y >> 13) & 0x3FFFFu) << 1), 0u);
         }
      }
   }
}

/* last '\n' among the 16 bytes of v: index + 1, or 0 */
__device__ __forceinline__ uint32_t order_last_nl16(const fused_v4u &v)
{
   const uint32_t g3 = nl_flags(v.w), g2 = nl_flags(v.z), g1 = nl_flags(v.y), g0 = nl_flags(v.x);
   if (g3) return 13u + ((31u - (uint32_t)__builtin_clz(g3)) >> 3);
   if (g2) return 9u + ((31u - (uint32_t)__builtin_clz(g2)) >> 3);
   if (g1) return 5u + ((31u - (uint32_t)__builtin_clz(g1)) >> 3);
   if (g0) return 1u + ((31u - (uint32_t)__builtin_clz(g0)) >> 3);
   return 0u;
}

/* Offset just after the last '\n' in text[0, hi), or 0 when there is none: 128 bytes per step, the eight loads in flight together */
__device__ __forceinline__ uint64_t order_line_start(const uint8_t *text, uint64_t hi)
{
   uint64_t q = hi;
   while (q >= 128) {
      fused_v4u v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const fused_v4u_unaligned *>(text + q - 128 + 16 * u);
#pragma unroll
      for (int u = 7; u >= 0; u--) {
         const uint32_t r = order_last_nl16(v[u]);
         if (r) return q - 128 + 16 * u + r;
      }
      q -= 128;
   }
   while (q > 0) { if (text[q - 1] == '\n') return q; q--; }
   return 0;
}

__global__ __launch_bounds__(256) void k_bounds2(ScanArgs a, const uint4 *ent, uint32_t *hit_col)
{
   Counters *c = a.cnt;
   const uint32_t nhl = c->seg_nhitlines;
   const uint32_t stride = gridDim.x * 256;
   const uint64_t segb = a.seg_base + a.pos_bias;         /* the segment proper (a.seg_base is the biased base) */
   const uint32_t prev0 = c->prev_hit_line;
   for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < nhl; k += stride) {
      const uint4 e = ent[k];
      const uint32_t prev = k ? ent[k - 1].y : prev0;
      const uint32_t unresolved = e.z & 1u, col = e.z >> 1;
      a.hit_line[k] = e.y;
      if (e.y == prev) {                                  /* a repeat: keep the candidate's position for the exact pass's window */
         hit_col[k] = unresolved ? e.x : e.x + col;
         /* windows: the line's first candidate belongs to the segment before this one, whose exact pass could not know of this
            one -- the run is void, the next one scans candidate lines to their ends (seeqdevScanFetch) */
         if (a.window_ok && e.y == prev0) atomicOr(&c->overflow, 128u);
         a.hit_start[k] = 0xFFFFFFFFu;
         continue;
      }
      if (!unresolved) { a.hit_start[k] = e.x; hit_col[k] = col; continue; }
      const uint64_t hp = a.seg_base + e.x;               /* a byte of the line (inside the segment); never '\n' */
      if (hp >= a.nbytes || hp < segb || hp >= segb + a.seg_len) {       /* cannot be: an entry the scan kernel never wrote -- fail loudly, touch nothing */
         atomicOr(&c->overflow, 64u);
         a.hit_start[k] = 0xFFFFFFFFu;
         hit_col[k] = 0u;
         continue;
      }
      const uint64_t q = order_line_start(a.text, hp);
      if (q < a.seg_base) { atomicOr(&c->overflow, 8u); a.hit_start[k] = 0xFFFFFFFFu; hit_col[k] = 0u; }
      else {
         hit_col[k] = (uint32_t)(hp - q);
         a.hit_start[k] = (uint32_t)(q - a.seg_base);
      }
   }
}

#endif
