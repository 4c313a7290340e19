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
            1-based counted line number; bit 0: unresolved, bits 1..: the candidate's column in its line (resolved entries);
            bit 1: a line MARKER (k_pair under SQ_IGNORE: the line holds a skipped byte -- scan it whole), bit 0: ... that stands on a candidate of the walk,
            bit 2 (set by k_bounds2): ... that did not hold -- a marker no more} */
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
            const uint32_t tile = e[u].x & 0x1FFFFFFFu;          /* (bit 31: unresolved, bit 30: a line marker of k_pair under SQ_IGNORE, bit 29: ... on a candidate) */
            const bool ok = i0 + 64u * u < n;
            th[u] = ok ? f.tile_hits[tile] + s_ph[tile / ORDER_SCAN_BLOCK] : 0u;
            tc[u] = ok ? f.tile_cl[tile] + s_pc[tile / ORDER_SCAN_BLOCK] : 0u;
         }
#pragma unroll
         for (int u = 0; u < 4; u++) {
            if (i0 + 64u * u >= n) continue;
            const uint32_t dst = th[u] + (e[u].y & 0x1FFFu);
            ent[dst] = make_uint4(e[u].z, lines0 + tc[u] + e[u].w + 1u /* reference seeq.c:377 */, (e[u].x >> 31) | (((e[u].y >> 13) & 0x3FFFFu) << 1), (e[u].x >> 29) & 3u);
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

/* SQ_IGNORE, a line MARKER of k_pair (seeq_pair.h IG): its line starts at text[q] and runs to the next newline (or the buffer's end).  Can a skipped
 * byte hide an occurrence in it?  Only if the line holds (1) a skipped byte (libseeq.c:265-266: anything but A C G T U N in either case), (2) at least
 * m - tau characters that are not skipped, and (3) -- the frequency bound -- at least n_b - tau copies of the base b the pattern's plain positions hold
 * most often (ScanArgs.ig_bval / ig_bmask / ig_need: an occurrence with <= tau errors keeps all but tau of them; FASTQ quality lines, which hold a few
 * A C G but no T, fail it for every pattern with tau + 1 T's).  k_pair has checked (1) and (2) for the lines it could see whole; all three are checked
 * here, four bytes at a time. */
/* the counts of up to four bytes in front of a newline: w = the word, nby = its bytes that count (0 .. 4) */
__device__ __forceinline__ void order_marker_word(uint32_t w, uint32_t nby, uint32_t bm, uint32_t bv, uint32_t &ns, uint32_t &nb)
{
   const uint32_t keepm = nby >= 4u ? 0xFFFFFFFFu : (1u << (8u * nby)) - 1u;
   const uint32_t y = (w ^ __builtin_amdgcn_perm(0x474E5554u, 0x43FF41FFu, w & 0x07070707u)) & keepm;    /* (as pair_nonbase_mask32, seeq_pair.h: no newline among these bytes) */
   ns += (uint32_t)__popc((((y & 0x5F5F5F5Fu) + 0x7F7F7F7Fu) | y) & 0x80808080u);                        /* bytes outside the alphabet */
   const uint32_t z = ((w & bm) ^ bv) | ~keepm;
   nb += (uint32_t)__popc(~(((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u);                       /* copies of the base */
}

__device__ __forceinline__ bool order_marker_holds(const ScanArgs &a, uint64_t q)
{
   const uint8_t *text = a.text;
   uint32_t ns = 0, nb = 0, len = 0;
   const uint32_t bm = a.ig_bmask * 0x01010101u, bv = a.ig_bval * 0x01010101u;
   uint64_t p = q;
   for (; p + 16 <= a.nbytes; p += 16) {                   /* sixteen bytes a step, to the line's newline (a marker may stand anywhere in its line) */
      const fused_v4u v = *reinterpret_cast<const fused_v4u_unaligned *>(text + p);
      const uint32_t f0 = nl_flags(v.x), f1 = nl_flags(v.y), f2 = nl_flags(v.z), f3 = nl_flags(v.w);
      if ((f0 | f1 | f2 | f3) == 0u) {
         order_marker_word(v.x, 4u, bm, bv, ns, nb); order_marker_word(v.y, 4u, bm, bv, ns, nb);
         order_marker_word(v.z, 4u, bm, bv, ns, nb); order_marker_word(v.w, 4u, bm, bv, ns, nb);
         len += 16u;
         if (len > (1u << 20)) return true;                 /* (a line of a megabyte: not this kernel's text -- the exact pass decides) */
         continue;
      }
      /* the newline is among these sixteen: the bytes in front of it */
      const uint32_t wf = f0 ? 0u : f1 ? 1u : f2 ? 2u : 3u;
      const uint32_t fw = f0 ? f0 : f1 ? f1 : f2 ? f2 : f3;
      const uint32_t nby = (uint32_t)__builtin_ctz(fw) >> 3;
      order_marker_word(v.x, wf > 0u ? 4u : nby, bm, bv, ns, nb);
      if (wf >= 1u) order_marker_word(v.y, wf > 1u ? 4u : nby, bm, bv, ns, nb);
      if (wf >= 2u) order_marker_word(v.z, wf > 2u ? 4u : nby, bm, bv, ns, nb);
      if (wf >= 3u) order_marker_word(v.w, nby, bm, bv, ns, nb);
      len += 4u * wf + nby;
      return ns != 0u && len - ns >= a.ig_thr && nb >= a.ig_need;
   }
   for (; p < a.nbytes && text[p] != '\n'; p++) {
      const uint32_t ch = text[p], up = ch & 0xDFu;
      if (!(up == 'A' || up == 'C' || up == 'G' || up == 'T' || up == 'U' || up == 'N')) ns++;
      if (((ch & a.ig_bmask) ^ a.ig_bval) == 0u) nb++;
      len++;
   }
   return ns != 0u && len - ns >= a.ig_thr && nb >= a.ig_need;
}

__global__ __launch_bounds__(256) void k_bounds2(ScanArgs a, uint4 *ent, uint32_t *hit_col)
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
      /* SQ_IGNORE: the entry before me may be a line marker that its own thread DROPS (a marker made unseen for a line that turns out to hold no
         skipped byte: it stands on the tile's last pair, IN FRONT of what the next tile finds in the line) -- then I am the line's first entry,
         not its repeat.  Its verdict is recomputed here (its thread's lies in bit 2 of its word 3, which may not be written yet). */
      uint32_t repeat = e.y == prev ? 1u : 0u;
      if (repeat && k != 0u && a.ig_thr != 0u) {
         const uint4 pe = ent[k - 1];
         const uint32_t pprev = k > 1u ? ent[k - 2].y : prev0;
         if ((pe.w & 3u) == 2u && pprev != pe.y) {        /* a pure marker, the first entry of this line */
            const uint64_t pq = (pe.z & 1u) ? order_line_start(a.text, a.seg_base + pe.x) : a.seg_base + pe.x;
            const uint32_t pholds = (pq < a.seg_base || order_marker_holds(a, pq)) ? 1u : 0u;
            if (!pholds) repeat = 0u;
         }
      }
      if (repeat) {                                       /* a repeat: keep the candidate's position for the exact pass's window */
         hit_col[k] = unresolved ? e.x : e.x + col;
         /* windows: the line's first candidate belongs to the segment before this one, whose exact pass could not know of this
            one -- the run is void, the next one scans candidate lines to their ends (seeqdevScanFetch).  (SQ_IGNORE: every segment's first
            line end comes with a marker made unseen -- it counts only when the line really holds a skipped byte and enough characters.) */
         if (a.window_ok && e.y == prev0) {
            bool voids = true;
            if (a.ig_thr && (e.w & 2u) && unresolved) {
               const uint64_t hp0 = a.seg_base + e.x;
               voids = hp0 < a.nbytes && order_marker_holds(a, order_line_start(a.text, hp0));
            }
            if (voids) atomicOr(&c->overflow, 128u);
         }
         a.hit_start[k] = 0xFFFFFFFFu;
         continue;
      }
      if (!unresolved) {
         /* (SQ_IGNORE: a marker k_pair made with the line in sight has passed its counts there; the frequency bound is checked here) */
         uint32_t keep = 1u;
         if (a.ig_need != 0u && (e.w & 2u) != 0u) {
            const uint32_t holds = order_marker_holds(a, a.seg_base + e.x) ? 1u : 0u;
            if (!holds) { keep = e.w & 1u; ent[k].w = e.w | 4u; }
         }
         a.hit_start[k] = keep ? e.x : 0xFFFFFFFFu; hit_col[k] = keep ? col : 0u;
         continue;
      }
      const uint64_t hp = a.seg_base + e.x;               /* a byte of the line (inside the segment); never '\n' */
      if (hp >= a.nbytes || hp < segb || hp >= segb + a.seg_len) {       /* cannot be: an entry the scan kernel never wrote -- fail loudly, touch nothing */
         atomicOr(&c->overflow, 64u);
         a.hit_start[k] = 0xFFFFFFFFu;
         hit_col[k] = 0u;
         continue;
      }
      const uint64_t q = order_line_start(a.text, hp);
      /* k_pair under SQ_IGNORE named this line unseen (it began before the candidate's tile): without a skipped byte in it, or with fewer than
         m - tau other characters, there is nothing a skipped byte could hide -- the marker goes (were it a repeat of its line it would not be
         here).  (Evaluated BEFORE the branches below: as the condition of an `else if` between them the loop of order_marker_holds made the
         gfx950 build store the start in hit_col and nothing in hit_start -- gone with a printf in the branch; ROCm 7.2.0.) */
      uint32_t keep = 1u, holds = 1u;
      if (a.ig_thr != 0u && (e.w & 2u) != 0u && q >= a.seg_base) holds = order_marker_holds(a, q) ? 1u : 0u;
      if (!holds) {
         keep = e.w & 1u;                                 /* (the marker stood on a candidate of the walk: that stays, as under SQ_FAIL) */
         ent[k].w = e.w | 4u;                             /* (bit 2: a marker no more) */
      }
      uint32_t out_start, out_col;
      if (q < a.seg_base) { atomicOr(&c->overflow, 8u); out_start = 0xFFFFFFFFu; out_col = 0u; }
      else if (!keep) { out_start = 0xFFFFFFFFu; out_col = 0u; }
      else { out_start = (uint32_t)(q - a.seg_base); out_col = (uint32_t)(hp - q); }
      a.hit_start[k] = out_start;
      hit_col[k] = out_col;
   }
}

#endif
