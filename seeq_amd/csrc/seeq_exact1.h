/*
 * seeq_exact1.h -- k_exact1<MODE, W>: the exact pass for patterns of <= 30 (W = 1) / 31..62 (W = 2) positions.
 *
 * Same job as k_exact<1,*> (seeq_device.hip): over the HIT lines only, apply the reference's
 * acceptance rules (libseeq.c:277-331) and recover match starts (libseeq.c:289-316); but with the
 * top-aligned 12-op Myers step and the byte-indexed EQ tables of the scan kernels instead of the
 * generic multi-word column, and with the text staged per lane through LDS so that the per-character
 * loop is a small rolled loop.  One hit line per lane.
 */
#ifndef SEEQ_EXACT1_H_
#define SEEQ_EXACT1_H_

#define EXACT1_ROW 80          /* LDS bytes per lane: 64 text bytes + pad, 16-byte aligned */
typedef __attribute__((address_space(3))) uint8_t exact1_lds_u8;
typedef __attribute__((address_space(3))) const uint8_t exact1_lds_cu8;

/* Reverse start recovery, reference libseeq.c:289-316, on the reversed-pattern EQ table.  `text + off` is
 * the line, i the column the match ends before.  With `row` (the lane's 64-byte LDS row, free at that
 * point) the 64 bytes before the match end are fetched with four loads up front instead of one dependent
 * byte load per step; columns further back (non-base bytes skipped under SQ_IGNORE) come from memory. */
template <int W>
__device__ __forceinline__ uint32_t exact1_reverse(const uint8_t *text, uint64_t off, uint64_t nbytes, uint32_t i, uint32_t streak,
                                                   uint32_t eqr_base, uint32_t m, uint32_t tau1, uint8_t *row)
{
   const uint8_t *line = text + off;
   uint32_t lim = 0;                                      /* the row holds the `lim` bytes before the match end */
   if (row) {
      const uint64_t end = off + i;
      const uint64_t base = end >= 64 ? end - 64 : 0;
      lim = (uint32_t)(end - base);
#pragma unroll
      for (int q = 0; q < 4; q++)
         *reinterpret_cast<fused_v4u *>(row + 16 * q) = direct_load16(text, base + 16 * q, nbytes);
   }
   fused_state_t<W> st;
   st.init(m);
   uint32_t j = 0, d = tau1, last_d, ignores = 0;
   do {
      ++j;
      uint32_t b;
      /* (an LDS load and a global load: through one selected generic pointer it would be a flat load, per step) */
      if (j <= lim) b = (uint32_t)*(exact1_lds_cu8 *)(uintptr_t)((uint32_t)(uintptr_t)(exact1_lds_u8 *)row + lim - j);
      else b = (uint32_t)line[i - j];
      const fused_eq_t<W> ev = fused_eq_load<W>(eqr_base + (b << (W == 1 ? 2 : 3)));
      const uint32_t e = ev.w0;
      last_d = d;
      if ((e & FUSED_FLAGS) == 0) {
         ignores = 0;
         st.step(ev);
         d = st.score < tau1 ? st.score : tau1;
      } else {
         ignores++;                                   /* any non-base byte: skipped, counted (libseeq.c:308-311) */
      }
   } while (d > streak && j < i);
   const int jj = (int)(last_d < d ? j - 1 : j) - (int)ignores;     /* libseeq.c:315 */
   return (uint32_t)((int)i - jj);
}

/* Only alphabet bytes (A C G T N, either case, newline) in text[lo, hi)?  The window walk may skip such a stretch of
 * a line: no byte in it ends the line (SQ_FAIL), and k_stream's verdict "no hit ends in it" is exact.  Whole tiles are
 * answered by the exclusive prefix of k_stream's per-tile flags, the partial tiles at both ends by reading the bytes. */
__device__ __forceinline__ bool exact1_bytes_clean(const uint8_t *text, uint64_t lo, uint64_t hi)
{
   uint32_t bad = 0;
   uint64_t p = lo;
   for (; p + 128 <= hi && !bad; p += 128) {              /* eight independent loads per step: latency, not bytes, is the cost */
      fused_v4u v[8];
#pragma unroll
      for (int q = 0; q < 8; q++) v[q] = *reinterpret_cast<const fused_v4u_unaligned *>(text + p + 16 * q);
#pragma unroll
      for (int q = 0; q < 8; q++) bad |= fused_bad4(v[q].x) | fused_bad4(v[q].y) | fused_bad4(v[q].z) | fused_bad4(v[q].w);
   }
   for (; p + 16 <= hi && !bad; p += 16) {
      const fused_v4u v = *reinterpret_cast<const fused_v4u_unaligned *>(text + p);
      bad = fused_bad4(v.x) | fused_bad4(v.y) | fused_bad4(v.z) | fused_bad4(v.w);
   }
   for (; p < hi && !bad; p++) bad = fused_bad4(0x0A0A0A00u | text[p]);
   return bad == 0;
}

/* text[x, y) inside ONE tile (base tb): whole 128-byte chunks by the tile's chunk mask, the partial chunks by bytes */
__device__ __forceinline__ bool exact1_clean_in_tile(const ScanArgs &a, uint64_t tb, uint64_t dmask, uint64_t x, uint64_t y)
{
   if (y <= x) return true;
   const uint64_t c_lo = (x - tb + 127) >> 7, c_hi = (y - tb) >> 7;          /* whole chunks [c_lo, c_hi) */
   if (c_lo >= c_hi) return exact1_bytes_clean(a.text, x, y);
   const uint64_t below_hi = c_hi >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << c_hi) - 1);
   const uint64_t below_lo = ((uint64_t)1 << c_lo) - 1;                      /* c_lo < c_hi <= 64 */
   if (dmask & below_hi & ~below_lo) return false;
   return exact1_bytes_clean(a.text, x, tb + (c_lo << 7)) && exact1_bytes_clean(a.text, tb + (c_hi << 7), y);
}

__device__ __forceinline__ bool exact1_clean(const ScanArgs &a, uint64_t lo, uint64_t hi)
{
   if (hi <= lo) return true;
   if (!a.cnt->dirty) return true;                        /* nothing outside the alphabet in the text scanned so far */
   const uint64_t segb = a.seg_base + a.pos_bias;         /* the segment proper */
   if (!a.tile_dirty || lo < segb) return false;
   const uint64_t TB = a.stream_tile_bytes;
   const uint64_t t_lo = (lo - segb + TB - 1) / TB, t_hi = (hi - segb) / TB;          /* whole tiles [t_lo, t_hi) */
   const uint64_t t_first = (lo - segb) / TB;
   if (t_lo >= t_hi) {                                    /* no whole tile: one or two partial tiles */
      if (t_first == t_hi || t_hi >= a.stream_ntiles)
         return exact1_clean_in_tile(a, segb + t_first * TB, a.tile_dmask[t_first], lo, hi);
      return exact1_clean_in_tile(a, segb + t_first * TB, a.tile_dmask[t_first], lo, segb + t_hi * TB) &&
             exact1_clean_in_tile(a, segb + t_hi * TB, a.tile_dmask[t_hi], segb + t_hi * TB, hi);
   }
   const uint32_t d_hi = t_hi < a.stream_ntiles ? a.tile_dirty[t_hi] : a.cnt->seg_dirty_tiles;
   if (d_hi != a.tile_dirty[t_lo]) return false;
   if (t_first < t_lo && !exact1_clean_in_tile(a, segb + t_first * TB, a.tile_dmask[t_first], lo, segb + t_lo * TB)) return false;
   if (t_hi < a.stream_ntiles && hi > segb + t_hi * TB &&
       !exact1_clean_in_tile(a, segb + t_hi * TB, a.tile_dmask[t_hi], segb + t_hi * TB, hi)) return false;
   return true;
}

template <int W> __device__ __forceinline__ void exact1_take(fused_state_t<W> &st, const fused_state_t<W> &s2, bool take);
template <> __device__ __forceinline__ void exact1_take<1>(fused_state_t<1> &st, const fused_state_t<1> &s2, bool take)
{
   st.pv = take ? s2.pv : st.pv; st.mv = take ? s2.mv : st.mv; st.score = take ? s2.score : st.score;
}
template <> __device__ __forceinline__ void exact1_take<2>(fused_state_t<2> &st, const fused_state_t<2> &s2, bool take)
{
   st.pv0 = take ? s2.pv0 : st.pv0; st.pv1 = take ? s2.pv1 : st.pv1;
   st.mv0 = take ? s2.mv0 : st.mv0; st.mv1 = take ? s2.mv1 : st.mv1; st.score = take ? s2.score : st.score;
}

/* hit_col (k_stream only, else NULL): column of the FIRST hit end of each line.  On clean text no earlier
 * column has D[m] <= tau, so the scan may start 32 >= m + tau - 1 columns before it with a fresh column:
 * from the hit column on the saturated scores -- all the acceptance rules look at -- are the same. */
/* OPT: the match option (SQ_FIRST / SQ_BEST / SQ_ALL; SQ_COUNT behaves as SQ_FIRST) as a compile-time constant for
 * the EMIT kernels -- the per-character body then has no option branches; -1 = read it from a.options (COUNT). */
/* cache (16 B per hit line, the scan kernels' slice buffer, free by now; NULL = off): when records are wanted, the
 * COUNT pass leaves {end, dist} of the FIRST emission of every line there (for SQ_BEST the best one, and COUNT then
 * scans the whole line instead of stopping at the first hit).  Further emissions of a line (SQ_ALL: 7 % of the hit
 * lines of configs[4] have a second one) go to overflow lists {hit-list entry, index in the line, end, dist} in the
 * free entries above the per-line ones: ONE LIST PER WAVE of the grid (entry 0 of a wave's region = its count),
 * filled through a counter in LDS -- no global atomics (a single global counter, even with one atomic per wave,
 * cost 85-165 us per launch: same-address atomics serialise at ~10 ns each).  The EMIT pass (same grid) recovers
 * the start of every line's first record and then, wave by wave, of its own overflow list, one emission per lane:
 * no line is scanned twice and no lane scans a line alone while its workgroup waits (a single lane takes 0.3 us per
 * character: with a cache of two emissions per line and re-scans for the rest, that tail was most of the pass).
 * Only when a region is too small (Counters.seg_novf set) does EMIT scan the lines with more than one record again. */
/* WALK: compile the window walk in (long-line inputs); without it the per-character loop carries no walk state */
template <int MODE, int W, int OPT, bool WALK>
__device__ __forceinline__ void exact1_body(const ScanArgs &a, const uint32_t *eq2, const uint32_t *hit_col, uint4 *cache)
{
   __shared__ __align__(8) uint32_t s_eqf[256 * W];
   __shared__ __align__(8) uint32_t s_eqr[256 * W];
   __shared__ __align__(16) uint8_t s_blk[256 * EXACT1_ROW];
   __shared__ uint32_t s_novf[4];                          /* COUNT: entries in each wave's overflow list */
   if (threadIdx.x < 4) s_novf[threadIdx.x] = 0;
   for (int i = threadIdx.x; i < 256 * W; i += 256) { s_eqf[i] = eq2[i]; s_eqr[i] = eq2[256 * W + i]; }
   __syncthreads();
   const uint32_t eqf_base = (uint32_t)(uintptr_t)(fused_lds_cu32 *)s_eqf;
   const uint32_t eqr_base = (uint32_t)(uintptr_t)(fused_lds_cu32 *)s_eqr;
   const Counters *c = a.cnt;
   const uint32_t nhl = c->seg_nhitlines;
   const int match_opt = OPT >= 0 ? OPT : (a.options & 3);
   if (MODE == SQ_MODE_EMIT && (c->overflow & 4u)) return;
   const uint32_t m = (uint32_t)a.m, tau1 = (uint32_t)a.tau + 1;
   const bool count_any = a.want != SEEQDEV_WANT_COUNTMATCH && !(a.want == SEEQDEV_WANT_RECORDS && match_opt == SQ_ALL);
   const bool by_nh = a.use_nh != 0;                       /* record slots come from the scanned per-line counts */
   const bool trusted = a.use_nh == 3 && !a.filter && !c->dirty;   /* k_stream, complete automaton, clean text: exact verdicts */
   const bool caching = MODE == SQ_MODE_COUNT && cache != nullptr && a.want == SEEQDEV_WANT_RECORDS;
   const bool count_best = caching && match_opt == SQ_BEST;
   const bool cache_ok = MODE == SQ_MODE_EMIT && cache != nullptr && by_nh && !(trusted && count_any);
   /* overflow lists: the free entries above the per-line ones, split evenly over the waves of the grid (COUNT and EMIT
      are launched with the same grid) */
   const uint32_t ovf_r = (a.cap_hitlines > nhl ? a.cap_hitlines - nhl : 0u) / (gridDim.x * 4u);   /* entries per wave, the count included */
   const uint32_t wave_id = threadIdx.x >> 6;
   uint4 *ovf = cache ? cache + nhl + (size_t)(blockIdx.x * 4u + wave_id) * ovf_r : nullptr;
   const bool ovf_lost = MODE == SQ_MODE_EMIT && c->seg_novf != 0;               /* (EMIT: the lists are complete or they are not used) */
   const bool walk = WALK && a.use_nh == 3 && hit_col != nullptr && a.stream_ch != 0;     /* window walk (below); kernel-uniform */
   const uint32_t wback = a.skip_back > 32u ? a.skip_back : 32u;      /* columns a fresh column needs before a candidate (the table walks warm up over <= 32; the Myers mode over m + tau - 1) */
   /* no byte is skipped under these options: a flagged byte ends the line, so the column of a lane needs no protecting
      from it (nothing reads the column of a finished line) -- the per-character bodies below drop the predicated copy */
   const bool noskip = (a.options & (SQ_IGNORE | SQ_STREAM)) == 0;
   uint8_t *row = s_blk + threadIdx.x * EXACT1_ROW;
   const uint32_t stride = gridDim.x * 256;
   /* wave-uniform trip count so that every lane of a wave takes part in the wave-level votes */
   const uint32_t kmax = (nhl + stride - 1) / stride * stride;
   for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < kmax; k += stride) {
      bool done = k >= nhl;
      /* hit_idx (several patterns, one walk -- seeq_multi.h): entry k of this pattern's list is entry hit_idx[k] of the shared per-line arrays */
      const uint32_t kk = (a.hit_idx && !done) ? a.hit_idx[k] : k;
      const uint32_t hs = done ? 0u : a.hit_start[kk];
      if (hs == 0xFFFFFFFFu) done = true;                  /* k_stream: repeat of the previous entry's line */
      const uint64_t off = done ? a.seg_base : a.seg_base + hs;
      if (MODE == SQ_MODE_COUNT && trusted && count_any) {
         if (k < nhl) a.nh[k] = done ? 0u : 1u;            /* (wave-uniform branch) */
         continue;
      }
      if (a.use_nh == 3 && (a.options & SEEQDEV_FASTA) && !done && a.text[off] == '>') done = true;   /* k_stream candidate inside a FASTA header */
      fused_state_t<W> st;
      st.init(m);
      uint32_t streak = tau1, nhits = 0, best_d = tau1, best_end = 0, pos = 0;
      /* Window walk (k_stream, clean text, lines that end inside the segment): only the candidate chunks of the
         line are scanned -- the first entry and the repeats of the line that follow it in the hit list, each from
         32 columns before the candidate to the end of its chunk, on while the last 32 columns still hold a
         sub-threshold score (such a tail is what hides a hit from the next lane's walk), then a jump to the next
         candidate with a fresh column.  A chromosome-long line costs its hits, not its length. */
      bool win = false;
      uint32_t wend = 0, knext = k + 1;                    /* end of the current window (column), next hit-list entry */
      /* k_pair (window_ok): the candidates of this line are this entry and the repeats behind it, and no chain dropped any.
         Every occurrence of the line starts within m + tau before one of them and ends within m + tau behind it
         (seeq_pair.h): beyond `stop_at` the capped score is tau + 1 as at a terminator, and the scan ends there. */
      uint32_t stop_at = 0xFFFFFFFFu;
      int32_t lastsub = -0x40000000;                       /* column of the last score <= tau seen */
      if (hit_col && (trusted || walk || a.filter) && !done) {
         const uint32_t col = hit_col[kk];
         /* nothing ends the line before the first candidate (clean text), so the scan may start just before it: no
            occurrence ends before `col` (a filter: every occurrence contains a part that ends at or after the first
            candidate), and a column started skip_back >= m + tau - 1 bytes earlier has the line's own scores from there */
         /* (not trusted: no byte outside the alphabet up to the candidate itself -- a skipped byte inside the warm-up
            columns, SQ_IGNORE, would leave the fresh column short of real characters) */
         /* k_pair under SQ_IGNORE (round 5, seeq_pair.h IG): a line that holds a skipped byte and enough other characters for an occurrence has a
            MARKER among its entries -- this one or a repeat of the line behind it -- and is scanned from its first byte to its end; a line without
            one holds no skipped byte (or too few characters for any occurrence): its windows are what they are under SQ_FAIL */
         bool ig_whole = false;
         if (a.ig_ent) {
            ig_whole = (a.ig_ent[kk].w & 6u) == 2u;
            for (uint32_t j = k + 1; !ig_whole && j < nhl && a.hit_start[j] == 0xFFFFFFFFu && a.hit_line[j] == a.hit_line[kk]; j++) ig_whole = (a.ig_ent[j].w & 6u) == 2u;
         }
         if (col > a.skip_back && !ig_whole && (trusted || a.ig_ent || exact1_clean(a, off, off + col))) pos = col - a.skip_back;
         /* (COUNT only: nh[] holds offsets by the time EMIT runs; an EMIT that scans again scans on -- and finds nothing more) */
         if (MODE == SQ_MODE_COUNT && a.window_ok && !ig_whole) {
            /* the line's further candidates are the repeats behind this entry (a third of the hit lines has one: an occurrence
               near the end of a chain is seen by the next chain's warm-up, which then reports its own first pair): the scan
               runs from before the first candidate to behind the last one */
            uint32_t lastcol = a.hit_last ? a.hit_last[kk] : col, unbounded = 0u;      /* (hit_last: packed read batches -- the candidates of a read come as one entry) */
            if (a.hit_idx) {                               /* the line's window is the union's: it ends maxspan (= skip_back) + 2 behind the last candidate; no last candidate: no end */
               unbounded = lastcol == 0xFFFFFFFFu ? 2u : 0u;
               lastcol += a.skip_back - (m + tau1 - 1u);
            }
            else for (uint32_t j = k + 1; j < nhl && a.hit_start[j] == 0xFFFFFFFFu && (!a.ig_ent || a.hit_line[j] == a.hit_line[kk]); j++) lastcol = hit_col[j] - hs;      /* (ig_ent: a marker k_bounds2 dropped looks like a repeat -- of ANOTHER line) */
            if (!unbounded) stop_at = lastcol + m + tau1 + 1u;
         }
         /* the line ends in this segment: a newline at or after its start, or the buffer ends with the segment */
         const uint32_t lastnl = c->seg_last_nl;
         const bool last_seg = a.seg_base + a.pos_bias + a.seg_len >= a.nbytes;
         win = walk && (last_seg || (lastnl != 0 && (int64_t)hs - (int64_t)a.pos_bias < (int64_t)lastnl));
         /* (ll_restart, round 5: the filter's restart table flags every part occurrence, so every occurrence of the pattern holds a candidate
            and ends within m + tau behind it -- the window ends there, not at the end of the candidate's chunk) */
         wend = a.ll_restart ? col + m + tau1 + 1u : (((hs + col) | (a.stream_ch - 1u)) + 1u) - hs + a.walk_ext;
      }
      bool latch = false;
      seeqdev_hit_t *out = nullptr;
      uint32_t out_cap = 0, line_no = 0;
      uint32_t ce0 = 0, ce1 = 0, ncached = 0;                         /* {end, dist} of the first emission -- COUNT: what goes to the cache; EMIT: what came from it */
      uint32_t cstart = 0, chas = 0;                                  /* EMIT: the cached record's start, when COUNT recovered it */
      bool from_cache = false;
      if (MODE == SQ_MODE_EMIT && !done) {
         line_no = a.hit_line[kk];
         if (match_opt == SQ_ALL) { out = a.records + c->records + nh_at(a, k); out_cap = 0xFFFFFFFFu; }
         else { out = a.records + c->records + (by_nh ? nh_at(a, k) : k); out_cap = 1; }
         if (cache_ok) {
            ncached = (k + 1 < nhl ? nh_at(a, k + 1) : c->seg_nrec) - nh_at(a, k);
            if (ncached <= 1 || !ovf_lost) {               /* the first record from the cache, the others from the overflow list */
               if (ncached) { const uint4 ce = cache[k]; ce0 = ce.x; ce1 = ce.y; cstart = ce.z; chas = ce.w; }
               from_cache = true;
               done = true;
            }
         }
      }
      while (__any(!done)) {
         /* next 64 bytes of my line -> my LDS row (bytes beyond the buffer read as NUL) */
         if (!done) {
#pragma unroll
            for (int q = 0; q < 4; q++)
               *reinterpret_cast<fused_v4u *>(row + 16 * q) = direct_load16(a.text, off + pos + 16 * q, a.nbytes);
         }
         uint32_t adv = 64;                                /* columns of this block the wave steps */
#pragma unroll 1
         for (uint32_t t4 = 0; t4 < 64; t4 += 4) {
            if (!__any(!done)) break;
            /* window walk: once EVERY lane still at work stands behind the end of its window the block ends here (round 5: the restart
               table's windows are all 2 (m + tau) + 2 columns from a start 64 apart -- they end together, mid-block: 128 columns stepped for 86
               before this); what a lane does next -- the next candidate, a longer window, the end -- is decided below as at a block's end */
            if (walk && t4 != 0u && !__any(!done && !(win && pos + t4 >= wend))) { adv = t4; break; }
            /* four characters: the EQ lookups go out together, the column steps are predicated (no branches) */
            const uint32_t w4 = *reinterpret_cast<const uint32_t *>(row + t4);
            fused_eq_t<W> ev[4];
#pragma unroll
            for (int cc = 0; cc < 4; cc++)
               ev[cc] = fused_eq_load<W>(eqf_base + (((w4 >> (8 * cc)) & 0xFFu) << (W == 1 ? 2 : 3)));
            if ((MODE == SQ_MODE_EMIT && match_opt == SQ_BEST) || (MODE == SQ_MODE_COUNT && count_best)) {
               /* SQ_BEST: smallest distance, first position where it is left (score rises) or repeated as 0.
                  No latch is needed: an emission the latch suppresses never beats best_d -- after a rise from s,
                  best_d <= s already (by induction over consecutive rises), and after a zero-distance emission
                  best_d = 0.  A finished lane is fed the line-end flag: that step is idempotent. */
               if (noskip) {
                  /* (flags as integers and bitwise logic: with `||` the compiler builds exec-masked regions per
                     character just to let finished lanes skip the wait for their EQ word) */
                  uint32_t dn = done ? 1u : 0u;
#pragma unroll
                  for (int cc = 0; cc < 4; cc++) {
                     dn |= (ev[cc].w0 & FUSED_FLAG_TERM) | (pos + t4 + cc >= stop_at ? 1u : 0u);
                     st.step(ev[cc]);
                     const uint32_t sc = st.score < tau1 ? st.score : tau1;
                     const uint32_t cur = dn ? tau1 : sc;
                     const bool upd = (streak < best_d) & ((streak < cur) | (streak == 0));
                     best_d = upd ? streak : best_d;
                     best_end = upd ? pos + t4 + cc : best_end;
                     if (walk) lastsub = cur < tau1 ? (int32_t)(pos + t4 + cc) : lastsub;
                     streak = cur;
                  }
                  done = dn != 0;
                  continue;
               }
#pragma unroll
               for (int cc = 0; cc < 4; cc++) {
                  const uint32_t e = done ? FUSED_FLAG_TERM : ev[cc].w0;
                  const bool term = (e & FUSED_FLAG_TERM) != 0, skip = (e & FUSED_FLAG_SKIP) != 0;
                  fused_state_t<W> s2 = st;
                  s2.step(ev[cc]);
                  exact1_take<W>(st, s2, (e & FUSED_FLAGS) == 0);
                  const uint32_t sc = s2.score < tau1 ? s2.score : tau1;
                  const uint32_t cur = term ? tau1 : sc;
                  const bool upd = !skip && streak < best_d && (streak < cur || streak == 0);
                  best_d = upd ? streak : best_d;
                  best_end = upd ? pos + t4 + cc : best_end;
                  if (walk) lastsub = cur < tau1 && !skip ? (int32_t)(pos + t4 + cc) : lastsub;
                  streak = skip ? streak : cur;
                  done = done || term;
               }
               continue;
            }
            /* SQ_FIRST / SQ_ALL / counting: the acceptance rules of libseeq.c:277-331 per character.  NS (no byte is
               skipped, see `noskip`): the column is stepped in place and a finished lane's bookkeeping runs on unprotected --
               `act` keeps it from emitting, nothing else of it is read again. */
#define EXACT1_CHARS4(NS) \
            _Pragma("unroll") \
            for (int cc = 0; cc < 4; cc++) { \
               const uint32_t e = ev[cc].w0; \
               const bool term = (e & FUSED_FLAG_TERM) != 0; \
               const bool act = NS ? !done : (!done & !(e & FUSED_FLAG_SKIP)); \
               uint32_t score; \
               if (NS) { st.step(ev[cc]); score = st.score; } \
               else { fused_state_t<W> s2 = st; s2.step(ev[cc]); exact1_take<W>(st, s2, act & !term); score = s2.score; } \
               const uint32_t sc = score < tau1 ? score : tau1; \
               const uint32_t cur = term ? tau1 : sc; \
               const bool stop = streak < cur, zero = streak == 0; \
               const uint32_t p = pos + t4 + cc; \
               if (walk) lastsub = ((NS || act) & (cur < tau1)) ? (int32_t)p : lastsub; \
               bool end = term; \
               const bool emit = act & (stop ? !latch : zero); \
               latch = (NS || act) ? (stop ? true : zero) : latch; \
               if (MODE == SQ_MODE_COUNT) { \
                  if (caching) { \
                     if (__any(emit)) {                       /* (a few times per line) */ \
                        const bool f0 = emit & (nhits == 0); \
                        ce0 = f0 ? p : ce0; ce1 = f0 ? streak : ce1; \
                        if (emit && nhits >= 1) {             /* second and later: to my wave's overflow list */ \
                           const uint32_t idx = atomicAdd(&s_novf[wave_id], 1u) + 1u; \
                           if (idx < ovf_r) ovf[idx] = make_uint4(k, nhits, p, streak); \
                        } \
                     } \
                  } \
                  nhits += emit ? 1u : 0u; \
                  end = end | (count_any & emit);        /* presence is enough: FIRST/BEST/COUNTLINES */ \
               } else { \
                  /* EMIT: only {end, dist} into the record slot now; the starts are recovered after the forward scan \
                     (below), when the lane's LDS row is free and the lanes of the wave do it together */ \
                  if (emit && nhits < out_cap) { out[nhits].end = p; out[nhits].dist = streak; } \
                  nhits += emit ? 1u : 0u; \
                  end = end | (emit & (match_opt != SQ_ALL));           /* SQ_FIRST / SQ_COUNT: libseeq.c:330 */ \
               } \
               streak = (NS || act) ? cur : streak; \
               done = done | ((NS || act) & end); \
            }
            if (noskip) {
               /* The same rules in integer arithmetic (flags as 0 / 1 words, compares as sign bits: scores are <= tau + 1
                  < 64): as bools the compiler keeps materialising them between mask registers and 0 / 1 words, ~15 of
                  the ~60 instructions per character of the two-word COUNT. */
               uint32_t dn = done ? 1u : 0u, latch_u = latch ? 1u : 0u;
               const uint32_t any_u = (MODE == SQ_MODE_COUNT ? count_any : match_opt != SQ_ALL) ? 1u : 0u;
#pragma unroll
               for (int cc = 0; cc < 4; cc++) {
                  const uint32_t term_u = (ev[cc].w0 & FUSED_FLAG_TERM) | (pos + t4 + cc >= stop_at ? 1u : 0u);      /* (= 1) */
                  st.step(ev[cc]);
                  const uint32_t m1 = st.score | (0u - term_u);                   /* the terminator's step: tau + 1 */
                  const uint32_t cur = m1 < tau1 ? m1 : tau1;
                  const uint32_t stop_u = (streak - cur) >> 31, zero_u = (streak - 1u) >> 31;
                  const uint32_t emit_u = (dn ^ 1u) & ((stop_u & (latch_u ^ 1u)) | ((stop_u ^ 1u) & zero_u));
                  latch_u = stop_u | zero_u;
                  const uint32_t p = pos + t4 + cc;
                  if (walk) lastsub = cur < tau1 ? (int32_t)p : lastsub;
                  if (MODE == SQ_MODE_COUNT) {
                     if (caching) {
                        if (__any(emit_u != 0)) {                /* (a few times per line) */
                           const bool f0 = emit_u != 0 && nhits == 0;
                           ce0 = f0 ? p : ce0; ce1 = f0 ? streak : ce1;
                           if (emit_u != 0 && nhits >= 1) {      /* second and later: to my wave's overflow list */
                              const uint32_t idx = atomicAdd(&s_novf[wave_id], 1u) + 1u;
                              if (idx < ovf_r) ovf[idx] = make_uint4(k, nhits, p, streak);
                           }
                        }
                     }
                  } else {
                     if (emit_u != 0 && nhits < out_cap) { out[nhits].end = p; out[nhits].dist = streak; }
                  }
                  nhits += emit_u;
                  dn |= term_u | (any_u & emit_u);               /* presence is enough / SQ_FIRST: libseeq.c:330 */
                  streak = cur;
               }
               done = dn != 0; latch = latch_u != 0;
            } else { EXACT1_CHARS4(false) }
#undef EXACT1_CHARS4
         }
         pos += adv;
         if (walk && win && !done) {
            /* A score <= tau in the last 32 columns of a chunk (or inside the chunk the walk stands in) may have put
               that chunk's lane into the accepting state before it could report: the chunk has to be scanned whole. */
            const int32_t b = (int32_t)(((hs + pos) & ~(a.stream_ch - 1u)) - hs);      /* start of the chunk holding `pos` */
            /* (not behind the restart table: no chain is ever absorbed there -- every part occurrence is a candidate of its own, or the first byte
               of the chain that met it inside its warm-up window is -- and the leaders' spacing counts on windows that end where they say) */
            if (!a.ll_restart && lastsub >= b - (int32_t)wback && b + (int32_t)a.stream_ch > (int32_t)wend) wend = (uint32_t)(b + (int32_t)a.stream_ch);
            if (pos >= wend) {
               /* the window is done and the columns behind are clean: nothing can hide before the next candidate */
               for (;;) {
                  if (!(knext < nhl && a.hit_start[knext] == 0xFFFFFFFFu)) {                       /* no candidate left */
                     /* (seeq_stream.h, leaders: the entry behind my group learns where my walk ended -- k_lead_check holds its
                        fresh start against it) */
                     if (MODE == SQ_MODE_COUNT && a.walk_end && knext < nhl) a.walk_end[knext] = hs + pos;
                     done = true;
                     break;
                  }
                  const uint32_t cpos = hit_col[knext++];                 /* position of the repeat's first hit */
                  const uint32_t ccol = cpos - hs, cend = a.ll_restart ? ccol + m + tau1 + 1u : ((cpos | (a.stream_ch - 1u)) + 1u) - hs + a.walk_ext;
                  /* FASTA input: a header line carries the rank of the line before it (its newline is not counted), so a "repeat" may lie in the
                     header BEHIND this line's end.  The stretch jumped over must then hold no newline: a header's first byte, '>', is outside
                     the alphabet and fails the test -- unless the jump lands exactly on it, the newline being the stretch's last byte: the
                     landing byte is looked at too (found by profiles/ignore_fuzz.py: a hit inside a header reported for the line before). */
                  const uint64_t land = (a.options & SEEQDEV_FASTA) ? 1u : 0u;
                  if (ccol > pos + wback && exact1_clean(a, off + pos, off + ccol - wback + land)) {   /* jump: fresh column `wback` columns before it */
                     pos = ccol - wback; wend = cend;
                     st.init(m); streak = tau1; latch = false; lastsub = -0x40000000;
                     break;
                  }
                  if (cend > wend) wend = cend;                           /* adjacent or behind: the walk just goes on */
                  if (wend > pos) break;
               }
            }
         }
      }
      if (k < nhl) {
         if (MODE == SQ_MODE_COUNT) {
            if (count_best) { nhits = best_d < tau1 ? 1u : 0u; ce0 = best_end; ce1 = best_d; }
            a.nh[k] = nhits;
            /* one record per line at most (SQ_BEST / SQ_FIRST): its start is recovered here, where the line has just been
               scanned (its bytes are in L1 / L2 and the lane's LDS row is free), and EMIT only copies {start, end, dist} */
            uint32_t ce2 = 0, ce3 = 0;
            if (caching && count_any && nhits) { ce2 = exact1_reverse<W>(a.text, off, a.nbytes, ce0, ce1, eqr_base, m, tau1, row); ce3 = 1u; }
            if (caching) cache[k] = make_uint4(ce0, ce1, ce2, ce3);
         } else if (match_opt == SQ_BEST && !from_cache) {
            if (best_d < tau1) {
               seeqdev_hit_t h;
               h.line = line_no;
               h.start = exact1_reverse<W>(a.text, off, a.nbytes, best_end, best_d, eqr_base, m, tau1, row);
               h.end = best_end;
               h.dist = best_d;
               out[0] = h;
               a.rec_off[out - a.records] = rec_off_of(a, off, line_no);
            }
         } else {
            /* the first record whole from the COUNT pass's cache (the others: the overflow list, below), or the
               emissions of the scan above -- then recover the starts */
            const uint32_t n = from_cache ? (ncached ? 1u : 0u) : (nhits < out_cap ? nhits : out_cap);
            for (uint32_t i = 0; i < n; i++) {
               seeqdev_hit_t h;
               h.line = line_no;
               if (from_cache) { h.end = ce0; h.dist = ce1; }
               else { h.end = out[i].end; h.dist = out[i].dist; }      /* (written by this lane, above) */
               h.start = (from_cache && chas) ? cstart : exact1_reverse<W>(a.text, off, a.nbytes, h.end, h.dist, eqr_base, m, tau1, row);
               out[i] = h;
               a.rec_off[(out - a.records) + i] = rec_off_of(a, off, line_no);     /* byte offset of the record's line (seeqdevScanCopyOffsets) */
            }
         }
      }
   }
   /* COUNT: publish the length of my wave's overflow list */
   if (MODE == SQ_MODE_COUNT && caching && ovf_r && (threadIdx.x & 63u) == 0) {
      const uint32_t n = s_novf[wave_id];                  /* (this wave's own LDS atomics: in program order) */
      ovf[0] = make_uint4(n, 0u, 0u, 0u);
      if (n + 1u > ovf_r) a.cnt->seg_novf = 1u;            /* it did not fit: EMIT scans again */
   }
   if (MODE == SQ_MODE_COUNT && caching && !ovf_r && (threadIdx.x & 63u) == 0 && s_novf[wave_id]) a.cnt->seg_novf = 1u;
   /* EMIT: the emissions beyond the first of their lines, one per lane */
   if (MODE == SQ_MODE_EMIT && cache_ok && !ovf_lost && ovf_r) {
      const uint32_t novf = ovf[0].x;
      for (uint32_t e = 1u + (threadIdx.x & 63u); e <= novf; e += 64u) {
         const uint4 o = ovf[e];                                    /* {hit-list entry, index in the line, end, dist} */
         const uint32_t ox = a.hit_idx ? a.hit_idx[o.x] : o.x;
         const uint64_t off = a.seg_base + a.hit_start[ox];
         const uint64_t slot = c->records + nh_at(a, o.x) + o.y;
         seeqdev_hit_t h;
         h.line = a.hit_line[ox];
         h.start = exact1_reverse<W>(a.text, off, a.nbytes, o.z, o.w, eqr_base, m, tau1, row);
         h.end = o.z;
         h.dist = o.w;
         a.records[slot] = h;
         a.rec_off[slot] = rec_off_of(a, off, h.line);
      }
   }
}

template <int MODE, int W, int OPT, bool WALK>
__global__ __launch_bounds__(256, 6) void k_exact1(ScanArgs a, const uint32_t *eq2, const uint32_t *hit_col, uint4 *cache)
{
   exact1_body<MODE, W, OPT, WALK>(a, eq2, hit_col, cache);
}

/* Several patterns in ONE launch (seeq_multi.h): blockIdx.y selects the pattern -- its arguments (its own index list, counters,
   record region, m, tau), its EQ tables, its COUNT -> EMIT cache -- from an array in HBM (scalar loads: the index is uniform). */
struct MultiExact {
   ScanArgs        a;
   const uint32_t *eq;
   const uint32_t *hcol;
   uint4          *cache;
   uint32_t       *scan_ws;      /* this pattern's block sums of the scan over its per-pair counts */
   uint32_t        nb;           /* blocks of that scan */
   int             seg_end_flags;
};

template <int MODE, int W, int OPT>
__global__ __launch_bounds__(256, 6) void k_exact1m(const MultiExact *mx)
{
   const MultiExact &m = mx[blockIdx.y];
   exact1_body<MODE, W, OPT, false>(m.a, m.eq, m.hcol, m.cache);
}

#endif
