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

/* Reverse start recovery, reference libseeq.c:289-316, on the reversed-pattern EQ table. */
template <int W>
__device__ __forceinline__ uint32_t exact1_reverse(const uint8_t *line, uint32_t i, uint32_t streak,
                                                   uint32_t eqr_base, uint32_t m, uint32_t tau1)
{
   fused_state_t<W> st;
   st.init(m);
   uint32_t j = 0, d = tau1, last_d, ignores = 0;
   do {
      ++j;
      const fused_eq_t<W> ev = fused_eq_load<W>(eqr_base + ((uint32_t)line[i - j] << (W == 1 ? 2 : 3)));
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

template <int MODE, int W>
__global__ __launch_bounds__(256) void k_exact1(ScanArgs a, const uint32_t *eq2)
{
   __shared__ __align__(8) uint32_t s_eqf[256 * W];
   __shared__ __align__(8) uint32_t s_eqr[256 * W];
   __shared__ __align__(16) uint8_t s_blk[256 * EXACT1_ROW];
   for (int i = threadIdx.x; i < 256 * W; i += 256) { s_eqf[i] = eq2[i]; s_eqr[i] = eq2[256 * W + i]; }
   __syncthreads();
   const uint32_t eqf_base = (uint32_t)(uintptr_t)(fused_lds_cu32 *)s_eqf;
   const uint32_t eqr_base = (uint32_t)(uintptr_t)(fused_lds_cu32 *)s_eqr;
   const Counters *c = a.cnt;
   const uint32_t nhl = c->seg_nhitlines;
   const int match_opt = a.options & 3;
   if (MODE == SQ_MODE_EMIT && (c->overflow & 4u)) return;
   const uint32_t m = (uint32_t)a.m, tau1 = (uint32_t)a.tau + 1;
   const bool count_any = a.want != SEEQDEV_WANT_COUNTMATCH && !(a.want == SEEQDEV_WANT_RECORDS && match_opt == SQ_ALL);
   const bool by_nh = a.use_nh != 0;                       /* record slots come from the scanned per-line counts */
   uint8_t *row = s_blk + threadIdx.x * EXACT1_ROW;
   const uint32_t stride = gridDim.x * 256;
   /* wave-uniform trip count so that every lane of a wave takes part in the wave-level votes */
   const uint32_t kmax = (nhl + stride - 1) / stride * stride;
   for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < kmax; k += stride) {
      bool done = k >= nhl;
      const uint32_t hs = done ? 0u : a.hit_start[k];
      if (hs == 0xFFFFFFFFu) done = true;                  /* k_stream: repeat of the previous entry's line */
      if (MODE == SQ_MODE_COUNT && a.use_nh == 3 && count_any && !c->dirty) {
         /* clean text: k_stream's verdict is exact (wave-uniform branch) */
         if (k < nhl) a.nh[k] = done ? 0u : 1u;
         continue;
      }
      const uint64_t off = done ? a.seg_base : a.seg_base + hs;
      const uint8_t *line = a.text + off;
      fused_state_t<W> st;
      st.init(m);
      uint32_t streak = tau1, nhits = 0, best_d = tau1, best_end = 0, pos = 0;
      bool latch = false;
      seeqdev_hit_t *out = nullptr;
      uint32_t out_cap = 0, line_no = 0;
      if (MODE == SQ_MODE_EMIT && !done) {
         line_no = a.hit_line[k];
         if (match_opt == SQ_ALL) { out = a.records + c->records + a.nh[k]; out_cap = 0xFFFFFFFFu; }
         else { out = a.records + c->records + (by_nh ? a.nh[k] : k); out_cap = 1; }
      }
      while (__any(!done)) {
         /* next 64 bytes of my line -> my LDS row (bytes beyond the buffer read as NUL) */
         if (!done) {
#pragma unroll
            for (int q = 0; q < 4; q++)
               *reinterpret_cast<fused_v4u *>(row + 16 * q) = direct_load16(a.text, off + pos + 16 * q, a.nbytes);
         }
         for (uint32_t t = 0; t < 64; t++) {
            if (!__any(!done)) break;
            if (!done) {
               const fused_eq_t<W> ev = fused_eq_load<W>(eqf_base + ((uint32_t)row[t] << (W == 1 ? 2 : 3)));
               const uint32_t e = ev.w0;
               if (!(e & FUSED_FLAG_SKIP)) {
                  uint32_t cur = tau1;
                  bool end = false;
                  if (!(e & FUSED_FLAG_TERM)) {
                     st.step(ev);
                     cur = st.score < tau1 ? st.score : tau1;
                  } else {
                     end = true;
                  }
                  const bool stop = streak < cur, zero = streak == 0;
                  const bool emit = stop ? !latch : zero;
                  latch = stop ? true : zero;
                  if (emit) {
                     const uint32_t p = pos + t;
                     if (MODE == SQ_MODE_COUNT) {
                        nhits++;
                        if (count_any) end = true;                   /* presence is enough: FIRST/BEST/COUNTLINES */
                     } else if (match_opt == SQ_BEST) {
                        if (streak < best_d) { best_d = streak; best_end = p; nhits = 1; }
                     } else {
                        if (nhits < out_cap) {
                           seeqdev_hit_t h;
                           h.line = line_no;
                           h.start = exact1_reverse<W>(line, p, streak, eqr_base, m, tau1);
                           h.end = p;
                           h.dist = streak;
                           out[nhits] = h;
                        }
                        nhits++;
                        if (match_opt != SQ_ALL) end = true;            /* SQ_FIRST / SQ_COUNT: libseeq.c:330 */
                     }
                  }
                  if (end) done = true;
                  streak = cur;
               }
            }
         }
         pos += 64;
      }
      if (k < nhl) {
         if (MODE == SQ_MODE_COUNT) {
            a.nh[k] = nhits;
         } else if (match_opt == SQ_BEST && nhits) {
            seeqdev_hit_t h;
            h.line = line_no;
            h.start = exact1_reverse<W>(line, best_end, best_d, eqr_base, m, tau1);
            h.end = best_end;
            h.dist = best_d;
            out[0] = h;
         }
      }
   }
}

#endif
