/*
 * seeq_verify.h -- k_verify<W, VAR>: the exact pass over CANDIDATE WINDOWS (behind k_pair, the partition filters of
 * k_stream, packed read batches), round 4.  Same job as k_exact1<COUNT> (seeq_exact1.h) -- the reference's acceptance
 * rules (libseeq.c:277-331) and, for one record per line, its reverse start recovery (libseeq.c:289-316) -- on text
 * where no byte is skipped (SQ_FAIL / SQ_CONVERT, line input), in two phases:
 *
 *   phase 1  a bare sweep of the bit-vector column over 64 columns of the window, the text in registers (four 16-byte
 *            loads), fully unrolled: per column one SDWA shift (byte -> LDS address), one EQ look-up, the 12-op Myers step
 *            and FOUR one-instruction bit streams (v_alignbit): "score > tau", the top bits of the horizontal deltas
 *            (+1 / -1 on D[m][j]) and the terminator flag of the EQ word.  18 VALU per column; k_exact1 spends 31 VALU
 *            and 4 SALU on the same column because it runs the rules on every one.
 *   phase 2  the rules, only where they can fire.  Every emission needs streak = sc[j-1] <= tau (libseeq.c:287: a stop
 *            needs streak < cur <= tau + 1; a zero needs streak = 0), and where sc[j-1] > tau the step leaves
 *            latch = 0: the state machine is inert outside the columns that follow a sub-threshold score.  So the
 *            lane walks the set bits of Q = L >> 1 (a handful around each occurrence), gets sc[j-1] exactly as
 *            s_in + popc(PH below j) - popc(MH below j), "the score rises" from PH's bit j (or j is the line's
 *            terminator / the end of the window), and applies
 *               emit = rise ? !latch : zero;  latch = rise ? 1 : zero        (sq_scan_line, seeq_kernel_core.h)
 *            with latch = 0 whenever position j - 1 was not walked.
 *   reverse  (one record per line: SQ_FIRST / SQ_BEST) the reversed pattern over the 32 (two words: 64) bytes before the
 *            match end, from registers, 15 VALU per step and one stream "score > dist"; the first clear bit is the
 *            reference's j.  A lane whose recovery would run past that block (never for the patterns the filters
 *            serve) or whose line sits in the first bytes of the buffer takes exact1_reverse.
 *
 * Windows, stop_at, the FASTA check, the candidate columns and the cache / overflow-list layout for EMIT are
 * k_exact1<COUNT>'s, so k_exact1<EMIT> runs behind it unchanged.  What is new behind it: the per-entry counts are
 * scanned HERE -- every workgroup leaves its 256 entries' exclusive offsets in nh[] and their sum in nh_sum[]; one small
 * launch (k_nh_top) scans the sums, checks the record capacity and, when no EMIT follows, ends the segment: the three scan
 * launches, k_count_nonzero, k_rec_check and k_seg_end are gone from the segment.
 */
#ifndef SEEQ_VERIFY_H_
#define SEEQ_VERIFY_H_

#include "seeq_post.h"            /* VERIFY_ANY / _BEST / _ALL */

/* x << (32 - n), n in 0 .. 32 (0 gives 0): the n stream bits of a register moved to its top */
__device__ __forceinline__ uint32_t verify_top(uint32_t x, uint32_t n) { return (uint32_t)((((uint64_t)x) << 32) >> n); }

/* byte SEL of a text word -> LDS byte address of its EQ entry (entry = 4 << (W - 1) bytes) */
#define VERIFY_ADDR(dst, word, SEL) \
   asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #SEL : "=v"(dst) : "v"(sh), "v"(word))

/* One column: the Myers step of seeq_scan_common.h.  MODE 2 (forward): four streams -- the terminator flag of the EQ word, the top
 * bits of ph / mh (+1 / -1 on D[m][j]) and "score > lim" (lim = tau); MODE 1 (reverse): "score > lim" only (lim = the match
 * distance); MODE 0 (the warm-up columns of a window, where no score can be <= tau): the bare step, 13 VALU with its look-up. */
#define VERIFY_BARE 0
#define VERIFY_REV  1
#define VERIFY_FWD  2
template <int W> struct verify_col;
template <> struct verify_col<1> {
   template <int MODE>
   static __device__ __forceinline__ void run(const fused_eq_t<1> &e, fused_state_t<1> &st, uint32_t lim, uint32_t &L, uint32_t &PH, uint32_t &MH, uint32_t &T)
   {
      const uint32_t eq = e.w0, pv = st.pv, mv = st.mv;
      if (MODE == VERIFY_FWD) T = __builtin_amdgcn_alignbit(eq, T, 1);
      const uint32_t s = (eq & pv) + pv;
      const uint32_t d0 = ((s ^ pv) | eq) | mv;
      const uint32_t ph = mv | ~(d0 | pv);
      const uint32_t mh = pv & d0;
      if (MODE == VERIFY_FWD) { PH = __builtin_amdgcn_alignbit(PH, ph, 31); MH = __builtin_amdgcn_alignbit(MH, mh, 31); }
      uint32_t ph2, mh2, score = st.score;
      asm("v_add_co_u32 %0, vcc, %2, %2\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "=v"(ph2), "+v"(score) : "v"(ph) : "vcc");
      asm("v_add_co_u32 %0, vcc, %2, %2\n\tv_subbrev_co_u32 %1, vcc, 0, %1, vcc" : "=v"(mh2), "+v"(score) : "v"(mh) : "vcc");
      st.pv = mh2 | ~(d0 | ph2);
      st.mv = ph2 & d0;
      st.score = score;
      if (MODE != VERIFY_BARE) L = __builtin_amdgcn_alignbit(L, lim - score, 31);
   }
};
template <> struct verify_col<2> {
   template <int MODE>
   static __device__ __forceinline__ void run(const fused_eq_t<2> &e, fused_state_t<2> &st, uint32_t lim, uint32_t &L, uint32_t &PH, uint32_t &MH, uint32_t &T)
   {
      const uint32_t pv0 = st.pv0, pv1 = st.pv1, mv0 = st.mv0, mv1 = st.mv1;
      if (MODE == VERIFY_FWD) T = __builtin_amdgcn_alignbit(e.w0, T, 1);
      const uint64_t pv = ((uint64_t)pv1 << 32) | pv0, eq = ((uint64_t)e.w1 << 32) | e.w0;
      const uint64_t s = (eq & pv) + pv;
      const uint32_t s0 = (uint32_t)s, s1 = (uint32_t)(s >> 32);
      const uint32_t d00 = ((s0 ^ pv0) | e.w0) | mv0, d01 = ((s1 ^ pv1) | e.w1) | mv1;
      const uint32_t ph0 = mv0 | ~(d00 | pv0), ph1 = mv1 | ~(d01 | pv1);
      const uint32_t mh0 = pv0 & d00, mh1 = pv1 & d01;
      if (MODE == VERIFY_FWD) { PH = __builtin_amdgcn_alignbit(PH, ph1, 31); MH = __builtin_amdgcn_alignbit(MH, mh1, 31); }
      uint32_t p0, p1, m0, m1, score = st.score;
      asm("v_add_co_u32 %0, vcc, %3, %3\n\tv_addc_co_u32 %1, vcc, %4, %4, vcc\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc"
          : "=&v"(p0), "=&v"(p1), "+v"(score) : "v"(ph0), "v"(ph1) : "vcc");
      asm("v_add_co_u32 %0, vcc, %3, %3\n\tv_addc_co_u32 %1, vcc, %4, %4, vcc\n\tv_subbrev_co_u32 %2, vcc, 0, %2, vcc"
          : "=&v"(m0), "=&v"(m1), "+v"(score) : "v"(mh0), "v"(mh1) : "vcc");
      st.pv0 = m0 | ~(d00 | p0); st.pv1 = m1 | ~(d01 | p1);
      st.mv0 = p0 & d00;         st.mv1 = p1 & d01;
      st.score = score;
      if (MODE != VERIFY_BARE) L = __builtin_amdgcn_alignbit(L, lim - score, 31);
   }
};

/* the four columns of one text word (reverse: bytes 3, 2, 1, 0) */
template <int W, int MODE>
__device__ __forceinline__ void verify_word(uint32_t word, uint32_t eq_base, uint32_t sh, fused_state_t<W> &st, uint32_t lim,
                                            uint32_t &L, uint32_t &PH, uint32_t &MH, uint32_t &T)
{
   uint32_t a0, a1, a2, a3;
   VERIFY_ADDR(a0, word, 0); VERIFY_ADDR(a1, word, 1); VERIFY_ADDR(a2, word, 2); VERIFY_ADDR(a3, word, 3);
   const fused_eq_t<W> e0 = fused_eq_load<W>(eq_base + a0), e1 = fused_eq_load<W>(eq_base + a1),
                       e2 = fused_eq_load<W>(eq_base + a2), e3 = fused_eq_load<W>(eq_base + a3);
   if (MODE != VERIFY_REV) {
      verify_col<W>::template run<MODE>(e0, st, lim, L, PH, MH, T);
      verify_col<W>::template run<MODE>(e1, st, lim, L, PH, MH, T);
      verify_col<W>::template run<MODE>(e2, st, lim, L, PH, MH, T);
      verify_col<W>::template run<MODE>(e3, st, lim, L, PH, MH, T);
   } else {
      verify_col<W>::template run<MODE>(e3, st, lim, L, PH, MH, T);
      verify_col<W>::template run<MODE>(e2, st, lim, L, PH, MH, T);
      verify_col<W>::template run<MODE>(e1, st, lim, L, PH, MH, T);
      verify_col<W>::template run<MODE>(e0, st, lim, L, PH, MH, T);
   }
}

__device__ __forceinline__ uint32_t verify_word_of(const fused_v4u (&v)[4], int g)
{
   const fused_v4u &q = v[g >> 2];
   return (g & 3) == 0 ? q.x : (g & 3) == 1 ? q.y : (g & 3) == 2 ? q.z : q.w;
}

/* Reverse start recovery of the lanes with `need` (libseeq.c:289-316; no byte is skipped here): the match ends before
 * column i of the line at `off`, distance `dist`.  Wave-wide: every lane of the wave calls it. */
template <int W>
__device__ __forceinline__ uint32_t verify_reverse(const ScanArgs &a, bool need, uint64_t off, uint32_t i, uint32_t dist,
                                                   uint32_t eqr_base, uint32_t sh, uint32_t m, uint32_t tau1)
{
   constexpr int NB = W == 1 ? 32 : 64;                   /* bytes before the match end held in registers = steps of the fast path */
   constexpr int NV = NB / 16;
   const uint64_t end = off + i;
   const bool fast = need && end >= (uint64_t)NB;
   fused_v4u v[4];
   {
#pragma unroll
      for (int q = 0; q < NV; q++) v[q] = fused_v4u{0u, 0u, 0u, 0u};
      if (fast) {
         const uint8_t *p = a.text + (end - NB);
#pragma unroll
         for (int q = 0; q < NV; q++) v[q] = *reinterpret_cast<const fused_v4u_unaligned *>(p + 16 * q);
      }
   }
   fused_state_t<W> st;
   st.init(m);
   uint32_t Rh = 0, Rl = 0, dummy = 0;
   uint32_t nsteps = NB;                                  /* wave-uniform */
#pragma unroll
   for (int g = 0; g < NB / 4; g++) {
      /* a lane goes on while all its scores were above dist and its line has columns left */
      const uint32_t seen = g < 8 ? Rh : Rl;
      const uint32_t full = (g & 7) == 0 ? 0u : (1u << (4 * (g & 7))) - 1u;      /* every step of this register so far "above" */
      const bool more = fast && (g < 8 || Rh == 0xFFFFFFFFu) && seen == full && 4u * g < i;
      if (!__any(more)) { nsteps = 4u * g; break; }
      const uint32_t word = verify_word_of(v, NV * 4 - 1 - g);
      verify_word<W, VERIFY_REV>(word, eqr_base, sh, st, dist, g < 8 ? Rh : Rl, dummy, dummy, dummy);
   }
   uint32_t start = 0;
   bool slow = need && !fast;
   if (fast) {
      const uint32_t nh_ = nsteps < 32u ? nsteps : 32u, nl_ = nsteps > 32u ? nsteps - 32u : 0u;
      const uint64_t F = ((uint64_t)verify_top(~Rh, nh_) << 32) | verify_top(~Rl, nl_);      /* bit 63 - (j - 1): step j reached score <= dist */
      const uint32_t j = F ? (uint32_t)__builtin_clzll(F) + 1u : 0xFFFFFFFFu;
      if (j <= i) start = i - j;                          /* (libseeq.c:315 with last_d > d: jj = j) */
      else slow = true;                                   /* not within this block / not before the line's first byte: the literal loop */
   }
   if (slow) start = exact1_reverse<W>(a.text, off, a.nbytes, i, dist, eqr_base, m, tau1, nullptr);
   return start;
}

/* Phase 2 over the 32 columns of one stream register (first column in bit 31): the acceptance rules at the columns that follow a
 * score <= tau.  L: sub-threshold columns (already cut at the terminator), tcol: column of the terminator's step within the group
 * (> 32: none here), s_in: the score before the group, p0: line column of the group's first one. */
template <int VAR>
struct verify_rules {
   uint32_t prevL, latch, nhits, best_d, best_end, ce0, ce1;
   bool done;
   template <typename EMIT2>
   __device__ __forceinline__ void group(uint32_t L, uint32_t PH, uint32_t MH, uint32_t tcol, uint32_t s_in, uint32_t p0, EMIT2 second)
   {
      uint32_t Q = (L >> 1) | (prevL << 31);
      int lastj = -1;
      while (Q) {
         const uint32_t j = (uint32_t)__builtin_clz(Q);
         Q &= ~(0x80000000u >> j);
         const uint32_t below = ~(0xFFFFFFFFu >> j);                                   /* columns 0 .. j-1 of the group */
         const uint32_t streak = s_in + (uint32_t)__popc(PH & below) - (uint32_t)__popc(MH & below);   /* sc[j-1], exact (<= tau) */
         const bool rise = j == tcol || ((PH << j) & 0x80000000u) != 0u;
         const bool zero = streak == 0u;
         const uint32_t p = p0 + j;
         if (VAR == VERIFY_BEST) {
            /* no latch: an emission the latch suppresses never beats best_d (exact1_body) */
            if ((rise || zero) && streak < best_d) { best_d = streak; best_end = p; }
         } else {
            if ((int)j != lastj + 1) latch = 0u;
            const bool emit = rise ? latch == 0u : zero;
            latch = (rise || zero) ? 1u : 0u;
            lastj = (int)j;
            if (emit) {
               if (nhits == 0u) { ce0 = p; ce1 = streak; }
               else second(p, streak);
               nhits++;
               if (VAR == VERIFY_ANY) { Q = 0; done = true; }
            }
         }
      }
      if (VAR != VERIFY_BEST && lastj != 31) latch = 0u;
      prevL = L & 1u;
   }
};

/* Text with bytes outside the alphabet somewhere (Counters.dirty: FASTQ quality lines, lower case, binary junk): may the window of
 * this lane's candidate be trusted?  The walk itself meets every byte of the window and ends the line at one that ends it; what
 * it cannot see is such a byte BEFORE the window, in text[off, off + pos) -- the reference stopped there (libseeq.c:267-270) and
 * the line has no occurrence behind it.  Under SQ_FAIL any byte outside { A C G T U N a c g t u n } counts (fused_bad4),
 * under SQ_CONVERT a NUL only (every other byte is an N there, libseeq.c:223-228).  Returns true when the
 * stretch holds one: the lane then scans its line from the first byte, which is exact whatever the bytes are.  Wave-wide (every
 * lane calls it; lanes with pos = 0 look at nothing): 64 bytes per round, the four loads in flight together. */
__device__ __forceinline__ bool verify_prefix_dirty(const ScanArgs &a, uint64_t off, uint32_t pos)
{
   const bool nul_only = (a.options & SQ_CONVERT) != 0;
   uint32_t bad = 0;
   for (uint32_t o = 0; __any(o < pos && bad == 0u); o += 64u) {
      if (o >= pos || bad) continue;
      const uint64_t p0 = off + o;
      fused_v4u v[4];
      if (p0 + 64 <= a.nbytes) {
#pragma unroll
         for (int q = 0; q < 4; q++) v[q] = *reinterpret_cast<const fused_v4u_unaligned *>(a.text + p0 + 16 * q);
      } else {
#pragma unroll
         for (int q = 0; q < 4; q++) v[q] = direct_load16(a.text, p0 + 16 * q, a.nbytes);
      }
      const uint32_t n = pos - o;                           /* bytes of this block that lie before the window */
#pragma unroll
      for (int g = 0; g < 16; g++) {
         const uint32_t w = verify_word_of(v, g);
         const uint32_t b = nul_only ? (~(((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w) & 0x80808080u) : fused_bad4(w);
         const uint32_t keep = n >= 4u * g + 4u ? 0xFFFFFFFFu : n <= 4u * g ? 0u : (1u << (8u * (n - 4u * g))) - 1u;
         bad |= b & keep;
      }
   }
   return bad != 0u;
}

template <int W, int VAR>
__device__ __forceinline__ void verify_body(const ScanArgs &a, const uint32_t *eq2, const uint32_t *hit_col, uint4 *cache)
{
   __shared__ __align__(8) uint32_t s_eqf[256 * W];
   __shared__ __align__(8) uint32_t s_eqr[256 * W];
   __shared__ uint32_t s_novf[4], s_wave[4];
   if (threadIdx.x < 4) s_novf[threadIdx.x] = 0;
   for (int i = threadIdx.x; i < 256 * W; i += 256) { s_eqf[i] = eq2[i]; s_eqr[i] = eq2[256 * W + i]; }
   __syncthreads();
   const uint32_t eqf_base = (uint32_t)(uintptr_t)(fused_lds_cu32 *)s_eqf;
   const uint32_t eqr_base = (uint32_t)(uintptr_t)(fused_lds_cu32 *)s_eqr;
   Counters *c = a.cnt;
   const uint32_t nhl = c->seg_nhitlines;
   const uint32_t m = (uint32_t)a.m, tau = (uint32_t)a.tau, tau1 = tau + 1u;
   const bool caching = cache != nullptr && a.want == SEEQDEV_WANT_RECORDS;
   const bool dirty = c->dirty != 0;
   uint32_t sh = W == 1 ? 2u : 3u;
   asm volatile("" : "+v"(sh));                            /* (SDWA takes the shift from a register) */
   /* overflow lists (VERIFY_ALL with records): k_exact1's layout -- the free entries above the per-line ones, one list per wave of the grid */
   const uint32_t ovf_r = (a.cap_hitlines > nhl ? a.cap_hitlines - nhl : 0u) / (gridDim.x * 4u);
   const uint32_t wave_id = threadIdx.x >> 6;
   uint4 *const ovf_all = cache ? cache + nhl : nullptr;
   /* Round 5: the hit list holds REPEATS (an entry per candidate; a line's second and later candidates only stretch its window:
      hit_start = 0xFFFFFFFF) -- a quarter of the entries behind the headline pattern's prefix automaton, more than half behind a
      three-part filter (configs[4]: every part of an occurrence reports) -- and a lane that draws one idles while its wave walks.  So a
      workgroup takes a RANGE of 512 consecutive entries (ScanArgs.vrange), packs the indices of those that are not repeats into LDS (one
      block scan per 256) and walks them 256 at a time: full waves but for the range's last round.  The counts go back to the entries' own
      places (s_cnt), the per-chunk scan of nh[] / nh_sum[] is what it was.  Measured (profiles/r05/ab_verify_range.txt): configs[4]'s post-pass
      2.95 -> 2.17 ms per step with ranges of 512 or 1 024 (2.52 with 256); behind the headline's prefix automaton, where a quarter of the
      entries are repeats, the packing costs more than the idle lanes (0.715 -> 0.75 / 0.77 / 0.81 ms with 256 / 512 / 1 024: an extra block
      scan per chunk, fewer and longer workgroups) -- so the host asks for it behind partition filters only (vrange = 0: as before). */
   __shared__ uint16_t s_idx[1024];
   __shared__ uint32_t s_cnt[1024];
   const bool compact = a.vrange != 0u;                    /* (kernel-uniform; 0: the list holds few repeats -- the packing costs more than the idle lanes) */
   const uint32_t range = (a.vrange == 512u || a.vrange == 1024u) && nhl >= 65536u ? a.vrange : 256u;
   const uint32_t nranges = (nhl + range - 1u) / range;
   /* the overflow lists are one per wave of the GRID (k_emit_all walks them all); with fewer ranges than workgroups the workgroups beyond the
      ranges have nothing to do, and workgroup b fills -- one after the other -- the lists of workgroups b, b + nr, b + 2 nr ... as well */
   /* (recomputed where they are used -- an emission beyond a line's first is rare --: kept live across the walk they cost registers it does not have) */
#define VERIFY_NR        ((nranges == 0u || nranges > gridDim.x) ? gridDim.x : nranges)
#define VERIFY_MY_LISTS  (blockIdx.x < VERIFY_NR ? (gridDim.x - blockIdx.x + VERIFY_NR - 1u) / VERIFY_NR : 0u)
#define VERIFY_LIST_ROOM (ovf_r > 1u ? ovf_r - 1u : 0u)                         /* entries of a list (entry 0 is its count) */
   for (uint32_t rg = blockIdx.x; rg < nranges; rg += gridDim.x) {
    const uint32_t r0 = rg * range;
    uint32_t nact = 0;                                      /* block-uniform */
    if (!compact) {                                         /* every entry of the (256-entry) range in its own lane, repeats included */
       s_cnt[threadIdx.x] = 0u;
       nact = nhl - r0 < 256u ? nhl - r0 : 256u;
    }
    else for (uint32_t u = 0; u < range; u += 256u) {
       const uint32_t k = r0 + u + threadIdx.x;
       const bool live = k < nhl;
       const uint32_t kk = (a.hit_idx && live) ? a.hit_idx[k] : k;
       const bool act = live && a.hit_start[kk] != 0xFFFFFFFFu;      /* (a repeat of the previous entry's line: no work) */
       s_cnt[u + threadIdx.x] = 0u;
       uint32_t tot;
       const uint32_t ex = block_excl_scan(act ? 1u : 0u, &tot, s_wave);
       if (act) s_idx[nact + ex] = (uint16_t)(u + threadIdx.x);
       if (live && !act && caching) cache[k] = make_uint4(0u, 0u, 0u, 0u);
       nact += tot;
    }
    __syncthreads();
    for (uint32_t j0 = 0; j0 < nact; j0 += 256u) {
      bool done = j0 + threadIdx.x >= nact;
      const uint32_t k = r0 + (done ? 0u : compact ? (uint32_t)s_idx[j0 + threadIdx.x] : threadIdx.x);
#define VERIFY_MINE (j0 + threadIdx.x < nact)                /* (recomputed where it is needed: one register less across the walk) */
      const uint32_t kk = (a.hit_idx && !done) ? a.hit_idx[k] : k;
      const uint32_t hs = done ? 0u : a.hit_start[kk];
      if (hs == 0xFFFFFFFFu) done = true;                  /* repeat of the previous entry's line (not packed away) */
      const uint64_t off = done ? a.seg_base : a.seg_base + hs;
      if (a.use_nh == 3 && (a.options & SEEQDEV_FASTA) && !done && a.text[off] == '>') done = true;
      uint32_t pos = 0, stop_at = 0xFFFFFFFFu;
      if (!done) {
         /* the window: from skip_back columns before the line's first candidate (clean text: nothing ends the line before
            it, no occurrence ends before it) to m + tau + 1 behind its last one (seeq_pair.h; exact1_body has the argument) */
         const uint32_t col = hit_col[kk];
         if (col > a.skip_back) pos = col - a.skip_back;
         if (a.window_ok) {
            uint32_t lastcol = a.hit_last ? a.hit_last[kk] : col;
            bool unbounded = false;
            if (a.hit_idx) { unbounded = lastcol == 0xFFFFFFFFu; lastcol += a.skip_back - (m + tau1 - 1u); }
            else for (uint32_t j = k + 1; j < nhl && a.hit_start[j] == 0xFFFFFFFFu; j++) lastcol = hit_col[j] - hs;
            if (!unbounded) stop_at = lastcol + m + tau1 + 1u;
         }
      }
      /* dirty (kernel-uniform): a byte outside the alphabet somewhere in the text scanned so far -- it may end this line before the window */
      if (dirty && verify_prefix_dirty(a, off, pos)) pos = 0;
      fused_state_t<W> st;
      st.init(m);
      verify_rules<VAR> r;
      r.prevL = 0; r.latch = 0; r.nhits = 0; r.best_d = tau1; r.best_end = 0; r.ce0 = 0; r.ce1 = 0; r.done = false;
      auto second = [&](uint32_t p, uint32_t streak) {      /* second and later emissions of a line: to my wave's overflow list */
         if (VAR == VERIFY_ALL && caching) {
            const uint32_t idx = atomicAdd(&s_novf[wave_id], 1u);
            const uint32_t list_room = VERIFY_LIST_ROOM;
            if (idx < VERIFY_MY_LISTS * list_room) {
               const uint32_t t = idx / list_room;
               ovf_all[(size_t)((blockIdx.x + t * VERIFY_NR) * 4u + wave_id) * ovf_r + 1u + (idx - t * list_room)] = make_uint4(k, r.nhits, p, streak);
            }
         }
      };
      while (__any(!done)) {
         /* ---- phase 1: up to 64 columns from registers ---- */
         fused_v4u v[4];
         {
            const uint64_t p0 = off + pos;
            if (p0 + 64 <= a.nbytes) {
#pragma unroll
               for (int q = 0; q < 4; q++) v[q] = *reinterpret_cast<const fused_v4u_unaligned *>(a.text + p0 + 16 * q);
            } else {
#pragma unroll
               for (int q = 0; q < 4; q++) v[q] = direct_load16(a.text, p0 + 16 * q, a.nbytes);      /* bytes beyond the buffer read as NUL */
            }
         }
         const uint32_t stop_rel = done ? 0u : stop_at - pos;          /* the window ends before this column of the block */
         const uint32_t s_in = st.score;
         uint32_t Lh = 0, Ll = 0, Ph = 0, Pl = 0, Mh = 0, Ml = 0, Th = 0, Tl = 0;
         uint32_t ncols = 64;                                           /* wave-uniform */
#pragma unroll
         for (int g = 0; g < 16; g++) {
            if (!__any(stop_rel > 4u * g)) { ncols = 4u * g; break; }
            verify_word<W, VERIFY_FWD>(verify_word_of(v, g), eqf_base, sh, st, tau, g < 8 ? Lh : Ll, g < 8 ? Ph : Pl, g < 8 ? Mh : Ml, g < 8 ? Th : Tl);
         }
         /* the streams, first column of a register in bit 31 */
         const uint32_t nh_ = ncols < 32u ? ncols : 32u, nl_ = ncols > 32u ? ncols - 32u : 0u;
         Lh = verify_top(~Lh, nh_); Ph = verify_top(Ph, nh_); Mh = verify_top(Mh, nh_); Th = verify_top(__builtin_bitreverse32(Th), nh_);
         Ll = verify_top(~Ll, nl_); Pl = verify_top(Pl, nl_); Ml = verify_top(Ml, nl_); Tl = verify_top(__builtin_bitreverse32(Tl), nl_);
         /* the line's terminator (a flagged EQ word) or the end of the window, whichever comes first: that step sees tau + 1 */
         uint32_t tcol = Th ? (uint32_t)__builtin_clz(Th) : Tl ? 32u + (uint32_t)__builtin_clz(Tl) : 64u;
         tcol = stop_rel < tcol ? stop_rel : tcol;
         if (done) { Lh = 0; Ll = 0; r.prevL = 0; }
         if (tcol < 32u) { Lh &= ~(0xFFFFFFFFu >> tcol); Ll = 0; }
         else if (tcol < 64u) Ll &= ~(0xFFFFFFFFu >> (tcol - 32u));
         /* ---- phase 2: the acceptance rules at the columns that follow a score <= tau ---- */
         r.group(Lh, Ph, Mh, tcol, s_in, pos, second);
         if (__any(!r.done && (Ll | r.prevL) != 0u)) {
            const uint32_t s_mid = s_in + (uint32_t)__popc(Ph) - (uint32_t)__popc(Mh);
            if (!r.done) r.group(Ll, Pl, Ml, tcol - 32u, s_mid, pos + 32u, second);
         } else {
            if (VAR != VERIFY_BEST) r.latch = 0u;            /* (nothing walked in the second half) */
            r.prevL = 0u;
         }
         if (tcol < 64u || r.done) done = true;
         pos += done ? 0u : 64u;
      }
      uint32_t nhits = r.nhits, ce0 = r.ce0, ce1 = r.ce1;
      if (VAR == VERIFY_BEST) { nhits = r.best_d < tau1 ? 1u : 0u; ce0 = r.best_end; ce1 = r.best_d; }
      /* the start of the line's (first) record is recovered here: EMIT only copies it (SQ_ALL: k_emit_all recovers the starts of the
         second and later records of a line from the overflow lists) */
      uint32_t ce2 = 0, ce3 = 0;
      if (caching) {
         const bool need = VERIFY_MINE && nhits != 0u;
         if (__any(need)) {
            const uint32_t s0 = verify_reverse<W>(a, need, off, ce0, ce1, eqr_base, sh, m, tau1);
            if (need) { ce2 = s0; ce3 = 1u; }
         }
      }
      if (caching && VERIFY_MINE) cache[k] = make_uint4(ce0, ce1, ce2, ce3);
      if (VERIFY_MINE) s_cnt[k - r0] = nhits;
#undef VERIFY_MINE
    }
    __syncthreads();
    /* offsets inside every chunk of 256 entries, the chunks' sums */
    for (uint32_t u = 0; u < range && r0 + u < nhl; u += 256u) {
      const uint32_t k0 = r0 + u, k = k0 + threadIdx.x;
      const uint32_t nhits = s_cnt[u + threadIdx.x];
      uint32_t tot;
      const uint32_t ex = block_excl_scan(nhits, &tot, s_wave);
      if (k < nhl) a.nh[k] = ex;
      if (threadIdx.x == 0) a.nh_sum[k0 >> 8] = tot;
      if (a.nz_sum) {                                       /* entries with >= 1 hit (kernel-uniform branch) */
         const uint32_t nzw = (uint32_t)__popcll(__ballot(nhits != 0u));
         if ((threadIdx.x & 63u) == 0) s_wave[wave_id] = nzw;
         __syncthreads();
         if (threadIdx.x == 0) a.nz_sum[k0 >> 8] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
         __syncthreads();
      }
    }
    __syncthreads();
   }
   /* publish the length of my wave's overflow list (EMIT reads it whatever the match option: an empty list otherwise) */
   if (caching && (threadIdx.x & 63u) == 0) {
      const uint32_t n = s_novf[wave_id];
      if (ovf_r) {
         const uint32_t my_lists = VERIFY_MY_LISTS, list_room = VERIFY_LIST_ROOM, nr = VERIFY_NR;
         for (uint32_t t = 0; t < my_lists; t++) {
            const uint32_t before = t * list_room;
            const uint32_t cnt = n > before ? (n - before < list_room ? n - before : list_room) : 0u;
            ovf_all[(size_t)((blockIdx.x + t * nr) * 4u + wave_id) * ovf_r] = make_uint4(cnt, 0u, 0u, 0u);
         }
         if (n > my_lists * list_room) c->seg_novf = 1u;
      }
      else if (n) c->seg_novf = 1u;
#undef VERIFY_NR
#undef VERIFY_MY_LISTS
#undef VERIFY_LIST_ROOM
   }
}

/* Behind k_verify: ONE workgroup scans the chunk sums in place (exclusive), publishes the segment's record count and its lines
 * with a hit, checks the record capacity and -- when no EMIT pass follows -- ends the segment.  (A ticket inside k_verify would do
 * it without this launch, and was measured: for one workgroup to read what the others wrote, every workgroup has to write back
 * its XCD's L2 first (buffer_wbl2, the agent-scope release) -- 4 096 of them took the post-pass from 1.17 to 4.9 ms per step; and a
 * ticket at the end of the EMIT pass to save k_seg_end costs 100 us per launch: 4 096 atomics on one address at ~25 ns each.) */
__global__ __launch_bounds__(256) void k_nh_top(ScanArgs a)
{
   __shared__ uint32_t s_wave[4];
   Counters *c = a.cnt;
   const uint32_t nhl = c->seg_nhitlines;
   const uint32_t nch = (nhl + 255u) >> 8;
   uint32_t running = 0, nz = 0;
   for (uint32_t b0 = 0; b0 < nch; b0 += 2048u) {          /* eight loads per thread in flight */
      uint32_t item[8], v = 0;
#pragma unroll
      for (int q = 0; q < 8; q++) {
         const uint32_t i = b0 + threadIdx.x * 8u + q;
         item[q] = i < nch ? a.nh_sum[i] : 0u;
         if (a.nz_sum && i < nch) nz += a.nz_sum[i];
      }
#pragma unroll
      for (int q = 0; q < 8; q++) v += item[q];
      uint32_t tot;
      uint32_t ex = running + block_excl_scan(v, &tot, s_wave);
#pragma unroll
      for (int q = 0; q < 8; q++) {
         const uint32_t i = b0 + threadIdx.x * 8u + q;
         if (i < nch) a.nh_sum[i] = ex;
         ex += item[q];
      }
      running += tot;
   }
   if (a.nz_sum) {
      uint32_t tot;
      block_excl_scan(nz, &tot, s_wave);
      nz = tot;
   }
   if (threadIdx.x == 0) {
      c->seg_nrec = running;
      if (a.nz_sum) c->seg_nmatch = nz;
      if (a.want == SEEQDEV_WANT_RECORDS) rec_check_body(a);
      c->emit_nhl = nhl;                                    /* (k_emit1 runs behind the end of the segment) */
      c->emit_base = c->records;
      if (a.fin) seg_end_body(a, (int)a.fin - 1);
   }
}

/* One record per line (SQ_FIRST / SQ_BEST) behind k_verify: everything of the record is in the cache -- copy it to its slot. */
__global__ __launch_bounds__(256) void k_emit1(ScanArgs a, const uint4 *cache)
{
   const Counters *c = a.cnt;
   if (c->overflow & 4u) return;
   const uint32_t nhl = c->emit_nhl;
   const uint64_t base = c->emit_base;
   for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < nhl; k += gridDim.x * 256u) {
      const uint4 ce = cache[k];                            /* {end, dist, start, has a record} */
      if (!ce.w) continue;
      const uint32_t kk = a.hit_idx ? a.hit_idx[k] : k;
      const uint64_t slot = base + nh_at(a, k);
      seeqdev_hit_t h;
      h.line = a.hit_line[kk];
      h.start = ce.z;
      h.end = ce.x;
      h.dist = ce.y;
      a.records[slot] = h;
      a.rec_off[slot] = rec_off_of(a, a.seg_base + a.hit_start[kk], h.line);       /* byte offset of the record's line (seeqdevScanCopyOffsets) */
   }
}

/* SQ_ALL records behind k_verify<VERIFY_ALL>: the first record of every line is whole in the cache (as for k_emit1); the further
 * emissions of a line sit in the overflow lists k_verify's waves filled ({hit-list entry, index in the line, end, dist}; `vgrid` =
 * the workgroups k_verify ran with: one list per wave of that grid) -- their starts are recovered here, sixty-four entries of a list
 * at a time (verify_reverse is wave-wide).  (Until round 4 k_exact1's EMIT pass did both, one lane per LINE with its own reverse
 * loop: 124 us per segment of configs[4], 87 us of the headline reads' --all.)  A list that did not fit (Counters.seg_novf): that
 * EMIT pass, which scans the lines with several records again. */
template <int W>
__global__ __launch_bounds__(256, W == 1 ? 6 : 5) void k_emit_all(ScanArgs a, const uint32_t *eq2, const uint32_t *hit_col, uint4 *cache, uint32_t vgrid)
{
   const Counters *c = a.cnt;
   if (c->overflow & 4u) return;
   if (c->seg_novf) { exact1_body<SQ_MODE_EMIT, W, -1, false>(a, eq2, hit_col, cache); return; }      /* (kernel-uniform) */
   __shared__ __align__(8) uint32_t s_eqr[256 * W];
   for (int i = threadIdx.x; i < 256 * W; i += 256) s_eqr[i] = eq2[256 * W + i];
   __syncthreads();
   const uint32_t eqr_base = (uint32_t)(uintptr_t)(fused_lds_cu32 *)s_eqr;
   const uint32_t nhl = c->seg_nhitlines;
   const uint64_t base = c->records;
   const uint32_t m = (uint32_t)a.m, tau1 = (uint32_t)a.tau + 1u;
   uint32_t sh = W == 1 ? 2u : 3u;
   asm volatile("" : "+v"(sh));
   for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < nhl; k += gridDim.x * 256u) {
      const uint4 ce = cache[k];                            /* {end, dist, start, has a record} */
      if (!ce.w) continue;
      const uint32_t kk = a.hit_idx ? a.hit_idx[k] : k;
      const uint64_t slot = base + nh_at(a, k);
      seeqdev_hit_t h;
      h.line = a.hit_line[kk];
      h.start = ce.z;
      h.end = ce.x;
      h.dist = ce.y;
      a.records[slot] = h;
      a.rec_off[slot] = rec_off_of(a, a.seg_base + a.hit_start[kk], h.line);
   }
   const uint32_t nlists = vgrid * 4u;
   const uint32_t ovf_r = (a.cap_hitlines > nhl ? a.cap_hitlines - nhl : 0u) / nlists;      /* (k_verify's formula) */
   if (!ovf_r) return;
   const uint32_t lane = threadIdx.x & 63u;
   for (uint32_t l = blockIdx.x * 4u + (threadIdx.x >> 6); l < nlists; l += gridDim.x * 4u) {
      const uint4 *ovf = cache + nhl + (size_t)l * ovf_r;
      const uint32_t novf = (uint32_t)__builtin_amdgcn_readfirstlane((int)ovf[0].x);
      for (uint32_t e0 = 1u; e0 <= novf; e0 += 64u) {
         const bool need = e0 + lane <= novf;
         const uint4 o = need ? ovf[e0 + lane] : make_uint4(0u, 0u, 0u, 0u);
         const uint32_t ox = (need && a.hit_idx) ? a.hit_idx[o.x] : o.x;
         const uint32_t hs = need ? a.hit_start[ox] : 0u;
         const uint64_t off = a.seg_base + hs;
         const uint32_t s0 = verify_reverse<W>(a, need, off, o.z, o.w, eqr_base, sh, m, tau1);
         if (need) {
            const uint64_t slot = base + nh_at(a, o.x) + o.y;
            seeqdev_hit_t h;
            h.line = a.hit_line[ox];
            h.start = s0;
            h.end = o.z;
            h.dist = o.w;
            a.records[slot] = h;
            a.rec_off[slot] = rec_off_of(a, off, h.line);
         }
      }
   }
}

template <int W, int VAR, int OCC>
__global__ __launch_bounds__(256, OCC) void k_verify(ScanArgs a, const uint32_t *eq2, const uint32_t *hit_col, uint4 *cache)
{
   verify_body<W, VAR>(a, eq2, hit_col, cache);
}

#undef VERIFY_ADDR

#endif
