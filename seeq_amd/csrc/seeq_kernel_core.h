/*
 * seeq_kernel_core.h -- per-line device functions of the seeq-mi355x hot path.
 *
 * Everything here runs once per lane (one text line per lane).  The functions
 * are SEEQ_HD (= __host__ __device__ under hipcc) only so that
 * tests/host_harness.cpp can compile the very same code with g++ and fuzz it
 * against the oracle inside the CPU-only container; the shipped library
 * (libseeq_amd.so) calls them from HIP kernels exclusively.
 *
 * What is computed (reference src/libseeq.c):
 *   - per character the capped distance min(tau+1, D[m][j]) of the semi-global
 *     edit-distance matrix (what the reference reads from its lazily built DFA,
 *     libseeq.c:255-264, whose states are the columns of libseeq.c:767-786).
 *     Here it comes from a multi-word Myers/Hyyro bit-vector column update:
 *     ~17 integer VALU ops per character per 32-bit word, no memory traffic
 *     besides the Peq word (LDS) and the text byte.
 *   - the acceptance rules of libseeq.c:277-331 (latch / perfect / stop),
 *   - the reverse start recovery of libseeq.c:289-316.
 */
#ifndef SEEQ_KERNEL_CORE_H_
#define SEEQ_KERNEL_CORE_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define SEEQ_HD __host__ __device__ __forceinline__
#else
#define SEEQ_HD static inline
#endif

/* Character classes after option folding (sq_build_lut). */
#define SQC_TERM 5   /* ends the line            (libseeq.c:267-270) */
#define SQC_SKIP 6   /* skipped, still counted in coordinates (libseeq.c:265-266) */

/* libseeq.h option bits (duplicated so that this header is self-contained). */
#define SQK_BEST     0x01
#define SQK_ALL      0x02
#define SQK_CONVERT  0x04
#define SQK_IGNORE   0x08
#define SQK_STREAM   0x10

/* Byte -> class, folding the reference's two translate tables
 * (seeqcore.h:89-111) with the non-DNA and input options
 * (libseeq.c:223-228,265-270):
 *   Aa Cc Gg TtUu Nn -> 0..4; NUL -> TERM; '\n' -> SKIP if SQ_STREAM else TERM;
 *   anything else -> 4 (SQ_CONVERT) | SKIP (SQ_IGNORE) | TERM (SQ_FAIL).
 * Bytes >= 0x80 are "anything else" (the reference indexes its table with a
 * negative char there, which is undefined). */
SEEQ_HD uint8_t sq_class_of(uint32_t b, int options)
{
   switch (b) {
   case 'A': case 'a': return 0;
   case 'C': case 'c': return 1;
   case 'G': case 'g': return 2;
   case 'T': case 't': case 'U': case 'u': return 3;
   case 'N': case 'n': return 4;
   case 0:    return SQC_TERM;
   case '\n': return (options & SQK_STREAM) ? SQC_SKIP : SQC_TERM;
   default:
      if ((options & 0x0C) == SQK_CONVERT) return 4;
      if ((options & 0x0C) == SQK_IGNORE)  return SQC_SKIP;
      return SQC_TERM;
   }
}

/* ---- multi-word Myers column -------------------------------------------- */
/* Bit i of word w is pattern row 32*w+i+1.  Vertical deltas Pv/Mv, score =
 * D[m][j].  Initial column D[i][0] = i: Pv all ones (libseeq.c:681-682 is the
 * same column saturated at tau+1).  Row 0 is identically 0 (free start in the
 * text, libseeq.c:768), i.e. the horizontal delta shifted into bit 0 is 0. */
template <int W>
struct sq_myers_t {
   uint32_t pv[W];
   uint32_t mv[W];
   int      score;
};

template <int W>
SEEQ_HD void sq_myers_init(sq_myers_t<W> &s, int m)
{
#pragma unroll
   for (int w = 0; w < W; w++) { s.pv[w] = 0xFFFFFFFFu; s.mv[w] = 0u; }
   s.score = m;
}

/* One text character.  eq[w] = Peq word of that character.  topw/topbit
 * locate pattern row m.  The addition is one carry chain over all words,
 * so this is exactly the single-word recurrence on a 32*W-bit integer. */
template <int W>
SEEQ_HD void sq_myers_step(sq_myers_t<W> &s, const uint32_t *eq, int topw, int topbit)
{
   uint32_t carry = 0, ph_in = 0, mh_in = 0;
#pragma unroll
   for (int w = 0; w < W; w++) {
      const uint32_t e = eq[w], pv = s.pv[w], mv = s.mv[w];
      const uint32_t xv = e | mv;
      const uint64_t sum = (uint64_t)(e & pv) + pv + carry;
      carry = (uint32_t)(sum >> 32);
      const uint32_t xh = ((uint32_t)sum ^ pv) | e;
      uint32_t ph = mv | ~(xh | pv);
      uint32_t mh = pv & xh;
      if (w == topw) s.score += (int)((ph >> topbit) & 1u) - (int)((mh >> topbit) & 1u);
      const uint32_t ph_out = ph >> 31, mh_out = mh >> 31;
      ph = (ph << 1) | ph_in;
      mh = (mh << 1) | mh_in;
      ph_in = ph_out;
      mh_in = mh_out;
      s.pv[w] = mh | ~(xv | ph);
      s.mv[w] = ph & xv;
   }
}

/* ---- line access ---------------------------------------------------------- */
struct sq_chunk16_t { uint32_t w[4]; };

/* 16 text bytes starting at absolute offset pos; bytes at or beyond nbytes
 * read as NUL (= line terminator, like the reference's strlen, libseeq.c:245). */
SEEQ_HD sq_chunk16_t sq_load16(const uint8_t *text, uint64_t pos, uint64_t nbytes)
{
   sq_chunk16_t c;
   if (pos + 16 <= nbytes) {
      __builtin_memcpy(&c, text + pos, 16);
   } else {
      c.w[0] = c.w[1] = c.w[2] = c.w[3] = 0;
#pragma unroll
      for (int k = 0; k < 16; k++)
         if (pos + (uint64_t)k < nbytes) c.w[k >> 2] |= (uint32_t)text[pos + k] << ((k & 3) * 8);
   }
   return c;
}

/* ---- reverse start recovery: libseeq.c:289-316 ---------------------------- */
/* text points at the first byte of the line; i = exclusive end of the hit,
 * streak = its distance.  peq_r = Peq of the reversed pattern. */
template <int W, typename LUT, typename PEQ>
SEEQ_HD uint32_t sq_reverse_start(const uint8_t *line, uint32_t i, int streak, const PEQ peq_r, const LUT lut,
                                  int m, int tau)
{
   const int topw = (m - 1) >> 5, topbit = (m - 1) & 31;
   sq_myers_t<W> r;
   sq_myers_init<W>(r, m);
   uint32_t j = 0;
   int d = tau + 1, last_d, ignores = 0;
   do {
      ++j;
      const uint32_t cls = lut[line[i - j]];
      last_d = d;
      if (cls < 5) {
         ignores = 0;
         sq_myers_step<W>(r, &peq_r[cls * W], topw, topbit);
         d = r.score < tau + 1 ? r.score : tau + 1;
      } else {
         ignores++;
      }
   } while (d > streak && j < i);
   const int jj = (int)(last_d < d ? j - 1 : j) - ignores;   /* libseeq.c:315 */
   return (uint32_t)((int)i - jj);
}

/* ---- the acceptance rules per position (k_string: positions shared out over threads) ---- */
/* The rules of libseeq.c:277-331 only look at three consecutive capped scores: with streak = sc[j-1], cur = sc[j]
 * (tau+1 at the terminator's step and before the line start),
 *    latch before step j = stop(j-1) ? 1 : zero(j-1) = sc[j-2] < sc[j-1] ? 1 : sc[j-2] == 0,
 *    emit(j)             = stop(j) ? !latch : zero(j)  with stop(j) = sc[j-1] < sc[j], zero(j) = sc[j-1] == 0,
 * and sc[j] itself only depends on the m + tau characters up to j: a fresh column started m + tau + 1 characters
 * before the first position of interest gives the line's own capped scores from two positions before it on.
 * text[0, len): the line up to its terminator, every byte of class 0..4 (no skipped bytes); positions j0 <= j < j1
 * <= len + 1 (position len is the terminator's step).  ed[j] = emitted distance + 1, or 0.  Returns the emissions. */
template <int W, typename TEXT, typename LUT, typename PEQ>
SEEQ_HD uint32_t sq_emit_window(const TEXT text, uint32_t len, uint32_t j0, uint32_t j1, const PEQ peq_f, const LUT lut,
                                int m, int tau, uint16_t *ed)
{
   const int tau1 = tau + 1, topw = (m - 1) >> 5, topbit = (m - 1) & 31;
   long start = (long)j0 - 2 - (long)(m + tau - 1);
   if (start < 0) start = 0;
   sq_myers_t<W> st;
   sq_myers_init<W>(st, m);
   int s2 = tau1, s1 = tau1;                        /* capped scores of the two positions before j */
   uint32_t cnt = 0;
   for (long j = start; j < (long)j1; j++) {
      int cur = tau1;
      if ((uint32_t)j < len) {
         sq_myers_step<W>(st, &peq_f[(uint32_t)lut[text[j]] * W], topw, topbit);
         cur = st.score < tau1 ? st.score : tau1;
      }
      if (j >= (long)j0) {
         const bool latch = s2 < s1 ? true : s2 == 0;
         const bool stop = s1 < cur, zero = s1 == 0;
         const bool emit = stop ? !latch : zero;
         ed[j] = emit ? (uint16_t)(s1 + 1) : (uint16_t)0;
         cnt += emit ? 1u : 0u;
      }
      s2 = s1; s1 = cur;
   }
   return cnt;
}

/* ---- forward scan of one line: libseeq.c:250-338 --------------------------- */
#define SQ_MODE_ANY   0   /* does the line have >= 1 hit?  (stops at the first)      */
#define SQ_MODE_COUNT 1   /* number of hits under SQ_ALL rules                        */
#define SQ_MODE_EMIT  2   /* write hit records for match_opt FIRST / BEST / ALL       */

struct sq_hit_t { uint32_t line, start, end, dist; };   /* == seeqdev_hit_t */

/* Returns the number of hits found (ANY: 0/1; EMIT+FIRST/BEST: 0/1).
 * The acceptance rules (libseeq.c:277-331) reduce to this per-position update
 * with streak = previous capped distance, cur = current one (both <= tau+1):
 *    stop  = streak < cur                 (:287, "streak <= tau" is implied)
 *    emit  = stop ? !latch : streak == 0  (:278,286,288)
 *    latch = stop ? 1      : streak == 0  (:278,289)
 * SQ_BEST keeps the first emission with the smallest distance (:288,321-325);
 * SQ_FIRST the first emission (:330).  The reference's early exit on
 * min_to_match (:272-275) cannot change any result and is not reproduced. */
template <int W, int MODE, typename LUT, typename PEQ>
SEEQ_HD uint32_t sq_scan_line(const uint8_t *text, uint64_t nbytes, uint64_t line_off, const PEQ peq_f,
                              const PEQ peq_r, const LUT lut, int m, int tau, int match_opt, uint32_t line_no,
                              sq_hit_t *out, uint32_t out_cap)
{
   const int topw = (m - 1) >> 5, topbit = (m - 1) & 31;
   const int tau1 = tau + 1;
   sq_myers_t<W> st;
   sq_myers_init<W>(st, m);
   int streak = tau1;
   bool latch = false;
   bool done = false;
   uint32_t nhits = 0;
   int best_d = tau1;
   uint32_t best_end = 0;
   uint32_t i = 0;
   while (!done) {
      const sq_chunk16_t c = sq_load16(text, line_off + i, nbytes);
#pragma unroll
      for (int k = 0; k < 16; k++) {
         /* no break/continue in here: the 16 steps must unroll so that c.w[] stays in registers */
         const uint32_t b = (c.w[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
         const uint32_t cls = lut[b];
         if (!done && cls != SQC_SKIP) {
            int cur = tau1;
            bool end = false;
            if (cls < 5) {
               sq_myers_step<W>(st, &peq_f[cls * W], topw, topbit);
               cur = st.score < tau1 ? st.score : tau1;
            } else {
               end = true;
            }
            const bool stop = streak < cur;
            const bool zero = streak == 0;
            const bool emit = stop ? !latch : zero;
            latch = stop ? true : zero;
            if (emit) {
               const uint32_t pos = i + (uint32_t)k;
               if (MODE == SQ_MODE_ANY) {
                  nhits = 1;
                  end = true;
               } else if (MODE == SQ_MODE_COUNT) {
                  nhits++;
               } else {
                  if (match_opt == SQK_BEST) {
                     if (streak < best_d) { best_d = streak; best_end = pos; nhits = 1; }
                  } else {
                     if (nhits < out_cap) {
                        sq_hit_t h;
                        h.line = line_no;
                        h.start = sq_reverse_start<W>(text + line_off, pos, streak, peq_r, lut, m, tau);
                        h.end = pos;
                        h.dist = (uint32_t)streak;
                        out[nhits] = h;
                     }
                     nhits++;
                     if (match_opt != SQK_ALL) end = true;   /* SQ_FIRST / SQ_COUNT: libseeq.c:330 */
                  }
               }
            }
            if (end) done = true;
            streak = cur;
         }
      }
      i += 16;
   }
   if (MODE == SQ_MODE_EMIT && match_opt == SQK_BEST && nhits && out_cap) {
      sq_hit_t h;
      h.line = line_no;
      h.start = sq_reverse_start<W>(text + line_off, best_end, best_d, peq_r, lut, m, tau);
      h.end = best_end;
      h.dist = (uint32_t)best_d;
      out[0] = h;
   }
   return nhits;
}

#endif
