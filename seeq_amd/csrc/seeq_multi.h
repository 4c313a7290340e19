/*
 * seeq_multi.h -- SEVERAL PATTERNS, ONE WALK over the text (barcode sets; seeqdevScanRunMulti).  The reference names the
 * multi-pattern search as the one place where its algorithm has parallel work (doc/response.tex:358-360); its own API runs
 * one pattern per scan (seeq.c:307-437), and so did this library until round 3 -- N patterns, N walks over the text.
 *
 * One k_pair walk over the UNION pair automaton of the patterns (seeq_dfa.h section 4) makes the candidate list of the
 * text: lines in which ANY pattern may occur, with the first and last candidate column of each.  Then, per segment:
 *
 *   k_multi_resolve   one lane per candidate line: the resolve automaton (the same patterns advanced together without
 *                     restart, byte by byte, table in LDS when it fits 64 KB, else read through L2) walked over the line's
 *                     window [first candidate - maxspan, last candidate + maxspan + 2) gives the SET of patterns that can
 *                     occur in the line (a superset; the exact set when the automata carry whole patterns).
 *   k_multi_reduce / _top / _apply   a scan per pattern (blockIdx.y = pattern) over bit p of the sets: the (line, pattern)
 *                     pairs, grouped by pattern, in line order -- pattern p's candidate list, in its own region of the
 *                     workspace, with its own Counters.
 *   k_exact1 COUNT / EMIT per pattern over its own list (window = the line's union window): counts and records as a scan
 *                     of the pattern alone produces them -- the kernels, the rules and the record order are the same.
 *
 * Host side of the superset / window argument: tests/test_kernel_core_host.py::test_multi_pattern_automata_...
 */
#ifndef SEEQ_MULTI_H_
#define SEEQ_MULTI_H_

#define SEEQ_MULTI_MAX 32

struct MultiArgs {
   const uint8_t *text;
   uint64_t       nbytes;
   uint64_t       seg_base;       /* the union scan's biased base: hit_start[] is relative to it */
   /* the union's hit list (after k_stream_bounds): first entries and repeats */
   const uint32_t *hit_start, *hit_line, *hit_col, *nh;
   Counters      *ucnt;
   /* resolve automaton */
   const uint16_t *res_next;      /* 8 entries (16 bytes) per state: classes A C G T N */
   const uint32_t *res_mask;
   uint32_t       res_states;
   uint32_t       maxspan;
   uint32_t       window_ok;
   int            options;
   /* per candidate line */
   uint32_t      *lmask, *lfirst, *llast;
   /* per pattern: regions of capP entries */
   uint32_t       npat, capP;
   uint32_t      *p_start, *p_line, *p_col, *p_last, *p_nh;
   Counters      *pcnt;           /* [npat] */
   uint32_t      *bsum;           /* [npat * nb] block sums */
   uint32_t       nb;
   uint32_t       span[SEEQ_MULTI_MAX];      /* m + tau of every pattern */
};

static constexpr int MULTI_ITEMS = 8;
static constexpr int MULTI_BLOCK = 256 * MULTI_ITEMS;

/* INLDS: the resolve table and the masks are staged in LDS (res_states * 20 bytes) */
template <bool INLDS>
__global__ __launch_bounds__(256) void k_multi_resolve(MultiArgs a)
{
   extern __shared__ __align__(16) uint8_t ms_tab[];
   __shared__ uint8_t s_cls[256];
   for (int b = threadIdx.x; b < 256; b += 256) s_cls[b] = sq_class_of((uint32_t)b, a.options);
   if (INLDS) {                                           /* the table, then the masks */
      const fused_v4u *src = reinterpret_cast<const fused_v4u *>(a.res_next);
      for (uint32_t i = threadIdx.x; i < a.res_states; i += 256) reinterpret_cast<fused_v4u *>(ms_tab)[i] = src[i];
      uint32_t *mk = reinterpret_cast<uint32_t *>(ms_tab + (size_t)a.res_states * 16);
      for (uint32_t i = threadIdx.x; i < a.res_states; i += 256) mk[i] = a.res_mask[i];
   }
   __syncthreads();
   const uint16_t *tab = INLDS ? reinterpret_cast<const uint16_t *>(ms_tab) : a.res_next;
   const uint32_t *masks = INLDS ? reinterpret_cast<const uint32_t *>(ms_tab + (size_t)a.res_states * 16) : a.res_mask;
   const Counters *c = a.ucnt;
   const uint32_t nhl = c->seg_nhitlines;
   const bool whole = !a.window_ok || c->dirty != 0;      /* no windows: every candidate line from its first byte to its end */
   const uint32_t stride = gridDim.x * 256;
   for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < nhl; k += stride) {
      const uint32_t hs = a.hit_start[k];
      if (hs == 0xFFFFFFFFu) { a.lmask[k] = 0u; continue; }      /* a repeat: its line's first entry speaks for it */
      const uint32_t col = a.hit_col[k];
      uint32_t lastcol = col, unbounded = a.nh[k] & 2u;
      for (uint32_t j = k + 1; j < nhl && a.hit_start[j] == 0xFFFFFFFFu; j++) { lastcol = a.hit_col[j] - hs; unbounded |= a.nh[j] & 2u; }
      const bool all = whole || unbounded != 0;
      const uint32_t from = (all || col <= a.maxspan) ? 0u : col - a.maxspan;
      const uint32_t to = all ? 0xFFFFFFFFu : lastcol + a.maxspan + 2u;
      const uint64_t off = a.seg_base + hs;
      uint32_t q = 0, acc = 0;
      bool done = false;
      for (uint32_t pos = from; pos < to && !done; pos += 16) {
         const fused_v4u v = direct_load16(a.text, off + pos, a.nbytes);       /* (bytes beyond the buffer read as NUL: a terminator) */
         const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
         for (int i = 0; i < 16; i++) {
            const uint32_t cls = s_cls[(w[i >> 2] >> (8 * (i & 3))) & 0xFFu];
            done = done || cls >= 5u || pos + (uint32_t)i >= to;                /* (SQC_SKIP cannot occur: SQ_IGNORE is not served) */
            if (!done) {
               q = tab[q * 8u + cls];
               acc |= masks[q];
            }
         }
      }
      a.lmask[k] = acc;
      a.lfirst[k] = all ? 0u : col;                       /* (column 0: k_exact1 starts at the line's first byte) */
      a.llast[k] = all ? 0xFFFFFFFFu : lastcol;
   }
}

/* bit y of the sets, summed per block of MULTI_BLOCK entries */
__global__ __launch_bounds__(256) void k_multi_reduce(MultiArgs a)
{
   __shared__ uint32_t s_wave[4];
   const uint32_t nhl = a.ucnt->seg_nhitlines;
   const uint32_t y = blockIdx.y, base = blockIdx.x * MULTI_BLOCK;
   if (base >= nhl) return;                                /* (the grid is sized for the workspace; k_multi_top reads the blocks below nhl only) */
   uint32_t v = 0;
#pragma unroll
   for (int k = 0; k < MULTI_ITEMS; k++) {
      const uint32_t i = base + threadIdx.x * MULTI_ITEMS + k;
      if (i < nhl) v += (a.lmask[i] >> y) & 1u;
   }
   uint32_t tot;
   block_excl_scan(v, &tot, s_wave);
   if (threadIdx.x == 0) a.bsum[y * a.nb + blockIdx.x] = tot;
}

/* per pattern (blockIdx.x): exclusive scan of its block sums; its Counters for this segment */
__global__ __launch_bounds__(256) void k_multi_top(MultiArgs a)
{
   __shared__ uint32_t s_wave[4];
   const uint32_t y = blockIdx.x;
   const uint32_t nhl = a.ucnt->seg_nhitlines;
   const uint32_t nbu = (nhl + MULTI_BLOCK - 1) / MULTI_BLOCK;
   uint32_t *b = a.bsum + y * a.nb;
   uint32_t running = 0;
   for (uint32_t b0 = 0; b0 < nbu; b0 += 256) {
      const uint32_t i = b0 + threadIdx.x;
      const uint32_t v = i < nbu ? b[i] : 0;
      uint32_t tot;
      const uint32_t ex = block_excl_scan(v, &tot, s_wave);
      if (i < nbu) b[i] = running + ex;
      running += tot;
      __syncthreads();
   }
   if (threadIdx.x == 0) {
      Counters *pc = a.pcnt + y;
      const Counters *u = a.ucnt;
      uint32_t n = running;
      if (n > pc->need_hitlines) pc->need_hitlines = n;
      if (n > a.capP) { pc->overflow |= 2u; n = 0; }
      if (pc->overflow & 2u) n = 0;
      pc->seg_nhitlines = n;
      pc->seg_nlines = u->seg_nlines;
      pc->seg_nheaders = u->seg_nheaders;
      pc->seg_nrec = 0; pc->seg_nmatch = 0; pc->seg_novf = 0;
      pc->dirty = u->dirty;
      pc->seg_last_nl = u->seg_last_nl;
   }
}

/* the (line, pattern) pairs of pattern y into its region, in line order */
__global__ __launch_bounds__(256) void k_multi_apply(MultiArgs a)
{
   __shared__ uint32_t s_wave[4];
   const uint32_t nhl = a.ucnt->seg_nhitlines;
   const uint32_t y = blockIdx.y, base = blockIdx.x * MULTI_BLOCK;
   const Counters *pc = a.pcnt + y;
   if (base >= nhl) return;
   uint32_t item[MULTI_ITEMS];
   uint32_t v = 0;
#pragma unroll
   for (int k = 0; k < MULTI_ITEMS; k++) {
      const uint32_t i = base + threadIdx.x * MULTI_ITEMS + k;
      item[k] = i < nhl ? (a.lmask[i] >> y) & 1u : 0u;
      v += item[k];
   }
   uint32_t tot;
   uint32_t ex = block_excl_scan(v, &tot, s_wave) + a.bsum[y * a.nb + blockIdx.x];
   if (pc->overflow & 2u) return;                          /* the region is too small: the scan is run again */
   const size_t r0 = (size_t)y * a.capP;
#pragma unroll
   for (int k = 0; k < MULTI_ITEMS; k++) {
      if (item[k]) {
         const uint32_t i = base + threadIdx.x * MULTI_ITEMS + k;
         const uint32_t last = a.llast[i];
         a.p_start[r0 + ex] = a.hit_start[i];
         a.p_line[r0 + ex] = a.hit_line[i];
         a.p_col[r0 + ex] = a.lfirst[i];
         /* k_exact1 ends the scan at last + m + tau + 2: the union window ends at last candidate + maxspan + 2 */
         a.p_last[r0 + ex] = last == 0xFFFFFFFFu ? 0u : last + a.maxspan - a.span[y];
         a.p_nh[r0 + ex] = last == 0xFFFFFFFFu ? 2u : 0u;            /* bit 1: no window end (k_exact1 scans to the end of the line) */
         ex++;
      }
   }
}

#endif
