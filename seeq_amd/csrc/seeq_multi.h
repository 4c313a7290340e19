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
 *   k_exact1m COUNT / EMIT (blockIdx.y = pattern: every pattern in one launch, its arguments from an array in HBM) over each
 *                     pattern's own list (window = the line's union window): counts and records as a scan of the pattern
 *                     alone produces them -- the kernel body, the rules and the record order are k_exact1's; the scans of
 *                     the per-pair counts, the record check and the segment's bookkeeping likewise one launch each.
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
   uint32_t       trust;          /* the sets are exact and only lines are counted: k_multi_top books them, no exact pass */
   int            options;
   /* per candidate line */
   uint32_t      *lmask, *lfirst, *llast;
   /* per pattern: regions of capP entries */
   uint32_t       npat, capP;
   uint32_t      *p_idx;          /* pattern p's list: entries of the union's list whose set holds p */
   Counters      *pcnt;           /* [npat] */
   uint32_t      *bsum;           /* [npat * nb] block sums */
   uint32_t       nb;
};

static constexpr int MULTI_ITEMS = 8;                      /* rounds of 64 lines per wave */
static constexpr int MULTI_BLOCK = 256 * MULTI_ITEMS;      /* lines per workgroup of the split kernels: wave w owns lines [512 w, 512 w + 512) of them */
static constexpr int MULTI_RESOLVE_WG = 1024;              /* the table is staged once per workgroup: large workgroups, full occupancy */

/* INLDS 2: the resolve table and the masks are staged in LDS (res_states * 20 bytes); 1: the table only (* 16 bytes), the
   masks are read through L2 (off the chain of the walk); 0: both through L2 */
template <int INLDS>
__global__ __launch_bounds__(MULTI_RESOLVE_WG) void k_multi_resolve(MultiArgs a)
{
   extern __shared__ __align__(16) uint8_t ms_tab[];
   __shared__ uint8_t s_cls[256];
   if (threadIdx.x < 256) s_cls[threadIdx.x] = sq_class_of((uint32_t)threadIdx.x, a.options);
   if (INLDS) {                                           /* the table, then the masks */
      const fused_v4u *src = reinterpret_cast<const fused_v4u *>(a.res_next);
      for (uint32_t i = threadIdx.x; i < a.res_states; i += MULTI_RESOLVE_WG) reinterpret_cast<fused_v4u *>(ms_tab)[i] = src[i];
      if (INLDS == 2) {
         uint32_t *mk = reinterpret_cast<uint32_t *>(ms_tab + (size_t)a.res_states * 16);
         for (uint32_t i = threadIdx.x; i < a.res_states; i += MULTI_RESOLVE_WG) mk[i] = a.res_mask[i];
      }
   }
   __syncthreads();
   const Counters *c = a.ucnt;
   const uint32_t nhl = c->seg_nhitlines;
   const bool whole = !a.window_ok || c->dirty != 0;      /* no windows: every candidate line from its first byte to its end */
   const uint32_t stride = gridDim.x * MULTI_RESOLVE_WG;
   for (uint32_t k = blockIdx.x * MULTI_RESOLVE_WG + threadIdx.x; k < nhl; k += stride) {
      /* One lane per ENTRY of the list -- a line's first candidate or a repeat: every occurrence of a pattern lies within
         maxspan of ITS candidate (seeq_dfa.h section 4), so each lane walks [c - maxspan, c + maxspan + 2) of its own
         candidate c from the root and ORs what it finds into the line's set.  k_pair lists the FIRST and the LAST candidate
         of a chain (64 bytes of text): an entry in the chain of the entry before it starts where that one starts, the
         stretch between them may hold unlisted candidates.  (A lane per LINE walking from its first candidate to its last
         held its wave for ten blocks of text where a read had a barcode in front and a chance candidate at its end: 0.85 ms
         per 10 M reads; a lane per line jumping from window to window: 0.67 ms, the wave still waits for its longest line.) */
      uint32_t hs = a.hit_start[k];
      uint32_t f = k;                                      /* the line's first entry */
      const bool repeat = hs == 0xFFFFFFFFu;
      if (repeat) {
         if (whole) continue;                              /* the first entry's lane walks the whole line */
         while (f > 0 && hs == 0xFFFFFFFFu) hs = a.hit_start[--f];
         if (hs == 0xFFFFFFFFu) continue;                  /* the line's first candidate belongs to the segment before this one (the run is void: overflow 128) */
      }
      const uint32_t c = repeat ? a.hit_col[k] - hs : a.hit_col[k];
      const uint64_t off = a.seg_base + hs;
      if ((a.options & SEEQDEV_FASTA) && a.text[off] == '>') continue;      /* a candidate inside a FASTA header */
      bool all = whole;
      if (!repeat) {
         a.lfirst[k] = c;
         if (all) { a.lfirst[k] = 0u; atomicMax(&a.llast[k], 0xFFFFFFFFu); }      /* (column 0: k_exact1 starts at the line's first byte and scans to its end) */
      }
      uint32_t from = c;
      if (k > f) {                                         /* same chain as the entry before: from its window's start */
         const uint32_t pc = k - 1 == f ? a.hit_col[f] : a.hit_col[k - 1] - hs;
         if (((hs + pc) >> 6) == ((hs + c) >> 6)) from = pc;
      }
      uint32_t pos = (all || from <= a.maxspan) ? 0u : from - a.maxspan;
      const uint32_t to = all ? 0xFFFFFFFFu : c + a.maxspan + 2u;
      uint32_t q = 0, acc = 0;                             /* q: INLDS the byte offset of the state's row, else the state */
      bool done = false;
      while (!done && pos < to) {
         const fused_v4u v = direct_load16(a.text, off + pos, a.nbytes);       /* (bytes beyond the buffer read as NUL: a terminator) */
         const uint32_t w[4] = {v.x, v.y, v.z, v.w};
         uint32_t cls[16];
#pragma unroll
         for (int i = 0; i < 16; i++) cls[i] = s_cls[(w[i >> 2] >> (8 * (i & 3))) & 0xFFu];      /* (off the chain of the walk: all sixteen go out together) */
#pragma unroll
         for (int i = 0; i < 16; i++) {
            done = done || cls[i] >= 5u || pos + (uint32_t)i >= to;             /* a terminator ends the line (SQC_SKIP cannot occur: SQ_IGNORE is not served) */
            if (INLDS) {
               const uint32_t nq = (uint32_t)*reinterpret_cast<const uint16_t *>(ms_tab + q + 2u * (cls[i] & 7u)) << 4;
               q = done ? q : nq;
               if (INLDS == 2) {
                  const uint32_t mk = *reinterpret_cast<const uint32_t *>(ms_tab + (size_t)a.res_states * 16 + (q >> 2));
                  acc |= done ? 0u : mk;
               } else if (!done) {
                  acc |= a.res_mask[q >> 4];
               }
            } else if (!done) {
               q = a.res_next[q * 8u + cls[i]];
               acc |= a.res_mask[q];
            }
         }
         pos += 16u;
      }
      if (acc) atomicOr(&a.lmask[f], acc);                 /* (lmask[] and llast[] are zeroed before the launch) */
      if (!all) atomicMax(&a.llast[f], c);
   }
}

/* A wave's pattern counts over its 512 lines: lane p ends up with the number of lines whose set holds pattern p. */
__device__ __forceinline__ uint32_t multi_wave_counts(const MultiArgs &a, uint32_t wbase, uint32_t nhl, uint32_t lane)
{
   uint32_t mine = 0;
   for (int r = 0; r < MULTI_ITEMS; r++) {
      const uint32_t i = wbase + (uint32_t)r * 64u + lane;
      const uint32_t mk = i < nhl ? a.lmask[i] : 0u;
      for (uint32_t p = 0; p < a.npat; p++) {
         const uint32_t n = (uint32_t)__popcll(__ballot((mk >> p) & 1u));
         mine += lane == p ? n : 0u;
      }
   }
   return mine;
}

/* per workgroup of 2048 lines: how many of them hold pattern p -> bsum[p][workgroup], every pattern in one pass */
__global__ __launch_bounds__(256) void k_multi_reduce(MultiArgs a)
{
   __shared__ uint32_t s_cnt[4][SEEQ_MULTI_MAX];
   const uint32_t nhl = a.ucnt->seg_nhitlines;
   const uint32_t base = blockIdx.x * MULTI_BLOCK;
   if (base >= nhl) return;                                /* (the grid is sized for the workspace; k_multi_top reads the workgroups below nhl only) */
   const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
   const uint32_t mine = multi_wave_counts(a, base + wave * 64u * MULTI_ITEMS, nhl, lane);
   if (lane < SEEQ_MULTI_MAX) s_cnt[wave][lane] = mine;
   __syncthreads();
   if (threadIdx.x < a.npat) a.bsum[threadIdx.x * a.nb + blockIdx.x] = s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
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
      if (!a.trust && n > pc->need_hitlines) pc->need_hitlines = n;
      if (!a.trust && n > a.capP) { pc->overflow |= 2u; n = 0; }
      if (pc->overflow & 2u) n = 0;
      pc->seg_nhitlines = n;
      pc->seg_nlines = u->seg_nlines;
      pc->seg_nheaders = u->seg_nheaders;
      pc->seg_nrec = 0; pc->seg_nmatch = 0; pc->seg_novf = 0;
      pc->dirty = u->dirty;
      pc->seg_last_nl = u->seg_last_nl;
      if (a.trust) {                                       /* what k_seg_end books behind an exact pass of 0 / 1 verdicts */
         pc->lines += u->seg_nlines - u->seg_nheaders;
         pc->headers += u->seg_nheaders;
         pc->matchlines += running;
         pc->hits += running;
         pc->seg_nhitlines = 0;
      }
   }
}

/* the (line, pattern) pairs into the patterns' regions, in line order: a wave walks its 512 lines 64 at a time; per pattern
   present in the round one ballot gives the ranks, lane p carries pattern p's running position */
__global__ __launch_bounds__(256) void k_multi_apply(MultiArgs a)
{
   __shared__ uint32_t s_cnt[4][SEEQ_MULTI_MAX];
   const uint32_t nhl = a.ucnt->seg_nhitlines;
   const uint32_t base = blockIdx.x * MULTI_BLOCK;
   if (base >= nhl) return;
   const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
   const uint32_t wbase = base + wave * 64u * MULTI_ITEMS;
   const uint32_t mine = multi_wave_counts(a, wbase, nhl, lane);
   if (lane < SEEQ_MULTI_MAX) s_cnt[wave][lane] = mine;
   __syncthreads();
   uint32_t mybase = 0;                                    /* lane p: position of this wave's next pair of pattern p inside region p */
   bool room = true;
   if (lane < a.npat) {
      mybase = a.bsum[lane * a.nb + blockIdx.x];
      for (uint32_t w = 0; w < wave; w++) mybase += s_cnt[w][lane];
      room = !(a.pcnt[lane].overflow & 2u);                /* the region is too small: the scan is run again */
   }
   for (int r = 0; r < MULTI_ITEMS; r++) {
      const uint32_t i = wbase + (uint32_t)r * 64u + lane;
      const uint32_t mk = i < nhl ? a.lmask[i] : 0u;
      if (__ballot(mk != 0u) == 0ull) continue;
      for (uint32_t p = 0; p < a.npat; p++) {
         const bool has = (mk >> p) & 1u;
         const uint64_t bal = __ballot(has);
         if (bal == 0ull) continue;
         const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)mybase, (int)p);
         const bool ok = __builtin_amdgcn_readlane((int)(room ? 1 : 0), (int)p) != 0;
         if (has && ok) a.p_idx[(size_t)p * a.capP + b0 + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = i;
         mybase += lane == p ? (uint32_t)__popcll(bal) : 0u;
      }
   }
}

/* ---- the per-pattern bookkeeping of the exact pass, every pattern in one launch (blockIdx.y / blockIdx.x = pattern) ---- */
__global__ __launch_bounds__(WG) void k_multi_count_nonzero(const MultiExact *mx) { count_nonzero_body(mx[blockIdx.y].a); }

/* exclusive scan of a pattern's per-pair counts nh[0 .. seg_nhitlines) in place, total -> seg_nrec: reduce, top, apply */
__global__ __launch_bounds__(WG) void k_multi_scan_reduce(const MultiExact *mx)
{
   __shared__ uint32_t s_wave[4];
   const MultiExact &m = mx[blockIdx.y];
   const uint32_t n = m.a.cnt->seg_nhitlines;
   const uint32_t base = blockIdx.x * SCAN_BLOCK;
   if (base >= n) return;
   uint32_t v = 0;
#pragma unroll
   for (int k = 0; k < SCAN_ITEMS; k++) {
      const uint32_t i = base + threadIdx.x * SCAN_ITEMS + k;
      if (i < n) v += m.a.nh[i];
   }
   uint32_t tot;
   block_excl_scan(v, &tot, s_wave);
   if (threadIdx.x == 0) m.scan_ws[blockIdx.x] = tot;
}

/* (one workgroup per pattern; with records wanted also what k_rec_check does) */
__global__ __launch_bounds__(WG) void k_multi_scan_top(const MultiExact *mx, int check_records)
{
   __shared__ uint32_t s_wave[4];
   const MultiExact &m = mx[blockIdx.x];
   Counters *c = m.a.cnt;
   const uint32_t n = c->seg_nhitlines;
   const uint32_t nbu = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
   uint32_t running = 0;
   for (uint32_t b0 = 0; b0 < nbu; b0 += WG) {
      const uint32_t i = b0 + threadIdx.x;
      const uint32_t v = i < nbu ? m.scan_ws[i] : 0;
      uint32_t tot;
      const uint32_t ex = block_excl_scan(v, &tot, s_wave);
      if (i < nbu) m.scan_ws[i] = running + ex;
      running += tot;
      __syncthreads();
   }
   if (threadIdx.x == 0) {
      c->seg_nrec = running;
      if (check_records) rec_check_body(m.a);
   }
}

__global__ __launch_bounds__(WG) void k_multi_scan_apply(const MultiExact *mx)
{
   __shared__ uint32_t s_wave[4];
   const MultiExact &m = mx[blockIdx.y];
   const uint32_t n = m.a.cnt->seg_nhitlines;
   const uint32_t base = blockIdx.x * SCAN_BLOCK;
   if (base >= n) return;
   uint32_t item[SCAN_ITEMS];
   uint32_t v = 0;
#pragma unroll
   for (int k = 0; k < SCAN_ITEMS; k++) {
      const uint32_t i = base + threadIdx.x * SCAN_ITEMS + k;
      item[k] = i < n ? m.a.nh[i] : 0;
      v += item[k];
   }
   uint32_t tot;
   uint32_t ex = block_excl_scan(v, &tot, s_wave) + m.scan_ws[blockIdx.x];
#pragma unroll
   for (int k = 0; k < SCAN_ITEMS; k++) {
      const uint32_t i = base + threadIdx.x * SCAN_ITEMS + k;
      if (i < n) m.a.nh[i] = ex;
      ex += item[k];
   }
}

__global__ void k_multi_seg_end(const MultiExact *mx) { seg_end_body(mx[blockIdx.x].a, mx[blockIdx.x].seg_end_flags); }

#endif
