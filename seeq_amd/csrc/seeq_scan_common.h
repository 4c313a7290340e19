/*
 * seeq_scan_common.h -- what the one-pass scan kernels (k_stream: seeq_stream.h, k_direct: seeq_direct.h) and the
 * exact pass (k_exact1: seeq_exact1.h) share: the argument block, the top-aligned Myers step for one- and two-word
 * columns, the byte-indexed EQ table entries, DPP wave scans, and the two small kernels that turn per-wave slices
 * into ordered per-line arrays.
 *
 * EQ[byte] (256 entries per direction, built on the host in seeq_device.hip): the top-aligned Peq word of the byte's
 * class (reference seeqcore.h:89-111 folded with the non-DNA option, libseeq.c:223-228,265-270), or a flag for bytes
 * that end the line / are skipped but counted in coordinates.
 */
#ifndef SEEQ_SCAN_COMMON_H_
#define SEEQ_SCAN_COMMON_H_

#include "libseeq.h"
#include "seeq_types.h"

#ifndef SEEQ_WG
#define SEEQ_WG 256            /* threads of the post-pass workgroups (4 waves) */
#endif

#define FUSED_FLAG_TERM 1u      /* byte ends the line                          */
#define FUSED_FLAG_SKIP 2u      /* byte is skipped but counted in coordinates  */
#define FUSED_FLAGS     3u
#define FUSED_MAX_WLEN  30      /* 32-bit word minus the two flag bits         */
#define FUSED_MAX_WLEN2 62      /* two words minus the two flag bits (k_direct<.,2>, k_exact1<.,2>) */

typedef unsigned int fused_v4u __attribute__((ext_vector_type(4)));
typedef fused_v4u fused_v4u_unaligned __attribute__((aligned(1)));   /* the text pointer may have any alignment */

struct FusedArgs {
   const uint8_t *text;        /* whole buffer                                 */
   uint64_t       nbytes;
   uint64_t       seg_base;    /* first byte of the segment                    */
   uint32_t       seg_len;
   uint32_t       first_seg;
   uint32_t       tile_bytes;  /* multiple of 16                               */
   uint32_t       pos_bias;    /* k_stream: hit offsets are relative to seg_base - pos_bias (a line can start before the segment) */
   uint32_t       ntiles;
   const uint32_t *eqtab;      /* [256] top-aligned Peq word or flag, per byte */
   const uint32_t *peq;        /* [2][5][1] bottom-aligned (long-line fallback)*/
   int            m, tau, options, want;
   uint32_t      *tile_cl;     /* per tile: counted lines (headers excluded)   */
   uint32_t      *tile_hits;   /* per tile: hit lines                          */
   uint4         *tmp;         /* hit entries {tile, seq, start, counted rank}: one slice per wave */
   uint32_t       cap_tmp;     /* total entries                                */
   uint32_t       slice_cap;   /* entries per slice = cap_tmp / slices           */
   uint32_t      *wg_hits;     /* per slice (= per wave of k_stream / k_direct): entries stored */
   uint32_t      *wg_part;     /* per slice: {lines, headers, hit lines | overflow<<31, flags (k_stream: 1 = met a byte outside
                                  its alphabet, 2 = wants the long-line variant, 4 = its hit lines are a superset)} */
   uint32_t      *tile_dirty;  /* k_stream, long-line mode: per tile, 1 when it holds a byte outside the alphabet (then its exclusive prefix); else NULL */
   uint64_t      *tile_dmask;  /* k_stream, long-line mode: per tile, one bit per 128-byte chunk (lane) that holds a non-alphabet byte */
   uint32_t      *wg_lastnl;   /* k_stream, per wave: segment-relative offset + 1 of the last newline it saw (0: none); else NULL */
   const uint16_t *dfa;        /* k_stream: transition table, dfa_rows x 8 u16 (seeq_dfa.h) */
   uint32_t       dfa_rows;
   uint32_t       dfa_final_base;   /* k_stream: state value of ACC_NEW; k_pair: state values >= this are flagged rows */
   unsigned long long *clk_probe;   /* NULL, or 4 words the scan kernel's first wave fills: shader clock (s_memtime) and the constant 100 MHz counter
                                       (s_memrealtime) at its start and at its end -- their ratio is the core clock the kernel actually ran at */
   uint32_t       ll_filter;        /* k_stream's long-line variant walks a partition filter: a chain that starts blind (accepting state) reports its first byte */
   uint32_t       skip_thr;         /* k_stream under SQ_IGNORE: m - tau when the automaton is the complete one -- a chain whose line holds fewer characters that
                                       are not skipped makes up no candidate (0: every chain with a skipped byte in its warm-up window does) */
   uint32_t       ig_thr;           /* k_pair under SQ_IGNORE (IG): m - tau, the characters that are not skipped an occurrence needs at least */
   uint32_t       pair;             /* 1: k_pair (two bytes per step; everything it reports is a candidate); 2: k_stream's Myers mode (exact on clean text
                                       under SQ_CONVERT too).  Nonzero: text with non-DNA bytes needs no re-run on another kernel */
   Counters      *cnt;
};

/* One Myers column step on a TOP-aligned pattern (row m = bit 31).  The two
 * left shifts double as the extraction of the horizontal delta of row m: the
 * carry out of ph+ph / mh+mh is +1 / -1 on D[m][j]. */
__device__ __forceinline__ void fused_step(uint32_t eq, uint32_t &pv, uint32_t &mv, uint32_t &score)
{
   /* Hyyro's form of the Myers recurrence: D0 = zero-diagonal vector (12 VALU ops with 3-input bitops) */
   const uint32_t s = (eq & pv) + pv;
   const uint32_t d0 = ((s ^ pv) | eq) | mv;
   const uint32_t ph = mv | ~(d0 | pv);
   const uint32_t mh = pv & d0;
   uint32_t ph2, mh2;
   asm("v_add_co_u32 %0, vcc, %2, %2\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
       : "=v"(ph2), "+v"(score) : "v"(ph) : "vcc");
   asm("v_add_co_u32 %0, vcc, %2, %2\n\tv_subbrev_co_u32 %1, vcc, 0, %1, vcc"
       : "=v"(mh2), "+v"(score) : "v"(mh) : "vcc");
   pv = mh2 | ~(d0 | ph2);
   mv = ph2 & d0;
}

/* alphabet check of four characters: nonzero when a byte is outside {ACGTUN, acgtun, '\n'} (k_stream's table columns
 * are exact for those -- U and u share T's column, as they share its class in the reference, seeqcore.h:89-111 --; any other
 * byte aliases onto one of them).  (Until round 5 U and u counted as outside: harmless where the flag only makes the exact pass
 * look, wrong under SQ_CONVERT / SQ_IGNORE on k_stream, whose corrected copy turned them into N / a skipped byte -- found by
 * profiles/ignore_fuzz.py.) */
__device__ __forceinline__ uint32_t fused_bad4(uint32_t w)
{
   /* canonical byte of each table column (A C T G . \n . N); the text must equal it -- letters in either case,
      the newline exactly ('*' = 0x2A is '\n' with the case bit set: it must NOT pass) */
   const uint32_t idx = (w & 0x0E0E0E0Eu) >> 1;
   const uint32_t canon = __builtin_amdgcn_perm(0x4EFF0AFFu, 0x47544341u, idx);
   const uint32_t fold = __builtin_amdgcn_perm(0xDFFFFFFFu, 0xDFDEDFDFu, idx);      /* per column: case-fold mask (column 2 folds bit 0 as well: T and U) */
   return (w & fold) ^ canon;
}

/* Inclusive prefix sum over the 64 lanes of a wave with DPP row shifts / broadcasts
 * (VALU-speed, no LDS crossbar round trips). */
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x)
{
   x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);   /* row_shr:1 */
   x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);   /* row_shr:2 */
   x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);   /* row_shr:4 */
   x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);   /* row_shr:8 */
   x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, true);   /* row_bcast:15 -> rows 1,3 */
   x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, true);   /* row_bcast:31 -> rows 2,3 */
   return x;
}

/* ---- one- and two-word variants behind one interface (k_direct, k_exact1 are templated on W) ---- */
/* W = 2 serves patterns of 31..62 positions: the pattern sits in the top m bits of a 64-bit column
   (lo word = low rows), the two flag bits in bits 0-1 of the low word. */
template <int W> struct fused_eq_t;
template <> struct fused_eq_t<1> { uint32_t w0; };
template <> struct fused_eq_t<2> { uint32_t w0, w1; };

typedef __attribute__((address_space(3))) const uint32_t fused_lds_cu32;
typedef __attribute__((address_space(3))) const uint64_t fused_lds_cu64;

template <int W>
__device__ __forceinline__ fused_eq_t<W> fused_eq_load(uint32_t lds_byte_addr);
template <>
__device__ __forceinline__ fused_eq_t<1> fused_eq_load<1>(uint32_t addr)
{
   fused_eq_t<1> e;
   e.w0 = *(fused_lds_cu32 *)(uintptr_t)addr;
   return e;
}
template <>
__device__ __forceinline__ fused_eq_t<2> fused_eq_load<2>(uint32_t addr)
{
   const uint64_t v = *(fused_lds_cu64 *)(uintptr_t)addr;          /* ds_read_b64 */
   fused_eq_t<2> e;
   e.w0 = (uint32_t)v;
   e.w1 = (uint32_t)(v >> 32);
   return e;
}

template <int W> struct fused_state_t;
template <> struct fused_state_t<1> {
   uint32_t pv, mv, score;
   __device__ __forceinline__ void init(uint32_t m) { pv = 0xFFFFFFFFu; mv = 0u; score = m; }
   __device__ __forceinline__ void step(const fused_eq_t<1> &e) { fused_step(e.w0, pv, mv, score); }
};
template <> struct fused_state_t<2> {
   uint32_t pv0, pv1, mv0, mv1, score;
   __device__ __forceinline__ void init(uint32_t m) { pv0 = pv1 = 0xFFFFFFFFu; mv0 = mv1 = 0u; score = m; }
   /* the 64-bit version of fused_step: one carry chain through both words; the carry out of the
      high word of ph+ph / mh+mh is the +1 / -1 on D[m][j] */
   __device__ __forceinline__ void step(const fused_eq_t<2> &e)
   {
      const uint64_t pv = ((uint64_t)pv1 << 32) | pv0, eq = ((uint64_t)e.w1 << 32) | e.w0;
      const uint64_t s = (eq & pv) + pv;
      const uint32_t s0 = (uint32_t)s, s1 = (uint32_t)(s >> 32);
      const uint32_t d00 = ((s0 ^ pv0) | e.w0) | mv0, d01 = ((s1 ^ pv1) | e.w1) | mv1;
      const uint32_t ph0 = mv0 | ~(d00 | pv0), ph1 = mv1 | ~(d01 | pv1);
      const uint32_t mh0 = pv0 & d00, mh1 = pv1 & d01;
      uint32_t p0, p1, m0, m1;
      asm("v_add_co_u32 %0, vcc, %3, %3\n\tv_addc_co_u32 %1, vcc, %4, %4, vcc\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc"
          : "=&v"(p0), "=&v"(p1), "+v"(score) : "v"(ph0), "v"(ph1) : "vcc");
      asm("v_add_co_u32 %0, vcc, %3, %3\n\tv_addc_co_u32 %1, vcc, %4, %4, vcc\n\tv_subbrev_co_u32 %2, vcc, 0, %2, vcc"
          : "=&v"(m0), "=&v"(m1), "+v"(score) : "v"(mh0), "v"(mh1) : "vcc");
      pv0 = m0 | ~(d00 | p0); pv1 = m1 | ~(d01 | p1);
      mv0 = p0 & d00;         mv1 = p1 & d01;
   }
};

/* ---- block-level helpers (256 threads) ---- */
/* Exclusive prefix sum over the 256 threads of a block; *total = block sum. */
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *total, uint32_t *s_wave /* >= 4 */)
{
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   uint32_t x = v;
#pragma unroll
   for (int d = 1; d < 64; d <<= 1) {
      uint32_t y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
   }
   if (lane == 63) s_wave[wave] = x;
   __syncthreads();
   uint32_t base = 0, tot = 0;
#pragma unroll
   for (int w = 0; w < SEEQ_WG / 64; w++) {
      uint32_t s = s_wave[w];
      if (w < wave) base += s;
      tot += s;
   }
   __syncthreads();
   *total = tot;
   return base + x - v;
}

/* Exact per-byte "== '\n'" flags of 4 packed bytes (bit 7 of each byte). */
__device__ __forceinline__ uint32_t nl_flags(uint32_t w)
{
   const uint32_t x = w ^ 0x0A0A0A0Au;
   const uint32_t t = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;
   return ~t & 0x80808080u;
}


/* Before the EMIT pass: do the records fit? */
__device__ __forceinline__ void rec_check_body(const ScanArgs &a)
{
   Counters *c = a.cnt;
   c->need_records = c->records + c->seg_nrec;     /* running total incl. this segment */
   if (c->records + c->seg_nrec > a.cap_records) atomicOr(&c->overflow, 4u);
}

__device__ __forceinline__ void seg_end_body(const ScanArgs &a, int flags /* 1: hits come from nh[]; 2: nh[] holds 0/1 verdicts, their sum = matching lines */)
{
   Counters *c = a.cnt;
   const uint32_t counted = c->seg_nlines - c->seg_nheaders;
   const uint32_t seg_hits = (flags & 1) ? c->seg_nrec : c->seg_nhitlines;
   if (flags & 2) c->seg_nmatch = c->seg_nrec;
   c->lines += counted;
   c->headers += c->seg_nheaders;
   c->matchlines += a.use_nh >= 2 ? c->seg_nmatch : c->seg_nhitlines;   /* >= 2: the filter was a superset */
   if (a.use_nh == 3 && c->seg_nhitlines) {
      const uint32_t nhl = c->seg_nhitlines, lastl = a.hit_line[nhl - 1];
      uint32_t covered = 1u;
      if (a.ig_thr != 0u) {
         /* SQ_IGNORE on k_pair: the segment's last line may be there through a marker alone, made unseen at the segment's last tile and dropped
            by k_bounds2 (the line holds no skipped byte) -- then this segment scans nothing of it, and what the next segment finds in the line
            is the line's FIRST entry, not a repeat (found by profiles/ignore_fuzz.py: a 1 200-byte line across a seam, its hits behind it).
            Covered: one of the line's entries here is live, or the line came in as the covered line of the segment before. */
         covered = lastl == c->prev_hit_line ? 1u : 0u;
         for (uint32_t k = nhl; !covered && k-- > 0u && a.hit_line[k] == lastl; ) covered = a.hit_start[k] != 0xFFFFFFFFu ? 1u : 0u;
      }
      c->prev_hit_line = covered ? lastl : 0xFFFFFFFFu;
   }
   c->hits += seg_hits;
   if (a.want == SEEQDEV_WANT_RECORDS) c->records += seg_hits;
   c->seg_nlines = c->seg_nhitlines = c->seg_nheaders = c->seg_nrec = c->seg_nmatch = c->seg_novf = 0;
}

/* After k_stream / k_direct: reduce the per-slice partial counts (no atomics in the hot kernels)
   and publish the hit-line count of the segment, or the overflow. */
__device__ __forceinline__ void fused_post_body(const FusedArgs &a, uint32_t nslices)
{
   __shared__ uint32_t s_red[4][4];
   uint32_t lines = 0, hdrs = 0, hits = 0, mx = 0, ovf = 0, lastnl = 0, flags = 0, busy = 0, crowded = 0;
   /* one 16-byte load per slice, eight slices per thread in flight (this kernel is one workgroup on an idle chip: its time is
      the latency of its loads) */
   const uint4 *part = reinterpret_cast<const uint4 *>(a.wg_part);
   for (uint32_t i0 = threadIdx.x; i0 < nslices; i0 += 2048) {
      uint4 pv[8];
      uint32_t lv[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
         const uint32_t i = i0 + 256u * u;
         pv[u] = i < nslices ? part[i] : make_uint4(0u, 0u, 0u, 0u);
         lv[u] = a.wg_lastnl && i < nslices ? a.wg_lastnl[i] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
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


#endif
