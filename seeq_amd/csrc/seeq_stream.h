/*
 * seeq_stream.h -- k_stream: line-agnostic, table-driven scan.  The text is read ONCE, coalesced by
 * construction, and the per-character work is one LDS gather.
 *
 * k_direct gives every lane one LINE; the lane then fetches its line with strided loads, which
 * re-reads the text from L2 / HBM a second time and is what bounds those kernels.  Here a lane owns a
 * fixed CHUNK of CH consecutive bytes (a wave owns a tile of 64 * CH bytes), whatever the line structure;
 * with CH = 128 every 128-byte memory line is consumed whole by one lane (eight back-to-back 16-byte loads
 * that L1 merges; measured HBM traffic 1.02x the text):
 *
 *   - The lane walks the streaming Levenshtein automaton of the pattern (seeq_dfa.h, seeq_dfa_build_stream:
 *     the reference's DFA of saturated NW columns, libseeq.c:698-842, complete, in LDS) over its chunk,
 *     after a WARM-UP over the 4*WU >= m+tau-1 bytes before it, started from the root state.  A hit ending at
 *     text position j only depends on the m+tau bytes up to j, and the root column is the largest column
 *     there is, so at every OWNED position "D[m][j] <= tau" is decided exactly as the line-long scan would.
 *   - '\n' resets the automaton (ROOT_NL).  The first hit of a line inside a lane's window is marked by the
 *     state ACC_NEW; the lane records it in a bit mask (v_cmp + v_addc per character).  A line that spans
 *     chunks can be reported by more than one lane: duplicates are adjacent after the ordered compaction
 *     and are dropped by k_stream_bounds.
 *   - Newlines are found exactly, apart from the walk (stream_nl_mask32: per text word four flags by SWAR, dropped into
 *     the mask by v_dot4_u32_u8); the rank of a hit's line is the number of newlines before the hit, so lines are
 *     numbered without ever building a line index.
 *   - Bytes outside {A,C,G,T,N,a,c,g,t,n,'\n'} alias onto table columns; such a byte sets Counters.dirty and
 *     every reported line is then verified by the exact pass (k_exact1 COUNT), and only then.  On
 *     clean input the exact pass trusts the filter.
 *
 *   - The chunk is walked as two independent chains (bytes 0-63 warm up on the previous lane's tail, bytes 64-127 on
 *     the lane's own bytes 40-63), so two gathers are in flight per lane.
 *
 * Per tile the wave emits {tile | unresolved, rank among the tile's hits | column of the hit, start of the hit's
 * line (or the hit position when the line starts before the tile), line rank inside the tile} into its private
 * slice, and tile_cl[] = line starts owned, tile_hits[] (same layout as k_direct).  k_stream_reorder orders the
 * entries, k_stream_bounds finishes the unresolved ones and drops repeats of a line.
 * Only for line mode and automata of <= 4 000 states that warm up within 32 bytes: the pattern's complete automaton
 * (exact verdicts) or a partition filter (candidates; seeq_dfa.h).
 * What bounds it: the LDS gather unit (32 banks: 5.6 cycles per 64-lane gather), HBM hidden.
 */
#ifndef SEEQ_STREAM_H_
#define SEEQ_STREAM_H_

#define STREAM_NW 16

/* 16 bytes at an arbitrary address; bytes at or beyond `nbytes` read as '\n' (a line that the buffer
 * cuts short ends there). */
__device__ __forceinline__ fused_v4u dfa_load16(const uint8_t *text, uint64_t off, uint64_t nbytes)
{
   if (off + 16 <= nbytes) return *reinterpret_cast<const fused_v4u_unaligned *>(text + off);
   uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#pragma unroll 1
   for (int k = 15; k >= 0; k--) {
      const uint32_t b = off + (uint64_t)k < nbytes ? (uint32_t)text[off + k] : 0x0Au;
      w3 = (w3 << 8) | (w2 >> 24);
      w2 = (w2 << 8) | (w1 >> 24);
      w1 = (w1 << 8) | (w0 >> 24);
      w0 = (w0 << 8) | b;
   }
   return fused_v4u{w0, w1, w2, w3};
}

#define STREAM_OR(K) asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #K \
                         : "=v"(ad) : "v"(state), "v"(wm))
#define STREAM_HIT asm("v_cmp_eq_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(hm) : "v"(state), "v"(acc_new) : "vcc")
typedef __attribute__((address_space(3))) const uint16_t stream_lds_cu16;

/* The exact newline mask of 32 characters (eight text words; first character = bit 31), without a compare per
 * character: per word the four "is '\n'" flags as bytes of 0 / 1, then v_dot4_u32_u8 with the weights 8 4 2 1 (x 16 for
 * the first word of a pair) drops them into the mask, two words per shift.  `clean` (wave-uniform: the tile holds only
 * bytes of the alphabet, or has been corrected): every such byte but '\n' has bit 6 set, so ~bit 6 is the flag -- the
 * dot product then sums the NON-newline flags and the mask is its complement.  Otherwise the exact zero-byte test of
 * w ^ 0x0A0A0A0A.  3.5 (clean) / 6.5 VALU per word. */
__device__ __forceinline__ uint32_t stream_nl_mask32(const fused_v4u &a, const fused_v4u &b, bool clean)
{
   const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
   uint32_t nm = 0;
   if (clean) {
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
         const uint32_t p0 = (w[k] >> 6) & 0x01010101u, p1 = (w[k + 1] >> 6) & 0x01010101u;
         nm = __builtin_amdgcn_udot4(p0, 0x10204080u, __builtin_amdgcn_udot4(p1, 0x01020408u, nm << 8, false), false);
      }
      return ~nm;
   }
   /* (the flags stay at bit 7 of their bytes: the dot products carry a factor 128, taken out per 16 characters) */
   uint32_t h[2];
#pragma unroll
   for (int g = 0; g < 2; g++) {
      nm = 0;
#pragma unroll
      for (int k = 4 * g; k < 4 * g + 4; k += 2) {
         const uint32_t x0 = w[k] ^ 0x0A0A0A0Au, x1 = w[k + 1] ^ 0x0A0A0A0Au;
         const uint32_t f0 = ~(((x0 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x0) & 0x80808080u, f1 = ~(((x1 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x1) & 0x80808080u;
         nm = __builtin_amdgcn_udot4(f0, 0x10204080u, __builtin_amdgcn_udot4(f1, 0x01020408u, nm << 8, false), false);
      }
      h[g] = nm;
   }
   return (h[0] << 9) | (h[1] >> 7);
}

/* Two independent walks interleaved (chains A and B of one lane): twice the gathers in flight per wave. */
#define STREAM_X2(K) \
   asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #K : "=v"(ada) : "v"(sa), "v"(wma)); \
   asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #K : "=v"(adb) : "v"(sb), "v"(wmb)); \
   sa = *(stream_lds_cu16 *)(uintptr_t)ada; sb = *(stream_lds_cu16 *)(uintptr_t)adb;
#define STREAM_EV2(K) \
   asm("v_cmp_eq_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(hma) : "v"(sa), "v"(acc_new) : "vcc"); \
   asm("v_cmp_eq_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(hmb) : "v"(sb), "v"(acc_new) : "vcc");

__device__ __forceinline__ void stream_warm4x2(uint32_t &sa, uint32_t wa, uint32_t &sb, uint32_t wb)
{
   const uint32_t wma = wa & 0x0E0E0E0Eu, wmb = wb & 0x0E0E0E0Eu;
   uint32_t ada, adb;
   STREAM_X2(0) STREAM_X2(1) STREAM_X2(2) STREAM_X2(3)
}

/* the same, remembering whether a chain accepted (restart tables: a chain that restarted inside its warm-up window is out of step with the
   walk the line's start would have made -- it names its own first byte a candidate, as k_pair's chains do) */
__device__ __forceinline__ void stream_warm4x2_seen(uint32_t &sa, uint32_t wa, uint32_t &sb, uint32_t wb, uint32_t acc_new, uint32_t &seena, uint32_t &seenb)
{
   const uint32_t wma = wa & 0x0E0E0E0Eu, wmb = wb & 0x0E0E0E0Eu;
   uint32_t ada, adb;
   STREAM_X2(0) seena |= sa == acc_new ? 1u : 0u; seenb |= sb == acc_new ? 1u : 0u;
   STREAM_X2(1) seena |= sa == acc_new ? 1u : 0u; seenb |= sb == acc_new ? 1u : 0u;
   STREAM_X2(2) seena |= sa == acc_new ? 1u : 0u; seenb |= sb == acc_new ? 1u : 0u;
   STREAM_X2(3) seena |= sa == acc_new ? 1u : 0u; seenb |= sb == acc_new ? 1u : 0u;
}

__device__ __forceinline__ void stream_own4x2(uint32_t &sa, uint32_t wa, uint32_t &hma, uint32_t &sb, uint32_t wb, uint32_t &hmb, uint32_t acc_new)
{
   const uint32_t wma = wa & 0x0E0E0E0Eu, wmb = wb & 0x0E0E0E0Eu;
   uint32_t ada, adb;
   STREAM_X2(0) STREAM_EV2(0) STREAM_X2(1) STREAM_EV2(1) STREAM_X2(2) STREAM_EV2(2) STREAM_X2(3) STREAM_EV2(3)
}

/* MY (k_stream's Myers mode, below): four text bytes through the bit-vector column instead of the table.  EQ entries: the
 * top-aligned Peq word(s) of the byte's class, or a flag in bits 0-1: 1 = the byte ends the line (NUL, a non-DNA byte under
 * SQ_FAIL), 3 = newline.  The column is stepped in place whatever the byte (after a flagged byte it is garbage until the
 * next newline re-initialises it, and `dead` keeps hits out until then); `seen` makes only the FIRST position with
 * D[m][j] <= tau of a line report (what ACC_NEW / ACC_OLD do for the table walk).  OWN: record into the mask. */
template <int W>
__device__ __forceinline__ void stream_myers_lookup4(uint32_t w, uint32_t eqb, fused_eq_t<W> (&ev)[4])
{
#pragma unroll
   for (int cc = 0; cc < 4; cc++) ev[cc] = fused_eq_load<W>(eqb + (((w >> (8 * cc)) & 0xFFu) << (W == 1 ? 2 : 3)));
}

template <int W, bool OWN>
__device__ __forceinline__ void stream_myers_step4(const fused_eq_t<W> (&ev)[4], fused_state_t<W> &st, uint32_t m, uint32_t tau,
                                                   uint32_t &dead, uint32_t &seen, uint32_t &hm)
{
#pragma unroll
   for (int cc = 0; cc < 4; cc++) {
      const uint32_t fl = ev[cc].w0 & 3u;
      st.step(ev[cc]);
      const bool nl = fl == 3u;
      fused_state_t<W> fresh;
      fresh.init(m);
      exact1_take<W>(st, fresh, nl);
      dead = nl ? 0u : (dead | (fl == 1u ? 1u : 0u));
      const uint32_t hit = (fl == 0u && st.score <= tau && dead == 0u) ? 1u : 0u;
      if (OWN) hm = (hm << 1) | (hit & (seen ^ 1u));
      seen = nl ? 0u : (seen | hit);
   }
}

/* Round 5: the same four bytes where NO lane of the wave holds a flagged byte (no newline, no terminator: on chromosome-long lines that is
 * nearly every word) -- the column steps and ONE instruction per byte beside them: v_alignbit shifts the sign of tau - score into a stream
 * (k_verify's trick, seeq_verify.h); the dead / first-hit-of-the-line logic runs once per word on the four bits.  18.75 VALU per text byte
 * instead of 32 (PMC, profiles/r05/pmc_myers.txt: the per-byte flag logic was 17 of the 32, and the kernel issued VALU instructions in 76 %
 * of its 4-cycle slots -- it was not short of issue rate but long on instructions).  `dead` cannot change here (no flag), `seen` only grows. */
template <int W, bool OWN>
__device__ __forceinline__ void stream_myers_fast4(const fused_eq_t<W> (&ev)[4], fused_state_t<W> &st, uint32_t tau, uint32_t dead, uint32_t &seen, uint32_t &hm)
{
   uint32_t L = 0;
#pragma unroll
   for (int cc = 0; cc < 4; cc++) {
      st.step(ev[cc]);
      if (OWN) L = __builtin_amdgcn_alignbit(L, tau - st.score, 31);      /* bit = 1: score > tau; first byte ends up in bit 3 */
   }
   if (OWN) {
      uint32_t r4 = ~L & 0xFu;                                         /* scores <= tau */
      r4 = (dead | seen) ? 0u : r4;                                    /* only the FIRST such position of a line reports, and none behind a byte that ended it */
      const uint32_t first = r4 ? 1u << (31u - (uint32_t)__builtin_clz(r4)) : 0u;
      hm = (hm << 4) | first;
      seen |= r4 ? 1u : 0u;
   }
}

/* word k (0..31) of a lane's 128-byte chunk */
__device__ __forceinline__ uint32_t stream_word32(const fused_v4u (&v)[8], int k)
{
   const fused_v4u &q = v[k >> 2];
   return (k & 3) == 0 ? q.x : (k & 3) == 1 ? q.y : (k & 3) == 2 ? q.z : q.w;
}

/* SQ_CONVERT (reference libseeq.c:223-228: a byte that is not A C G T N or a terminator counts as 'N'): four text bytes
 * with every byte outside the alphabet replaced by 'N' -- and NUL, which ends the line in every mode (seeqcore.h:89-111,
 * libseeq.c:267-270), by a byte of the DEAD column -- so that the walk over them is exact.
 * SQ_IGNORE (libseeq.c:265-266: such a byte is skipped, but counted in coordinates): replaced by 'H', a byte of column 4,
 * which the skip variant of the table (seeq_dfa_skip_variant) maps every state onto itself with. */
template <int SUB = 1>
__device__ __forceinline__ uint32_t stream_sub4(uint32_t w, uint32_t &nuls)
{
   /* opaque: the tile-wide alphabet check has computed fused_bad4 of the same word; without this the compiler keeps all
      its intermediates (three values per word, 32 words) in scratch on EVERY tile to reuse them here */
   asm volatile("" : "+v"(w));
   const uint32_t bad = fused_bad4(w);
   if (bad == 0) return w;
   uint32_t f = (((bad & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | bad) & 0x80808080u;      /* 0x80 per byte outside the alphabet */
   uint32_t z = ~(((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w) & 0x80808080u;         /* 0x80 per NUL byte */
   nuls |= z;
   f = f | (f - (f >> 7));                                                      /* -> 0xFF */
   z = z | (z - (z >> 7));
   w = (w & ~f) | ((SUB == 2 ? 0x48484848u : 0x4E4E4E4Eu) & f);                 /* 'N' / 'H' (column 4: skip) */
   return (w & ~z) | (0x4C4C4C4Cu & z);                                         /* column 6: DEAD until the next newline */
}

/* the value of `x` in the previous lane; lane 0 gets `first` (DPP wave_shr:1) */
__device__ __forceinline__ uint32_t stream_from_prev_lane(uint32_t x, uint32_t first)
{
   return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)x, 0x138, 0xf, 0xf, false);
}

/* inclusive prefix maximum over the 64 lanes of a wave (DPP; lanes shifted in from outside read 0) */
__device__ __forceinline__ uint32_t wave_incl_max_u32(uint32_t x)
{
   uint32_t y;
   y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true); x = y > x ? y : x;   /* row_shr:1 */
   y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true); x = y > x ? y : x;   /* row_shr:2 */
   y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true); x = y > x ? y : x;   /* row_shr:4 */
   y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true); x = y > x ? y : x;   /* row_shr:8 */
   y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, true); x = y > x ? y : x;   /* row_bcast:15 */
   y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, true); x = y > x ? y : x;   /* row_bcast:31 */
   return x;
}

/* FA: FASTA input (SEEQDEV_FASTA): a line that starts with '>' is a header -- not counted, never a hit line
 * (reference seeq.c:367-374).  Headers are found per NEWLINE (is the byte after it a '>'?), so the walk itself is
 * unchanged; candidates inside a header are discarded by the exact pass, which looks at the first byte of the line. */
/* LL: long-line mode -- per-tile "dirty" flags and the segment's last newline for the window walk of the exact pass
 * (kept out of the read-length kernel: its tile loop has no register to spare). */
/* SUB = 1 (SQ_CONVERT): tiles that hold such bytes are walked over a corrected copy (stream_sub4) held in registers: the
 * verdicts are exact on any text and nothing needs re-running.
 * A NUL ends its line for the chain that meets it, but not for the chains behind it on the same line: a tile with a NUL
 * flags the scan as a superset (wg_part flag 4) and the exact pass verifies the candidates.
 * SUB = 2 (SQ_IGNORE, read-length lines): the same with skip bytes.  Skipped bytes stretch the text a match spans, so a
 * chain whose warm-up window holds one has not seen enough of the line when its own bytes begin: it reports the line its
 * first byte lies in as a candidate whatever the walk says (a made-up first hit), the wave flags the scan as a superset
 * (wg_part flag 4) and the exact pass, which knows how to skip, verifies the candidates. */
/* MY = 1 / 2 (words of the bit-vector column): the MYERS MODE -- patterns of <= 62 positions that have no automaton small
 * enough for LDS, on text whose lines are too long for the per-line kernels (a chromosome per line: the reference's own
 * benchmark, doc/response.tex:181-232, m = 27 .. 42 with up to 15 errors).  Same frame, same bookkeeping, same exact
 * verdicts on clean text; the walk is the Myers/Hyyro column itself, one chain per lane warmed up over the 4 * WU >= m + tau
 * - 1 bytes before it (WU = 16 or 32 words, all from the previous lane through DPP).  ~25 (W = 1) / ~35 (W = 2) VALU
 * instructions per text byte and no table: VALU-bound at ~1 TB/s -- where the generic path, one LINE per lane, took 45 to
 * 85 s for the 3.2 GB of that benchmark, 7 times the reference's own time on one core. */
template <int WU, bool FA, bool LL, int SUB = 0, int MY = 0>
__global__ __launch_bounds__(64 * STREAM_NW, MY ? 4 : 8) void k_stream(FusedArgs a)
{
   constexpr int NW = STREAM_NW;
   constexpr int CH = 128;                                /* bytes per lane: one 128-byte memory line, consumed whole by its lane */
   constexpr int NQ = CH / 16;                            /* 16-byte pieces per lane */
   constexpr int NM = CH / 32;                            /* mask registers per lane */
   constexpr uint32_t TB = 64u * CH;                      /* tile bytes */
   static_assert(MY ? (WU == 16 || WU == 32) : (WU == 4 || WU == 6 || WU == 8), "warm-up is 16, 24 or 32 bytes (Myers mode: 64 or 128)");
   static_assert(!MY || SUB == 0, "the Myers mode steps every byte through its own EQ table");
   extern __shared__ __align__(16) uint8_t dsmem[];
   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   {
      const fused_v4u *src = reinterpret_cast<const fused_v4u *>(a.dfa);
      for (uint32_t i = tid; i < a.dfa_rows; i += 64 * NW) reinterpret_cast<fused_v4u *>(dsmem)[i] = src[i];
   }
   __syncthreads();                                       /* the only barrier: the table is read-only from here */
   const uint32_t acc_new = a.dfa_final_base;              /* state value of ACC_NEW (seeq_dfa_build_stream) */

   const uint32_t gwave = blockIdx.x * NW + wave, nwaves = gridDim.x * NW;
   uint32_t wv_lines = 0, wv_hitlines = 0, wv_hdrs = 0, slice_pos = 0, wv_lastnl = 0;    /* wave-uniform */
   bool wv_overflow = false;
   uint32_t wv_dirty = 0, wv_fakes = 0;
   uint4 *slice = a.tmp + (size_t)gwave * a.slice_cap;
   const uint64_t lim = a.seg_base + a.seg_len;           /* bytes at or beyond it are not this segment's */
   const uint64_t last = a.nbytes - 1;

   /* persistent grid: wave w of the grid takes tiles w, w + waves, ... */
   for (uint32_t tile = gwave; tile < a.ntiles; tile += nwaves) {
      const uint64_t t0 = a.seg_base + (uint64_t)tile * TB;
      /* opaque per tile: keeps the compiler from hoisting the per-lane 64-bit addresses of the guarded loads
         out of the tile loop (that costs ~20 VGPRs and spills) */
      uint32_t lane_off = (uint32_t)lane * CH;
      asm volatile("" : "+v"(lane_off));
      const uint64_t my = t0 + lane_off;
      /* wave-uniform; the segment's last tile, cut short (32-bit scalar compares: a 64-bit one would go through VGPRs) */
      const bool partial = tile + 1 == a.ntiles && (a.seg_len % TB) != 0;
      fused_v4u v[NQ];
      if (!partial) {
         const uint8_t *p = a.text + my;
#pragma unroll
         for (int q = 0; q < NQ; q++)
            v[q] = *reinterpret_cast<const fused_v4u_unaligned *>(p + 16 * q);   /* (nt loads: the eight loads of a 128-B line no longer merge in L1 -- 2x slower;
                                                                                     a uniform base + per-load 32-bit offsets instead of one 64-bit address with immediates: 20 % slower) */
      } else {
#pragma unroll
         for (int q = 0; q < NQ; q++) v[q] = dfa_load16(a.text, my + 16 * q, lim);       /* '\n' beyond the segment */
      }
      /* the 32 bytes before the tile (lane 0's warm-up); '\n' when the buffer starts here */
      fused_v4u pa = fused_v4u{0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au}, pb = pa;
      if (!MY && t0 >= 32) {
         pa = *reinterpret_cast<const fused_v4u_unaligned *>(a.text + t0 - 32);
         pb = *reinterpret_cast<const fused_v4u_unaligned *>(a.text + t0 - 16);
      }
      /* ---- alphabet check, done with before the walk starts (nothing of it stays live) ---- */
      uint32_t fake = 0;                                  /* SUB == 2: bit 0 / 1 = chain A / B reports its first byte's line unseen */
      bool tile_clean = false;                            /* wave-uniform: only alphabet bytes in my tile (or corrected): the cheap newline test holds */
      {
         uint32_t bad = 0, bad_mid = 0, bad_tail = 0;     /* (SUB == 2: bytes 32..63 and 96..127 apart -- the warm-up windows of chain B and of the next lane's chain A) */
#pragma unroll
         for (int q = 0; q < NQ; q++) {
            const uint32_t b = fused_bad4(v[q].x) | fused_bad4(v[q].y) | fused_bad4(v[q].z) | fused_bad4(v[q].w);
            if (SUB == 2 && (q == NQ / 2 - 2 || q == NQ / 2 - 1)) bad_mid |= b;
            else if (SUB == 2 && q >= NQ - 2) bad_tail |= b;
            else bad |= b;
         }
         bad |= bad_mid | bad_tail;
         /* a byte outside the alphabet anywhere in the tile: the scan's verdicts need verifying */
         uint64_t badlanes = __ballot(bad != 0);
         uint32_t flag = (uint32_t)__builtin_amdgcn_readfirstlane(badlanes != 0 ? 1 : 0);
         if (SUB) {
            /* the 32 bytes before the tile belong to another tile: looked at on their own (the same for all lanes) */
            const uint32_t pbad = fused_bad4(pa.x) | fused_bad4(pa.y) | fused_bad4(pa.z) | fused_bad4(pa.w) |
                                  fused_bad4(pb.x) | fused_bad4(pb.y) | fused_bad4(pb.z) | fused_bad4(pb.w);
            const bool pre = __builtin_amdgcn_readfirstlane(pbad != 0 ? 1 : 0) != 0;
            if (SUB == 2 && (flag || pre)) {              /* (wave-uniform) whose warm-up window holds a byte that will be skipped? */
               fake = (stream_from_prev_lane(bad_tail, pbad) ? 1u : 0u) | (bad_mid ? 2u : 0u);
               const uint32_t nfake = (uint32_t)__popcll(__ballot((fake & 1u) != 0)) + (uint32_t)__popcll(__ballot((fake & 2u) != 0));
               if (nfake) wv_dirty |= 4u;                 /* the hit lines of this scan are a superset: the exact pass decides */
            }
            uint32_t nuls = 0, pnuls = 0;
            if (flag) {                                   /* wave-uniform */
#pragma unroll
               for (int q = 0; q < NQ; q++) {
                  v[q].x = stream_sub4<SUB>(v[q].x, nuls); v[q].y = stream_sub4<SUB>(v[q].y, nuls);
                  v[q].z = stream_sub4<SUB>(v[q].z, nuls); v[q].w = stream_sub4<SUB>(v[q].w, nuls);
               }
            }
            if (pre) {
               pa.x = stream_sub4<SUB>(pa.x, pnuls); pa.y = stream_sub4<SUB>(pa.y, pnuls); pa.z = stream_sub4<SUB>(pa.z, pnuls); pa.w = stream_sub4<SUB>(pa.w, pnuls);
               pb.x = stream_sub4<SUB>(pb.x, pnuls); pb.y = stream_sub4<SUB>(pb.y, pnuls); pb.z = stream_sub4<SUB>(pb.z, pnuls); pb.w = stream_sub4<SUB>(pb.w, pnuls);
            }
            /* handled: only a newline or a NUL ends a line now.  A NUL does so for the chain that meets it, not for the chains
               behind it on the same line: the scan's hit lines become candidates, and for the window walk of the exact pass
               (LL) the lanes that hold one count as not clean */
            flag = 0;
            badlanes = __ballot(nuls != 0);
            if (badlanes | __ballot(pnuls != 0)) wv_dirty |= 4u;
         }
         asm volatile("" : "+s"(flag));                   /* pinned here: the walk below needs the registers */
         wv_dirty |= flag;
         tile_clean = flag == 0;
         if (LL && lane == 0) { a.tile_dirty[tile] = (flag || badlanes) ? 1u : 0u; a.tile_dmask[tile] = badlanes; }
      }
      uint32_t hmask[NM], nmask[NM];
      uint32_t blind = 0;                                 /* LL, filter: bit 0 / 1 = chain A / B entered its bytes in the accepting state */
      if (MY) {
         /* ---- Myers mode: one chain per lane, warmed up over the previous lane's last WU words (lane 0: the words before
                 the tile -- one per lane 0..31 in `hal`, '\n' where the buffer starts) ---- */
         const uint32_t eqb = (uint32_t)(uintptr_t)(fused_lds_cu32 *)(const uint32_t *)dsmem;
         uint32_t hal = 0x0A0A0A0Au;
         if (t0 >= 128) hal = *reinterpret_cast<const uint32_t __attribute__((aligned(1))) *>(a.text + t0 - 128 + 4 * (lane & 31));
         constexpr int MW = MY ? MY : 1;
         fused_state_t<MW> st;
         st.init((uint32_t)a.m);
         uint32_t dead = 0, seen = 0, hm = 0;
         /* WU + 32 words, the EQ lookups of a word issued one word ahead of its column steps and no further: left alone the
            compiler hoists the lookups of ALL words to the top (every one of them independent of the column) and spills
            their 250 results.  The empty asm ties the next word to the state after the previous word's steps. */
         fused_eq_t<MW> ev[2][4];
         {
            uint32_t w0 = stream_from_prev_lane(stream_word32(v, 32 - WU), (uint32_t)__builtin_amdgcn_readlane((int)hal, 32 - WU));
            stream_myers_lookup4<MW>(w0, eqb, ev[0]);
         }
         /* Round 5: the column is exact m + tau - 1 bytes behind ANY start, so the warm-up begins that far back and no further: the words of the
            instance's window (64 / 128 bytes) in front of it keep their look-ups (the pipeline below stays one piece of straight code) and skip
            their column steps -- a scalar branch per warm-up word.  m = 20, tau = 4: 6 words of 16, 38 of the chain's 48 words stepped. */
         const int wu_skip = __builtin_amdgcn_readfirstlane(WU - (int)(((uint32_t)a.m + (uint32_t)a.tau - 1u + 3u) >> 2));
#pragma unroll
         for (int k = 0; k < WU + 32; k++) {
            if (k + 1 < WU + 32) {
               const int kk = k + 1 - WU;                   /* word index in my chunk; < 0: the previous lane's */
               uint32_t wn = kk < 0 ? stream_from_prev_lane(stream_word32(v, 32 + kk), (uint32_t)__builtin_amdgcn_readlane((int)hal, 32 + kk))
                                    : stream_word32(v, kk);
               asm volatile("" : "+v"(wn) : "v"(seen), "v"(st.score));
               stream_myers_lookup4<MW>(wn, eqb, ev[(k + 1) & 1]);
            }
            if (k == WU) { seen = 0; hm = 0; }              /* every chain reports the first hit of a line inside its OWN bytes */
            /* a flagged byte (newline, terminator) in this word of ANY lane: the per-byte logic; else the lean steps (wave-uniform branch) */
            const fused_eq_t<MW> (&e4)[4] = ev[k & 1];
            if (k < WU && k < wu_skip) continue;            /* (wave-uniform) in front of the warm-up this pattern needs */
            if (__any(((e4[0].w0 | e4[1].w0 | e4[2].w0 | e4[3].w0) & 3u) != 0u)) {
               if (k < WU) stream_myers_step4<MW, false>(e4, st, (uint32_t)a.m, (uint32_t)a.tau, dead, seen, hm);
               else stream_myers_step4<MW, true>(e4, st, (uint32_t)a.m, (uint32_t)a.tau, dead, seen, hm);
            } else {
               if (k < WU) stream_myers_fast4<MW, false>(e4, st, (uint32_t)a.tau, dead, seen, hm);
               else stream_myers_fast4<MW, true>(e4, st, (uint32_t)a.tau, dead, seen, hm);
            }
            if (k >= WU && ((k - WU) & 7) == 7) { hmask[(k - WU) >> 3] = hm; hm = 0; }
         }
      } else {
         /* Two chains per lane: A = bytes 0..63 (warm-up: the previous lane's last bytes), B = bytes 64..127
            (warm-up: my own bytes before 64).  Same result, 16 % more gathers, but two of them in flight. */
         uint32_t sa = 0, sb = 0;
         const bool restart = LL && a.ll_filter == 2u;     /* (wave-uniform) the filter's restart table: every part occurrence is flagged */
         if (restart) {
            uint32_t seena = 0, seenb = 0;
            if (WU == 8) {
               stream_warm4x2_seen(sa, stream_from_prev_lane(v[NQ - 2].x, pa.x), sb, v[NQ / 2 - 2].x, acc_new, seena, seenb);
               stream_warm4x2_seen(sa, stream_from_prev_lane(v[NQ - 2].y, pa.y), sb, v[NQ / 2 - 2].y, acc_new, seena, seenb);
            }
            if (WU >= 6) {
               stream_warm4x2_seen(sa, stream_from_prev_lane(v[NQ - 2].z, pa.z), sb, v[NQ / 2 - 2].z, acc_new, seena, seenb);
               stream_warm4x2_seen(sa, stream_from_prev_lane(v[NQ - 2].w, pa.w), sb, v[NQ / 2 - 2].w, acc_new, seena, seenb);
            }
            stream_warm4x2_seen(sa, stream_from_prev_lane(v[NQ - 1].x, pb.x), sb, v[NQ / 2 - 1].x, acc_new, seena, seenb);
            stream_warm4x2_seen(sa, stream_from_prev_lane(v[NQ - 1].y, pb.y), sb, v[NQ / 2 - 1].y, acc_new, seena, seenb);
            stream_warm4x2_seen(sa, stream_from_prev_lane(v[NQ - 1].z, pb.z), sb, v[NQ / 2 - 1].z, acc_new, seena, seenb);
            stream_warm4x2_seen(sa, stream_from_prev_lane(v[NQ - 1].w, pb.w), sb, v[NQ / 2 - 1].w, acc_new, seena, seenb);
            blind = seena | (seenb << 1);
         } else {
         if (WU == 8) {
            stream_warm4x2(sa, stream_from_prev_lane(v[NQ - 2].x, pa.x), sb, v[NQ / 2 - 2].x);
            stream_warm4x2(sa, stream_from_prev_lane(v[NQ - 2].y, pa.y), sb, v[NQ / 2 - 2].y);
         }
         if (WU >= 6) {
            stream_warm4x2(sa, stream_from_prev_lane(v[NQ - 2].z, pa.z), sb, v[NQ / 2 - 2].z);
            stream_warm4x2(sa, stream_from_prev_lane(v[NQ - 2].w, pa.w), sb, v[NQ / 2 - 2].w);
         }
         stream_warm4x2(sa, stream_from_prev_lane(v[NQ - 1].x, pb.x), sb, v[NQ / 2 - 1].x);
         stream_warm4x2(sa, stream_from_prev_lane(v[NQ - 1].y, pb.y), sb, v[NQ / 2 - 1].y);
         stream_warm4x2(sa, stream_from_prev_lane(v[NQ - 1].z, pb.z), sb, v[NQ / 2 - 1].z);
         stream_warm4x2(sa, stream_from_prev_lane(v[NQ - 1].w, pb.w), sb, v[NQ / 2 - 1].w);
         /* Long lines behind a partition FILTER (round 4): a chain that enters its own bytes in the accepting state -- a part occurrence
            ended inside its warm-up window -- stays there until a newline that may be megabytes away and reports nothing.  Behind the
            complete automaton the exact pass notices (a score <= tau in the columns before the chunk: it then scans the chunk whole);
            a part occurrence it cannot see -- so the blind chain names its own first byte a candidate and its chunk is scanned. */
         if (LL && a.ll_filter) blind = ((sa == 16u || sa == acc_new) ? 1u : 0u) | ((sb == 16u || sb == acc_new) ? 2u : 0u);
         }
#pragma unroll
         for (int r = 0; r < NM / 2; r++) {
            uint32_t hma = 0, hmb = 0;
#pragma unroll
            for (int q = 2 * r; q < 2 * r + 2; q++) {
               stream_own4x2(sa, v[q].x, hma, sb, v[q + NQ / 2].x, hmb, acc_new);
               stream_own4x2(sa, v[q].y, hma, sb, v[q + NQ / 2].y, hmb, acc_new);
               stream_own4x2(sa, v[q].z, hma, sb, v[q + NQ / 2].z, hmb, acc_new);
               stream_own4x2(sa, v[q].w, hma, sb, v[q + NQ / 2].w, hmb, acc_new);
            }
            hmask[r] = hma; hmask[r + NM / 2] = hmb;
         }
      }
      /* ---- newline masks, apart from the walk: SWAR flags + dot products (stream_nl_mask32) ---- */
#pragma unroll
      for (int r = 0; r < NM; r++) nmask[r] = stream_nl_mask32(v[2 * r], v[2 * r + 1], tile_clean);
      if (LL && !MY) {
         if (blind & 1u) hmask[0] |= 0x80000000u;
         if (blind & 2u) hmask[NM / 2] |= 0x80000000u;
      }
      if (SUB == 2) {                /* made-up first hits at the first byte of a chain (see SUB) */
         if (a.skip_thr && __ballot(fake != 0)) {
            /* Round 4: most made-up candidates can be ruled out by COUNTING.  An occurrence takes at least m - tau characters that
               are not skipped, all between its line's start and its end -- so a hit this chain's walk may have missed (it ends inside
               the chain) needs that many between the last newline before the chain and the chain's end.  Per chain: its characters
               that are not skipped (the bytes stream_sub4 turned into 'H' / 'L' and the newlines taken off); per lane {a newline in
               it, the characters behind its last newline}; a segmented scan
               over the lanes gives every lane the count since the last newline before it (unknown when the line starts before the
               tile: the candidate stays).  FASTQ quality lines hold ~10 such characters in 150: their chains stay silent. */
            /* rm[g]: the characters of 32-byte group g that are neither skipped nor a newline (first character = bit 31) */
            uint32_t rm[NM], pc[NM], smk[NM];
#pragma unroll
            for (int g = 0; g < NM; g++) {
               const uint32_t w[8] = {v[2 * g].x, v[2 * g].y, v[2 * g].z, v[2 * g].w, v[2 * g + 1].x, v[2 * g + 1].y, v[2 * g + 1].z, v[2 * g + 1].w};
               uint32_t sm = 0;
#pragma unroll
               for (int k = 0; k < 8; k += 2) {
                  const uint32_t y0 = (w[k] ^ 0x48484848u) & 0xFBFBFBFBu, y1 = (w[k + 1] ^ 0x48484848u) & 0xFBFBFBFBu;      /* zero byte: 'H' (skipped) or 'L' (dead) */
                  const uint32_t f0 = ~(((y0 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y0) & 0x80808080u, f1 = ~(((y1 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y1) & 0x80808080u;
                  sm = __builtin_amdgcn_udot4(f0 >> 7, 0x10204080u, __builtin_amdgcn_udot4(f1 >> 7, 0x01020408u, sm << 8, false), false);
               }
               smk[g] = sm;
               rm[g] = ~(sm | nmask[g]);
               pc[g] = (uint32_t)__popc(rm[g]);
            }
            const uint32_t rc[2] = {pc[0] + pc[1], pc[2] + pc[3]};
            static_assert(NM == 4, "two groups per chain");
            /* (1) a newline in the warm-up window resets the walk: only bytes skipped BEHIND the window's last newline shorten it.
               Window of chain B: my bytes 32..63 (group 1); of chain A: the previous lane's bytes 96..127 (its group 3; lane 0 keeps
               the verdict made from the bytes before the tile).  Bits below a mask's lowest one = the characters behind its last newline
               (no newline: all of them). */
#define STREAM_BEHIND_MASK(g) ((nmask[g] & (0u - nmask[g])) - 1u)
            {
               const uint32_t winB = smk[1] & STREAM_BEHIND_MASK(1), winT = smk[3] & STREAM_BEHIND_MASK(3);
               const uint32_t keepA = stream_from_prev_lane(winT ? 1u : 0u, 1u);
               fake &= keepA | (winB ? 2u : 0u);
            }
            /* (2) the count: characters that are not skipped since the last newline before the chain + those of the chain in front of
               ITS first newline (the line the made-up candidate would name ends there) */
#define STREAM_BEHIND(g) ((uint32_t)__popc(rm[g] & STREAM_BEHIND_MASK(g)))
#define STREAM_AHEAD(g) ((uint32_t)__popc(rm[g] & ~(0xFFFFFFFFu >> (uint32_t)__builtin_clz(nmask[g]))))      /* (nmask[g] != 0) in front of the group's FIRST newline */
            const uint32_t nlA = (nmask[0] | nmask[1]) ? 1u : 0u, nlB = (nmask[2] | nmask[3]) ? 1u : 0u;
            const uint32_t tailA = nmask[1] ? STREAM_BEHIND(1) : STREAM_BEHIND(0) + pc[1];                    /* (used when nlA) */
            const uint32_t tailB = nmask[3] ? STREAM_BEHIND(3) : STREAM_BEHIND(2) + pc[3];                    /* (used when nlB) */
            const uint32_t headA = nmask[0] ? STREAM_AHEAD(0) : pc[0] + (nmask[1] ? STREAM_AHEAD(1) : pc[1]);
            const uint32_t headB = nmask[2] ? STREAM_AHEAD(2) : pc[2] + (nmask[3] ? STREAM_AHEAD(3) : pc[3]);
#undef STREAM_BEHIND
#undef STREAM_AHEAD
#undef STREAM_BEHIND_MASK
            /* {bit 31: a newline in the lane, low bits: characters since its last one}; combine(before, cur) = cur.flag ? cur : before + cur */
            uint32_t P = ((nlA | nlB) << 31) | (nlB ? tailB : nlA ? tailA + rc[1] : rc[0] + rc[1]);
#define STREAM_SEG(ctrl, rmask) { const uint32_t y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, ctrl, rmask, 0xf, true); P = (P >> 31) ? P : y + P; }
            STREAM_SEG(0x111, 0xf) STREAM_SEG(0x112, 0xf) STREAM_SEG(0x114, 0xf) STREAM_SEG(0x118, 0xf) STREAM_SEG(0x142, 0xa) STREAM_SEG(0x143, 0xc)
#undef STREAM_SEG
            const uint32_t before = stream_from_prev_lane(P, 0u);                               /* lane 0: unknown */
            const uint32_t knownA = before >> 31, cntA = before & 0x7FFFFFFFu;
            if ((fake & 1u) && knownA && cntA + headA < a.skip_thr) fake &= ~1u;
            const uint32_t knownB = knownA | nlA, cntB = nlA ? tailA : cntA + rc[0];
            if ((fake & 2u) && knownB && cntB + headB < a.skip_thr) fake &= ~2u;
         }
         wv_fakes += (uint32_t)__popcll(__ballot((fake & 1u) != 0)) + (uint32_t)__popcll(__ballot((fake & 2u) != 0));
         if (fake & 1u) hmask[0] |= 0x80000000u;
         if (fake & 2u) hmask[NM / 2] |= 0x80000000u;
      }
      /* ---- bookkeeping: what the tile owns ---- */
      if (partial) {                                      /* filler bytes are nobody's newlines */
         const uint32_t valid = lim > my ? (lim - my < CH ? (uint32_t)(lim - my) : (uint32_t)CH) : 0u;
#pragma unroll
         for (int r = 0; r < NM; r++) {
            const uint32_t lo = 32u * r;
            const uint32_t keep = valid <= lo ? 0u : (valid >= lo + 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> (valid - lo)));
            nmask[r] &= keep; hmask[r] &= keep;
         }
      }
      if (t0 <= last && last < t0 + TB) {                 /* a newline in the very last byte starts no line */
         const uint32_t o = (uint32_t)(last - t0);
         if ((uint32_t)lane == o / CH) {
            const uint32_t pos = o % CH;
#pragma unroll
            for (int r = 0; r < NM; r++)
               if ((pos >> 5) == (uint32_t)r) nmask[r] &= ~(0x80000000u >> (pos & 31));
         }
      }
      uint32_t lane_hits = 0, lane_nl = 0;
#pragma unroll
      for (int r = 0; r < NM; r++) { lane_hits += (uint32_t)__popc(hmask[r]); lane_nl += (uint32_t)__popc(nmask[r]); }
      const uint32_t incl_h = wave_incl_scan_u32(lane_hits), incl_n = wave_incl_scan_u32(lane_nl);
      const uint32_t tot_h = (uint32_t)__builtin_amdgcn_readlane((int)incl_h, 63);
      const uint32_t tot_n = (uint32_t)__builtin_amdgcn_readlane((int)incl_n, 63);
      const uint32_t extra = (uint32_t)__builtin_amdgcn_readfirstlane((a.first_seg && tile == 0) ? 1 : 0);   /* the line starting at byte 0 */
      /* FASTA: which of my newlines start a header line?  (few newlines per lane: one byte load each) */
      uint32_t dmask[NM], lane_hd = 0, excl_d = 0, tot_d = 0, hd_extra = 0;
#pragma unroll
      for (int r = 0; r < NM; r++) dmask[r] = 0;
      if (FA) {
#pragma unroll
         for (int r = 0; r < NM; r++) {
            uint32_t mm = nmask[r];
            while (mm) {
               const uint32_t lz = (uint32_t)__builtin_clz(mm);
               mm &= ~(0x80000000u >> lz);
               const uint64_t nxt = my + 32u * r + lz + 1;             /* < nbytes: a newline in the last byte was dropped */
               if (a.text[nxt] == '>') dmask[r] |= 0x80000000u >> lz;
            }
            lane_hd += (uint32_t)__popc(dmask[r]);
         }
         const uint32_t incl_d = wave_incl_scan_u32(lane_hd);
         excl_d = incl_d - lane_hd;
         tot_d = (uint32_t)__builtin_amdgcn_readlane((int)incl_d, 63);
         hd_extra = extra && a.text[0] == '>' ? 1u : 0u;                /* (first_seg: the buffer starts at byte 0) */
      }
      /* last newline per lane (tile-relative + 2 = start of the next line + 1; 0: none) and its prefix maximum */
      uint32_t incl_last = 0;
      if (tot_n && (tot_h || LL)) {                       /* wave-uniform */
         uint32_t my_last = 0;
#pragma unroll
         for (int r = 0; r < NM; r++)
            if (nmask[r]) my_last = (uint32_t)lane * CH + 32u * r + (31u - (uint32_t)__builtin_ctz(nmask[r])) + 2u;
         incl_last = wave_incl_max_u32(my_last);
         /* the segment's last newline decides which line runs on into the next segment (k_exact1) */
         if (LL) wv_lastnl = tile * TB + (uint32_t)__builtin_amdgcn_readlane((int)incl_last, 63) - 1u;
      }
      /* ---- ordered compaction of the hit lines: per-wave slice, no atomics ---- */
      if (tot_h) {
         if (slice_pos + tot_h <= a.slice_cap) {
            /* Where does the line of a hit start?  After the last newline before it: in my chunk, else in a
               lower lane's chunk (prefix maximum), else before the tile -- then the entry carries the hit
               position and k_stream_bounds searches backwards.  Offsets are tile-relative, +1 so 0 = none. */
            uint32_t before = stream_from_prev_lane(incl_last, 0u);        /* start+1 of the line my chunk begins in */
            if (extra && before == 0) before = 1;                          /* ... the buffer starts here */
            if (lane_hits) {
               uint32_t ord = incl_h - lane_hits;
               uint32_t nlb = incl_n - lane_nl + extra - 1u - excl_d - hd_extra;   /* counted rank of the line my chunk starts in */
#pragma unroll
               for (int r = 0; r < NM; r++) {
                  uint32_t mm = hmask[r];
                  while (mm) {
                     const uint32_t lz = (uint32_t)__builtin_clz(mm);
                     mm &= ~(0x80000000u >> lz);
                     const uint32_t nlt = lz ? nmask[r] >> (32 - lz) : 0u;  /* newlines before the hit, same group */
                     const uint32_t nb = (uint32_t)__popc(nlt) - (FA && lz ? (uint32_t)__popc(dmask[r] >> (32 - lz)) : 0u);
                     const uint32_t st1 = nlt ? (uint32_t)lane * CH + 32u * r + lz - (uint32_t)__builtin_ctz(nlt) + 1u : before;
                     const uint32_t hp = (uint32_t)lane * CH + 32u * r + lz;           /* the hit, tile-relative */
                     const uint32_t pos = st1 ? st1 - 1u : hp;
                     /* {tile | unresolved, rank | column of the hit << 13, line start (or hit) position, line rank} */
                     slice[slice_pos + ord] = make_uint4(tile | (st1 ? 0u : 0x80000000u), ord | ((hp - pos) << 13),
                                                         tile * TB + pos + a.pos_bias, nlb + nb);
                     ord++;
                  }
                  if (nmask[r]) before = (uint32_t)lane * CH + 32u * r + (31u - (uint32_t)__builtin_ctz(nmask[r])) + 2u;
                  nlb += (uint32_t)__popc(nmask[r]) - (uint32_t)__popc(dmask[r]);
               }
            }
            slice_pos += tot_h;
         } else {
            wv_overflow = true;
         }
      }
      if (lane == 0) {
         a.tile_cl[tile] = tot_n + extra - tot_d - hd_extra;    /* counted lines: headers excluded */
         a.tile_hits[tile] = tot_h;
      }
      /* a hit inside a line of >= a whole tile: ask for the long-line variant (of this configuration).  (Not for the tile
         the buffer ends in: without a newline it need not be a long line, just the tail of the last one.) */
      if (!LL && tot_h && !tot_n && !partial && !(t0 <= last && last < t0 + TB)) wv_dirty |= 2u;
      wv_lines += tot_n + extra;
      wv_hdrs += tot_d + hd_extra;
      wv_hitlines += tot_h;
   }
   if (lane == 0) {
      a.wg_hits[gwave] = wv_overflow ? 0u : slice_pos;
      a.wg_part[4 * gwave + 0] = wv_lines;
      a.wg_part[4 * gwave + 1] = wv_hdrs;
      if (LL) a.wg_lastnl[gwave] = wv_lastnl;    /* offset + 1 of the last newline this wave saw */
      a.wg_part[4 * gwave + 2] = wv_overflow ? (wv_hitlines | 0x80000000u) : wv_hitlines;
      if (SUB == 2 && wv_fakes * 4 > wv_lines) wv_dirty |= 8u;    /* more made-up candidates than a quarter of my lines */
      a.wg_part[4 * gwave + 3] = wv_dirty;       /* 1: a byte outside the alphabet, 2: wants the long-line variant, 4: superset, 8: see above (k_fused_post acts on them) */
   }
}

/* Slices -> ordered per-line arrays (tile_hits / tile_cl hold exclusive prefixes by now). */
__global__ __launch_bounds__(256) void k_stream_reorder(FusedArgs a, uint32_t nslices, uint32_t *hit_start, uint32_t *hit_line,
                                                        uint32_t *unresolved, uint32_t *hit_col)
{
   const Counters *c = a.cnt;
   if (c->overflow & 2u) return;
   const uint32_t lane = threadIdx.x & 63;
   for (uint32_t sl = blockIdx.x * 4 + (threadIdx.x >> 6); sl < nslices; sl += gridDim.x * 4) {      /* one wave per slice */
      const uint32_t n = a.wg_hits[sl];
      const uint4 *slice = a.tmp + (size_t)sl * a.slice_cap;
      const uint32_t lines0 = (uint32_t)c->lines;
      for (uint32_t i0 = lane; i0 < n; i0 += 256) {          /* four entries per lane in flight: the time of this kernel is load latency */
         uint4 e[4];
         uint32_t th[4], tc[4];
#pragma unroll
         for (int u = 0; u < 4; u++) e[u] = i0 + 64u * u < n ? slice[i0 + 64u * u] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
         for (int u = 0; u < 4; u++) {
            const uint32_t tile = e[u].x & 0x1FFFFFFFu;
            const bool ok = i0 + 64u * u < n;
            th[u] = ok ? a.tile_hits[tile] : 0u;
            tc[u] = ok ? a.tile_cl[tile] : 0u;
         }
#pragma unroll
         for (int u = 0; u < 4; u++) {
            if (i0 + 64u * u >= n) continue;
            const uint32_t dst = th[u] + (e[u].y & 0x1FFFu);
            hit_start[dst] = e[u].z;
            hit_line[dst] = lines0 + tc[u] + e[u].w + 1u;   /* 1-based, reference seeq.c:377 */
            /* 1: e.z is the hit itself (the line starts before the tile) */
            unresolved[dst] = e[u].x >> 31;
            hit_col[dst] = (e[u].y >> 13) & 0x3FFFFu;
         }
      }
   }
}

/* After the reorder: hit_start[k] holds the (biased) position of the first hit of a line, hit_line[k] its
 * line number, both ascending.  Turn the position into the start of its line (the exact pass scans whole
 * lines), and drop the repeats of a line (hit_start = 0xFFFFFFFF: k_exact1 skips the entry, nh = 0). */
/* Offset just after the last '\n' in text[lo, hi), or ~0 when there is none (16 bytes per step, backwards). */
__device__ __forceinline__ uint64_t stream_line_start_in(const uint8_t *text, uint64_t lo, uint64_t hi)
{
   uint64_t q = hi;
   while (q >= lo + 64) {                                  /* 64 bytes per step, the four loads in flight together (a line of reads: 2-3 steps) */
      fused_v4u v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const fused_v4u_unaligned *>(text + q - 64 + 16 * u);
#pragma unroll
      for (int u = 3; u >= 0; u--) {
         const uint32_t g3 = nl_flags(v[u].w), g2 = nl_flags(v[u].z), g1 = nl_flags(v[u].y), g0 = nl_flags(v[u].x);
         if (g3 | g2 | g1 | g0) {
            uint32_t byte;                                 /* index of the LAST newline among these 16 bytes */
            if (g3) byte = 12 + ((31 - (uint32_t)__builtin_clz(g3)) >> 3);
            else if (g2) byte = 8 + ((31 - (uint32_t)__builtin_clz(g2)) >> 3);
            else if (g1) byte = 4 + ((31 - (uint32_t)__builtin_clz(g1)) >> 3);
            else byte = (31 - (uint32_t)__builtin_clz(g0)) >> 3;
            return q - 64 + 16 * u + byte + 1;
         }
      }
      q -= 64;
   }
   while (q >= lo + 16) {
      const fused_v4u v = *reinterpret_cast<const fused_v4u_unaligned *>(text + q - 16);
      const uint32_t g3 = nl_flags(v.w), g2 = nl_flags(v.z), g1 = nl_flags(v.y), g0 = nl_flags(v.x);
      if (g3 | g2 | g1 | g0) {
         uint32_t byte;                                    /* index of the LAST newline among the 16 bytes */
         if (g3) byte = 12 + ((31 - (uint32_t)__builtin_clz(g3)) >> 3);
         else if (g2) byte = 8 + ((31 - (uint32_t)__builtin_clz(g2)) >> 3);
         else if (g1) byte = 4 + ((31 - (uint32_t)__builtin_clz(g1)) >> 3);
         else byte = (31 - (uint32_t)__builtin_clz(g0)) >> 3;
         return q - 16 + byte + 1;
      }
      q -= 16;
   }
   while (q > lo) { if (text[q - 1] == '\n') return q; q--; }
   return ~(uint64_t)0;
}

__global__ __launch_bounds__(256) void k_stream_bounds(ScanArgs a, uint32_t *hit_col, const uint32_t *tile_cl, uint32_t ntiles,
                                                       uint32_t tile_bytes)
{
   Counters *c = a.cnt;
   const uint32_t nhl = c->seg_nhitlines;
   const uint32_t stride = gridDim.x * 256;
   const uint64_t segb = a.seg_base + a.pos_bias;         /* the segment proper (a.seg_base is the biased base) */
   for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < nhl; k += stride) {
      const uint32_t ln = a.hit_line[k];
      const uint32_t prev = k ? a.hit_line[k - 1] : c->prev_hit_line;
      if (ln == prev) {                                   /* a repeat: keep the candidate's position for k_exact1's window walk */
         hit_col[k] = (a.nh[k] & 1u) ? a.hit_start[k] : a.hit_start[k] + hit_col[k];
         /* k_pair, windows: the line's first candidate belongs to the segment before this one, whose exact pass could not
            know of this one -- the run is void, the next one scans candidate lines to their ends (seeqdevScanFetch) */
         if (a.window_ok && ln == c->prev_hit_line) atomicOr(&c->overflow, 128u);
         a.hit_start[k] = 0xFFFFFFFFu;
         continue;
      }
      if (!(a.nh[k] & 1u)) continue;                      /* k_stream already found the start of the line */
      const uint64_t hp = a.seg_base + a.hit_start[k];   /* a byte of the line (inside the segment); never '\n' */
      if (hp >= a.nbytes || hp < segb || hp >= segb + a.seg_len) {       /* cannot be: an entry the scan kernel never wrote -- fail loudly, touch nothing */
         atomicOr(&c->overflow, 64u);
         a.hit_start[k] = 0xFFFFFFFFu;
         continue;
      }
      /* backwards to the byte after the previous '\n': first inside the hit's tile, then -- chromosome-long lines --
         tile by tile through the per-tile line counts (tile_cl[] holds their exclusive prefix: 4 bytes per 8 KB of
         text), and only when the line starts before the segment through the text in front of it */
      uint64_t q = ~(uint64_t)0;
      uint64_t floor_ = segb;                              /* nothing searched below this yet */
      if (tile_cl && hp >= segb) {
         uint32_t t = (uint32_t)((hp - segb) / tile_bytes);
         q = stream_line_start_in(a.text, segb + (uint64_t)t * tile_bytes, hp);
         if (q == ~(uint64_t)0 && t > 0) {
            /* The last tile before t that holds a newline which starts a counted line: tile_cl[] is the exclusive prefix of the tiles' counts, so it
               is the tile before the FIRST index whose prefix equals tile t's -- a binary search (round 5: the search went back tile by tile, two
               dependent loads each -- 16 384 tiles per 128 MiB chromosome line: 2.1 of the 6.5 ms of the sweep's cell m = 42, k = 9, for 74
               records).  Tile 0 carries the corrections for the line at byte 0 / a FASTA header there: when nothing lies between, look into it. */
            const uint32_t v = tile_cl[t];
            uint32_t lo = 0, hi = t;
            while (lo < hi) {
               const uint32_t mid = (lo + hi) >> 1;
               if (tile_cl[mid] >= v) hi = mid; else lo = mid + 1;
            }
            uint32_t tt = lo >= 1 ? lo - 1 : 0;
            q = stream_line_start_in(a.text, segb + (uint64_t)tt * tile_bytes, segb + (uint64_t)(tt + 1) * tile_bytes);
            if (q == ~(uint64_t)0 && tt > 0) q = stream_line_start_in(a.text, segb, segb + tile_bytes);      /* (tile 0, as the walk would have reached it) */
         }
         (void)ntiles;
      } else {
         floor_ = hp;
      }
      if (q == ~(uint64_t)0) {                            /* the line starts before the segment (or at byte 0) */
         q = stream_line_start_in(a.text, 0, floor_ < hp ? floor_ : hp);
         if (q == ~(uint64_t)0) q = 0;
      }
      if (q < a.seg_base) { atomicOr(&c->overflow, 8u); a.hit_start[k] = 0xFFFFFFFFu; }
      else {
         hit_col[k] = (uint32_t)(hp - q);
         a.hit_start[k] = (uint32_t)(q - a.seg_base);
      }
   }
}

/* ==========================================================================================================================
 * Long lines with MANY hits (the dense cells of the reference's published sweep: 10^4 .. 10^6 records in 24 chromosome
 * lines): the exact pass walks a line's candidates one after the other in one lane.  Where every hit is counted (SQ_ALL
 * records, COUNTMATCH) a candidate far enough behind the one before it may be walked by a lane of its own: between two
 * candidate chunks with a candidate-free chunk between them no hit ends (the scan kernel's verdict), the window of the
 * earlier one ends inside that free chunk at the latest (it is extended while the last columns of a chunk hold a score
 * <= tau, and a free chunk holds none), and the walk would jump to `wback` columns before the later candidate with a fresh
 * column anyway (k_exact1, WALK).  Such a candidate -- a LEADER -- gets what a line's first entry has: the line's start in
 * hit_start[], its column in hit_col[]; the entries behind it up to the next leader stay its repeats.  k_exact1 runs
 * unchanged; hits per entry add up per line as before, and the lines with a hit are counted once each (k_lead_lines).
 * Clean text only (a gap is jumped only when it is proven clean: with foreign bytes about, the line stays with one lane).
 * ========================================================================================================================== */
static constexpr int LEAD_ITEMS = 8;
static constexpr int LEAD_BLOCK = 256 * LEAD_ITEMS;

__device__ __forceinline__ uint32_t lead_block_incl_max(uint32_t v, uint32_t *s_wave /* >= 4 */)
{
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   uint32_t x = v;
#pragma unroll
   for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(x, d, 64);
      if (lane >= d) x = x > y ? x : y;
   }
   if (lane == 63) s_wave[wave] = x;
   __syncthreads();
   uint32_t before = 0;
   for (int w = 0; w < wave; w++) before = before > s_wave[w] ? before : s_wave[w];
   __syncthreads();
   return x > before ? x : before;
}

/* per block of LEAD_BLOCK entries: index + 1 of its last first-entry (0: none) */
__global__ __launch_bounds__(256) void k_lead_reduce(ScanArgs a, uint32_t *bmax)
{
   __shared__ uint32_t s_wave[4];
   const uint32_t nhl = a.cnt->seg_nhitlines;
   const uint32_t base = blockIdx.x * LEAD_BLOCK;
   if (base >= nhl) return;
   uint32_t v = 0;
#pragma unroll
   for (int i = 0; i < LEAD_ITEMS; i++) {
      const uint32_t k = base + threadIdx.x * LEAD_ITEMS + i;
      if (k < nhl && a.hit_start[k] != 0xFFFFFFFFu) v = k + 1u;
   }
   const uint32_t m = lead_block_incl_max(v, s_wave);
   if (threadIdx.x == 255) bmax[blockIdx.x] = m;
}

/* exclusive prefix maximum over the blocks (one workgroup) */
__global__ __launch_bounds__(256) void k_lead_top(ScanArgs a, uint32_t *bmax)
{
   __shared__ uint32_t s_wave[4];
   const uint32_t nb = (a.cnt->seg_nhitlines + LEAD_BLOCK - 1) / LEAD_BLOCK;
   uint32_t running = 0;
   for (uint32_t b0 = 0; b0 < nb; b0 += 256) {
      const uint32_t i = b0 + threadIdx.x;
      const uint32_t v = i < nb ? bmax[i] : 0u;
      const uint32_t incl = lead_block_incl_max(v, s_wave);
      const uint32_t prev = __shfl_up(incl, 1, 64);
      __shared__ uint32_t s_last[4];
      if ((threadIdx.x & 63) == 63) s_last[threadIdx.x >> 6] = incl;
      __syncthreads();
      uint32_t excl = (threadIdx.x & 63) ? prev : (threadIdx.x >> 6 ? s_last[(threadIdx.x >> 6) - 1] : 0u);
      excl = excl > running ? excl : running;
      if (i < nb) bmax[i] = excl;
      running = running > s_last[3] ? running : s_last[3];
      __syncthreads();
   }
}

/* fidx[k] = index of the first entry of k's line (0xFFFFFFFF: that entry belongs to the segment before); leaders marked in
   tmp[k] = {leader ? 1 : 0, line start, column, 0} -- committed by k_lead_commit once every lane has read its neighbour */
__global__ __launch_bounds__(256) void k_lead_apply(ScanArgs a, const uint32_t *hit_col, const uint32_t *bmax, uint32_t *fidx, uint32_t *lflag, uint4 *tmp,
                                                    uint32_t wback, unsigned long long *lkey)
{
   __shared__ uint32_t s_wave[4];
   const Counters *c = a.cnt;
   const uint32_t nhl = c->seg_nhitlines;
   const uint32_t base = blockIdx.x * LEAD_BLOCK;
   if (base >= nhl) return;
   uint32_t item[LEAD_ITEMS];
   uint32_t v = 0;
#pragma unroll
   for (int i = 0; i < LEAD_ITEMS; i++) {
      const uint32_t k = base + threadIdx.x * LEAD_ITEMS + i;
      if (k < nhl && a.hit_start[k] != 0xFFFFFFFFu) v = k + 1u;
      item[i] = v;                                         /* running maximum inside the thread */
   }
   const uint32_t incl = lead_block_incl_max(v, s_wave);
   uint32_t before = __shfl_up(incl, 1, 64);
   __shared__ uint32_t s_last[4];
   if ((threadIdx.x & 63) == 63) s_last[threadIdx.x >> 6] = incl;
   __syncthreads();
   before = (threadIdx.x & 63) ? before : (threadIdx.x >> 6 ? s_last[(threadIdx.x >> 6) - 1] : 0u);
   const uint32_t bprev = bmax[blockIdx.x];
   before = before > bprev ? before : bprev;
   const bool promote = c->dirty == 0u;
   const uint32_t ch = a.stream_ch;
   const uint32_t lastnl = c->seg_last_nl;
   const bool last_seg = a.seg_base + a.pos_bias + a.seg_len >= a.nbytes;
#pragma unroll
   for (int i = 0; i < LEAD_ITEMS; i++) {
      const uint32_t k = base + threadIdx.x * LEAD_ITEMS + i;
      if (k >= nhl) break;
      const uint32_t f1 = item[i] > before ? item[i] : before;      /* first entry of my line, + 1 */
      fidx[k] = f1 ? f1 - 1u : 0xFFFFFFFFu;
      const uint32_t hs = a.hit_start[k];
      uint4 t = make_uint4(0u, 0u, 0u, 0u);
      if (hs != 0xFFFFFFFFu && lkey) lkey[k] = ~0ull;       /* (SQ_BEST: the line's best group, k_lead_best) */
      if (hs != 0xFFFFFFFFu || !(promote && f1 && k > 0)) lflag[k] = 0u;      /* (a first entry's flag: "the line has a hit", set by k_lead_lines) */
      if (hs == 0xFFFFFFFFu && promote && f1 && k > 0) {
         /* positions relative to the segment's (biased) base: a repeat holds its own, a first entry start + column */
         const uint32_t ps = a.hit_start[k - 1];
         const uint32_t pabs = ps != 0xFFFFFFFFu ? ps + hit_col[k - 1] : hit_col[k - 1];
         const uint32_t abs_ = hit_col[k];
         /* end of the chunk behind the previous candidate's -- behind the restart table (round 5) the end of the previous candidate's WINDOW: m + tau + 2
            columns behind it (k_exact1; the walk may step up to a block further, which the 64 below allow for, and k_lead_check holds every fresh
            start against where the walk before it really ended) */
         const uint32_t free_end = a.ll_restart ? pabs + (uint32_t)a.m + (uint32_t)a.tau + 2u : ((pabs / ch) + 2u) * ch;
         const uint32_t ls = a.hit_start[f1 - 1u];
         /* (k_exact1 walks windows only in lines that end inside the segment: the same test) */
         const bool win = ls != 0xFFFFFFFFu && (last_seg || (lastnl != 0u && (int64_t)ls - (int64_t)a.pos_bias < (int64_t)lastnl));
         if (win && abs_ >= free_end + wback + 64u) {          /* (the walk advances in blocks of 64 columns) */
            t = make_uint4(1u, ls, abs_ - ls, 0u);
            lflag[k] = 2u;                                  /* a leader: k_lead_check looks at it */
         } else {
            lflag[k] = 0u;
         }
      }
      tmp[k] = t;
   }
}

__global__ __launch_bounds__(256) void k_lead_commit(ScanArgs a, uint32_t *hit_col, const uint4 *tmp)
{
   const uint32_t nhl = a.cnt->seg_nhitlines;
   const uint32_t stride = gridDim.x * 256;
   for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < nhl; k += stride) {
      const uint4 t = tmp[k];
      if (t.x) { a.hit_start[k] = t.y; hit_col[k] = t.z; a.walk_end[k] = 0xFFFFFFFFu; }      /* (walk_end: "not yet vouched for" -- the walk before it writes where it stopped) */
   }
}

/* lines with >= 1 hit, each once: nh[] holds the entries' hit counts; per wave one atomic pair per line it sees */
__global__ __launch_bounds__(256) void k_lead_lines(ScanArgs a, const uint32_t *fidx, uint32_t *lflag)
{
   const uint32_t nhl = a.cnt->seg_nhitlines;
   const uint32_t stride = gridDim.x * 256;
   const uint32_t kmax = (nhl + stride - 1) / stride * stride;
   const uint32_t lane = threadIdx.x & 63u;
   for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < kmax; k += stride) {
      const bool live = k < nhl;
      const uint32_t f = live ? fidx[k] : 0xFFFFFFFEu;
      const bool has = live && f != 0xFFFFFFFFu && a.nh[k] != 0u;
      const uint32_t fprev = __shfl_up(f, 1, 64);
      const bool head = lane == 0 || f != fprev;
      const uint64_t heads = __ballot(head), hass = __ballot(has);
      if (head) {
         const uint64_t above = lane == 63 ? 0ull : heads >> (lane + 1u);
         const uint32_t len = above ? (uint32_t)__builtin_ctzll(above) + 1u : 64u - lane;
         const uint64_t seg = (hass >> lane) & (len >= 64u ? ~0ull : ((1ull << len) - 1ull));
         if (seg && f < 0xFFFFFFFEu && atomicExch(&lflag[f], 1u) == 0u) atomicAdd(&a.cnt->seg_nmatch, 1u);
      }
   }
}

/* after COUNT: every leader's fresh start (wback columns before its candidate) must lie behind the end of the walk before
   it -- else that walk would have gone on into the leader's window (a score <= tau near the end of every chunk between them:
   periodic patterns in periodic text) and the two lanes have counted the stretch twice: the run is void (overflow 256) */
__global__ __launch_bounds__(256) void k_lead_check(ScanArgs a, const uint32_t *hit_col, const uint32_t *lflag, uint32_t wback)
{
   const uint32_t nhl = a.cnt->seg_nhitlines;
   const uint32_t stride = gridDim.x * 256;
   for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < nhl; k += stride) {
      if (lflag[k] != 2u) continue;
      const uint32_t abs_ = a.hit_start[k] + hit_col[k];
      /* (a walk that ran on to the end of its line never looked at this entry: the mark of k_lead_commit is still there) */
      if (!((uint64_t)abs_ > (uint64_t)a.walk_end[k] + wback)) atomicOr(&a.cnt->overflow, 256u);
   }
}

/* SQ_BEST with leaders: every group of a line has found ITS best hit (nh[k] = 1, the hit in the COUNT -> EMIT cache); the
   line's record is the one with the smallest distance, the first of those: atomicMin over {distance, entry} per line, then
   the groups that lost give up their record slot. */
__global__ __launch_bounds__(256) void k_lead_best(ScanArgs a, const uint32_t *fidx, unsigned long long *lkey, const uint4 *cache, int pass)
{
   const uint32_t nhl = a.cnt->seg_nhitlines;
   const uint32_t stride = gridDim.x * 256;
   for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < nhl; k += stride) {
      if (a.nh[k] == 0u) continue;
      const uint32_t f = fidx[k];
      if (f == 0xFFFFFFFFu) continue;
      if (pass == 0) atomicMin(&lkey[f], ((unsigned long long)cache[k].y << 32) | k);
      else if ((uint32_t)lkey[f] != k) a.nh[k] = 0u;
   }
}

#endif
