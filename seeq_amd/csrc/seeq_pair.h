/*
 * seeq_pair.h -- k_pair: the line-agnostic table walk of k_stream (seeq_stream.h) with TWO text bytes per table step.
 *
 * k_stream is held by the LDS gather unit: one 64-lane ds_read_u16 over a 53 KB table costs 5.6 LDS cycles (bank
 * conflicts; profiles/microbench/lds_bank_model.c) and it pays 1.375 of them per text byte.  Here a step consumes a PAIR
 * of bases: the table (seeq_dfa.h section 3) has 16 columns per state -- 32-byte rows, row offsets in 16 bits, so at
 * most 2 047 rows: the walk carries the pattern's longest prefix that fits (headline pattern: 17 of 20 positions,
 * 1 700 states after minimisation, 55 KB) or a partition filter, whichever makes fewer false candidates.  Gathers per
 * text byte: 0.5 x (64 + 20) / 64 = 0.66.
 *
 *   - Text bytes are taken as 2-bit codes (bits 1-2: A 0, C 1, T 2, G 3); per text word two VALU instructions build both
 *     pair indices (u = w & 0x06060606; t = u | u << 10: bytes 1 and 3 of t hold {first code, second code} << 1), and
 *     the SDWA v_xor of k_stream picks them: address = state ^ t.BYTE_1 / BYTE_3.  Every other byte aliases onto a base
 *     ('\n' -> C, N -> G, anything else -> whatever its bits say): an alias only turns mismatches into matches.
 *   - There is no newline column and no absorbing state: the walk runs across line ends and RESTARTS at the root when
 *     it accepts (the accepting transition leads to a flagged copy of the row it restarts in: state >= hit_base marks the
 *     pair; v_cmp + v_addc shift the flag into a mask of 32 pairs = one 64-byte chain).
 *   - Two chains per lane (bytes 0-63 / 64-127) as in k_stream, each warmed up over the 4 * WU >= warm bytes before it.
 *     A walk that accepts DURING its warm-up restarts there and may then miss an occurrence that ends in its own first
 *     bytes (the restart sits inside it) while the flag is somebody else's position -- possibly on the line before: such
 *     a chain reports its own first pair as a candidate (v_max over the warm-up states, one compare at the end).
 *
 * Why every line with a hit gets a candidate, and why the exact pass may start m + tau columns before a line's FIRST
 * candidate (tests/test_kernel_core_host.py::test_pair_automaton_... checks both on the host against the oracle):
 * an occurrence O of the pattern in line L contains an occurrence P = [s, j] of the prefix (of a part) with no more than
 * its threshold of errors, j - s <= warm.  The chain that owns j has P inside its window; it flags the pair of j unless
 * it restarted at some j1 in [s, j) -- then j1 is flagged: by this chain if it owns j1, else (j1 in the warm-up) the
 * made-up candidate at the chain's first pair, which lies in (j1, j].  Either way a candidate position e with s <= e - 1
 * and e <= j + 1 exists, inside O or on the byte after it: in line L (a newline right after O belongs to the line it
 * ends -- the line of a position is the number of newlines strictly before it).  And e <= s + m + tau for EVERY
 * occurrence of the line, so the first candidate c of the line satisfies c - (m + tau) <= the start of every occurrence:
 * a fresh column started there sees every alignment with <= tau errors the line holds.
 *
 * Everything k_pair reports is a CANDIDATE (ScanArgs.filter): the exact pass (k_exact1) verifies each one -- under
 * SQ_FAIL and SQ_CONVERT alike, since aliasing is harmless for a superset (SQ_IGNORE, where a skipped byte stretches a
 * match, stays with k_stream).  The alphabet check and Counters.dirty are kept: when the text holds a byte that could
 * end a line early, the exact pass starts at the beginning of the line instead of before the candidate.
 * Bookkeeping (newline masks, line ranks, line starts, slices, FASTA headers) is k_stream's.
 */
#ifndef SEEQ_PAIR_H_
#define SEEQ_PAIR_H_

/* one pair of each chain: address = state ^ pair index (byte K of the prepared word), then the gather */
#define PAIR_X2(K) \
   asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #K : "=v"(ada) : "v"(sa), "v"(ta)); \
   asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #K : "=v"(adb) : "v"(sb), "v"(tb)); \
   sa = *(stream_lds_cu16 *)(uintptr_t)ada; sb = *(stream_lds_cu16 *)(uintptr_t)adb; \
   __builtin_amdgcn_sched_barrier(0);              /* both gathers go out together (left alone the scheduler walks the chains one after the other) */
/* flagged row? -> shifted into the chain's pair mask (first pair ends up in bit 31) */
#define PAIR_EV2 \
   asm("v_cmp_ge_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(hma) : "v"(sa), "v"(hit_base) : "vcc"); \
   asm("v_cmp_ge_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(hmb) : "v"(sb), "v"(hit_base) : "vcc");

/* the two pair indices of a text word, in bytes 1 and 3: {code of the first byte, code of the second} << 1 */
__device__ __forceinline__ uint32_t pair_prep(uint32_t w)
{
   const uint32_t u = w & 0x06060606u;
   return u | (u << 10);
}

/* four warm-up bytes of each chain: walk, and remember the largest state seen (a flagged row is >= hit_base) */
__device__ __forceinline__ void pair_warm4x2(uint32_t &sa, uint32_t wa, uint32_t &xa, uint32_t &sb, uint32_t wb, uint32_t &xb)
{
   const uint32_t ta = pair_prep(wa), tb = pair_prep(wb);
   uint32_t ada, adb;
   PAIR_X2(1) xa = sa > xa ? sa : xa; xb = sb > xb ? sb : xb;
   PAIR_X2(3) xa = sa > xa ? sa : xa; xb = sb > xb ? sb : xb;
}

/* four owned bytes of each chain: walk + pair mask */
__device__ __forceinline__ void pair_own4x2(uint32_t &sa, uint32_t wa, uint32_t &hma, uint32_t &sb, uint32_t wb, uint32_t &hmb, uint32_t hit_base)
{
   const uint32_t ta = pair_prep(wa), tb = pair_prep(wb);
   uint32_t ada, adb;
   PAIR_X2(1) PAIR_EV2 PAIR_X2(3) PAIR_EV2
}

/* word k (0..7) of the 32 bytes held in two 16-byte pieces */
__device__ __forceinline__ uint32_t pair_word8(const fused_v4u &p, const fused_v4u &q, int k)
{
   return k == 0 ? p.x : k == 1 ? p.y : k == 2 ? p.z : k == 3 ? p.w : k == 4 ? q.x : k == 5 ? q.y : k == 6 ? q.z : q.w;
}

/* WU: warm-up dwords (4 .. 8); FA: FASTA input (header lines: see k_stream) */
template <int WU, bool FA>
__global__ __launch_bounds__(64 * STREAM_NW, 8) void k_pair(FusedArgs a)
{
   constexpr int NW = STREAM_NW;
   constexpr int CH = 128;
   constexpr int NQ = CH / 16;                            /* 16-byte pieces per lane */
   constexpr int NM = CH / 32;                            /* newline mask registers per lane */
   constexpr uint32_t TB = 64u * CH;                      /* tile bytes */
   static_assert(WU >= 4 && WU <= 8, "warm-up is 16 .. 32 bytes");
   extern __shared__ __align__(16) uint8_t dsmem[];
   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   {
      const fused_v4u *src = reinterpret_cast<const fused_v4u *>(a.dfa);
      for (uint32_t i = tid; i < a.dfa_rows; i += 64 * NW) reinterpret_cast<fused_v4u *>(dsmem)[i] = src[i];
   }
   __syncthreads();                                       /* the only barrier: the table is read-only from here */
   const uint32_t hit_base = a.dfa_final_base;             /* state values >= this are flagged rows (seeq_pair_from_next) */

   const uint32_t gwave = blockIdx.x * NW + wave, nwaves = gridDim.x * NW;
   uint32_t wv_lines = 0, wv_hitlines = 0, wv_hdrs = 0, slice_pos = 0;    /* wave-uniform */
   bool wv_overflow = false;
   uint32_t wv_dirty = 0;
   uint4 *slice = a.tmp + (size_t)gwave * a.slice_cap;
   const uint64_t lim = a.seg_base + a.seg_len;           /* bytes at or beyond it are not this segment's */
   const uint64_t last = a.nbytes - 1;

   for (uint32_t tile = gwave; tile < a.ntiles; tile += nwaves) {
      const uint64_t t0 = a.seg_base + (uint64_t)tile * TB;
      uint32_t lane_off = (uint32_t)lane * CH;
      asm volatile("" : "+v"(lane_off));                  /* (see k_stream: keeps the per-lane 64-bit addresses out of the loop-invariant set) */
      const uint64_t my = t0 + lane_off;
      const bool partial = tile + 1 == a.ntiles && (a.seg_len % TB) != 0;
      fused_v4u v[NQ];
      if (!partial) {
         const uint8_t *p = a.text + my;
#pragma unroll
         for (int q = 0; q < NQ; q++) v[q] = *reinterpret_cast<const fused_v4u_unaligned *>(p + 16 * q);
      } else {
#pragma unroll
         for (int q = 0; q < NQ; q++) v[q] = dfa_load16(a.text, my + 16 * q, lim);       /* '\n' beyond the segment */
      }
      /* the 32 bytes before the tile (lane 0's warm-up); '\n' when the buffer starts here */
      fused_v4u pa = fused_v4u{0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au}, pb = pa;
      if (t0 >= 32) {
         pa = *reinterpret_cast<const fused_v4u_unaligned *>(a.text + t0 - 32);
         pb = *reinterpret_cast<const fused_v4u_unaligned *>(a.text + t0 - 16);
      }
      /* ---- alphabet check (a byte that could end a line early: the exact pass then starts at the line's first byte) ---- */
      bool tile_clean;
      {
         uint32_t bad = 0;
#pragma unroll
         for (int q = 0; q < NQ; q++) bad |= fused_bad4(v[q].x) | fused_bad4(v[q].y) | fused_bad4(v[q].z) | fused_bad4(v[q].w);
         uint32_t flag = (uint32_t)__builtin_amdgcn_readfirstlane(__ballot(bad != 0) != 0 ? 1 : 0);
         asm volatile("" : "+s"(flag));
         wv_dirty |= flag;
         tile_clean = flag == 0;
      }
      /* ---- the walk: chain A = bytes 0..63 (warm-up: the previous lane's last bytes), chain B = bytes 64..127 ---- */
      uint32_t hm[2];
      {
         uint32_t sa = 0, sb = 0, xa = 0, xb = 0, hma = 0, hmb = 0;
#pragma unroll
         for (int k = 8 - WU; k < 8; k++)
            pair_warm4x2(sa, stream_from_prev_lane(pair_word8(v[NQ - 2], v[NQ - 1], k), pair_word8(pa, pb, k)), xa,
                         sb, pair_word8(v[NQ / 2 - 2], v[NQ / 2 - 1], k), xb);
#pragma unroll
         for (int q = 0; q < NQ / 2; q++) {
            pair_own4x2(sa, v[q].x, hma, sb, v[q + NQ / 2].x, hmb, hit_base);
            pair_own4x2(sa, v[q].y, hma, sb, v[q + NQ / 2].y, hmb, hit_base);
            pair_own4x2(sa, v[q].z, hma, sb, v[q + NQ / 2].z, hmb, hit_base);
            pair_own4x2(sa, v[q].w, hma, sb, v[q + NQ / 2].w, hmb, hit_base);
         }
         /* accepted during the warm-up: my first pair is a candidate (see the header) */
         hm[0] = hma | (xa >= hit_base ? 0x80000000u : 0u);
         hm[1] = hmb | (xb >= hit_base ? 0x80000000u : 0u);
      }
      /* ---- newline masks, apart from the walk ---- */
      uint32_t nmask[NM];
#pragma unroll
      for (int r = 0; r < NM; r++) nmask[r] = stream_nl_mask32(v[2 * r], v[2 * r + 1], tile_clean);
      /* ---- bookkeeping: what the tile owns ---- */
      uint32_t valid = CH;                                /* bytes of my chunk inside the segment */
      if (partial) {
         valid = lim > my ? (lim - my < CH ? (uint32_t)(lim - my) : (uint32_t)CH) : 0u;
#pragma unroll
         for (int r = 0; r < NM; r++) {
            const uint32_t lo = 32u * r;
            nmask[r] &= valid <= lo ? 0u : (valid >= lo + 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> (valid - lo)));
         }
#pragma unroll
         for (int x = 0; x < 2; x++) {                    /* a pair counts when its first byte is the segment's */
            const uint32_t vx = valid <= 64u * x ? 0u : (valid - 64u * x >= 64u ? 64u : valid - 64u * x);
            const uint32_t np = (vx + 1u) >> 1;
            hm[x] &= np >= 32u ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> np);
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
      uint32_t lane_nl = 0;
      const uint32_t lane_hits = (uint32_t)__popc(hm[0]) + (uint32_t)__popc(hm[1]);
#pragma unroll
      for (int r = 0; r < NM; r++) lane_nl += (uint32_t)__popc(nmask[r]);
      const uint32_t incl_h = wave_incl_scan_u32(lane_hits), incl_n = wave_incl_scan_u32(lane_nl);
      const uint32_t tot_h = (uint32_t)__builtin_amdgcn_readlane((int)incl_h, 63);
      const uint32_t tot_n = (uint32_t)__builtin_amdgcn_readlane((int)incl_n, 63);
      const uint32_t extra = (uint32_t)__builtin_amdgcn_readfirstlane((a.first_seg && tile == 0) ? 1 : 0);   /* the line starting at byte 0 */
      /* FASTA: which of my newlines start a header line? */
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
         hd_extra = extra && a.text[0] == '>' ? 1u : 0u;
      }
      /* last newline per lane (tile-relative + 2 = start of the next line + 1; 0: none) and its prefix maximum */
      uint32_t incl_last = 0;
      if (tot_n && tot_h) {                               /* wave-uniform */
         uint32_t my_last = 0;
#pragma unroll
         for (int r = 0; r < NM; r++)
            if (nmask[r]) my_last = (uint32_t)lane * CH + 32u * r + (31u - (uint32_t)__builtin_ctz(nmask[r])) + 2u;
         incl_last = wave_incl_max_u32(my_last);
      }
      /* ---- ordered compaction of the candidates: per-wave slice, no atomics ---- */
      if (tot_h) {
         if (slice_pos + tot_h <= a.slice_cap) {
            uint32_t before = stream_from_prev_lane(incl_last, 0u);        /* start+1 of the line my chunk begins in */
            if (extra && before == 0) before = 1;                          /* ... the buffer starts here */
            if (lane_hits) {
               uint32_t ord = incl_h - lane_hits;
               uint32_t nlb = incl_n - lane_nl + extra - 1u - excl_d - hd_extra;   /* counted rank of the line my chunk starts in */
#pragma unroll
               for (int r = 0; r < NM; r++) {
                  /* the pairs of this 32-byte group, first pair in bit 31 */
                  uint32_t mm = (r & 1) ? hm[r >> 1] << 16 : hm[r >> 1] & 0xFFFF0000u;
                  while (mm) {
                     const uint32_t lp = (uint32_t)__builtin_clz(mm);
                     mm &= ~(0x80000000u >> lp);
                     uint32_t lz = 2u * lp + 1u;                            /* the pair's second byte, within the group */
                     if (partial && 32u * r + lz >= valid) lz = valid - 1u - 32u * r;      /* ... or its first, when the segment ends between them */
                     const uint32_t nlt = lz ? nmask[r] >> (32 - lz) : 0u;  /* newlines before it, same group */
                     const uint32_t nb = (uint32_t)__popc(nlt) - (FA && lz ? (uint32_t)__popc(dmask[r] >> (32 - lz)) : 0u);
                     const uint32_t st1 = nlt ? (uint32_t)lane * CH + 32u * r + lz - (uint32_t)__builtin_ctz(nlt) + 1u : before;
                     const uint32_t hp = (uint32_t)lane * CH + 32u * r + lz;           /* the candidate, tile-relative */
                     const uint32_t pos = st1 ? st1 - 1u : hp;
                     /* {tile | unresolved, rank | column of the candidate << 13, line start (or candidate) position, line rank} */
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
      /* a candidate inside a line of >= a whole tile: this is long-line input -- k_stream's long-line variant takes over */
      if (tot_h && !tot_n && !partial && !(t0 <= last && last < t0 + TB)) wv_dirty |= 2u;
      wv_lines += tot_n + extra;
      wv_hdrs += tot_d + hd_extra;
      wv_hitlines += tot_h;
   }
   if (lane == 0) {
      a.wg_hits[gwave] = wv_overflow ? 0u : slice_pos;
      a.wg_part[4 * gwave + 0] = wv_lines;
      a.wg_part[4 * gwave + 1] = wv_hdrs;
      a.wg_part[4 * gwave + 2] = wv_overflow ? (wv_hitlines | 0x80000000u) : wv_hitlines;
      a.wg_part[4 * gwave + 3] = wv_dirty;       /* 1: a byte outside the alphabet, 2: long-line input (k_fused_post acts on them) */
   }
}

#undef PAIR_X2
#undef PAIR_EV2

#endif
