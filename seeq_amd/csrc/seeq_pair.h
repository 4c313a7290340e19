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
 *     it accepts (the accepting transition leads to a flagged copy of the row it restarts in; flagged rows have odd state
 *     values, so one v_alignbit per step shifts the flag into a mask of 32 pairs = one 64-byte chain).
 *   - Two chains per lane (bytes 0-63 / 64-127) as in k_stream, each warmed up over the 4 * WU >= warm bytes before it.
 *     A walk that accepts DURING its warm-up restarts there and may then miss an occurrence that ends in its own first
 *     bytes (the restart sits inside it) while the flag is somebody else's position -- possibly on the line before: such
 *     a chain reports its own first pair as a candidate (the warm-up states OR-ed together, bit 0 tested once).
 *   - What else a tile needs from its text -- the alphabet check and the newline masks -- is computed word by word
 *     BETWEEN a step's two gathers and the instructions that need their results, i.e. in the shadow of the LDS latency
 *     the walk is made of (fast check: upper case A C G T N and newlines only -- v_perm + v_sad_u8 per word; newline flags:
 *     v_perm + v_dot4 per word; a wave that meets a tile which fails the check walks on without both and makes exact newline
 *     masks from the tile's registers: DM in the kernel, round 5 -- FASTQ records stay on this kernel).
 *     With the gathers halved the kernel is no longer held by the LDS unit alone (60 % busy, of which half bank conflicts):
 *     it sits 1.2 - 1.4 x above both the memory system's floor and the gather unit's at once.
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
 * end a line early, the exact pass looks at the bytes between the line's start and a candidate's window before it trusts
 * the window (verify_prefix_dirty, seeq_verify.h).
 * Bookkeeping (newline masks, line ranks, line starts, slices, FASTA headers) is k_stream's.
 */
#ifndef SEEQ_PAIR_H_
#define SEEQ_PAIR_H_

/* one pair of each chain: address = state ^ pair index (byte K of the prepared word), then the gathers -- both out before
 * anything else (left alone the scheduler walks the chains one after the other) */
#define PAIR_X2(K) \
   asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #K : "=v"(ada) : "v"(sa), "v"(ta)); \
   asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #K : "=v"(adb) : "v"(sb), "v"(tb)); \
   sa = *(stream_lds_cu16 *)(uintptr_t)ada; sb = *(stream_lds_cu16 *)(uintptr_t)adb; \
   __builtin_amdgcn_sched_barrier(0);

/* the two pair indices of a text word, in bytes 1 and 3: {code of the first byte, code of the second} << 1 */
__device__ __forceinline__ uint32_t pair_prep(uint32_t w)
{
   const uint32_t u = w & 0x06060606u;
   return u | (u << 10);
}

/* Per text word, in the shadow of a gather: the FAST alphabet check -- the canonical byte of the word's table columns
 * (A C T G . \n . N, upper case) against the word, differences summed by v_sad_u8: zero over a tile = nothing but upper
 * case A C G T N and newlines (anything else, lower case included, sends the tile through the exact check) -- and the
 * newline flags of the word (column 5), dropped into the mask by v_dot4_u32_u8 (two words per shift).
 * (Round 5: indexing the two tables by the bytes' low three bits, which tell the six bytes apart as well, saves the shift -- and was 3 % SLOWER:
 * the allocator, at its 64 registers, spilled one more value inside the tile loop.  profiles/r05/ab_fast_check_index.txt) */
__device__ __forceinline__ void pair_chk(uint32_t w, uint32_t &bad, uint32_t &nm, bool first_of_two)
{
   const uint32_t idx = (w >> 1) & 0x07070707u;
   bad = __builtin_amdgcn_sad_u8(w, __builtin_amdgcn_perm(0x4EFF0AFFu, 0x47544341u, idx), bad);
   const uint32_t nf = __builtin_amdgcn_perm(0x00000100u, 0u, idx);
   nm = first_of_two ? __builtin_amdgcn_udot4(nf, 0x10204080u, nm << 8, false) : __builtin_amdgcn_udot4(nf, 0x01020408u, nm, false);
}

/* four warm-up bytes of each chain: walk; the states of a chain are OR-ed into its `seen` (bit 0: the walk accepted) */
__device__ __forceinline__ void pair_warm4x2(uint32_t &sa, uint32_t wa, uint32_t &sb, uint32_t wb, uint32_t &seena, uint32_t &seenb)
{
   const uint32_t ta = pair_prep(wa), tb = pair_prep(wb);
   uint32_t ada, adb;
   /* (asm: as plain C the chain of ORs is re-associated into a tree evaluated after the warm-up, with every state kept -- spilled -- until then) */
   PAIR_X2(1) asm("v_or_b32 %0, %0, %1" : "+v"(seena) : "v"(sa)); asm("v_or_b32 %0, %0, %1" : "+v"(seenb) : "v"(sb));
   PAIR_X2(3) asm("v_or_b32 %0, %0, %1" : "+v"(seena) : "v"(sa)); asm("v_or_b32 %0, %0, %1" : "+v"(seenb) : "v"(sb));
}

/* four owned bytes of each chain: walk + pair masks (first pair of a chain ends up in bit 0), and the per-word checks of
 * both words between the gathers and their use */
template <bool CHK>
__device__ __forceinline__ void pair_own4x2(uint32_t &sa, uint32_t wa, uint32_t &hma, uint32_t &nma, uint32_t &sb, uint32_t wb, uint32_t &hmb, uint32_t &nmb,
                                            uint32_t &bad, bool first_of_two)
{
   const uint32_t ta = pair_prep(wa), tb = pair_prep(wb);
   uint32_t ada, adb;
   /* (the empty volatile asm statements pin the side-effect-free work where it is written: without them it is all moved
      behind the walk, with every state and every intermediate kept -- spilled -- until then) */
   PAIR_X2(1)
   asm volatile("" : "+v"(wa));
   if (CHK) pair_chk(wa, bad, nma, first_of_two);
   asm volatile("" : "+v"(bad), "+v"(nma));
   __builtin_amdgcn_sched_barrier(0);
   hma = __builtin_amdgcn_alignbit(sa, hma, 1); hmb = __builtin_amdgcn_alignbit(sb, hmb, 1);
   asm volatile("" : "+v"(hma), "+v"(hmb));
   PAIR_X2(3)
   asm volatile("" : "+v"(wb));
   if (CHK) pair_chk(wb, bad, nmb, first_of_two);
   asm volatile("" : "+v"(bad), "+v"(nmb));
   __builtin_amdgcn_sched_barrier(0);
   hma = __builtin_amdgcn_alignbit(sa, hma, 1); hmb = __builtin_amdgcn_alignbit(sb, hmb, 1);
   asm volatile("" : "+v"(hma), "+v"(hmb));
}

/* word k (0..7) of the 32 bytes held in two 16-byte pieces */
__device__ __forceinline__ uint32_t pair_word8(const fused_v4u &p, const fused_v4u &q, int k)
{
   return k == 0 ? p.x : k == 1 ? p.y : k == 2 ? p.z : k == 3 ? p.w : k == 4 ? q.x : k == 5 ? q.y : k == 6 ? q.z : q.w;
}

typedef uint32_t pair_u32_unaligned __attribute__((aligned(1)));

/* SQ_IGNORE (IG, round 5): which of 32 characters are NOT bases -- everything but A C G T U N in either case (seeqcore.h:89-111) --, as a mask,
 * first character = bit 31.  The reference skips such a byte (libseeq.c:265-266) unless it ends the line: skipped = this mask without the newlines
 * (a NUL counts as skipped here, which only makes the rule below more careful).  Per word: the low three bits of a byte pick its column's one base
 * from an eight-entry table (v_perm; A 1, C 3, T 4, U 5, N 6, G 7; columns 0 and 2 -- the newline's -- hold none), the byte is a base when it equals
 * it apart from bit 5; the four flags drop into the mask by v_dot4 (the flags stay at bit 7: the sums carry a factor 128, taken out per 16
 * characters).  8.5 VALU per word. */
__device__ __forceinline__ uint32_t pair_nonbase_mask32(const fused_v4u &a, const fused_v4u &b)
{
   const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
   uint32_t h[2];
#pragma unroll
   for (int g = 0; g < 2; g++) {
      uint32_t acc = 0;
#pragma unroll
      for (int k = 4 * g; k < 4 * g + 4; k += 2) {
         uint32_t f[2];
#pragma unroll
         for (int u = 0; u < 2; u++) {
            const uint32_t x = w[k + u];
            const uint32_t t = x ^ __builtin_amdgcn_perm(0x474E5554u, 0x43FF41FFu, x & 0x07070707u);
            f[u] = (((t & 0x5F5F5F5Fu) + 0x7F7F7F7Fu) | t) & 0x80808080u;
         }
         acc = __builtin_amdgcn_udot4(f[0], 0x10204080u, __builtin_amdgcn_udot4(f[1], 0x01020408u, acc << 8, false), false);
      }
      h[g] = acc;
   }
   return (h[0] << 9) | (h[1] >> 7);
}

/* characters of the lane's 128 in front of lane-relative position b (0 .. 128) that are set in the masks m[] (first character of a group = bit 31);
   c[] = the groups' exclusive prefix counts, c[4] = the total */
__device__ __forceinline__ uint32_t pair_count_before(const uint32_t (&m)[4], const uint32_t (&c)[5], uint32_t b)
{
   const uint32_t r = b >> 5, lz = b & 31u;
   const uint32_t mr = r == 0 ? m[0] : r == 1 ? m[1] : r == 2 ? m[2] : r == 3 ? m[3] : 0u;
   const uint32_t cr = r == 0 ? c[0] : r == 1 ? c[1] : r == 2 ? c[2] : r == 3 ? c[3] : c[4];
   return cr + (lz ? (uint32_t)__popc(mr >> (32u - lz)) : 0u);
}

/* The walk of one tile: warm-up of both chains, then their 64 owned bytes each; CHK: the fast alphabet check and the fast newline masks
 * ride along (nmask[], bad); else the caller makes the masks. */
template <int WU, bool CHK>
__device__ __forceinline__ void pair_walk(const fused_v4u (&v)[8], uint32_t halo, bool halo_nl, uint32_t (&hm)[2], uint32_t (&nmask)[4], uint32_t &tile_bad)
{
   constexpr int NQ = 8, NM = 4;
   uint32_t sa = 0, sb = 0, hma = 0, hmb = 0, seena = 0, seenb = 0, bad = 0, nma = 0, nmb = 0;
#pragma unroll
   for (int k = 8 - WU; k < 8; k++)
      pair_warm4x2(sa, stream_from_prev_lane(pair_word8(v[NQ - 2], v[NQ - 1], k), halo_nl ? 0x0A0A0A0Au : (uint32_t)__builtin_amdgcn_readlane((int)halo, k)),
                        sb, pair_word8(v[NQ / 2 - 2], v[NQ / 2 - 1], k), seena, seenb);
#pragma unroll
   for (int q = 0; q < NQ / 2; q++) {
      pair_own4x2<CHK>(sa, v[q].x, hma, nma, sb, v[q + NQ / 2].x, hmb, nmb, bad, true);
      pair_own4x2<CHK>(sa, v[q].y, hma, nma, sb, v[q + NQ / 2].y, hmb, nmb, bad, false);
      pair_own4x2<CHK>(sa, v[q].z, hma, nma, sb, v[q + NQ / 2].z, hmb, nmb, bad, true);
      pair_own4x2<CHK>(sa, v[q].w, hma, nma, sb, v[q + NQ / 2].w, hmb, nmb, bad, false);
      if (CHK && (q & 1)) { nmask[q >> 1] = nma; nmask[(q >> 1) + NM / 2] = nmb; nma = 0; nmb = 0; }
   }
   tile_bad = bad;
   /* first pair of a chain in bit 31; a walk that accepted during its warm-up: my first pairs are candidates (see the header) */
   hm[0] = __builtin_bitreverse32(hma) | ((seena & 1u) << 31);
   hm[1] = __builtin_bitreverse32(hmb) | ((seenb & 1u) << 31);
}

/* WU: warm-up dwords (4 .. 8); FA: FASTA input (header lines: see k_stream) */
/* IG (round 5): SQ_IGNORE on read-length lines.  A skipped byte stretches the text a match spans, so the walk over aliased bytes says nothing about
 * a line that holds one -- but every such line can be NAMED: an occurrence with <= tau errors takes at least m - tau characters that are not skipped,
 * all inside its line, so a line with a skipped byte and that many other characters is a candidate from its first byte to its end whatever the walk
 * said (a MARKER entry at the line's last character), a line with a skipped byte and fewer holds no occurrence, and a line without one is what it is
 * under SQ_FAIL: the walk's candidates, windows and all.  The counts come from bit masks of the skipped bytes and a segmented prefix sum over the lanes
 * (FusedArgs.ig_thr = m - tau); a line belongs to the tile it starts in, which reads the part of it that lies behind the tile (256 bytes) to count it
 * whole: FASTQ quality lines hold ~10 such characters in 150 and stay silent, headers and '+' lines hold none.  k_bounds2 (seeq_order.h) adds a
 * frequency bound on the pattern's most frequent base, and counts the rare line that was marked unseen (longer than a tile, or than the 256 bytes). */
/* LL (round 5): long lines -- the text of k_stream's long-line variant (a chromosome per line) on this walk: per-tile flags and the
 * segment's last newline for the window walk of the exact pass (seeq_exact1.h), EVERY flag of a chain kept (the walk jumps from
 * candidate to candidate: what lies between two of them is not scanned), no request for another kernel.  The candidates are this
 * kernel's as ever: every occurrence holds one or ends on the byte before one, and the exact pass scans m + tau + 1 either side. */
template <int WU, bool FA, bool IG = false, bool LL = false>
__global__ __launch_bounds__(64 * STREAM_NW, 8) void k_pair(FusedArgs a)
{
   static_assert(!(FA && IG), "SQ_IGNORE on FASTA input stays with k_stream");
   static_assert(!(LL && (FA || IG)), "the long-line variant serves plain text under SQ_FAIL / SQ_CONVERT");
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

   const uint32_t gwave = blockIdx.x * NW + wave, nwaves = gridDim.x * NW;
   /* round 4: the launch comes in two speeds box by box -- what clock does it run at?  Wave 0 of the grid reads the shader clock
      and the constant 100 MHz counter before and after its tiles (it works from the first tile to one of the last) */
   if (a.clk_probe != nullptr && blockIdx.x == 0 && tid == 0) { a.clk_probe[0] = __builtin_readcyclecounter(); a.clk_probe[1] = wall_clock64(); }
   uint32_t wv_lines = 0, wv_hitlines = 0, wv_hdrs = 0, slice_pos = 0;    /* wave-uniform */
   bool wv_overflow = false;
   uint32_t wv_dirty = 0, wv_lastnl = 0;
   uint32_t dmode = 0;                                    /* wave-uniform: a tile of mine failed the fast alphabet check (DM, below) */
   uint4 *slice = a.tmp + (size_t)gwave * a.slice_cap;
   const uint64_t lim = a.seg_base + a.seg_len;           /* bytes at or beyond it are not this segment's */
   const uint64_t last = a.nbytes - 1;

   /* persistent grid: wave w of the grid takes tiles w, w + waves, ....  (Requesting the next tile's text early was tried
      four ways -- piece by piece into the registers the walk has passed, all of it right after the walk, double-buffered
      in 128 registers at half the occupancy, and through LDS with global_load_lds_dwordx4 at 12 waves per CU: 1.06 / 0.92 /
      0.81 / 0.78 ms per launch against 0.77 - 0.78 without.  A lane's eight pieces lie inside one 128-byte line and only loads
      issued back to back are merged into one fetch of it; and with every load behind a wave-uniform branch awaited where the
      branch ends, or a spilled value reloaded through the same in-order counter, the early request waits anyway.  Without the
      bookkeeping the kernel runs at 0.645 ms per launch (5.9 TB/s; a plain read sweep of this layout: 6.2): DESIGN.md
      section 5; the variants are in the history -- tags r03-experiment-*, r05-before-prune.) */
   fused_v4u v[NQ];
   uint32_t halo = 0;                                     /* lanes 0..7: the eight words before the tile (lane 0's warm-up comes from them) */
   uint32_t tile = gwave;
   while (tile < a.ntiles) {
      const uint64_t t0 = a.seg_base + (uint64_t)tile * TB;
      const bool partial = tile + 1 == a.ntiles && (a.seg_len % TB) != 0;
      const uint32_t next = tile + nwaves;
      uint32_t lane_off = (uint32_t)lane * CH;
      asm volatile("" : "+v"(lane_off));                  /* (see k_stream: keeps the per-lane 64-bit addresses out of the loop-invariant set) */
      const uint64_t my = t0 + lane_off;
      {
         if (!partial) {                                  /* (all nine in one block, back to back: the eight pieces are merged into one fetch of each 128-byte line) */
            const uint8_t *p = a.text + my;
#pragma unroll
            for (int q = 0; q < NQ; q++) v[q] = *reinterpret_cast<const fused_v4u_unaligned *>(p + 16 * q);
            /* lanes 0..7: the eight words before the tile ('\n' where the buffer starts: chosen where they are used); the lane's
               offset is made afresh from its id -- kept alive from the top of the kernel it is spilled, and the reload waits for
               every load in flight */
            uint32_t lid;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lid));
            halo = *reinterpret_cast<const pair_u32_unaligned *>(a.text + (t0 >= 32 ? t0 - 32 : 0) + ((lid & 7u) << 2));
         } else {
#pragma unroll
            for (int q = 0; q < NQ; q++) v[q] = dfa_load16(a.text, my + 16 * q, lim);       /* '\n' beyond the segment */
            const fused_v4u h0 = dfa_load16(a.text, t0 >= 32 ? t0 - 32 : 0, lim), h1 = dfa_load16(a.text, t0 >= 32 ? t0 - 16 : 0, lim);
            halo = pair_word8(h0, h1, 0);
#pragma unroll
            for (int k = 1; k < 8; k++) halo = (lane & 7) == k ? pair_word8(h0, h1, k) : halo;
         }
      }
      const bool halo_nl = t0 < 32;                       /* the buffer starts here: lane 0 warms up over newlines */
      /* ---- the walk: chain A = bytes 0..63 (warm-up: the previous lane's last bytes), chain B = bytes 64..127; the
              alphabet check and the newline masks ride along -- until a tile fails the check: from then on this wave walks
              without them (DM, below) ---- */
      uint32_t hm[2], nmask[NM];
      if (!dmode) {
         uint32_t tile_bad;
         pair_walk<WU, true>(v, halo, halo_nl, hm, nmask, tile_bad);
         /* the fast check failed somewhere in the tile (wave-uniform): a byte that is not an upper-case base, N or a newline */
         const uint64_t badlanes = __ballot(tile_bad != 0);
         dmode = (uint32_t)__builtin_amdgcn_readfirstlane(badlanes != 0 ? 1 : 0);
         if (LL && lane == 0) { a.tile_dirty[tile] = badlanes != 0 ? 1u : 0u; a.tile_dmask[tile] = badlanes; }   /* (the window walk jumps over clean stretches only) */
      } else {
         uint32_t unused;
         pair_walk<WU, false>(v, halo, halo_nl, hm, nmask, unused);
         if (LL && lane == 0) { a.tile_dirty[tile] = 1u; a.tile_dmask[tile] = ~0ull; }      /* (no check in this mode: nothing of the tile counts as clean) */
      }
      /* DM (round 5): text with bytes outside { A C G T N \n } -- FASTQ quality lines, lower case.  Such a byte may alias onto the
         newline column ('+', ':', 'J' ...), so the newline masks are made exactly, from the registers, which still hold the tile
         (until round 5 the tile was fetched a second time, and text full of such bytes was kept off this kernel), and the scan is
         flagged: a byte that could end a line early (SQ_FAIL; SQ_CONVERT: a NUL) may sit between a line's start and a candidate's
         window, so the exact pass looks at that stretch before it trusts the window (verify_prefix_dirty, seeq_verify.h).  The flag
         is conservative -- lower-case bases set it too -- and costs the exact pass one look per candidate.  Where one tile of a wave
         is like that the next ones are too: the wave stays in this mode and its walks drop the fast check and the fast masks (6.5 of
         the walk's 21 VALU per text word), which pays for the exact masks -- FASTQ records run at the speed of clean reads. */
      if (dmode) {
         wv_dirty |= 1u;
#pragma unroll
         for (int r = 0; r < NM; r++) nmask[r] = stream_nl_mask32(v[2 * r], v[2 * r + 1], false);
      }
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
      uint32_t mk[2] = {0u, 0u};                          /* IG: pairs of hm[] that are line markers */
      const uint32_t walk_hm[2] = {hm[0], hm[1]};         /* IG: ... and what the walk itself flagged (a marker k_bounds2 drops may stand on a candidate: that one stays) */
      if (IG) {
         /* line ENDS of the tile: its newlines (the one in the buffer's very last byte included: it starts no line, but it ends one) and
            -- e_tail, tile-relative 1 .. TB -- the end of the buffer when no newline closes it / the byte behind the tile when it is a newline */
         uint32_t e_tail = 0;
         if (t0 <= last && last < t0 + TB) { if (a.text[last] != '\n') e_tail = (uint32_t)(last - t0) + 1u; }
         else if (!partial && a.text[t0 + TB] == '\n') e_tail = TB;
         e_tail = (uint32_t)__builtin_amdgcn_readfirstlane((int)e_tail);
         uint32_t vm[NM], sk[NM], dn[NM];
#pragma unroll
         for (int r = 0; r < NM; r++) {
            const uint32_t lo = 32u * r;
            vm[r] = valid <= lo ? 0u : (valid >= lo + 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> (valid - lo)));
            sk[r] = dmode ? pair_nonbase_mask32(v[2 * r], v[2 * r + 1]) & ~nmask[r] & vm[r] : 0u;      /* (a tile that passed the fast check holds no skipped byte) */
            dn[r] = ~(nmask[r] | sk[r]) & vm[r];
         }
         uint32_t cD[5], cS[5];
         cD[0] = 0; cS[0] = 0;
#pragma unroll
         for (int r = 0; r < NM; r++) { cD[r + 1] = cD[r] + (uint32_t)__popc(dn[r]); cS[r + 1] = cS[r] + (uint32_t)__popc(sk[r]); }
         /* a newline in my first byte ends the line of the lane before me: it is that lane's end at b = 128 */
         const uint32_t end_after = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(nmask[0] >> 31), 0x130, 0xf, 0xf, true);      /* wave_shl:1 -- lane 63: 0 */
         uint32_t b_tail = 0;                             /* a line end at lane-relative 1 .. 128 that no newline bit of mine stands for */
         if (e_tail && (uint32_t)lane == (e_tail - 1u) / CH) b_tail = e_tail - (uint32_t)lane * CH;
         if (end_after) b_tail = CH;
         /* My line ends, in order.  A line's counts are those since the end before it; the FIRST end of a lane closes a line that began in an earlier
            lane (or tile): its part in me is kept (first*) until the lanes before me have been summed up; every later line lies in me whole.
            (baseD / baseS: minus the counts at the end before.) */
         uint32_t baseD = 0, baseS = 0, seen_end = 0, firstP = 0, first_pq = 0;       /* firstP: bit 31 = there is one | skipped << 14 | characters */
         auto mark = [&](uint32_t pq) { if (pq < 32u) mk[0] |= 0x80000000u >> pq; else mk[1] |= 0x80000000u >> (pq - 32u); };
         auto line_end = [&](uint32_t b, uint32_t Dc, uint32_t Sc) {          /* Dc / Sc: characters / skipped bytes of mine in front of b */
            if (b >= 1u) {                                /* (b = 0: the lane before me has it as its b = 128) */
               const uint32_t lineD = Dc + baseD, lineS = Sc + baseS;
               if (!seen_end) { firstP = 0x80000000u | (lineS << 14) | lineD; first_pq = (b - 1u) >> 1; }
               else if (lineS != 0u && lineD >= a.ig_thr) mark((b - 1u) >> 1);
            }
            baseD = 0u - Dc; baseS = 0u - Sc; seen_end = 1u;
         };
#pragma unroll
         for (int r = 0; r < NM; r++) {
            uint32_t mm = nmask[r];
            while (mm) {
               const uint32_t lz = (uint32_t)__builtin_clz(mm);
               mm &= ~(0x80000000u >> lz);
               line_end(32u * r + lz, cD[r] + (uint32_t)__popc((dn[r] >> 1) >> (31u - lz)), cS[r] + (uint32_t)__popc((sk[r] >> 1) >> (31u - lz)));
            }
         }
         if (__ballot(b_tail != 0u)) {                    /* (wave-uniform; all but the buffer's own end stand behind the lane's last byte) */
            if (b_tail == CH) line_end(CH, cD[4], cS[4]);
            else if (b_tail) line_end(b_tail, pair_count_before(dn, cD, b_tail), pair_count_before(sk, cS, b_tail));
         }
         /* what the lanes behind me need: {a line ended in me, characters / skipped bytes behind my last line end} -- a segmented sum */
         uint32_t P = (seen_end << 31) | (cD[4] + baseD) | ((cS[4] + baseS) << 14);
#define PAIR_SEG(ctrl, rmask) { const uint32_t y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, ctrl, rmask, 0xf, true); P = (P >> 31) ? P : y + P; }
         PAIR_SEG(0x111, 0xf) PAIR_SEG(0x112, 0xf) PAIR_SEG(0x114, 0xf) PAIR_SEG(0x118, 0xf) PAIR_SEG(0x142, 0xa) PAIR_SEG(0x143, 0xc)
#undef PAIR_SEG
         /* A line belongs to the tile it STARTS in: the tile whose first byte follows a newline (or is the buffer's first) knows its first line
            from its start; a line that began before the tile was the business of the tile before (below: the tail) */
         const uint32_t starts_line = (uint32_t)__builtin_amdgcn_readfirstlane((t0 == 0 || (t0 >= 32 && ((uint32_t)__builtin_amdgcn_readlane((int)halo, 7) >> 24) == 0x0Au)) ? 1 : 0);
         const uint32_t before = stream_from_prev_lane(P, starts_line << 31);
         /* the lane's first end: + what the lanes before me bring.  Known when the line's start was seen: a line end in a lane before me, or the
            tile's own first byte (starts_line: every lane's sums then run from there -- the buffer's first line, found unmarked by
            profiles/ignore_fuzz.py, and every line that begins with its tile) */
         if ((firstP >> 31) && ((before >> 31) | starts_line)) {
            const uint32_t sum = (firstP & 0x0FFFFFFFu) + (before & 0x0FFFFFFFu);
            if ((sum >> 14) != 0u && (sum & 0x3FFFu) >= a.ig_thr) mark(first_pq);
         }
         /* The TAIL: the line that runs past my last byte (not the buffer's last tile, no line end right at the tile's end).  Its remainder is
            read from the text behind the tile -- 256 bytes, four per lane: a read-length line ends there -- and its marker, when it needs one,
            stands on the tile's last pair: inside the line, in front of whatever the next tile finds in it (repeats of the line).  A line that
            is longer, or that began before this tile as well, gets the marker unseen: k_bounds2 (seeq_order.h) reads it whole. */
         {
            const uint32_t tile_valid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(partial ? (uint32_t)(lim - t0) : TB));
            const uint32_t lv = (tile_valid - 1u) >> 7, bb = (tile_valid - 1u) & 127u;
            const uint32_t ng = (bb >> 5) == 0 ? nmask[0] : (bb >> 5) == 1 ? nmask[1] : (bb >> 5) == 2 ? nmask[2] : nmask[3];
            const uint32_t last_is_nl = (uint32_t)__builtin_amdgcn_readlane((int)((ng >> (31u - (bb & 31u))) & 1u), (int)lv);
            const bool buffer_last = t0 <= last && last < t0 + TB;
            if (!buffer_last && e_tail == 0u && !last_is_nl) {               /* (wave-uniform) */
               const uint32_t P63 = (uint32_t)__builtin_amdgcn_readlane((int)P, 63);
               const uint32_t started_here = (P63 >> 31) | starts_line;
               const uint64_t pk = t0 + tile_valid + 4u * (uint32_t)lane;
               uint32_t w = 0x0A0A0A0Au;                                       /* (behind the buffer: a line end) */
               if (pk + 4 <= a.nbytes) w = *reinterpret_cast<const pair_u32_unaligned *>(a.text + pk);
               else for (int i = 3; i >= 0; i--) w = (w << 8) | (pk + (uint64_t)i < a.nbytes ? (uint32_t)a.text[pk + i] : 0x0Au);
               const uint32_t nlf = nl_flags(w);
               const uint64_t nlb = __ballot(nlf != 0u);
               uint32_t qualifies = 1u;                                        /* unseen: the marker goes out, k_bounds2 decides */
               if (nlb != 0ull && started_here) {
                  const uint32_t F = (uint32_t)__builtin_ctzll(nlb);
                  const uint32_t nbytes_mine = (uint32_t)lane < F ? 4u : (uint32_t)lane == F ? ((uint32_t)__builtin_ctz(nlf) >> 3) : 0u;      /* my bytes in front of the newline */
                  const uint32_t keepm = nbytes_mine >= 4u ? 0xFFFFFFFFu : (1u << (8u * nbytes_mine)) - 1u;
                  const uint32_t y = (w ^ __builtin_amdgcn_perm(0x474E5554u, 0x43FF41FFu, w & 0x07070707u)) & keepm;      /* (as pair_nonbase_mask32) */
                  const uint32_t cs = (uint32_t)__popc((((y & 0x5F5F5F5Fu) + 0x7F7F7F7Fu) | y) & 0x80808080u);
                  const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32((nbytes_mine - cs) | (cs << 16)), 63);
                  const uint32_t lineD = (P63 & 0x3FFFu) + (tot & 0xFFFFu), lineS = ((P63 >> 14) & 0x3FFFu) + (tot >> 16);
                  qualifies = (lineS != 0u && lineD >= a.ig_thr) ? 1u : 0u;
               }
               if (qualifies && (uint32_t)lane == lv) mark(bb >> 1);
            }
         }
         hm[0] |= mk[0]; hm[1] |= mk[1];
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
      /* A chain that flags one line several times (the walk restarts when it accepts: every part of a partition filter
         reports on its own) keeps the FIRST and the LAST flag per line only: of the flags before the chain's first newline
         and of those behind its last one; between two newlines (short lines) all of them.  The exact pass scans a line
         from before its first candidate to behind its last one: what is dropped lies in between.  Wave-uniform, rare on
         read-length lines with a prefix automaton, the rule with a partition filter. */
      if (!LL && __ballot(((hm[0] & (hm[0] - 1u)) | (hm[1] & (hm[1] - 1u))) != 0)) {      /* (LL: the window walk scans around every candidate, not from the first to the last) */
#pragma unroll
         for (int x = 0; x < 2; x++) {
            const uint32_t n0 = nmask[2 * x], n1 = nmask[2 * x + 1], h = hm[x];
            /* first / last newline of the chain as byte offsets (64: none); pair q = bit 31 - q, its second byte = 2q + 1 */
            const uint32_t f = n0 ? (uint32_t)__builtin_clz(n0) : (n1 ? 32u + (uint32_t)__builtin_clz(n1) : 64u);
            const uint32_t l = n1 ? 63u - (uint32_t)__builtin_ctz(n1) : (n0 ? 31u - (uint32_t)__builtin_ctz(n0) : 64u);
            const uint32_t na = f >= 64u ? 32u : (f + 1u) >> 1;           /* pairs whose second byte is not behind the first newline */
            const uint32_t sc = f >= 64u ? 32u : (l + 1u) >> 1;           /* first pair whose second byte is behind the last newline */
            const uint32_t ma = na >= 32u ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> na);
            const uint32_t mc = sc >= 32u ? 0u : 0xFFFFFFFFu >> sc;
            const uint32_t ha = h & ma, hc = h & mc & ~ma;
            hm[x] = (h & ~ma & ~mc) | (ha ? (0x80000000u >> (uint32_t)__builtin_clz(ha)) | (ha & (0u - ha)) : 0u)
                                    | (hc ? (0x80000000u >> (uint32_t)__builtin_clz(hc)) | (hc & (0u - hc)) : 0u);
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
      if (tot_n && (tot_h || LL)) {                       /* wave-uniform */
         uint32_t my_last = 0;
#pragma unroll
         for (int r = 0; r < NM; r++)
            if (nmask[r]) my_last = (uint32_t)lane * CH + 32u * r + (31u - (uint32_t)__builtin_ctz(nmask[r])) + 2u;
         incl_last = wave_incl_max_u32(my_last);
         /* the segment's last newline decides which line runs on into the next segment (k_exact1) */
         if (LL) wv_lastnl = tile * TB + (uint32_t)__builtin_amdgcn_readlane((int)incl_last, 63) - 1u;
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
                  const uint32_t mkm = (r & 1) ? mk[r >> 1] << 16 : mk[r >> 1] & 0xFFFF0000u;
                  const uint32_t wkm = (r & 1) ? walk_hm[r >> 1] << 16 : walk_hm[r >> 1] & 0xFFFF0000u;
                  while (mm) {
                     const uint32_t lp = (uint32_t)__builtin_clz(mm);
                     mm &= ~(0x80000000u >> lp);
                     /* bit 30 of the entry's first word: a line marker; bit 29: ... on a pair the walk flagged as well */
                     const uint32_t isk = IG ? (((mkm << lp) >> 31) << 30) | ((((mkm & wkm) << lp) >> 31) << 29) : 0u;
                     uint32_t lz = 2u * lp + 1u;                            /* the pair's second byte, within the group */
                     if (partial && 32u * r + lz >= valid) lz = valid - 1u - 32u * r;      /* ... or its first, when the segment ends between them */
                     const uint32_t nlt = lz ? nmask[r] >> (32 - lz) : 0u;  /* newlines before it, same group */
                     const uint32_t nb = (uint32_t)__popc(nlt) - (FA && lz ? (uint32_t)__popc(dmask[r] >> (32 - lz)) : 0u);
                     const uint32_t st1 = nlt ? (uint32_t)lane * CH + 32u * r + lz - (uint32_t)__builtin_ctz(nlt) + 1u : before;
                     const uint32_t hp = (uint32_t)lane * CH + 32u * r + lz;           /* the candidate, tile-relative */
                     const uint32_t pos = st1 ? st1 - 1u : hp;
                     /* {tile | unresolved, rank | column of the candidate << 13, line start (or candidate) position, line rank} */
                     slice[slice_pos + ord] = make_uint4(tile | (st1 ? 0u : 0x80000000u) | isk, ord | ((hp - pos) << 13),
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
      if (!LL && tot_h && !tot_n && !partial && !(t0 <= last && last < t0 + TB)) wv_dirty |= 2u;
      wv_lines += tot_n + extra;
      wv_hdrs += tot_d + hd_extra;
      wv_hitlines += tot_h;
      tile = next;
   }
   if (lane == 0) {
      a.wg_hits[gwave] = wv_overflow ? 0u : slice_pos;
      a.wg_part[4 * gwave + 0] = wv_lines;
      a.wg_part[4 * gwave + 1] = wv_hdrs;
      if (LL) a.wg_lastnl[gwave] = wv_lastnl;    /* offset + 1 of the last newline this wave saw */
      a.wg_part[4 * gwave + 2] = wv_overflow ? (wv_hitlines | 0x80000000u) : wv_hitlines;
      a.wg_part[4 * gwave + 3] = wv_dirty;       /* 1: a byte outside the alphabet, 2: long-line input (k_fused_post acts on them) */
   }   if (a.clk_probe != nullptr && blockIdx.x == 0 && tid == 0) { a.clk_probe[2] = __builtin_readcyclecounter(); a.clk_probe[3] = wall_clock64(); }
}

#undef PAIR_X2

#endif
