/*
 * seeq_stream2.h -- k_stream2: k_stream (seeq_stream.h) with every walk running over ONE LONG STRETCH of the text.
 *
 * k_stream gives a lane 128 consecutive bytes, walked as two chains of 64, and pays a warm-up walk over the 24..32
 * bytes before each chain: 37.5 % more table gathers and VALU work than the text has characters.  Here a lane owns S =
 * 1024 consecutive bytes (a wave owns a 64 KB tile) and walks them as two chains of 512 bytes -- chain A the first
 * half of the stretch, chain B the second -- that carry their automaton state from line to line.  The text still
 * arrives as whole 128-byte memory lines, eight back-to-back 16-byte loads per lane as in k_stream, only 1 KB instead of
 * 128 B apart (same HBM rate: profiles/microbench/stream_layout*).  A chain consumes a line in two half-steps of 64
 * characters; the two chains are staggered by one half-step, so that at any time a lane holds one fresh line (32
 * registers) and the second half of the other chain's line (16): 48 text registers feed two independent walks.
 * (Two chains over 64-byte half lines -- k_stream's register budget -- would fetch every line twice, a phase apart.)
 *
 *     half-step  0   1   2   3   4   5   6   7   8
 *     chain A   a0' a0" a1' a1" a2' a2" a3' a3"  -          ' / " = first / second half of a line
 *     chain B   w   b0' b0" b1' b1" b2' b2" b3' b3"         w = B's warm-up: the 64 bytes before its stretch
 *     loads     a0  b0  a1  b1  a2  b2  a3  b3
 *
 * One warm-up per 512 bytes, and chain B's falls into the half-step in which it would idle anyway: 1.03 walk steps per
 * text byte for chain A's 24..32-byte warm-up, plus one idle half-step in nine.  The per-character code is k_stream's.
 *
 * Bookkeeping changes shape with it.  A hit's line rank and line start depend on the newlines in EARLIER stretches,
 * which the wave has not seen yet when it meets the hit (the last half-step of one chain comes after the first of the
 * next).  So a hit is stored with what its chain knows -- {tile, chain, hits and newlines of the chain's stretch before
 * it, the line start if the stretch holds it} -- and at the end of a tile the wave leaves three values per chain
 * (newlines and hits in earlier stretches, start of the line the stretch begins in: three wave scans per 64 KB instead
 * of per 8 KB); k_stream2_reorder adds them while it orders the entries.  Everything behind that (k_stream_bounds, the
 * exact pass) is shared with k_stream.
 *
 * The hot kernel only takes whole tiles; the segment's last tile, when it is cut short or holds the buffer's last
 * byte, goes to a one-wave launch of the TAIL variant (guarded loads, filler masking), which keeps all of that out of
 * the hot kernel's registers.  Serves read-length lines without FASTA headers and without the long-line bookkeeping
 * (those stay on k_stream).  CHK / SUB as there.
 */
#ifndef SEEQ_STREAM2_H_
#define SEEQ_STREAM2_H_

#define STREAM2_S 1024                                       /* bytes per lane: two chains of 512 */
#define STREAM2_TB (64u * STREAM2_S)                         /* tile bytes */
#define STREAM2_NW 12                                        /* waves per workgroup: 2 workgroups = 6 waves per SIMD at <= 80 VGPRs */
#define STREAM2_WS 384                                       /* u32 of lane workspace per tile: 3 values x 128 chains */

/* What a chain knows about its stretch so far. */
struct stream2_chain_t {
   uint32_t nl;          /* newlines */
   uint32_t hits;        /* hits (first hits of lines, as this chain sees them) */
   uint32_t last1;       /* tile-relative offset + 1 of the line start behind its last newline; 0: none yet */
};

/* 64 characters of one chain are walked: hit masks h0 h1 / newline masks n0 n1 of its two groups of 32 (first
 * character = bit 31), pofs = tile-relative offset of the 64 bytes.  Stores the hits (rare), updates the chain. */
__device__ __forceinline__ void stream2_book(uint32_t h0, uint32_t h1, uint32_t n0, uint32_t n1, uint32_t pofs, stream2_chain_t &c,
                                             uint32_t vlane, uint32_t tile, uint4 *slice, uint32_t &slice_pos, uint32_t slice_cap,
                                             bool &overflow)
{
   const uint32_t c0 = (uint32_t)__popc(n0);
   const uint32_t l0 = n0 ? pofs + (31u - (uint32_t)__builtin_ctz(n0)) + 2u : 0u;      /* line start + 1 behind group 0's last newline */
   uint32_t anyh = h0 | h1;
   while (__any(anyh != 0)) {
      const bool has = anyh != 0;
      const uint64_t bal = __ballot(has);
      const uint32_t nb = (uint32_t)__popcll(bal);
      const bool room = slice_pos + nb <= slice_cap;         /* wave-uniform */
      if (has) {
         const bool g1 = h0 == 0;                            /* my first group with a hit left */
         const uint32_t mm = g1 ? h1 : h0, nmr = g1 ? n1 : n0;
         const uint32_t lz = (uint32_t)__builtin_clz(mm);
         const uint32_t keep = ~(0x80000000u >> lz);
         h0 = g1 ? h0 : h0 & keep; h1 = g1 ? h1 & keep : h1;
         anyh = h0 | h1;
         if (room) {
            const uint32_t nlt = lz ? nmr >> (32 - lz) : 0u;                  /* newlines before the hit, same group */
            const uint32_t hp = pofs + (g1 ? 32u : 0u) + lz;                  /* the hit, tile-relative */
            const uint32_t lprev = g1 && l0 ? l0 : c.last1;
            /* start + 1 of the hit's line when my stretch holds it (0: it starts in an earlier stretch or before the tile) */
            const uint32_t st1 = nlt ? hp - (uint32_t)__builtin_ctz(nlt) + 1u : lprev;
            const uint32_t col = st1 ? hp - (st1 - 1u) : 0u;
            const uint32_t slot = slice_pos + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            /* {tile | line start unknown here, hits of my stretch before it | column << 16, hit position | chain << 16,
               newlines of my stretch before it} */
            slice[slot] = make_uint4(tile | (st1 ? 0u : 0x80000000u), c.hits | (col << 16), hp | (vlane << 16),
                                     c.nl + (g1 ? c0 : 0u) + (uint32_t)__popc(nlt));
         }
         c.hits++;
      }
      if (room) slice_pos += nb; else overflow = true;
   }
   c.nl += c0 + (uint32_t)__popc(n1);
   const uint32_t l1 = n1 ? pofs + 32u + (31u - (uint32_t)__builtin_ctz(n1)) + 2u : l0;
   c.last1 = l1 ? l1 : c.last1;
}

/* 64 characters of both chains (four 16-byte pieces each), interleaved: two gathers in flight per lane. */
__device__ __forceinline__ void stream2_walk64x2(uint32_t &sa, const fused_v4u &a0, const fused_v4u &a1, const fused_v4u &a2, const fused_v4u &a3,
                                                 uint32_t &sb, const fused_v4u &b0, const fused_v4u &b1, const fused_v4u &b2, const fused_v4u &b3,
                                                 uint32_t &ha0, uint32_t &ha1, uint32_t &na0, uint32_t &na1,
                                                 uint32_t &hb0, uint32_t &hb1, uint32_t &nb0, uint32_t &nb1, uint32_t acc_new, uint32_t ten)
{
   ha0 = ha1 = na0 = na1 = hb0 = hb1 = nb0 = nb1 = 0;
   stream_own4x2(sa, a0.x, ha0, na0, sb, b0.x, hb0, nb0, acc_new, ten);
   stream_own4x2(sa, a0.y, ha0, na0, sb, b0.y, hb0, nb0, acc_new, ten);
   stream_own4x2(sa, a0.z, ha0, na0, sb, b0.z, hb0, nb0, acc_new, ten);
   stream_own4x2(sa, a0.w, ha0, na0, sb, b0.w, hb0, nb0, acc_new, ten);
   stream_own4x2(sa, a1.x, ha0, na0, sb, b1.x, hb0, nb0, acc_new, ten);
   stream_own4x2(sa, a1.y, ha0, na0, sb, b1.y, hb0, nb0, acc_new, ten);
   stream_own4x2(sa, a1.z, ha0, na0, sb, b1.z, hb0, nb0, acc_new, ten);
   stream_own4x2(sa, a1.w, ha0, na0, sb, b1.w, hb0, nb0, acc_new, ten);
   stream_own4x2(sa, a2.x, ha1, na1, sb, b2.x, hb1, nb1, acc_new, ten);
   stream_own4x2(sa, a2.y, ha1, na1, sb, b2.y, hb1, nb1, acc_new, ten);
   stream_own4x2(sa, a2.z, ha1, na1, sb, b2.z, hb1, nb1, acc_new, ten);
   stream_own4x2(sa, a2.w, ha1, na1, sb, b2.w, hb1, nb1, acc_new, ten);
   stream_own4x2(sa, a3.x, ha1, na1, sb, b3.x, hb1, nb1, acc_new, ten);
   stream_own4x2(sa, a3.y, ha1, na1, sb, b3.y, hb1, nb1, acc_new, ten);
   stream_own4x2(sa, a3.z, ha1, na1, sb, b3.z, hb1, nb1, acc_new, ten);
   stream_own4x2(sa, a3.w, ha1, na1, sb, b3.w, hb1, nb1, acc_new, ten);
}

/* masks of a 64-byte half line of which only the first `valid` bytes are text (TAIL: filler bytes are nobody's newlines) */
__device__ __forceinline__ void stream2_clip(uint32_t &h0, uint32_t &h1, uint32_t &n0, uint32_t &n1, uint32_t valid)
{
   const uint32_t k0 = valid >= 32 ? 0xFFFFFFFFu : (valid ? ~(0xFFFFFFFFu >> valid) : 0u);
   const uint32_t k1 = valid >= 64 ? 0xFFFFFFFFu : (valid > 32 ? ~(0xFFFFFFFFu >> (valid - 32)) : 0u);
   h0 &= k0; n0 &= k0; h1 &= k1; n1 &= k1;
}

template <int WU, bool CHK, bool SUB, bool TAIL>
__global__ __launch_bounds__(64 * STREAM2_NW, STREAM2_NW / 2) void k_stream2(FusedArgs a, uint32_t *lane_ws, uint32_t first_tile, uint32_t end_tile,
                                                                uint32_t slice0)
{
   constexpr int NW = STREAM2_NW;
   constexpr int S = STREAM2_S;
   constexpr int NL = S / 256;                            /* memory lines per chain */
   constexpr uint32_t TB = STREAM2_TB;
   static_assert(WU == 6 || WU == 8, "warm-up is 24 or 32 bytes");
   extern __shared__ __align__(16) uint8_t dsmem[];
   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   {
      const fused_v4u *src = reinterpret_cast<const fused_v4u *>(a.dfa);
      for (uint32_t i = tid; i < a.dfa_rows; i += 64 * NW) reinterpret_cast<fused_v4u *>(dsmem)[i] = src[i];
   }
   __syncthreads();                                       /* the only barrier: the table is read-only from here */
   if (TAIL && wave != 0) return;                         /* the tail is one tile: one wave */
   const uint32_t acc_new = a.dfa_final_base;              /* state value of ACC_NEW (seeq_dfa_build_stream) */
   const uint32_t ten = 0x0Au;

   const uint32_t gwave = blockIdx.x * NW + wave, nwaves = TAIL ? 1u : gridDim.x * NW;
   uint32_t wv_lines = 0, wv_hitlines = 0, slice_pos = 0;  /* wave-uniform */
   bool wv_overflow = false;
   uint32_t wv_dirty = 0;
   const uint32_t my_slice = slice0 + gwave;
   uint4 *slice = a.tmp + (size_t)my_slice * a.slice_cap;
   const uint64_t lim = a.seg_base + a.seg_len;           /* bytes at or beyond it are not this segment's */
   const uint64_t last = a.nbytes - 1;
   const fused_v4u nlv = fused_v4u{0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au};

   /* one 16-byte piece: plain in the hot kernel, guarded ('\n' at or beyond the segment's end) in the tail */
   auto ld = [&](uint64_t off) -> fused_v4u {
      if (TAIL) return dfa_load16(a.text, off, lim);
      return *reinterpret_cast<const fused_v4u_unaligned *>(a.text + off);
   };
   /* alphabet check of 128 bytes; SUB: walk a corrected copy */
   auto check = [&](fused_v4u (&v)[8]) {
      if (!CHK) return;
      uint32_t bad = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) bad |= fused_bad4(v[i].x) | fused_bad4(v[i].y) | fused_bad4(v[i].z) | fused_bad4(v[i].w);
      uint32_t flag = (uint32_t)__builtin_amdgcn_readfirstlane(__ballot(bad != 0) != 0 ? 1 : 0);
      if (SUB) {
         if (flag) {                                      /* wave-uniform */
#pragma unroll
            for (int i = 0; i < 8; i++) {
               v[i].x = stream_sub4(v[i].x); v[i].y = stream_sub4(v[i].y); v[i].z = stream_sub4(v[i].z); v[i].w = stream_sub4(v[i].w);
            }
         }
         flag = 0;                                        /* handled: only a newline (or NUL) ends a line now */
      }
      asm volatile("" : "+s"(flag));
      wv_dirty |= flag;
   };

   for (uint32_t tile = first_tile + gwave; tile < end_tile; tile += nwaves) {
      const uint64_t t0 = a.seg_base + (uint64_t)tile * TB;
      uint32_t lane_off = (uint32_t)lane * S;             /* opaque per tile: keeps 64-bit per-lane addresses out of the loop-invariant set */
      asm volatile("" : "+v"(lane_off));
      const uint64_t my = t0 + lane_off;                  /* chain A's stretch; chain B's starts S / 2 further on */
      const uint32_t va = 2u * (uint32_t)lane, vb = va + 1u;
      /* TAIL: bytes of a 64-byte half line at tile offset `o` that are text of this segment */
      auto valid64 = [&](uint32_t o) -> uint32_t {
         const uint64_t p = t0 + o;
         return lim > p ? (lim - p < 64 ? (uint32_t)(lim - p) : 64u) : 0u;
      };
      auto drop_last = [&](uint32_t &n0, uint32_t &n1, uint32_t o) {      /* a newline in the very last byte of the buffer starts no line */
         if (t0 <= last && last < t0 + TB) {
            const uint32_t q = (uint32_t)(last - t0);
            if (q >= o && q < o + 32) n0 &= ~(0x80000000u >> (q - o));
            if (q >= o + 32 && q < o + 64) n1 &= ~(0x80000000u >> (q - o - 32));
         }
      };
      /* ---- chain A warms up over the 4 * WU bytes before its stretch, from the root state ('\n' where the buffer starts) ---- */
      uint32_t sa = 0, sb = 0;
      {
         fused_v4u pa = nlv, pb = nlv;
         if (my >= 32) { pa = ld(my - 32); pb = ld(my - 16); }
         if (SUB) {
            pa.x = stream_sub4(pa.x); pa.y = stream_sub4(pa.y); pa.z = stream_sub4(pa.z); pa.w = stream_sub4(pa.w);
            pb.x = stream_sub4(pb.x); pb.y = stream_sub4(pb.y); pb.z = stream_sub4(pb.z); pb.w = stream_sub4(pb.w);
         }
         if (WU == 8) { stream_warm4(sa, pa.x); stream_warm4(sa, pa.y); }
         stream_warm4(sa, pa.z); stream_warm4(sa, pa.w);
         stream_warm4(sa, pb.x); stream_warm4(sa, pb.y); stream_warm4(sa, pb.z); stream_warm4(sa, pb.w);
      }
      stream2_chain_t ca = {0, 0, 0}, cb = {0, 0, 0};
      /* B's pending half line; at first: its warm-up, the 64 bytes before its stretch (chain A's last) */
      fused_v4u q0 = ld(my + S / 2 - 64), q1 = ld(my + S / 2 - 48), q2 = ld(my + S / 2 - 32), q3 = ld(my + S / 2 - 16);
      if (SUB) {
         fused_v4u *qq[4] = {&q0, &q1, &q2, &q3};
#pragma unroll
         for (int i = 0; i < 4; i++) { qq[i]->x = stream_sub4(qq[i]->x); qq[i]->y = stream_sub4(qq[i]->y); qq[i]->z = stream_sub4(qq[i]->z); qq[i]->w = stream_sub4(qq[i]->w); }
      }
#pragma unroll 1
      for (int j = 0; j < NL; j++) {
         const uint32_t oa = lane_off + 128u * (uint32_t)j, ob = lane_off + S / 2 + 128u * (uint32_t)j;     /* tile-relative: A's line, B's line */
         uint32_t ha0, ha1, na0, na1, hb0, hb1, nb0, nb1;
         /* ---- even half-step: A's line j arrives; A walks its first half, B the pending half (line j - 1, or its warm-up) ---- */
         fused_v4u v[8];
#pragma unroll
         for (int i = 0; i < 8; i++) v[i] = ld(t0 + oa + 16 * i);
         check(v);
         stream2_walk64x2(sa, v[0], v[1], v[2], v[3], sb, q0, q1, q2, q3, ha0, ha1, na0, na1, hb0, hb1, nb0, nb1, acc_new, ten);
         if (TAIL) { stream2_clip(ha0, ha1, na0, na1, valid64(oa)); drop_last(na0, na1, oa); }
         stream2_book(ha0, ha1, na0, na1, oa, ca, va, tile, slice, slice_pos, a.slice_cap, wv_overflow);
         if (j > 0) {                                     /* (j = 0: that was B's warm-up -- nothing of it counts) */
            if (TAIL) { stream2_clip(hb0, hb1, nb0, nb1, valid64(ob - 64)); drop_last(nb0, nb1, ob - 64); }
            stream2_book(hb0, hb1, nb0, nb1, ob - 64, cb, vb, tile, slice, slice_pos, a.slice_cap, wv_overflow);
         }
         /* ---- odd half-step: B's line j arrives; A walks its second half, B its first ---- */
         fused_v4u w[8];
#pragma unroll
         for (int i = 0; i < 8; i++) w[i] = ld(t0 + ob + 16 * i);
         check(w);
         stream2_walk64x2(sa, v[4], v[5], v[6], v[7], sb, w[0], w[1], w[2], w[3], ha0, ha1, na0, na1, hb0, hb1, nb0, nb1, acc_new, ten);
         if (TAIL) { stream2_clip(ha0, ha1, na0, na1, valid64(oa + 64)); drop_last(na0, na1, oa + 64); }
         stream2_book(ha0, ha1, na0, na1, oa + 64, ca, va, tile, slice, slice_pos, a.slice_cap, wv_overflow);
         if (TAIL) { stream2_clip(hb0, hb1, nb0, nb1, valid64(ob)); drop_last(nb0, nb1, ob); }
         stream2_book(hb0, hb1, nb0, nb1, ob, cb, vb, tile, slice, slice_pos, a.slice_cap, wv_overflow);
         q0 = w[4]; q1 = w[5]; q2 = w[6]; q3 = w[7];
      }
      /* ---- the last half-step: B's pending half; A has nothing left and walks newlines ---- */
      {
         uint32_t ha0, ha1, na0, na1, hb0, hb1, nb0, nb1;
         uint32_t idle = sa;
         const uint32_t ob = lane_off + S - 64;
         stream2_walk64x2(idle, nlv, nlv, nlv, nlv, sb, q0, q1, q2, q3, ha0, ha1, na0, na1, hb0, hb1, nb0, nb1, acc_new, ten);
         if (TAIL) { stream2_clip(hb0, hb1, nb0, nb1, valid64(ob)); drop_last(nb0, nb1, ob); }
         stream2_book(hb0, hb1, nb0, nb1, ob, cb, vb, tile, slice, slice_pos, a.slice_cap, wv_overflow);
      }
      /* ---- the tile is walked: three wave scans over the 128 chains (A of lane 0, B of lane 0, A of lane 1, ...) ---- */
      const uint32_t pair_h = ca.hits + cb.hits, pair_n = ca.nl + cb.nl;
      const uint32_t incl_h = wave_incl_scan_u32(pair_h), incl_n = wave_incl_scan_u32(pair_n);
      const uint32_t tot_h = (uint32_t)__builtin_amdgcn_readlane((int)incl_h, 63);
      const uint32_t tot_n = (uint32_t)__builtin_amdgcn_readlane((int)incl_n, 63);
      const uint32_t extra = (uint32_t)__builtin_amdgcn_readfirstlane((a.first_seg && tile == 0) ? 1 : 0);   /* the line starting at byte 0 */
      if (tot_h) {                                        /* wave-uniform */
         const uint32_t pair_last = cb.last1 ? cb.last1 : ca.last1;          /* offsets grow with the chain index: the later one wins */
         const uint32_t incl_last = wave_incl_max_u32(pair_last);
         uint32_t before_a = stream_from_prev_lane(incl_last, 0u);           /* start + 1 of the line chain A's stretch begins in */
         if (extra && before_a == 0) before_a = 1;                           /* ... the buffer starts here */
         const uint32_t before_b = ca.last1 ? ca.last1 : before_a;
         uint32_t *lw = lane_ws + (size_t)tile * STREAM2_WS;
         const uint32_t eh = incl_h - pair_h, en = incl_n - pair_n + extra - 1u;
         *reinterpret_cast<uint2 *>(lw + 2 * lane) = make_uint2(eh, eh + ca.hits);                      /* hits in earlier chains */
         *reinterpret_cast<uint2 *>(lw + 128 + 2 * lane) = make_uint2(en, en + ca.nl);                  /* counted rank of the line the stretch begins in */
         *reinterpret_cast<uint2 *>(lw + 256 + 2 * lane) = make_uint2(before_a, before_b);
      }
      if (lane == 0) {
         a.tile_cl[tile] = tot_n + extra;
         a.tile_hits[tile] = tot_h;
      }
      if (tot_h && !tot_n) wv_dirty |= 2u;                /* a hit inside a line of >= a whole tile: ask for the long-line kernel */
      wv_lines += tot_n + extra;
      wv_hitlines += tot_h;
   }
   if (lane == 0) {
      a.wg_hits[my_slice] = wv_overflow ? 0u : slice_pos;
      a.wg_part[4 * my_slice + 0] = wv_lines;
      a.wg_part[4 * my_slice + 1] = 0;
      a.wg_part[4 * my_slice + 2] = wv_overflow ? (wv_hitlines | 0x80000000u) : wv_hitlines;
      a.wg_part[4 * my_slice + 3] = wv_dirty;    /* 1: a byte outside the alphabet, 2: wants the long-line kernel (k_fused_post acts on them) */
   }
}

/* Slices -> ordered per-line arrays (tile_hits / tile_cl hold exclusive prefixes by now), finishing each entry with its
 * tile's per-chain values: position among the tile's hits, counted line rank, and -- when the chain's own stretch did
 * not hold it -- the line start found in an earlier stretch.  Same outputs as k_stream_reorder. */
__global__ __launch_bounds__(256) void k_stream2_reorder(FusedArgs a, uint32_t nslices, const uint32_t *lane_ws, uint32_t *hit_start,
                                                         uint32_t *hit_line, uint32_t *unresolved, uint32_t *hit_col)
{
   const Counters *c = a.cnt;
   if (c->overflow & 2u) return;
   const uint32_t lane = threadIdx.x & 63;
   for (uint32_t sl = blockIdx.x * 4 + (threadIdx.x >> 6); sl < nslices; sl += gridDim.x * 4) {      /* one wave per slice */
      const uint32_t n = a.wg_hits[sl];
      const uint4 *slice = a.tmp + (size_t)sl * a.slice_cap;
      for (uint32_t i = lane; i < n; i += 64) {
         const uint4 e = slice[i];
         const uint32_t tile = e.x & 0x7FFFFFFFu;
         const uint32_t hp = e.z & 0xFFFFu, ch = e.z >> 16;
         const uint32_t *lw = lane_ws + (size_t)tile * STREAM2_WS;
         const uint32_t dst = a.tile_hits[tile] + lw[ch] + (e.y & 0xFFFFu);
         uint32_t col = e.y >> 16, unres = e.x >> 31;
         if (unres) {
            const uint32_t before = lw[256 + ch];
            if (before) { col = hp - (before - 1u); unres = 0; }
         }
         hit_start[dst] = tile * STREAM2_TB + (unres ? hp : hp - col) + a.pos_bias;
         hit_line[dst] = (uint32_t)(c->lines + a.tile_cl[tile] + lw[128 + ch] + e.w + 1);     /* 1-based, reference seeq.c:377 */
         unresolved[dst] = unres;                            /* hit_start is the hit itself: the line starts before the tile */
         hit_col[dst] = unres ? 0u : col;
      }
   }
}

#endif
