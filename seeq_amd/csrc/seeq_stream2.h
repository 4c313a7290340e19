/*
 * seeq_stream2.h -- k_stream2: k_stream (seeq_stream.h) with every lane walking ONE LONG STRETCH of the text.
 *
 * k_stream gives a lane 128 consecutive bytes and pays, per chunk, a warm-up walk over the 24..32 bytes before it
 * (two chains per lane: 37.5 % more table gathers and VALU work than the text has characters).  Here a lane owns S
 * consecutive bytes (S = 1024: a wave owns a 64 KB tile) and walks them in S / 128 phases of 128 bytes -- one memory
 * line, eight back-to-back 16-byte loads, exactly k_stream's access shape per lane, only 1 KB instead of 128 B apart
 * (measured: the HBM rate is the same, profiles/microbench/stream_layout*) -- carrying the automaton state from phase
 * to phase.  One warm-up per KILOBYTE: 1.03 gathers per text byte.  The per-character code is k_stream's (one SDWA
 * xor, one 2-byte LDS gather, v_cmp + v_addc into the first-hit mask, SDWA byte compare + v_addc into the newline
 * mask), the table is the same (seeq_dfa.h).
 *
 * Bookkeeping changes shape with it.  A hit's line rank and line start depend on the newlines in LOWER lanes' stretches,
 * which the wave has not seen yet when it meets the hit (lane l - 1's last phase comes after lane l's first).  So a hit
 * is stored with what its lane knows -- {tile, lane, hits and newlines of the lane's stretch before it, the line start if
 * the stretch holds it} -- and at the end of a tile the wave leaves three values per lane (newlines and hits in lower
 * lanes, start of the line the lane's stretch begins in: three wave scans per 64 KB instead of per 8 KB);
 * k_stream2_reorder adds them while it orders the entries.  Everything behind that (k_stream_bounds, the exact pass) is
 * shared with k_stream.
 *
 * Serves read-length lines (no FASTA headers, no long-line bookkeeping: those stay on k_stream).  CHK / SUB as there.
 */
#ifndef SEEQ_STREAM2_H_
#define SEEQ_STREAM2_H_

#define STREAM2_S 1024                                       /* bytes per lane stretch */
#define STREAM2_TB (64u * STREAM2_S)                         /* tile bytes */

template <int WU, bool CHK, bool SUB>
__global__ __launch_bounds__(64 * STREAM_NW, 8) void k_stream2(FusedArgs a, uint32_t *lane_ws)
{
   constexpr int NW = STREAM_NW;
   constexpr int S = STREAM2_S;
   constexpr int NP = S / 128;                            /* phases */
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
   const uint32_t acc_new = a.dfa_final_base;              /* state value of ACC_NEW (seeq_dfa_build_stream) */
   const uint32_t ten = 0x0Au;

   const uint32_t gwave = blockIdx.x * NW + wave, nwaves = gridDim.x * NW;
   uint32_t wv_lines = 0, wv_hitlines = 0, slice_pos = 0;  /* wave-uniform */
   bool wv_overflow = false;
   uint32_t wv_dirty = 0;
   uint4 *slice = a.tmp + (size_t)gwave * a.slice_cap;
   const uint64_t lim = a.seg_base + a.seg_len;           /* bytes at or beyond it are not this segment's */
   const uint64_t last = a.nbytes - 1;

   for (uint32_t tile = gwave; tile < a.ntiles; tile += nwaves) {
      const uint64_t t0 = a.seg_base + (uint64_t)tile * TB;
      uint32_t lane_off = (uint32_t)lane * S;             /* opaque per tile: keeps 64-bit per-lane addresses out of the loop-invariant set */
      asm volatile("" : "+v"(lane_off));
      const uint64_t my = t0 + lane_off;
      const bool partial = t0 + TB > lim;                 /* wave-uniform */
      const bool has_last = t0 <= last && last < t0 + TB; /* wave-uniform: the buffer's last byte is in this tile */
      /* ---- warm-up over the 4 * WU bytes before my stretch, from the root state ('\n' where the buffer starts) ---- */
      uint32_t state = 0;
      {
         fused_v4u pa = fused_v4u{0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au}, pb = pa;
         if (my >= 32) {
            if (!partial) {
               pa = *reinterpret_cast<const fused_v4u_unaligned *>(a.text + my - 32);
               pb = *reinterpret_cast<const fused_v4u_unaligned *>(a.text + my - 16);
            } else {
               pa = dfa_load16(a.text, my - 32, lim);
               pb = dfa_load16(a.text, my - 16, lim);
            }
         }
         if (SUB) {
            pa.x = stream_sub4(pa.x); pa.y = stream_sub4(pa.y); pa.z = stream_sub4(pa.z); pa.w = stream_sub4(pa.w);
            pb.x = stream_sub4(pb.x); pb.y = stream_sub4(pb.y); pb.z = stream_sub4(pb.z); pb.w = stream_sub4(pb.w);
         }
         if (WU == 8) { stream_warm4(state, pa.x); stream_warm4(state, pa.y); }
         stream_warm4(state, pa.z); stream_warm4(state, pa.w);
         stream_warm4(state, pb.x); stream_warm4(state, pb.y); stream_warm4(state, pb.z); stream_warm4(state, pb.w);
      }
      /* what my stretch holds so far: newlines, hits, tile-relative offset + 1 of the line start behind my last newline */
      uint32_t nl_cnt = 0, hit_cnt = 0, last_nl1 = 0;
#pragma unroll 1
      for (int p = 0; p < NP; p++) {
         fused_v4u v[8];
         const uint64_t pbase = my + 128u * (uint32_t)p;
         if (!partial) {
            const uint8_t *q = a.text + pbase;
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = *reinterpret_cast<const fused_v4u_unaligned *>(q + 16 * i);
         } else {
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = dfa_load16(a.text, pbase + 16 * i, lim);      /* '\n' beyond the segment */
         }
         if (CHK) {
            uint32_t bad = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) bad |= fused_bad4(v[i].x) | fused_bad4(v[i].y) | fused_bad4(v[i].z) | fused_bad4(v[i].w);
            uint32_t flag = (uint32_t)__builtin_amdgcn_readfirstlane(__ballot(bad != 0) != 0 ? 1 : 0);
            if (SUB) {
               if (flag) {                                /* wave-uniform */
#pragma unroll
                  for (int i = 0; i < 8; i++) {
                     v[i].x = stream_sub4(v[i].x); v[i].y = stream_sub4(v[i].y); v[i].z = stream_sub4(v[i].z); v[i].w = stream_sub4(v[i].w);
                  }
               }
               flag = 0;                                  /* handled: only a newline (or NUL) ends a line now */
            }
            asm volatile("" : "+s"(flag));
            wv_dirty |= flag;
         }
         /* ---- the walk: four groups of 32 characters, one hit mask and one newline mask each (first character = bit 31) ---- */
         uint32_t hmask[4], nmask[4];
#pragma unroll
         for (int r = 0; r < 4; r++) {
            uint32_t hm = 0, nm = 0;
#pragma unroll
            for (int i = 2 * r; i < 2 * r + 2; i++) {
               stream_own4(state, v[i].x, hm, nm, acc_new, ten);
               stream_own4(state, v[i].y, hm, nm, acc_new, ten);
               stream_own4(state, v[i].z, hm, nm, acc_new, ten);
               stream_own4(state, v[i].w, hm, nm, acc_new, ten);
            }
            hmask[r] = hm; nmask[r] = nm;
         }
         const uint32_t pofs = lane_off + 128u * (uint32_t)p;                 /* tile-relative offset of these 128 bytes */
         if (partial) {                                   /* filler bytes are nobody's newlines */
            const uint32_t valid = lim > pbase ? (lim - pbase < 128 ? (uint32_t)(lim - pbase) : 128u) : 0u;
#pragma unroll
            for (int r = 0; r < 4; r++) {
               const uint32_t lo = 32u * r;
               const uint32_t keep = valid <= lo ? 0u : (valid >= lo + 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> (valid - lo)));
               nmask[r] &= keep; hmask[r] &= keep;
            }
         }
         if (has_last) {                                  /* a newline in the very last byte of the buffer starts no line */
            const uint32_t o = (uint32_t)(last - t0);
#pragma unroll
            for (int r = 0; r < 4; r++)
               if (o >= pofs + 32u * r && o < pofs + 32u * r + 32u) nmask[r] &= ~(0x80000000u >> (o - pofs - 32u * r));
         }
         uint32_t h0 = hmask[0], h1 = hmask[1], h2 = hmask[2], h3 = hmask[3];
         const uint32_t n0 = nmask[0], n1 = nmask[1], n2 = nmask[2], n3 = nmask[3];
         /* newlines before groups 1, 2, 3; start + 1 of the line behind the last newline of groups 0..r (0: none yet) */
         const uint32_t c0 = (uint32_t)__popc(n0), c1 = c0 + (uint32_t)__popc(n1), c2 = c1 + (uint32_t)__popc(n2);
         const uint32_t l0 = n0 ? pofs + (31u - (uint32_t)__builtin_ctz(n0)) + 2u : 0u;
         const uint32_t l1 = n1 ? pofs + 32u + (31u - (uint32_t)__builtin_ctz(n1)) + 2u : l0;
         const uint32_t l2 = n2 ? pofs + 64u + (31u - (uint32_t)__builtin_ctz(n2)) + 2u : l1;
         /* ---- hits of the phase (a lane rarely has one): one entry each into the wave's slice, no atomics ---- */
         uint32_t anyh = h0 | h1 | h2 | h3;
         while (__any(anyh != 0)) {
            const bool has = anyh != 0;
            const uint64_t bal = __ballot(has);
            const uint32_t nb = (uint32_t)__popcll(bal);
            const bool room = slice_pos + nb <= a.slice_cap;         /* wave-uniform */
            if (has) {
               const uint32_t r = h0 ? 0u : h1 ? 1u : h2 ? 2u : 3u;                 /* my first group with a hit left */
               const uint32_t mm = r == 0 ? h0 : r == 1 ? h1 : r == 2 ? h2 : h3;
               const uint32_t nmr = r == 0 ? n0 : r == 1 ? n1 : r == 2 ? n2 : n3;
               const uint32_t lz = (uint32_t)__builtin_clz(mm);
               const uint32_t keep = ~(0x80000000u >> lz);
               h0 = r == 0 ? h0 & keep : h0; h1 = r == 1 ? h1 & keep : h1; h2 = r == 2 ? h2 & keep : h2; h3 = r == 3 ? h3 & keep : h3;
               anyh = h0 | h1 | h2 | h3;
               if (room) {
                  const uint32_t nlt = lz ? nmr >> (32 - lz) : 0u;                  /* newlines before the hit, same group */
                  const uint32_t hp = pofs + 32u * r + lz;                          /* the hit, tile-relative */
                  const uint32_t lprev = r == 0 ? 0u : r == 1 ? l0 : r == 2 ? l1 : l2;   /* ... in the groups before, this phase */
                  /* start + 1 of the hit's line when my stretch holds it (0: it starts in a lower lane's stretch or before the tile) */
                  const uint32_t st1 = nlt ? hp - (uint32_t)__builtin_ctz(nlt) + 1u : (lprev ? lprev : last_nl1);
                  const uint32_t col = st1 ? hp - (st1 - 1u) : 0u;
                  const uint32_t cb = r == 0 ? 0u : r == 1 ? c0 : r == 2 ? c1 : c2;
                  const uint32_t slot = slice_pos + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                  /* {tile | line start unknown here, hits of my stretch before it | column << 16, hit position | lane << 16,
                     newlines of my stretch before it} */
                  slice[slot] = make_uint4(tile | (st1 ? 0u : 0x80000000u), hit_cnt | (col << 16), hp | ((uint32_t)lane << 16),
                                           nl_cnt + cb + (uint32_t)__popc(nlt));
               }
               hit_cnt++;
            }
            if (room) slice_pos += nb; else wv_overflow = true;
         }
         nl_cnt += c2 + (uint32_t)__popc(n3);
         const uint32_t l3 = n3 ? pofs + 96u + (31u - (uint32_t)__builtin_ctz(n3)) + 2u : l2;
         last_nl1 = l3 ? l3 : last_nl1;
      }
      /* ---- the tile is walked: three wave scans, three values per lane for k_stream2_reorder ---- */
      const uint32_t incl_h = wave_incl_scan_u32(hit_cnt), incl_n = wave_incl_scan_u32(nl_cnt);
      const uint32_t tot_h = (uint32_t)__builtin_amdgcn_readlane((int)incl_h, 63);
      const uint32_t tot_n = (uint32_t)__builtin_amdgcn_readlane((int)incl_n, 63);
      const uint32_t extra = (uint32_t)__builtin_amdgcn_readfirstlane((a.first_seg && tile == 0) ? 1 : 0);   /* the line starting at byte 0 */
      if (tot_h) {                                        /* wave-uniform */
         const uint32_t incl_last = wave_incl_max_u32(last_nl1);
         uint32_t before = stream_from_prev_lane(incl_last, 0u);           /* start + 1 of the line my stretch begins in */
         if (extra && before == 0) before = 1;                             /* ... the buffer starts here */
         uint32_t *lw = lane_ws + (size_t)tile * 192u;
         lw[lane] = incl_h - hit_cnt;                                      /* hits in lower lanes */
         lw[64 + lane] = incl_n - nl_cnt + extra - 1u;                     /* counted rank of the line my stretch begins in */
         lw[128 + lane] = before;
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
      a.wg_hits[gwave] = wv_overflow ? 0u : slice_pos;
      a.wg_part[4 * gwave + 0] = wv_lines;
      a.wg_part[4 * gwave + 1] = 0;
      a.wg_part[4 * gwave + 2] = wv_overflow ? (wv_hitlines | 0x80000000u) : wv_hitlines;
      a.wg_part[4 * gwave + 3] = wv_dirty;       /* 1: a byte outside the alphabet, 2: wants the long-line kernel (k_fused_post acts on them) */
   }
}

/* Slices -> ordered per-line arrays (tile_hits / tile_cl hold exclusive prefixes by now), finishing each entry with its
 * tile's per-lane values: position among the tile's hits, counted line rank, and -- when the lane's own stretch did not
 * hold it -- the line start found in a lower lane's stretch.  Same outputs as k_stream_reorder. */
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
         const uint32_t hp = e.z & 0xFFFFu, hl = e.z >> 16;
         const uint32_t *lw = lane_ws + (size_t)tile * 192u;
         const uint32_t dst = a.tile_hits[tile] + lw[hl] + (e.y & 0xFFFFu);
         uint32_t col = e.y >> 16, unres = e.x >> 31;
         if (unres) {
            const uint32_t before = lw[128 + hl];
            if (before) { col = hp - (before - 1u); unres = 0; }
         }
         hit_start[dst] = tile * STREAM2_TB + (unres ? hp : hp - col) + a.pos_bias;
         hit_line[dst] = (uint32_t)(c->lines + a.tile_cl[tile] + lw[64 + hl] + e.w + 1);     /* 1-based, reference seeq.c:377 */
         unresolved[dst] = unres;                            /* hit_start is the hit itself: the line starts before the tile */
         hit_col[dst] = unres ? 0u : col;
      }
   }
}

#endif
