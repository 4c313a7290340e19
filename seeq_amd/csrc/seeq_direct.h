/*
 * seeq_direct.h -- k_direct: the per-line bit-vector scan kernel (one line per lane, text in registers).
 *
 * One pass over the text does what the reference does per line in seeq.c:361-380 + libseeq.c:250-275 (getline,
 * newline strip, FASTA header skip, byte classification, per-character distance) for patterns of up to 62
 * positions (one or two 32-bit Myers words with two spare flag bits), plus the ordered hit-line compaction.
 * It serves what the table-driven k_stream (seeq_stream.h) cannot: patterns whose (filter) automaton does not
 * fit LDS or is not selective enough.
 *
 * Every WAVE works alone on a ~62-line region.  It reads the region once with coalesced 16-byte loads (HBM) only
 * to find the newlines (SWAR test + DPP wave scans, all in registers), then every lane loads ITS OWN line straight
 * into VGPRs, 160 characters (10 x dwordx4, all in flight together; L2 hits, the bytes were just fetched) and runs
 * the top-aligned Myers step per character from registers.  LDS only holds the 256-entry EQ table and 64..256 line
 * starts per wave, so occupancy is bounded by VGPRs (~100 -> 5 waves/SIMD) and the ~10-cycle dependent-issue
 * latency of the integer pipe is covered.  No workgroup barriers, no atomics.
 *
 * Lines of any length are handled by the same loop (one 160-character window after the other).
 */
#ifndef SEEQ_DIRECT_H_
#define SEEQ_DIRECT_H_

#define DIRECT_MAXRR   16      /* coalesced rounds of 1 KiB per region: region <= 16 KiB       */
#define DIRECT_WIN     10      /* 16-byte chunks of a line loaded into registers at once      */
#define DIRECT_SCAP    128     /* line starts kept in LDS per wave and pass                    */

/* 16 bytes at an arbitrary address; bytes at or beyond `nbytes` read as NUL.  The guarded branch only
 * runs for the last few lines of a buffer: a rolled byte loop keeps it small (no unrolled byte loads to
 * inflate the register budget of the hot path). */
__device__ __forceinline__ fused_v4u direct_load16(const uint8_t *text, uint64_t off, uint64_t nbytes)
{
   if (off + 16 <= nbytes) return *reinterpret_cast<const fused_v4u_unaligned *>(text + off);
   uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#pragma unroll 1
   for (int k = 15; k >= 0; k--) {
      const uint32_t b = off + (uint64_t)k < nbytes ? (uint32_t)text[off + k] : 0u;
      w3 = (w3 << 8) | (w2 >> 24);
      w2 = (w2 << 8) | (w1 >> 24);
      w1 = (w1 << 8) | (w0 >> 24);
      w0 = (w0 << 8) | b;
   }
   return fused_v4u{w0, w1, w2, w3};
}

template <int NW, int W>
__global__ __launch_bounds__(64 * NW, (W == 1 ? 4 : 3)) void k_direct(FusedArgs a)
{
   __shared__ __align__(8) uint32_t s_eq[256 * W];
   __shared__ uint32_t s_starts_all[NW][DIRECT_SCAP];

   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   uint32_t *s_starts = s_starts_all[wave];
   const bool fasta = (a.options & SEEQDEV_FASTA) != 0;
   const uint32_t tau = (uint32_t)a.tau;
   const uint32_t TB = a.tile_bytes;
   const uint32_t two = W == 1 ? 2u : 3u;                 /* log2 of the EQ entry size */

   for (int i = tid; i < 256 * W; i += 64 * NW) s_eq[i] = a.eqtab[i];
   __syncthreads();                                       /* the only barrier: tables are read-only from here */
   typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
   const uint32_t eq_base = (uint32_t)(uintptr_t)(lds_cu32 *)s_eq;   /* LDS byte offset of the table */

   const uint32_t gwave = blockIdx.x * NW + wave, nwaves = gridDim.x * NW;
   uint32_t wv_lines = 0, wv_hdrs = 0, wv_hitlines = 0, slice_pos = 0;
   bool wv_overflow = false;
   uint4 *slice = a.tmp + (size_t)gwave * a.slice_cap;
   const uint64_t last = a.nbytes - 1;

   for (uint32_t region = gwave; region < a.ntiles; region += nwaves) {
      const uint64_t t0 = a.seg_base + (uint64_t)region * TB;
      const uint32_t tb = (uint32_t)(((uint64_t)a.seg_len - (uint64_t)region * TB) < TB
                                     ? ((uint64_t)a.seg_len - (uint64_t)region * TB) : TB);
      /* ---- 1+2. coalesced read of the region; newline masks; ranks; line starts -> LDS ---- */
      /* One pass keeps DIRECT_SCAP starts; a region with more lines (very short lines) simply runs the
         pass again for the next window of ranks (the region is in L2 by then).  Nothing per lane has to
         stay in registers across the per-line scan. */
      const uint32_t rr = (tb + 1023) >> 10;                                /* <= DIRECT_MAXRR */
      const uint32_t extra = (a.first_seg && region == 0) ? 1u : 0u;        /* the line starting at byte 0 */
      const bool region_safe = t0 + (uint64_t)DIRECT_MAXRR * 1024 + 16 <= a.nbytes;   /* wave-uniform */
      const bool region_plain = region_safe && t0 + (uint64_t)tb <= last;              /* no owned byte is the last one */
      uint32_t nl = 0, reg_hdrs = 0, reg_hits = 0;
      /* opaque per iteration: keeps the compiler from hoisting 16 rounds of per-lane 64-bit addresses out
         of the region loop (loop-invariant code motion there costs ~60 VGPRs and forces spills) */
      for (uint32_t p0 = 0; p0 == 0 || p0 < nl; p0 += DIRECT_SCAP) {
         uint32_t lane16 = (uint32_t)lane * 16;
         asm volatile("" : "+v"(lane16));
         uint32_t running = extra;
         if (extra && p0 == 0 && lane == 0) s_starts[0] = 0;
#pragma unroll
         for (int rb = 0; rb < DIRECT_MAXRR; rb += 8) {
            if ((uint32_t)rb < rr) {                                        /* wave-uniform */
               fused_v4u pre[8];                                            /* 8 KiB of the region in flight per wave */
#pragma unroll
               for (int i = 0; i < 8; i++) {
                  const uint32_t q0 = (uint32_t)(rb + i) * 1024 + lane16;
                  if (region_safe) pre[i] = *reinterpret_cast<const fused_v4u_unaligned *>(a.text + t0 + q0);
                  else pre[i] = direct_load16(a.text, t0 + q0, a.nbytes);
               }
#pragma unroll
               for (int i = 0; i < 8; i++) {
                  const uint32_t r = (uint32_t)(rb + i);
                  const uint32_t q0 = r * 1024 + lane16;
                  if (r < rr) {                                             /* wave-uniform */
                     const fused_v4u v = pre[i];
                     /* 0x80 in every byte that is '\n' -- plus, possibly, in the byte right above one (borrow):
                        the LOWEST flag of a word is always a true newline */
                     const uint32_t x0 = v.x ^ 0x0A0A0A0Au, x1 = v.y ^ 0x0A0A0A0Au, x2 = v.z ^ 0x0A0A0A0Au, x3 = v.w ^ 0x0A0A0A0Au;
                     const uint32_t f0 = (x0 - 0x01010101u) & ~x0 & 0x80808080u, f1 = (x1 - 0x01010101u) & ~x1 & 0x80808080u;
                     const uint32_t f2 = (x2 - 0x01010101u) & ~x2 & 0x80808080u, f3 = (x3 - 0x01010101u) & ~x3 & 0x80808080u;
                     const uint32_t cflag = (uint32_t)__popc(f0) + (uint32_t)__popc(f1) + (uint32_t)__popc(f2) + (uint32_t)__popc(f3);
                     const bool edge = !region_plain || (r + 1) * 1024 > tb;  /* wave-uniform */
                     if (!edge && !__any(cflag > 1)) {
                        /* common case: at most one newline per 16-byte piece in the whole wave -> a ballot ranks them */
                        const bool has = cflag != 0;
                        const uint64_t bm = __ballot(has);
                        if (has) {
                           const uint32_t rk = running + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32),
                                                            __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0));
                           /* bit index of the single flag among the 128 bits (ffbl of an empty word = 0xFFFFFFFF) */
                           const uint32_t b0 = (uint32_t)__builtin_ctz(f0 | 0x80000000u) | (f0 ? 0u : 0xFFFFFF00u);
                           const uint32_t b1 = (32u + (uint32_t)__builtin_ctz(f1 | 0x80000000u)) | (f1 ? 0u : 0xFFFFFF00u);
                           const uint32_t b2 = (64u + (uint32_t)__builtin_ctz(f2 | 0x80000000u)) | (f2 ? 0u : 0xFFFFFF00u);
                           const uint32_t b3 = (96u + (uint32_t)__builtin_ctz(f3 | 0x80000000u)) | (f3 ? 0u : 0xFFFFFF00u);
                           const uint32_t bl = b0 < b1 ? b0 : b1, bh = b2 < b3 ? b2 : b3;
                           const uint32_t bit = bl < bh ? bl : bh;
                           if (rk >= p0 && rk < p0 + DIRECT_SCAP) s_starts[rk - p0] = q0 + (bit >> 3) + 1;
                        }
                        running += (uint32_t)__popcll(bm);
                     } else {
                        uint32_t m16 = 0;
                        if (cflag) {                                        /* exact per-byte mask */
                           const uint32_t g0 = nl_flags(v.x), g1 = nl_flags(v.y), g2 = nl_flags(v.z), g3 = nl_flags(v.w);
                           m16 = (((g0 >> 7) * 0x00204081u >> 21) & 0xFu) | ((((g1 >> 7) * 0x00204081u >> 21) & 0xFu) << 4) |
                                 ((((g2 >> 7) * 0x00204081u >> 21) & 0xFu) << 8) | ((((g3 >> 7) * 0x00204081u >> 21) & 0xFu) << 12);
                        }
                        if (edge) {                                         /* edges of the owned range */
                           if (q0 >= tb) m16 = 0;
                           else if (q0 + 16 > tb) m16 &= (1u << (tb - q0)) - 1u;   /* a newline must be owned: q < tb */
                           if (t0 + q0 <= last && last < t0 + q0 + 16)      /* ... and not the last byte       */
                              m16 &= ~(1u << (uint32_t)(last - (t0 + q0)));
                        }
                        const uint32_t c = (uint32_t)__popc(m16);
                        const uint32_t incl = wave_incl_scan_u32(c);
                        uint32_t rk = running + incl - c;                   /* rank of my first newline */
                        running += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                        while (m16) {
                           const uint32_t b = (uint32_t)__builtin_ctz(m16);
                           m16 &= m16 - 1;
                           if (rk >= p0 && rk < p0 + DIRECT_SCAP) s_starts[rk - p0] = q0 + b + 1;
                           rk++;
                        }
                     }
                  }
               }
            }
         }
         nl = running;                                                      /* raw lines owned by the region */
         const uint32_t npass = nl - p0 < DIRECT_SCAP ? nl - p0 : DIRECT_SCAP;
         __builtin_amdgcn_wave_barrier();
         /* ---- 3. one line per lane, text in registers ---- */
         for (uint32_t b0 = 0; b0 < npass; b0 += 64) {
            const uint32_t rl = b0 + lane;
            bool active = rl < npass;
            const uint32_t lstart = active ? s_starts[rl] : 0;             /* offset of the line inside the region */
            const uint64_t lbase = t0 + lstart;                            /* absolute offset */
            uint32_t ahead = 0;                                            /* bytes of my line requested so far */
            fused_state_t<W> st;
            st.init((uint32_t)a.m);
            uint32_t minscore = (uint32_t)a.m;
            bool hit = false, hdr = false;
            auto next_window = [&](fused_v4u (&v)[DIRECT_WIN]) {
               const uint64_t o = lbase + ahead;
               if (!__any(o + 16 * DIRECT_WIN > a.nbytes)) {
                  const uint8_t *p = a.text + o;
#pragma unroll
                  for (int c = 0; c < DIRECT_WIN; c++) v[c] = *reinterpret_cast<const fused_v4u_unaligned *>(p + 16 * c);
               } else {
#pragma unroll
                  for (int c = 0; c < DIRECT_WIN; c++) v[c] = direct_load16(a.text, o + 16 * c, a.nbytes);
               }
               if (active) ahead += 16 * DIRECT_WIN;
            };
            /* one 16-character chunk from registers: EQ lookups, flag test, 16 Myers steps */
            auto process = [&](const fused_v4u &q) {
               fused_eq_t<W> eq[16];
#pragma unroll
               for (int k = 0; k < 16; k += 4) {
                  const uint32_t word = k == 0 ? q.x : k == 4 ? q.y : k == 8 ? q.z : q.w;
                  uint32_t a0, a1, a2, a3;
                  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0"
                      : "=v"(a0) : "v"(two), "v"(word));
                  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"
                      : "=v"(a1) : "v"(two), "v"(word));
                  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2"
                      : "=v"(a2) : "v"(two), "v"(word));
                  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3"
                      : "=v"(a3) : "v"(two), "v"(word));
                  eq[k + 0] = fused_eq_load<W>(eq_base + a0);
                  eq[k + 1] = fused_eq_load<W>(eq_base + a1);
                  eq[k + 2] = fused_eq_load<W>(eq_base + a2);
                  eq[k + 3] = fused_eq_load<W>(eq_base + a3);
               }
               uint32_t fl[4];
#pragma unroll
               for (int g = 0; g < 4; g++)
                  fl[g] = (eq[4 * g].w0 | eq[4 * g + 1].w0 | eq[4 * g + 2].w0 | eq[4 * g + 3].w0) & FUSED_FLAGS;
               const uint32_t flall = fl[0] | fl[1] | fl[2] | fl[3];
               if (!__any(active && flall != 0)) {
#pragma unroll
                  for (int k = 0; k < 16; k++) {
                     st.step(eq[k]);
                     minscore = st.score < minscore ? st.score : minscore;
                  }
               } else {
#pragma unroll
                  for (int g = 0; g < 4; g++) {
                     if (!__any(active && fl[g] != 0)) {
#pragma unroll
                        for (int k = 4 * g; k < 4 * g + 4; k++) {
                           st.step(eq[k]);
                           minscore = st.score < minscore ? st.score : minscore;
                        }
                     } else {
#pragma unroll
                        for (int k = 4 * g; k < 4 * g + 4; k++) {
                           if (active) {
                              const uint32_t e = eq[k].w0;
                              if ((e & FUSED_FLAGS) == 0) {
                                 st.step(eq[k]);
                                 minscore = st.score < minscore ? st.score : minscore;
                              } else if (e & FUSED_FLAG_TERM) {
                                 active = false;                            /* line over: latch the verdict */
                                 hit = minscore <= tau;
                              }
                           }
                        }
                     }
                  }
               }
            };
            /* The next DIRECT_WIN x 16 characters of my line, all loads issued back to back right after the
               coalesced read of the region: they hit L2 (the lines were fetched microseconds ago) and L1
               merges the loads that fall into one cache line. */
            fused_v4u win[DIRECT_WIN];
            bool first_window = true;
            while (__any(active)) {
               next_window(win);
               if (first_window) {
                  if (fasta && active && (win[0].x & 0xFFu) == '>') { hdr = true; active = false; }
                  first_window = false;
               }
#pragma unroll
               for (int c = 0; c < DIRECT_WIN; c++)
                  if (__any(active)) process(win[c]);
            }
            /* ---- 4. ordered compaction: per-wave slice, no atomics ---- */
            const uint64_t hm = __ballot(hit), dm = __ballot(hdr);
            const uint32_t nh = (uint32_t)__popcll(hm);
            if (nh && a.want != SEEQDEV_WANT_COUNTLINES) {
               if (slice_pos + nh <= a.slice_cap) {
                  if (hit) {
                     const uint32_t below_h = __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32),
                                                 __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0));
                     const uint32_t below_d = __builtin_amdgcn_mbcnt_hi((uint32_t)(dm >> 32),
                                                 __builtin_amdgcn_mbcnt_lo((uint32_t)dm, 0));
                     const uint32_t crank = p0 + rl - (reg_hdrs + below_d);  /* counted rank inside the region */
                     const uint64_t start_seg = (uint64_t)region * TB + lstart;
                     slice[slice_pos + below_h] = make_uint4(region, reg_hits + below_h, (uint32_t)start_seg, crank);
                  }
                  slice_pos += nh;
               } else {
                  wv_overflow = true;
               }
            }
            reg_hits += nh;
            reg_hdrs += (uint32_t)__popcll(dm);
         }
         __builtin_amdgcn_wave_barrier();
      }
      if (lane == 0) {
         a.tile_cl[region] = nl - reg_hdrs;
         a.tile_hits[region] = reg_hits;
      }
      wv_lines += nl;
      wv_hdrs += reg_hdrs;
      wv_hitlines += reg_hits;
   }
   if (lane == 0) {
      a.wg_hits[gwave] = wv_overflow ? 0u : slice_pos;
      a.wg_part[4 * gwave + 0] = wv_lines;
      a.wg_part[4 * gwave + 1] = wv_hdrs;
      a.wg_part[4 * gwave + 2] = wv_overflow ? (wv_hitlines | 0x80000000u) : wv_hitlines;
      a.wg_part[4 * gwave + 3] = 0u;
   }
}

#endif
