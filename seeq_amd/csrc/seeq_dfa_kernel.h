/*
 * seeq_dfa_kernel.h -- k_dfa: the table-driven variant of the one-pass scan kernel.
 *
 * Same outputs as k_direct (seeq_direct.h); what differs is the per-character work.  Instead of a
 * bit-vector column update (12.5 VALU ops per character) the lane walks the COMPLETE Levenshtein
 * automaton of the pattern (seeq_dfa.h: the reference's DFA of saturated NW columns, reference
 * libseeq.c:698-842, built breadth first on the host, accepting states absorbed) held in LDS:
 *
 *        state = TABLE[ state | (byte & 0xE) ]          1.25 VALU ops + one 2-byte LDS gather
 *
 * The table row of a state is 16 bytes (8 columns selected by bits 1-3 of the text byte, which
 * separate A, C, G, T/U, N and '\n' in both cases); entries are row byte offsets, so the address is
 * an OR.  '\n' (and the two columns no DNA byte maps to) lead to two absorbing final rows, ACC_FINAL
 * / DEAD_FINAL, so a lane needs no line length: it is done when its state is a final row.
 * Only for SQ_FAIL + SQ_LINES (the default options): there every non-DNA byte ends the line, so the
 * aliasing of such bytes onto DNA columns can only add spurious hit lines (the exact pass verifies
 * every flagged line: k_exact1 in COUNT mode), never lose one.
 * One workgroup = 16 waves sharing one table (<= 64 KB); two workgroups per CU = 8 waves/SIMD.
 */
#ifndef SEEQ_DFA_KERNEL_H_
#define SEEQ_DFA_KERNEL_H_

#define DFA_NW 16

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

__global__ __launch_bounds__(64 * DFA_NW, 8) void k_dfa(FusedArgs a)
{
   constexpr int NW = DFA_NW;
   extern __shared__ __align__(16) uint8_t dsmem[];
   /* LDS: [0, nrows*16) transition table (row offsets are LDS addresses), then DIRECT_SCAP starts per wave */
   const uint32_t table_bytes = a.dfa_rows * 16;
   uint32_t *s_starts_base = reinterpret_cast<uint32_t *>(dsmem + ((table_bytes + 15) & ~15u));

   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   uint32_t *s_starts = s_starts_base + wave * DIRECT_SCAP;
   const bool fasta = (a.options & SEEQDEV_FASTA) != 0;
   const uint32_t TB = a.tile_bytes;
   {
      const fused_v4u *src = reinterpret_cast<const fused_v4u *>(a.dfa);
      for (uint32_t i = tid; i < a.dfa_rows; i += 64 * NW) reinterpret_cast<fused_v4u *>(dsmem)[i] = src[i];
   }
   __syncthreads();                                       /* the only barrier: the table is read-only from here */
   const uint32_t final_base = a.dfa_final_base, acc_final = a.dfa_final_base, dead_final = a.dfa_final_base + 16;
   typedef __attribute__((address_space(3))) const uint16_t lds_cu16;

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
         /* ---- 3. one line per lane: walk the automaton ---- */
         for (uint32_t b0 = 0; b0 < npass; b0 += 64) {
            const uint32_t rl = b0 + lane;
            const bool mine = rl < npass;
            const uint32_t lstart = mine ? s_starts[rl] : 0;               /* offset of the line inside the region */
            const uint64_t lbase = t0 + lstart;                            /* absolute offset */
            uint32_t ahead = 0;                                            /* bytes of my line requested so far */
            uint32_t state = mine ? 0u : dead_final;                       /* row offset; 0 = root */
            bool hdr = false;
            if (a.debug & 1u) state = dead_final;
            /* next 64 bytes of my line: four back-to-back 16-byte loads (L1 merges them) */
            auto next_block = [&](fused_v4u (&v)[4]) {
               const uint64_t o = lbase + ahead;
               if (!__any(o + 64 > a.nbytes)) {
                  const uint8_t *p = a.text + o;
#pragma unroll
                  for (int c = 0; c < 4; c++) v[c] = *reinterpret_cast<const fused_v4u_unaligned *>(p + 16 * c);
               } else {
#pragma unroll
                  for (int c = 0; c < 4; c++) v[c] = dfa_load16(a.text, o + 16 * c, a.nbytes);
               }
               if (state < final_base) ahead += 64;
            };
            /* 4 characters: pre-masked column offsets, then OR + gather per character */
            auto walk4 = [&](uint32_t word) {
               const uint32_t wm = word & 0x0E0E0E0Eu;
               uint32_t ad;
               asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0"
                   : "=v"(ad) : "v"(state), "v"(wm));
               state = *(lds_cu16 *)(uintptr_t)ad;
               asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"
                   : "=v"(ad) : "v"(state), "v"(wm));
               state = *(lds_cu16 *)(uintptr_t)ad;
               asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2"
                   : "=v"(ad) : "v"(state), "v"(wm));
               state = *(lds_cu16 *)(uintptr_t)ad;
               asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3"
                   : "=v"(ad) : "v"(state), "v"(wm));
               state = *(lds_cu16 *)(uintptr_t)ad;
            };
            fused_v4u cur[4], nxt[4];
            next_block(cur);
            if (fasta && mine && (cur[0].x & 0xFFu) == '>') { hdr = true; state = dead_final; }
            while (__any(state < final_base)) {
               next_block(nxt);
#pragma unroll
               for (int c = 0; c < 4; c++) {
                  if (__any(state < final_base)) {                          /* wave-uniform */
                     walk4(cur[c].x); walk4(cur[c].y); walk4(cur[c].z); walk4(cur[c].w);
                  }
               }
#pragma unroll
               for (int c = 0; c < 4; c++) cur[c] = nxt[c];
            }
            const bool hit = state == acc_final;
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
      a.wg_part[3 * gwave + 0] = wv_lines;
      a.wg_part[3 * gwave + 1] = wv_hdrs;
      a.wg_part[3 * gwave + 2] = wv_overflow ? (wv_hitlines | 0x80000000u) : wv_hitlines;
   }
}

#endif
