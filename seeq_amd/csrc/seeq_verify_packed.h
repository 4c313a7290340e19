/*
 * seeq_verify_packed.h -- k_verify_packed<W, VAR>: k_verify (seeq_verify.h) reading its candidate windows straight from the PACKED
 * read batch (seeq_packed.h: four bases per byte, first base in bits 7-6; N as one bit per base), round 4.
 *
 * Until now the packed walk unpacked every candidate read into an ASCII staging text for the exact pass: 0.07 of its 0.38 ms per
 * 16 Mi reads, and 151 bytes written + read per candidate.  Here the window comes from the read itself: five big-endian words of the
 * base stream funnel-shifted to the window's first column, one byte of them = four columns = four 2-bit codes, and a code IS the
 * index of its EQ entry (a table of five entries per direction: A C T G by code, N) -- two VALU per column for the address instead
 * of the one SDWA shift of the ASCII form, no byte -> class table at all.  N is rare (one read in 256 here): the 64 N bits of the
 * window are fetched beside the bases, and a wave takes the phase-1 variant that patches the N entry in (+2 VALU per column) only
 * when one of its lanes has an N in its window (wave-uniform branch).  The line ends at column read_len: that column is the
 * terminator's step (the ASCII form meets its '\n' there), so stop_at = min(stop_at, read_len) and no terminator stream is kept.
 * Everything else -- windows, phase 2, the cache for k_emit1, the per-chunk scan of the counts -- is k_verify's.
 *
 * Served: one record per line at most or counts (VERIFY_ANY, VERIFY_BEST; VERIFY_ALL without records).  SQ_ALL records keep the
 * staging text: k_exact1<EMIT> recovers their starts from it.
 */
#ifndef SEEQ_VERIFY_PACKED_H_
#define SEEQ_VERIFY_PACKED_H_

struct VerifyPacked {
   const uint8_t *bases, *nmask;                          /* the whole batch (nmask may be NULL) */
   uint32_t stride, nstride, read_len;
   uint64_t total_bytes, ntotal_bytes;                    /* of bases / nmask */
};

typedef uint32_t vpk_u32_unaligned __attribute__((aligned(1)));

/* four bytes at byte `off` of `p` as a big-endian word (first byte on top); bytes outside [0, total) read as 0 */
__device__ __forceinline__ uint32_t vpk_be32(const uint8_t *p, int64_t off, uint64_t total)
{
   if (off >= 0 && (uint64_t)off + 4 <= total) return __builtin_bswap32(*reinterpret_cast<const vpk_u32_unaligned *>(p + off));
   uint32_t w = 0;
   for (int k = 0; k < 4; k++) {
      const int64_t q = off + k;
      w = (w << 8) | ((q >= 0 && (uint64_t)q < total) ? (uint32_t)p[q] : 0u);
   }
   return w;
}

/* hi:lo shifted left by s bits (0 .. 31), the top word */
__device__ __forceinline__ uint32_t vpk_funnel(uint32_t hi, uint32_t lo, uint32_t s) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (32u - s)); }

/* NT * 16 columns of read r from column `pos` on (pos may be negative: the columns in front of the read are garbage nobody uses):
   T[w] = columns 16 w .. 16 w + 15, first column in bits 31-30; Nb[w] = their N bits, first column in bit 31 (two words per 64 columns) */
template <int NT>
__device__ __forceinline__ void vpk_window(const VerifyPacked &pk, uint64_t r, int32_t pos, uint32_t (&T)[NT], uint32_t (&Nb)[(NT + 1) / 2])
{
   const int64_t c0 = (int64_t)(r * 4ull * pk.stride) + pos;      /* column of the base stream */
   const int64_t b0 = c0 >> 2;
   const uint32_t sh = 2u * (uint32_t)(c0 & 3);
   uint32_t S[NT + 1];
   /* all words of the window with one range test, the loads back to back (a test per word puts a wait behind every load: the kernel
      then spends its time in eight load latencies per window -- 135 against 70 us per 16 Mi reads) */
   if (b0 >= 0 && (uint64_t)b0 + 4u * (NT + 1) <= pk.total_bytes) {
      const uint8_t *p = pk.bases + b0;
#pragma unroll
      for (int w = 0; w <= NT; w++) S[w] = *reinterpret_cast<const vpk_u32_unaligned *>(p + 4 * w);
#pragma unroll
      for (int w = 0; w <= NT; w++) S[w] = __builtin_bswap32(S[w]);
   } else {
#pragma unroll
      for (int w = 0; w <= NT; w++) S[w] = vpk_be32(pk.bases, b0 + 4 * w, pk.total_bytes);
   }
#pragma unroll
   for (int w = 0; w < NT; w++) T[w] = vpk_funnel(S[w], S[w + 1], sh);
   constexpr int NN = (NT + 1) / 2;
   if (pk.nmask) {
      const int64_t n0 = (int64_t)(r * 8ull * pk.nstride) + pos;  /* bit of the N stream */
      const int64_t nb0 = n0 >> 3;
      const uint32_t nsh = (uint32_t)(n0 & 7);
      uint32_t M[NN + 1];
      if (nb0 >= 0 && (uint64_t)nb0 + 4u * (NN + 1) <= pk.ntotal_bytes) {
         const uint8_t *p = pk.nmask + nb0;
#pragma unroll
         for (int w = 0; w <= NN; w++) M[w] = *reinterpret_cast<const vpk_u32_unaligned *>(p + 4 * w);
#pragma unroll
         for (int w = 0; w <= NN; w++) M[w] = __builtin_bswap32(M[w]);
      } else {
#pragma unroll
         for (int w = 0; w <= NN; w++) M[w] = vpk_be32(pk.nmask, nb0 + 4 * w, pk.ntotal_bytes);
      }
#pragma unroll
      for (int w = 0; w < NN; w++) Nb[w] = vpk_funnel(M[w], M[w + 1], nsh);
   } else {
#pragma unroll
      for (int w = 0; w < NN; w++) Nb[w] = 0u;
   }
}

/* the four columns of byte x (codes first to last = bits 7-6 .. 1-0) -> LDS byte offsets of their EQ entries (entry = 4 << (W - 1)
   bytes; entry 4 = N where the column's N bit -- bit 3 - k of n4 -- is set, NP only) */
template <int W, bool NP>
__device__ __forceinline__ void vpk_addr4(uint32_t x, uint32_t n4, uint32_t &a0, uint32_t &a1, uint32_t &a2, uint32_t &a3)
{
   if (W == 1) { a0 = (x >> 4) & 0xCu; a1 = (x >> 2) & 0xCu; a2 = x & 0xCu; a3 = (x << 2) & 0xCu; }
   else { a0 = (x >> 3) & 0x18u; a1 = (x >> 1) & 0x18u; a2 = (x << 1) & 0x18u; a3 = (x << 3) & 0x18u; }
   if (NP) {
      const uint32_t an = W == 1 ? 16u : 32u;
      a0 = (n4 & 8u) ? an : a0; a1 = (n4 & 4u) ? an : a1; a2 = (n4 & 2u) ? an : a2; a3 = (n4 & 1u) ? an : a3;
   }
}

template <int W, int MODE>
__device__ __forceinline__ void vpk_cols4(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t eq_base, fused_state_t<W> &st, uint32_t lim,
                                          uint32_t &L, uint32_t &PH, uint32_t &MH, uint32_t &T)
{
   const fused_eq_t<W> e0 = fused_eq_load<W>(eq_base + a0), e1 = fused_eq_load<W>(eq_base + a1),
                       e2 = fused_eq_load<W>(eq_base + a2), e3 = fused_eq_load<W>(eq_base + a3);
   if (MODE != VERIFY_REV) {
      verify_col<W>::template run<MODE>(e0, st, lim, L, PH, MH, T);
      verify_col<W>::template run<MODE>(e1, st, lim, L, PH, MH, T);
      verify_col<W>::template run<MODE>(e2, st, lim, L, PH, MH, T);
      verify_col<W>::template run<MODE>(e3, st, lim, L, PH, MH, T);
   } else {
      verify_col<W>::template run<MODE>(e3, st, lim, L, PH, MH, T);
      verify_col<W>::template run<MODE>(e2, st, lim, L, PH, MH, T);
      verify_col<W>::template run<MODE>(e1, st, lim, L, PH, MH, T);
      verify_col<W>::template run<MODE>(e0, st, lim, L, PH, MH, T);
   }
}

/* one base of read r (column c < read_len): LDS byte offset of its EQ entry */
template <int W>
__device__ __forceinline__ uint32_t vpk_addr1(const VerifyPacked &pk, uint64_t r, uint32_t c)
{
   const uint32_t code = (pk.bases[r * (uint64_t)pk.stride + (c >> 2)] >> (6u - 2u * (c & 3u))) & 3u;
   const bool isn = pk.nmask && ((pk.nmask[r * (uint64_t)pk.nstride + (c >> 3)] >> (7u - (c & 7u))) & 1u);
   return (isn ? 4u : code) << (W == 1 ? 2 : 3);
}

/* Reverse start recovery (libseeq.c:289-316) of the lanes with `need`: the match ends before column i of read r, distance dist. */
template <int W>
__device__ __forceinline__ uint32_t vpk_reverse(const VerifyPacked &pk, bool need, uint64_t r, uint32_t i, uint32_t dist, uint32_t eqr_base, uint32_t m, uint32_t tau1)
{
   constexpr int NB = W == 1 ? 32 : 64;                   /* columns before the match end held in registers */
   constexpr int NT = NB / 16;
   uint32_t T[NT], Nb[(NT + 1) / 2];
   {
#pragma unroll
      for (int w = 0; w < NT; w++) T[w] = 0u;
#pragma unroll
      for (int w = 0; w < (NT + 1) / 2; w++) Nb[w] = 0u;
      if (need) vpk_window<NT>(pk, r, (int32_t)i - NB, T, Nb);
   }
   bool anyn = false;
#pragma unroll
   for (int w = 0; w < (NT + 1) / 2; w++) anyn = anyn || Nb[w] != 0u;
   fused_state_t<W> st;
   st.init(m);
   uint32_t Rh = 0, Rl = 0, dummy = 0;
   uint32_t nsteps = NB;                                  /* wave-uniform */
   const bool patch = __any(need && anyn) != 0;           /* (wave-uniform) */
#pragma unroll
   for (int g = 0; g < NB / 4; g++) {
      const uint32_t seen = g < 8 ? Rh : Rl;
      const uint32_t full = (g & 7) == 0 ? 0u : (1u << (4 * (g & 7))) - 1u;
      const bool more = need && (g < 8 || Rh == 0xFFFFFFFFu) && seen == full && 4u * g < i;
      if (!__any(more)) { nsteps = 4u * g; break; }
      const int q = NB / 4 - 1 - g;                       /* byte of the window, from its end */
      const uint32_t x = (T[q >> 2] >> (24 - 8 * (q & 3))) & 0xFFu;
      uint32_t a0, a1, a2, a3;
      if (patch) {
         const uint32_t n4 = (Nb[q >> 3] >> (28 - 4 * (q & 7))) & 0xFu;
         vpk_addr4<W, true>(x, n4, a0, a1, a2, a3);
      } else vpk_addr4<W, false>(x, 0u, a0, a1, a2, a3);
      vpk_cols4<W, VERIFY_REV>(a0, a1, a2, a3, eqr_base, st, dist, g < 8 ? Rh : Rl, dummy, dummy, dummy);
   }
   uint32_t start = 0;
   if (need) {
      const uint32_t nh_ = nsteps < 32u ? nsteps : 32u, nl_ = nsteps > 32u ? nsteps - 32u : 0u;
      const uint64_t F = ((uint64_t)verify_top(~Rh, nh_) << 32) | verify_top(~Rl, nl_);
      const uint32_t j = F ? (uint32_t)__builtin_clzll(F) + 1u : 0xFFFFFFFFu;
      if (j <= i) start = i - j;                          /* (libseeq.c:315 with last_d > d: jj = j) */
      else {
         /* not within this block / not before the read's first base: the literal loop of libseeq.c:289-316 (no byte of a read is skipped) */
         fused_state_t<W> s2;
         s2.init(m);
         uint32_t jj = 0, d = tau1, last_d;
         do {
            ++jj;
            const fused_eq_t<W> ev = fused_eq_load<W>(eqr_base + vpk_addr1<W>(pk, r, i - jj));
            last_d = d;
            s2.step(ev);
            d = s2.score < tau1 ? s2.score : tau1;
         } while (d > dist && jj < i);
         start = i - (last_d < d ? jj - 1u : jj);
      }
   }
   return start;
}

template <int W, int VAR>
__device__ __forceinline__ void verify_packed_body(const ScanArgs &a, const VerifyPacked &pk, const uint32_t *eq2, const uint32_t *hit_col, uint4 *cache)
{
   /* EQ entries by code: A C T G (code = bits 1-2 of the letter), N; forward and reversed pattern */
   __shared__ __align__(8) uint32_t s_eqf[8 * W];
   __shared__ __align__(8) uint32_t s_eqr[8 * W];
   __shared__ uint32_t s_wave[4];
   if (threadIdx.x < 5 * W) {
      const uint32_t e = threadIdx.x / W, w = threadIdx.x % W;
      const uint32_t byte = e == 0 ? 'A' : e == 1 ? 'C' : e == 2 ? 'T' : e == 3 ? 'G' : 'N';
      s_eqf[e * W + w] = eq2[byte * W + w];
      s_eqr[e * W + w] = eq2[256 * W + byte * W + w];
   }
   __syncthreads();
   const uint32_t eqf_base = (uint32_t)(uintptr_t)(fused_lds_cu32 *)s_eqf;
   const uint32_t eqr_base = (uint32_t)(uintptr_t)(fused_lds_cu32 *)s_eqr;
   Counters *c = a.cnt;
   const uint32_t nhl = c->seg_nhitlines;
   const uint32_t m = (uint32_t)a.m, tau = (uint32_t)a.tau, tau1 = tau + 1u;
   const bool caching = cache != nullptr && a.want == SEEQDEV_WANT_RECORDS;
   const uint32_t Lr = pk.read_len;
   const uint32_t stride = gridDim.x * 256u;
   const uint32_t kmax = (nhl + stride - 1u) / stride * stride;
   for (uint32_t k0 = blockIdx.x * 256u; k0 < kmax; k0 += stride) {
      const uint32_t k = k0 + threadIdx.x;
      bool done = k >= nhl;
      const uint64_t r = done ? 0ull : (uint64_t)a.hit_line[k] - 1ull;      /* the read (its line number is its index + 1) */
      uint32_t pos = 0, stop_at = Lr;
      if (!done) {
         const uint32_t col = hit_col[k];
         if (col > a.skip_back) pos = col - a.skip_back;
         if (a.window_ok) {
            const uint32_t lastcol = a.hit_last ? a.hit_last[k] : col;
            const uint32_t sa = lastcol + m + tau1 + 1u;
            stop_at = sa < Lr ? sa : Lr;
         }
      }
      fused_state_t<W> st;
      st.init(m);
      verify_rules<VAR> rl;
      rl.prevL = 0; rl.latch = 0; rl.nhits = 0; rl.best_d = tau1; rl.best_end = 0; rl.ce0 = 0; rl.ce1 = 0; rl.done = false;
      auto second = [&](uint32_t, uint32_t) {};            /* (VERIFY_ALL is served for counts only: no overflow lists) */
      while (__any(!done)) {
         /* ---- phase 1: up to 64 columns of the read, from its packed words ---- */
         uint32_t T[4], Nb[2];
         vpk_window<4>(pk, r, (int32_t)pos, T, Nb);
         const uint32_t stop_rel = done ? 0u : stop_at - pos;          /* the line / the window ends before this column of the block */
         const uint32_t s_in = st.score;
         uint32_t Lh = 0, Ll = 0, Ph = 0, Pl = 0, Mh = 0, Ml = 0, dummy = 0;
         uint32_t ncols = 64;                                           /* wave-uniform */
         if (__any(!done && (Nb[0] | Nb[1]) != 0u)) {                   /* an N in some lane's window: the patching variant */
#pragma unroll
            for (int g = 0; g < 16; g++) {
               if (!__any(stop_rel > 4u * g)) { ncols = 4u * g; break; }
               const uint32_t x = (T[g >> 2] >> (24 - 8 * (g & 3))) & 0xFFu, n4 = (Nb[g >> 3] >> (28 - 4 * (g & 7))) & 0xFu;
               uint32_t a0, a1, a2, a3;
               vpk_addr4<W, true>(x, n4, a0, a1, a2, a3);
               vpk_cols4<W, VERIFY_FWD>(a0, a1, a2, a3, eqf_base, st, tau, g < 8 ? Lh : Ll, g < 8 ? Ph : Pl, g < 8 ? Mh : Ml, dummy);
            }
         } else {
#pragma unroll
            for (int g = 0; g < 16; g++) {
               if (!__any(stop_rel > 4u * g)) { ncols = 4u * g; break; }
               const uint32_t x = (T[g >> 2] >> (24 - 8 * (g & 3))) & 0xFFu;
               uint32_t a0, a1, a2, a3;
               vpk_addr4<W, false>(x, 0u, a0, a1, a2, a3);
               vpk_cols4<W, VERIFY_FWD>(a0, a1, a2, a3, eqf_base, st, tau, g < 8 ? Lh : Ll, g < 8 ? Ph : Pl, g < 8 ? Mh : Ml, dummy);
            }
         }
         const uint32_t nh_ = ncols < 32u ? ncols : 32u, nl_ = ncols > 32u ? ncols - 32u : 0u;
         Lh = verify_top(~Lh, nh_); Ph = verify_top(Ph, nh_); Mh = verify_top(Mh, nh_);
         Ll = verify_top(~Ll, nl_); Pl = verify_top(Pl, nl_); Ml = verify_top(Ml, nl_);
         const uint32_t tcol = stop_rel < 64u ? stop_rel : 64u;         /* the only terminator: the end of the window / of the read */
         if (done) { Lh = 0; Ll = 0; rl.prevL = 0; }
         if (tcol < 32u) { Lh &= ~(0xFFFFFFFFu >> tcol); Ll = 0; }
         else if (tcol < 64u) Ll &= ~(0xFFFFFFFFu >> (tcol - 32u));
         /* ---- phase 2 ---- */
         rl.group(Lh, Ph, Mh, tcol, s_in, pos, second);
         if (__any(!rl.done && (Ll | rl.prevL) != 0u)) {
            const uint32_t s_mid = s_in + (uint32_t)__popc(Ph) - (uint32_t)__popc(Mh);
            if (!rl.done) rl.group(Ll, Pl, Ml, tcol - 32u, s_mid, pos + 32u, second);
         } else {
            if (VAR != VERIFY_BEST) rl.latch = 0u;
            rl.prevL = 0u;
         }
         if (tcol < 64u || rl.done) done = true;
         pos += done ? 0u : 64u;
      }
      uint32_t nhits = rl.nhits, ce0 = rl.ce0, ce1 = rl.ce1;
      if (VAR == VERIFY_BEST) { nhits = rl.best_d < tau1 ? 1u : 0u; ce0 = rl.best_end; ce1 = rl.best_d; }
      uint32_t ce2 = 0, ce3 = 0;
      if (VAR != VERIFY_ALL && caching) {
         const bool need = k < nhl && nhits != 0u;
         if (__any(need)) {
            const uint32_t s0 = vpk_reverse<W>(pk, need, r, ce0, ce1, eqr_base, m, tau1);
            if (need) { ce2 = s0; ce3 = 1u; }
         }
      }
      if (caching && k < nhl) cache[k] = make_uint4(ce0, ce1, ce2, ce3);
      uint32_t tot;
      const uint32_t ex = block_excl_scan(nhits, &tot, s_wave);
      if (k < nhl) a.nh[k] = ex;
      if (threadIdx.x == 0 && k0 < nhl) a.nh_sum[k0 >> 8] = tot;
      if (a.nz_sum) {
         const uint32_t nzw = (uint32_t)__popcll(__ballot(nhits != 0u));
         const uint32_t wave_id = threadIdx.x >> 6;
         if ((threadIdx.x & 63u) == 0) s_wave[wave_id] = nzw;
         __syncthreads();
         if (threadIdx.x == 0 && k0 < nhl) a.nz_sum[k0 >> 8] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
         __syncthreads();
      }
   }
   /* (k_emit1 reads no overflow lists; nothing to publish) */
}

template <int W, int VAR>
__global__ __launch_bounds__(256, W == 1 ? 6 : 5) void k_verify_packed(ScanArgs a, VerifyPacked pk, const uint32_t *eq2, const uint32_t *hit_col, uint4 *cache)
{
   verify_packed_body<W, VAR>(a, pk, eq2, hit_col, cache);
}

#endif
