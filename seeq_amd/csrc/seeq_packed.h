/*
 * seeq_packed.h -- the scan of PACKED read batches (include/seeq_amd.h: seeqdev_packed_t): 2 bits per base, four bases per
 * byte, every read at a fixed stride -- the north star's "coalesced HBM loads of packed 2-bit read batches", 38 bytes per
 * 150 bp read instead of 151.
 *
 *   k_packed_walk   one READ per lane, 64 consecutive reads per wave: the lane loads its read (<= 64 bytes, 16-byte
 *                   loads) and walks k_pair's pair automaton over it straight from the root -- a read begins where its
 *                   line begins, so there is no warm-up, no newline, no alphabet check, and a nibble of the packed byte IS
 *                   the pair index: per text word two VALU instructions prepare eight steps (k_pair: two).  Per pair: the
 *                   SDWA v_xor, the gather, one v_alignbit.  Output: cand[r] = {first, last candidate column} + 1 or 0.
 *                   A base that is N is stored as some code and flagged in the optional N mask: an alias, as in k_pair.
 *   k_packed_stage  the candidate reads (5 % of a read set with planted hits) are written out as ASCII lines, N restored,
 *                   into a staging text of one line per candidate, with the hit-list arrays the exact pass reads:
 *                   from here on k_exact1 COUNT / EMIT run as behind k_pair -- windows, every match option, bit-exact.
 *
 * Results are those of the ASCII scan of the same reads, one per line (tests/test_gpu_packed.py against the oracle).
 */
#ifndef SEEQ_PACKED_H_
#define SEEQ_PACKED_H_

struct PackedArgs {
   const uint8_t *bases;        /* device; read r at bases + r * stride */
   const uint8_t *nmask;        /* device or NULL; read r at nmask + r * nstride, first base = bit 7 */
   uint64_t       first;        /* first read of this segment */
   uint32_t       nreads;       /* reads of this segment */
   uint32_t       read_len, stride, nstride;
   uint64_t       total_bytes;  /* of `bases`: reads of the whole batch x stride */
   const uint16_t *dfa;         /* pair table (seeq_dfa.h section 3) */
   uint32_t       dfa_units;    /* 16-byte units of it */
   uint32_t      *cand;         /* [nreads] */
   const uint32_t *coff;        /* [nreads] exclusive prefix of (cand != 0) */
   uint8_t       *stage;        /* [cap * (read_len + 1)] ASCII lines of the candidates */
   uint32_t      *hit_start, *hit_line, *hit_col, *hit_last, *nh;
   uint32_t       cap;          /* candidates the staging text and the hit list hold */
   uint64_t       line_base;    /* counted lines before this segment */
   Counters      *cnt;
};

/* eight pairs = one word of packed text: bytes in text order, high nibble first */
#define PACKED_STEP(T, K) \
   asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #K : "=v"(ad) : "v"(st), "v"(T)); \
   st = *(stream_lds_cu16 *)(uintptr_t)ad; \
   hm = __builtin_amdgcn_alignbit(st, hm, 1);

__device__ __forceinline__ void packed_word(uint32_t w, uint32_t &st, uint32_t &hm)
{
   const uint32_t thi = (w >> 3) & 0x1E1E1E1Eu, tlo = (w << 1) & 0x1E1E1E1Eu;      /* {first code, second code} << 1 of both nibbles of every byte */
   uint32_t ad;
   PACKED_STEP(thi, 0) PACKED_STEP(tlo, 0) PACKED_STEP(thi, 1) PACKED_STEP(tlo, 1)
   PACKED_STEP(thi, 2) PACKED_STEP(tlo, 2) PACKED_STEP(thi, 3) PACKED_STEP(tlo, 3)
}

__global__ __launch_bounds__(64 * STREAM_NW, 8) void k_packed_walk(PackedArgs a)
{
   extern __shared__ __align__(16) uint8_t dsmem[];
   const int tid = threadIdx.x, lane = tid & 63;
   {
      const fused_v4u *src = reinterpret_cast<const fused_v4u *>(a.dfa);
      for (uint32_t i = tid; i < a.dfa_units; i += 64 * STREAM_NW) reinterpret_cast<fused_v4u *>(dsmem)[i] = src[i];
   }
   __syncthreads();
   const uint32_t gwave = blockIdx.x * STREAM_NW + (uint32_t)(tid >> 6), nwaves = gridDim.x * STREAM_NW;
   const uint32_t nb = (a.read_len + 3) >> 2;              /* packed bytes of a read */
   const uint32_t nwords = (nb + 3) >> 2;                  /* <= 16 */
   const uint32_t npairs = (a.read_len + 1) >> 1;
   const uint32_t nblocks = (a.nreads + 63) >> 6;
   for (uint32_t blk = gwave; blk < nblocks; blk += nwaves) {
      const uint32_t r = blk * 64 + (uint32_t)lane;
      const bool live = r < a.nreads;
      const uint64_t roff = (a.first + (live ? r : 0)) * (uint64_t)a.stride;
      uint32_t st = 0, first = 0xFFFFFFFFu, last = 0;
      /* 16 bytes at a time: four words, 32 pairs, one mask (bytes beyond the read belong to the next read or to the
         padding: their steps are walked and their flags dropped) */
#pragma unroll 1
      for (uint32_t j = 0; j < nwords; j += 4) {
         const fused_v4u v = dfa_load16(a.bases, roff + 4 * j, a.total_bytes);      /* (bounds-checked: the batch's last read ends the allocation) */
         uint32_t hm = 0;
         packed_word(v.x, st, hm); packed_word(v.y, st, hm); packed_word(v.z, st, hm); packed_word(v.w, st, hm);
         hm = __builtin_bitreverse32(hm);                  /* first pair of the 32 in bit 31 */
         const uint32_t lo = j * 8;                        /* pair index of bit 31 */
         const uint32_t keep = npairs <= lo ? 0u : (npairs - lo >= 32u ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> (npairs - lo)));
         hm &= keep;
         if (hm) {
            const uint32_t f = lo + (uint32_t)__builtin_clz(hm), l = lo + 31u - (uint32_t)__builtin_ctz(hm);
            first = first == 0xFFFFFFFFu ? f : first;
            last = l;
         }
      }
      if (live) {
         /* candidate columns: the second base of the flagged pairs (the read's last base when it has no second) */
         uint32_t fc = 2u * first + 1u, lc = 2u * last + 1u;
         fc = fc < a.read_len ? fc : a.read_len - 1u;
         lc = lc < a.read_len ? lc : a.read_len - 1u;
         a.cand[r] = first == 0xFFFFFFFFu ? 0u : ((fc << 16) | lc) + 1u;
      }
   }
}

/* The hit list of the candidate reads (one lane per read of the segment: a load, and for one read in twenty five stores). */
__global__ __launch_bounds__(256) void k_packed_list(PackedArgs a)
{
   const uint32_t stride = gridDim.x * 256;
   const uint32_t L = a.read_len;
   for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < a.nreads; r += stride) {
      const uint32_t cd = a.cand[r];
      if (!cd) continue;
      const uint32_t k = a.coff[r];
      if (k >= a.cap) continue;                            /* (the overflow is reported by k_packed_counts) */
      a.hit_start[k] = k * (L + 1);
      a.hit_line[k] = (uint32_t)(a.line_base + r + 1u);
      a.hit_col[k] = (cd - 1u) >> 16;
      a.hit_last[k] = (cd - 1u) & 0xFFFFu;
      a.nh[k] = 0u;
   }
}

/* Candidate reads -> ASCII lines of the staging text, N restored (one lane per CANDIDATE: run per read, the 151 byte stores of
   one candidate in a wave of 64 reads held the whole wave -- 5.8 ms per 100 M reads). */
__global__ __launch_bounds__(256) void k_packed_stage(PackedArgs a)
{
   const uint32_t n = a.cnt->seg_nhitlines;
   const uint32_t stride = gridDim.x * 256;
   const uint32_t L = a.read_len;
   for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < n; k += stride) {
      const uint64_t r = (uint64_t)a.hit_line[k] - 1u;     /* read index in the batch (line numbers are read indices + 1) */
      const uint8_t *p = a.bases + r * (uint64_t)a.stride;
      const uint8_t *nm = a.nmask ? a.nmask + r * (uint64_t)a.nstride : nullptr;
      uint8_t *out = a.stage + (uint64_t)k * (L + 1);
      for (uint32_t i = 0; i < L; i += 4) {
         const uint32_t b = p[i >> 2];
         const uint32_t nbits = nm ? (uint32_t)nm[i >> 3] >> (4u - (i & 4u)) : 0u;      /* the four N bits of these bases, first base in bit 3 */
         for (uint32_t q = 0; q < 4 && i + q < L; q++) {
            const uint32_t code = (b >> (6 - 2 * q)) & 3u;
            out[i + q] = ((nbits >> (3 - q)) & 1u) ? (uint8_t)'N' : (uint8_t)("ACTG"[code]);
         }
      }
      out[L] = '\n';
   }
}

/* After the scan of the candidate flags: the segment's line / candidate counts, the capacity check. */
__global__ void k_packed_counts(PackedArgs a)
{
   Counters *c = a.cnt;
   uint32_t n = c->seg_nhitlines;                          /* total of the flag scan */
   if (n > c->need_hitlines) c->need_hitlines = n;
   if (n > a.cap) { atomicOr(&c->overflow, 2u); n = 0; }
   if (c->overflow & 2u) n = 0;
   c->seg_nhitlines = n;
   c->seg_nrec = n;
   c->seg_nlines = a.nreads;
   c->seg_nheaders = 0;
   c->seg_novf = 0;
}

/* ASCII reads resident in HBM (one per line, each exactly read_len bases + newline) -> the packed layout: one lane per read.
   (Bench / test input; a byte that is not A C G T U N in either case counts as N when there is a mask, as A without.) */
__global__ __launch_bounds__(256) void k_pack_ascii(const uint8_t *text, uint64_t nreads, uint32_t L, uint8_t *bases, uint8_t *nmask, uint32_t stride, uint32_t nstride)
{
   const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
   if (r >= nreads) return;
   const uint8_t *t = text + r * (uint64_t)(L + 1);
   uint8_t *b = bases + r * (uint64_t)stride, *n = nmask ? nmask + r * (uint64_t)nstride : nullptr;
   for (uint32_t i = 0; i < L; i += 8) {
      uint32_t packed = 0, nbits = 0;
      for (uint32_t q = 0; q < 8 && i + q < L; q++) {
         const uint32_t ch = t[i + q], up = ch & 0xDFu;
         const bool base = up == 'A' || up == 'C' || up == 'G' || up == 'T' || up == 'U';
         packed |= (base ? (ch >> 1) & 3u : 0u) << (14 - 2 * q);
         nbits |= (base ? 0u : 1u) << (7 - q);
      }
      b[i >> 2] = (uint8_t)(packed >> 8);
      if (i + 4 < L) b[(i >> 2) + 1] = (uint8_t)packed;
      if (n) n[i >> 3] = (uint8_t)nbits;
   }
}

#undef PACKED_STEP

#endif
