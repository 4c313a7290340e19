/*
 * seeq_packed.h -- the scan of PACKED read batches (include/seeq_amd.h: seeqdev_packed_t): 2 bits per base, four bases per
 * byte, every read at a fixed stride -- the north star's "coalesced HBM loads of packed 2-bit read batches", 38 bytes per
 * 150 bp read instead of 151.
 *
 *   k_packed_walk   one READ per lane, 64 consecutive reads per wave: the lane loads its read (<= 64 bytes, 16-byte
 *                   loads) and walks k_pair's pair automaton over it straight from the root -- a read begins where its
 *                   line begins, so there is no warm-up, no newline, no alphabet check, and a nibble of the packed byte IS
 *                   the pair index: per text word two VALU instructions prepare eight steps (k_pair: two).  Per pair: the
 *                   SDWA v_xor, the gather, one v_alignbit.  Output: per block of 64 reads the mask of candidate reads and
 *                   their number, per candidate read cand[r] = {first, last candidate column}.
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
   uint32_t      *cand;         /* [nreads], written for candidate reads only */
   uint64_t      *bmask;        /* [ceil(nreads / 64)] candidate reads of every block of 64 reads, read r = bit r & 63 */
   uint32_t      *boff;         /* [ceil(nreads / 64)] their number; after the scan: candidates before the block */
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
         packed_word(v.x, st, hm); packed_word(v.y, st, hm);
         if (j + 2 < nwords) { packed_word(v.z, st, hm); packed_word(v.w, st, hm); }      /* (150 bp: ten words, the third load's last two are the next read) */
         else hm >>= 16;                                   /* (the first pair belongs in bit 0 before the reversal) */
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
      const bool is_cand = live && first != 0xFFFFFFFFu;
      const uint64_t mask = __ballot(is_cand);
      if (lane == 0) { a.bmask[blk] = mask; a.boff[blk] = (uint32_t)__popcll(mask); }
      if (is_cand) {
         /* candidate columns: the second base of the flagged pairs (the read's last base when it has no second) */
         uint32_t fc = 2u * first + 1u, lc = 2u * last + 1u;
         fc = fc < a.read_len ? fc : a.read_len - 1u;
         lc = lc < a.read_len ? lc : a.read_len - 1u;
         a.cand[r] = (fc << 16) | lc;
      }
   }
}

/* The hit list of the candidate reads: one lane per read, a wave per block of 64 (two loads per wave; for one read in twenty a
   load and five stores). */
__global__ __launch_bounds__(256) void k_packed_list(PackedArgs a)
{
   const uint32_t stride = gridDim.x * 256;
   const uint32_t L = a.read_len;
   const uint32_t nr = (a.nreads + 63u) & ~63u;
   for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < nr; r += stride) {
      const uint64_t mask = a.bmask[r >> 6];
      const uint32_t lane = r & 63u;
      if (!((mask >> lane) & 1u)) continue;
      const uint32_t k = a.boff[r >> 6] + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
      if (k >= a.cap) continue;                            /* (the overflow is reported by k_packed_counts) */
      const uint32_t cd = a.cand[r];
      a.hit_start[k] = k * (L + 1);
      a.hit_line[k] = (uint32_t)(a.line_base + r + 1u);
      a.hit_col[k] = cd >> 16;
      a.hit_last[k] = cd & 0xFFFFu;
      a.nh[k] = 0u;
   }
}

/* Candidate reads -> ASCII lines of the staging text, N restored: SIXTEEN lanes per candidate, a lane per word of the read
   (16 bases): one load of the packed bases and one of the N bits per lane, four v_perm look-ups, four word stores.
   (A lane per read held its wave for the 151 byte stores of one candidate in twenty: 5.8 ms per 100 M reads; a lane per
   candidate stored to 64 cache lines per instruction: 2.5 ms per 16 M reads; a lane per 16 bytes of the staging text
   loaded every byte of the read by itself: 0.33 ms per 16 M reads, 0.24 of them the byte loads.) */
typedef uint32_t packed_u32_unaligned __attribute__((aligned(1)));

__global__ __launch_bounds__(256) void k_packed_stage(PackedArgs a)
{
   const uint32_t n = a.cnt->seg_nhitlines;
   const uint32_t L = a.read_len, L1 = L + 1u;
   const uint32_t i = threadIdx.x & 15u;                   /* word of the read */
   const uint32_t groups = gridDim.x * 16u;
   for (uint32_t k = blockIdx.x * 16u + (threadIdx.x >> 4); k < n; k += groups) {
      const uint64_t r = (uint64_t)a.hit_line[k] - 1u;     /* read index in the batch (line numbers are read indices + 1) */
      uint8_t *out = a.stage + (uint64_t)k * L1;
      if (i == 0) out[L] = '\n';
      if (16u * i >= L) continue;
      const uint8_t *pb = a.bases + r * (uint64_t)a.stride + 4u * i;
      uint32_t w = 0, nb = 0;
      if (4u * i + 4u <= a.stride) w = *reinterpret_cast<const packed_u32_unaligned *>(pb);
      else for (uint32_t b = 4u * i; b < a.stride; b++) w |= (uint32_t)pb[b - 4u * i] << (8u * (b - 4u * i));
      if (a.nmask) {
         const uint8_t *pn = a.nmask + r * (uint64_t)a.nstride;
         nb = (uint32_t)pn[2u * i] << 8;
         if (2u * i + 1u < a.nstride) nb |= pn[2u * i + 1u];
      }
      uint32_t word[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
         const uint32_t b = (w >> (8 * j)) & 0xFFu;
         const uint32_t sel = (b >> 6) | ((b & 0x30u) << 4) | ((b & 0x0Cu) << 14) | ((b & 3u) << 24);      /* the four codes, first base in byte 0 */
         const uint32_t n4 = (nb >> (12 - 4 * j)) & 0xFu;                                                 /* their N bits, first base in bit 3 */
         const uint32_t nm = (((n4 >> 3) & 1u) | ((n4 & 4u) << 6) | ((n4 & 2u) << 15) | ((n4 & 1u) << 24)) * 0xFFu;
         word[j] = (__builtin_amdgcn_perm(0u, 0x47544341u, sel) & ~nm) | (0x4E4E4E4Eu & nm);              /* "ACTG"[code] or 'N' */
      }
      if (16u * i + 16u <= L) {
#pragma unroll
         for (int j = 0; j < 4; j++) *reinterpret_cast<packed_u32_unaligned *>(out + 16u * i + 4u * j) = word[j];
      } else {
         for (uint32_t c = 16u * i; c < L; c++) out[c] = (uint8_t)(word[(c >> 2) & 3u] >> (8u * (c & 3u)));
      }
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
