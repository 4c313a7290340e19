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
 *                   The candidate reads of a block (5 % of a read set with planted hits) are unpacked on the spot -- sixteen
 *                   lanes per candidate, N restored -- into lines of a staging text (every wave fills its own share of it).
 *   k_packed_list   after a scan over the blocks' candidate counts: the hit-list arrays the exact pass reads, in read order,
 *                   pointing at those lines: from here on k_exact1 COUNT / EMIT run as behind k_pair -- windows, every
 *                   match option, bit-exact.
 *
 *   k_packed_walk<true>   (round 4) the same walk over the QUAD table (seeq_dfa.h section 3b) where the pattern has one: a packed
 *                   BYTE -- four bases -- per gather, half the gathers; per byte one SDWA shift (byte -> index << 1), one v_bfi
 *                   (index under the state's row offset), the gather, one v_alignbit by 4 (the entry's four position flags into a
 *                   per-base mask).  The small automaton behind it is a partition filter: more false candidates (one read in 57
 *                   instead of one in 700 for the headline pattern), verified like the others.
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
   uint32_t      *cand;         /* [nreads], written for candidate reads only: {first, last candidate column} */
   uint32_t      *cslot;        /* [nreads], candidate reads only: the line of the staging text the walk kernel unpacked the read into */
   uint32_t       wave_cap;     /* lines of the staging text every wave of the walk grid owns */
   uint32_t       pitch;        /* bytes per line of the staging text: read_len + 1 rounded up to 16 */
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

/* the quad table: four gathers per word; entry = row offset (bits 9-15) | the four bases' accept flags (bits 0-3) */
#define PACKED_STEP4(K) \
   asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #K : "=v"(of) : "v"(one), "v"(w)); \
   ad = (of & 0x1FFu) | (st & ~0x1FFu); \
   st = *(stream_lds_cu16 *)(uintptr_t)ad; \
   hm = __builtin_amdgcn_alignbit(st, hm, 4);

__device__ __forceinline__ void packed_word4(uint32_t w, uint32_t one, uint32_t &st, uint32_t &hm)
{
   uint32_t of, ad;
   PACKED_STEP4(0) PACKED_STEP4(1) PACKED_STEP4(2) PACKED_STEP4(3)
}

typedef uint32_t packed_u32_unaligned __attribute__((aligned(1)));

/* Read `r` of the batch -> line `line` of the staging text, N restored: lane i of sixteen handles word i of the read (16
   bases).  In two halves so that the walk can put a block of text between them: packed_stage_load requests the lane's
   packed word and N bits, packed_stage_store -- a block later, when they have long arrived -- does the four v_perm look-ups
   and the four word stores. */
struct packed_pend_t { uint32_t w, nb, line, live; };

__device__ __forceinline__ packed_pend_t packed_stage_load(const PackedArgs &a, uint64_t r, uint32_t line, uint32_t i)
{
   packed_pend_t p = {0u, 0u, line, 0u};
   if (16u * i > a.read_len) return p;                     /* (lane read_len / 16 writes the newline, with or without bases in front of it) */
   p.live = 1u;
   if (16u * i == a.read_len) return p;
   const uint8_t *pb = a.bases + r * (uint64_t)a.stride + 4u * i;
   if (4u * i + 4u <= a.stride) p.w = *reinterpret_cast<const packed_u32_unaligned *>(pb);
   else for (uint32_t b = 4u * i; b < a.stride; b++) p.w |= (uint32_t)pb[b - 4u * i] << (8u * (b - 4u * i));
   if (a.nmask) {
      const uint8_t *pn = a.nmask + r * (uint64_t)a.nstride;
      p.nb = (uint32_t)pn[2u * i] << 8;
      if (2u * i + 1u < a.nstride) p.nb |= pn[2u * i + 1u];
   }
   return p;
}

/* The staging text has a pitch of a multiple of 16 bytes per line (read_len + 1 rounded up): every lane stores ONE aligned
   16-byte word -- the line's last word carries the newline and zeros behind it (never read as text: the line ends at the
   newline).  (Lines at a pitch of read_len + 1: four unaligned 4-byte stores per lane and a byte loop for the last word --
   the staging cost 0.11 ms per 16 Mi reads inside the walk kernel, as much as in a kernel of its own.) */
__device__ __forceinline__ void packed_stage_store(const PackedArgs &a, const packed_pend_t &p, uint32_t i)
{
   if (!p.live) return;
   const uint32_t L = a.read_len;
   uint8_t *out = a.stage + (uint64_t)p.line * a.pitch;
   uint32_t word[4];
#pragma unroll
   for (int j = 0; j < 4; j++) {
      const uint32_t b = (p.w >> (8 * j)) & 0xFFu;
      const uint32_t sel = (b >> 6) | ((b & 0x30u) << 4) | ((b & 0x0Cu) << 14) | ((b & 3u) << 24);      /* the four codes, first base in byte 0 */
      const uint32_t n4 = (p.nb >> (12 - 4 * j)) & 0xFu;                                               /* their N bits, first base in bit 3 */
      const uint32_t nm = (((n4 >> 3) & 1u) | ((n4 & 4u) << 6) | ((n4 & 2u) << 15) | ((n4 & 1u) << 24)) * 0xFFu;
      word[j] = (__builtin_amdgcn_perm(0u, 0x47544341u, sel) & ~nm) | (0x4E4E4E4Eu & nm);              /* "ACTG"[code] or 'N' */
   }
   if (16u * i + 16u > L) {                                /* the line's last word: bases, the newline, zeros */
      const uint32_t keep = L - 16u * i;                   /* 0 .. 15 bases */
#pragma unroll
      for (int j = 0; j < 4; j++) {
         const uint32_t lo = 4u * (uint32_t)j;
         const uint32_t nb_ = keep <= lo ? 0u : (keep - lo >= 4u ? 4u : keep - lo);      /* bases of this word that belong to the read */
         const uint32_t km = nb_ >= 4u ? 0xFFFFFFFFu : (1u << (8u * nb_)) - 1u;
         word[j] = (word[j] & km) | ((keep >= lo && keep < lo + 4u) ? 0x0Au << (8u * (keep - lo)) : 0u);
      }
   }
   *reinterpret_cast<fused_v4u *>(out + 16u * i) = fused_v4u{word[0], word[1], word[2], word[3]};
   if (L == 256u && i == 15u) out[256] = '\n';            /* (the one length whose newline has no lane of its own) */
}

template <bool QUAD>
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
   uint32_t wslots = 0;                                    /* wave-uniform: lines of the staging text this wave has filled */
   bool wave_over = false;
   packed_pend_t pend = {0u, 0u, 0u, 0u};                  /* a candidate's words on their way: unpacked and stored a block later */
   for (uint32_t blk = gwave; blk < nblocks; blk += nwaves) {
      const uint32_t r = blk * 64 + (uint32_t)lane;
      const bool live = r < a.nreads;
      const uint64_t roff = (a.first + (live ? r : 0)) * (uint64_t)a.stride;
      uint32_t st = 0, first = 0xFFFFFFFFu, last = 0;
      if (QUAD) {
         /* 16 bytes at a time: four words, sixteen table steps, two masks of 32 bases (first base of a mask in bit 0 until reversed) */
         uint32_t one = 1u;
         asm volatile("" : "+v"(one));                     /* (SDWA takes the shift from a register) */
#pragma unroll 1
         for (uint32_t j = 0; j < nwords; j += 4) {
            const fused_v4u v = dfa_load16(a.bases, roff + 4 * j, a.total_bytes);
            uint32_t hm0 = 0, hm1 = 0;
            packed_word4(v.x, one, st, hm0); packed_word4(v.y, one, st, hm0);
            if (j + 2 < nwords) { packed_word4(v.z, one, st, hm1); packed_word4(v.w, one, st, hm1); }
#pragma unroll
            for (int h = 0; h < 2; h++) {
               uint32_t hm = __builtin_bitreverse32(h ? hm1 : hm0);          /* first base of the 32 in bit 31 */
               const uint32_t lo = j * 16u + 32u * (uint32_t)h;               /* base index of bit 31 */
               hm &= a.read_len <= lo ? 0u : (a.read_len - lo >= 32u ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> (a.read_len - lo)));
               if (hm) {
                  const uint32_t f = lo + (uint32_t)__builtin_clz(hm), l = lo + 31u - (uint32_t)__builtin_ctz(hm);
                  first = first == 0xFFFFFFFFu ? f : first;
                  last = l;
               }
            }
         }
      } else {
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
      }
      const bool is_cand = live && first != 0xFFFFFFFFu;
      const uint64_t mask = __ballot(is_cand);
      if (lane == 0) { a.bmask[blk] = mask; a.boff[blk] = (uint32_t)__popcll(mask); }
      if (is_cand) {
         /* candidate columns: the second base of the flagged pairs (the read's last base when it has no second); the quad table
            names the base itself */
         uint32_t fc = QUAD ? first : 2u * first + 1u, lc = QUAD ? last : 2u * last + 1u;
         fc = fc < a.read_len ? fc : a.read_len - 1u;
         lc = lc < a.read_len ? lc : a.read_len - 1u;
         a.cand[r] = (fc << 16) | lc;
      }
      /* The candidates of this block are unpacked right here -- their bytes are in L1, a later kernel would gather 57-byte
         objects scattered over the batch (0.145 ms per 16 Mi reads) -- into lines of the staging text this WAVE owns (no
         atomics: a counter per wave, as the hit slices of k_pair); k_packed_list points the hit list at them. */
      const uint32_t nc = (uint32_t)__popcll(mask);
      if (nc && a.stage) {                                /* (no staging text: k_verify_packed reads the candidates' windows from the batch itself) */
         const uint32_t s0 = wslots;
         wslots += nc;
         if (wslots <= a.wave_cap) {
            const uint32_t line0 = gwave * a.wave_cap + s0;
            if (is_cand) a.cslot[r] = line0 + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            const uint32_t g = (uint32_t)lane >> 4, i = (uint32_t)lane & 15u;
            for (uint32_t b0 = 0; b0 < nc; b0 += 4) {       /* sixteen lanes per candidate, four candidates per round */
               packed_stage_store(a, pend, i);               /* the round before this one (the last round of a block: finished behind the next block's walk) */
               pend.live = 0u;
               const uint32_t idx = b0 + g;
               if (idx >= nc) continue;
               uint64_t mm = mask;
               for (uint32_t t = 0; t < idx; t++) mm &= mm - 1ull;
               const uint32_t src = (uint32_t)__builtin_ctzll(mm);
               pend = packed_stage_load(a, a.first + (uint64_t)blk * 64u + src, line0 + idx, i);
            }
         } else {
            wave_over = true;
         }
      }
   }
   packed_stage_store(a, pend, (uint32_t)lane & 15u);
   if (wave_over && lane == 0) {                           /* the wave's share of the staging text is too small: the scan is run again with more */
      atomicOr(&a.cnt->overflow, 2u);
      atomicMax(&a.cnt->need_hitlines, wslots * nwaves);
   }
}

/* The hit list of the candidate reads.  A wave takes sixteen blocks of 64 reads at a time: sixteen lanes spread their block's mask
   into read numbers in LDS (three bits each on a read set with planted hits), then all 64 lanes fetch {candidate columns, staging
   line} of one candidate each and store the entries side by side.  (One lane per block walking its own bits -- dependent loads,
   scattered stores -- took 60 us per 16 Mi reads; one lane per read 55 us, 95 % of them looking at nothing.) */
__global__ __launch_bounds__(256) void k_packed_list(PackedArgs a)
{
   __shared__ uint32_t s_r[4][1024];
   const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
   const uint32_t nblocks = (a.nreads + 63u) >> 6;
   const uint32_t ngroups = (nblocks + 15u) >> 4;
   for (uint32_t g = blockIdx.x * 4u + wave; g < ngroups; g += gridDim.x * 4u) {
      const uint32_t blk = g * 16u + lane;
      const bool own = lane < 16u && blk < nblocks;
      uint64_t mask = own ? a.bmask[blk] : 0ull;
      const uint32_t off = own ? a.boff[blk] : 0u;
      const uint32_t cnt = (uint32_t)__popcll(mask);
      const uint32_t incl = wave_incl_scan_u32(cnt);
      const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      const uint32_t k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);      /* (boff is the exclusive scan of the blocks' counts: lane i's first entry is k0 + the counts before it) */
      uint32_t e = incl - cnt;
      while (mask) {
         s_r[wave][e++] = blk * 64u + (uint32_t)__builtin_ctzll(mask);
         mask &= mask - 1ull;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (uint32_t e0 = 0; e0 < total; e0 += 64u) {
         const uint32_t ei = e0 + lane, k = k0 + ei;
         if (ei < total && k < a.cap) {                     /* (k >= cap: the overflow is reported by k_packed_counts) */
            const uint32_t r = s_r[wave][ei];
            const uint32_t cd = a.cand[r], sl = a.stage ? a.cslot[r] : 0u;      /* (no staging text: the exact pass finds the read through its line number) */
            a.hit_start[k] = sl * a.pitch;
            a.hit_line[k] = (uint32_t)(a.line_base + r + 1u);
            a.hit_col[k] = cd >> 16;
            a.hit_last[k] = cd & 0xFFFFu;
            a.nh[k] = 0u;
         }
      }
      __builtin_amdgcn_wave_barrier();                     /* (the next group's writes behind this group's reads) */
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

/* The whole batch as ASCII text (one read per line, read_len bases + newline, N restored): the fall-back of seeqdevScanPacked for
   patterns the packed walk does not serve (more than 62 positions, no pair automaton) -- the ASCII scan then runs over it.  One
   thread per 16 output characters. */
__global__ __launch_bounds__(256) void k_unpack_ascii(const uint8_t *bases, const uint8_t *nmask, uint64_t nreads, uint32_t L, uint32_t stride, uint32_t nstride,
                                                      uint8_t *out)
{
   const uint32_t pieces = L / 16u + 1u;                   /* positions 0 .. L (the newline) */
   const uint64_t gid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
   const uint64_t r = gid / pieces;
   const uint32_t i = (uint32_t)(gid % pieces);
   if (r >= nreads) return;
   const uint8_t *pb = bases + r * (uint64_t)stride, *pn = nmask ? nmask + r * (uint64_t)nstride : nullptr;
   uint8_t *o = out + r * (uint64_t)(L + 1u);
   for (uint32_t q = 0; q < 4u; q++) {
      uint32_t word = 0, nbytes = 0;
      for (uint32_t j = 0; j < 4u; j++) {
         const uint32_t p_ = 16u * i + 4u * q + j;
         if (p_ > L) break;
         uint32_t ch = '\n';
         if (p_ < L) {
            const uint32_t code = (pb[p_ >> 2] >> (6u - 2u * (p_ & 3u))) & 3u;
            ch = (0x47544341u >> (8u * code)) & 0xFFu;      /* "ACTG"[code]: the code is bits 1-2 of the letter */
            if (pn && ((pn[p_ >> 3] >> (7u - (p_ & 7u))) & 1u)) ch = 'N';
         }
         word |= ch << (8u * j);
         nbytes++;
      }
      uint8_t *dst = o + 16u * i + 4u * q;
      if (nbytes == 4u) *reinterpret_cast<packed_u32_unaligned *>(dst) = word;
      else for (uint32_t j = 0; j < nbytes; j++) dst[j] = (uint8_t)(word >> (8u * j));
   }
}

#undef PACKED_STEP
#undef PACKED_STEP4

#endif
