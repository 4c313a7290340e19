/*
 * seeq_fused.h -- the hot kernel of seeq-mi355x: k_fused<NW>.
 *
 * One pass over the text in HBM does what the reference does per line in
 * seeq.c:361-380 + libseeq.c:250-275 (getline, newline strip, FASTA header
 * skip, byte classification, per-character distance) for patterns of up to
 * 30 positions (one 32-bit Myers word with two spare flag bits):
 *
 *   1. a workgroup stages one TILE of text (~64*NW lines) from HBM into LDS
 *      with coalesced 16-byte loads -- every input byte is fetched from HBM
 *      exactly once (plus a 1 KiB halo per tile);
 *   2. it finds the newlines of the tile in LDS (SWAR compare, wave prefix
 *      sums) -> line starts, in order, in LDS;
 *   3. one line per lane: the lane walks its line through LDS 16 characters
 *      at a time.  Each character costs one LDS lookup EQ[byte] (a 256-entry
 *      table: top-aligned Peq word of the byte's class, or a flag for bytes
 *      that end the line / are skipped under the non-DNA option) and ~13
 *      integer VALU ops of the Myers column update; the running minimum of
 *      D[m][j] tells whether the line has a hit (SQ_COUNTLINES needs nothing
 *      else; for records the exact pass k_exact re-reads only the hit lines);
 *   4. hit lines are compacted in line order per tile (wave ballots + popc)
 *      and appended to a global list with one atomic per tile.
 *
 * No MFMA: this is bitwise integer work.  LDS holds the text window, the EQ
 * table and the line starts; the only HBM traffic is the text itself.
 */
#ifndef SEEQ_FUSED_H_
#define SEEQ_FUSED_H_

#define FUSED_FLAG_TERM 1u      /* byte ends the line                          */
#define FUSED_FLAG_SKIP 2u      /* byte is skipped but counted in coordinates  */
#define FUSED_FLAGS     3u
#define FUSED_MAX_WLEN  30      /* 32-bit word minus the two flag bits         */
#define FUSED_MAX_WLEN2 62      /* two words minus the two flag bits (k_direct<.,2>, k_exact1<.,2>) */
#define FUSED_HALO_MAX  1024    /* bytes staged beyond the tile (runtime, <= this): longest line handled from LDS */
#define FUSED_CAPL_PER_THREAD 2 /* line starts kept in LDS per pass = this * threads */
#define FUSED_MAXR      16      /* newline-detection rounds: tile <= MAXR * threads * 16 bytes */
#define FUSED_MAXS      10      /* staging rounds: tile + halo <= MAXS * threads * 16 bytes        */

typedef unsigned int fused_v4u __attribute__((ext_vector_type(4)));
typedef fused_v4u fused_v4u_unaligned __attribute__((aligned(1)));   /* the text pointer may have any alignment */

struct FusedArgs {
   const uint8_t *text;        /* whole buffer                                 */
   uint64_t       nbytes;
   uint64_t       seg_base;    /* first byte of the segment                    */
   uint32_t       seg_len;
   uint32_t       first_seg;
   uint32_t       tile_bytes;  /* multiple of 16                               */
   uint32_t       halo;        /* k_fused: multiple of 16, <= FUSED_HALO_MAX   */
   uint32_t       pos_bias;    /* k_stream: hit offsets are relative to seg_base - pos_bias (a line can start before the segment) */
   uint32_t       ntiles;
   const uint32_t *eqtab;      /* [256] top-aligned Peq word or flag, per byte */
   const uint32_t *peq;        /* [2][5][1] bottom-aligned (long-line fallback)*/
   int            m, tau, options, want;
   uint32_t      *tile_cl;     /* per tile: counted lines (headers excluded)   */
   uint32_t      *tile_hits;   /* per tile: hit lines                          */
   uint4         *tmp;         /* hit entries {tile, seq, start, counted rank}: one slice per workgroup */
   uint32_t       cap_tmp;     /* total entries                                */
   uint32_t       slice_cap;   /* entries per workgroup slice = cap_tmp / grid  */
   uint32_t      *wg_hits;     /* per slice (workgroup of k_fused / wave of k_direct): entries stored */
   uint32_t      *wg_part;     /* per slice: {lines, headers, hit lines | overflow<<31}               */
   uint32_t      *tile_dirty;  /* k_stream, long-line mode: per tile, 1 when it holds a byte outside the alphabet (then its exclusive prefix); else NULL */
   uint64_t      *tile_dmask;  /* k_stream, long-line mode: per tile, one bit per 128-byte chunk (lane) that holds a non-alphabet byte */
   uint32_t      *wg_lastnl;   /* k_stream, per wave: segment-relative offset + 1 of the last newline it saw (0: none); else NULL */
   uint32_t       debug;       /* profiling experiments only (SEEQ_FUSED_DEBUG): 1 = skip the per-line scan */
   const uint16_t *dfa;        /* k_dfa: transition table, dfa_rows x 8 u16 (seeq_dfa.h) */
   uint32_t       dfa_rows;
   uint32_t       dfa_final_base;   /* row offset of ACC_FINAL; DEAD_FINAL = +16 */
   Counters      *cnt;
};

/* One Myers column step on a TOP-aligned pattern (row m = bit 31).  The two
 * left shifts double as the extraction of the horizontal delta of row m: the
 * carry out of ph+ph / mh+mh is +1 / -1 on D[m][j]. */
__device__ __forceinline__ void fused_step(uint32_t eq, uint32_t &pv, uint32_t &mv, uint32_t &score)
{
   /* Hyyro's form of the Myers recurrence: D0 = zero-diagonal vector (12 VALU ops with 3-input bitops) */
   const uint32_t s = (eq & pv) + pv;
   const uint32_t d0 = ((s ^ pv) | eq) | mv;
   const uint32_t ph = mv | ~(d0 | pv);
   const uint32_t mh = pv & d0;
   uint32_t ph2, mh2;
   asm("v_add_co_u32 %0, vcc, %2, %2\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
       : "=v"(ph2), "+v"(score) : "v"(ph) : "vcc");
   asm("v_add_co_u32 %0, vcc, %2, %2\n\tv_subbrev_co_u32 %1, vcc, 0, %1, vcc"
       : "=v"(mh2), "+v"(score) : "v"(mh) : "vcc");
   pv = mh2 | ~(d0 | ph2);
   mv = ph2 & d0;
}

/* alphabet check of four characters: nonzero when a byte is outside {ACGTN, acgtn, '\n'} (k_stream's table columns
 * are exact for those; any other byte aliases onto one of them) */
__device__ __forceinline__ uint32_t fused_bad4(uint32_t w)
{
   /* canonical byte of each table column (A C T G . \n . N); the text must equal it -- letters in either case,
      the newline exactly ('*' = 0x2A is '\n' with the case bit set: it must NOT pass) */
   const uint32_t idx = (w & 0x0E0E0E0Eu) >> 1;
   const uint32_t canon = __builtin_amdgcn_perm(0x4EFF0AFFu, 0x47544341u, idx);
   const uint32_t fold = __builtin_amdgcn_perm(0xDFFFFFFFu, 0xDFDFDFDFu, idx);      /* per column: case-fold mask */
   return (w & fold) ^ canon;
}

/* Inclusive prefix sum over the 64 lanes of a wave with DPP row shifts / broadcasts
 * (VALU-speed, no LDS crossbar round trips). */
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x)
{
   x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);   /* row_shr:1 */
   x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);   /* row_shr:2 */
   x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);   /* row_shr:4 */
   x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);   /* row_shr:8 */
   x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, true);   /* row_bcast:15 -> rows 1,3 */
   x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, true);   /* row_bcast:31 -> rows 2,3 */
   return x;
}

/* LDS layout of k_fused: fixed-size tables first (compile-time offsets), text window last. */
/* ---- one- and two-word variants behind one interface (k_direct, k_exact1 are templated on W) ---- */
/* W = 2 serves patterns of 31..62 positions: the pattern sits in the top m bits of a 64-bit column
   (lo word = low rows), the two flag bits in bits 0-1 of the low word. */
template <int W> struct fused_eq_t;
template <> struct fused_eq_t<1> { uint32_t w0; };
template <> struct fused_eq_t<2> { uint32_t w0, w1; };

typedef __attribute__((address_space(3))) const uint32_t fused_lds_cu32;
typedef __attribute__((address_space(3))) const uint64_t fused_lds_cu64;

template <int W>
__device__ __forceinline__ fused_eq_t<W> fused_eq_load(uint32_t lds_byte_addr);
template <>
__device__ __forceinline__ fused_eq_t<1> fused_eq_load<1>(uint32_t addr)
{
   fused_eq_t<1> e;
   e.w0 = *(fused_lds_cu32 *)(uintptr_t)addr;
   return e;
}
template <>
__device__ __forceinline__ fused_eq_t<2> fused_eq_load<2>(uint32_t addr)
{
   const uint64_t v = *(fused_lds_cu64 *)(uintptr_t)addr;          /* ds_read_b64 */
   fused_eq_t<2> e;
   e.w0 = (uint32_t)v;
   e.w1 = (uint32_t)(v >> 32);
   return e;
}

template <int W> struct fused_state_t;
template <> struct fused_state_t<1> {
   uint32_t pv, mv, score;
   __device__ __forceinline__ void init(uint32_t m) { pv = 0xFFFFFFFFu; mv = 0u; score = m; }
   __device__ __forceinline__ void step(const fused_eq_t<1> &e) { fused_step(e.w0, pv, mv, score); }
};
template <> struct fused_state_t<2> {
   uint32_t pv0, pv1, mv0, mv1, score;
   __device__ __forceinline__ void init(uint32_t m) { pv0 = pv1 = 0xFFFFFFFFu; mv0 = mv1 = 0u; score = m; }
   /* the 64-bit version of fused_step: one carry chain through both words; the carry out of the
      high word of ph+ph / mh+mh is the +1 / -1 on D[m][j] */
   __device__ __forceinline__ void step(const fused_eq_t<2> &e)
   {
      const uint64_t pv = ((uint64_t)pv1 << 32) | pv0, eq = ((uint64_t)e.w1 << 32) | e.w0;
      const uint64_t s = (eq & pv) + pv;
      const uint32_t s0 = (uint32_t)s, s1 = (uint32_t)(s >> 32);
      const uint32_t d00 = ((s0 ^ pv0) | e.w0) | mv0, d01 = ((s1 ^ pv1) | e.w1) | mv1;
      const uint32_t ph0 = mv0 | ~(d00 | pv0), ph1 = mv1 | ~(d01 | pv1);
      const uint32_t mh0 = pv0 & d00, mh1 = pv1 & d01;
      uint32_t p0, p1, m0, m1;
      asm("v_add_co_u32 %0, vcc, %3, %3\n\tv_addc_co_u32 %1, vcc, %4, %4, vcc\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc"
          : "=&v"(p0), "=&v"(p1), "+v"(score) : "v"(ph0), "v"(ph1) : "vcc");
      asm("v_add_co_u32 %0, vcc, %3, %3\n\tv_addc_co_u32 %1, vcc, %4, %4, vcc\n\tv_subbrev_co_u32 %2, vcc, 0, %2, vcc"
          : "=&v"(m0), "=&v"(m1), "+v"(score) : "v"(mh0), "v"(mh1) : "vcc");
      pv0 = m0 | ~(d00 | p0); pv1 = m1 | ~(d01 | p1);
      mv0 = p0 & d00;         mv1 = p1 & d01;
   }
};

template <int NW>
struct FusedLds {
   static constexpr int NT = 64 * NW;
   static constexpr int CAPL = FUSED_CAPL_PER_THREAD * NT;
   static constexpr int ITERS = FUSED_CAPL_PER_THREAD;
   static constexpr uint32_t EQ = 0;                              /* u32[256]              */
   static constexpr uint32_t STARTS = EQ + 256 * 4;               /* u32[CAPL]             */
   static constexpr uint32_t WTOT = STARTS + CAPL * 4;      /* u32[NW] (+pad to 16)  */
   static constexpr uint32_t HIT = WTOT + 16 * ((NW * 4 + 15) / 16);   /* u64[ITERS*NW]    */
   static constexpr uint32_t HDR = HIT + ITERS * NW * 8;          /* u64[ITERS*NW]         */
   static constexpr uint32_t MISC = HDR + ITERS * NW * 8;         /* u32[8]                */
   static constexpr uint32_t PEQ = MISC + 32;                     /* u32[12]               */
   static constexpr uint32_t LUT = PEQ + 48;                      /* u8[256]               */
   static constexpr uint32_t TEXT = (LUT + 256 + 15) & ~15u;      /* u8[WIN + 32]          */
};

template <int NW>
static size_t fused_lds_bytes(uint32_t tile_bytes, uint32_t halo)
{
   return (size_t)FusedLds<NW>::TEXT + tile_bytes + halo + 32;
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_fused(FusedArgs a)
{
   typedef FusedLds<NW> L;
   constexpr int NT = 64 * NW;
   extern __shared__ __align__(16) uint8_t smem[];
   const uint32_t TB = a.tile_bytes;
   const uint32_t WIN = TB + a.halo;                     /* staged bytes */
   constexpr uint32_t CAPL = L::CAPL;
   uint32_t *s_eq = reinterpret_cast<uint32_t *>(smem + L::EQ);
   uint32_t *s_starts = reinterpret_cast<uint32_t *>(smem + L::STARTS);
   uint32_t *s_wtot = reinterpret_cast<uint32_t *>(smem + L::WTOT);
   uint64_t *s_hit = reinterpret_cast<uint64_t *>(smem + L::HIT);
   uint64_t *s_hdr = reinterpret_cast<uint64_t *>(smem + L::HDR);
   uint32_t *s_misc = reinterpret_cast<uint32_t *>(smem + L::MISC);
   uint32_t *s_peq = reinterpret_cast<uint32_t *>(smem + L::PEQ);
   uint8_t  *s_lut = smem + L::LUT;
   uint8_t  *s_text = smem + L::TEXT;

   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   const int uwave = __builtin_amdgcn_readfirstlane(wave);             /* same value, provably wave-uniform */
   const bool fasta = (a.options & SEEQDEV_FASTA) != 0;
   const uint32_t tau = (uint32_t)a.tau;

   for (int i = tid; i < 256; i += NT) { s_eq[i] = a.eqtab[i]; s_lut[i] = sq_class_of((uint32_t)i, a.options); }
   if (tid < 10) s_peq[tid] = a.peq[tid];

   /* Per-workgroup accumulators: no global atomics inside the tile loop (same-address atomics serialise
      at ~100/us chip-wide, which would cap the whole kernel at a few hundred thousand tiles per ms). */
   uint32_t wg_lines = 0, wg_hdrs = 0, wg_hitlines = 0, slice_pos = 0;
   bool wg_overflow = false;

   /* Software pipeline over tiles: the global loads of tile t+1 are issued (into registers) before the
      per-line scan of tile t and only written to LDS when tile t is done, so the HBM round trip hides
      behind ~10 us of VALU work instead of stalling all waves of the workgroup. */
   fused_v4u pre[FUSED_MAXS];
   bool pre_valid = false;                                /* pre[] holds the window of the tile about to start */
   auto prefetch = [&](uint32_t tl) {
      const uint64_t p0 = a.seg_base + (uint64_t)tl * TB;
      pre_valid = tl < a.ntiles && p0 + WIN + 32 <= a.nbytes;          /* interior tiles only */
      if (pre_valid) {
         const uint8_t *src = a.text + p0;
#pragma unroll
         for (int r = 0; r < FUSED_MAXS; r++) {
            const uint32_t off = ((uint32_t)r * NT + tid) * 16;
            if (off < WIN) pre[r] = *reinterpret_cast<const fused_v4u_unaligned *>(src + off);
         }
      }
   };
   prefetch(blockIdx.x);

   for (uint32_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
      const uint64_t t0 = a.seg_base + (uint64_t)tile * TB;                 /* absolute offset of the tile */
      const uint32_t tb = (uint32_t)(((uint64_t)a.seg_len - (uint64_t)tile * TB) < TB
                                     ? ((uint64_t)a.seg_len - (uint64_t)tile * TB) : TB);   /* owned bytes */
      __syncthreads();                                   /* previous tile fully consumed */
      /* ---- 1. stage [t0, t0 + WIN) into LDS, zero beyond the buffer ---- */
      if (pre_valid) {
#pragma unroll
         for (int r = 0; r < FUSED_MAXS; r++) {
            const uint32_t off = ((uint32_t)r * NT + tid) * 16;
            if (off < WIN) *reinterpret_cast<fused_v4u *>(s_text + off) = pre[r];
         }
         if (tid < 2) *reinterpret_cast<fused_v4u *>(s_text + WIN + tid * 16) = fused_v4u{0, 0, 0, 0};
      } else {
         /* last tile(s) of the buffer: bounds-checked byte loads */
         for (uint32_t off = (uint32_t)tid * 16; off < WIN + 32; off += NT * 16) {
            const uint64_t g = t0 + off;
            uint32_t w[4] = {0, 0, 0, 0};
            if (off < WIN) {
#pragma unroll
               for (int k = 0; k < 16; k++)
                  if (g + (uint64_t)k < a.nbytes) w[k >> 2] |= (uint32_t)a.text[g + k] << ((k & 3) * 8);
            }
            *reinterpret_cast<fused_v4u *>(s_text + off) = fused_v4u{w[0], w[1], w[2], w[3]};
         }
      }
      prefetch(tile + gridDim.x);                        /* in flight during everything below */
      __syncthreads();
      /* ---- 2. newlines of the owned range [0, tb) -> ranks ---- */
      /* A newline at q starts a line at q+1 unless it is the last byte of the buffer.  Thread t owns
         the R consecutive 16-byte pieces [t*R, (t+1)*R), so line order = (thread, piece, bit). */
      const uint32_t npieces = (tb + 15) >> 4;
      const uint32_t R = (npieces + NT - 1) / NT;                           /* <= FUSED_MAXR */
      auto exact_mask = [&](uint32_t piece) -> uint32_t {
         const fused_v4u v = *reinterpret_cast<const fused_v4u *>(s_text + piece * 16);
         const uint32_t f0 = nl_flags(v.x), f1 = nl_flags(v.y), f2 = nl_flags(v.z), f3 = nl_flags(v.w);
         uint32_t m16 = (((f0 >> 7) * 0x00204081u >> 21) & 0xFu) | ((((f1 >> 7) * 0x00204081u >> 21) & 0xFu) << 4) |
                        ((((f2 >> 7) * 0x00204081u >> 21) & 0xFu) << 8) | ((((f3 >> 7) * 0x00204081u >> 21) & 0xFu) << 12);
         const uint32_t q0 = piece * 16;
         if (q0 + 16 > tb) m16 &= (1u << (tb - q0)) - 1u;                   /* q < tb */
         const uint64_t last = a.nbytes - 1;                                 /* q + 1 < nbytes */
         if (t0 + q0 <= last && last < t0 + q0 + 16) m16 &= ~(1u << (uint32_t)(last - (t0 + q0)));
         return m16;
      };
      uint32_t pmask = 0, cnt = 0;                                          /* pieces with newlines; their count */
      {
         const uint32_t p0 = (uint32_t)tid * R;
         const uint32_t pend = p0 + R < npieces ? p0 + R : npieces;
#pragma unroll 2
         for (uint32_t piece = p0; piece < pend; piece++) {
            const fused_v4u v = *reinterpret_cast<const fused_v4u *>(s_text + piece * 16);
            /* cheap superset test (a borrow can flag the byte above a newline), exact mask only then */
            const uint32_t x0 = v.x ^ 0x0A0A0A0Au, x1 = v.y ^ 0x0A0A0A0Au, x2 = v.z ^ 0x0A0A0A0Au, x3 = v.w ^ 0x0A0A0A0Au;
            const uint32_t any = (((x0 - 0x01010101u) & ~x0) | ((x1 - 0x01010101u) & ~x1) |
                                  ((x2 - 0x01010101u) & ~x2) | ((x3 - 0x01010101u) & ~x3)) & 0x80808080u;
            if (any) {
               const uint32_t m16 = exact_mask(piece);
               if (m16) { pmask |= 1u << (piece - p0); cnt += (uint32_t)__popc(m16); }
            }
         }
      }
      const uint32_t incl = wave_incl_scan_u32(cnt);
      if (lane == 63) s_wtot[wave] = incl;
      __syncthreads();
      const uint32_t extra = (a.first_seg && tile == 0) ? 1u : 0u;          /* the line starting at byte 0 */
      uint32_t my_base = extra + incl - cnt, nl_tile = extra;
#pragma unroll
      for (int w = 0; w < NW; w++) {
         const uint32_t tw = s_wtot[w];
         if (w < wave) my_base += tw;
         nl_tile += tw;                                                     /* raw lines owned by the tile */
      }

      uint32_t tile_hdrs = 0, tile_hits = 0;
      for (uint32_t r0 = 0; r0 < nl_tile; r0 += CAPL) {
         const uint32_t npass = nl_tile - r0 < CAPL ? nl_tile - r0 : CAPL;
         __syncthreads();                                                   /* s_starts / s_hit free */
         if (extra && r0 == 0 && tid == 0) s_starts[0] = 0;
         {
            uint32_t r = my_base, pm = pmask;
            while (pm) {
               const uint32_t j = (uint32_t)__builtin_ctz(pm);
               pm &= pm - 1;
               const uint32_t piece = (uint32_t)tid * R + j;
               uint32_t mm = exact_mask(piece);
               while (mm) {
                  const uint32_t b = (uint32_t)__builtin_ctz(mm);
                  mm &= mm - 1;
                  if (r >= r0 && r < r0 + CAPL) s_starts[r - r0] = piece * 16 + b + 1;
                  r++;
               }
            }
         }
         __syncthreads();
         /* ---- 3. one line per lane ---- */
         const uint32_t niter = (npass + NT - 1) / NT;
         for (uint32_t it = 0; it < niter; it++) {
            const uint32_t rl = it * NT + tid;                              /* rank within the pass */
            bool active = rl < npass;
            uint32_t p = active ? s_starts[rl] : 0;                         /* LDS offset of the next chunk */
            const uint32_t lstart = p;
            bool hdr = false;
            if (fasta && active && s_text[p] == '>') { hdr = true; active = false; }
            uint32_t pv = 0xFFFFFFFFu, mv = 0u, score = (uint32_t)a.m, minscore = (uint32_t)a.m;
            bool hit = false, toolong = false;
            const uint32_t two = 2u;
            if (a.debug & 1u) active = false;
            while (__any(active)) {
               if (active && p + 16 > WIN) { toolong = true; active = false; }
               /* 16 text bytes at arbitrary alignment: 5 aligned dwords + funnel shifts */
               const uint32_t *src = reinterpret_cast<const uint32_t *>(s_text + (p & ~3u));
               const uint32_t sh = p & 3u;
               const uint32_t d0 = src[0], d1 = src[1], d2 = src[2], d3 = src[3], d4 = src[4];
               uint32_t w[4];
               w[0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
               w[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
               w[2] = __builtin_amdgcn_alignbyte(d3, d2, sh);
               w[3] = __builtin_amdgcn_alignbyte(d4, d3, sh);
               /* EQ[byte]: the table sits at LDS offset 0, so the address is byte << 2 -- one SDWA
                  instruction per character (byte select + shift) */
               uint32_t eq[16];
#pragma unroll
               for (int k = 0; k < 16; k += 4) {
                  uint32_t a0, a1, a2, a3;
                  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0"
                      : "=v"(a0) : "v"(two), "v"(w[k >> 2]));
                  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"
                      : "=v"(a1) : "v"(two), "v"(w[k >> 2]));
                  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2"
                      : "=v"(a2) : "v"(two), "v"(w[k >> 2]));
                  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3"
                      : "=v"(a3) : "v"(two), "v"(w[k >> 2]));
                  eq[k + 0] = *reinterpret_cast<const uint32_t *>(smem + a0);
                  eq[k + 1] = *reinterpret_cast<const uint32_t *>(smem + a1);
                  eq[k + 2] = *reinterpret_cast<const uint32_t *>(smem + a2);
                  eq[k + 3] = *reinterpret_cast<const uint32_t *>(smem + a3);
               }
               uint32_t fl[4];
#pragma unroll
               for (int g = 0; g < 4; g++) fl[g] = (eq[4 * g] | eq[4 * g + 1] | eq[4 * g + 2] | eq[4 * g + 3]) & FUSED_FLAGS;
               const uint32_t flall = fl[0] | fl[1] | fl[2] | fl[3];
               if (!__any(active && flall != 0)) {
                  /* fast path: no lane of the wave sees a special byte in these 16 */
#pragma unroll
                  for (int k = 0; k < 16; k++) {
                     fused_step(eq[k], pv, mv, score);
                     minscore = score < minscore ? score : minscore;
                  }
                  if (active) p += 16;
               } else {
#pragma unroll
                  for (int g = 0; g < 4; g++) {
                     if (!__any(active && fl[g] != 0)) {
#pragma unroll
                        for (int k = 4 * g; k < 4 * g + 4; k++) {
                           fused_step(eq[k], pv, mv, score);
                           minscore = score < minscore ? score : minscore;
                        }
                     } else {
#pragma unroll
                        for (int k = 4 * g; k < 4 * g + 4; k++) {
                           if (active) {
                              const uint32_t e = eq[k];
                              if ((e & FUSED_FLAGS) == 0) {
                                 fused_step(e, pv, mv, score);
                                 minscore = score < minscore ? score : minscore;
                              } else if (e & FUSED_FLAG_TERM) {
                                 active = false;                            /* line over: latch the verdict */
                                 hit = minscore <= tau;
                              }
                           }
                        }
                     }
                  }
                  if (active) p += 16;
               }
            }
            if (toolong) {
               /* the line leaves the LDS window: scan it from HBM with the generic per-line function */
               hit = sq_scan_line<1, SQ_MODE_ANY>(a.text, a.nbytes, t0 + lstart, (const uint32_t *)s_peq,
                                                  (const uint32_t *)(s_peq + 5), (const uint8_t *)s_lut, a.m, a.tau,
                                                  a.options & 3, 0, nullptr, 0) != 0;
            }
            const uint64_t hm = __ballot(hit);
            const uint64_t dm = __ballot(hdr);
            if (lane == 0) { s_hit[it * NW + wave] = hm; s_hdr[it * NW + wave] = dm; }
         }
         __syncthreads();
         /* ---- 4. ordered compaction of this pass's hit lines ---- */
         /* Masks are wave-uniform: totals and prefixes are scalar work (readfirstlane + s_bcnt1). */
         auto uni64 = [](uint64_t v) -> uint64_t {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
            return ((uint64_t)hi << 32) | lo;
         };
         uint32_t pass_hits = 0, pass_hdrs = 0;
         for (uint32_t i = 0; i < niter * NW; i++) {
            pass_hits += (uint32_t)__popcll(uni64(s_hit[i]));
            pass_hdrs += (uint32_t)__popcll(uni64(s_hdr[i]));
         }
         const bool keep = a.want != SEEQDEV_WANT_COUNTLINES;
         if (pass_hits && keep) {
            if (slice_pos + pass_hits <= a.slice_cap) {
               uint4 *slice = a.tmp + (size_t)blockIdx.x * a.slice_cap + slice_pos;
               uint32_t hb = 0, db = 0;                                     /* hits / headers before (it, wave) */
               for (uint32_t it = 0; it < niter; it++) {
                  for (int w = 0; w < NW; w++) {
                     const uint64_t hm = uni64(s_hit[it * NW + w]), dm = uni64(s_hdr[it * NW + w]);
                     if (w == uwave && ((hm >> lane) & 1)) {
                        const uint32_t below_h = __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32),
                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0));
                        const uint32_t below_d = __builtin_amdgcn_mbcnt_hi((uint32_t)(dm >> 32),
                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)dm, 0));
                        const uint32_t rl = it * NT + tid;
                        const uint32_t seq = hb + below_h;
                        /* counted rank inside the tile = raw rank - headers before it */
                        const uint32_t crank = r0 + rl - (tile_hdrs + db + below_d);
                        const uint64_t start_seg = (uint64_t)tile * TB + s_starts[rl];      /* segment-relative */
                        slice[seq] = make_uint4(tile, tile_hits + seq, (uint32_t)start_seg, crank);
                     }
                     hb += (uint32_t)__popcll(hm);
                     db += (uint32_t)__popcll(dm);
                  }
               }
               slice_pos += pass_hits;
            } else {
               wg_overflow = true;
            }
         }
         tile_hits += pass_hits;
         tile_hdrs += pass_hdrs;
      }
      if (tid == 0) {
         a.tile_cl[tile] = nl_tile - tile_hdrs;
         a.tile_hits[tile] = tile_hits;
      }
      wg_lines += nl_tile;
      wg_hdrs += tile_hdrs;
      wg_hitlines += tile_hits;
   }
   if (tid == 0) {
      a.wg_hits[blockIdx.x] = wg_overflow ? 0u : slice_pos;
      a.wg_part[3 * blockIdx.x + 0] = wg_lines;
      a.wg_part[3 * blockIdx.x + 1] = wg_hdrs;
      a.wg_part[3 * blockIdx.x + 2] = wg_overflow ? (wg_hitlines | 0x80000000u) : wg_hitlines;
   }
}

/* After k_fused / k_direct: reduce the per-slice partial counts (no atomics in the hot kernels)
   and publish the hit-line count of the segment, or the overflow. */
__global__ __launch_bounds__(256) void k_fused_post(FusedArgs a, uint32_t nslices)
{
   __shared__ uint32_t s_red[4][4];
   uint32_t lines = 0, hdrs = 0, hits = 0, mx = 0, ovf = 0, lastnl = 0;
   for (uint32_t i = threadIdx.x; i < nslices; i += 256) {
      if (a.wg_lastnl) { const uint32_t l = a.wg_lastnl[i]; lastnl = l > lastnl ? l : lastnl; }
      lines += a.wg_part[3 * i + 0];
      hdrs += a.wg_part[3 * i + 1];
      const uint32_t h = a.wg_part[3 * i + 2];
      hits += h & 0x7FFFFFFFu;
      mx = (h & 0x7FFFFFFFu) > mx ? (h & 0x7FFFFFFFu) : mx;
      ovf |= h >> 31;
   }
#pragma unroll
   for (int d = 32; d >= 1; d >>= 1) {
      lines += __shfl_xor(lines, d, 64);
      hdrs += __shfl_xor(hdrs, d, 64);
      hits += __shfl_xor(hits, d, 64);
      const uint32_t o = __shfl_xor(mx, d, 64);
      mx = o > mx ? o : mx;
      ovf |= __shfl_xor(ovf, d, 64);
      const uint32_t ol = __shfl_xor(lastnl, d, 64);
      lastnl = ol > lastnl ? ol : lastnl;
   }
   __shared__ uint32_t s_last[4];
   const int w = threadIdx.x >> 6;
   if ((threadIdx.x & 63) == 0) s_last[w] = lastnl;
   if ((threadIdx.x & 63) == 0) { s_red[w][0] = lines; s_red[w][1] = hdrs; s_red[w][2] = hits; s_red[w][3] = mx | (ovf << 31); }
   __syncthreads();
   if (threadIdx.x == 0) {
      lines = hdrs = hits = mx = ovf = 0;
      for (int k = 0; k < 4; k++) {
         lines += s_red[k][0]; hdrs += s_red[k][1]; hits += s_red[k][2];
         const uint32_t m = s_red[k][3] & 0x7FFFFFFFu;
         mx = m > mx ? m : mx;
         ovf |= s_red[k][3] >> 31;
      }
      Counters *c = a.cnt;
      c->seg_nlines = lines;
      c->seg_nheaders = hdrs;
      /* capacity wanted next time: every slice as large as the fullest one, plus slack */
      const uint64_t need = (uint64_t)mx * nslices + (uint64_t)nslices * 64;
      if (need > c->need_hitlines) c->need_hitlines = need > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)need;
      if (ovf) { atomicOr(&c->overflow, 2u); hits = 0; }
      c->seg_nhitlines = hits;
      c->seg_nrec = hits;                                   /* (k_seg_mid's job; the slices cannot hold more than cap_hitlines) */
      if (hits > c->need_hitlines) c->need_hitlines = hits;
      c->seg_tmp_hits = 0;
      lastnl = 0;
      for (int k = 0; k < 4; k++) lastnl = s_last[k] > lastnl ? s_last[k] : lastnl;
      c->seg_last_nl = lastnl;
   }
}

/* Slices -> ordered (hit_start, hit_line).  tile_hits / tile_cl hold exclusive prefixes by now.
   One workgroup per k_fused workgroup slice. */
__global__ __launch_bounds__(256) void k_fused_reorder(FusedArgs a, uint32_t *hit_start, uint32_t *hit_line)
{
   const Counters *c = a.cnt;
   if (c->overflow & 2u) return;
   const uint32_t n = a.wg_hits[blockIdx.x];
   const uint4 *slice = a.tmp + (size_t)blockIdx.x * a.slice_cap;
   for (uint32_t i = threadIdx.x; i < n; i += 256) {
      const uint4 e = slice[i];
      const uint32_t dst = a.tile_hits[e.x] + e.y;
      hit_start[dst] = e.z;
      hit_line[dst] = (uint32_t)(c->lines + a.tile_cl[e.x] + e.w + 1);      /* 1-based, reference seeq.c:377 */
   }
}

#endif
