/*
 * seeq_device.hip -- HIP kernels and the device-level C-ABI (include/seeq_amd.h)
 * of seeq-mi355x.  Written for gfx950 (MI355X, CDNA4): 64-wide wavefronts,
 * one text line per lane, wave ballots for per-line flags, LDS for the Peq and
 * class tables.  Integer/bitwise work only -- no MFMA.
 *
 * Pipeline of one seeqdevScanRun over a text buffer resident in HBM, per
 * segment of < 4 GiB (all offsets inside a segment are u32):
 *
 *   K0  k_nl_count / k_nl_write      newline index -> line_start[]        (HBM stream)
 *   K1  k_forward<W>                 one line per lane: Myers column per character,
 *                                    acceptance rules; ballot -> hitmask[] (1 bit/line)
 *   K2  scan of popc(hitmask)        ordered ranks of hit lines
 *   K3  k_compact                    hitlines[]
 *   K4  k_exact<W,COUNT>             (SQ_ALL / COUNTMATCH) hits per hit line, then scan
 *   K5  k_exact<W,EMIT>              acceptance rules + reverse start recovery -> records[]
 *
 * replacing the reference's per-line loop seeq.c:361-387 -> libseeq.c:171-352.
 * There is no host-side matcher: without a GPU every entry point fails.
 */
#include <hip/hip_runtime.h>

#include <errno.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "seeq_amd.h"
#include "seeq_kernel_core.h"
extern "C" {
#include "seeq_pattern.h"
}

/* ========================================================================== */
/* Error plumbing                                                             */
/* ========================================================================== */
static thread_local char g_last_error[256] = "";

static int hip_fail(hipError_t e, const char *what, int err_no)
{
   snprintf(g_last_error, sizeof g_last_error, "%s: %s", what, hipGetErrorString(e));
   seeqerr = 0;
   errno = err_no;
   return -1;
}

#define HIP_TRY(call, err_no)                                        \
   do {                                                              \
      hipError_t e_ = (call);                                        \
      if (e_ != hipSuccess) return hip_fail(e_, #call, (err_no));    \
   } while (0)

extern "C" const char *seeqdevLastError(void) { return g_last_error; }

/* Every entry point that takes a scan context or a pattern runs on THAT object's device, whatever device the calling
 * thread used last (a caller that spreads chunks over several GPUs goes from one context to the next). */
static int use_device(int device)
{
   int cur = -1;
   if (hipGetDevice(&cur) == hipSuccess && cur == device) return 0;
   HIP_TRY(hipSetDevice(device), ENODEV);
   return 0;
}

extern "C" int seeqdevDeviceCount(void)
{
   int n = 0;
   if (hipGetDeviceCount(&n) != hipSuccess) return 0;
   return n;
}

extern "C" int seeqdevSetDevice(int device)
{
   HIP_TRY(hipSetDevice(device), ENODEV);
   return 0;
}

#include "seeq_types.h"
#include "seeq_scan_common.h"

static constexpr int WG = 256;           /* 4 waves */
static constexpr int TILE = 16384;       /* bytes per newline-index workgroup: 64 B per thread */
static constexpr size_t FUSED_MIN_TILE = 512;    /* smallest text tile / region of the one-pass kernels */
static constexpr int STREAM_NW_HOST = 16;         /* = STREAM_NW (seeq_stream.h): waves per k_stream workgroup */
static constexpr size_t MAX_FUSED_GRID = 16384;   /* upper bound of the waves (= hit slices) of a persistent scan grid */
static constexpr size_t SAMPLE_BYTES = 65536;     /* prefix sampled to estimate the line length */

/* ========================================================================== */
/* Block-level helpers                                                        */
/* ========================================================================== */
/* 64-bit mask of newline positions among the 64 bytes this thread owns
 * (bytes seg_base + tile*TILE + tid*64 ...), restricted to positions q with
 * q < seg_base+seg_len and q + 1 < nbytes (a final '\n' starts no line). */
__device__ __forceinline__ uint64_t thread_nl_mask(const ScanArgs &a, uint32_t tile)
{
   const uint64_t seg_off = (uint64_t)tile * TILE + (uint64_t)threadIdx.x * 64;
   if (seg_off >= a.seg_len) return 0;
   const uint64_t q0 = a.seg_base + seg_off;
   uint64_t limit = a.seg_base + a.seg_len;                /* exclusive */
   if (a.nbytes - 1 < limit) limit = a.nbytes - 1;         /* q + 1 < nbytes  (nbytes > 0 here) */
   uint64_t mask = 0;
   if (q0 + 64 <= limit && ((uintptr_t)(a.text + q0) & 15) == 0) {
      const uint4 *p = reinterpret_cast<const uint4 *>(a.text + q0);
#pragma unroll
      for (int j = 0; j < 4; j++) {
         const uint4 v = p[j];
         const uint32_t f0 = nl_flags(v.x), f1 = nl_flags(v.y), f2 = nl_flags(v.z), f3 = nl_flags(v.w);
         /* gather bit 7 of each byte into 4 consecutive bits */
         const uint32_t b0 = ((f0 >> 7) * 0x00204081u >> 21) & 0xFu;
         const uint32_t b1 = ((f1 >> 7) * 0x00204081u >> 21) & 0xFu;
         const uint32_t b2 = ((f2 >> 7) * 0x00204081u >> 21) & 0xFu;
         const uint32_t b3 = ((f3 >> 7) * 0x00204081u >> 21) & 0xFu;
         const uint64_t m16 = b0 | (b1 << 4) | (b2 << 8) | (b3 << 12);
         mask |= m16 << (16 * j);
      }
   } else {
      for (int k = 0; k < 64; k++) {
         const uint64_t q = q0 + k;
         if (q < limit && a.text[q] == '\n') mask |= 1ull << k;
      }
   }
   return mask;
}

/* ========================================================================== */
/* K0: newline index                                                          */
/* ========================================================================== */
__global__ __launch_bounds__(WG) void k_nl_count(ScanArgs a)
{
   __shared__ uint32_t s_wave[4];
   const uint64_t m = thread_nl_mask(a, blockIdx.x);
   uint32_t tot;
   block_excl_scan((uint32_t)__popcll(m), &tot, s_wave);
   if (threadIdx.x == 0) a.tile_cnt[blockIdx.x] = tot;
}

/* After tile_cnt has been scanned in place (exclusive) and the total written
 * to cnt->seg_nlines: add the line that starts at byte 0 of the buffer. */
__global__ void k_index_finalize(ScanArgs a)
{
   Counters *c = a.cnt;
   uint32_t n = c->seg_nlines;
   if (a.first_seg && a.nbytes > 0) n += 1;
   if (n > a.cap_lines) {
      atomicOr(&c->overflow, 1u);
      if (n > c->need_lines) c->need_lines = n;
      n = 0;                       /* later kernels of this segment do nothing */
   } else if (n > c->need_lines) {
      c->need_lines = n;
   }
   c->seg_nlines = n;
   if (n && a.first_seg) a.line_start[0] = 0;
}

__global__ __launch_bounds__(WG) void k_nl_write(ScanArgs a)
{
   __shared__ uint32_t s_wave[4];
   if (a.cnt->seg_nlines == 0) return;
   uint64_t m = thread_nl_mask(a, blockIdx.x);
   uint32_t tot;
   uint32_t rank = block_excl_scan((uint32_t)__popcll(m), &tot, s_wave);
   rank += a.tile_cnt[blockIdx.x] + (a.first_seg ? 1u : 0u);
   const uint32_t off0 = blockIdx.x * TILE + threadIdx.x * 64 + 1;   /* start = newline position + 1 */
   while (m) {
      const int b = __builtin_ctzll(m);
      m &= m - 1;
      a.line_start[rank++] = off0 + (uint32_t)b;
   }
}

/* ========================================================================== */
/* Generic two-level exclusive scan over u32 items with a device-side length   */
/*   XF 0: in = u32[];  XF 1: in = u64[], item = popcount;  XF 2: u32 != 0     */
/*   n = (*n_ptr + add) >> shift                                              */
/* ========================================================================== */
static constexpr int SCAN_ITEMS = 8;                     /* per thread */
static constexpr int SCAN_BLOCK = WG * SCAN_ITEMS;       /* 2048 per block */

template <int XF>
__device__ __forceinline__ uint32_t scan_item(const void *in, uint32_t i)
{
   if (XF == 0) return reinterpret_cast<const uint32_t *>(in)[i];
   if (XF == 2) return reinterpret_cast<const uint32_t *>(in)[i] != 0u ? 1u : 0u;
   return (uint32_t)__popcll(reinterpret_cast<const uint64_t *>(in)[i]);
}

template <int XF>
__global__ __launch_bounds__(WG) void k_scan_reduce(const void *in, uint32_t *bsum, const uint32_t *n_ptr, uint32_t add,
                                                    uint32_t shift)
{
   __shared__ uint32_t s_wave[4];
   const uint32_t n = n_ptr ? (*n_ptr + add) >> shift : add;
   const uint32_t base = blockIdx.x * SCAN_BLOCK;
   if (base >= n) return;
   uint32_t v = 0;
#pragma unroll
   for (int k = 0; k < SCAN_ITEMS; k++) {
      const uint32_t i = base + threadIdx.x * SCAN_ITEMS + k;
      if (i < n) v += scan_item<XF>(in, i);
   }
   uint32_t tot;
   block_excl_scan(v, &tot, s_wave);
   if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

/* One block: exclusive scan of bsum[0..nb) in place, total -> *total_out. */
__global__ __launch_bounds__(WG) void k_scan_top(uint32_t *bsum, const uint32_t *n_ptr, uint32_t add, uint32_t shift,
                                                 uint32_t *total_out)
{
   __shared__ uint32_t s_wave[4];
   const uint32_t n = n_ptr ? (*n_ptr + add) >> shift : add;
   const uint32_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
   uint32_t running = 0;
   for (uint32_t b0 = 0; b0 < nb; b0 += WG) {
      const uint32_t i = b0 + threadIdx.x;
      const uint32_t v = i < nb ? bsum[i] : 0;
      uint32_t tot;
      const uint32_t ex = block_excl_scan(v, &tot, s_wave);
      if (i < nb) bsum[i] = running + ex;
      running += tot;
   }
   if (threadIdx.x == 0) *total_out = running;
}

template <int XF>
__global__ __launch_bounds__(WG) void k_scan_apply(const void *in, uint32_t *out, const uint32_t *bsum,
                                                   const uint32_t *n_ptr, uint32_t add, uint32_t shift)
{
   __shared__ uint32_t s_wave[4];
   const uint32_t n = n_ptr ? (*n_ptr + add) >> shift : add;
   const uint32_t base = blockIdx.x * SCAN_BLOCK;
   if (base >= n) return;
   uint32_t item[SCAN_ITEMS];
   uint32_t v = 0;
#pragma unroll
   for (int k = 0; k < SCAN_ITEMS; k++) {
      const uint32_t i = base + threadIdx.x * SCAN_ITEMS + k;
      item[k] = i < n ? scan_item<XF>(in, i) : 0;
      v += item[k];
   }
   uint32_t tot;
   uint32_t ex = block_excl_scan(v, &tot, s_wave) + bsum[blockIdx.x];
#pragma unroll
   for (int k = 0; k < SCAN_ITEMS; k++) {
      const uint32_t i = base + threadIdx.x * SCAN_ITEMS + k;
      if (i < n) out[i] = ex;
      ex += item[k];
   }
}

/* ========================================================================== */
/* K1: forward scan, one line per lane, 64 consecutive lines per wave          */
/* ========================================================================== */
template <int W>
__device__ __forceinline__ void load_tables(const ScanArgs &a, uint32_t *s_peq, uint8_t *s_lut)
{
   for (int i = threadIdx.x; i < 10 * W; i += WG) {
      /* a.peq holds [2][5][Wp] with Wp = words of the pattern; pad to W */
      const int Wp = (a.m + 31) >> 5;
      const int dir = i / (5 * W), rem = i % (5 * W), cls = rem / W, w = rem % W;
      s_peq[i] = w < Wp ? a.peq[(dir * 5 + cls) * Wp + w] : 0u;
   }
   for (int b = threadIdx.x; b < 256; b += WG) s_lut[b] = sq_class_of((uint32_t)b, a.options);
   __syncthreads();
}

template <int W>
__global__ __launch_bounds__(WG) void k_forward(ScanArgs a)
{
   __shared__ uint32_t s_peq[10 * W];
   __shared__ uint8_t s_lut[256];
   load_tables<W>(a, s_peq, s_lut);
   const uint32_t nlines = a.cnt->seg_nlines;
   const int lane = threadIdx.x & 63;
   const uint32_t wave = (blockIdx.x * WG + threadIdx.x) >> 6;
   const uint32_t nwaves = (gridDim.x * WG) >> 6;
   const bool fasta = (a.options & SEEQDEV_FASTA) != 0;
   const uint32_t nchunks = (nlines + 63) >> 6;
   for (uint32_t chunk = wave; chunk < nchunks; chunk += nwaves) {
      const uint32_t idx = chunk * 64 + lane;
      bool hit = false, hdr = false;
      if (idx < nlines) {
         const uint64_t off = a.seg_base + a.line_start[idx];
         if (fasta && a.text[off] == '>') hdr = true;   /* off < nbytes: every line has >= 1 byte */
         else
            hit = sq_scan_line<W, SQ_MODE_ANY>(a.text, a.nbytes, off, (const uint32_t *)s_peq,
                                               (const uint32_t *)(s_peq + 5 * W), (const uint8_t *)s_lut, a.m, a.tau,
                                               a.options & 3, 0, nullptr, 0) != 0;
      }
      const uint64_t hm = __ballot(hit);
      const uint64_t dm = __ballot(hdr);
      if (lane == 0) {
         a.hitmask[chunk] = hm;
         if (fasta) a.hdrmask[chunk] = dm;
      }
   }
}

__device__ __forceinline__ uint32_t counted_line_no(const ScanArgs &a, uint32_t idx, bool fasta)
{
   /* 1-based index among counted lines of the whole buffer (reference seeq.c:377) */
   uint64_t n = a.cnt->lines + idx + 1;
   if (fasta) {
      const uint32_t chunk = idx >> 6;
      n -= a.hdr_off[chunk] + (uint32_t)__popcll(a.hdrmask[chunk] & ((1ull << (idx & 63)) - 1));
   }
   return (uint32_t)n;
}

/* ========================================================================== */
/* K3: ordered compaction of hit lines                                        */
/* ========================================================================== */
__global__ __launch_bounds__(WG) void k_compact(ScanArgs a)
{
   const uint32_t nlines = a.cnt->seg_nlines;
   const uint32_t nchunks = (nlines + 63) >> 6;
   const int lane = threadIdx.x & 63;
   const uint32_t wave = (blockIdx.x * WG + threadIdx.x) >> 6;
   const uint32_t nwaves = (gridDim.x * WG) >> 6;
   const bool fasta = (a.options & SEEQDEV_FASTA) != 0;
   for (uint32_t chunk = wave; chunk < nchunks; chunk += nwaves) {
      const uint64_t hm = a.hitmask[chunk];
      if ((hm >> lane) & 1) {
         const uint32_t k = a.wave_off[chunk] + (uint32_t)__popcll(hm & ((1ull << lane) - 1));
         if (k < a.cap_hitlines) {
            const uint32_t idx = chunk * 64 + lane;
            a.hit_start[k] = a.line_start[idx];
            a.hit_line[k] = counted_line_no(a, idx, fasta);
         }
      }
   }
}

/* After compaction: overflow check of the hit-line list; for FIRST/BEST the
 * number of records of the segment is the number of hit lines. */
__global__ void k_seg_mid(ScanArgs a)
{
   Counters *c = a.cnt;
   uint32_t nhl = c->seg_nhitlines;
   if (nhl > c->need_hitlines) c->need_hitlines = nhl;
   if (nhl > a.cap_hitlines) {
      atomicOr(&c->overflow, 2u);
      nhl = 0;
      c->seg_nhitlines = 0;      /* totals of this run are void anyway */
   }
   c->seg_nrec = nhl;            /* overwritten by the nh scan for SQ_ALL / COUNTMATCH */
}

__global__ void k_rec_check(ScanArgs a) { rec_check_body(a); }

/* ========================================================================== */
/* K4/K5: exact pass over the hit lines                                       */
/* ========================================================================== */
template <int W, int MODE>
__global__ __launch_bounds__(WG) void k_exact(ScanArgs a)
{
   __shared__ uint32_t s_peq[10 * W];
   __shared__ uint8_t s_lut[256];
   load_tables<W>(a, s_peq, s_lut);
   const Counters *c = a.cnt;
   const uint32_t nhl = c->seg_nhitlines;
   const int match_opt = a.options & 3;
   if (MODE == SQ_MODE_EMIT && (c->overflow & 4u)) return;
   const uint32_t stride = gridDim.x * WG;
   for (uint32_t k = blockIdx.x * WG + threadIdx.x; k < nhl; k += stride) {
      const uint64_t off = a.seg_base + a.hit_start[k];
      if (MODE == SQ_MODE_COUNT) {
         a.nh[k] = sq_scan_line<W, SQ_MODE_COUNT>(a.text, a.nbytes, off, (const uint32_t *)s_peq,
                                                  (const uint32_t *)(s_peq + 5 * W), (const uint8_t *)s_lut, a.m,
                                                  a.tau, match_opt, 0, nullptr, 0);
      } else {
         const uint32_t line_no = a.hit_line[k];
         uint64_t dst;
         uint32_t cap;
         if (match_opt == SQ_ALL) {
            dst = c->records + a.nh[k];
            cap = 0xFFFFFFFFu;       /* exact count known from the COUNT pass */
         } else {
            dst = c->records + (a.use_nh ? a.nh[k] : k);
            cap = 1;
         }
         sq_scan_line<W, SQ_MODE_EMIT>(a.text, a.nbytes, off, (const uint32_t *)s_peq,
                                       (const uint32_t *)(s_peq + 5 * W), (const uint8_t *)s_lut, a.m, a.tau,
                                       match_opt, line_no, reinterpret_cast<sq_hit_t *>(a.records + dst), cap);
      }
   }
}

/* Per record: where its line starts in the buffer (lets the host jump from hit to hit instead of
   walking every line: the replay of seeqFileMatch, seeq.c:361-386, becomes O(hits)). */
__global__ __launch_bounds__(WG) void k_rec_offsets(ScanArgs a)
{
   const Counters *c = a.cnt;
   if (c->overflow & 4u) return;
   const uint32_t nhl = c->seg_nhitlines;
   const bool all = (a.options & 3) == SQ_ALL || a.use_nh;
   const uint32_t stride = gridDim.x * WG;
   for (uint32_t k = blockIdx.x * WG + threadIdx.x; k < nhl; k += stride) {
      const uint64_t off = a.seg_base + a.hit_start[k];
      if (all) {
         const uint32_t lo = a.nh[k], hi = k + 1 < nhl ? a.nh[k + 1] : c->seg_nrec;
         for (uint32_t r = lo; r < hi; r++) a.rec_off[c->records + r] = off;
      } else {
         a.rec_off[c->records + k] = off;
      }
   }
}

/* Lines with >= 1 verified hit, from the per-line counts (before they are scanned into offsets). */
__device__ __forceinline__ void count_nonzero_body(const ScanArgs &a)
{
   __shared__ uint32_t s_n[WG / 64];
   const uint32_t nhl = a.cnt->seg_nhitlines;
   const uint32_t stride = gridDim.x * WG;
   uint32_t n = 0;
   for (uint32_t k = blockIdx.x * WG + threadIdx.x; k < nhl; k += stride) n += a.nh[k] != 0;
#pragma unroll
   for (int d = 32; d >= 1; d >>= 1) n += __shfl_xor(n, d, 64);
   if ((threadIdx.x & 63) == 0) s_n[threadIdx.x >> 6] = n;
   __syncthreads();
   if (threadIdx.x == 0) {                                 /* one atomic per block: same-address atomics serialise */
      n = 0;
      for (int w = 0; w < WG / 64; w++) n += s_n[w];
      if (n) atomicAdd(&a.cnt->seg_nmatch, n);
   }
}
__global__ __launch_bounds__(WG) void k_count_nonzero(ScanArgs a) { count_nonzero_body(a); }

__global__ void k_seg_end(ScanArgs a, int flags) { seg_end_body(a, flags); }

/* SINGLELINE: the buffer is one string -> one line starting at 0. */
__global__ void k_single_line(ScanArgs a)
{
   a.cnt->seg_nlines = 1;
   if (a.cnt->need_lines < 1) a.cnt->need_lines = 1;
   a.line_start[0] = 0;
}

#include "seeq_fused_post.h"
#include "seeq_direct.h"
#include "seeq_exact1.h"
#include "seeq_stream.h"
#include "seeq_pair.h"
#include "seeq_packed.h"
#include "seeq_multi.h"
#include "seeq_post.h"
static_assert(STREAM_NW == STREAM_NW_HOST, "waves per k_stream workgroup");
extern "C" {
#include "seeq_dfa.h"
}

/* ========================================================================== */
/* Synthetic reads (bench / test input; CPU twin: oracle/seeq_oracle.c)        */
/* ========================================================================== */
__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
   x += 0x9E3779B97F4A7C15ull;
   uint64_t z = x;
   z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
   z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
   return z ^ (z >> 31);
}

struct SynthArgs {
   uint8_t *out;
   uint64_t first, n, seed;
   int len, plen, tau;
   char pattern[96];
};

__global__ __launch_bounds__(WG) void k_synth(SynthArgs s)
{
   const uint64_t k = (uint64_t)blockIdx.x * WG + threadIdx.x;
   if (k >= s.n) return;
   const uint64_t r = s.first + k;
   uint8_t *line = s.out + k * (uint64_t)(s.len + 1);
   const char B[4] = {'A', 'C', 'G', 'T'};
   for (int p = 0; p < s.len; p++) line[p] = B[splitmix64(s.seed ^ (r * 256 + (uint64_t)p)) >> 62];
   line[s.len] = '\n';
   const uint64_t hr = splitmix64(s.seed ^ 0xA5A5A5A5DEADBEEFull ^ (r * 0x100000001B3ull));
   if ((hr & 15) == 0 && s.plen > 0 && s.plen + s.tau + 2 <= 96 && s.plen + s.tau + 2 <= s.len) {
      char t[96];
      int cur = s.plen;
      for (int i = 0; i < s.plen; i++) t[i] = s.pattern[i];
      const int e = (int)((hr >> 4) % (uint64_t)(s.tau + 3));
      for (int q = 0; q < e; q++) {
         const uint64_t hk = splitmix64(hr + (uint64_t)q + 1);
         const int type = (int)(hk % 3);
         const int pos = (int)((hk >> 8) % (uint64_t)cur);
         const char b = B[(hk >> 40) & 3];
         if (type == 0) t[pos] = b;
         else if (type == 1) {
            for (int u = cur; u > pos; u--) t[u] = t[u - 1];
            t[pos] = b;
            cur++;
         } else if (cur > 1) {
            for (int u = pos; u < cur - 1; u++) t[u] = t[u + 1];
            cur--;
         }
      }
      const int off = (int)((hr >> 20) % (uint64_t)(s.len - cur + 1));
      for (int i = 0; i < cur; i++) line[off + i] = (uint8_t)t[i];
   }
   const uint64_t hn = splitmix64(s.seed ^ 0x5BD1E9955BD1E995ull ^ (r * 0x9E3779B1ull));
   if ((hn & 255) == 0) line[(hn >> 8) % (uint64_t)s.len] = 'N';
}

extern "C" int seeqdevSynthReads(void *d_out, uint64_t first, uint64_t n, int len, const char *pattern_plain, int plen,
                                 int tau, uint64_t seed, void *hip_stream)
{
   if (!d_out || len <= 0 || plen < 0 || plen > 96) { seeqerr = 0; errno = EINVAL; return -1; }
   if (n == 0) return 0;
   SynthArgs s;
   memset(&s, 0, sizeof s);
   s.out = (uint8_t *)d_out; s.first = first; s.n = n; s.seed = seed; s.len = len; s.plen = plen; s.tau = tau;
   memcpy(s.pattern, pattern_plain, (size_t)plen);
   const uint64_t blocks = (n + WG - 1) / WG;
   if (blocks > 0x7FFFFFFFull) { seeqerr = 0; errno = E2BIG; return -1; }
   hipLaunchKernelGGL(k_synth, dim3((unsigned)blocks), dim3(WG), 0, (hipStream_t)hip_stream, s);
   HIP_TRY(hipGetLastError(), EIO);
   return 0;
}

/* ========================================================================== */
/* Pattern handle                                                             */
/* ========================================================================== */
static unsigned long g_pattern_ids = 0;    /* one per seeqdevPatternNew: cache key (a freed pattern's address can come back) */

struct seeqdev_pattern {
   unsigned long id;
   int       wlen, tau, words;
   int       device;
   uint32_t *d_peq;          /* [2][5][words] */
   char     *keys;           /* host copy of the key bytes (DFA construction) */
   uint32_t *h_peq;          /* host copy of d_peq */
   int       sdfa_state;     /* the streaming automaton of k_stream (seeq_dfa.h): 0 not tried, 1 built, -1 none fits */
   uint16_t *d_sdfa;         /* transition table in HBM, staged into LDS by k_stream */
   uint16_t *d_sdfa_skip;    /* its skip variant (SQ_IGNORE: column 4 maps every state onto itself) */
   uint16_t *d_sdfa_restart; /* a filter's restart variant (long lines: acceptance goes on from the root, every part occurrence is flagged) */
   uint32_t  sdfa_rows, sdfa_final_base;
   int       sdfa_parts;     /* 1: the complete automaton (exact verdicts); > 1: partition filter (candidates) */
   int       sdfa_warm;      /* bytes of warm-up a chunk walk needs */
   double    sdfa_pacc;      /* filter: probability that a random DNA character completes a candidate */
   int       pair_state;     /* the pair automaton of k_pair (seeq_dfa.h section 3): 0 not tried, 1 built, -1 none fits */
   uint16_t *d_pair;         /* its table in HBM (32-byte rows), staged into LDS by k_pair */
   uint32_t  pair_units;     /* 16-byte units of the table (2 per row) */
   uint32_t  pair_states;
   int       pair_parts, pair_mp, pair_warm;
   double    pair_pacc;      /* probability that a random DNA character completes a candidate */
   int       quad_state;     /* the quad automaton of the packed walk (seeq_dfa.h section 3b): 0 not tried, 1 built, -1 none fits */
   uint16_t *d_quad;         /* its table in HBM (512-byte rows) */
   uint32_t  quad_units;     /* 16-byte units of it */
   uint32_t  quad_states;
   int       quad_parts, quad_mp;
   double    quad_pacc;
   pthread_mutex_t plan_lock;   /* the automata are built on first use; scan contexts on several threads may share a pattern */
};

extern "C" seeqdev_pattern_t *seeqdevPatternNew(const char *keys, int wlen, int tau)
{
   seeqerr = 0;
   if (!keys || wlen < 1 || tau < 0 || tau >= wlen) { errno = EINVAL; return NULL; }
   if (wlen > SEEQDEV_MAX_WLEN) {
      snprintf(g_last_error, sizeof g_last_error, "pattern longer than %d positions", SEEQDEV_MAX_WLEN);
      errno = E2BIG;
      return NULL;
   }
   if (seeqdevDeviceCount() < 1) {
      snprintf(g_last_error, sizeof g_last_error, "no HIP device: seeq-mi355x has no CPU matcher");
      errno = ENODEV;
      return NULL;
   }
   seeqdev_pattern *p = (seeqdev_pattern *)calloc(1, sizeof *p);
   if (!p) return NULL;
   p->id = __atomic_add_fetch(&g_pattern_ids, 1ul, __ATOMIC_RELAXED);
   pthread_mutex_init(&p->plan_lock, NULL);
   p->wlen = wlen; p->tau = tau; p->words = seeq_words_for(wlen);
   p->keys = (char *)malloc((size_t)wlen);
   if (p->keys) memcpy(p->keys, keys, (size_t)wlen);
   const size_t nw = (size_t)10 * p->words;
   uint32_t *h = (uint32_t *)malloc(nw * sizeof(uint32_t));
   char *rkeys = (char *)malloc((size_t)wlen);
   if (!h || !rkeys || !p->keys) { free(h); free(rkeys); free(p->keys); free(p); errno = ENOMEM; return NULL; }
   for (int i = 0; i < wlen; i++) rkeys[i] = keys[wlen - 1 - i];     /* reference libseeq.c:89 */
   seeq_build_peq(keys, wlen, p->words, h);
   seeq_build_peq(rkeys, wlen, p->words, h + 5 * p->words);
   hipError_t e = hipGetDevice(&p->device);
   if (e == hipSuccess) e = hipMalloc((void **)&p->d_peq, nw * sizeof(uint32_t));
   if (e == hipSuccess) e = hipMemcpy(p->d_peq, h, nw * sizeof(uint32_t), hipMemcpyHostToDevice);
   free(rkeys);
   p->h_peq = h;
   if (e != hipSuccess) {
      hip_fail(e, "seeqdevPatternNew", e == hipErrorOutOfMemory ? ENOMEM : EIO);
      if (p->d_peq) (void)hipFree(p->d_peq);
      free(p->keys);
      free(h);
      free(p);
      return NULL;
   }
   return p;
}

extern "C" int seeqdevPatternDevice(const seeqdev_pattern_t *p) { return p ? p->device : -1; }

extern "C" void seeqdevPatternFree(seeqdev_pattern_t *p)
{
   if (!p) return;
   (void)use_device(p->device);
   if (p->d_peq) (void)hipFree(p->d_peq);
   if (p->d_sdfa) (void)hipFree(p->d_sdfa);
   if (p->d_sdfa_skip) (void)hipFree(p->d_sdfa_skip);
   if (p->d_sdfa_restart) (void)hipFree(p->d_sdfa_restart);
   if (p->d_pair) (void)hipFree(p->d_pair);
   if (p->d_quad) (void)hipFree(p->d_quad);
   pthread_mutex_destroy(&p->plan_lock);
   free(p->keys);
   free(p->h_peq);
   free(p);
}

/* The automata of a pattern are built on first use, once, under the pattern's lock (scan contexts on several threads
 * may share a pattern); a table is published -- state 1 -- only when all of it sits in HBM, and a failed upload frees
 * what it had allocated. */
static void pattern_plan_stream(seeqdev_pattern *mp, bool complete_only)
{
   pthread_mutex_lock(&mp->plan_lock);
   if (mp->sdfa_state == 0 && mp->keys) {
      int state = -1;
      seeq_dfa_t *d = seeq_dfa_plan_stream(mp->keys, mp->wlen, mp->tau, complete_only ? 1 : 0);
      if (d) {
         const size_t bytes = (size_t)d->nrows * 16;
         uint16_t *skip = seeq_dfa_skip_variant(d);
         uint16_t *rst = d->nparts > 1 ? seeq_dfa_restart_variant(d) : nullptr;
         uint16_t *t0 = nullptr, *t1 = nullptr, *t2 = nullptr;
         if (skip && (rst || d->nparts <= 1) && hipMalloc((void **)&t0, bytes) == hipSuccess && hipMemcpy(t0, d->table, bytes, hipMemcpyHostToDevice) == hipSuccess &&
             hipMalloc((void **)&t1, bytes) == hipSuccess && hipMemcpy(t1, skip, bytes, hipMemcpyHostToDevice) == hipSuccess &&
             (!rst || (hipMalloc((void **)&t2, bytes) == hipSuccess && hipMemcpy(t2, rst, bytes, hipMemcpyHostToDevice) == hipSuccess))) {
            mp->d_sdfa = t0; mp->d_sdfa_skip = t1; mp->d_sdfa_restart = t2;
            mp->sdfa_rows = d->nrows;
            mp->sdfa_final_base = d->acc_final;            /* state value of ACC_NEW */
            mp->sdfa_parts = d->nparts;
            mp->sdfa_warm = d->warm;
            mp->sdfa_pacc = d->p_accept;
            state = 1;
         } else {
            if (t0) (void)hipFree(t0);
            if (t1) (void)hipFree(t1);
            if (t2) (void)hipFree(t2);
         }
         free(skip);
         free(rst);
         seeq_dfa_free(d);
      }
      __atomic_store_n(&mp->sdfa_state, state, __ATOMIC_RELEASE);
   }
   pthread_mutex_unlock(&mp->plan_lock);
}

static void pattern_plan_pair(seeqdev_pattern *mp)
{
   pthread_mutex_lock(&mp->plan_lock);
   if (mp->pair_state == 0 && mp->keys) {
      int state = -1;
      seeq_pair_t *d = seeq_pair_plan(mp->keys, mp->wlen, mp->tau);
      if (d) {
         const size_t bytes = d->table_bytes;
         uint16_t *t0 = nullptr;
         if (hipMalloc((void **)&t0, bytes) == hipSuccess && hipMemcpy(t0, d->table, bytes, hipMemcpyHostToDevice) == hipSuccess) {
            mp->d_pair = t0;
            mp->pair_units = d->table_bytes / 16;
            mp->pair_states = d->nstates;
            mp->pair_parts = d->nparts; mp->pair_mp = d->mp; mp->pair_warm = d->warm;
            mp->pair_pacc = d->p_accept;
            state = 1;
         } else if (t0) {
            (void)hipFree(t0);
         }
         seeq_pair_free(d);
      }
      __atomic_store_n(&mp->pair_state, state, __ATOMIC_RELEASE);
   }
   pthread_mutex_unlock(&mp->plan_lock);
}

static void pattern_plan_quad(seeqdev_pattern *mp)
{
   pthread_mutex_lock(&mp->plan_lock);
   if (mp->quad_state == 0 && mp->keys) {
      int state = -1;
      seeq_quad_t *d = seeq_quad_plan(mp->keys, mp->wlen, mp->tau);
      if (d) {
         uint16_t *t0 = nullptr;
         if (hipMalloc((void **)&t0, d->table_bytes) == hipSuccess && hipMemcpy(t0, d->table, d->table_bytes, hipMemcpyHostToDevice) == hipSuccess) {
            mp->d_quad = t0;
            mp->quad_units = d->table_bytes / 16;
            mp->quad_states = d->nstates;
            mp->quad_parts = d->nparts; mp->quad_mp = d->mp;
            mp->quad_pacc = d->p_accept;
            state = 1;
         } else if (t0) {
            (void)hipFree(t0);
         }
         seeq_quad_free(d);
      }
      __atomic_store_n(&mp->quad_state, state, __ATOMIC_RELEASE);
   }
   pthread_mutex_unlock(&mp->plan_lock);
}

/* ---- one-pass multi-pattern scans: the automata of a pattern SET (seeq_dfa.h section 4, seeq_multi.h) ---- */
struct MultiPlan {
   int            npat;
   unsigned long  ids[SEEQ_MULTI_MAX];      /* generation ids of the patterns it was built for */
   int            state;                    /* 1 usable, -1 no union automaton for this set: a scan per pattern */
   seeqdev_pattern upat;                    /* the union as a pattern: what run_segments walks (wlen = maxspan, tau = 0: skip_back = maxspan) */
   uint16_t      *d_res_next;
   uint32_t      *d_res_mask;
   uint32_t       res_states;
   int            maxspan;
   int            m[SEEQ_MULTI_MAX], tau[SEEQ_MULTI_MAX], fw[SEEQ_MULTI_MAX];
   uint32_t      *d_eq;                     /* [npat][768 * 2] the patterns' EQ tables of the exact pass */
   int            eq_options;               /* the option bits d_eq was made for (-1: none yet) */
   uint32_t       pair_states, raw_states;
   int            lp_min, exact;
};

static void multi_plan_free(MultiPlan *mp)
{
   if (!mp) return;
   if (mp->d_res_next) (void)hipFree(mp->d_res_next);
   if (mp->d_res_mask) (void)hipFree(mp->d_res_mask);
   if (mp->d_eq) (void)hipFree(mp->d_eq);
   if (mp->upat.d_pair) (void)hipFree(mp->upat.d_pair);
   if (mp->upat.d_peq) (void)hipFree(mp->upat.d_peq);
   free(mp->upat.h_peq);
   free(mp);
}

/* The plan for this set (built once per set and context; a context keeps the plan of its last set). */
static MultiPlan *multi_plan_for(MultiPlan **slot, const seeqdev_pattern_t *const *pats, int npat)
{
   MultiPlan *mp = *slot;
   if (mp && mp->npat == npat) {
      bool same = true;
      for (int k = 0; k < npat; k++) same = same && mp->ids[k] == pats[k]->id;
      if (same) return mp;
   }
   multi_plan_free(mp);
   *slot = mp = (MultiPlan *)calloc(1, sizeof *mp);
   if (!mp) return NULL;
   mp->npat = npat;
   mp->state = -1;
   mp->eq_options = -1;
   for (int k = 0; k < npat; k++) mp->ids[k] = pats[k]->id;
   if (npat < 2 || npat > SEEQ_MULTI_MAX) return mp;
   const char *keys[SEEQ_MULTI_MAX];
   for (int k = 0; k < npat; k++) {
      if (!pats[k]->keys || pats[k]->wlen > FUSED_MAX_WLEN2) return mp;
      keys[k] = pats[k]->keys; mp->m[k] = pats[k]->wlen; mp->tau[k] = pats[k]->tau;
      mp->fw[k] = pats[k]->wlen <= FUSED_MAX_WLEN ? 1 : 2;
   }
   seeq_multi_t *d = seeq_multi_build(keys, mp->m, mp->tau, npat);
   if (!d) return mp;
   if (d->maxspan <= FUSED_MAX_WLEN2) {
      seeqdev_pattern &u = mp->upat;
      u.id = __atomic_add_fetch(&g_pattern_ids, 1ul, __ATOMIC_RELAXED);
      u.wlen = d->maxspan; u.tau = 0; u.words = seeq_words_for(d->maxspan);
      u.device = pats[0]->device;
      u.sdfa_state = -1;
      u.h_peq = (uint32_t *)calloc((size_t)10 * u.words, sizeof(uint32_t));
      bool ok = u.h_peq != NULL;
      ok = ok && hipMalloc((void **)&u.d_peq, (size_t)10 * u.words * sizeof(uint32_t)) == hipSuccess;
      ok = ok && hipMemset(u.d_peq, 0, (size_t)10 * u.words * sizeof(uint32_t)) == hipSuccess;
      ok = ok && hipMalloc((void **)&u.d_pair, d->pair->table_bytes) == hipSuccess;
      ok = ok && hipMemcpy(u.d_pair, d->pair->table, d->pair->table_bytes, hipMemcpyHostToDevice) == hipSuccess;
      ok = ok && hipMalloc((void **)&mp->d_res_next, (size_t)d->res_states * 16) == hipSuccess;
      ok = ok && hipMemcpy(mp->d_res_next, d->res_next, (size_t)d->res_states * 16, hipMemcpyHostToDevice) == hipSuccess;
      ok = ok && hipMalloc((void **)&mp->d_res_mask, (size_t)d->res_states * 4) == hipSuccess;
      ok = ok && hipMemcpy(mp->d_res_mask, d->res_mask, (size_t)d->res_states * 4, hipMemcpyHostToDevice) == hipSuccess;
      ok = ok && hipMalloc((void **)&mp->d_eq, (size_t)npat * 1536 * sizeof(uint32_t)) == hipSuccess;
      if (ok) {
         u.pair_state = 1;
         u.pair_units = d->pair->table_bytes / 16;
         u.pair_states = d->pair->nstates;
         u.pair_parts = npat; u.pair_mp = d->pair->mp; u.pair_warm = d->pair->warm;
         u.pair_pacc = d->pair->p_accept;
         mp->res_states = d->res_states;
         mp->maxspan = d->maxspan;
         mp->pair_states = d->pair->nstates; mp->raw_states = d->pair->nstates_raw;
         mp->exact = d->res_exact;
         mp->lp_min = d->lp[0];
         for (int k = 1; k < npat; k++) if (d->lp[k] < mp->lp_min) mp->lp_min = d->lp[k];
         mp->state = 1;
      }
   }
   seeq_multi_free(d);
   return mp;
}

/* ========================================================================== */
/* Scan context                                                               */
/* ========================================================================== */
#include "seeq_plan.h"          /* ScanKnobs, seeq_plan_scan: which kernels serve a scan (pure host code) */

struct OccMemo { const void *fn; size_t lds; int per_cu; };

/* Packed runs (seeq_packed.h), reads per segment: 64 Mi (16 Mi until the quad walk: the per-segment launches behind the walk -- the scan of the
   candidate counts, the list, k_nh_top, k_emit1 -- are latency-bound and cost as much for a quarter of the reads: 100 M reads, 2.28 ms per step
   in six segments, 2.11 in three, 2.07 in two; 8 bytes of workspace per read of a segment); SEEQ_PACKED_SEG_READS sets another size (tests) */
static constexpr size_t PACKED_SEG_READS_DEFAULT = (size_t)1 << 26;

struct seeqdev_scan {
   hipStream_t stream;
   bool        own_stream;
   int         device;            /* the HIP device this context (its stream, its workspace) lives on */
   hipEvent_t  ev_h2d[2];         /* profiling: around the H2D copy of seeqdevScanHostBegin */
   bool        have_h2d_ev;
   float       h2d_ms;            /* ... of the last fetched scan */
   float      *launch_ms; size_t cap_launch_ms;      /* per forward-scan launch of the last fetched scan (profiling) */
   unsigned long long *clk_probe;  /* page-locked, 4 words per segment: the scan kernel's own clock readings (profiling; k_pair) */
   size_t      cap_clk_probe;
   float       clk_mhz;           /* core clock the last run's scan launches ran at (mean over the launches; 0: not measured) */
   bool        clk_valid;         /* the last run's scan kernel filled clk_probe (k_pair under profiling): else the readings are an earlier run's */
   int         ncu;               /* compute units of the device (cached) */
   size_t      lds_per_wg;        /* LDS a workgroup may allocate on it */
   size_t      lds_per_cu;        /* LDS of a compute unit (what a workgroup gets when it asks for it: hipFuncAttributeMaxDynamicSharedMemorySize) */
   ScanKnobs   knobs;
   OccMemo     occ[8]; int nocc;  /* hipOccupancyMaxActiveBlocksPerMultiprocessor results */
   bool        last_filter;       /* the last run walked a partition filter automaton */
   bool        last_packed_quad;  /* the last packed run walked the quad table */
   size_t      pk_seg_reads;      /* reads per segment of a packed run */
   /* workspace (device) */
   uint32_t *line_start;  size_t cap_lines;
   uint32_t *tile_cnt;    size_t cap_tiles;
   uint64_t *hitmask, *hdrmask; uint32_t *wave_off, *hdr_off; size_t cap_chunks;
   uint4    *ent;                /* [cap_hitlines] k_order -> k_bounds2 (seeq_order.h) */
   uint32_t *hit_start, *hit_line, *nh, *hit_col, *nh_sum; size_t cap_hitlines;      /* nh_sum: per 256 entries (k_verify) */
   /* one-pass kernels: what the scan kernel of a segment writes and its post-pass reads */
   struct OnePassWs {
      uint32_t *tile_cl, *tile_hits, *tile_dirty; uint64_t *tile_dmask;   /* [cap_ftiles] */
      uint4    *tmp;                 /* [cap_hitlines] hit slices, then the COUNT -> EMIT cache */
      uint32_t *wg_hits;             /* [cap_slices] */
      uint32_t *wg_part;             /* [4 * cap_slices] */
      uint32_t *wg_lastnl;           /* [cap_slices] k_stream: last newline seen by each wave */
   } ow;
   size_t cap_ftiles, cap_slices;
   uint32_t *d_eqtab, *h_eqtab;   /* [256]; h_ is pinned */
   unsigned long eq_pat_id; int eq_options;   /* what d_eqtab holds (pattern generation id, option bits) */
   double avg_line;               /* average bytes per line incl. newline (hint or sampled) */
   double line_hint;              /* caller's hint; 0 = sample the buffer */
   uint8_t *h_sample;             /* pinned, SAMPLE_BYTES */
   const void *avg_text; size_t avg_nbytes;
   int force_path;                /* 0 auto, 1 generic, 2 fused (SEEQ_PATH env / tests) */
   int last_path;                 /* 1 generic, 2 fused: what the last run used */
   seeqdev_hit_t *records; uint64_t *rec_off; size_t cap_records;
   uint32_t *scan_ws;           size_t cap_scan_ws;
   uint32_t *lead_fidx, *lead_flag, *lead_wend; unsigned long long *lead_key; size_t cap_lead;      /* long lines, leaders (seeq_stream.h): per hit-list entry */
   Counters *d_cnt;
   Counters *h_cnt;            /* pinned */
   /* seeqdevStringMatch: one string per call in ONE launch (pinned, device-visible) */
   uint8_t *h_str; size_t cap_str;            /* the string (strings below STRING_ZC_MAX are read by the kernel over the link) */
   uint32_t *h_strout; size_t cap_strout;     /* {nhits, ticket, pad[2]} + records; fine-grained (coherent) page-locked memory */
   uint32_t  str_seq;                         /* ticket of the last k_string launch */
   /* staging for seeqdevScanHost */
   uint8_t *d_text; size_t cap_text;
   /* seeqdevScanRunMulti: per pattern of the last multi scan its counts and (host copy) its records */
   seeqdev_counts_t *multi_cnt; size_t *multi_first; int multi_n, cap_multi_n;
   seeqdev_hit_t *multi_rec; size_t cap_multi_rec, multi_nrec;
   /* one-pass multi-pattern scans (seeq_multi.h): the plan of the last pattern set, per-pattern workspace */
   struct MultiPlan *mplan;                               /* owned */
   bool      multi_active;                                /* run_segments: stop after the candidate list, hand over to multi_post */
   int       multi_rc;                                    /* multi_post's verdict inside run_segments */
   uint32_t *ml_mask, *ml_first, *ml_last; size_t cap_ml;            /* per candidate line */
   uint32_t *mp_idx, *mp_nh; size_t cap_mp;               /* npat regions of cap_mp / npat entries: index into the per-line arrays, hits */
   uint32_t *m_bsum; size_t cap_m_bsum;
   Counters *d_mcnt, *h_mcnt;                             /* [SEEQ_MULTI_MAX], h_ pinned */
   MultiExact *d_mx, *h_mx; size_t mx_slots, mx_next;      /* per segment the patterns' exact-pass arguments: device copy, pinned ring of mx_slots segments */
   uint32_t *m_scan_ws; size_t cap_m_scan_ws;             /* block sums of the per-pattern scans */
   int       last_multi;                                  /* the last multi scan: 1 = one walk for all patterns, 0 = a scan per pattern */
   /* packed read batches (seeqdevScanPacked) */
   uint32_t *pk_cand, *pk_slot, *pk_coff; uint64_t *pk_bmask; size_t cap_pk_reads;      /* candidate columns per read of a segment; per block of 64 reads: candidates before it, their mask */
   uint8_t  *pk_stage; size_t cap_pk_stage;               /* ASCII lines of the candidate reads */
   uint8_t  *d_unpack; size_t cap_unpack;                 /* the whole batch as ASCII text: patterns the packed walk does not serve */
   uint32_t *pk_last; size_t cap_pk_last;                 /* per candidate: column of its last candidate */
   bool      is_packed; seeqdev_packed_t packed;          /* the last run was a packed one (re-run on overflow) */
   /* last run (for the transparent re-run on overflow) */
   const seeqdev_pattern *pat; const void *text; size_t nbytes; int options, want;
   bool ran;
   seeqdev_counts_t counts;
   /* profiling */
   bool prof;
   hipEvent_t *ev;             /* 4 events per segment: index start, forward start, forward end, segment end */
   size_t nev_seg;             /* segments that have events */
   size_t prof_segs;           /* segments of the last run */
   float acc_ms[4];
   float fwd_ms_avg;           /* mean k_forward launch duration of the last run */
   size_t seg_bytes;           /* segment size */
   bool user_reserved;         /* caller sized the per-line workspace: trust it */
   bool no_stream;             /* k_stream met a line it cannot address (starts > 1 GiB before its segment): use the per-line kernels */
   bool force_ll;              /* a read-length looking buffer had hits inside very long lines: use k_stream's long-line variant */
   bool no_stream_nd;          /* SQ_CONVERT / SQ_IGNORE: the text has non-DNA bytes, k_stream (exact for clean text only) is off */
   bool no_leaders;            /* long lines with many hits: a leader's fresh start lay inside the walk before it -- every line stays with one lane */
   bool no_window;             /* k_pair's candidates: a line had candidates on both sides of a segment seam -- whole lines are scanned */
   int  fallback_ttl;          /* scans left before the three fall-back flags above are dropped and the fast path is tried again (one text with a
                                  long line or foreign bytes must not slow a long-lived context down for good) */
   bool sample_dirty;          /* the sampled prefix holds more than one byte outside the alphabet per 4 KB: k_pair stays out */
   unsigned sample_age;        /* runs since the line-length sample was taken (a reused buffer may hold other text by now) */
};

/* A workspace array of a new size: the new block first, then the old one goes -- a refused allocation (ENOMEM) leaves the array, and with it the
   capacity its group was sized for, as it was: the context stays usable (tests/test_gpu_parity.py::test_a_refused_reserve_leaves_the_context_usable). */
static int ws_alloc(void **p, size_t bytes)
{
   void *g = NULL;
   hipError_t e = hipMalloc(&g, bytes ? bytes : 16);
   if (e != hipSuccess) { (void)hipGetLastError(); return hip_fail(e, "hipMalloc(workspace)", ENOMEM); }
   if (*p) (void)hipFree(*p);
   *p = g;
   return 0;
}

extern "C" seeqdev_scan_t *seeqdevScanNew(void *hip_stream)
{
   seeqerr = 0;
   if (seeqdevDeviceCount() < 1) {
      snprintf(g_last_error, sizeof g_last_error, "no HIP device: seeq-mi355x has no CPU matcher");
      errno = ENODEV;
      return NULL;
   }
   seeqdev_scan *s = (seeqdev_scan *)calloc(1, sizeof *s);
   if (!s) return NULL;
   if (hipGetDevice(&s->device) != hipSuccess) s->device = 0;
   s->seg_bytes = (size_t)0xF0000000u;      /* 3.75 GiB segments: u32 offsets with room for k_stream's bias; multiple of every tile size */
   const char *env = getenv("SEEQ_SEGMENT_BYTES");
   if (env && atoll(env) >= 65536) s->seg_bytes = ((size_t)atoll(env) + 15) & ~(size_t)15;
   if (s->seg_bytes > 0xFFFF0000ull) s->seg_bytes = 0xFFFF0000ull;
   hipError_t e = hipSuccess;
   if (hip_stream) s->stream = (hipStream_t)hip_stream;
   else {
      /* a private stream of the default ("blocking") kind: work a caller queued on the legacy null stream -- torch's
         default stream has handle 0, which arrives here as NULL -- is ordered before ours and ours before theirs */
      e = hipStreamCreateWithFlags(&s->stream, hipStreamDefault);
      s->own_stream = true;
   }
   if (e == hipSuccess) e = hipMalloc((void **)&s->d_cnt, sizeof(Counters));
   if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_cnt, sizeof(Counters), hipHostMallocDefault);
   if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_eqtab, 2048 * sizeof(uint32_t), hipHostMallocDefault);
   if (e == hipSuccess) e = hipMalloc((void **)&s->d_eqtab, 2048 * sizeof(uint32_t));
   if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_sample, SAMPLE_BYTES, hipHostMallocDefault);
   {
      const char *pe = getenv("SEEQ_PATH");
      s->force_path = pe ? (!strcmp(pe, "generic") ? 1 : !strcmp(pe, "fused") ? 2 : 0) : 0;
      ScanKnobs &kn = s->knobs;
      const char *v;
      v = getenv("SEEQ_FUSED_KERNEL"); kn.kernel = v ? (!strcmp(v, "stream") ? 1 : !strcmp(v, "direct") ? 2 : !strcmp(v, "pair") ? 3 : 0) : 0;
      v = getenv("SEEQ_NO_LEADERS");   kn.no_leaders = v && atoi(v) == 1;
      v = getenv("SEEQ_TILE_BYTES");   kn.tile_bytes = v ? atoi(v) : 0;
      v = getenv("SEEQ_NO_FILTER");    kn.no_filter = v && atoi(v) == 1;
      v = getenv("SEEQ_STREAM_SUB");   kn.no_sub = v && atoi(v) == 0;
      v = getenv("SEEQ_STREAM_WU");    kn.min_wu = v ? atoi(v) : 0;
      v = getenv("SEEQ_NO_MYERS");     kn.no_myers = v && atoi(v) == 1;
      v = getenv("SEEQ_NO_WINDOW");    kn.no_window = v && atoi(v) == 1;
      v = getenv("SEEQ_EXPLAIN");      kn.explain = v && atoi(v) == 1;
      v = getenv("SEEQ_PACKED_QUAD");  kn.no_packed_quad = v && atoi(v) == 0;
      v = getenv("SEEQ_PACKED_SEG_READS");
      s->pk_seg_reads = v && atol(v) >= 64 && (size_t)atol(v) <= ((size_t)1 << 26) ? (size_t)atol(v) & ~(size_t)63 : PACKED_SEG_READS_DEFAULT;
      s->ncu = 256;
      s->lds_per_wg = 65536;
      s->lds_per_cu = 65536;
      int dev = 0;
      hipDeviceProp_t prop;
      if (e == hipSuccess && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      { s->ncu = prop.multiProcessorCount; s->lds_per_wg = prop.sharedMemPerBlock; s->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor; }
   }
   if (e != hipSuccess) {
      hip_fail(e, "seeqdevScanNew", EIO);
      seeqdevScanFree(s);
      return NULL;
   }
   return s;
}

extern "C" void seeqdevScanFree(seeqdev_scan_t *s)
{
   if (!s) return;
   (void)use_device(s->device);
   if (s->have_h2d_ev) { (void)hipEventDestroy(s->ev_h2d[0]); (void)hipEventDestroy(s->ev_h2d[1]); }
   (void)hipStreamSynchronize(s->stream);
   {
      void *ob[] = {s->ow.tile_cl, s->ow.tile_hits, s->ow.tile_dirty, s->ow.tile_dmask, s->ow.tmp, s->ow.wg_hits, s->ow.wg_part, s->ow.wg_lastnl};
      for (void *b : ob) if (b) (void)hipFree(b);
   }
   { void *pk[] = {s->pk_cand, s->pk_slot, s->pk_coff, s->pk_bmask, s->pk_stage, s->pk_last, s->d_unpack}; for (void *b : pk) if (b) (void)hipFree(b); }
   { void *mw[] = {s->ml_mask, s->ml_first, s->ml_last, s->mp_idx, s->mp_nh, s->m_bsum, s->d_mcnt, s->d_mx, s->m_scan_ws}; for (void *b : mw) if (b) (void)hipFree(b); }
   if (s->h_mcnt) (void)hipHostFree(s->h_mcnt);
   if (s->h_mx) (void)hipHostFree(s->h_mx);
   multi_plan_free(s->mplan);
   void *bufs[] = {s->rec_off, s->line_start, s->tile_cnt, s->hitmask, s->hdrmask, s->wave_off, s->hdr_off, s->hit_start,
                   s->hit_line, s->d_eqtab,
                   s->nh, s->hit_col, s->nh_sum, s->ent, s->records, s->scan_ws, s->lead_fidx, s->lead_flag, s->lead_wend, s->lead_key, s->d_cnt, s->d_text};
   for (void *b : bufs) if (b) (void)hipFree(b);
   if (s->h_cnt) (void)hipHostFree(s->h_cnt);
   if (s->h_eqtab) (void)hipHostFree(s->h_eqtab);
   if (s->h_sample) (void)hipHostFree(s->h_sample);
   if (s->h_str) (void)hipHostFree(s->h_str);
   if (s->h_strout) (void)hipHostFree(s->h_strout);
   for (size_t i = 0; i < 4 * s->nev_seg; i++) (void)hipEventDestroy(s->ev[i]);
   free(s->ev);
   free(s->launch_ms);
   if (s->clk_probe) (void)hipHostFree(s->clk_probe);
   free(s->multi_cnt); free(s->multi_first);
   if (s->multi_rec) (void)hipHostFree(s->multi_rec);
   if (s->own_stream && s->stream) (void)hipStreamDestroy(s->stream);
   free(s);
}

static int reserve_impl(seeqdev_scan *s, size_t max_bytes, size_t max_lines, size_t max_hitlines, size_t max_records)
{
   seeqerr = 0;
   if (!s) { errno = EINVAL; return -1; }
   /* Everything per-line is per SEGMENT; only records span the whole buffer. */
   if (max_bytes) {
      const size_t seg = max_bytes < s->seg_bytes ? max_bytes : s->seg_bytes;
      const size_t tiles = (seg + TILE - 1) / TILE + 1;
      if (tiles > s->cap_tiles) {
         if (ws_alloc((void **)&s->tile_cnt, tiles * sizeof(uint32_t))) return -1;
         s->cap_tiles = tiles;
      }
      const size_t ftiles = seg / FUSED_MIN_TILE + 2;
      const size_t slices = MAX_FUSED_GRID;               /* hit slices: one per wave of a persistent grid */
      if (ftiles > s->cap_ftiles) {
         if (ws_alloc((void **)&s->ow.tile_cl, ftiles * sizeof(uint32_t))) return -1;
         if (ws_alloc((void **)&s->ow.tile_dirty, ftiles * sizeof(uint32_t))) return -1;
         if (ws_alloc((void **)&s->ow.tile_dmask, ftiles * sizeof(uint64_t))) return -1;
         if (ws_alloc((void **)&s->ow.tile_hits, ftiles * sizeof(uint32_t))) return -1;
         s->cap_ftiles = ftiles;
      }
      if (slices > s->cap_slices) {
         if (ws_alloc((void **)&s->ow.wg_hits, slices * sizeof(uint32_t))) return -1;
         if (ws_alloc((void **)&s->ow.wg_part, 4 * slices * sizeof(uint32_t))) return -1;
         if (ws_alloc((void **)&s->ow.wg_lastnl, slices * sizeof(uint32_t))) return -1;
         s->cap_slices = slices;
      }
   }
   if (max_lines > s->cap_lines) {
      const size_t chunks = (max_lines + 63) / 64 + 1;
      if (ws_alloc((void **)&s->line_start, (max_lines + 1) * sizeof(uint32_t))) return -1;
      if (ws_alloc((void **)&s->hitmask, chunks * sizeof(uint64_t))) return -1;
      if (ws_alloc((void **)&s->hdrmask, chunks * sizeof(uint64_t))) return -1;
      if (ws_alloc((void **)&s->wave_off, chunks * sizeof(uint32_t))) return -1;
      if (ws_alloc((void **)&s->hdr_off, chunks * sizeof(uint32_t))) return -1;
      s->cap_lines = max_lines;
      s->cap_chunks = chunks;
   }
   if (max_hitlines > s->cap_hitlines) {
      if (ws_alloc((void **)&s->hit_start, max_hitlines * sizeof(uint32_t))) return -1;
      if (ws_alloc((void **)&s->hit_line, max_hitlines * sizeof(uint32_t))) return -1;
      if (ws_alloc((void **)&s->ow.tmp, max_hitlines * sizeof(uint4))) return -1;
      if (ws_alloc((void **)&s->nh, max_hitlines * sizeof(uint32_t))) return -1;
      if (ws_alloc((void **)&s->hit_col, max_hitlines * sizeof(uint32_t))) return -1;
      if (ws_alloc((void **)&s->ent, max_hitlines * sizeof(uint4))) return -1;
      if (ws_alloc((void **)&s->nh_sum, 2 * (max_hitlines / 256 + 2) * sizeof(uint32_t))) return -1;      /* + the chunks' entries with a hit */
      s->cap_hitlines = max_hitlines;
   }
   if (max_records > s->cap_records) {
      if (ws_alloc((void **)&s->records, max_records * sizeof(seeqdev_hit_t))) return -1;
      if (ws_alloc((void **)&s->rec_off, max_records * sizeof(uint64_t))) return -1;
      s->cap_records = max_records;
   }
   /* block sums for the two-level scans: the largest scanned array */
   size_t largest = s->cap_tiles;
   if (s->cap_chunks > largest) largest = s->cap_chunks;
   if (s->cap_hitlines > largest) largest = s->cap_hitlines;
   if (s->cap_ftiles > largest) largest = s->cap_ftiles;
   size_t nb = largest / SCAN_BLOCK + 2;
   if (3 * (s->cap_ftiles / SCAN_BLOCK + 2) > nb) nb = 3 * (s->cap_ftiles / SCAN_BLOCK + 2);   /* launch_scanset: three tile arrays at once */
   if (nb > s->cap_scan_ws) {
      if (ws_alloc((void **)&s->scan_ws, nb * sizeof(uint32_t))) return -1;
      s->cap_scan_ws = nb;
   }
   return 0;
}

extern "C" int seeqdevScanReserve(seeqdev_scan_t *s, size_t max_bytes, size_t max_lines, size_t max_hitlines,
                                  size_t max_records)
{
   if (reserve_impl(s, max_bytes, max_lines, max_hitlines, max_records)) return -1;
   if (max_lines) s->user_reserved = true;
   return 0;
}

extern "C" int seeqdevScanSetLineHint(seeqdev_scan_t *s, double avg_bytes_per_line)
{
   if (!s || avg_bytes_per_line < 0) { errno = EINVAL; return -1; }
   s->line_hint = avg_bytes_per_line;
   return 0;
}

extern "C" int seeqdevScanLastPath(const seeqdev_scan_t *s) { return s ? s->last_path : 0; }
extern "C" int seeqdevScanLastFilter(const seeqdev_scan_t *s) { return s && s->last_filter ? 1 : 0; }
extern "C" int seeqdevScanLastPackedQuad(const seeqdev_scan_t *s) { return s && s->last_packed_quad ? 1 : 0; }

extern "C" int seeqdevScanSetProfiling(seeqdev_scan_t *s, int on)
{
   if (!s) { errno = EINVAL; return -1; }
   s->prof = on != 0;
   return 0;
}

extern "C" int seeqdevScanLastTimes(const seeqdev_scan_t *s, float ms[4])
{
   if (!s || !ms) { errno = EINVAL; return -1; }
   for (int i = 0; i < 4; i++) ms[i] = s->acc_ms[i];
   return 0;
}

/* Up to three u32 arrays of the same host-known length scanned in place by ONE set of three launches (the per-tile
 * arrays of the one-pass kernels): blockIdx.y selects the array, bsum has one region of `nb` partial sums per array. */
struct ScanSet { uint32_t *arr[3]; uint32_t *total[3]; };

__global__ __launch_bounds__(WG) void k_scanset_reduce(ScanSet ss, uint32_t *bsum, uint32_t n, uint32_t nb)
{
   __shared__ uint32_t s_wave[4];
   const uint32_t *in = ss.arr[blockIdx.y];
   const uint32_t base = blockIdx.x * SCAN_BLOCK;
   uint32_t v = 0;
#pragma unroll
   for (int k = 0; k < SCAN_ITEMS; k++) {
      const uint32_t i = base + threadIdx.x * SCAN_ITEMS + k;
      if (i < n) v += in[i];
   }
   uint32_t tot;
   block_excl_scan(v, &tot, s_wave);
   if (threadIdx.x == 0) bsum[blockIdx.y * nb + blockIdx.x] = tot;
}

__global__ __launch_bounds__(WG) void k_scanset_top(ScanSet ss, uint32_t *bsum, uint32_t nb)
{
   __shared__ uint32_t s_wave[4];
   uint32_t *b = bsum + blockIdx.x * nb;
   uint32_t running = 0;
   for (uint32_t b0 = 0; b0 < nb; b0 += WG) {
      const uint32_t i = b0 + threadIdx.x;
      const uint32_t v = i < nb ? b[i] : 0;
      uint32_t tot;
      const uint32_t ex = block_excl_scan(v, &tot, s_wave);
      if (i < nb) b[i] = running + ex;
      running += tot;
      __syncthreads();
   }
   if (threadIdx.x == 0 && ss.total[blockIdx.x]) *ss.total[blockIdx.x] = running;
}

__global__ __launch_bounds__(WG) void k_scanset_apply(ScanSet ss, const uint32_t *bsum, uint32_t n, uint32_t nb)
{
   __shared__ uint32_t s_wave[4];
   uint32_t *arr = ss.arr[blockIdx.y];
   const uint32_t base = blockIdx.x * SCAN_BLOCK;
   uint32_t item[SCAN_ITEMS];
   uint32_t v = 0;
#pragma unroll
   for (int k = 0; k < SCAN_ITEMS; k++) {
      const uint32_t i = base + threadIdx.x * SCAN_ITEMS + k;
      item[k] = i < n ? arr[i] : 0;
      v += item[k];
   }
   uint32_t tot;
   uint32_t ex = block_excl_scan(v, &tot, s_wave) + bsum[blockIdx.y * nb + blockIdx.x];
#pragma unroll
   for (int k = 0; k < SCAN_ITEMS; k++) {
      const uint32_t i = base + threadIdx.x * SCAN_ITEMS + k;
      if (i < n) arr[i] = ex;
      ex += item[k];
   }
}

/* ---- launch helpers ------------------------------------------------------- */
template <int XF>
static void launch_scan(seeqdev_scan *s, hipStream_t st, const void *in, uint32_t *out, size_t cap_items, const uint32_t *n_ptr,
                        uint32_t add, uint32_t shift, uint32_t *total_out)
{
   const unsigned nb = (unsigned)((cap_items + SCAN_BLOCK - 1) / SCAN_BLOCK);
   if (nb == 0) return;
   hipLaunchKernelGGL(k_scan_reduce<XF>, dim3(nb), dim3(WG), 0, st, in, s->scan_ws, n_ptr, add, shift);
   hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(WG), 0, st, s->scan_ws, n_ptr, add, shift, total_out);
   hipLaunchKernelGGL(k_scan_apply<XF>, dim3(nb), dim3(WG), 0, st, in, out, (const uint32_t *)s->scan_ws,
                      n_ptr, add, shift);
}

static void launch_scanset(seeqdev_scan *s, hipStream_t st, uint32_t *a0, uint32_t *a1, uint32_t *a2, uint32_t n, uint32_t *t0, uint32_t *t1, uint32_t *t2)
{
   const unsigned nb = (unsigned)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
   if (nb == 0) return;
   ScanSet ss = {{a0, a1, a2}, {t0, t1, t2}};
   const unsigned na = a2 ? 3 : a1 ? 2 : 1;
   hipLaunchKernelGGL(k_scanset_reduce, dim3(nb, na), dim3(WG), 0, st, ss, s->scan_ws, n, nb);
   hipLaunchKernelGGL(k_scanset_top, dim3(na), dim3(WG), 0, st, ss, s->scan_ws, nb);
   hipLaunchKernelGGL(k_scanset_apply, dim3(nb, na), dim3(WG), 0, st, ss, (const uint32_t *)s->scan_ws, n, nb);
}

/* The tile_cnt scan has a host-known length (ntiles); a dedicated small kernel
 * avoids routing a host constant through device memory. */
__global__ __launch_bounds__(WG) void k_scan_tiles(uint32_t *tile_cnt, uint32_t ntiles, uint32_t *total_out)
{
   __shared__ uint32_t s_wave[4];
   uint32_t running = 0;
   for (uint32_t b0 = 0; b0 < ntiles; b0 += WG * SCAN_ITEMS) {
      uint32_t item[SCAN_ITEMS];
      uint32_t v = 0;
#pragma unroll
      for (int k = 0; k < SCAN_ITEMS; k++) {
         const uint32_t i = b0 + threadIdx.x * SCAN_ITEMS + k;
         item[k] = i < ntiles ? tile_cnt[i] : 0;
         v += item[k];
      }
      uint32_t tot;
      uint32_t ex = running + block_excl_scan(v, &tot, s_wave);
#pragma unroll
      for (int k = 0; k < SCAN_ITEMS; k++) {
         const uint32_t i = b0 + threadIdx.x * SCAN_ITEMS + k;
         if (i < ntiles) tile_cnt[i] = ex;
         ex += item[k];
      }
      running += tot;
   }
   if (threadIdx.x == 0) *total_out = running;
}

/* Workgroups of `fn` (threads per workgroup, dynamic LDS) that fit one CU; asked once per kernel and LDS size.
 * Also raises the kernel's dynamic-LDS limit.  -1 (errno set) when HIP refuses. */
static int occupancy_of(seeqdev_scan *s, const void *fn, int threads, size_t lds)
{
   for (int i = 0; i < s->nocc; i++)
      if (s->occ[i].fn == fn && s->occ[i].lds == lds) return s->occ[i].per_cu;
   if (lds) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), EIO);
   int per_cu = 0;
   if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds) != hipSuccess || per_cu < 1) per_cu = 1;
   OccMemo &m = s->occ[s->nocc < 8 ? s->nocc++ : 7];
   m.fn = fn; m.lds = lds; m.per_cu = per_cu;
   return per_cu;
}

static int multi_post(seeqdev_scan *s, const ScanArgs &ua, hipStream_t st);

/* what the planner (seeq_plan.h) needs to know of the pattern's automata, and how it asks for one that has not been tried yet */
static void pattern_automata(const seeqdev_pattern *p, PlanAutomata *au)
{
   au->sdfa_state = __atomic_load_n(&p->sdfa_state, __ATOMIC_ACQUIRE);
   au->sdfa_parts = p->sdfa_parts; au->sdfa_warm = p->sdfa_warm; au->sdfa_pacc = p->sdfa_pacc;
   au->pair_state = __atomic_load_n(&p->pair_state, __ATOMIC_ACQUIRE);
   au->pair_warm = p->pair_warm; au->pair_pacc = p->pair_pacc;
}

static void plan_ensure(void *ctx, int which, int complete_only, PlanAutomata *au)
{
   seeqdev_pattern *mp = (seeqdev_pattern *)ctx;
   if (which == 0) pattern_plan_stream(mp, complete_only != 0);
   else pattern_plan_pair(mp);
   pattern_automata(mp, au);
}

/* ---- a run of seeqdevScanRun, in pieces (round 5: run_segments was one function of 385 lines; the decisions had moved to seeq_plan.h in round 4,
        what is left executes the plan):  run_setup -- the plan, the kernel instance and its grid, the EQ tables, the profiling events;
        seg_onepass -- a segment's one-pass scan kernel (k_pair / k_stream / k_direct) and the ordering of its hit slices;
        seg_index_forward<W> -- the generic path's newline index and k_forward<W>;  seg_post<W> -- the exact pass and the records ---- */
struct SegRun {
   const seeqdev_pattern *pat;
   int       options, want;
   ScanPlan  plan;
   int       fw, nw;                  /* column words of the one-pass kernels; waves per workgroup of the scan kernel */
   uint32_t  tile_bytes;
   unsigned  fused_grid, nslices, grid_lines;
   const void *stream_fn;             /* the k_stream / k_pair instance of this run */
   size_t    dfa_lds, seg_bytes, nseg;
};

/* EQ[dir][byte][fw]: the top-aligned Peq column of the byte's class, or a flag (reference seeqcore.h:89-111 folded with the non-DNA option,
   libseeq.c:223-228,265-270) -- uploaded when the pattern or the options changed since the context's last scan */
static int eq_tables_upload(seeqdev_scan *s, const seeqdev_pattern *pat, int options, int fw)
{
   if (s->eq_pat_id == pat->id && s->eq_options == options) return 0;
   const int Wp = pat->words;
   for (int dir = 0; dir < 2; dir++)
      for (int b = 0; b < 256; b++) {
         const uint8_t cls = sq_class_of((uint32_t)b, options);
         uint64_t v;
         if (cls < 5) {
            const uint32_t *q = pat->h_peq + (dir * 5 + cls) * Wp;
            const uint64_t col = (uint64_t)q[0] | (Wp > 1 ? (uint64_t)q[1] << 32 : 0);
            v = col << (32 * fw - pat->wlen);                /* row m lands on the top bit */
         } else {
            v = cls == SQC_TERM ? FUSED_FLAG_TERM : FUSED_FLAG_SKIP;
         }
         uint32_t *dst = s->h_eqtab + (size_t)(dir * 256 + b) * fw;
         dst[0] = (uint32_t)v;
         if (fw == 2) dst[1] = (uint32_t)(v >> 32);
      }
   /* third table, k_stream's Myers mode: the forward table with the newline marked (flag bits 0-1 = 3) */
   memcpy(s->h_eqtab + (size_t)512 * fw, s->h_eqtab, (size_t)256 * fw * sizeof(uint32_t));
   s->h_eqtab[(size_t)512 * fw + (size_t)'\n' * fw] |= 3u;
   /* the pinned staging table may still be read by an earlier copy on this stream: wait before the next rewrite */
   HIP_TRY(hipMemcpyAsync(s->d_eqtab, s->h_eqtab, (size_t)768 * fw * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream), EIO);
   HIP_TRY(hipStreamSynchronize(s->stream), EIO);
   s->eq_pat_id = pat->id;
   s->eq_options = options;
   return 0;
}

static int run_setup(seeqdev_scan *s, SegRun &r)
{
   const seeqdev_pattern *pat = s->pat;
   const int options = s->options, want = s->want;
   const bool fasta = (options & SEEQDEV_FASTA) != 0;
   const bool single = (options & SEEQDEV_SINGLELINE) != 0;
   const size_t nbytes = s->nbytes;


   const int ncu = s->ncu;
   const ScanKnobs &kn = s->knobs;

   const size_t line_blocks = (s->cap_lines + WG - 1) / WG;
   unsigned grid_lines = (unsigned)(line_blocks < (size_t)ncu * 16 ? line_blocks : (size_t)ncu * 16);
   if (grid_lines == 0) grid_lines = 1;

   /* ---- the plan (seeq_plan.h: a pure function of the pattern, the options, the text's line length and this context's fall-back
           flags); the rest of this function executes it ---- */
   PlanIn pin;
   memset(&pin, 0, sizeof pin);
   pin.wlen = pat->wlen; pin.tau = pat->tau; pin.options = options; pin.want = want;
   pin.avg_line = s->avg_line; pin.line_hint = s->line_hint; pin.force_path = s->force_path;
   pin.no_stream = s->no_stream; pin.force_ll = s->force_ll; pin.no_stream_nd = s->no_stream_nd; pin.no_window = s->no_window;
   pin.no_leaders = s->no_leaders; pin.sample_dirty = s->sample_dirty; pin.multi_active = s->multi_active;
   pin.seg_bytes = s->seg_bytes; pin.kn = &kn;
   PlanAutomata au;
   pattern_automata(pat, &au);
   r.plan = seeq_plan_scan(pin, au, plan_ensure, const_cast<seeqdev_pattern *>(pat));
   const ScanPlan &plan = r.plan;
   if (kn.explain) seeq_plan_print(stderr, pin, au, plan);
   if (plan.rc) return plan.rc;
   const int fw = plan.fw;
   const bool use_stream = plan.use_stream, use_pair = plan.use_pair, use_myers = plan.use_myers, filter = plan.filter, use_fused = plan.use_fused;
   const int stream_ch = 128;                 /* bytes per lane of k_stream / k_pair */
   const int stream_wu = plan.stream_wu;
   uint32_t tile_bytes = 0;
   unsigned fused_grid = 1;
   int nw = 4;
   unsigned nslices = 1;                      /* hit slices: one per wave */
   const void *stream_fn = nullptr;
   const bool stream_ll = plan.stream_ll;
   const int stream_sub = plan.stream_sub;    /* 0, 1: SQ_CONVERT ('N' for non-DNA bytes), 2: SQ_IGNORE (skip bytes) */
   size_t dfa_lds = 0;
   if (use_fused) {
      if (use_stream) {
         nw = STREAM_NW;
         tile_bytes = 64u * (uint32_t)stream_ch;
         /* the k_stream instance of this scan: <warm-up dwords, FASTA, long lines, SUB> */
#define SEEQ_STREAM_FN(...) (stream_wu == 4 ? (const void *)k_stream<4, __VA_ARGS__> : stream_wu == 6 ? (const void *)k_stream<6, __VA_ARGS__> \
                                                                                                      : (const void *)k_stream<8, __VA_ARGS__>)
         stream_fn = stream_sub == 2 ? SEEQ_STREAM_FN(false, false, 2)
                   : stream_sub ? (stream_ll ? SEEQ_STREAM_FN(false, true, 1) : SEEQ_STREAM_FN(false, false, 1))
                   : stream_ll ? (fasta ? SEEQ_STREAM_FN(true, true) : SEEQ_STREAM_FN(false, true))
                   : fasta ? SEEQ_STREAM_FN(true, false) : SEEQ_STREAM_FN(false, false);
#undef SEEQ_STREAM_FN
         dfa_lds = ((size_t)pat->sdfa_rows * 16 + 15) & ~(size_t)15;
         if (use_myers) {
#define SEEQ_MYERS_FN(FA, MY) (stream_wu == 16 ? (const void *)k_stream<16, FA, true, 0, MY> : (const void *)k_stream<32, FA, true, 0, MY>)
            stream_fn = fw == 1 ? (fasta ? SEEQ_MYERS_FN(true, 1) : SEEQ_MYERS_FN(false, 1)) : (fasta ? SEEQ_MYERS_FN(true, 2) : SEEQ_MYERS_FN(false, 2));
#undef SEEQ_MYERS_FN
            dfa_lds = (size_t)256 * fw * sizeof(uint32_t);
         }
         if (use_pair) {
#define SEEQ_PAIR_FN(...) (stream_wu == 4 ? (const void *)k_pair<4, __VA_ARGS__> : stream_wu == 5 ? (const void *)k_pair<5, __VA_ARGS__> : stream_wu == 6 ? (const void *)k_pair<6, __VA_ARGS__> \
                          : stream_wu == 7 ? (const void *)k_pair<7, __VA_ARGS__> : (const void *)k_pair<8, __VA_ARGS__>)
            stream_fn = fasta ? SEEQ_PAIR_FN(true) : plan.ig ? SEEQ_PAIR_FN(false, true) : plan.pair_ll ? SEEQ_PAIR_FN(false, false, true) : SEEQ_PAIR_FN(false);
            dfa_lds = (size_t)pat->pair_units * 16;
#undef SEEQ_PAIR_FN
         }
         int per_cu = occupancy_of(s, stream_fn, 64 * nw, dfa_lds);
         if (per_cu < 0) return -1;
         fused_grid = (unsigned)(ncu * per_cu);
         if ((size_t)fused_grid * nw > MAX_FUSED_GRID) fused_grid = (unsigned)(MAX_FUSED_GRID / nw);
         nslices = fused_grid * nw;                         /* one hit slice per wave */
      } else {
         nw = 4;
         double want = s->avg_line * 63.5;                /* <= 64 lines per region: one per lane */
         if (want < 512) want = 512;
         if (want > DIRECT_MAXRR * 1024) want = DIRECT_MAXRR * 1024;      /* (k_direct reads a region in at most DIRECT_MAXRR rounds of 1 KiB: lines that average more than 258 bytes
                                                                              once made regions of 16 KiB + 48 bytes, whose last 48 bytes no round looked at -- profiles/ignore_fuzz.py) */
         tile_bytes = ((uint32_t)want) & ~15u;
         if (kn.tile_bytes >= 512 && kn.tile_bytes <= DIRECT_MAXRR * 1024) tile_bytes = (uint32_t)kn.tile_bytes & ~15u;
         int per_cu = occupancy_of(s, fw == 1 ? (const void *)k_direct<4, 1> : (const void *)k_direct<4, 2>, 256, 0);
         if (per_cu < 0) return -1;
         fused_grid = (unsigned)(ncu * per_cu);
         if ((size_t)fused_grid * nw > MAX_FUSED_GRID) fused_grid = (unsigned)(MAX_FUSED_GRID / nw);
         nslices = fused_grid * nw;                         /* one hit slice per wave */
      }
      if (eq_tables_upload(s, pat, options, fw)) return -1;
   }
   const bool use_direct = plan.use_direct;
   s->last_path = plan.path;
   s->last_filter = filter;
   const bool superset = plan.superset;                  /* the scan kernel's hit lines are candidates: nh[] decides */
   const bool need_nh = plan.need_nh, nh_is_count = plan.nh_is_count;

   const size_t seg_bytes = single ? (nbytes ? nbytes : 1) : s->seg_bytes;
   if (single && nbytes > 0xFFFF0000ull) { seeqerr = 0; errno = E2BIG; return -1; }
   const size_t nseg = nbytes ? (nbytes + seg_bytes - 1) / seg_bytes : 0;
   s->prof_segs = 0;
   if (s->prof && nseg > s->nev_seg) {
      hipEvent_t *g = (hipEvent_t *)realloc(s->ev, 4 * nseg * sizeof(hipEvent_t));
      if (!g) { seeqerr = 0; errno = ENOMEM; return -1; }
      s->ev = g;
      for (size_t i = 4 * s->nev_seg; i < 4 * nseg; i++) HIP_TRY(hipEventCreate(&s->ev[i]), EIO);
      s->nev_seg = nseg;
   }
   if (s->prof) s->prof_segs = nseg;
   if (s->prof && nseg > s->cap_clk_probe) {
      if (s->clk_probe) (void)hipHostFree(s->clk_probe);
      s->clk_probe = nullptr; s->cap_clk_probe = 0;
      if (hipHostMalloc((void **)&s->clk_probe, nseg * 4 * sizeof(unsigned long long), hipHostMallocDefault) == hipSuccess) s->cap_clk_probe = nseg;
   }
   if (s->prof && s->clk_probe) memset(s->clk_probe, 0, nseg * 4 * sizeof(unsigned long long));
   s->clk_valid = s->prof && s->clk_probe && use_pair && use_fused;
   /* (Tried: the post-pass of segment k on a second stream under k_pair of segment k + 1, k_pair on one workgroup per CU --
      it is as fast there.  The post-pass kernels do run beside it, and take 3 to 14 times as long as alone: they are
      made of scattered loads and the memory system is what k_pair saturates.  Net: +2 % .. -3 % per step.  Not kept;
      tag r03-experiment-overlap-postpass, profiles/r03/overlap_trace.txt.) */
   r.pat = pat; r.options = options; r.want = want;
   r.fw = fw; r.nw = nw; r.tile_bytes = tile_bytes; r.fused_grid = fused_grid; r.nslices = nslices; r.grid_lines = grid_lines;
   r.stream_fn = stream_fn; r.dfa_lds = dfa_lds; r.seg_bytes = seg_bytes; r.nseg = nseg;
   (void)use_direct; (void)superset; (void)need_nh; (void)nh_is_count; (void)use_myers;
   return 0;
}

/* a segment's one-pass scan kernel + the ordering of its hit slices: newline handling, forward scan and per-tile compaction in ONE kernel */
static int seg_onepass(seeqdev_scan *s, const SegRun &r, ScanArgs &a, size_t sg, hipEvent_t *ev, bool &order2, uint32_t &stream_ntiles)
{
   const seeqdev_pattern *pat = r.pat;
   const int options = r.options, want = r.want, match_opt = r.options & 3, fw = r.fw, ncu = s->ncu;
   const ScanPlan &plan = r.plan;
   const bool fasta = (options & SEEQDEV_FASTA) != 0, single = (options & SEEQDEV_SINGLELINE) != 0;
   const bool use_stream = plan.use_stream, use_pair = plan.use_pair, use_myers = plan.use_myers, filter = plan.filter, use_fused = plan.use_fused, use_direct = plan.use_direct;
   const bool stream_ll = plan.stream_ll, superset = plan.superset, need_nh = plan.need_nh, nh_is_count = plan.nh_is_count;
   const int stream_sub = plan.stream_sub, stream_ch = 128, nw = r.nw;
   const uint32_t tile_bytes = r.tile_bytes;
   const unsigned fused_grid = r.fused_grid, nslices = r.nslices, grid_lines = r.grid_lines;
   const void *stream_fn = r.stream_fn;
   const size_t dfa_lds = r.dfa_lds, nbytes = s->nbytes;
   Counters *c = s->d_cnt;
   seeqdev_scan::OnePassWs &ow = s->ow;
   hipStream_t st = s->stream;
   (void)pat; (void)want; (void)match_opt; (void)fw; (void)ncu; (void)fasta; (void)single; (void)use_stream; (void)use_pair; (void)use_myers; (void)filter; (void)use_fused;
   (void)use_direct; (void)stream_ll; (void)superset; (void)need_nh; (void)nh_is_count; (void)stream_sub; (void)stream_ch; (void)nw; (void)tile_bytes; (void)fused_grid;
   (void)nslices; (void)grid_lines; (void)stream_fn; (void)dfa_lds; (void)nbytes; (void)c; (void)ow; (void)st;

   /* ---- fused path: newline index + forward scan + per-tile compaction in ONE kernel ---- */
   FusedArgs f;
   memset(&f, 0, sizeof f);
   f.text = a.text; f.nbytes = nbytes; f.seg_base = a.seg_base; f.seg_len = a.seg_len; f.first_seg = a.first_seg;
   f.tile_bytes = tile_bytes;
   f.ntiles = (uint32_t)(((uint64_t)a.seg_len + tile_bytes - 1) / tile_bytes);
   stream_ntiles = f.ntiles;
   f.eqtab = s->d_eqtab; f.peq = pat->d_peq;
   f.m = pat->wlen; f.tau = pat->tau; f.options = options; f.want = want;
   f.tile_cl = ow.tile_cl; f.tile_hits = ow.tile_hits; f.tmp = ow.tmp; f.cap_tmp = (uint32_t)s->cap_hitlines;
   f.wg_hits = ow.wg_hits; f.wg_part = ow.wg_part; f.wg_lastnl = stream_ll ? ow.wg_lastnl : nullptr;   /* only the window walk (long lines) needs it */
   f.tile_dirty = f.wg_lastnl ? ow.tile_dirty : nullptr;
   f.tile_dmask = f.wg_lastnl ? ow.tile_dmask : nullptr;
   f.cnt = c;
   f.clk_probe = (s->prof && s->clk_probe && use_pair) ? s->clk_probe + 4 * sg : nullptr;
   uint32_t pos_bias = 0;
   if (use_stream) {
      f.dfa = stream_sub == 2 ? pat->d_sdfa_skip : plan.ll_restart ? pat->d_sdfa_restart : pat->d_sdfa; f.dfa_rows = pat->sdfa_rows; f.dfa_final_base = pat->sdfa_final_base;
      f.ll_filter = plan.ll_restart ? 2u : plan.ll_filter ? 1u : 0u;      /* (2: the restart table -- a chain that accepted inside its warm-up window names its first byte) */
      f.skip_thr = plan.skip_thr;
      if (use_pair) { f.dfa = pat->d_pair; f.dfa_rows = pat->pair_units; f.dfa_final_base = 0; f.pair = 1; f.ig_thr = plan.ig ? (uint32_t)(pat->wlen - pat->tau) : 0u; }
      if (use_myers) { f.dfa = (const uint16_t *)(s->d_eqtab + (size_t)512 * fw); f.dfa_rows = (uint32_t)(64 * fw); f.dfa_final_base = 0; f.pair = 2; }
      /* A hit line can start before the segment: hit offsets of this segment are relative to seg_base - pos_bias */
      uint64_t room = 0xFFFFFFF0ull - a.seg_len;
      if (room > ((uint64_t)1 << 30)) room = (uint64_t)1 << 30;
      pos_bias = (uint32_t)(a.seg_base < room ? a.seg_base : room) & ~127u;     /* chunk boundaries stay multiples of the chunk */
      f.pos_bias = pos_bias;
   }
   if (ev) { HIP_TRY(hipEventRecord(ev[0], st), EIO); HIP_TRY(hipEventRecord(ev[1], st), EIO); }
   unsigned fgrid = fused_grid;                     /* persistent: workgroups without a tile just publish zeros */
   const unsigned nsl = nslices;
   f.slice_cap = f.cap_tmp / nsl;
   if (use_stream) {
      void *kargs[] = {&f};
      HIP_TRY(hipLaunchKernel(stream_fn, dim3(fgrid), dim3(64 * (unsigned)nw), kargs, dfa_lds, st), EIO);
   }
   else if (use_direct && fw == 2) hipLaunchKernelGGL((k_direct<4, 2>), dim3(fgrid), dim3(256), 0, st, f);
   else hipLaunchKernelGGL((k_direct<4, 1>), dim3(fgrid), dim3(256), 0, st, f);
   if (ev) HIP_TRY(hipEventRecord(ev[2], st), EIO);
   /* read-length lines behind k_pair / k_stream: the three launches of seeq_order.h; else (long lines, k_direct) the seven of before */
   const uint32_t order_nb = (f.ntiles + SEEQ_ORDER_BLOCK - 1) / SEEQ_ORDER_BLOCK;
   order2 = plan.order2 && order_nb <= SEEQ_ORDER_MAX_BLOCKS && 2 * (size_t)order_nb <= s->cap_scan_ws;
   if (plan.ig) {
      /* SQ_IGNORE behind k_pair: the line markers travel in the ordered entries (seeq_order.h) -- the older ordering kernels know nothing of them */
      if (!order2) { snprintf(g_last_error, sizeof g_last_error, "SQ_IGNORE on k_pair needs the three-launch ordering (segment too large for its block sums)"); seeqerr = 0; errno = EIO; return -1; }
      a.ig_thr = f.ig_thr; a.ig_ent = (const uint4 *)s->ent;
      {  /* the base the pattern's plain positions hold most often (ties: T first -- the rarest byte of FASTQ quality lines): an occurrence keeps all but tau copies */
         static const char base_of_key[9] = {0, 'A', 'C', 0, 'G', 0, 0, 0, 'T'};
         int cnt[4] = {0, 0, 0, 0}, best = 3;
         for (int i = 0; i < pat->wlen; i++) { const int kb = pat->keys[i] & 0x1F; if (kb == 1) cnt[0]++; else if (kb == 2) cnt[1]++; else if (kb == 4) cnt[2]++; else if (kb == 8) cnt[3]++; }
         for (int b = 2; b >= 0; b--) if (cnt[b] > cnt[best]) best = b;
         const int need = cnt[best] - pat->tau;
         a.ig_need = need > 0 ? (uint32_t)need : 0u;
         a.ig_bval = (uint32_t)(unsigned char)base_of_key[1 << best];
         a.ig_bmask = best == 3 ? 0xDEu : 0xDFu;           /* T: U and either case too */
      }
   }
   if (order2) {
      const unsigned rgrid = nsl / 4 + 1 < 2048 ? nsl / 4 + 1 : 2048;       /* one wave per slice, strided */
      seeq_launch_tiles_post(st, f, (uint32_t)nsl, s->scan_ws, order_nb);
      seeq_launch_order(rgrid, st, f, (uint32_t)nsl, (const uint32_t *)s->scan_ws, order_nb, s->ent);
   }
   else hipLaunchKernelGGL(k_fused_post, dim3(1), dim3(256), 0, st, f, (uint32_t)nsl);
   if (!order2 && (want != SEEQDEV_WANT_COUNTLINES || superset)) {
      launch_scanset(s, st, f.tile_hits, f.tile_cl, f.tile_dirty, f.ntiles, nullptr, nullptr, f.tile_dirty ? &c->seg_dirty_tiles : nullptr);
      const unsigned rgrid = nsl / 4 + 1 < 2048 ? nsl / 4 + 1 : 2048;       /* one wave per slice, strided */
      if (use_stream) hipLaunchKernelGGL(k_stream_reorder, dim3(rgrid), dim3(256), 0, st, f, (uint32_t)nsl, s->hit_start, s->hit_line, s->nh, s->hit_col);
      else hipLaunchKernelGGL(k_fused_reorder, dim3(rgrid), dim3(256), 0, st, f, (uint32_t)nsl, s->hit_start, s->hit_line);
   }
   a.seg_base -= pos_bias;                           /* the exact pass addresses lines through hit_start */
   a.pos_bias = pos_bias;
   a.tile_dirty = f.tile_dirty; a.tile_dmask = f.tile_dmask; a.stream_ntiles = f.ntiles; a.stream_tile_bytes = tile_bytes;
   /* the exact pass walks candidate windows instead of whole lines where lines are long (sampled average);
      read-length lines are scanned whole -- the bookkeeping of the walk costs more than it saves there */
   a.stream_ch = stream_ll ? (uint32_t)stream_ch : 0u;
   a.walk_ext = plan.walk_ext; a.ll_restart = plan.ll_restart ? 1u : 0u;
   return 0;
}

/* the generic path's segment: newline index (K0), k_forward<W> (K1), ranks of the hit lines and of the FASTA headers (K2) */
template <int W>
static int seg_index_forward(seeqdev_scan *s, const SegRun &r, ScanArgs &a, hipEvent_t *ev)
{
   const seeqdev_pattern *pat = r.pat;
   const int options = r.options, want = r.want, match_opt = r.options & 3, fw = r.fw, ncu = s->ncu;
   const ScanPlan &plan = r.plan;
   const bool fasta = (options & SEEQDEV_FASTA) != 0, single = (options & SEEQDEV_SINGLELINE) != 0;
   const bool use_stream = plan.use_stream, use_pair = plan.use_pair, use_myers = plan.use_myers, filter = plan.filter, use_fused = plan.use_fused, use_direct = plan.use_direct;
   const bool stream_ll = plan.stream_ll, superset = plan.superset, need_nh = plan.need_nh, nh_is_count = plan.nh_is_count;
   const int stream_sub = plan.stream_sub, stream_ch = 128, nw = r.nw;
   const uint32_t tile_bytes = r.tile_bytes;
   const unsigned fused_grid = r.fused_grid, nslices = r.nslices, grid_lines = r.grid_lines;
   const void *stream_fn = r.stream_fn;
   const size_t dfa_lds = r.dfa_lds, nbytes = s->nbytes;
   Counters *c = s->d_cnt;
   seeqdev_scan::OnePassWs &ow = s->ow;
   hipStream_t st = s->stream;
   (void)pat; (void)want; (void)match_opt; (void)fw; (void)ncu; (void)fasta; (void)single; (void)use_stream; (void)use_pair; (void)use_myers; (void)filter; (void)use_fused;
   (void)use_direct; (void)stream_ll; (void)superset; (void)need_nh; (void)nh_is_count; (void)stream_sub; (void)stream_ch; (void)nw; (void)tile_bytes; (void)fused_grid;
   (void)nslices; (void)grid_lines; (void)stream_fn; (void)dfa_lds; (void)nbytes; (void)c; (void)ow; (void)st;

   /* ---- K0: newline index ---- */
   if (ev) HIP_TRY(hipEventRecord(ev[0], st), EIO);
   if (single) {
      hipLaunchKernelGGL(k_single_line, dim3(1), dim3(1), 0, st, a);
   } else {
      hipLaunchKernelGGL(k_nl_count, dim3(a.ntiles), dim3(WG), 0, st, a);
      hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(WG), 0, st, a.tile_cnt, a.ntiles, &c->seg_nlines);
      hipLaunchKernelGGL(k_index_finalize, dim3(1), dim3(1), 0, st, a);
      hipLaunchKernelGGL(k_nl_write, dim3(a.ntiles), dim3(WG), 0, st, a);
   }
   /* ---- K1: forward scan ---- */
   if (ev) HIP_TRY(hipEventRecord(ev[1], st), EIO);
   hipLaunchKernelGGL(k_forward<W>, dim3(grid_lines), dim3(WG), 0, st, a);
   if (ev) HIP_TRY(hipEventRecord(ev[2], st), EIO);
   /* ---- K2: ranks of hit lines (and FASTA headers) ---- */
   launch_scan<1>(s, st, a.hitmask, a.wave_off, s->cap_chunks, &c->seg_nlines, 63u, 6u, &c->seg_nhitlines);
   if (fasta) launch_scan<1>(s, st, a.hdrmask, a.hdr_off, s->cap_chunks, &c->seg_nlines, 63u, 6u, &c->seg_nheaders);
   return 0;
}

/* behind a segment's scan: hit list -> (multi-pattern hand-over | leaders) -> the exact pass (K4) -> the records (K5).  Returns 1 when the segment
   is finished here (several patterns: multi_post ended it), 0 to go on, < 0 on failure */
template <int W>
static int seg_post(seeqdev_scan *s, const SegRun &r, ScanArgs &a, hipEvent_t *ev, bool order2, uint32_t stream_ntiles)
{
   const seeqdev_pattern *pat = r.pat;
   const int options = r.options, want = r.want, match_opt = r.options & 3, fw = r.fw, ncu = s->ncu;
   const ScanPlan &plan = r.plan;
   const bool fasta = (options & SEEQDEV_FASTA) != 0, single = (options & SEEQDEV_SINGLELINE) != 0;
   const bool use_stream = plan.use_stream, use_pair = plan.use_pair, use_myers = plan.use_myers, filter = plan.filter, use_fused = plan.use_fused, use_direct = plan.use_direct;
   const bool stream_ll = plan.stream_ll, superset = plan.superset, need_nh = plan.need_nh, nh_is_count = plan.nh_is_count;
   const int stream_sub = plan.stream_sub, stream_ch = 128, nw = r.nw;
   const uint32_t tile_bytes = r.tile_bytes;
   const unsigned fused_grid = r.fused_grid, nslices = r.nslices, grid_lines = r.grid_lines;
   const void *stream_fn = r.stream_fn;
   const size_t dfa_lds = r.dfa_lds, nbytes = s->nbytes;
   Counters *c = s->d_cnt;
   seeqdev_scan::OnePassWs &ow = s->ow;
   hipStream_t st = s->stream;
   (void)pat; (void)want; (void)match_opt; (void)fw; (void)ncu; (void)fasta; (void)single; (void)use_stream; (void)use_pair; (void)use_myers; (void)filter; (void)use_fused;
   (void)use_direct; (void)stream_ll; (void)superset; (void)need_nh; (void)nh_is_count; (void)stream_sub; (void)stream_ch; (void)nw; (void)tile_bytes; (void)fused_grid;
   (void)nslices; (void)grid_lines; (void)stream_fn; (void)dfa_lds; (void)nbytes; (void)c; (void)ow; (void)st;

   /* ---- K3: compaction ---- */
   if (!use_fused) hipLaunchKernelGGL(k_compact, dim3(grid_lines), dim3(WG), 0, st, a);
   if (!use_fused) hipLaunchKernelGGL(k_seg_mid, dim3(1), dim3(1), 0, st, a);   /* the fused paths: done by k_fused_post */
   const size_t hit_blocks = (s->cap_hitlines + WG - 1) / WG;
   unsigned grid_hits = (unsigned)(hit_blocks < (size_t)ncu * 16 ? hit_blocks : (size_t)ncu * 16);
   if (grid_hits == 0) grid_hits = 1;
   if (order2) seeq_launch_bounds2(grid_hits, st, a, s->ent, s->hit_col);
   else if (use_stream) hipLaunchKernelGGL(k_stream_bounds, dim3(grid_hits), dim3(256), 0, st, a, s->hit_col,
                                      (const uint32_t *)ow.tile_cl, stream_ntiles, tile_bytes);   /* hit position -> line start; repeats dropped */
   if (s->multi_active) {
      /* several patterns: the candidate list is the union's -- pattern sets per line, a list per pattern, the exact pass per pattern */
      { const int mr = multi_post(s, a, st); if (mr > 0) return -2; if (mr) return -1; }      /* (1: a launch was refused -- a scan per pattern) */
      hipLaunchKernelGGL(k_seg_end, dim3(1), dim3(1), 0, st, a, 3);
      if (ev) HIP_TRY(hipEventRecord(ev[3], st), EIO);
      HIP_TRY(hipGetLastError(), EIO);
      return 1;
   }
   /* long lines, every hit counted: candidates far behind the one before them get a lane of their own (seeq_stream.h, leaders) */
   const bool lead_best = plan.lead_best, leaders = plan.leaders;
   const uint32_t lead_wback = a.skip_back > 32u ? a.skip_back : 32u;
   if (leaders) {
      if (s->cap_hitlines > s->cap_lead) {
         if (ws_alloc((void **)&s->lead_fidx, s->cap_hitlines * sizeof(uint32_t))) return -1;
         if (ws_alloc((void **)&s->lead_flag, s->cap_hitlines * sizeof(uint32_t))) return -1;
         if (ws_alloc((void **)&s->lead_wend, s->cap_hitlines * sizeof(uint32_t))) return -1;
         if (ws_alloc((void **)&s->lead_key, s->cap_hitlines * sizeof(unsigned long long))) return -1;
         s->cap_lead = s->cap_hitlines;
      }
      const unsigned nbl = (unsigned)(s->cap_hitlines / LEAD_BLOCK + 1);
      hipLaunchKernelGGL(k_lead_reduce, dim3(nbl), dim3(256), 0, st, a, s->scan_ws);
      hipLaunchKernelGGL(k_lead_top, dim3(1), dim3(256), 0, st, a, s->scan_ws);
      hipLaunchKernelGGL(k_lead_apply, dim3(nbl), dim3(256), 0, st, a, (const uint32_t *)s->hit_col, (const uint32_t *)s->scan_ws, s->lead_fidx, s->lead_flag, ow.tmp, lead_wback, lead_best ? s->lead_key : (unsigned long long *)nullptr);
      a.walk_end = s->lead_wend;
      /* SQ_BEST: COUNT has to walk every group itself (and leave each group's best hit in the cache) instead of trusting the hit
         list and leaving the scan to EMIT, one lane per line */
      if (lead_best) a.filter = 1u;
      hipLaunchKernelGGL(k_lead_commit, dim3(grid_hits), dim3(256), 0, st, a, s->hit_col, (const uint4 *)ow.tmp);
   }
   const uint32_t *hcol = use_stream ? s->hit_col : nullptr;      /* first-hit columns: the exact pass may skip ahead */
   uint4 *ecache = (use_fused && need_nh && want == SEEQDEV_WANT_RECORDS) ? ow.tmp : nullptr;   /* COUNT -> EMIT */
   /* (Tried behind k_pair: a lane-queue kernel -- a wave owns 256 .. 512 hit-list entries staged in LDS and a lane that has
      finished its line takes the next entry at the next 64-byte block -- 22 % fewer instructions than k_exact1 COUNT, and
      slower, 296 against 257 us per segment: at 4 waves per SIMD the per-block loads of a lane are not hidden.  Not kept;
      tag r03-experiment-overlap-postpass holds it, profiles/r03/verify_ab.txt the numbers.) */
   /* ---- K4: hits per hit line ---- */
   /* behind the filters (every hit line is a candidate) on text where no byte is skipped: k_verify (seeq_verify.h) -- the lean
      two-phase exact pass with the scan of its counts inside; the EMIT pass behind it ends the segment */
   bool emitted = false;                                /* the records are out (k_emit1) */
   const bool verify = plan.verify;
   const int seg_flags = (need_nh ? 1 : 0) | (superset && !nh_is_count ? 2 : 0);
   if (verify) {
      const bool count_any = want != SEEQDEV_WANT_COUNTMATCH && !(want == SEEQDEV_WANT_RECORDS && match_opt == SQ_ALL);
      const int var = (want == SEEQDEV_WANT_RECORDS && match_opt == SQ_BEST) ? VERIFY_BEST : count_any ? VERIFY_ANY : VERIFY_ALL;
      a.nh_sum = s->nh_sum;
      a.nz_sum = superset && nh_is_count ? s->nh_sum + (s->cap_hitlines / 256 + 2) : nullptr;
      /* k_nh_top ends the segment unless k_exact1's EMIT pass follows (SQ_ALL records): k_emit1 works from what k_nh_top saved */
      a.fin = (want == SEEQDEV_WANT_RECORDS && var == VERIFY_ALL) ? 0u : 1u + (uint32_t)seg_flags;
      /* behind a partition filter every part of an occurrence reports: more than half of the entries are repeats of their line, and
         k_verify packs them away, 512 entries per workgroup (seeq_verify.h); behind a prefix automaton it does not pay */
      a.vrange = (use_pair ? pat->pair_parts > 1 : pat->sdfa_parts > 1) ? 512u : 0u;
      seeq_launch_verify(fw, var, grid_hits, st, a, (const uint32_t *)s->d_eqtab, hcol, ecache);
      if (want == SEEQDEV_WANT_RECORDS && var != VERIFY_ALL) seeq_launch_emit1(grid_hits, st, a, ecache);
      else if (want == SEEQDEV_WANT_RECORDS && ecache)      /* SQ_ALL: the first records from the cache, the others from the overflow lists */
         seeq_launch_emit_all(fw, grid_hits, grid_hits, st, a, (const uint32_t *)s->d_eqtab, hcol, ecache);
      emitted = want == SEEQDEV_WANT_RECORDS && (var != VERIFY_ALL || ecache);
   }
   else if (need_nh) {
      if (use_fused) {
         const uint32_t *eqp = (const uint32_t *)s->d_eqtab;
   #define SEEQ_COUNT1(WW, WK) hipLaunchKernelGGL((k_exact1<SQ_MODE_COUNT, WW, -1, WK>), dim3(grid_hits), dim3(WG), 0, st, a, eqp, hcol, ecache)
         if (fw == 2) { if (a.stream_ch) SEEQ_COUNT1(2, true); else SEEQ_COUNT1(2, false); }
         else { if (a.stream_ch) SEEQ_COUNT1(1, true); else SEEQ_COUNT1(1, false); }
   #undef SEEQ_COUNT1
      }
      else hipLaunchKernelGGL((k_exact<W, SQ_MODE_COUNT>), dim3(grid_hits), dim3(WG), 0, st, a);
      /* lines with >= 1 verified hit: with 0/1 verdicts that is the scan total (seg_nrec) -- no extra pass */
      if (leaders) {
         hipLaunchKernelGGL(k_lead_check, dim3(grid_hits), dim3(256), 0, st, a, (const uint32_t *)s->hit_col, (const uint32_t *)s->lead_flag, lead_wback);
         if (lead_best) {
            hipLaunchKernelGGL(k_lead_best, dim3(grid_hits), dim3(256), 0, st, a, (const uint32_t *)s->lead_fidx, s->lead_key, (const uint4 *)ecache, 0);
            hipLaunchKernelGGL(k_lead_best, dim3(grid_hits), dim3(256), 0, st, a, (const uint32_t *)s->lead_fidx, s->lead_key, (const uint4 *)ecache, 1);
         }
         else hipLaunchKernelGGL(k_lead_lines, dim3(grid_hits < 512 ? grid_hits : 512), dim3(256), 0, st, a, (const uint32_t *)s->lead_fidx, s->lead_flag);
      }
      else if (superset && nh_is_count) hipLaunchKernelGGL(k_count_nonzero, dim3(grid_hits < 512 ? grid_hits : 512), dim3(WG), 0, st, a);
      launch_scan<0>(s, st, a.nh, a.nh, s->cap_hitlines, &c->seg_nhitlines, 0u, 0u, &c->seg_nrec);
   }
   /* ---- K5: records ---- */
   if (want == SEEQDEV_WANT_RECORDS && !emitted) {
      if (!verify) hipLaunchKernelGGL(k_rec_check, dim3(1), dim3(1), 0, st, a);      /* (k_verify's last workgroup did) */
      if (use_fused) {
         const uint32_t *eqp = (const uint32_t *)s->d_eqtab;
         const int mo = (options & 3) == SQ_COUNT ? SQ_FIRST : (options & 3);
   #define SEEQ_EMIT1(WW, OO, WK) hipLaunchKernelGGL((k_exact1<SQ_MODE_EMIT, WW, OO, WK>), dim3(grid_hits), dim3(WG), 0, st, a, eqp, hcol, ecache)
         if (a.stream_ch) {
            if (fw == 2) { if (mo == SQ_BEST) SEEQ_EMIT1(2, SQ_BEST, true); else SEEQ_EMIT1(2, -1, true); }
            else { if (mo == SQ_BEST) SEEQ_EMIT1(1, SQ_BEST, true); else SEEQ_EMIT1(1, -1, true); }
         } else {
            if (fw == 2) { if (mo == SQ_BEST) SEEQ_EMIT1(2, SQ_BEST, false); else SEEQ_EMIT1(2, -1, false); }
            else { if (mo == SQ_BEST) SEEQ_EMIT1(1, SQ_BEST, false); else SEEQ_EMIT1(1, -1, false); }
         }
   #undef SEEQ_EMIT1
      }
      else {
         hipLaunchKernelGGL((k_exact<W, SQ_MODE_EMIT>), dim3(grid_hits), dim3(WG), 0, st, a);
         hipLaunchKernelGGL(k_rec_offsets, dim3(grid_hits), dim3(WG), 0, st, a);    /* k_exact1 writes them itself */
      }
   }
   return 0;
}

template <int W>
static int run_segments(seeqdev_scan *s)
{
   Counters *c = s->d_cnt;
   HIP_TRY(hipMemsetAsync(c, 0, sizeof(Counters), s->stream), EIO);
   SegRun r;
   memset(&r, 0, sizeof r);
   { const int rc = run_setup(s, r); if (rc) return rc; }
   const seeqdev_pattern *pat = r.pat;
   const ScanPlan &plan = r.plan;
   const int options = r.options, want = r.want;
   const size_t nbytes = s->nbytes, seg_bytes = r.seg_bytes, nseg = r.nseg;
   const bool use_stream = plan.use_stream, use_fused = plan.use_fused, filter = plan.filter, superset = plan.superset, need_nh = plan.need_nh, nh_is_count = plan.nh_is_count;
   for (size_t sg = 0; sg < nseg; sg++) {

      hipEvent_t *ev = s->prof ? s->ev + 4 * sg : NULL;
      uint32_t stream_ntiles = 0;
      bool order2 = false;                                /* the hit list is made by seeq_order.h's kernels */
      hipStream_t st = s->stream;
      ScanArgs a;
      memset(&a, 0, sizeof a);
      a.text = (const uint8_t *)s->text;
      a.nbytes = nbytes;
      a.seg_base = (uint64_t)sg * seg_bytes;
      a.seg_len = (uint32_t)((nbytes - a.seg_base) < seg_bytes ? (nbytes - a.seg_base) : seg_bytes);
      a.first_seg = sg == 0;
      a.peq = pat->d_peq;
      a.m = pat->wlen; a.tau = pat->tau; a.options = options; a.want = want;
      a.line_start = s->line_start; a.cap_lines = (uint32_t)s->cap_lines;
      a.tile_cnt = s->tile_cnt; a.ntiles = (a.seg_len + TILE - 1) / TILE;
      a.hitmask = s->hitmask; a.hdrmask = s->hdrmask; a.wave_off = s->wave_off; a.hdr_off = s->hdr_off;
      a.hit_start = s->hit_start; a.hit_line = s->hit_line; a.cap_hitlines = (uint32_t)s->cap_hitlines; a.nh = s->nh;
      a.records = s->records; a.cap_records = s->cap_records; a.rec_off = s->rec_off;
      a.use_nh = need_nh ? (use_stream ? 3u : 1u) : 0u;
      a.filter = filter ? 1u : 0u;
      a.skip_back = plan.skip_back;
      a.window_ok = plan.window_ok ? 1u : 0u;
      a.cnt = c;

      if (use_fused) { if (seg_onepass(s, r, a, sg, ev, order2, stream_ntiles)) return -1; }
      else if (seg_index_forward<W>(s, r, a, ev)) return -1;
      if (want != SEEQDEV_WANT_COUNTLINES || superset) {
         const int pr = seg_post<W>(s, r, a, ev, order2, stream_ntiles);
         if (pr < 0) return pr;
         if (pr > 0) continue;
      }
      if (!a.fin) hipLaunchKernelGGL(k_seg_end, dim3(1), dim3(1), 0, st, a, (need_nh ? 1 : 0) | (superset && !nh_is_count ? 2 : 0));      /* (a.fin: the segment's last launch ended it) */
      if (ev) HIP_TRY(hipEventRecord(ev[3], st), EIO);
      HIP_TRY(hipGetLastError(), EIO);
   }
   HIP_TRY(hipMemcpyAsync(s->h_cnt, c, sizeof(Counters), hipMemcpyDeviceToHost, s->stream), EIO);
   return 0;
}

/* ========================================================================== */
/* Packed read batches (seeq_packed.h)                                        */
/* ========================================================================== */
/* (reads per segment: seeqdev_scan.pk_seg_reads) */

static int run_packed(seeqdev_scan *s)
{
   const seeqdev_pattern *pat = s->pat;
   const seeqdev_packed_t &b = s->packed;
   const int options = s->options, want = s->want;
   const int match_opt = options & 3;
   const bool nh_is_count = want == SEEQDEV_WANT_COUNTMATCH || (want == SEEQDEV_WANT_RECORDS && match_opt == SQ_ALL);
   const int fw = pat->wlen <= FUSED_MAX_WLEN ? 1 : 2;
   Counters *c = s->d_cnt;
   hipStream_t st = s->stream;
   const uint32_t L = b.read_len;
   /* the exact pass reads the candidates' windows from the batch itself (seeq_verify_packed.h) -- no staging text -- unless SQ_ALL records are
      wanted (k_emit_all recovers their starts from text) or the round-3 exact pass is asked for */
   const bool direct = !(want == SEEQDEV_WANT_RECORDS && match_opt == SQ_ALL);
   const uint32_t pitch = (L + 1u + 15u) & ~15u;            /* bytes per line of the staging text (L <= 256: the newline's word exists for every lane count up to 17; 16 lanes serve L <= 255, L = 256 below) */
   /* workspace: per read of a segment, per candidate */
   size_t PACKED_SEG_READS = s->pk_seg_reads;
   /* the staging text is addressed with 32-bit offsets: where it is used a segment holds no more reads than it has lines for (every wave fills its own
      share of it and a share never sees more candidates than its wave has reads) -- smaller segments, not E2BIG (round 4 refused above 26.8 M hit-list
      entries at read_len 150, whether or not the staging text was used at all) */
   const size_t stage_lines_max = (size_t)(0xFFFF0000ull / pitch);
   if (!direct && PACKED_SEG_READS + 64u * MAX_FUSED_GRID > stage_lines_max)
      PACKED_SEG_READS = (stage_lines_max - 64u * MAX_FUSED_GRID) & ~(size_t)63;
   const size_t seg_reads = b.nreads < PACKED_SEG_READS ? (size_t)b.nreads : PACKED_SEG_READS;
   const size_t cap_use = (!direct && s->cap_hitlines > stage_lines_max) ? stage_lines_max : s->cap_hitlines;      /* hit-list entries a segment may make */
   if (seg_reads > s->cap_pk_reads) {
      if (ws_alloc((void **)&s->pk_cand, seg_reads * sizeof(uint32_t))) return -1;
      if (ws_alloc((void **)&s->pk_slot, seg_reads * sizeof(uint32_t))) return -1;
      if (ws_alloc((void **)&s->pk_coff, (seg_reads / 64 + 1) * sizeof(uint32_t))) return -1;
      if (ws_alloc((void **)&s->pk_bmask, (seg_reads / 64 + 1) * sizeof(uint64_t))) return -1;
      s->cap_pk_reads = seg_reads;
   }
   if (!direct && cap_use * (size_t)pitch > s->cap_pk_stage) {
      if (ws_alloc((void **)&s->pk_stage, cap_use * (size_t)pitch + 64)) return -1;
      s->cap_pk_stage = cap_use * (size_t)pitch;
   }
   if (s->cap_hitlines > s->cap_pk_last) {
      if (ws_alloc((void **)&s->pk_last, s->cap_hitlines * sizeof(uint32_t))) return -1;
      s->cap_pk_last = s->cap_hitlines;
   }
   {  /* block sums of the scans over per-read arrays */
      const size_t nb = seg_reads / SCAN_BLOCK + 2;
      if (nb > s->cap_scan_ws) { if (ws_alloc((void **)&s->scan_ws, nb * sizeof(uint32_t))) return -1; s->cap_scan_ws = nb; }
   }
   /* EQ tables of the exact pass (as run_segments makes them) */
   if (eq_tables_upload(s, pat, options, fw)) return -1;
   HIP_TRY(hipMemsetAsync(c, 0, sizeof(Counters), st), EIO);
   s->clk_valid = false;                                    /* (the packed walk reads no clock) */
   /* four bases per gather over the quad table (seeq_dfa.h section 3b) when the pattern has one and the false candidates it adds
      -- each an exact-pass window, ~13 walks' worth -- stay below the gathers it saves: 4 % of the reads */
   const bool quad = pat->quad_state == 1 && !s->knobs.no_packed_quad && (pat->quad_pacc - pat->pair_pacc) * (double)L <= 0.04;
   s->last_packed_quad = quad;
   const size_t dfa_lds = quad ? (size_t)pat->quad_units * 16 : (size_t)pat->pair_units * 16;
   const void *walk_fn = quad ? (const void *)k_packed_walk<true> : (const void *)k_packed_walk<false>;
   int per_cu = occupancy_of(s, walk_fn, 64 * STREAM_NW, dfa_lds);
   if (per_cu < 0) return -1;
   /* persistent grid, but no more waves than blocks of 64 reads: every wave owns cap / waves lines of the staging text */
   unsigned wgrid = (unsigned)(s->ncu * per_cu);
   {
      const size_t nblocks = (seg_reads + 63) / 64, need = (nblocks + STREAM_NW_HOST - 1) / STREAM_NW_HOST;
      if (need < wgrid) wgrid = (unsigned)(need ? need : 1);
   }
   const size_t hit_blocks = (s->cap_hitlines + WG - 1) / WG;
   unsigned grid_hits = (unsigned)(hit_blocks < (size_t)s->ncu * 16 ? hit_blocks : (size_t)s->ncu * 16);
   if (grid_hits == 0) grid_hits = 1;
   const size_t nseg = (size_t)((b.nreads + PACKED_SEG_READS - 1) / PACKED_SEG_READS);
   s->prof_segs = 0;
   if (s->prof && nseg > s->nev_seg) {
      hipEvent_t *g = (hipEvent_t *)realloc(s->ev, 4 * nseg * sizeof(hipEvent_t));
      if (!g) { seeqerr = 0; errno = ENOMEM; return -1; }
      s->ev = g;
      for (size_t i = 4 * s->nev_seg; i < 4 * nseg; i++) HIP_TRY(hipEventCreate(&s->ev[i]), EIO);
      s->nev_seg = nseg;
   }
   if (s->prof) s->prof_segs = nseg;
   for (size_t sg = 0; sg < nseg; sg++) {
      hipEvent_t *ev = s->prof ? s->ev + 4 * sg : NULL;
      PackedArgs p;
      memset(&p, 0, sizeof p);
      p.bases = (const uint8_t *)b.bases; p.nmask = (const uint8_t *)b.nmask;
      p.first = (uint64_t)sg * PACKED_SEG_READS;
      p.nreads = (uint32_t)(b.nreads - p.first < PACKED_SEG_READS ? b.nreads - p.first : PACKED_SEG_READS);
      p.read_len = L; p.stride = b.stride; p.nstride = b.nstride;
      p.total_bytes = b.nreads * (uint64_t)b.stride;
      p.dfa = quad ? pat->d_quad : pat->d_pair; p.dfa_units = quad ? pat->quad_units : pat->pair_units;
      p.cand = s->pk_cand; p.cslot = s->pk_slot; p.boff = s->pk_coff; p.bmask = s->pk_bmask; p.stage = direct ? nullptr : s->pk_stage;
      p.wave_cap = (uint32_t)(cap_use / ((size_t)wgrid * STREAM_NW_HOST));
      p.pitch = pitch;
      p.hit_start = s->hit_start; p.hit_line = s->hit_line; p.hit_col = s->hit_col; p.hit_last = s->pk_last; p.nh = s->nh;
      p.cap = (uint32_t)cap_use;
      p.line_base = p.first;
      p.cnt = c;
      if (ev) { HIP_TRY(hipEventRecord(ev[0], st), EIO); HIP_TRY(hipEventRecord(ev[1], st), EIO); }
      {
         void *kargs[] = {&p};
         HIP_TRY(hipLaunchKernel(walk_fn, dim3(wgrid), dim3(64 * STREAM_NW), kargs, dfa_lds, st), EIO);
      }
      if (ev) HIP_TRY(hipEventRecord(ev[2], st), EIO);
      /* candidates before every block of 64 reads, their number */
      launch_scanset(s, st, s->pk_coff, nullptr, nullptr, (p.nreads + 63u) >> 6, &c->seg_nhitlines, nullptr, nullptr);
      hipLaunchKernelGGL(k_packed_counts, dim3(1), dim3(1), 0, st, p);
      hipLaunchKernelGGL(k_packed_list, dim3((unsigned)(((size_t)p.nreads / 1024 + 4) / 4)), dim3(256), 0, st, p);      /* a wave per 16 blocks of 64 reads */
      /* from here: the exact pass over the staging text, as behind k_pair */
      ScanArgs a;
      memset(&a, 0, sizeof a);
      a.text = direct ? nullptr : s->pk_stage;              /* (direct: nothing reads text -- the windows come from the batch) */
      a.nbytes = direct ? 0 : cap_use * (uint64_t)pitch;
      a.seg_base = 0; a.seg_len = (uint32_t)a.nbytes; a.first_seg = sg == 0;
      a.peq = pat->d_peq;
      a.m = pat->wlen; a.tau = pat->tau; a.options = options & ~(MASK_NONDNA | MASK_INPUT); a.want = want;
      a.hit_start = s->hit_start; a.hit_line = s->hit_line; a.cap_hitlines = (uint32_t)s->cap_hitlines; a.nh = s->nh;
      a.records = s->records; a.cap_records = s->cap_records; a.rec_off = s->rec_off;
      a.use_nh = 3u; a.filter = 1u;
      a.skip_back = (uint32_t)(pat->wlen + pat->tau);
      a.hit_last = s->pk_last;
      a.window_ok = 1u;
      a.cnt = c;
      a.rec_pitch = L + 1u;                                 /* seeqdevScanCopyOffsets: the read's offset in the ASCII form of the batch */
      const uint32_t *eqp = (const uint32_t *)s->d_eqtab;
      const uint32_t *hcol = s->hit_col;
      uint4 *ecache = want == SEEQDEV_WANT_RECORDS ? s->ow.tmp : nullptr;
      const int seg_flags = 1 | (!nh_is_count ? 2 : 0);
      {
         /* the exact pass of the candidates: k_verify_packed on the batch itself, or (SQ_ALL records) k_verify over the staging text -- ASCII lines,
            no byte of them is skipped (seeq_verify.h) */
         const bool count_any = want != SEEQDEV_WANT_COUNTMATCH && !(want == SEEQDEV_WANT_RECORDS && match_opt == SQ_ALL);
         const int var = (want == SEEQDEV_WANT_RECORDS && match_opt == SQ_BEST) ? VERIFY_BEST : count_any ? VERIFY_ANY : VERIFY_ALL;
         a.nh_sum = s->nh_sum;
         a.nz_sum = nh_is_count ? s->nh_sum + (s->cap_hitlines / 256 + 2) : nullptr;
         a.fin = (want == SEEQDEV_WANT_RECORDS && var == VERIFY_ALL) ? 0u : 1u + (uint32_t)seg_flags;
         if (direct) seeq_launch_verify_packed(fw, var, grid_hits, st, a, b.bases, b.nmask, b.stride, b.nstride, L, p.total_bytes,
                                               b.nmask ? b.nreads * (uint64_t)b.nstride : 0ull, eqp, hcol, ecache);
         else seeq_launch_verify(fw, var, grid_hits, st, a, eqp, hcol, ecache);
         if (want == SEEQDEV_WANT_RECORDS && var != VERIFY_ALL) seeq_launch_emit1(grid_hits, st, a, ecache);
         else if (want == SEEQDEV_WANT_RECORDS) {
            seeq_launch_emit_all(fw, grid_hits, grid_hits, st, a, eqp, hcol, ecache);
            hipLaunchKernelGGL(k_seg_end, dim3(1), dim3(1), 0, st, a, seg_flags);
         }
      }
      if (ev) HIP_TRY(hipEventRecord(ev[3], st), EIO);
      HIP_TRY(hipGetLastError(), EIO);
   }
   HIP_TRY(hipMemcpyAsync(s->h_cnt, c, sizeof(Counters), hipMemcpyDeviceToHost, st), EIO);
   s->last_path = 8;
   s->last_filter = true;
   return 0;
}

/* ========================================================================== */
/* Several patterns, one walk (seeq_multi.h)                                    */
/* ========================================================================== */
static int multi_ws_ensure(seeqdev_scan *s, int npat)
{
   if (!s->d_mcnt) {
      HIP_TRY(hipMalloc((void **)&s->d_mcnt, SEEQ_MULTI_MAX * sizeof(Counters)), ENOMEM);
      HIP_TRY(hipHostMalloc((void **)&s->h_mcnt, SEEQ_MULTI_MAX * sizeof(Counters), hipHostMallocDefault), ENOMEM);
   }
   if (s->cap_hitlines > s->cap_ml) {
      if (ws_alloc((void **)&s->ml_mask, s->cap_hitlines * sizeof(uint32_t))) return -1;
      if (ws_alloc((void **)&s->ml_first, s->cap_hitlines * sizeof(uint32_t))) return -1;
      if (ws_alloc((void **)&s->ml_last, s->cap_hitlines * sizeof(uint32_t))) return -1;
      s->cap_ml = s->cap_hitlines;
   }
   if (s->cap_hitlines > s->cap_mp) {
      if (ws_alloc((void **)&s->mp_idx, s->cap_hitlines * sizeof(uint32_t))) return -1;
      if (ws_alloc((void **)&s->mp_nh, s->cap_hitlines * sizeof(uint32_t))) return -1;
      s->cap_mp = s->cap_hitlines;
   }
   if (!s->d_mx) {
      s->mx_slots = 64;                                    /* segments whose argument arrays may be in flight (a run of more segments waits for the stream in between) */
      HIP_TRY(hipMalloc((void **)&s->d_mx, s->mx_slots * SEEQ_MULTI_MAX * sizeof(MultiExact)), ENOMEM);
      HIP_TRY(hipHostMalloc((void **)&s->h_mx, s->mx_slots * SEEQ_MULTI_MAX * sizeof(MultiExact), hipHostMallocDefault), ENOMEM);
      s->mx_next = 0;
   }
   {
      const size_t nbp = (s->cap_hitlines / (size_t)npat) / SCAN_BLOCK + 2;
      if ((size_t)npat * nbp > s->cap_m_scan_ws) {
         if (ws_alloc((void **)&s->m_scan_ws, (size_t)npat * nbp * sizeof(uint32_t))) return -1;
         s->cap_m_scan_ws = (size_t)npat * nbp;
      }
   }
   const size_t nb = s->cap_hitlines / MULTI_BLOCK + 2;
   if ((size_t)npat * nb > s->cap_m_bsum) {
      if (ws_alloc((void **)&s->m_bsum, (size_t)npat * nb * sizeof(uint32_t))) return -1;
      s->cap_m_bsum = (size_t)npat * nb;
   }
   return 0;
}

/* The part of a segment behind the union walk: `ua` = the union scan's arguments (hit list made, bounds done). */
static int multi_post(seeqdev_scan *s, const ScanArgs &ua, hipStream_t st)
{
   const MultiPlan *mp = s->mplan;
   const int npat = mp->npat;
   const int options = s->options, want = s->want;
   const int match_opt = options & 3;
   const bool nh_is_count = want == SEEQDEV_WANT_COUNTMATCH || (want == SEEQDEV_WANT_RECORDS && match_opt == SQ_ALL);
   const uint32_t capP = (uint32_t)(s->cap_hitlines / (size_t)npat);
   const uint64_t capR = s->cap_records / (uint64_t)npat;
   MultiArgs m;
   memset(&m, 0, sizeof m);
   m.text = ua.text; m.nbytes = ua.nbytes; m.seg_base = ua.seg_base;
   m.hit_start = s->hit_start; m.hit_line = s->hit_line; m.hit_col = s->hit_col; m.nh = s->nh;
   m.ucnt = s->d_cnt;
   m.res_next = mp->d_res_next; m.res_mask = mp->d_res_mask; m.res_states = mp->res_states;
   m.maxspan = (uint32_t)mp->maxspan;
   m.window_ok = ua.window_ok;
   m.options = options;
   /* whole patterns in the resolve automaton: its sets are exact -- counting lines needs no exact pass (as behind k_stream's complete automata) */
   const bool trust = mp->exact && want == SEEQDEV_WANT_COUNTLINES;
   m.trust = trust ? 1u : 0u;
   m.lmask = s->ml_mask; m.lfirst = s->ml_first; m.llast = s->ml_last;
   m.npat = (uint32_t)npat; m.capP = capP;
   m.p_idx = s->mp_idx;
   m.pcnt = s->d_mcnt;
   m.bsum = s->m_bsum;
   m.nb = (uint32_t)(s->cap_hitlines / MULTI_BLOCK + 2);
   {
      const size_t blocks = (s->cap_hitlines + MULTI_RESOLVE_WG - 1) / MULTI_RESOLVE_WG;
      const unsigned grid = (unsigned)(blocks < (size_t)s->ncu * 2 ? blocks : (size_t)s->ncu * 2);      /* persistent: the table is staged once per workgroup */
      HIP_TRY(hipMemsetAsync(s->ml_mask, 0, s->cap_hitlines * sizeof(uint32_t), st), EIO);      /* the lanes of a line's entries OR / MAX into them */
      HIP_TRY(hipMemsetAsync(s->ml_last, 0, s->cap_hitlines * sizeof(uint32_t), st), EIO);
      /* the automaton in LDS when it fits what a workgroup may ask for beside the kernel's static arrays (the device's limit, not a literal) */
      const size_t lds2 = (size_t)mp->res_states * 20, lds1 = (size_t)mp->res_states * 16;
      const size_t lds_room = s->lds_per_wg > 1024 ? s->lds_per_wg - 1024 : 0;
      HIP_TRY(hipGetLastError(), EIO);                       /* (an error of an EARLIER launch of this segment is a failure, not a reason for the per-pattern fall-back) */
      if (lds2 <= lds_room && lds2 <= 65536) hipLaunchKernelGGL(k_multi_resolve<2>, dim3(grid ? grid : 1), dim3(MULTI_RESOLVE_WG), lds2, st, m);
      else if (lds1 <= lds_room && lds1 <= 65536) hipLaunchKernelGGL(k_multi_resolve<1>, dim3(grid ? grid : 1), dim3(MULTI_RESOLVE_WG), lds1, st, m);
      else hipLaunchKernelGGL(k_multi_resolve<0>, dim3(grid ? grid : 1), dim3(MULTI_RESOLVE_WG), 0, st, m);
      {
         const hipError_t le = hipGetLastError();            /* this launch refused for its resources: a scan per pattern (seeqdevScanRunMulti); anything else fails */
         if (le == hipErrorInvalidValue || le == hipErrorLaunchOutOfResources || le == hipErrorInvalidConfiguration) return 1;
         if (le != hipSuccess) return hip_fail(le, "k_multi_resolve", EIO);
      }
      hipLaunchKernelGGL(k_multi_reduce, dim3(m.nb), dim3(256), 0, st, m);
      hipLaunchKernelGGL(k_multi_top, dim3((unsigned)npat), dim3(256), 0, st, m);
      if (trust) { HIP_TRY(hipGetLastError(), EIO); return 0; }
      hipLaunchKernelGGL(k_multi_apply, dim3(m.nb), dim3(256), 0, st, m);
   }
   /* The exact pass, every pattern in one launch per step (blockIdx.y = pattern; one-word patterns first, then the two-word
      ones): the patterns' arguments go to HBM through a page-locked ring, one slot per segment. */
   const size_t hit_blocks = ((size_t)capP + WG - 1) / WG;
   unsigned grid_hits = (unsigned)(hit_blocks < (size_t)s->ncu * 4 ? hit_blocks : (size_t)s->ncu * 4);      /* (x npat workgroups per launch) */
   if (grid_hits == 0) grid_hits = 1;
   if (s->mx_next == s->mx_slots) { HIP_TRY(hipStreamSynchronize(st), EIO); s->mx_next = 0; }
   MultiExact *hx = s->h_mx + s->mx_next * SEEQ_MULTI_MAX, *dx = s->d_mx + s->mx_next * SEEQ_MULTI_MAX;
   s->mx_next++;
   const uint32_t nbp = (uint32_t)((size_t)capP / SCAN_BLOCK + 2);
   int order[SEEQ_MULTI_MAX], n1 = 0, n2 = 0;
   for (int k = 0; k < npat; k++) if (mp->fw[k] == 1) order[n1++] = k;
   for (int k = 0; k < npat; k++) if (mp->fw[k] != 1) order[n1 + n2++] = k;
   for (int q = 0; q < npat; q++) {
      const int k = order[q];
      MultiExact &x = hx[q];
      ScanArgs &a = x.a;
      a = ua;
      a.m = mp->m[k]; a.tau = mp->tau[k];
      a.hit_start = s->hit_start; a.hit_line = s->hit_line; a.cap_hitlines = capP;      /* the union's lines, through this pattern's index list */
      a.hit_idx = s->mp_idx + (size_t)k * capP;
      a.nh = s->mp_nh + (size_t)k * capP;
      a.records = s->records + (uint64_t)k * capR; a.cap_records = capR; a.rec_off = s->rec_off + (uint64_t)k * capR;
      a.use_nh = 3u; a.filter = 1u;
      a.skip_back = (uint32_t)mp->maxspan;
      a.hit_last = s->ml_last;
      a.window_ok = 1u;
      a.tile_dirty = nullptr; a.tile_dmask = nullptr; a.stream_ntiles = 0; a.stream_ch = 0;
      a.cnt = s->d_mcnt + k;
      x.eq = mp->d_eq + (size_t)k * 1536;
      x.hcol = s->ml_first;
      x.cache = want == SEEQDEV_WANT_RECORDS ? s->ow.tmp + (size_t)k * capP : nullptr;
      x.scan_ws = s->m_scan_ws + (size_t)k * nbp;
      x.nb = nbp;
      x.seg_end_flags = 1 | (!nh_is_count ? 2 : 0);
   }
   HIP_TRY(hipMemcpyAsync(dx, hx, (size_t)npat * sizeof(MultiExact), hipMemcpyHostToDevice, st), EIO);
   const int mo = match_opt == SQ_COUNT ? SQ_FIRST : match_opt;
   if (n1) hipLaunchKernelGGL((k_exact1m<SQ_MODE_COUNT, 1, -1>), dim3(grid_hits, (unsigned)n1), dim3(WG), 0, st, (const MultiExact *)dx);
   if (n2) hipLaunchKernelGGL((k_exact1m<SQ_MODE_COUNT, 2, -1>), dim3(grid_hits, (unsigned)n2), dim3(WG), 0, st, (const MultiExact *)(dx + n1));
   if (nh_is_count) hipLaunchKernelGGL(k_multi_count_nonzero, dim3(grid_hits < 128 ? grid_hits : 128, (unsigned)npat), dim3(WG), 0, st, (const MultiExact *)dx);
   hipLaunchKernelGGL(k_multi_scan_reduce, dim3(nbp, (unsigned)npat), dim3(WG), 0, st, (const MultiExact *)dx);
   hipLaunchKernelGGL(k_multi_scan_top, dim3((unsigned)npat), dim3(WG), 0, st, (const MultiExact *)dx, want == SEEQDEV_WANT_RECORDS ? 1 : 0);
   hipLaunchKernelGGL(k_multi_scan_apply, dim3(nbp, (unsigned)npat), dim3(WG), 0, st, (const MultiExact *)dx);
   if (want == SEEQDEV_WANT_RECORDS) {
      if (mo == SQ_BEST) {
         if (n1) hipLaunchKernelGGL((k_exact1m<SQ_MODE_EMIT, 1, SQ_BEST>), dim3(grid_hits, (unsigned)n1), dim3(WG), 0, st, (const MultiExact *)dx);
         if (n2) hipLaunchKernelGGL((k_exact1m<SQ_MODE_EMIT, 2, SQ_BEST>), dim3(grid_hits, (unsigned)n2), dim3(WG), 0, st, (const MultiExact *)(dx + n1));
      } else {
         if (n1) hipLaunchKernelGGL((k_exact1m<SQ_MODE_EMIT, 1, -1>), dim3(grid_hits, (unsigned)n1), dim3(WG), 0, st, (const MultiExact *)dx);
         if (n2) hipLaunchKernelGGL((k_exact1m<SQ_MODE_EMIT, 2, -1>), dim3(grid_hits, (unsigned)n2), dim3(WG), 0, st, (const MultiExact *)(dx + n1));
      }
   }
   hipLaunchKernelGGL(k_multi_seg_end, dim3((unsigned)npat), dim3(1), 0, st, (const MultiExact *)dx);
   HIP_TRY(hipGetLastError(), EIO);
   return 0;
}

static int dispatch_run(seeqdev_scan *s)
{
   if (s->is_packed) return run_packed(s);
   const int W = s->pat->words;
   if (W <= 1) return run_segments<1>(s);
   if (W <= 2) return run_segments<2>(s);
   if (W <= 4) return run_segments<4>(s);
   if (W <= 8) return run_segments<8>(s);
   return run_segments<16>(s);
}

static int scan_setup(seeqdev_scan_t *s, const seeqdev_pattern_t *pat, const void *d_text, size_t nbytes, int options, int want, int hl_div);

extern "C" int seeqdevScanRun(seeqdev_scan_t *s, const seeqdev_pattern_t *pat, const void *d_text, size_t nbytes,
                              int options, int want)
{
   if (scan_setup(s, pat, d_text, nbytes, options, want, 8)) return -1;
   if (dispatch_run(s)) return -1;
   s->ran = true;
   return 0;
}

/* Everything of a run before its launches: arguments, fall-back flags, the optimistic workspace (hit lines: one line in
   `hl_div`), the line-length sample. */
static int scan_setup(seeqdev_scan_t *s, const seeqdev_pattern_t *pat, const void *d_text, size_t nbytes, int options, int want, int hl_div)
{
   seeqerr = 0;
   if (!s || !pat || (!d_text && nbytes) || want < 0 || want > 2) { errno = EINVAL; return -1; }
   if (pat->device != s->device) {
      snprintf(g_last_error, sizeof g_last_error, "pattern lives on device %d, scan context on device %d", pat->device, s->device);
      errno = EINVAL;
      return -1;
   }
   if (use_device(s->device)) return -1;
   s->pat = pat; s->text = d_text; s->nbytes = nbytes; s->options = options; s->want = want;
   s->ran = false;
   s->is_packed = false;
   if ((s->no_stream || s->no_stream_nd || s->force_ll || s->no_window || s->no_leaders) && --s->fallback_ttl <= 0) s->no_stream = s->no_stream_nd = s->force_ll = s->no_window = s->no_leaders = false;
   /* Optimistic default workspace: lines average >= 32 bytes, one line in 8 hits, 1 record per hit line.
      A too-small workspace is detected on the device and fixed by one re-run in seeqdevScanFetch. */
   const size_t seg = nbytes < s->seg_bytes ? nbytes : s->seg_bytes;
   size_t want_lines = s->cap_lines, want_hl = s->cap_hitlines, want_rec = s->cap_records;
   if (!s->user_reserved) {
      const size_t guess = (options & SEEQDEV_SINGLELINE) ? 1 : seg / 32 + 1024;
      if (guess > want_lines) want_lines = guess;
      if (want_lines / (size_t)hl_div + 1024 > want_hl) want_hl = want_lines / (size_t)hl_div + 1024;
      /* the one-pass kernels cut the hit-line workspace into one slice per wave (<= 8 192 of them): room for 64
         entries each, or the first scan with a hit always costs a second pass */
      if (!(options & SEEQDEV_SINGLELINE) && want_hl < (size_t)8192 * 64) want_hl = (size_t)8192 * 64;
      if (want_hl > want_rec) want_rec = want_hl;
   }
   if (reserve_impl(s, nbytes ? nbytes : 1, want_lines, want_hl, want_rec)) return -1;
   /* Average line length (tile sizing of the fused kernel): caller's hint, else a 64 KiB sample. */
   if (s->line_hint > 0) {
      s->avg_line = s->line_hint;
   } else if (!(options & SEEQDEV_SINGLELINE) && nbytes && (s->avg_text != d_text || s->avg_nbytes != nbytes || ++s->sample_age >= 64)) {
      s->sample_age = 0;
      const size_t n = nbytes < SAMPLE_BYTES ? nbytes : SAMPLE_BYTES;
      HIP_TRY(hipMemcpyAsync(s->h_sample, d_text, n, hipMemcpyDeviceToHost, s->stream), EIO);
      HIP_TRY(hipStreamSynchronize(s->stream), EIO);
      size_t nl = 0;
      size_t foreign = 0;                                  /* bytes outside A C G T N (either case) and newline -- FASTA header lines included: their tiles take k_pair's slow path like any other (2-line FASTA records: 5.1 against 11.3 G lines/s on k_stream) */
      for (size_t i = 0; i < n; i++) {
         const uint8_t b = s->h_sample[i];
         nl += b == '\n';
         foreign += b != '\n' && sq_class_of(b, 0) >= 5;
      }
      s->avg_line = nl ? (double)n / (double)nl : 1e9;
      /* more than one foreign byte per 4 KB (FASTQ: every quality line): nearly every 8 KB tile of k_pair would take its slow
         path (exact alphabet check over the text fetched again) -- k_stream walks such text at full speed */
      s->sample_dirty = foreign * 4096 > n;
      s->avg_text = d_text;
      s->avg_nbytes = nbytes;
   }
   return 0;
}

extern "C" int seeqdevScanFetch(seeqdev_scan_t *s, seeqdev_counts_t *counts)
{
   seeqerr = 0;
   if (!s || !s->ran) { errno = EINVAL; return -1; }
   if (use_device(s->device)) return -1;
   /* Overflows surface one stage at a time (lines, hit lines, records, then k_stream's fall-backs): up to six
      re-runs, and the result of the last one is checked too. */
   for (int attempt = 0; attempt < 8; attempt++) {
      HIP_TRY(hipStreamSynchronize(s->stream), EIO);
      const Counters h = *s->h_cnt;
      if (!h.overflow) {
         s->counts.nlines = h.lines;
         s->counts.nmatchlines = h.matchlines;
         s->counts.nhits = h.hits;
         s->counts.nrecords = h.records;
         s->counts.nheaders = h.headers;
         if (counts) *counts = s->counts;
         for (int i = 0; i < 4; i++) s->acc_ms[i] = 0.f;
         s->fwd_ms_avg = 0.f;
         for (size_t sg = 0; sg < s->prof_segs; sg++) {
            hipEvent_t *ev = s->ev + 4 * sg;
            float t01 = 0, t12 = 0, t23 = 0;
            if (sg == 0 && s->prof_segs > s->cap_launch_ms) {
               float *g = (float *)realloc(s->launch_ms, s->prof_segs * sizeof(float));
               if (g) { s->launch_ms = g; s->cap_launch_ms = s->prof_segs; }
            }
            (void)hipEventElapsedTime(&t01, ev[0], ev[1]);
            (void)hipEventElapsedTime(&t12, ev[1], ev[2]);
            (void)hipEventElapsedTime(&t23, ev[2], ev[3]);
            s->acc_ms[0] += t01; s->acc_ms[1] += t12; s->acc_ms[2] += t23; s->acc_ms[3] += t01 + t12 + t23;
            if (sg < s->cap_launch_ms) s->launch_ms[sg] = t12;
         }
         s->clk_mhz = 0.f;
         if (s->clk_valid && s->prof_segs && s->clk_probe && s->cap_clk_probe >= s->prof_segs) {
            double sum = 0; size_t nn = 0;
            for (size_t sg = 0; sg < s->prof_segs; sg++) {
               const unsigned long long *q = s->clk_probe + 4 * sg;
               if (q[3] > q[1] && q[2] > q[0]) { sum += (double)(q[2] - q[0]) / (double)(q[3] - q[1]) * 100.0; nn++; }      /* s_memrealtime: 100 MHz */
            }
            if (nn) s->clk_mhz = (float)(sum / (double)nn);
         }
         if (s->prof_segs) s->fwd_ms_avg = s->acc_ms[1] / (float)s->prof_segs;
         s->h2d_ms = 0.f;
         if (s->prof && s->have_h2d_ev) (void)hipEventElapsedTime(&s->h2d_ms, s->ev_h2d[0], s->ev_h2d[1]);
         return 0;
      }
      /* Grow to what the device reported (plus slack for the parts it could not see) and re-run. */
      size_t nl = s->cap_lines, nhl = s->cap_hitlines, nrec = s->cap_records;
      if (h.overflow & 1u) nl = (size_t)h.need_lines + (h.need_lines >> 3) + 64;
      if (h.overflow & 2u) nhl = (size_t)h.need_hitlines + (h.need_hitlines >> 3) + 64;
      if (h.overflow & 64u) {
         snprintf(g_last_error, sizeof g_last_error, "internal inconsistency in the hit list (k_stream_bounds)");
         errno = EIO;
         return -1;
      }
      if (h.overflow & 8u) s->no_stream = true;
      if (h.overflow & 16u) s->no_stream_nd = true;
      if (h.overflow & 32u) s->force_ll = true;
      if (h.overflow & 128u) s->no_window = true;            /* a line with candidates on both sides of a segment seam */
      if (h.overflow & 256u) s->no_leaders = true;           /* a leader's fresh start inside the walk before it */
      if (h.overflow & (8u | 16u | 32u | 128u | 256u)) s->fallback_ttl = 32;
      if (h.overflow & 4u) {
         /* need_records keeps counting after the overflow, so it is the total of this run. */
         nrec = (size_t)h.need_records + (size_t)(h.need_records >> 3) + 64;
      }
      if ((h.overflow & 1u) && nhl < nl / 8) nhl = nl / 8 + 64;
      if (attempt == 7) break;
      if (reserve_impl(s, s->nbytes, nl, nhl, nrec)) return -1;
      if (dispatch_run(s)) return -1;
   }
   snprintf(g_last_error, sizeof g_last_error, "workspace did not converge");
   errno = ENOMEM;
   return -1;
}

extern "C" const seeqdev_hit_t *seeqdevScanRecordsDevice(const seeqdev_scan_t *s) { return s ? s->records : NULL; }

extern "C" int seeqdevScanCopyRecords(seeqdev_scan_t *s, seeqdev_hit_t *host_out, size_t first, size_t n)
{
   seeqerr = 0;
   if (!s || (!host_out && n)) { errno = EINVAL; return -1; }
   if (first + n > s->counts.nrecords) { errno = EINVAL; return -1; }
   if (n == 0) return 0;
   if (use_device(s->device)) return -1;
   HIP_TRY(hipMemcpyAsync(host_out, s->records + first, n * sizeof(seeqdev_hit_t), hipMemcpyDeviceToHost, s->stream),
           EIO);
   HIP_TRY(hipStreamSynchronize(s->stream), EIO);
   return 0;
}

/* A packed read batch resident in HBM (seeq_amd.h: seeqdev_packed_t, seeq_packed.h): asynchronous, seeqdevScanFetch waits. */
extern "C" int seeqdevScanPacked(seeqdev_scan_t *s, const seeqdev_pattern_t *pat, const seeqdev_packed_t *batch, int options, int want)
{
   seeqerr = 0;
   if (!s || !pat || !batch || want < 0 || want > 2 || (batch->nreads && !batch->bases)) { errno = EINVAL; return -1; }
   if (batch->read_len < 1 || batch->read_len > 256 || batch->stride < (batch->read_len + 3) / 4 ||
       (batch->nmask && batch->nstride < (batch->read_len + 7) / 8) || (options & (MASK_INPUT | SEEQDEV_FASTA | SEEQDEV_SINGLELINE))) { errno = EINVAL; return -1; }
   if (pat->device != s->device) { errno = EINVAL; return -1; }
   if (use_device(s->device)) return -1;
   seeqdev_pattern *mp = const_cast<seeqdev_pattern *>(pat);
   if (pat->wlen <= FUSED_MAX_WLEN2 && __atomic_load_n(&mp->pair_state, __ATOMIC_ACQUIRE) == 0) pattern_plan_pair(mp);
   if (pat->wlen <= FUSED_MAX_WLEN2 && mp->pair_state == 1 && __atomic_load_n(&mp->quad_state, __ATOMIC_ACQUIRE) == 0) pattern_plan_quad(mp);
   if (pat->wlen > FUSED_MAX_WLEN2 || mp->pair_state != 1) {
      /* not the packed walk's pattern (more than 62 positions, or no pair automaton): the batch is unpacked on the device and the
         ASCII scan runs over it -- the reference takes any pattern (libseeq.c:43-138), so does this entry */
      const size_t tbytes = (size_t)batch->nreads * (batch->read_len + 1u);
      if (tbytes > s->cap_unpack) {
         if (ws_alloc((void **)&s->d_unpack, tbytes + 64)) return -1;
         s->cap_unpack = tbytes;
      }
      if (batch->nreads) {
         const uint64_t threads = batch->nreads * (uint64_t)(batch->read_len / 16u + 1u), blocks = (threads + 255) / 256;
         if (blocks > 0x7FFFFFFFull) { errno = E2BIG; return -1; }
         hipLaunchKernelGGL(k_unpack_ascii, dim3((unsigned)blocks), dim3(256), 0, s->stream, (const uint8_t *)batch->bases, (const uint8_t *)batch->nmask,
                            batch->nreads, batch->read_len, batch->stride, batch->nstride, s->d_unpack);
         HIP_TRY(hipGetLastError(), EIO);
      }
      s->avg_text = NULL;                                   /* (new contents behind the same pointer: sample again) */
      return seeqdevScanRun(s, pat, s->d_unpack, tbytes, options, want);
   }
   s->pat = pat; s->text = NULL; s->nbytes = 0; s->options = options; s->want = want;
   s->ran = false;
   s->is_packed = true;
   s->packed = *batch;
   /* optimistic workspace: one read in eight is a candidate (grown by the re-run of seeqdevScanFetch when it is not) */
   const size_t seg = batch->nreads < s->pk_seg_reads ? (size_t)batch->nreads : s->pk_seg_reads;
   size_t want_hl = s->cap_hitlines, want_rec = s->cap_records;
   if (!s->user_reserved) {
      if (seg / 8 + 1024 > want_hl) want_hl = seg / 8 + 1024;
      if (want_hl > want_rec) want_rec = want_hl;
   }
   if (reserve_impl(s, 1, 0, want_hl, want_rec)) return -1;
   if (dispatch_run(s)) return -1;
   s->ran = true;
   return 0;
}

/* The same for ASCII reads resident in HBM (one per line, each read_len bases + '\n'): device to device, asynchronous on `hip_stream`. */
extern "C" int seeqdevPackReadsDevice(const void *d_text, uint64_t nreads, uint32_t read_len, void *d_bases, void *d_nmask, uint32_t stride, uint32_t nstride,
                                      void *hip_stream)
{
   seeqerr = 0;
   if ((!d_text || !d_bases) && nreads) { errno = EINVAL; return -1; }
   if (read_len < 1 || read_len > 256 || stride < (read_len + 3) / 4 || (d_nmask && nstride < (read_len + 7) / 8)) { errno = EINVAL; return -1; }
   if (nreads == 0) return 0;
   const uint64_t blocks = (nreads + 255) / 256;
   if (blocks > 0x7FFFFFFFull) { errno = E2BIG; return -1; }
   hipLaunchKernelGGL(k_pack_ascii, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hip_stream, (const uint8_t *)d_text, nreads, read_len,
                      (uint8_t *)d_bases, (uint8_t *)d_nmask, stride, nstride);
   HIP_TRY(hipGetLastError(), EIO);
   return 0;
}

/* ASCII reads, one per line, every line exactly read_len bases -> the packed layout (host helper for callers that hold text:
 * a caller that holds packed reads already -- BAM, .2bit -- fills seeqdev_packed_t itself).  A C G T U N in either case; any
 * other byte, or a line of another length: -1, errno = EINVAL.  Returns the number of reads. */
extern "C" long seeqdevPackReads(const char *text, size_t nbytes, uint32_t read_len, void *bases_out, void *nmask_out, uint32_t stride, uint32_t nstride)
{
   seeqerr = 0;
   if (!text || !bases_out || read_len < 1 || read_len > 256 || stride < (read_len + 3) / 4 || (nmask_out && nstride < (read_len + 7) / 8)) { errno = EINVAL; return -1; }
   uint8_t *bo = (uint8_t *)bases_out, *no = (uint8_t *)nmask_out;
   long r = 0;
   size_t p = 0;
   while (p < nbytes) {
      if (p + read_len > nbytes) { errno = EINVAL; return -1; }
      uint8_t *b = bo + (size_t)r * stride, *n = no ? no + (size_t)r * nstride : NULL;
      memset(b, 0, stride);
      if (n) memset(n, 0, nstride);
      for (uint32_t i = 0; i < read_len; i++) {
         const unsigned char ch = (unsigned char)text[p + i];
         const unsigned char up = ch & 0xDF;
         unsigned code;
         if (up == 'A' || up == 'C' || up == 'G' || up == 'T' || up == 'U') code = (ch >> 1) & 3u;
         else if (up == 'N') { code = 0; if (!n) { errno = EINVAL; return -1; } n[i >> 3] |= (uint8_t)(0x80u >> (i & 7)); }
         else { errno = EINVAL; return -1; }
         b[i >> 2] |= (uint8_t)(code << (6 - 2 * (i & 3)));
      }
      p += read_len;
      if (p < nbytes) { if (text[p] != '\n') { errno = EINVAL; return -1; } p++; }
      r++;
   }
   return r;
}

/* Page-locked host memory for staging buffers (H2D at link speed instead of through a bounce buffer). */
extern "C" void *seeqdevHostAlloc(size_t bytes)
{
   /* Called from seeqFileMatch's READER THREAD (seeq_file.c slot_reserve): seeqerr is the reference's plain global (libseeq.h:38) and belongs to the
      caller's thread -- this entry reports through errno (thread-local) alone; seeq_file.c clears seeqerr where it hands the failure to the caller. */
   void *p = NULL;
   hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable);      /* every device may copy from it */
   if (e != hipSuccess) { snprintf(g_last_error, sizeof g_last_error, "hipHostMalloc: %s", hipGetErrorString(e)); errno = ENOMEM; return NULL; }
   return p;
}

extern "C" void seeqdevHostFree(void *p)
{
   if (p) (void)hipHostFree(p);
}

/* Device memory for RESIDENT TEXT, chosen by measurement.  The scan kernel's time follows the physical pages a buffer gets from the driver
 * (0.77 / 0.87 / 0.92 ms per 3.75 GiB for the same text, stable for the life of the allocation; power-of-two blocks are fast far more often
 * than requests of an odd size: DESIGN.md section 5 (i)-(l)), so a caller that keeps text resident chooses its buffer once: up to twelve
 * candidate allocations (the plain one, then blocks of p bytes, p = the power of two >= bytes), each filled with synthetic reads and
 * scanned twice with the benchmark pattern; the one whose scan kernel was fastest is returned, the
 * others are freed.  probe_ms (may be NULL): the candidates' scan-kernel times, *nprobed of them.  The buffer's contents are undefined. */
extern "C" void *seeqdevTextAllocFor(seeqdev_scan_t *scan, size_t bytes, int candidates, seeqdev_textinfo_t *info)
{
   seeqerr = 0;
   if (info) memset(info, 0, sizeof *info);
   if (bytes == 0) bytes = 1;
   void *blk[12] = {nullptr};
   size_t blk_bytes[12] = {0};
   float ms[12] = {0};
   int n = 0;
   if (candidates > 12) candidates = 12;
   if (candidates < 2 || bytes < ((size_t)64 << 20)) candidates = 1;      /* (nothing to tell apart on a scan of microseconds) */
   size_t p2 = 1;
   while (p2 < bytes) p2 <<= 1;
   /* what the probing scan context allocates beside the candidates (reserve_impl for one segment of `bytes`: per-line, per-hit-line, per-tile arrays
      and the records: about 0.46 bytes per text byte of a segment), kept free while the candidates are taken */
   seeqdev_scan_t *sc = nullptr;
   const size_t seg = bytes < (size_t)0xF0000000u ? bytes : (size_t)0xF0000000u;
   const size_t headroom = seg / 2 + ((size_t)256 << 20);
   size_t peak = 0;
   for (int i = 0; i < candidates; i++) {
      size_t want = bytes;
      if (i > 0) {
         want = p2;
         size_t freeb = 0, total = 0;
         if (hipMemGetInfo(&freeb, &total) != hipSuccess) break;
         if (want + headroom > freeb) { want = bytes; if (want + headroom > freeb) break; }
      }
      if (hipMalloc(&blk[n], want) != hipSuccess) { (void)hipGetLastError(); blk[n] = nullptr; break; }
      blk_bytes[n] = want;
      peak += want;
      n++;
   }
   if (n == 0) { hip_fail(hipErrorOutOfMemory, "seeqdevTextAlloc", ENOMEM); return NULL; }
   int best = 0;
   if (n > 1) {
      static const char plain[] = "GATGTAGCGCGATTAGCCTG";
      char keys[20];
      for (int i = 0; i < 20; i++) keys[i] = plain[i] == 'A' ? 1 : plain[i] == 'C' ? 2 : plain[i] == 'G' ? 4 : 8;
      seeqdev_pattern_t *pat = seeqdevPatternNew(keys, 20, 3);
      /* Round 5: the launch time is a property of the PAIR (text buffer, scan context's workspace) -- the same text runs at 0.72 or 0.84 ms with
         two contexts of one process, reproducibly (profiles/r05/workspace_probe.txt) -- so a caller that scans the text with a context of its own
         (`scan`: reserve it first, so that its workspace is the one that stays) has the candidates probed with THAT context; NULL: a context made here. */
      sc = pat ? (scan ? scan : seeqdevScanNew(NULL)) : NULL;
      const bool own_sc = scan == nullptr;
      const bool prof_was = sc ? sc->prof : false;
      const uint64_t nreads = bytes / 151;
      bool ok = pat && sc && nreads > 0 && seeqdevScanSetProfiling(sc, 1) == 0;
      for (int i = 0; ok && i < n; i++) {
         ok = seeqdevSynthReads(blk[i], 0, nreads, 150, plain, 20, 3, 0x5EE92025ull, NULL) == 0 && hipStreamSynchronize(NULL) == hipSuccess;
         for (int rep = 0; ok && rep < 2; rep++) {
            seeqdev_counts_t cnt;
            ok = seeqdevScanRun(sc, pat, blk[i], (size_t)nreads * 151, SQ_BEST, SEEQDEV_WANT_COUNTLINES) == 0 && seeqdevScanFetch(sc, &cnt) == 0;
         }
         float t[4] = {0, 0, 0, 0};
         if (ok) ok = seeqdevScanLastTimes(sc, t) == 0;
         ms[i] = t[1];
      }
      if (sc && own_sc) seeqdevScanFree(sc);
      else if (sc) (void)seeqdevScanSetProfiling(sc, prof_was ? 1 : 0);
      if (pat) seeqdevPatternFree(pat);
      if (ok) {
         for (int i = 1; i < n; i++) if (ms[i] < ms[best]) best = i;
         if (info) { for (int i = 0; i < n; i++) info->probe_ms[i] = ms[i]; info->nprobed = n; }
      }                                                     /* (a failed probe: the plain allocation, nprobed = 0) */
      for (int i = 0; i < n; i++) if (i != best) (void)hipFree(blk[i]);
      seeqerr = 0;
   }
   if (info) { info->chosen = best; info->allocated_bytes = blk_bytes[best]; info->probe_peak_bytes = n > 1 ? peak + headroom : peak; }
   return blk[best];
}

extern "C" void *seeqdevTextAllocInfo(size_t bytes, int candidates, seeqdev_textinfo_t *info) { return seeqdevTextAllocFor(NULL, bytes, candidates, info); }

extern "C" void *seeqdevTextAlloc(size_t bytes, int candidates, float *probe_ms, int *nprobed)
{
   seeqdev_textinfo_t info;
   void *p = seeqdevTextAllocInfo(bytes, candidates, &info);
   if (nprobed) *nprobed = p ? info.nprobed : 0;
   if (p && probe_ms) for (int i = 0; i < info.nprobed; i++) probe_ms[i] = info.probe_ms[i];
   return p;
}

extern "C" void seeqdevTextFree(void *d_text)
{
   if (d_text) (void)hipFree(d_text);
}

extern "C" int seeqdevScanCopyOffsets(seeqdev_scan_t *s, uint64_t *host_out, size_t first, size_t n)
{
   seeqerr = 0;
   if (!s || (!host_out && n)) { errno = EINVAL; return -1; }
   if (first + n > s->counts.nrecords) { errno = EINVAL; return -1; }
   if (n == 0) return 0;
   if (use_device(s->device)) return -1;
   HIP_TRY(hipMemcpyAsync(host_out, s->rec_off + first, n * sizeof(uint64_t), hipMemcpyDeviceToHost, s->stream), EIO);
   HIP_TRY(hipStreamSynchronize(s->stream), EIO);
   return 0;
}

extern "C" int seeqdevScanHostBegin(seeqdev_scan_t *s, const seeqdev_pattern_t *pat, const char *host_text, size_t nbytes,
                                    int options, int want)
{
   seeqerr = 0;
   if (!s || !pat || (!host_text && nbytes)) { errno = EINVAL; return -1; }
   if (use_device(s->device)) return -1;
   if (nbytes > s->cap_text) {
      const size_t cap = nbytes + (nbytes >> 2) + 4096;
      if (ws_alloc((void **)&s->d_text, cap)) return -1;
      s->cap_text = cap;
   }
   if (s->prof && !s->have_h2d_ev) {
      HIP_TRY(hipEventCreate(&s->ev_h2d[0]), EIO);
      HIP_TRY(hipEventCreate(&s->ev_h2d[1]), EIO);
      s->have_h2d_ev = true;
   }
   if (s->prof) HIP_TRY(hipEventRecord(s->ev_h2d[0], s->stream), EIO);
   if (nbytes) HIP_TRY(hipMemcpyAsync(s->d_text, host_text, nbytes, hipMemcpyHostToDevice, s->stream), EIO);
   if (s->prof) HIP_TRY(hipEventRecord(s->ev_h2d[1], s->stream), EIO);
   /* the line-length sample (kernel selection) comes from the host copy: no round trip, nothing stale */
   if (s->line_hint <= 0 && !(options & SEEQDEV_SINGLELINE) && nbytes) {
      const size_t n = nbytes < SAMPLE_BYTES ? nbytes : SAMPLE_BYTES;
      size_t nl = 0;
      for (size_t i = 0; i < n; i++) nl += host_text[i] == '\n';
      s->avg_line = nl ? (double)n / (double)nl : 1e9;
      s->avg_text = s->d_text;
      s->avg_nbytes = nbytes;
   } else {
      s->avg_text = NULL;
   }
   return seeqdevScanRun(s, pat, s->d_text, nbytes, options, want);
}

extern "C" int seeqdevScanHost(seeqdev_scan_t *s, const seeqdev_pattern_t *pat, const char *host_text, size_t nbytes,
                               int options, int want, seeqdev_counts_t *counts)
{
   if (seeqdevScanHostBegin(s, pat, host_text, nbytes, options, want)) return -1;
   return seeqdevScanFetch(s, counts);
}

/* ========================================================================== */
/* Several patterns, one text (barcode demultiplexing: reference doc/response.tex:358-360)  */
/* ========================================================================== */
/* Several patterns over one text.  ONE walk for all of them when the set has a union automaton and the text is k_pair's
 * (read-length lines, SQ_FAIL / SQ_CONVERT; seeq_multi.h) -- else, and under SEEQ_MULTI=sequential, a scan per pattern over
 * the resident text, back to back on the context's stream.  Per pattern: counts, and for SEEQDEV_WANT_RECORDS its ordered
 * records, kept on the host until the next multi scan.  Either way the results are those of a scan of each pattern alone. */
static int multi_grow_host(seeqdev_scan_t *s, size_t n)
{
   if (s->multi_nrec + n > s->cap_multi_rec) {             /* page-locked: the records of a barcode set are hundreds of MB, and a pageable copy runs at a fifth of the link */
      const size_t cap = (s->multi_nrec + n) + ((s->multi_nrec + n) >> 1) + 1024;
      seeqdev_hit_t *g = nullptr;
      if (hipHostMalloc((void **)&g, cap * sizeof *g, hipHostMallocDefault) != hipSuccess || !g) { errno = ENOMEM; return -1; }
      if (s->multi_rec) {
         if (hipStreamSynchronize(s->stream) != hipSuccess) { (void)hipHostFree(g); errno = EIO; return -1; }      /* (copies into the old buffer may be in flight) */
         if (s->multi_nrec) memcpy(g, s->multi_rec, s->multi_nrec * sizeof *g);
         (void)hipHostFree(s->multi_rec);
      }
      s->multi_rec = g;
      s->cap_multi_rec = cap;
   }
   return 0;
}

/* 0: done; 1: not for this set / text / options (the caller scans pattern by pattern); -1: error */
static int multi_one_pass(seeqdev_scan_t *s, const seeqdev_pattern_t *const *pats, int npat, const void *d_text, size_t nbytes,
                          int options, int want, seeqdev_counts_t *counts)
{
   const char *env = getenv("SEEQ_MULTI");
   if (env && !strcmp(env, "sequential")) return 1;
   if (npat < 2 || npat > SEEQ_MULTI_MAX || nbytes == 0) return 1;
   const int nd = options & MASK_NONDNA;
   if ((options & (MASK_INPUT | SEEQDEV_SINGLELINE)) || !(nd == SQ_FAIL || nd == SQ_CONVERT)) return 1;
   for (int k = 0; k < npat; k++) if (!pats[k] || pats[k]->device != s->device) return 1;
   if (use_device(s->device)) return -1;
   MultiPlan *mp = multi_plan_for(&s->mplan, pats, npat);
   if (!mp) { errno = ENOMEM; return -1; }
   if (mp->state != 1) return 1;
   /* the patterns' EQ tables of the exact pass (as run_segments makes the one of a single pattern) */
   if (mp->eq_options != options) {
      uint32_t *h = (uint32_t *)calloc((size_t)npat * 1536, sizeof(uint32_t));
      if (!h) { errno = ENOMEM; return -1; }
      for (int k = 0; k < npat; k++) {
         const seeqdev_pattern *pat = pats[k];
         const int Wp = pat->words, fw = mp->fw[k];
         uint32_t *tab = h + (size_t)k * 1536;
         for (int dir = 0; dir < 2; dir++)
            for (int b = 0; b < 256; b++) {
               const uint8_t cls = sq_class_of((uint32_t)b, options);
               uint64_t v;
               if (cls < 5) {
                  const uint32_t *q = pat->h_peq + (dir * 5 + cls) * Wp;
                  const uint64_t col = (uint64_t)q[0] | (Wp > 1 ? (uint64_t)q[1] << 32 : 0);
                  v = col << (32 * fw - pat->wlen);
               } else {
                  v = cls == SQC_TERM ? FUSED_FLAG_TERM : FUSED_FLAG_SKIP;
               }
               uint32_t *dst = tab + (size_t)(dir * 256 + b) * fw;
               dst[0] = (uint32_t)v;
               if (fw == 2) dst[1] = (uint32_t)(v >> 32);
            }
      }
      const hipError_t e = hipMemcpy(mp->d_eq, h, (size_t)npat * 1536 * sizeof(uint32_t), hipMemcpyHostToDevice);
      free(h);
      if (e != hipSuccess) return hip_fail(e, "hipMemcpy(EQ tables)", EIO);
      mp->eq_options = options;
   }
   if (scan_setup(s, &mp->upat, d_text, nbytes, options, want, 2)) return -1;
   s->multi_active = true;
   int rc = -1;
   for (int attempt = 0; attempt < 8; attempt++) {
      if (multi_ws_ensure(s, npat)) break;
      if (hipMemsetAsync(s->d_mcnt, 0, SEEQ_MULTI_MAX * sizeof(Counters), s->stream) != hipSuccess) { errno = EIO; break; }
      const int r = dispatch_run(s);
      if (r == -2) { rc = 1; break; }
      if (r) break;
      if (hipMemcpyAsync(s->h_mcnt, s->d_mcnt, (size_t)npat * sizeof(Counters), hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
          hipStreamSynchronize(s->stream) != hipSuccess) { errno = EIO; break; }
      const Counters u = *s->h_cnt;
      uint32_t povf = 0, need_hl = 0;
      uint64_t need_rec = 0;
      for (int k = 0; k < npat; k++) {
         povf |= s->h_mcnt[k].overflow;
         if (s->h_mcnt[k].need_hitlines > need_hl) need_hl = s->h_mcnt[k].need_hitlines;
         if (s->h_mcnt[k].need_records > need_rec) need_rec = s->h_mcnt[k].need_records;
      }
      if (u.overflow & 64u) { snprintf(g_last_error, sizeof g_last_error, "internal inconsistency in the hit list (k_stream_bounds)"); errno = EIO; break; }
      if (u.overflow & (8u | 16u | 32u)) { rc = 1; break; }          /* not k_pair's text after all: a scan per pattern */
      if (!u.overflow && !povf) {
         /* results: counts, then every pattern's records from its region */
         const uint64_t capR = s->cap_records / (uint64_t)npat;
         s->multi_nrec = 0;
         rc = 0;
         if (want == SEEQDEV_WANT_RECORDS) {
            size_t total = 0;
            for (int k = 0; k < npat; k++) total += (size_t)s->h_mcnt[k].records;
            if (multi_grow_host(s, total)) { rc = -1; break; }
         }
         for (int k = 0; k < npat && rc == 0; k++) {
            const Counters &h = s->h_mcnt[k];
            seeqdev_counts_t &o = s->multi_cnt[k];
            o.nlines = h.lines; o.nmatchlines = h.matchlines; o.nhits = h.hits; o.nrecords = h.records; o.nheaders = h.headers;
            s->multi_first[k] = s->multi_nrec;
            const size_t n = want == SEEQDEV_WANT_RECORDS ? (size_t)h.records : 0;
            if (n) {
               if (multi_grow_host(s, n)) { rc = -1; break; }
               if (hipMemcpyAsync(s->multi_rec + s->multi_nrec, s->records + (uint64_t)k * capR, n * sizeof(seeqdev_hit_t), hipMemcpyDeviceToHost, s->stream) != hipSuccess) { errno = EIO; rc = -1; break; }
               s->multi_nrec += n;
            }
            if (counts) counts[k] = o;
         }
         if (rc == 0 && hipStreamSynchronize(s->stream) != hipSuccess) { errno = EIO; rc = -1; }
         if (rc == 0) { s->multi_first[npat] = s->multi_nrec; s->multi_n = npat; s->last_multi = 1; }
         break;
      }
      size_t nl = s->cap_lines, nhl = s->cap_hitlines, nrec = s->cap_records;
      if (u.overflow & 1u) nl = (size_t)u.need_lines + (u.need_lines >> 3) + 64;
      if (u.overflow & 2u) nhl = (size_t)u.need_hitlines + (u.need_hitlines >> 3) + 64;
      if (u.overflow & 128u) { s->no_window = true; s->fallback_ttl = 32; }
      if (povf & 2u) { const size_t w = ((size_t)need_hl + (need_hl >> 3) + 64) * (size_t)npat; if (w > nhl) nhl = w; }
      if (povf & 4u) { const size_t w = ((size_t)need_rec + (size_t)(need_rec >> 3) + 64) * (size_t)npat; if (w > nrec) nrec = w; }
      if ((u.overflow & 1u) && nhl < nl / 2) nhl = nl / 2 + 64;
      if (attempt == 7) { snprintf(g_last_error, sizeof g_last_error, "workspace did not converge"); errno = ENOMEM; break; }
      if (reserve_impl(s, s->nbytes, nl, nhl, nrec)) break;
   }
   s->multi_active = false;
   s->ran = false;                                         /* (seeqdevScanFetch has nothing to fetch: the multi scan is complete) */
   return rc;
}

extern "C" int seeqdevScanRunMulti(seeqdev_scan_t *s, const seeqdev_pattern_t *const *pats, int npat, const void *d_text, size_t nbytes,
                                   int options, int want, seeqdev_counts_t *counts)
{
   seeqerr = 0;
   if (!s || !pats || npat < 1 || (!d_text && nbytes) || want < 0 || want > 2) { errno = EINVAL; return -1; }
   if (npat > s->cap_multi_n) {
      seeqdev_counts_t *c = (seeqdev_counts_t *)realloc(s->multi_cnt, (size_t)npat * sizeof *c);
      if (c) s->multi_cnt = c;
      size_t *f = (size_t *)realloc(s->multi_first, ((size_t)npat + 1) * sizeof *f);
      if (f) s->multi_first = f;
      if (!c || !f) { errno = ENOMEM; return -1; }
      s->cap_multi_n = npat;
   }
   s->multi_n = 0;
   s->multi_nrec = 0;
   s->last_multi = 0;
   {
      const int r = multi_one_pass(s, pats, npat, d_text, nbytes, options, want, counts);
      if (r <= 0) return r;
   }
   s->multi_nrec = 0;
   for (int k = 0; k < npat; k++) {
      if (seeqdevScanRun(s, pats[k], d_text, nbytes, options, want)) return -1;
      if (seeqdevScanFetch(s, &s->multi_cnt[k])) return -1;
      s->multi_first[k] = s->multi_nrec;
      const size_t n = want == SEEQDEV_WANT_RECORDS ? (size_t)s->multi_cnt[k].nrecords : 0;
      if (n) {
         if (multi_grow_host(s, n)) return -1;
         if (seeqdevScanCopyRecords(s, s->multi_rec + s->multi_nrec, 0, n)) return -1;
         s->multi_nrec += n;
      }
      if (counts) counts[k] = s->multi_cnt[k];
   }
   s->multi_first[npat] = s->multi_nrec;
   s->multi_n = npat;
   return 0;
}

/* 1: the last multi scan walked the text once for all its patterns; 0: a scan per pattern. */
extern "C" int seeqdevScanLastMulti(const seeqdev_scan_t *s) { return s ? s->last_multi : 0; }

extern "C" int seeqdevScanHostMulti(seeqdev_scan_t *s, const seeqdev_pattern_t *const *pats, int npat, const char *host_text, size_t nbytes,
                                    int options, int want, seeqdev_counts_t *counts)
{
   seeqerr = 0;
   if (!s || !pats || npat < 1 || (!host_text && nbytes)) { errno = EINVAL; return -1; }
   if (use_device(s->device)) return -1;
   if (nbytes > s->cap_text) {
      const size_t cap = nbytes + (nbytes >> 2) + 4096;
      if (ws_alloc((void **)&s->d_text, cap)) return -1;
      s->cap_text = cap;
   }
   if (nbytes) HIP_TRY(hipMemcpyAsync(s->d_text, host_text, nbytes, hipMemcpyHostToDevice, s->stream), EIO);   /* once, for all patterns */
   s->avg_text = NULL;                                     /* new contents behind the same pointer: sample again */
   return seeqdevScanRunMulti(s, pats, npat, s->d_text, nbytes, options, want, counts);
}

extern "C" int seeqdevScanMultiRecords(const seeqdev_scan_t *s, int k, const seeqdev_hit_t **rec, size_t *nrec)
{
   if (!s || !rec || !nrec || k < 0 || k >= s->multi_n) { errno = EINVAL; return -1; }
   *rec = s->multi_rec + s->multi_first[k];
   *nrec = s->multi_first[k + 1] - s->multi_first[k];
   return 0;
}

extern "C" int seeqdevScanLastCopyMs(const seeqdev_scan_t *s, float *h2d_ms)
{
   if (!s || !h2d_ms) { errno = EINVAL; return -1; }
   *h2d_ms = s->h2d_ms;
   return 0;
}

extern "C" int seeqdevScanLastLaunches(const seeqdev_scan_t *s) { return s ? (int)s->prof_segs : 0; }

extern "C" float seeqdevScanLastClockMHz(const seeqdev_scan_t *s) { return s ? s->clk_mhz : 0.f; }

extern "C" int seeqdevScanLastLaunchTimes(const seeqdev_scan_t *s, float *ms, int cap)
{
   if (!s || (!ms && cap > 0)) { errno = EINVAL; return -1; }
   const size_t n = s->prof_segs < s->cap_launch_ms ? s->prof_segs : s->cap_launch_ms;
   for (size_t i = 0; i < n && (int)i < cap; i++) ms[i] = s->launch_ms[i];
   return (int)n;
}

/* ========================================================================== */
/* One string, one launch: seeqStringMatch (reference libseeq.c:171-352)        */
/* ========================================================================== */
/* The per-string entry point is what the reference's Python module calls for every string (seeqmodule.c:858).  One
 * workgroup: all threads stage the string (read over the link from page-locked host memory when it is short, else
 * from HBM) and the tables into LDS; then the string's positions are shared out over the 256 threads: every thread
 * computes the capped scores of its positions from a fresh column started m + tau + 1 characters earlier (exact from
 * there on, as in k_stream), applies the acceptance rules -- which only look at the scores of a position and the two
 * before it (libseeq.c:277-331: emit = stop ? !latch : zero, latch = stop ? 1 : zero) -- and the emissions are compacted
 * in order (block scan) / reduced (first, best), starts recovered by their threads, count + records written straight
 * into page-locked host memory.  A single lane walking the string took 0.17 us per character (41 us per call at 150
 * characters, 15 of them launch + synchronisation).  Strings with skipped bytes (SQ_IGNORE, SQ_STREAM) or longer than
 * STRING_PAR_MAX keep the one-lane scan.  One launch, one stream synchronisation, no device allocation.  Long strings
 * in line mode take the batched scan instead (libseeq_api.c). */
static constexpr uint32_t STRING_LDS_MAX = 48u * 1024;      /* strings up to this are staged in LDS */
static constexpr uint32_t STRING_ZC_MAX = 4096;             /* ... and up to this read straight from host memory */
static constexpr uint32_t STRING_PAR_MAX = 32768;           /* ... and up to this scanned by all threads (positions fit 16 bits; longer strings in line mode take the batched scan) */

template <int W>
__global__ __launch_bounds__(WG) void k_string(const uint8_t *text, uint32_t n, const uint32_t *peq, int m, int tau, int options,
                                               uint32_t *out, uint32_t cap, uint32_t seq)
{
   extern __shared__ __align__(16) uint8_t s_text[];       /* n + 16 bytes when staged */
   __shared__ uint32_t s_peq[10 * W];
   __shared__ uint8_t s_lut[256];
   const int Wp = (m + 31) >> 5;
   for (int i = threadIdx.x; i < 10 * W; i += WG) {
      const int dir = i / (5 * W), rem = i % (5 * W), cls = rem / W, w = rem % W;
      s_peq[i] = w < Wp ? peq[(dir * 5 + cls) * Wp + w] : 0u;
   }
   for (int b = threadIdx.x; b < 256; b += WG) s_lut[b] = sq_class_of((uint32_t)b, options);
   const bool staged = n <= STRING_LDS_MAX;
   if (staged) {
      /* 16 bytes per thread and round, all loads of a round in flight together (the link's latency is paid once per round) */
      for (uint32_t o = threadIdx.x * 16; o < n; o += WG * 16) {
         const sq_chunk16_t c = sq_load16(text, o, n);
         *reinterpret_cast<uint4 *>(s_text + o) = make_uint4(c.w[0], c.w[1], c.w[2], c.w[3]);
      }
   }
   __shared__ uint32_t s_first, s_skip, s_key, s_wave[WG / 64];
   const bool par = n <= STRING_PAR_MAX;                  /* (=> staged) */
   if (threadIdx.x == 0) { s_first = n; s_skip = 0; s_key = 0xFFFFFFFFu; }
   __syncthreads();
   if (par) {
      /* the line ends at its first terminator (or at n: bytes beyond read as NUL); a skipped byte before it -> one lane */
      for (uint32_t j = threadIdx.x; j < n; j += WG) if (s_lut[s_text[j]] == SQC_TERM) atomicMin(&s_first, j);
      __syncthreads();
      const uint32_t len = s_first;
      for (uint32_t j = threadIdx.x; j < len; j += WG) if (s_lut[s_text[j]] == SQC_SKIP) s_skip = 1u;
      __syncthreads();
      if (!s_skip) {
         const int match_opt = options & 3;
         const uint32_t *peq_f = s_peq, *peq_r = s_peq + 5 * W;
         uint16_t *ed = reinterpret_cast<uint16_t *>(s_text + (((size_t)n + 31) & ~(size_t)15));   /* per position: emitted distance + 1, or 0 */
         const uint32_t P = len + 1;                      /* positions 0..len; the last one is the terminator's step */
         const uint32_t B = (P + WG - 1) / WG;
         const uint32_t j0 = threadIdx.x * B, j1 = j0 + B < P ? j0 + B : P;
         const uint32_t cnt = j0 < P ? sq_emit_window<W>((const uint8_t *)s_text, len, j0, j1, peq_f, (const uint8_t *)s_lut, m, tau, ed) : 0u;
         uint32_t total = 0, nh_par = 0;
         const uint32_t excl = block_excl_scan(cnt, &total, s_wave);      /* (also orders the ed[] writes: barrier inside) */
         sq_hit_t *rec = reinterpret_cast<sq_hit_t *>(out + 4);
         if (match_opt == SQK_ALL) {
            uint32_t idx = excl;
            for (uint32_t j = j0; j < j1 && cnt; j++) {
               if (!ed[j]) continue;
               if (idx < cap) {
                  sq_hit_t h;
                  h.line = 1;
                  h.start = sq_reverse_start<W>((const uint8_t *)s_text, j, (int)ed[j] - 1, peq_r, (const uint8_t *)s_lut, m, tau);
                  h.end = j;
                  h.dist = (uint32_t)ed[j] - 1u;
                  rec[idx] = h;
               }
               idx++;
            }
            nh_par = total;
         } else {
            /* SQ_BEST: smallest distance, first position; SQ_FIRST / SQ_COUNT: first position */
            for (uint32_t j = j0; j < j1 && cnt; j++)
               if (ed[j]) { atomicMin(&s_key, (match_opt == SQK_BEST ? ((uint32_t)ed[j] - 1u) << 16 : 0u) | j); if (match_opt != SQK_BEST) break; }
            __syncthreads();
            const uint32_t key = s_key;
            if (key != 0xFFFFFFFFu) {
               const uint32_t j = key & 0xFFFFu;
               if (j >= j0 && j < j1 && cap) {
                  sq_hit_t h;
                  h.line = 1;
                  h.start = sq_reverse_start<W>((const uint8_t *)s_text, j, (int)ed[j] - 1, peq_r, (const uint8_t *)s_lut, m, tau);
                  h.end = j;
                  h.dist = (uint32_t)ed[j] - 1u;
                  rec[0] = h;
               }
            }
            nh_par = key != 0xFFFFFFFFu ? 1u : 0u;
         }
         /* records first (every writer fences), then the count, then the ticket the host spins on */
         __threadfence_system();
         __syncthreads();
         if (threadIdx.x == 0) {
            out[0] = nh_par;
            __threadfence_system();
            __hip_atomic_store(&out[1], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
         }
         return;
      }
   }
   if (threadIdx.x != 0) return;
   const uint8_t *tp = staged ? (const uint8_t *)s_text : text;
   const uint32_t nh = sq_scan_line<W, SQ_MODE_EMIT>(tp, (uint64_t)n, 0, (const uint32_t *)s_peq, (const uint32_t *)(s_peq + 5 * W),
                                                     (const uint8_t *)s_lut, m, tau, options & 3, 1,
                                                     reinterpret_cast<sq_hit_t *>(out + 4), cap);
   out[0] = nh;
   __threadfence_system();
   __hip_atomic_store(&out[1], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int W>
static void launch_string(seeqdev_scan *s, const seeqdev_pattern *pat, const uint8_t *text, uint32_t n, int options, uint32_t cap, uint32_t seq)
{
   size_t lds = n <= STRING_LDS_MAX ? (((size_t)n + 31) & ~(size_t)15) : 0;
   if (n <= STRING_PAR_MAX) lds += (2 * ((size_t)n + 2) + 15) & ~(size_t)15;        /* + per-position emissions */
   if (lds > 48u * 1024)                                    /* beyond the default limit of dynamic LDS per workgroup (set per device: every time) */
      (void)hipFuncSetAttribute((const void *)k_string<W>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
   hipLaunchKernelGGL(k_string<W>, dim3(1), dim3(WG), lds, s->stream, text, n, (const uint32_t *)pat->d_peq, pat->wlen, pat->tau,
                      options, s->h_strout, cap, seq);
}

/* data[0..n): the string (no NUL needed; a NUL inside ends it as in the reference).  On return *rec points at the
 * hit records (left to right; context-owned page-locked memory, valid until the next call) and *nrec is their number. */
extern "C" int seeqdevStringMatch(seeqdev_scan_t *s, const seeqdev_pattern_t *pat, const char *data, size_t n, int options,
                                  const seeqdev_hit_t **rec, size_t *nrec)
{
   seeqerr = 0;
   if (!s || !pat || (!data && n) || !rec || !nrec) { errno = EINVAL; return -1; }
   if (n > 0xFFFF0000ull) { errno = E2BIG; return -1; }
   if (use_device(s->device)) return -1;
   if (!s->h_strout) {
      s->cap_strout = 256;                                  /* records */
      HIP_TRY(hipHostMalloc((void **)&s->h_strout, 16 + s->cap_strout * sizeof(seeqdev_hit_t), hipHostMallocCoherent), ENOMEM);
   }
   if (!s->h_str) {
      s->cap_str = STRING_ZC_MAX + 16;
      HIP_TRY(hipHostMalloc((void **)&s->h_str, s->cap_str, hipHostMallocDefault), ENOMEM);
   }
   const uint8_t *dtext;
   if (n <= STRING_ZC_MAX) {
      memcpy(s->h_str, data, n);
      dtext = s->h_str;                                     /* page-locked host memory is device-visible at the same address */
   } else {
      if (n > s->cap_text) {
         const size_t cap = n + (n >> 2) + 4096;
         if (ws_alloc((void **)&s->d_text, cap)) return -1;
         s->cap_text = cap;
      }
      HIP_TRY(hipMemcpyAsync(s->d_text, data, n, hipMemcpyHostToDevice, s->stream), EIO);
      s->avg_text = NULL;
      dtext = s->d_text;
   }
   const int W = pat->words;
   for (int attempt = 0; attempt < 2; attempt++) {
      const uint32_t cap = (uint32_t)s->cap_strout;
      s->h_strout[0] = 0;
      const uint32_t seq = ++s->str_seq ? s->str_seq : ++s->str_seq;      /* never 0 */
      volatile uint32_t *ticket = s->h_strout + 1;
      *ticket = 0;
      if (W <= 1) launch_string<1>(s, pat, dtext, (uint32_t)n, options, cap, seq);
      else if (W <= 2) launch_string<2>(s, pat, dtext, (uint32_t)n, options, cap, seq);
      else if (W <= 4) launch_string<4>(s, pat, dtext, (uint32_t)n, options, cap, seq);
      else if (W <= 8) launch_string<8>(s, pat, dtext, (uint32_t)n, options, cap, seq);
      else launch_string<16>(s, pat, dtext, (uint32_t)n, options, cap, seq);
      HIP_TRY(hipGetLastError(), EIO);
      /* The kernel's last store is its ticket, into fine-grained page-locked memory: spinning on it is shorter than the
         runtime's completion path (hipStreamSynchronize: ~8 us).  After 200 us (long strings, a failed launch) the runtime
         takes over. */
      {
         struct timespec t0, t1;
         clock_gettime(CLOCK_MONOTONIC, &t0);
         unsigned spins = 0;
         while (__atomic_load_n(ticket, __ATOMIC_ACQUIRE) != seq) {
            if ((++spins & 63u) == 0) {
               clock_gettime(CLOCK_MONOTONIC, &t1);
               if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > 200000L) break;
            }
         }
         if (__atomic_load_n(ticket, __ATOMIC_ACQUIRE) != seq) HIP_TRY(hipStreamSynchronize(s->stream), EIO);
      }
      const uint32_t nh = s->h_strout[0];
      if (nh <= cap) {
         *rec = reinterpret_cast<const seeqdev_hit_t *>(s->h_strout + 4);
         *nrec = nh;
         return 0;
      }
      /* SQ_ALL with more hits than the record buffer holds: grow it and scan again */
      (void)hipHostFree(s->h_strout);
      s->h_strout = NULL;
      s->cap_strout = (size_t)nh + (nh >> 2) + 64;
      HIP_TRY(hipHostMalloc((void **)&s->h_strout, 16 + s->cap_strout * sizeof(seeqdev_hit_t), hipHostMallocCoherent), ENOMEM);
   }
   errno = EIO;
   return -1;
}
