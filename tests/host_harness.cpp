// tests/host_harness.cpp -- compiles the per-line device functions
// (seeq_amd/csrc/seeq_kernel_core.h) with g++ so that the CPU-only test suite
// can fuzz the Myers + acceptance + reverse-scan logic against the oracle
// before any GPU time is spent.  Test infrastructure: never linked into
// libseeq_amd.so.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../seeq_amd/csrc/seeq_kernel_core.h"
extern "C" {
#include "../seeq_amd/csrc/seeq_pattern.h"
}

namespace {

template <int W>
long run(const uint8_t *text, size_t n, const char *keys, int m, int tau, int options, int mode, uint32_t *out,
         size_t cap)
{
   std::vector<uint32_t> pf(5 * W), pr(5 * W);
   std::vector<char> rkeys(m);
   for (int i = 0; i < m; i++) rkeys[i] = keys[m - 1 - i];
   seeq_build_peq(keys, m, W, pf.data());
   seeq_build_peq(rkeys.data(), m, W, pr.data());
   uint8_t lut[256];
   for (int b = 0; b < 256; b++) lut[b] = sq_class_of((uint32_t)b, options);
   const uint32_t *f = pf.data(), *r = pr.data();
   const uint8_t *l = lut;
   std::vector<sq_hit_t> hits(cap ? cap : 1);
   uint32_t nh;
   if (mode == SQ_MODE_ANY)
      nh = sq_scan_line<W, SQ_MODE_ANY>(text, n, 0, f, r, l, m, tau, options & 3, 1, hits.data(), (uint32_t)cap);
   else if (mode == SQ_MODE_COUNT)
      nh = sq_scan_line<W, SQ_MODE_COUNT>(text, n, 0, f, r, l, m, tau, options & 3, 1, hits.data(), (uint32_t)cap);
   else
      nh = sq_scan_line<W, SQ_MODE_EMIT>(text, n, 0, f, r, l, m, tau, options & 3, 1, hits.data(), (uint32_t)cap);
   if (mode == SQ_MODE_EMIT)
      for (uint32_t k = 0; k < nh && k < cap; k++) {
         out[3 * k + 0] = hits[k].start;
         out[3 * k + 1] = hits[k].end;
         out[3 * k + 2] = hits[k].dist;
      }
   return (long)nh;
}

}  // namespace

// Scans text[0..n) as ONE line (seeqStringMatch semantics).  Hits come back
// left to right as (start,end,dist) triples.  wforce > 0 forces the word
// count (to exercise the multi-word carry chain on short patterns).
extern "C" long harness_scan(const uint8_t *text, size_t n, const char *keys, int m, int tau, int options, int mode,
                             int wforce, uint32_t *out, size_t cap)
{
   int W = wforce > 0 ? wforce : seeq_words_for(m);
   if (W <= 1) return run<1>(text, n, keys, m, tau, options, mode, out, cap);
   if (W <= 2) return run<2>(text, n, keys, m, tau, options, mode, out, cap);
   if (W <= 4) return run<4>(text, n, keys, m, tau, options, mode, out, cap);
   if (W <= 8) return run<8>(text, n, keys, m, tau, options, mode, out, cap);
   return run<16>(text, n, keys, m, tau, options, mode, out, cap);
}

// seeqStringMatch as k_string computes it (seeq_device.hip): the positions of the line shared out in blocks of `block`
// (what the 256 threads of the workgroup get), each block from a fresh column through sq_emit_window, emissions then
// taken in order (SQ_ALL), the first (SQ_FIRST) or the first with the smallest distance (SQ_BEST), starts by
// sq_reverse_start.  Returns the number of hits, or -1 for a line the kernel leaves to its one-lane scan (a skipped
// byte before the terminator).
namespace {
template <int W>
long run_par(const uint8_t *text, size_t n, const char *keys, int m, int tau, int options, int block, uint32_t *out, size_t cap)
{
   std::vector<uint32_t> pf(5 * W), pr(5 * W);
   std::vector<char> rkeys(m);
   for (int i = 0; i < m; i++) rkeys[i] = keys[m - 1 - i];
   seeq_build_peq(keys, m, W, pf.data());
   seeq_build_peq(rkeys.data(), m, W, pr.data());
   uint8_t lut[256];
   for (int b = 0; b < 256; b++) lut[b] = sq_class_of((uint32_t)b, options);
   uint32_t len = (uint32_t)n;
   for (uint32_t j = 0; j < n; j++) if (lut[text[j]] == SQC_TERM) { len = j; break; }
   for (uint32_t j = 0; j < len; j++) if (lut[text[j]] == SQC_SKIP) return -1;
   std::vector<uint16_t> ed(len + 2, 0);
   const uint32_t P = len + 1;
   const uint32_t *f = pf.data(), *r = pr.data();
   const uint8_t *l = lut;
   for (uint32_t j0 = 0; j0 < P; j0 += (uint32_t)block)
      sq_emit_window<W>(text, len, j0, j0 + (uint32_t)block < P ? j0 + (uint32_t)block : P, f, l, m, tau, ed.data());
   const int mo = options & 3;
   long nh = 0;
   long best = -1;
   for (uint32_t j = 0; j < P; j++) {
      if (!ed[j]) continue;
      if (mo == SQK_ALL) {
         if ((size_t)nh < cap) {
            out[3 * nh + 0] = sq_reverse_start<W>(text, j, (int)ed[j] - 1, r, l, m, tau);
            out[3 * nh + 1] = j;
            out[3 * nh + 2] = (uint32_t)ed[j] - 1u;
         }
         nh++;
      } else if (mo == SQK_BEST) {
         if (best < 0 || ed[j] < ed[best]) best = j;
      } else { best = j; break; }
   }
   if (mo != SQK_ALL && best >= 0) {
      if (cap) {
         out[0] = sq_reverse_start<W>(text, (uint32_t)best, (int)ed[best] - 1, r, l, m, tau);
         out[1] = (uint32_t)best;
         out[2] = (uint32_t)ed[best] - 1u;
      }
      nh = 1;
   }
   return nh;
}
}  // namespace

extern "C" long harness_string_par(const uint8_t *text, size_t n, const char *keys, int m, int tau, int options, int block,
                                   int wforce, uint32_t *out, size_t cap)
{
   int W = wforce > 0 ? wforce : seeq_words_for(m);
   if (W <= 1) return run_par<1>(text, n, keys, m, tau, options, block, out, cap);
   if (W <= 2) return run_par<2>(text, n, keys, m, tau, options, block, out, cap);
   if (W <= 4) return run_par<4>(text, n, keys, m, tau, options, block, out, cap);
   if (W <= 8) return run_par<8>(text, n, keys, m, tau, options, block, out, cap);
   return run_par<16>(text, n, keys, m, tau, options, block, out, cap);
}

extern "C" int harness_compile(const char *expr, char *keys, int *err) { return seeq_compile_pattern(expr, keys, err); }

// ---- the streaming automaton of k_stream (seeq_amd/csrc/seeq_dfa.h: host-side table builder) ----
// Emulates the kernel's decomposition on the host: every `chunk`-byte chunk of the text is walked on its own from the
// root state after a warm-up over the `warm` bytes before it ('\n' where the buffer starts), with the kernel's
// addressing "state ^ (byte & 0xE)"; positions where the walk enters ACC_NEW inside the owned chunk are reported.
// Returns the number of events (positions in out[], ascending), -1 when the automaton does not fit; *nstates = its size.
#include "../seeq_amd/csrc/seeq_dfa.h"
extern "C" long harness_dfa_stream(const uint8_t *text, size_t n, const char *keys, int m, int tau, int chunk, int warm,
                                   uint64_t *out, size_t cap, uint32_t *nstates)
{
   seeq_dfa_t *d = seeq_dfa_build_stream(keys, m, tau);
   if (!d) return -1;
   if (nstates) *nstates = d->nstates;
   size_t ne = 0;
   for (size_t c0 = 0; c0 < n; c0 += (size_t)chunk) {
      uint32_t state = 0;
      for (long long p = (long long)c0 - warm; p < (long long)c0 + chunk && p < (long long)n; p++) {
         const uint8_t b = p < 0 ? (uint8_t)'\n' : text[p];
         state = d->table[(state ^ (uint32_t)(b & 0xE)) >> 1];
         if (p >= (long long)c0 && state == d->acc_final) {
            if (ne < cap) out[ne] = (uint64_t)p;
            ne++;
         }
      }
   }
   seeq_dfa_free(d);
   return (long)ne;
}

// The same walk over what seeq_dfa_plan_stream() (parts == 0) or seeq_dfa_build_filter(parts >= 2) builds: a partition
// FILTER automaton flags candidates -- every line with a hit must get at least one event, lines without may get some.
// info[0..3] = states, parts, warm-up bytes, accept rate * 1e9.  Returns -1 when nothing fits.
extern "C" long harness_dfa_filter(const uint8_t *text, size_t n, const char *keys, int m, int tau, int parts, int chunk,
                                   uint64_t *out, size_t cap, uint32_t *info)
{
   seeq_dfa_t *d = parts >= 2 ? seeq_dfa_build_filter(keys, m, tau, parts) : seeq_dfa_plan_stream(keys, m, tau, 0);
   if (!d) return -1;
   if (info) { info[0] = d->nstates; info[1] = (uint32_t)d->nparts; info[2] = (uint32_t)d->warm; info[3] = (uint32_t)(d->p_accept * 1e9); }
   const int warm = d->warm;
   size_t ne = 0;
   for (size_t c0 = 0; c0 < n; c0 += (size_t)chunk) {
      uint32_t state = 0;
      for (long long p = (long long)c0 - warm; p < (long long)c0 + chunk && p < (long long)n; p++) {
         const uint8_t b = p < 0 ? (uint8_t)'\n' : text[p];
         state = d->table[(state ^ (uint32_t)(b & 0xE)) >> 1];
         if (p >= (long long)c0 && state == d->acc_final) {
            if (ne < cap) out[ne] = (uint64_t)p;
            ne++;
         }
      }
   }
   seeq_dfa_free(d);
   return (long)ne;
}

// The long-line filter's RESTART table (seeq_dfa_restart_variant; k_stream<.., LL> with FusedArgs.ll_filter == 2): the kernel's decomposition
// on the host -- every `chain`-byte chain (64 in the kernel) is walked on its own from the root after a warm-up over the `warm_bytes` bytes before
// it; a step that lands on ACC_NEW inside the chain reports its position, and a chain that landed on it inside its warm-up window reports its own
// first byte (it is out of step with the walk the line's start would have made).  Returns the number of events, -1 when no filter fits.
extern "C" long harness_dfa_filter_restart(const uint8_t *text, size_t n, const char *keys, int m, int tau, int parts, int chain, int warm_bytes,
                                           uint64_t *out, size_t cap, uint32_t *info)
{
   seeq_dfa_t *d = parts >= 2 ? seeq_dfa_build_filter(keys, m, tau, parts) : seeq_dfa_plan_stream(keys, m, tau, 0);
   if (!d) return -1;
   if (d->nparts < 2) { seeq_dfa_free(d); return -1; }
   uint16_t *t = seeq_dfa_restart_variant(d);
   if (!t) { seeq_dfa_free(d); return -1; }
   if (info) { info[0] = d->nstates; info[1] = (uint32_t)d->nparts; info[2] = (uint32_t)d->warm; info[3] = (uint32_t)(d->p_accept * 1e9); }
   const int warm = warm_bytes > 0 ? warm_bytes : ((d->warm + 3) & ~3);
   size_t ne = 0;
   for (size_t c0 = 0; c0 < n; c0 += (size_t)chain) {
      uint32_t state = 0;
      bool seen = false;
      for (long long p = (long long)c0 - warm; p < (long long)c0 + chain && p < (long long)n; p++) {
         const uint8_t b = p < 0 ? (uint8_t)'\n' : text[p];
         state = t[(state ^ (uint32_t)(b & 0xE)) >> 1];
         const bool acc = state == d->acc_final;
         if (p < (long long)c0) { seen = seen || acc; continue; }
         if (acc || (seen && p == (long long)c0)) {
            if (ne < cap) out[ne] = (uint64_t)p;
            ne++;
         }
      }
   }
   free(t);
   seeq_dfa_free(d);
   return (long)ne;
}

// ---- the pair automaton of k_pair (seeq_dfa.h section 3, seeq_pair.h) ----
// Emulates the kernel's decomposition on the host: every `chain`-byte chain of the text (64 in the kernel) is walked on
// its own from the root state, two bytes per step (2-bit codes = bits 1-2 of the byte), after a warm-up over the
// `warm_bytes` bytes before it ('\n' where the buffer starts or has ended; 0 = the automaton's own warm-up rounded up to
// whole words).  A step that lands on a flagged row inside the owned chain reports the position of the pair's second
// byte; a flagged row met during the warm-up reports the chain's second byte (the made-up candidate of seeq_pair.h).
// info[0..5] = states, states before minimisation, prefix length, warm-up bytes, accept rate * 1e9, parts.
// Returns the number of events (positions in out[], ascending), -1 when no pair automaton fits.
extern "C" long harness_pair_walk(const uint8_t *text, size_t n, const char *keys, int m, int tau, int chain, int warm_bytes,
                                  uint64_t *out, size_t cap, uint32_t *info)
{
   seeq_pair_t *d = seeq_pair_plan(keys, m, tau);
   if (!d) return -1;
   if (info) { info[0] = d->nstates; info[1] = d->nstates_raw; info[2] = (uint32_t)d->mp; info[3] = (uint32_t)d->warm; info[4] = (uint32_t)(d->p_accept * 1e9); info[5] = (uint32_t)d->nparts; }
   const long long W = warm_bytes > 0 ? warm_bytes : 4 * ((d->warm + 3) / 4);
   size_t ne = 0;
   for (size_t c0 = 0; c0 < n; c0 += (size_t)chain) {
      uint32_t state = 0;
      bool warm_flag = false;
      for (long long p = (long long)c0 - W; p < (long long)c0 + chain && p < (long long)n; p += 2) {
         const uint8_t b1 = p < 0 ? (uint8_t)'\n' : text[p];
         const uint8_t b2 = (p + 1 < 0 || p + 1 >= (long long)n) ? (uint8_t)'\n' : text[p + 1];
         uint16_t nxt;
         memcpy(&nxt, d->table + state + 2u * (uint32_t)(((b1 >> 1) & 3) * 4 + ((b2 >> 1) & 3)), 2);      /* (the kernel: state ^ index -- the same, index bits 1-4 are free in a state value) */
         state = nxt;
         const bool flagged = (state & 1u) != 0;
         if (p < (long long)c0) { warm_flag |= flagged; continue; }
         const bool report = flagged || (p == (long long)c0 && warm_flag);
         if (report) {
            const uint64_t pos = (uint64_t)(p + 1 < (long long)n ? p + 1 : (long long)n - 1);
            if (ne < cap) out[ne] = pos;
            ne++;
         }
      }
   }
   seeq_pair_free(d);
   return (long)ne;
}

// ---- the quad automaton of the packed walk (seeq_dfa.h section 3b, seeq_packed.h) ----
// Emulates k_packed_walk<true> on the host: every line of the text is a read, walked from the root from its first base, four bases
// per table step (2-bit codes = bits 1-2 of the byte, packed first base high as seeqdev_packed_t stores them; the last step of a
// line is padded with code 0 and the flags of the padding are dropped).  Reports the absolute position of every base the walk
// accepted on.  info[0..4] = states, states before minimisation, positions carried, parts, accept rate * 1e9.
// Returns the number of events, -1 when no quad automaton fits.
extern "C" long harness_quad_walk(const uint8_t *text, size_t n, const char *keys, int m, int tau, uint64_t *out, size_t cap, uint32_t *info)
{
   seeq_quad_t *d = seeq_quad_plan(keys, m, tau);
   if (!d) return -1;
   if (info) { info[0] = d->nstates; info[1] = d->nstates_raw; info[2] = (uint32_t)d->mp; info[3] = (uint32_t)d->nparts; info[4] = (uint32_t)(d->p_accept * 1e9); }
   size_t ne = 0, p = 0;
   while (p < n) {
      size_t e = p;
      while (e < n && text[e] != '\n') e++;
      uint32_t state = 0;
      for (size_t q = p; q < e; q += 4) {
         uint32_t b = 0;
         for (int i = 0; i < 4; i++) b |= (q + i < e ? (uint32_t)(text[q + i] >> 1) & 3u : 0u) << (6 - 2 * i);
         const uint16_t ent = d->table[(size_t)(state >> 9) * 256 + b];
         for (int i = 0; i < 4; i++)
            if ((ent >> i) & 1u && q + i < e) { if (ne < cap) out[ne] = (uint64_t)(q + i); ne++; }
         state = ent & 0xFE00u;
      }
      p = e + 1;
   }
   seeq_quad_free(d);
   return (long)ne;
}

// The same table under the chunked ASCII walk (seeq_pair.h, QD): every `chain`-byte chain of the text walked on its own from the root, four
// bytes per step, after a warm-up over the `warm_bytes` bytes before it (0 = the automaton's own warm-up rounded up to whole words; '\n' where
// the buffer starts or has ended).  An accept inside the owned chain reports its byte; an accept during the warm-up reports the chain's
// first byte (the made-up candidate).  info as harness_quad_walk, info[5] = warm-up bytes of the automaton.
extern "C" long harness_quad_chain_walk(const uint8_t *text, size_t n, const char *keys, int m, int tau, int chain, int warm_bytes,
                                        uint64_t *out, size_t cap, uint32_t *info)
{
   seeq_quad_t *d = seeq_quad_plan(keys, m, tau);
   if (!d) return -1;
   if (info) { info[0] = d->nstates; info[1] = d->nstates_raw; info[2] = (uint32_t)d->mp; info[3] = (uint32_t)d->nparts; info[4] = (uint32_t)(d->p_accept * 1e9); info[5] = (uint32_t)d->warm; }
   const long long W = warm_bytes > 0 ? warm_bytes : 4 * ((d->warm + 3) / 4);
   size_t ne = 0;
   for (size_t c0 = 0; c0 < n; c0 += (size_t)chain) {
      uint32_t state = 0;
      bool warm_flag = false, made_up = false;
      for (long long p = (long long)c0 - W; p < (long long)c0 + chain && p < (long long)n; p += 4) {
         uint32_t b = 0;
         for (int i = 0; i < 4; i++) {
            const long long q = p + i;
            const uint8_t ch = (q < 0 || q >= (long long)n) ? (uint8_t)'\n' : text[q];
            b |= ((uint32_t)(ch >> 1) & 3u) << (6 - 2 * i);
         }
         const uint16_t ent = d->table[(size_t)(state >> 9) * 256 + b];
         state = ent & 0xFE00u;
         if (p < (long long)c0) { warm_flag |= (ent & 0xFu) != 0; continue; }
         if (p == (long long)c0 && warm_flag) made_up = true;
         for (int i = 0; i < 4; i++) {
            const bool hit = ((ent >> i) & 1u) != 0 || (i == 0 && made_up);
            if (i == 0) made_up = false;
            if (hit && p + i < (long long)n) { if (ne < cap) out[ne] = (uint64_t)(p + i); ne++; }
         }
      }
   }
   seeq_quad_free(d);
   return (long)ne;
}

// ---- several patterns, one walk (seeq_dfa.h section 4) ----
// keys: the patterns' key bytes concatenated (m[p] each).  harness_multi_walk: the union pair automaton walked as
// harness_pair_walk walks a single one (chains of `chain` bytes, warm-up, restart, made-up candidates) -> candidate positions.
// info[0..5] = pair states, raw states, longest prefix, warm-up bytes, resolve states, resolve exact.
static seeq_multi_t *harness_multi_make(const char *keys, const int *m, const int *tau, int npat)
{
   const char *kp[SEEQ_MULTI_MAX_PARTS];
   if (npat < 1 || npat > SEEQ_MULTI_MAX_PARTS) return nullptr;
   int o = 0;
   for (int p = 0; p < npat; p++) { kp[p] = keys + o; o += m[p]; }
   return seeq_multi_build(kp, m, tau, npat);
}

extern "C" long harness_multi_walk(const uint8_t *text, size_t n, const char *keys, const int *m, const int *tau, int npat, int chain,
                                   uint64_t *out, size_t cap, uint32_t *info)
{
   seeq_multi_t *d = harness_multi_make(keys, m, tau, npat);
   if (!d) return -1;
   if (info) { info[0] = d->pair->nstates; info[1] = d->pair->nstates_raw; info[2] = (uint32_t)d->pair->mp; info[3] = (uint32_t)d->pair->warm;
               info[4] = d->res_states; info[5] = (uint32_t)d->res_exact; info[6] = (uint32_t)d->maxspan; }
   const long long W = 4 * ((d->pair->warm + 3) / 4) < 16 ? 16 : 4 * ((d->pair->warm + 3) / 4);
   size_t ne = 0;
   for (size_t c0 = 0; c0 < n; c0 += (size_t)chain) {
      uint32_t state = 0;
      bool warm_flag = false;
      for (long long p = (long long)c0 - W; p < (long long)c0 + chain && p < (long long)n; p += 2) {
         const uint8_t b1 = p < 0 ? (uint8_t)'\n' : text[p];
         const uint8_t b2 = (p + 1 < 0 || p + 1 >= (long long)n) ? (uint8_t)'\n' : text[p + 1];
         uint16_t nxt;
         memcpy(&nxt, d->pair->table + state + 2u * (uint32_t)(((b1 >> 1) & 3) * 4 + ((b2 >> 1) & 3)), 2);
         state = nxt;
         const bool flagged = (state & 1u) != 0;
         if (p < (long long)c0) { warm_flag |= flagged; continue; }
         if (flagged || (p == (long long)c0 && warm_flag)) {
            const uint64_t pos = (uint64_t)(p + 1 < (long long)n ? p + 1 : (long long)n - 1);
            if (ne < cap) out[ne] = pos;
            ne++;
         }
      }
   }
   seeq_multi_free(d);
   return (long)ne;
}

// The resolve automaton walked from the root over text[lo, hi) (a byte outside A C G T N, either case, ends the walk):
// the union of the masks of the states it visits.  nwin windows at once (lo[i], hi[i]) -> masks[i].
extern "C" int harness_multi_resolve(const uint8_t *text, const char *keys, const int *m, const int *tau, int npat,
                                     const uint64_t *lo, const uint64_t *hi, size_t nwin, uint32_t *masks)
{
   seeq_multi_t *d = harness_multi_make(keys, m, tau, npat);
   if (!d) return -1;
   for (size_t i = 0; i < nwin; i++) {
      uint32_t q = 0, acc = 0;
      for (uint64_t p = lo[i]; p < hi[i]; p++) {
         const uint8_t b = text[p] & 0xDFu;
         const int c = b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : (b == 'T' || b == 'U') ? 3 : b == 'N' ? 4 : -1;
         if (c < 0) break;
         q = d->res_next[(size_t)q * 8 + c];
         acc |= d->res_mask[q];
      }
      masks[i] = acc;
   }
   seeq_multi_free(d);
   return 0;
}


// ---- the scan planner (seeq_amd/csrc/seeq_plan.h): the same pure function run_segments executes, with the automata built on the
//      host alone (seeq_dfa.h) -- which kernels would serve this pattern / these options / this line length? ----
#include "../seeq_amd/csrc/seeq_plan.h"

namespace {
struct plan_ctx { const char *keys; int m, tau; };
void plan_ensure_host(void *ctx, int which, int complete_only, PlanAutomata *au)
{
   const plan_ctx *c = (const plan_ctx *)ctx;
   if (which == 0) {
      seeq_dfa_t *d = c->m <= 62 ? seeq_dfa_plan_stream(c->keys, c->m, c->tau, complete_only) : NULL;
      au->sdfa_state = d ? 1 : -1;
      if (d) { au->sdfa_parts = d->nparts; au->sdfa_warm = d->warm; au->sdfa_pacc = d->p_accept; seeq_dfa_free(d); }
   } else {
      seeq_pair_t *d = c->m <= 62 ? seeq_pair_plan(c->keys, c->m, c->tau) : NULL;
      au->pair_state = d ? 1 : -1;
      if (d) { au->pair_warm = d->warm; au->pair_pacc = d->p_accept; seeq_pair_free(d); }
   }
}
}  // namespace

// flags: bit 0 force_ll, 1 no_stream, 2 no_stream_nd, 3 no_window, 4 sample_dirty, 5 multi_active; knob_kernel: SEEQ_FUSED_KERNEL (0 auto).
// out[16]: rc, path, fw, use_stream, use_pair, use_myers, filter, stream_ll, stream_sub, stream_wu, verify, order2, leaders, window_ok,
//          ll_filter (2: on the restart table), skip_back.  Returns 0.
extern "C" int harness_plan(const char *keys, int m, int tau, int options, int want, double avg_line, int flags, int knob_kernel, int *out)
{
   ScanKnobs kn;
   std::memset(&kn, 0, sizeof kn);
   kn.kernel = knob_kernel;
   PlanIn in;
   std::memset(&in, 0, sizeof in);
   in.wlen = m; in.tau = tau; in.options = options; in.want = want; in.avg_line = avg_line;
   in.force_ll = flags & 1; in.no_stream = (flags >> 1) & 1; in.no_stream_nd = (flags >> 2) & 1; in.no_window = (flags >> 3) & 1;
   in.sample_dirty = (flags >> 4) & 1; in.multi_active = (flags >> 5) & 1;
   in.seg_bytes = (size_t)0xF0000000u;
   in.kn = &kn;
   PlanAutomata au;
   std::memset(&au, 0, sizeof au);
   plan_ctx ctx = {keys, m, tau};
   const ScanPlan p = seeq_plan_scan(in, au, plan_ensure_host, &ctx);
   const int v[16] = {p.rc, p.path, p.fw, p.use_stream, p.use_pair, p.use_myers, p.filter, p.stream_ll, p.stream_sub, p.stream_wu, p.verify, p.order2,
                      p.leaders, p.window_ok, p.ll_restart ? 2 : (p.ll_filter ? 1 : 0), (int)p.skip_back};
   for (int i = 0; i < 16; i++) out[i] = v[i];
   return 0;
}
