// Microbenchmark: the DFA walk's LDS gather with the frequently visited ("hot") states replicated once per LDS bank.
//
// Today (seeq_stream.h): rows of 16 B (8 columns x u16), state value = row byte offset, address = state ^ (byte & 0xE);
// 64 lanes gather from a 53 KB table -> 5.6 LDS cycles per wave gather (32 banks, bank conflicts).
// Variant measured here: rows hold only the four DNA columns (8 B); the column bits sit in address bits 1 and 7, address
// bits 6..2 select the BANK.  Cold rows are interleaved 32 to a 256-B block (bank = row % 32).  The H hottest states
// have 32 copies, one per bank: lane l (l % 32) only ever reads bank l for them -> no conflict among hot lanes.
// Values stored in a hot copy already point into the same copy (or at a cold row); a cold row pointing at a hot state
// stores (hot base | 1) and the walk adds the lane's bank offset (two VALU ops per character).
// hipcc --offload-arch=gfx950 -O3 -I../../seeq_amd/csrc lds_gather2.hip -o lds_gather2 && ./lds_gather2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include "seeq_dfa.h"

typedef __attribute__((address_space(3))) const uint16_t lds_cu16;

#define XOR_B(K, AD, ST, WM) asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #K : "=v"(AD) : "v"(ST), "v"(WM))
#define HIT(HM, ST, ACC) asm("v_cmp_eq_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(HM) : "v"(ST), "v"(ACC) : "vcc")
#define NL(K, NM, W, TEN) asm("v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:BYTE_" #K " src1_sel:DWORD\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(NM) : "v"(W), "v"(TEN) : "vcc")
// newline: reset the state, then count (the compare's vcc feeds both)
#define NLR(K, NM, W, TEN, ST, ROOT) asm("v_cmp_eq_u32_sdwa vcc, %2, %3 src0_sel:BYTE_" #K " src1_sel:DWORD\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(NM), "+v"(ST) : "v"(W), "v"(TEN), "v"(ROOT) : "vcc")

// a cold row's pointer at a hot state carries bit 0: add the lane's bank offset (kfix = 4 * (lane % 32) - 1)
#define FIX(ST) { uint32_t t_; asm("v_and_b32 %0, 1, %1" : "=v"(t_) : "v"(ST)); asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(ST) : "v"(t_), "v"(kfix)); }
// MODE 0: today's table (16-B rows, rotated), today's per-character ops
// MODE 1: 8-B rows + hot copies, +2 VALU (lane fix-up) +1 (newline reset), 3 ops per 4 characters to place the column bits
template <int MODE>
__global__ __launch_bounds__(1024) void k_walk(const uint4 *table, uint32_t n16, uint32_t *out, int iters, uint32_t acc_new, uint32_t root_val, uint32_t hot_base)
{
   extern __shared__ __align__(16) uint8_t lds[];
   for (uint32_t i = threadIdx.x; i < n16; i += 1024) reinterpret_cast<uint4 *>(lds)[i] = table[i];
   __syncthreads();
   const uint32_t lane = threadIdx.x & 63;
   const int32_t kfix = (int32_t)((lane & 31) * 4) - 1;
   const uint32_t ten = 0x0Au;
   uint32_t sa = root_val, sb = root_val, hma = 0, hmb = 0, nma = 0, nmb = 0;
   if (MODE == 1) { sa += (lane & 31) * 4; sb = sa; }
   const uint32_t myroot = sa;
   uint32_t ra = (blockIdx.x * 1024 + threadIdx.x) * 2654435761u + 1u, rb = ra * 40503u + 77u;
   for (int it = 0; it < iters; it++) {
      ra = ra * 1664525u + 1013904223u; rb = rb * 1664525u + 1013904223u;
#pragma unroll
      for (int g = 0; g < 2; g++) {                       // 2 x 4 characters per rng word
         const uint32_t wa = ra >> (8 * g + 3), wb = rb >> (8 * g + 3);      // "text": 4 pseudo-random bytes per chain
         uint32_t wma, wmb, ada, adb;
         if (MODE == 0) { wma = wa & 0x06060606u; wmb = wb & 0x06060606u; }
         else {
            wma = ((wa & 0x04040404u) << 5) | (wa & 0x02020202u);
            wmb = ((wb & 0x04040404u) << 5) | (wb & 0x02020202u);
         }
#define STEP0(K) \
         XOR_B(K, ada, sa, wma); XOR_B(K, adb, sb, wmb); \
         sa = *(lds_cu16 *)(uintptr_t)ada; sb = *(lds_cu16 *)(uintptr_t)adb; \
         HIT(hma, sa, acc_new); NL(K, nma, wa, ten); HIT(hmb, sb, acc_new); NL(K, nmb, wb, ten);
#define STEP1(K) \
         XOR_B(K, ada, sa, wma); XOR_B(K, adb, sb, wmb); \
         sa = *(lds_cu16 *)(uintptr_t)ada; sb = *(lds_cu16 *)(uintptr_t)adb; \
         FIX(sa); FIX(sb); \
         HIT(hma, sa, acc_new); NLR(K, nma, wa, ten, sa, myroot); HIT(hmb, sb, acc_new); NLR(K, nmb, wb, ten, sb, myroot);
         if (MODE == 0) { STEP0(0) STEP0(1) STEP0(2) STEP0(3) }
         else { STEP1(0) STEP1(1) STEP1(2) STEP1(3) }
      }
   }
   out[blockIdx.x * 1024 + threadIdx.x] = sa ^ sb ^ hma ^ hmb ^ nma ^ nmb;
}

struct Built { std::vector<uint8_t> bytes; uint32_t acc_new, root_val, hot_base; double hotfrac; };

// logical automaton: next[s][c], c = 0..3 in TABLE column order (A C T G); ACC (state 1) restarts at the root
static Built build(const uint32_t *next5, uint32_t n, int H, int mode)
{
   static const int cls_of_col[4] = {0, 1, 3, 2};       // column (A C T G) -> class (A C G T)
   std::vector<std::array<uint32_t, 4>> nx(n);
   for (uint32_t s = 0; s < n; s++) for (int c = 0; c < 4; c++) { uint32_t t = next5[(size_t)s * 5 + cls_of_col[c]]; nx[s][c] = t == 1 ? 0 : t; }
   // stationary distribution (lines of 150 random bases: reset every 151st step is ignored here)
   std::vector<double> cur(n, 0.0), nxt(n);
   cur[0] = 1.0;
   for (int r = 0; r < 200; r++) { std::fill(nxt.begin(), nxt.end(), 0.0); for (uint32_t s = 0; s < n; s++) if (cur[s] > 0) for (int c = 0; c < 4; c++) nxt[nx[s][c]] += cur[s] * 0.25; cur.swap(nxt); }
   std::vector<uint32_t> order(n); for (uint32_t i = 0; i < n; i++) order[i] = i;
   std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cur[a] > cur[b]; });
   std::vector<int> hot(n, -1);
   double hf = 0; for (int i = 0; i < H && i < (int)n; i++) { hot[order[i]] = i; hf += cur[order[i]]; }
   Built b; b.hotfrac = hf;
   if (mode == 0) {                                      // today's layout via seeq_dfa_layout_stream
      uint32_t *cp = (uint32_t *)malloc((size_t)n * 5 * 4); memcpy(cp, next5, (size_t)n * 5 * 4);
      // make ACC restart at the root so that the walk stays spread out (as lds_gather.hip does)
      for (uint32_t s = 0; s < n; s++) for (int c = 0; c < 5; c++) if (cp[(size_t)s * 5 + c] == 1) cp[(size_t)s * 5 + c] = 0;
      seeq_dfa_t *d = seeq_dfa_layout_stream(cp, n);
      b.bytes.assign((uint8_t *)d->table, (uint8_t *)d->table + (size_t)d->nrows * 16);
      b.acc_new = d->acc_final; b.root_val = 0; b.hot_base = 0;
      seeq_dfa_free(d);
      return b;
   }
   const uint32_t nblocks = (n + 31) / 32, cold_bytes = nblocks * 256, hot_base = cold_bytes;
   b.bytes.assign((size_t)cold_bytes + (size_t)H * 256, 0);
   auto cold_addr = [&](uint32_t s) { return (s / 32) * 256 + (s % 32) * 4; };
   auto ent_off = [&](int c) { return (uint32_t)((c & 1) * 2 + (c >> 1) * 128); };
   for (uint32_t s = 0; s < n; s++)
      for (int c = 0; c < 4; c++) {
         const uint32_t t = nx[s][c];
         const uint16_t v = hot[t] >= 0 ? (uint16_t)((hot_base + (uint32_t)hot[t] * 256) | 1u) : (uint16_t)cold_addr(t);
         memcpy(&b.bytes[cold_addr(s) + ent_off(c)], &v, 2);
      }
   for (int h = 0; h < H && h < (int)n; h++) {
      const uint32_t s = order[h];
      for (uint32_t l = 0; l < 32; l++)
         for (int c = 0; c < 4; c++) {
            const uint32_t t = nx[s][c];
            const uint16_t v = hot[t] >= 0 ? (uint16_t)(hot_base + (uint32_t)hot[t] * 256 + l * 4) : (uint16_t)cold_addr(t);
            memcpy(&b.bytes[hot_base + (uint32_t)h * 256 + l * 4 + ent_off(c)], &v, 2);
         }
   }
   b.acc_new = 0xFFFFu; b.hot_base = hot_base;
   b.root_val = hot[0] >= 0 ? hot_base + (uint32_t)hot[0] * 256 : cold_addr(0);      // (+ lane offset when hot)
   if (hot[0] < 0) b.root_val = cold_addr(0);
   return b;
}

template <int MODE>
static void run(const Built &b, int wgs_per_cu, int ncu, const char *what)
{
   uint4 *d_table; uint32_t *d;
   const size_t lds = (b.bytes.size() + 15) & ~(size_t)15;
   hipMalloc(&d_table, lds); hipMemcpy(d_table, b.bytes.data(), b.bytes.size(), hipMemcpyHostToDevice);
   const int blocks = wgs_per_cu * ncu;
   hipMalloc(&d, (size_t)blocks * 1024 * 4);
   if (hipFuncSetAttribute((const void *)k_walk<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { printf("%s: LDS %zu refused\n", what, lds); return; }
   int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k_walk<MODE>, 1024, lds);
   const int iters = 4000;
   hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
   hipLaunchKernelGGL(k_walk<MODE>, dim3(blocks), dim3(1024), lds, 0, d_table, (uint32_t)(lds / 16), d, 10, b.acc_new, b.root_val, b.hot_base);
   hipDeviceSynchronize();
   hipEventRecord(e0);
   hipLaunchKernelGGL(k_walk<MODE>, dim3(blocks), dim3(1024), lds, 0, d_table, (uint32_t)(lds / 16), d, iters, b.acc_new, b.root_val, b.hot_base);
   hipEventRecord(e1); hipEventSynchronize(e1);
   float ms; hipEventElapsedTime(&ms, e0, e1);
   const double gathers = (double)blocks * 16 * iters * 8 * 2;             // wave-level ds_read_u16 (two chains)
   const double per_cu_per_us = gathers / ncu / (ms * 1e3);
   printf("%-44s LDS %6zu B  WG/CU %d (occupancy %d)  hot %.3f : %.3f ms  %.2f cycles/gather/CU @2.4GHz  %.2f T chars/s\n",
          what, lds, wgs_per_cu, occ, b.hotfrac, ms, 2400.0 / per_cu_per_us, gathers * 64 / (ms * 1e-3) / 1e12);
   hipFree(d); hipFree(d_table);
}

int main()
{
   hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
   const int ncu = p.multiProcessorCount;
   const char *pat = "GATGTAGCGCGATTAGCCTG";
   char keys[64]; int m = 0;
   for (const char *c = pat; *c; c++) keys[m++] = *c == 'A' ? 1 : *c == 'C' ? 2 : *c == 'G' ? 4 : 8;
   uint32_t *next = NULL;
   const uint32_t n = seeq_dfa_bfs(keys, m, 3, &next);
   printf("%s CUs=%d; automaton: %u states\n", p.name, ncu, n);
   { Built b = build(next, n, 0, 0); run<0>(b, 2, ncu, "today: 16-B rows, rotated"); run<0>(b, 1, ncu, "today: 16-B rows, rotated"); }
   for (int H : {0, 32, 64, 96, 128, 144}) {
      Built b = build(next, n, H, 1);
      char what[64]; snprintf(what, sizeof what, "8-B rows + %d hot states x 32 banks", H);
      if (b.bytes.size() * 2 <= 160 * 1024) run<1>(b, 2, ncu, what);
      run<1>(b, 1, ncu, what);
   }
   return 0;
}
