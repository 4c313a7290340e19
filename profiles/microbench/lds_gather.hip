// Microbenchmark: throughput / latency of the DFA walk's LDS gather on gfx950.
// Each lane walks the real Levenshtein automaton of the headline pattern (seeq_dfa.h, 3 342 states, 53 KB
// table in LDS) over pseudo-random DNA: state = TABLE[state | col]; ILP independent walks per lane.
// hipcc --offload-arch=gfx950 -O3 -I../../seeq_amd/csrc lds_gather.hip -o lds_gather && ./lds_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "seeq_dfa.h"

template <int ILP>
__global__ __launch_bounds__(1024) void k_walk(const uint4 *table, uint32_t rows, uint32_t *out, int iters)
{
   extern __shared__ __align__(16) uint8_t lds[];
   for (uint32_t i = threadIdx.x; i < rows; i += 1024) reinterpret_cast<uint4 *>(lds)[i] = table[i];
   __syncthreads();
   typedef __attribute__((address_space(3))) const uint16_t lds_cu16;
   uint32_t st[ILP], rng[ILP];
#pragma unroll
   for (int i = 0; i < ILP; i++) { st[i] = 0; rng[i] = (blockIdx.x * 1024 + threadIdx.x) * 2654435761u + i * 40503u + 1u; }
   for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < ILP; i++) rng[i] = rng[i] * 1664525u + 1013904223u;
#pragma unroll
      for (int k = 0; k < 8; k++) {                    // 8 characters per rng word: cols 0..3 (A C T G), 2 bits each from the top
#pragma unroll
         for (int i = 0; i < ILP; i++) {
            const uint32_t col = ((rng[i] >> (16 + 2 * k)) & 3u) << 1;
            st[i] = *(lds_cu16 *)(uintptr_t)(st[i] | col);
            if (st[i] == 16) st[i] = 0;                 // ACC (state 1) is absorbing: restart so the walk stays spread out
         }
      }
   }
   uint32_t acc = 0;
#pragma unroll
   for (int i = 0; i < ILP; i++) acc ^= st[i];
   out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

template <int ILP>
static void run(const uint4 *d_table, uint32_t rows, int wgs_per_cu, int ncu)
{
   uint32_t *d;
   const int blocks = wgs_per_cu * ncu;
   hipMalloc(&d, (size_t)blocks * 1024 * 4);
   const size_t lds = (size_t)rows * 16;
   hipFuncSetAttribute((const void *)k_walk<ILP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
   const int iters = 2000;
   hipEvent_t e0, e1;
   hipEventCreate(&e0); hipEventCreate(&e1);
   hipLaunchKernelGGL(k_walk<ILP>, dim3(blocks), dim3(1024), lds, 0, d_table, rows, d, 10);
   hipDeviceSynchronize();
   hipEventRecord(e0);
   hipLaunchKernelGGL(k_walk<ILP>, dim3(blocks), dim3(1024), lds, 0, d_table, rows, d, iters);
   hipEventRecord(e1);
   hipEventSynchronize(e1);
   float ms; hipEventElapsedTime(&ms, e0, e1);
   const double gathers = (double)blocks * 16 * iters * 8 * ILP;           // wave-level ds_read_u16
   const double per_cu_per_us = gathers / ncu / (ms * 1e3);
   printf("ILP=%d waves/SIMD=%d  %.3f ms  %.1f wave-gathers/us/CU => %.2f cycles/gather/CU @2.4GHz, %.2f T chars/s chip; per-wave chain %.0f cycles/char\n",
          ILP, wgs_per_cu * 4, ms, per_cu_per_us, 2400.0 / per_cu_per_us, gathers * 64 / (ms * 1e-3) / 1e12,
          ms * 1e-3 * 2.4e9 / ((double)iters * 8));
   hipFree(d);
}

int main()
{
   hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
   const int ncu = p.multiProcessorCount;
   const char *pat = "GATGTAGCGCGATTAGCCTG";
   char keys[64]; int m = 0;
   for (const char *c = pat; *c; c++) keys[m++] = *c == 'A' ? 1 : *c == 'C' ? 2 : *c == 'G' ? 4 : 8;
   seeq_dfa_t *d = seeq_dfa_build(keys, m, 3);
   if (!d) { printf("dfa build failed\n"); return 1; }
   printf("%s CUs=%d; automaton: %u states, table %u B\n", p.name, ncu, d->nstates, d->nrows * 16);
   uint4 *d_table; hipMalloc(&d_table, (size_t)d->nrows * 16);
   hipMemcpy(d_table, d->table, (size_t)d->nrows * 16, hipMemcpyHostToDevice);
   for (int w : {1, 2}) {
      run<1>(d_table, d->nrows, w, ncu);
      run<2>(d_table, d->nrows, w, ncu);
      run<4>(d_table, d->nrows, w, ncu);
   }
   return 0;
}
