// Microbenchmark: int32 VALU issue rate and dependent-chain latency on gfx950.
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int ILP>
__global__ __launch_bounds__(256) void k_chain(uint32_t *out, int iters, uint32_t seed)
{
   uint32_t x[ILP];
#pragma unroll
   for (int i = 0; i < ILP; i++) x[i] = seed + threadIdx.x * 7 + i;
   uint32_t y = seed ^ 0x9E3779B9u, z = seed * 3u + threadIdx.x;
   for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
#pragma unroll
         for (int i = 0; i < ILP; i++) {
            // one dependent op per chain per step; bitop-like mix so nothing folds
            asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x[i]) : "v"(y), "v"(z));   // chains are independent
         }
      }
   }
   uint32_t acc = 0;
#pragma unroll
   for (int i = 0; i < ILP; i++) acc ^= x[i];
   out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int ILP>
static void run(const char *name, int blocks_per_cu, int ncu)
{
   uint32_t *d;
   const int blocks = blocks_per_cu * ncu;
   hipMalloc(&d, (size_t)blocks * 256 * 4);
   const int iters = 2000;
   hipEvent_t e0, e1;
   hipEventCreate(&e0); hipEventCreate(&e1);
   hipLaunchKernelGGL(k_chain<ILP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1u);
   hipDeviceSynchronize();
   hipEventRecord(e0);
   hipLaunchKernelGGL(k_chain<ILP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
   hipEventRecord(e1);
   hipEventSynchronize(e1);
   float ms; hipEventElapsedTime(&ms, e0, e1);
   const double wave_instr = (double)blocks * 4 * iters * 16 * ILP;        // wave-level VALU instructions
   const double per_simd_per_us = wave_instr / (ncu * 4.0) / (ms * 1e3);
   printf("%-28s blocks/CU=%d  %.3f ms  %.1f wave-instr/us/SIMD  => %.2f cycles/instr @2.4GHz  (%.2f T lane-ops/s)\n",
          name, blocks_per_cu, ms, per_simd_per_us, 2400.0 / per_simd_per_us, wave_instr * 64 / (ms * 1e-3) / 1e12);
   hipFree(d);
}

int main()
{
   hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
   const int ncu = p.multiProcessorCount;
   printf("%s  CUs=%d clock=%d kHz\n", p.name, ncu, p.clockRate);
   for (int b : {1, 2, 3, 4, 8}) {
      run<1>("dependent chain (ILP=1)", b, ncu);
      run<2>("2 chains (ILP=2)", b, ncu);
      run<4>("4 chains (ILP=4)", b, ncu);
      run<8>("8 chains (ILP=8)", b, ncu);
   }
   return 0;
}
