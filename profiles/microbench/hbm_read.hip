// Microbenchmark: read-only HBM streaming rate on gfx950 for the access shapes the scan kernels use.
// hipcc --offload-arch=gfx950 -O3 hbm_read.hip -o hbm_read && ./hbm_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// MODE 0: coalesced (a wave instruction reads 1 KB contiguous), UNR loads in flight per lane
// MODE 1: per-lane 128-B chunks (a wave instruction reads 64 pieces of 16 B, 128 B apart), 8 loads per lane
template <int MODE, int UNR>
__global__ __launch_bounds__(1024) void k_read(const v4u *src, size_t n16, uint32_t *out)
{
   const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, nwaves = (size_t)gridDim.x * (blockDim.x / 64);
   const int lane = threadIdx.x & 63;
   uint32_t acc = 0;
   const size_t tile16 = 64 * UNR;                       // 16-B pieces per wave tile
   for (size_t t = wave; (t + 1) * tile16 <= n16; t += nwaves) {
      v4u v[UNR];
#pragma unroll
      for (int q = 0; q < UNR; q++) v[q] = MODE == 0 ? src[t * tile16 + q * 64 + lane] : src[t * tile16 + lane * UNR + q];
#pragma unroll
      for (int q = 0; q < UNR; q++) acc ^= v[q].x ^ v[q].y ^ v[q].z ^ v[q].w;
   }
   if (acc == 0x12345678u) out[0] = acc;
}

__global__ void k_fill(uint32_t *dst, size_t n)
{
   for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
      uint64_t z = i * 0x9E3779B97F4A7C15ull; z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
      const char *b = "ACGT";
      dst[i] = (uint32_t)b[z & 3] | ((uint32_t)b[(z >> 2) & 3] << 8) | ((uint32_t)b[(z >> 4) & 3] << 16) | ((uint32_t)b[(z >> 6) & 3] << 24);
   }
}

template <int MODE, int UNR>
static void run(const v4u *d, size_t bytes, uint32_t *o, int wgs_per_cu, int threads, int ncu)
{
   hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
   const int grid = wgs_per_cu * ncu;
   hipLaunchKernelGGL((k_read<MODE, UNR>), dim3(grid), dim3(threads), 0, 0, d, bytes / 16, o);
   hipDeviceSynchronize();
   hipEventRecord(e0);
   for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_read<MODE, UNR>), dim3(grid), dim3(threads), 0, 0, d, bytes / 16, o);
   hipEventRecord(e1); hipEventSynchronize(e1);
   float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
   printf("%-34s loads/lane=%d  %4d threads x %d WG/CU : %.3f ms  %.2f TB/s\n", MODE == 0 ? "coalesced 1 KB per instruction" : "128-B chunk per lane",
          UNR, threads, wgs_per_cu, ms, bytes / (ms * 1e-3) / 1e12);
}

int main()
{
   hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
   const int ncu = p.multiProcessorCount;
   const size_t bytes = (size_t)12 << 30;
   v4u *d; uint32_t *o;
   hipMalloc(&d, bytes); hipMalloc(&o, 4);
   hipMemset(d, 1, bytes);
   printf("%s CUs=%d, %zu GiB read-only sweep, constant bytes\n", p.name, ncu, bytes >> 30);
   run<0, 8>(d, bytes, o, 2, 1024, ncu);
   run<1, 8>(d, bytes, o, 2, 1024, ncu);
   hipLaunchKernelGGL(k_fill, dim3(ncu * 8), dim3(256), 0, 0, (uint32_t *)d, bytes / 4);
   hipDeviceSynchronize();
   printf("same buffer filled with pseudo-random DNA-like bytes\n");
   run<0, 4>(d, bytes, o, 2, 1024, ncu);
   run<0, 8>(d, bytes, o, 2, 1024, ncu);
   run<0, 8>(d, bytes, o, 1, 1024, ncu);
   run<0, 8>(d, bytes, o, 8, 256, ncu);
   run<0, 16>(d, bytes, o, 2, 1024, ncu);
   run<1, 8>(d, bytes, o, 2, 1024, ncu);
   run<1, 8>(d, bytes, o, 8, 256, ncu);
   run<1, 4>(d, bytes, o, 2, 1024, ncu);
   return 0;
}
