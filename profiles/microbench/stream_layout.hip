// Microbenchmark: HBM read rate when every LANE streams its own contiguous stretch of S bytes, 128 B (one memory line,
// eight 16-B loads) per phase -- the access shape of a k_stream variant whose lanes walk long stretches (no per-chunk
// warm-up).  S = 128 is today's k_stream tile (lane stride 128 B, wave tile 8 KB).
// hipcc --offload-arch=gfx950 -O3 stream_layout.hip -o stream_layout && ./stream_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

template <int S>
__global__ __launch_bounds__(1024) void k_read(const uint8_t *src, size_t ntiles, uint32_t *out, int spin)
{
   const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, nwaves = (size_t)gridDim.x * (blockDim.x / 64);
   const int lane = threadIdx.x & 63;
   uint32_t acc = 0;
   for (size_t t = wave; t < ntiles; t += nwaves) {
      const uint8_t *base = src + t * (size_t)(64 * S) + (size_t)lane * S;
#pragma unroll 1
      for (int p = 0; p < S / 128; p++) {
         v4u v[8];
#pragma unroll
         for (int q = 0; q < 8; q++) v[q] = *reinterpret_cast<const v4u *>(base + 128 * p + 16 * q);
#pragma unroll
         for (int q = 0; q < 8; q++) acc ^= v[q].x ^ v[q].y ^ v[q].z ^ v[q].w;
         for (int k = 0; k < spin; k++) acc = acc * 1664525u + 1013904223u;      // stand-in for the walk between phases
      }
   }
   if (acc == 0x12345678u) out[0] = acc;
}

// Two walks per lane sharing one set of text registers: per phase 64 B of the first half of the stretch and 64 B of the
// second half -- every 128-B memory line is consumed over two consecutive phases.
template <int S>
__global__ __launch_bounds__(1024) void k_read_half(const uint8_t *src, size_t ntiles, uint32_t *out, int spin)
{
   const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, nwaves = (size_t)gridDim.x * (blockDim.x / 64);
   const int lane = threadIdx.x & 63;
   uint32_t acc = 0;
   for (size_t t = wave; t < ntiles; t += nwaves) {
      const uint8_t *base = src + t * (size_t)(64 * S) + (size_t)lane * S;
#pragma unroll 1
      for (int p = 0; p < S / 128; p++) {
         v4u v[8];
#pragma unroll
         for (int q = 0; q < 4; q++) v[q] = *reinterpret_cast<const v4u *>(base + 64 * p + 16 * q);
#pragma unroll
         for (int q = 0; q < 4; q++) v[4 + q] = *reinterpret_cast<const v4u *>(base + S / 2 + 64 * p + 16 * q);
#pragma unroll
         for (int q = 0; q < 8; q++) acc ^= v[q].x ^ v[q].y ^ v[q].z ^ v[q].w;
         for (int k = 0; k < spin; k++) acc = acc * 1664525u + 1013904223u;
      }
   }
   if (acc == 0x12345678u) out[0] = acc;
}

template <int S>
static void run_half(const uint8_t *d, size_t bytes, uint32_t *o, int wgs_per_cu, int ncu, int spin)
{
   hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
   const int grid = wgs_per_cu * ncu;
   const size_t ntiles = bytes / (64 * (size_t)S);
   hipLaunchKernelGGL((k_read_half<S>), dim3(grid), dim3(1024), 0, 0, d, ntiles, o, spin);
   hipDeviceSynchronize();
   hipEventRecord(e0);
   for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_read_half<S>), dim3(grid), dim3(1024), 0, 0, d, ntiles, o, spin);
   hipEventRecord(e1); hipEventSynchronize(e1);
   float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
   printf("lane stretch %6d B, two half-line streams per lane  spin %4d  %d WG/CU : %.3f ms  %.2f TB/s\n", S, spin, wgs_per_cu, ms,
          ntiles * 64.0 * S / (ms * 1e-3) / 1e12);
}

template <int S>
static void run(const uint8_t *d, size_t bytes, uint32_t *o, int wgs_per_cu, int ncu, int spin)
{
   hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
   const int grid = wgs_per_cu * ncu;
   const size_t ntiles = bytes / (64 * (size_t)S);
   hipLaunchKernelGGL((k_read<S>), dim3(grid), dim3(1024), 0, 0, d, ntiles, o, spin);
   hipDeviceSynchronize();
   hipEventRecord(e0);
   for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_read<S>), dim3(grid), dim3(1024), 0, 0, d, ntiles, o, spin);
   hipEventRecord(e1); hipEventSynchronize(e1);
   float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
   printf("lane stretch %6d B (wave tile %5d KB)  spin %4d  %d WG/CU : %.3f ms  %.2f TB/s\n", S, 64 * S / 1024, spin, wgs_per_cu, ms,
          ntiles * 64.0 * S / (ms * 1e-3) / 1e12);
}

int main()
{
   hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
   const int ncu = p.multiProcessorCount;
   const size_t bytes = (size_t)12 << 30;
   uint8_t *d; uint32_t *o;
   hipMalloc(&d, bytes); hipMalloc(&o, 4);
   hipMemset(d, 'A', bytes);
   printf("%s CUs=%d, %zu GiB read-only sweep\n", p.name, ncu, bytes >> 30);
   for (int spin : {0, 200, 1500}) {
      run_half<1024>(d, bytes, o, 2, ncu, spin);
      run_half<2048>(d, bytes, o, 2, ncu, spin);
      run_half<4096>(d, bytes, o, 2, ncu, spin);
      run<1024>(d, bytes, o, 2, ncu, spin);
   }
   for (int spin : {0, 200}) {
      for (int w : {1, 2}) {
         run<128>(d, bytes, o, w, ncu, spin);
         run<256>(d, bytes, o, w, ncu, spin);
         run<512>(d, bytes, o, w, ncu, spin);
         run<1024>(d, bytes, o, w, ncu, spin);
         run<4096>(d, bytes, o, w, ncu, spin);
         run<16384>(d, bytes, o, w, ncu, spin);
      }
   }
   return 0;
}
