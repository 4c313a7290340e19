// Microbenchmark: what a HIP process pays before its first result (the fixed cost inside every `seeq` CLI run).
// hipcc --offload-arch=gfx950 -O2 hip_startup.hip -o hip_startup && ./hip_startup
#include <hip/hip_runtime.h>
#include <cstdio>
#include <ctime>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
__global__ void k_nop(int *p) { if (p) *p = 1; }
int main()
{
   double t0 = now(), t;
   int n = 0; hipGetDeviceCount(&n);                         t = now(); printf("hipGetDeviceCount      %7.1f ms (devices: %d)\n", (t - t0) * 1e3, n); t0 = t;
   hipSetDevice(0);                                          t = now(); printf("hipSetDevice           %7.1f ms\n", (t - t0) * 1e3); t0 = t;
   int *d; hipMalloc(&d, 16);                                t = now(); printf("first hipMalloc        %7.1f ms\n", (t - t0) * 1e3); t0 = t;
   hipStream_t s; hipStreamCreate(&s);                       t = now(); printf("hipStreamCreate        %7.1f ms\n", (t - t0) * 1e3); t0 = t;
   hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s, d); hipStreamSynchronize(s);
                                                             t = now(); printf("first launch + sync    %7.1f ms (code object load)\n", (t - t0) * 1e3); t0 = t;
   void *h; hipHostMalloc(&h, (size_t)64 << 20, 0);          t = now(); printf("hipHostMalloc 64 MiB   %7.1f ms\n", (t - t0) * 1e3); t0 = t;
   void *h2; hipHostMalloc(&h2, (size_t)64 << 20, 0);        t = now(); printf("hipHostMalloc 64 MiB   %7.1f ms (second)\n", (t - t0) * 1e3); t0 = t;
   void *b; hipMalloc(&b, (size_t)80 << 20);                 t = now(); printf("hipMalloc 80 MiB       %7.1f ms\n", (t - t0) * 1e3); t0 = t;
   hipMemcpyAsync(b, h, (size_t)64 << 20, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
                                                             t = now(); printf("H2D 64 MiB             %7.1f ms\n", (t - t0) * 1e3); t0 = t;
   return 0;
}
