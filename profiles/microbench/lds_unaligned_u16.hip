// Does ds_read_u16 at an ODD LDS byte address return the two bytes at [a, a+1] on gfx950 (unaligned access mode)?
// (k_stream idea: give the one accepting state an odd state value so that "state & 1" is the hit flag -- one
//  v_alignbit per step instead of v_cmp + v_addc -- with its table row stored one byte late.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) const uint16_t lds_cu16;
__global__ void k(uint32_t *out, int iters, uint32_t *cyc)
{
   __shared__ __align__(16) uint8_t s[4096];
   for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = (uint8_t)(i * 7 + 3);
   __syncthreads();
   const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)s;
   const uint32_t a = base + threadIdx.x;            // lane l reads at byte offset l: odd lanes misaligned, l % 4 == 3 crosses a dword
   uint32_t v = *(lds_cu16 *)(uintptr_t)a;
   out[threadIdx.x] = v;
   // timing: dependent chain of reads, aligned vs odd addresses
   uint32_t st = (threadIdx.x * 16) & 4095, st2 = ((threadIdx.x * 16) & 4095) | 1;
   uint64_t t0 = clock64();
   for (int i = 0; i < iters; i++) st = (*(lds_cu16 *)(uintptr_t)(base + st)) & 0xFFEu;
   uint64_t t1 = clock64();
   for (int i = 0; i < iters; i++) st2 = ((*(lds_cu16 *)(uintptr_t)(base + st2)) & 0xFFEu) | 1u;
   uint64_t t2 = clock64();
   if (threadIdx.x == 0) { cyc[0] = (uint32_t)(t1 - t0); cyc[1] = (uint32_t)(t2 - t1); }
   out[64 + threadIdx.x] = st + st2;
}
int main()
{
   uint32_t *d, *c, h[128], hc[2];
   hipMalloc(&d, sizeof h); hipMalloc(&c, sizeof hc);
   k<<<1, 64>>>(d, 1000, c);
   hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(hc, c, sizeof hc, hipMemcpyDeviceToHost);
   int bad = 0;
   for (int l = 0; l < 64; l++) {
      const uint32_t exp = (uint32_t)(uint8_t)(l * 7 + 3) | ((uint32_t)(uint8_t)((l + 1) * 7 + 3) << 8);
      if (h[l] != exp) { if (bad < 8) printf("lane %d: got %04x expected %04x\n", l, h[l], exp); bad++; }
   }
   printf("unaligned ds_read_u16: %s (%d of 64 lanes differ); dependent chain of 1000 reads: aligned %u cycles, odd %u cycles\n", bad ? "NOT byte-exact" : "byte-exact", bad, hc[0], hc[1]);
   return 0;
}
