/*
 * vmm_probe.hip -- round 5, the ONE bounded placement experiment the round-4 review asked for: is text mapped through the HIP
 * virtual-memory API (hipMemAddressReserve + hipMemCreate + hipMemMap at the recommended granularity) fast for k_pair where a
 * plain hipMalloc is not?  One process = one verdict: the headline text (100 M x 151 B) in (a) a plain hipMalloc, (b) ONE physical
 * allocation of the whole size mapped into a reserved range, (c) the range mapped from chunks of `chunk` bytes (argv[1] MiB,
 * default 1024), each scanned three times with the headline pattern through the library; prints the scan kernel's time per buffer.
 *
 *   hipcc --offload-arch=gfx950 -O2 -I include profiles/r05/vmm_probe.hip -L seeq_amd/lib -lseeq_amd -Wl,-rpath,$PWD/seeq_amd/lib -o gpurun_out/vmm_probe
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "seeq_amd.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static int scan_ms(seeqdev_scan_t *sc, seeqdev_pattern_t *pat, void *buf, uint64_t nreads, float *fwd, float *per_launch, int *nl)
{
   static const char plain[] = "GATGTAGCGCGATTAGCCTG";
   if (seeqdevSynthReads(buf, 0, nreads, 150, plain, 20, 3, 0x5EE92025ull, NULL) || hipStreamSynchronize(NULL) != hipSuccess) return 1;
   for (int rep = 0; rep < 3; rep++) {
      seeqdev_counts_t cnt;
      if (seeqdevScanRun(sc, pat, buf, (size_t)nreads * 151, 1 /* SQ_BEST */, 2 /* records */) || seeqdevScanFetch(sc, &cnt)) return 1;
   }
   float t[4];
   if (seeqdevScanLastTimes(sc, t)) return 1;
   *fwd = t[1];
   *nl = seeqdevScanLastLaunchTimes(sc, per_launch, 8);
   return 0;
}

int main(int argc, char **argv)
{
   const size_t chunk_mib = argc > 1 ? (size_t)atol(argv[1]) : 1024;
   const uint64_t nreads = 100000000ull;
   const size_t bytes = nreads * 151;
   static const char plain[] = "GATGTAGCGCGATTAGCCTG";
   char keys[20];
   for (int i = 0; i < 20; i++) keys[i] = plain[i] == 'A' ? 1 : plain[i] == 'C' ? 2 : plain[i] == 'G' ? 4 : 8;
   seeqdev_pattern_t *pat = seeqdevPatternNew(keys, 20, 3);
   seeqdev_scan_t *sc = pat ? seeqdevScanNew(NULL) : NULL;
   if (!sc || seeqdevScanSetProfiling(sc, 1)) { fprintf(stderr, "setup: %s\n", seeqdevLastError()); return 1; }
   int dev = 0;
   CK(hipGetDevice(&dev));
   hipMemAllocationProp prop;
   memset(&prop, 0, sizeof prop);
   prop.type = hipMemAllocationTypePinned;
   prop.location.type = hipMemLocationTypeDevice;
   prop.location.id = dev;
   size_t gran_min = 0, gran_rec = 0;
   CK(hipMemGetAllocationGranularity(&gran_min, &prop, hipMemAllocationGranularityMinimum));
   CK(hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended));
   hipMemAccessDesc acc;
   memset(&acc, 0, sizeof acc);
   acc.location = prop.location;
   acc.flags = hipMemAccessFlagsProtReadWrite;
   float fwd, pl[8];
   int nl;
   printf("{\"granularity_min\": %zu, \"granularity_recommended\": %zu, \"chunk_mib\": %zu", gran_min, gran_rec, chunk_mib);
   /* (a) the plain allocation: what a caller that hipMallocs once gets */
   void *plainbuf = nullptr;
   CK(hipMalloc(&plainbuf, bytes));
   if (scan_ms(sc, pat, plainbuf, nreads, &fwd, pl, &nl)) { fprintf(stderr, "scan: %s\n", seeqdevLastError()); return 1; }
   printf(", \"plain_hipMalloc\": {\"forward_ms\": %.4f, \"launch_ms\": [%.4f, %.4f, %.4f, %.4f]}", fwd, pl[0], pl[1], pl[2], pl[3]);
   /* (b), (c): a reserved range mapped from one / from many physical allocations */
   for (int mode = 0; mode < 2; mode++) {
      const size_t chunk = mode == 0 ? (bytes + gran_rec - 1) / gran_rec * gran_rec : ((chunk_mib << 20) + gran_rec - 1) / gran_rec * gran_rec;
      const size_t total = (bytes + chunk - 1) / chunk * chunk;
      void *va = nullptr;
      CK(hipMemAddressReserve(&va, total, gran_rec, nullptr, 0));
      std::vector<hipMemGenericAllocationHandle_t> hs;
      for (size_t off = 0; off < total; off += chunk) {
         hipMemGenericAllocationHandle_t h;
         CK(hipMemCreate(&h, chunk, &prop, 0));
         CK(hipMemMap((char *)va + off, chunk, 0, h, 0));
         hs.push_back(h);
      }
      CK(hipMemSetAccess(va, total, &acc, 1));
      if (scan_ms(sc, pat, va, nreads, &fwd, pl, &nl)) { fprintf(stderr, "scan (vmm): %s\n", seeqdevLastError()); return 1; }
      printf(", \"%s\": {\"chunks\": %zu, \"chunk_bytes\": %zu, \"forward_ms\": %.4f, \"launch_ms\": [%.4f, %.4f, %.4f, %.4f]}", mode == 0 ? "vmm_one_allocation" : "vmm_chunks",
             hs.size(), chunk, fwd, pl[0], pl[1], pl[2], pl[3]);
      CK(hipDeviceSynchronize());
      CK(hipMemUnmap(va, total));
      for (auto h : hs) CK(hipMemRelease(h));
      CK(hipMemAddressFree(va, total));
   }
   printf("}\n");
   CK(hipFree(plainbuf));
   seeqdevScanFree(sc);
   seeqdevPatternFree(pat);
   return 0;
}
