"""Does physically contiguous memory (hipExtMallocWithFlags + hipDeviceMallocContiguous) give the scan kernel its fast pages every time?
The same reads in: a plain torch allocation, two contiguous allocations (text size, 16 GiB), a 32 GiB torch block, another plain one; k_pair's
time per launch over each, two rounds."""
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import torch
from seeq_amd import device as dev
PATTERN, TAU, L, n = "GATGTAGCGCGATTAGCCTG", 3, 150, 100_000_000
stream = torch.cuda.current_stream().cuda_stream
pat = dev.Pattern(PATTERN, TAU)
sc = dev.Scanner(stream)
sc.set_profiling(True)
nb = n * (L + 1)
seg_lines = min(n, (0xF0000000 // (L + 1)) + 2)
sc.reserve(nb, seg_lines + 64, max(seg_lines // 6 + 1024, 8192 * 64), n // 8 + 1024)
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipFree.argtypes = [C.c_void_p]
t0 = torch.empty(nb, dtype=torch.uint8, device="cuda:0")
dev.synth_reads(t0.data_ptr(), 0, n, L, PATTERN, TAU, stream=stream)
torch.cuda.synchronize()
cands = [("plain torch.empty(text)", t0.data_ptr(), t0)]
for label, size in (("contiguous, text size", nb), ("contiguous, 16 GiB", 16 << 30)):
    p = C.c_void_p()
    rc = hip.hipExtMallocWithFlags(C.byref(p), size, 0x4)
    if rc != 0 or not p.value:
        print(label, ": hipExtMallocWithFlags failed, rc", rc); continue
    rc = hip.hipMemcpy(p, C.c_void_p(t0.data_ptr()), nb, 3)
    cands.append((label, p.value, None))
big = torch.empty(32 << 30, dtype=torch.uint8, device="cuda:0"); big[:nb].copy_(t0); cands.append(("torch 32 GiB block", big.data_ptr(), big))
t5 = torch.empty(nb, dtype=torch.uint8, device="cuda:0"); t5.copy_(t0); cands.append(("plain torch.empty(text) #2", t5.data_ptr(), t5))
torch.cuda.synchronize()
for rnd in range(2):
    for label, ptr, _ in cands:
        for _ in range(3):
            sc.run(pat, ptr, nb, dev.SQ_BEST, dev.WANT_RECORDS); c = sc.fetch()
        print("round %d  %-28s ptr 0x%x  %s  matching %d" % (rnd, label, ptr, [round(x, 3) for x in sc.last_launch_times_ms()], c["nmatchlines"]), flush=True)
