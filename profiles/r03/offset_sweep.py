#!/usr/bin/env python3
"""Does the scan kernel's time depend on where the text sits?  One process, one 15.1 GB text placed at several offsets inside a
larger allocation; mean launch time of the scan kernel per offset (k_pair's time is bimodal from process to process)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from seeq_amd import device as dev
PATTERN, TAU, L, n = "GATGTAGCGCGATTAGCCTG", 3, 150, 100_000_000
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
nb = n * (L + 1)
big = torch.empty(nb + (1 << 30), dtype=torch.uint8, device="cuda:0")
pat = dev.Pattern(PATTERN, TAU)
print("base address %x" % big.data_ptr())
for rep in range(2):
    for off in (0, 4096, 65536, 1 << 20, 2 << 20, 16 << 20, 256 << 20, (512 << 20) + 8192, 1 << 30):
        text = big[off:off + nb]
        dev.synth_reads(text.data_ptr(), 0, n, L, dev.plain_pattern(PATTERN), TAU, stream=stream)
        torch.cuda.synchronize()
        sc = dev.Scanner(stream)
        seg_lines = min(n, 0xF0000000 // (L + 1) + 2)
        sc.reserve(nb, seg_lines + 64, max(seg_lines // 6 + 1024, 8192 * 64), n // 4 + 1024)
        sc.set_profiling(True)
        fwd = launches = 0.0
        for it in range(8):
            sc.run(pat, text.data_ptr(), nb, dev.SQ_BEST, dev.WANT_RECORDS)
            sc.fetch()
            if it >= 2:
                tm = sc.last_times_ms(); fwd += tm["forward"]; launches += tm["forward_launches"]
        print("rep %d offset %10d  %s launch %.4f ms" % (rep, off, sc.last_kernel(), fwd / launches), flush=True)
        sc.close()
