#!/usr/bin/env python3
"""Which part of the workspace does the pairing hang on?  One text, one scan context; between measurements ONLY the hit-list arrays are
re-allocated (seeqdevScanReserve with a growing max_hitlines: the per-wave hit slices `tmp`, the ordered entries, hit_start / line / col) --
the per-tile and per-line arrays stay where they are."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from seeq_amd import device as dev
PATTERN, TAU, L, n = "GATGTAGCGCGATTAGCCTG", 3, 150, 100_000_000
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
x = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda:0")
dev.synth_reads(x.data_ptr(), 0, n, L, dev.plain_pattern(PATTERN), TAU, stream=stream)
torch.cuda.synchronize()
pat = dev.Pattern(PATTERN, TAU)
seg_lines = min(n, 0xF0000000 // (L + 1) + 2)
sc = dev.Scanner(stream)
hl = max(seg_lines // 6 + 1024, 8192 * 64)
sc.reserve(x.numel(), seg_lines + 64, hl, n // 4 + 1024)
sc.set_profiling(True)
def measure():
    fwd = launches = 0.0
    for it in range(5):
        sc.run(pat, x.data_ptr(), x.numel(), dev.SQ_BEST, dev.WANT_RECORDS); sc.fetch()
        if it >= 2:
            tm = sc.last_times_ms(); fwd += tm["forward"]; launches += tm["forward_launches"]
    return round(fwd / launches, 4)
row = [measure()]
for k in range(1, 10):
    sc.reserve(x.numel(), seg_lines + 64, hl + k * 8192 * 8, n // 4 + 1024)      # only the hit-list arrays grow (and move)
    row.append(measure())
print("hit-list arrays re-allocated between measurements:", row, flush=True)
row2 = [measure() for _ in range(4)]
print("nothing re-allocated:", row2, flush=True)
