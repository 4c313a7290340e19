#!/usr/bin/env python3
"""Does the scan kernel's launch time follow the WORKSPACE's placement as it follows the text buffer's?  One process, two text buffers (plain
allocations), six scan contexts each kept alive (six distinct workspaces): mean launch time of k_pair per (text, context)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from seeq_amd import device as dev
PATTERN, TAU, L, n = "GATGTAGCGCGATTAGCCTG", 3, 150, 100_000_000
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
texts = []
for t in range(2):
    x = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda:0")
    dev.synth_reads(x.data_ptr(), 0, n, L, dev.plain_pattern(PATTERN), TAU, stream=stream)
    texts.append(x)
torch.cuda.synchronize()
pat = dev.Pattern(PATTERN, TAU)
seg_lines = min(n, 0xF0000000 // (L + 1) + 2)
scs = []
for i in range(6):
    sc = dev.Scanner(stream)
    sc.reserve(texts[0].numel(), seg_lines + 64, max(seg_lines // 6 + 1024, 8192 * 64), n // 4 + 1024)
    sc.set_profiling(True)
    scs.append(sc)
for rnd in range(2):
    for ti, x in enumerate(texts):
        row = []
        for sc in scs:
            fwd = launches = 0.0
            for it in range(5):
                sc.run(pat, x.data_ptr(), x.numel(), dev.SQ_BEST, dev.WANT_RECORDS)
                try:
                    sc.fetch()
                except Exception:          # (an experiment library with void results may trip the library's own checks)
                    pass
                if it >= 2:
                    tm = sc.last_times_ms(); fwd += tm["forward"]; launches += tm["forward_launches"]
            row.append(round(fwd / launches, 4))
        print("round", rnd, "text", ti, "launch ms per context:", row, flush=True)
