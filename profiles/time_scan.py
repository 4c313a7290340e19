#!/usr/bin/env python3
"""Times the scan kernel of the headline workload without checking any result (for the timing-only experiment variants
of k_pair, SEEQ_PAIR_EXP, whose counts are void).  Prints: label, kernel, mean launch ms of the scan kernel, ms per step.
Usage: python profiles/time_scan.py <label> [reads] [steps] [mode: best|count|all]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from seeq_amd import device as dev  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "run"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
mode = sys.argv[4] if len(sys.argv) > 4 else "best"
PATTERN, TAU, L = os.environ.get("SEEQ_TS_PATTERN", "GATGTAGCGCGATTAGCCTG"), int(os.environ.get("SEEQ_TS_TAU", "3")), int(os.environ.get("SEEQ_TS_LEN", "150"))
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
text = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda:0")
dev.synth_reads(text.data_ptr(), 0, n, L, dev.plain_pattern(PATTERN)[:96], TAU, stream=stream)      # (the generator plants copies of at most 96 positions)
torch.cuda.synchronize()
pat = dev.Pattern(PATTERN, TAU)
sc = dev.Scanner(stream)
seg_lines = min(n, 0xF0000000 // (L + 1) + 2)
sc.reserve(text.numel(), seg_lines + 64, max(seg_lines // 6 + 1024, 8192 * 64), n // 4 + 1024)
sc.set_profiling(True)
opt = {"best": dev.SQ_BEST, "count": 0, "all": dev.SQ_ALL}[mode]
want = dev.WANT_COUNTLINES if mode == "count" else dev.WANT_RECORDS
fwd = launches = ex = 0.0
cnt = None
for it in range(steps + 3):
    if it == 3:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    sc.run(pat, text.data_ptr(), text.numel(), opt, want)
    try:
        cnt = sc.fetch()
    except Exception as e:          # (an experiment variant with void counts may trip the library's own checks)
        print(label, "fetch failed:", e)
        sys.exit(0)
    if it >= 3:
        tm = sc.last_times_ms()
        fwd += tm["forward"]
        launches += tm["forward_launches"]
        ex += tm["exact"]
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("%-16s %-8s launch %.4f ms  post-pass %.3f ms/step  step %.3f ms  matchlines %d" % (label, sc.last_kernel(), fwd / max(1, launches), ex / steps, 1e3 * el / steps, cnt["nmatchlines"]))
