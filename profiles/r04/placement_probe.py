"""Does k_pair's launch time go with WHERE the text lies?  Three 15.1 GB buffers with the same reads in one process, the same scan over
each, the duration of every launch (four segments per buffer), twice; then the first buffer again after the others were freed."""
import os, sys, json
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
bufs = []
for b in range(3):
    t = torch.empty(nb, dtype=torch.uint8, device="cuda:0")
    dev.synth_reads(t.data_ptr(), 0, n, L, PATTERN, TAU, stream=stream)
    torch.cuda.synchronize()
    bufs.append(t)
def run(t):
    for _ in range(2):
        sc.run(pat, t.data_ptr(), nb, dev.SQ_BEST, dev.WANT_RECORDS); sc.fetch()
    out = []
    for _ in range(3):
        sc.run(pat, t.data_ptr(), nb, dev.SQ_BEST, dev.WANT_RECORDS); sc.fetch()
        out.append([round(x, 3) for x in sc.last_launch_times_ms()])
    return out
for rnd in range(2):
    for b, t in enumerate(bufs):
        print("round", rnd, "buffer", b, "ptr 0x%x" % t.data_ptr(), run(t), "clock", round(sc.last_clock_mhz()), flush=True)
p0 = bufs[0].data_ptr()
del bufs[1:], t
torch.cuda.empty_cache()
print("buffer 0 alone again", run(bufs[0]))
