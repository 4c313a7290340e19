"""At what granularity does a buffer's speed vary, and does a plain read see it too?  Two buffers with the same reads (a plain 15.1 GB
allocation, a 32 GiB block), cut into pieces of 3.5 M lines (528 MB): per piece k_pair's launch time (one scan per piece, the kernel's own
duration) beside a plain reduction over the same bytes (torch.sum over int64 words), three rounds."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from seeq_amd import device as dev
PATTERN, TAU, L, n = "GATGTAGCGCGATTAGCCTG", 3, 150, 100_000_000
stream = torch.cuda.current_stream().cuda_stream
pat = dev.Pattern(PATTERN, TAU)
sc = dev.Scanner(stream)
sc.set_profiling(True)
nb = n * (L + 1)
PL = 3_500_000                     # lines per piece (a multiple of 8: the piece's bytes are whole int64 words)
pb = PL * (L + 1)
sc.reserve(pb, PL + 64, max(PL // 6 + 1024, 8192 * 64), PL // 8 + 1024)
bufs = []
for size in (nb, 32 << 30):
    big = torch.empty(size, dtype=torch.uint8, device="cuda:0")
    t = big[:nb]
    if bufs:
        t.copy_(bufs[0][1])
    else:
        dev.synth_reads(t.data_ptr(), 0, n, L, PATTERN, TAU, stream=stream)
    torch.cuda.synchronize()
    bufs.append((big, t))
npieces = n // PL
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for bi, (big, t) in enumerate(bufs):
    for rnd in range(3):
        kp, sm = [], []
        for p in range(npieces):
            sub = t[p * pb:(p + 1) * pb]
            for _ in range(2):
                sc.run(pat, sub.data_ptr(), pb, dev.SQ_BEST, dev.WANT_RECORDS); sc.fetch()
            kp.append(sc.last_launch_times_ms()[0])
            w = sub.view(torch.int64)
            w.sum(); torch.cuda.synchronize()
            ev0.record(); w.sum(); ev1.record(); torch.cuda.synchronize()
            sm.append(ev0.elapsed_time(ev1))
        print("buffer %d (%d GiB block) round %d" % (bi, big.numel() >> 30, rnd))
        print("   k_pair us per piece :", " ".join("%4.0f" % (1e3 * x) for x in kp))
        print("   plain sum us        :", " ".join("%4.0f" % (1e3 * x) for x in sm), flush=True)
