"""k_pair's tile -> wave map (SEEQ_PAIR_MAP = 0 interleaved over the grid / 1 a contiguous range per wave / 2 per workgroup) over three buffers
of one process: per-launch durations and the records' checksum.

(SEEQ_PAIR_MAP existed in an experiment build of round 4 only -- three lines in k_pair's tile loop: `tile = gwave * per + i` / `blockIdx.x * per + i * NW + wave`
instead of `gwave + i * nwaves` -- and was taken out again with this result, map_probe.txt; on HEAD the script measures map 0 four times.)"""
import os, sys, zlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from seeq_amd import device as dev
PATTERN, TAU, L, n = "GATGTAGCGCGATTAGCCTG", 3, 150, 100_000_000
stream = torch.cuda.current_stream().cuda_stream
pat = dev.Pattern(PATTERN, TAU)
nb = n * (L + 1)
seg_lines = min(n, (0xF0000000 // (L + 1)) + 2)
bufs = []
for b in range(3):
    t = torch.empty(nb, dtype=torch.uint8, device="cuda:0")
    dev.synth_reads(t.data_ptr(), 0, n, L, PATTERN, TAU, stream=stream)
    torch.cuda.synchronize()
    bufs.append(t)
for mp in ("0", "1", "2", "0"):
    os.environ["SEEQ_PAIR_MAP"] = mp
    sc = dev.Scanner(stream)
    sc.set_profiling(True)
    sc.reserve(nb, seg_lines + 64, max(seg_lines // 6 + 1024, 8192 * 64), n // 8 + 1024)
    for b, t in enumerate(bufs):
        for _ in range(2):
            sc.run(pat, t.data_ptr(), nb, dev.SQ_BEST, dev.WANT_RECORDS); c = sc.fetch()
        ms = []
        for _ in range(3):
            sc.run(pat, t.data_ptr(), nb, dev.SQ_BEST, dev.WANT_RECORDS); c = sc.fetch()
            ms.append([round(x, 3) for x in sc.last_launch_times_ms()])
        rec = sc.records(c["nrecords"])
        print("map", mp, "buffer", b, ms[-1], "mean of full segments %.3f" % np.mean([x for m_ in ms for x in m_[:3]]), c["nmatchlines"], zlib.crc32(rec.tobytes()), sc.last_kernel(), flush=True)
    sc.close()
