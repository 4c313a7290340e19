#!/usr/bin/env python3
"""Shape Q of SURVEY 8d (secondary workload): 4-line FASTQ records scanned as plain lines (-x 0).

    @r000000123            <- header: starts with '@', dead after one byte
    <150 bp read>          <- shape-R read (same generator as bench.py)
    +
    <150 Phred+33 bytes>   <- '!'..'J'; may start with A/C/G and alias onto the automaton's columns

Built on the device with torch (plumbing), scanned through the C-ABI; a 200 k-line prefix is checked against the
oracle.  Usage: python profiles/fastq_shape_bench.py [records] [count|best]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seeq_amd import device as dev                       # noqa: E402

PATTERN, TAU, L = "GATGTAGCGCGATTAGCCTG", 3, 150
nrec = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
mode = sys.argv[2] if len(sys.argv) > 2 else "best"
shape = sys.argv[3] if len(sys.argv) > 3 else "fastq"       # "fastq" (4-line records) or "fasta" (">id" + read, SEEQDEV_FASTA)
nondna = {"fail": 0, "convert": dev.SQ_CONVERT, "ignore": dev.SQ_IGNORE}[sys.argv[4] if len(sys.argv) > 4 else "fail"]   # -x 0 / 1 / 2
d = torch.device("cuda:0")
reads = torch.empty(nrec * (L + 1), dtype=torch.uint8, device=d)
dev.synth_reads(reads.data_ptr(), 0, nrec, L, PATTERN, TAU)
torch.cuda.synchronize()
HDR = 12                                                 # "@r%09d\n"
REC = HDR + (L + 1) + 2 + (L + 1)
buf = torch.empty((nrec, REC), dtype=torch.uint8, device=d)
idx = torch.arange(nrec, device=d, dtype=torch.int64)
buf[:, 0] = ord("@"); buf[:, 1] = ord("r")
for k in range(9):
    buf[:, 2 + k] = (48 + (idx // (10 ** (8 - k))) % 10).to(torch.uint8)
buf[:, 11] = 10
buf[:, HDR:HDR + L + 1] = reads.view(nrec, L + 1)
buf[:, HDR + L + 1] = ord("+"); buf[:, HDR + L + 2] = 10
g = torch.Generator(device=d); g.manual_seed(7)
buf[:, HDR + L + 3:HDR + L + 3 + L] = torch.randint(33, 75, (nrec, L), device=d, generator=g, dtype=torch.uint8)
buf[:, REC - 1] = 10
if shape == "fasta":
    buf[:, 0] = ord(">")
    text = buf[:, :HDR + L + 1].contiguous().view(-1)
    REC = HDR + L + 1
else:
    text = buf.view(-1)
del reads
torch.cuda.synchronize()

pat = dev.Pattern(PATTERN, TAU)
sc = dev.Scanner()
sc.set_profiling(True)
opt, want = (0, dev.WANT_COUNTLINES) if mode == "count" else (dev.SQ_BEST, dev.WANT_RECORDS)
FA = dev.SEEQDEV_FASTA if shape == "fasta" else 0
opt |= FA | nondna
for _ in range(2):
    cnt = sc.scan_tensor(pat, text, opt, want)
torch.cuda.synchronize()
t0 = time.perf_counter()
steps = 5
for _ in range(steps):
    cnt = sc.scan_tensor(pat, text, opt, want)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
# oracle check on a prefix
from oracle.pyoracle import Oracle                        # noqa: E402
k = 50_000
host = text[:k * REC].cpu().numpy()
exp = Oracle().buffer_scan(PATTERN, TAU, host, opt & ~FA, fasta=bool(FA))
c2 = dev.Scanner().scan_tensor(pat, text[:k * REC], opt, want)
ok = c2["nmatchlines"] == exp["nmatchlines"] and c2["nlines"] == exp["nlines"]
if want == dev.WANT_RECORDS:
    s2 = dev.Scanner(); c3 = s2.scan_tensor(pat, text[:k * REC], opt, want)
    ok = ok and np.array_equal(s2.records(c3["nrecords"]).astype(np.uint64), exp["records"])
print(json.dumps({"shape": "Q (4-line FASTQ records)" if shape == "fastq" else "2-line FASTA records", "mode": mode, "nondna": sys.argv[4] if len(sys.argv) > 4 else "fail", "records": nrec, "lines": int(cnt["nlines"]),
                  "bytes": int(text.numel()), "ms_per_step": dt * 1e3, "lines_per_s": cnt["nlines"] / dt,
                  "gb_per_s": text.numel() / dt / 1e9, "matching_lines": int(cnt["nmatchlines"]),
                  "kernel": sc.last_kernel(), "times_ms": sc.last_times_ms(),
                  "oracle_prefix_check": bool(ok)}))
