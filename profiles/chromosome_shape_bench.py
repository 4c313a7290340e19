#!/usr/bin/env python3
"""The reference's own published benchmark shape (doc/response.tex:181-187): a genome with one
chromosome per line -- here 24 lines x 128 MiB of uniform random DNA (3.2 GB) with 58 planted approximate copies of
the 20-mer, d = 3.  Reports device-resident scan times for -c, hit counting, --best and --all with positions, checks
the records against the oracle on a 64 MiB two-line sample, and (optionally) times the reference binary on the
same bytes.  Usage: python profiles/chromosome_shape_bench.py [lines] [MiB per line] [--ref]"""
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seeq_amd import device as dev                       # noqa: E402

PATTERN, TAU = "GATGTAGCGCGATTAGCCTG", 3
nlines = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 24
mib = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 128
L = mib << 20
d = torch.device("cuda:0")
g = torch.Generator(device=d); g.manual_seed(2025)
lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=d)
text = torch.empty(nlines * L, dtype=torch.uint8, device=d)
for i in range(nlines):                                  # line by line: bounded temporaries
    text[i * L:(i + 1) * L] = lut[torch.randint(0, 4, (L,), device=d, generator=g, dtype=torch.uint8).long()]
    text[(i + 1) * L - 1] = 10
rng = np.random.default_rng(7)
pat = np.frombuffer(PATTERN.encode(), dtype=np.uint8)
for _ in range(58):                                      # planted copies with 0..3 substitutions
    p = int(rng.integers(100, nlines * L - 100))
    c = pat.copy()
    for _e in range(int(rng.integers(0, TAU + 1))):
        c[int(rng.integers(0, len(c)))] = b"ACGT"[int(rng.integers(0, 4))]
    if (p % L) < L - 40:
        text[p:p + len(c)] = torch.from_numpy(c).to(d)
FA = 0
if "--fasta" in sys.argv:                                # a header line in front of every chromosome (SEEQDEV_FASTA)
    FA = dev.SEEQDEV_FASTA
    for i in range(nlines):
        text[i * L:i * L + 7] = torch.from_numpy(np.frombuffer((">chr%02d\n" % i).encode(), dtype=np.uint8).copy()).to(d)
torch.cuda.synchronize()

P = dev.Pattern(PATTERN, TAU)
sc = dev.Scanner()
sc.set_profiling(True)
out = {"shape": "%d lines x %d MiB random DNA, 58 planted copies%s" % (nlines, mib, ", FASTA headers" if FA else ""), "bytes": int(text.numel())}
for name, opt, want in (("count_lines", 0, dev.WANT_COUNTLINES), ("count_hits", 0, dev.WANT_COUNTMATCH),
                        ("best", dev.SQ_BEST, dev.WANT_RECORDS), ("all", dev.SQ_ALL, dev.WANT_RECORDS)):
    sc.scan_tensor(P, text, opt | FA, want)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cnt = sc.scan_tensor(P, text, opt | FA, want)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out[name] = {"ms": dt * 1e3, "gb_per_s": text.numel() / dt / 1e9, "matching_lines": int(cnt["nmatchlines"]),
                 "hits": int(cnt["nhits"]), "records": int(cnt["nrecords"]), "kernel": sc.last_kernel(), "times_ms": sc.last_times_ms()}
# parity on a sample: the first two lines (bounded to 64 MiB each) against the oracle
from oracle.pyoracle import Oracle                        # noqa: E402
k = min(L, 64 << 20)
sample = torch.cat([text[L - k:L], text[2 * L - k:2 * L]]).contiguous()
host = sample.cpu().numpy()
exp = Oracle().buffer_scan(PATTERN, TAU, host, dev.SQ_ALL)
s2 = dev.Scanner()
got = s2.scan_tensor(P, sample, dev.SQ_ALL, dev.WANT_RECORDS)      # (sample without the header bytes: plain lines)
out["oracle_sample_check"] = bool(np.array_equal(s2.records(got["nrecords"]).astype(np.uint64), exp["records"]))
if "--ref" in sys.argv and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "seeq_ref")):
    f = "/dev/shm/chrom.txt"
    text.cpu().numpy().tofile(f)
    t0 = time.perf_counter()
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "seeq_ref"), "-d", str(TAU), "-a", "-f", PATTERN, f],
                       capture_output=True, text=True)
    out["reference_cpu_all"] = {"seconds": time.perf_counter() - t0, "records": len(r.stdout.splitlines())}
    os.remove(f)
print(json.dumps(out))
