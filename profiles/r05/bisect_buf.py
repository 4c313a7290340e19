#!/usr/bin/env python3
"""Delta-debugging of a failing buffer: keeps the Scanner in the plan the whole buffer put it into (fallbacks are sticky), then looks
for a small contiguous range of lines on which GPU and oracle still differ."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.pyoracle import Oracle, SQ_ALL, SQ_BEST
from seeq_amd import device as dev
buf = open(sys.argv[1], "rb").read(); pattern = sys.argv[2]; tau = int(sys.argv[3]); opts = int(sys.argv[4]); fasta = int(sys.argv[5]) != 0
o = Oracle(); p = dev.Pattern(pattern, tau); sc = dev.Scanner()
FA = dev.SEEQDEV_FASTA if fasta else 0
lines = buf.split(b"\n")
import random
rng = random.Random(1)
core = dev.plain_pattern(pattern).encode()
LONG = b"".join((core if i % 7 == 0 else bytes(rng.choice(b"ACGT") for _ in range(40))) for i in range(700))      # ~27 KB with copies all over: keeps the long-line plan
def differs(ls):
    b = LONG + b"\n" + b"\n".join(ls) + b"\n"
    exp = o.buffer_scan(pattern, tau, b, opts, fasta=fasta)
    got = sc.scan_host(p, b, opts | FA, dev.WANT_RECORDS)
    return (not np.array_equal(got["records"].astype(np.uint64), exp["records"])), got, exp, sc.last_kernel()
d, got, exp, k = differs(lines)
print("whole buffer differs:", d, k, len(lines), "lines")
# keep the alignment: everything in front of the last `keep` lines of the failing range becomes same-length runs of T (no candidate there)
hi = 2473
keep = 2
ls = [b"T" * len(l) if j < hi - keep else l for j, l in enumerate(lines[:hi])]
print("=== debug scan (SQ_ALL | SQ_CONVERT)", flush=True)
d, got, exp, k = differs(ls)
g = {tuple(r) for r in got["records"].astype(np.uint64).tolist()}; e = {tuple(r) for r in exp["records"].tolist()}
print("differs", d, k, "extra", sorted(g - e)[:3], "missing", sorted(e - g)[:3], flush=True)
