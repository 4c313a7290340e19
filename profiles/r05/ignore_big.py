#!/usr/bin/env python3
"""Round 5: SQ_IGNORE on k_pair at a size where the hit slices, the marker entries and the workspace regrowth matter -- 2 M lines of
FASTQ-like records whose quality lines hold a share of bases (0.5: every one of them gets a marker and is scanned whole; 0.1: few do),
with copies of the pattern planted in reads and, with skipped bytes inside them, in quality lines; records of --best / --all and both
counts against the oracle.  numpy makes the text.  Usage: python profiles/r05/ignore_big.py [records] [share]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Oracle, SQ_ALL, SQ_BEST
from seeq_amd import device as dev

nrec = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
share = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 12345)
pattern, tau, L = "GATGTAGCGCGATTAGCCTG", 3, 150
pat = np.frombuffer(pattern.encode(), np.uint8)
REC = 12 + (L + 1) + 2 + (L + 1)
buf = np.empty((nrec, REC), np.uint8)
buf[:, 0] = ord("@"); buf[:, 1] = ord("r")
idx = np.arange(nrec)
for k in range(9):
    buf[:, 2 + k] = 48 + (idx // 10 ** (8 - k)) % 10
buf[:, 11] = 10
bases = np.frombuffer(b"ACGT", np.uint8)
buf[:, 12:12 + L] = bases[rng.integers(0, 4, (nrec, L))]
buf[:, 12 + L] = 10
buf[:, 13 + L] = ord("+"); buf[:, 14 + L] = 10
q0 = 15 + L
qual = rng.integers(33, 75, (nrec, L)).astype(np.uint8)
isb = rng.random((nrec, L)) < share
qual[isb] = np.frombuffer(b"ACGTN", np.uint8)[rng.integers(0, 5, int(isb.sum()))]
buf[:, q0:q0 + L] = qual
buf[:, REC - 1] = 10
# copies in 8 % of the reads (0 .. 2 substitutions)
rows = np.nonzero(rng.random(nrec) < 0.08)[0]
pos = rng.integers(0, L - 20, rows.size)
for r, p in zip(rows, pos):
    c = pat.copy()
    for _ in range(rng.integers(0, 3)):
        c[rng.integers(0, 20)] = bases[rng.integers(0, 4)]
    buf[r, 12 + p:12 + p + 20] = c
# copies with two skipped bytes inside in 5 % of the quality lines
rows = np.nonzero(rng.random(nrec) < 0.05)[0]
pos = rng.integers(0, L - 24, rows.size)
for r, p in zip(rows, pos):
    c = list(pat)
    for _ in range(2):
        c.insert(int(rng.integers(1, len(c))), int(rng.choice(np.frombuffer(b"!#*+:;5<", np.uint8))))
    buf[r, q0 + p:q0 + p + 22] = np.array(c, np.uint8)
text = buf.reshape(-1).tobytes()
print("text", len(text), "bytes", 4 * nrec, "lines, share", share, flush=True)
o = Oracle(); p = dev.Pattern(pattern, tau); sc = dev.Scanner()
for nd in (dev.SQ_IGNORE, 0):
    for mo in (SQ_BEST, SQ_ALL):
        t0 = time.time(); exp = o.buffer_scan(pattern, tau, text, mo | nd); t1 = time.time()
        got = sc.scan_host(p, text, mo | nd, dev.WANT_RECORDS); t2 = time.time()
        same = np.array_equal(got["records"].astype(np.uint64), exp["records"])
        print("nd", nd, "mo", mo, sc.last_kernel(), "lines", got["nlines"], exp["nlines"], "matching", got["nmatchlines"], exp["nmatchlines"], "records", len(got["records"]), len(exp["records"]),
              "identical", same, "oracle %.1f s, gpu call %.2f s" % (t1 - t0, t2 - t1), flush=True)
        assert same and got["nlines"] == exp["nlines"] and got["nmatchlines"] == exp["nmatchlines"]
    c1 = sc.scan_host(p, text, nd, dev.WANT_COUNTLINES); c2 = sc.scan_host(p, text, nd, dev.WANT_COUNTMATCH)
    expa = o.buffer_scan(pattern, tau, text, SQ_ALL | nd)
    assert c1["nmatchlines"] == expa["nmatchlines"] and c2["nhits"] == len(expa["records"]), (nd, c1, c2["nhits"], len(expa["records"]))
print("ignore big OK")
