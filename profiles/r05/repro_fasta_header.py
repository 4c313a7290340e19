#!/usr/bin/env python3
"""Repro: FASTA input, long-line plan (a long line with copies all over in front) -- a hit inside a header line."""
import os, sys, random, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.pyoracle import Oracle, SQ_ALL, SQ_BEST, SQ_FIRST
from seeq_amd import device as dev
o = Oracle()
rng = random.Random(3)
pattern, tau = "CAACCCCAACACCACAACCAAAAA", 4
def dna(n): return "".join(rng.choice("ACGT") for _ in range(n))
LONG = "".join((pattern if i % 7 == 0 else dna(40)) for i in range(700))
cases = {
  "hit line + header with hit": [dna(19) + pattern + dna(107), ">r000002472CA" + pattern + "AAAa2hnaGna83.62AG6H00D"],
  "plain line + header with hit": [dna(150), ">r000002472CA" + pattern + "AAAa2hnaGna83.62AG6H00D"],
  "hit line + header without": [dna(19) + pattern + dna(107), ">r000002472 plain"],
  "hit line + plain line with hit": [dna(19) + pattern + dna(107), "Ar000002472CA".replace("r", "C").replace("0", "A").replace("2", "G").replace("4", "T").replace("7", "C") + pattern + "AAAA"],
}
for name, tail in cases.items():
    lines = [LONG] + [dna(150) for _ in range(20)] + tail + [dna(150) for _ in range(5)]
    buf = ("\n".join(lines) + "\n").encode()
    p = dev.Pattern(pattern, tau); sc = dev.Scanner()
    for nd in (dev.SQ_CONVERT, 0):
        for mo in (SQ_ALL, SQ_BEST, SQ_FIRST):
            exp = o.buffer_scan(pattern, tau, buf, mo | nd, fasta=True)
            got = sc.scan_host(p, buf, mo | nd | dev.SEEQDEV_FASTA, dev.WANT_RECORDS)
            g = [tuple(r) for r in got["records"].astype(np.uint64).tolist() if r[0] > 1]; e = [tuple(r) for r in exp["records"].tolist() if r[0] > 1]
            print("%-32s nd %d mo %d %-8s %s   gpu %s   oracle %s" % (name, nd, mo, sc.last_kernel(), "same" if g == e else "DIFFERENT", g, e), flush=True)
    sc.close(); p.close()
