#!/usr/bin/env python3
"""Round 5: a fuzz aimed at the SQ_IGNORE markers of k_pair (seeq_pair.h IG, seeq_order.h) and at its dirty mode -- FASTQ-like text
made of line KINDS rather than of planted foreign bytes: reads, quality-like lines (bytes '!'..'J' with a tunable share of A C G T N
among them, so that the count of characters that are not skipped falls on either side of m - tau), headers, '+' lines, empty lines,
lines longer than the 256 bytes a tile looks ahead and longer than a tile, copies of the pattern with skipped bytes INSIDE them (a hit
under -x 2 only), copies cut by a byte that ends the line under -x 0, patterns poor in one base (the frequency bound of k_bounds2).
Every buffer is scanned under the three non-DNA modes, first / best / all records and both counts, against the oracle, with the
default plan and forced onto k_pair.  Usage: python profiles/ignore_fuzz.py [seed] [buffers]   (prints the seed; replay with it)"""
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.pyoracle import Oracle, SQ_ALL, SQ_BEST, SQ_FIRST      # noqa: E402
from seeq_amd import device as dev                                   # noqa: E402

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else (int(time.time() * 1000) ^ os.getpid()) % 1_000_000_007
nbuf = int(sys.argv[2]) if len(sys.argv) > 2 else 40
print("IGNORE_FUZZ_SEED=%d buffers=%d" % (seed0, nbuf), flush=True)
o = Oracle()
QUAL = "".join(chr(c) for c in range(33, 75))
SKIP = "!#$%&*+-./0123456789:;<=>?@BDEFHIJ"            # never a base, never a line end
tot = 0
bad = 0
kernels = {}


def mutate(rng, s, k):
    s = list(s)
    for _ in range(k):
        r = rng.random()
        if r < 0.5 and s:
            s[rng.randrange(len(s))] = rng.choice("ACGT")
        elif r < 0.75 and len(s) > 1:
            del s[rng.randrange(len(s))]
        else:
            s.insert(rng.randrange(len(s) + 1), rng.choice("ACGT"))
    return "".join(s)


ONLY = os.environ.get("IGNORE_FUZZ_ONLY")                 # replay one buffer of a seed, with a report on the lines that differ


def report(buf, got, exp, m):
    print("nheaders", got.get("nheaders"), exp.get("nheaders"))
    g = {tuple(r) for r in got["records"].astype(np.uint64).tolist()}
    e = {tuple(r) for r in exp["records"].tolist()}
    starts = [0] + [i + 1 for i, c in enumerate(buf) if c == 10]
    for tag, rows in (("MISSING (oracle only)", sorted(e - g)), ("EXTRA (GPU only)", sorted(g - e))):
        for r in rows[:6]:
            ln = int(r[0]); off = starts[ln - 1]; end = buf.find(b"\n", off); end = len(buf) if end < 0 else end
            line = buf[off:end]
            nsk = sum(1 for c in line if chr(c) not in "ACGTUNacgtun")
            print(tag, r, "line offset", off, "= tile", off // 8192, "+", off % 8192, "len", len(line), "skipped bytes", nsk, "bases", len(line) - nsk,
                  "ends in tile", end // 8192, "+", end % 8192, flush=True)
            print("   ", line[:400], flush=True)


for b in range(nbuf):
    if ONLY is not None and b != int(ONLY):
        continue
    rng = random.Random(seed0 * 1000 + b)
    m = rng.choice([8, 12, 16, 20, 20, 24, 30, 30, 36, 44, 62])                    # (> 32: two column words)
    alphabet = rng.choice(["ACGT", "ACGT", "ACG", "AC", "ACGTT", "AAAC"])          # (a pattern poor in a base: the frequency bound bites or not)
    core = "".join(rng.choice(alphabet) for _ in range(m))
    pattern = core
    if rng.random() < 0.25:                                                         # a class / N position or two
        pl = list(core)
        for _ in range(rng.randint(1, 2)):
            i = rng.randrange(m)
            pl[i] = rng.choice(["N", "[AC]", "[GT]", "[ACG]"])
        pattern = "".join(pl)
    tau = rng.randint(0, min(6 if m > 32 else 4, m // 5 + 1))
    lower = rng.choice([0.0, 0.0, 0.02, 0.5])                                       # share of reads written in lower case / with U for T
    crlf = rng.random() < 0.15                                                      # "\r\n" line ends: the \r is one more skipped byte
    high = rng.random() < 0.2                                                       # bytes >= 0x80 among the quality bytes
    align = rng.choice([0, 0, 128, 8192])                                           # now and then a line is stretched so that the next begins a lane / a tile
    qual = QUAL + ("\x80\xa7\xff" if high else "")
    fasta = rng.random() < 0.2                                                      # SEEQDEV_FASTA: lines that start with '>' are headers (seeq.c:367-374)
    pos = 0                                                                         # offset of the line being made
    dna_share = rng.choice([0.0, 0.05, 0.1, 0.15, 0.3, 0.6])                        # share of bases among a quality line's bytes
    lines = []
    nlines = rng.choice([400, 1500, 4000])
    for i in range(nlines):
        kind = rng.random()
        if kind < 0.30:                                                             # a read
            n = rng.choice([50, 100, 150, 150, 151, 250])
            t = [rng.choice("ACGT") for _ in range(n)]
        elif kind < 0.60:                                                           # a quality-like line
            n = rng.choice([50, 100, 150, 150, 151, 250])
            t = [rng.choice("ACGTN") if rng.random() < dna_share else rng.choice(qual) for _ in range(n)]
        elif kind < 0.70:                                                           # header
            t = list("%sr%09d %s" % (">" if fasta else "@", i, "".join(rng.choice("acgtnACGTlength=xyz0123") for _ in range(rng.randint(0, 40)))))
            if fasta and rng.random() < 0.3:
                t = list(">") + list(mutate(rng, core, 0)) + t[1:]                  # a header that holds the pattern: never a hit line
        elif kind < 0.78:
            t = list("+")
        elif kind < 0.80:
            t = []
        elif kind < 0.90:                                                           # longer than the look-ahead / than a tile, mixed
            # (few of the very long ones: the planner keeps k_pair for buffers whose sampled lines average <= 600 bytes)
            n = rng.choice([5000, 9000, 20000]) if rng.random() < 0.02 else rng.choice([300, 300, 600, 1200])
            sh = rng.choice([0.0, 0.02, 0.5, 1.0])
            t = [rng.choice("ACGT") if rng.random() >= sh * 0.2 else rng.choice(SKIP) for _ in range(n)]
        else:                                                                       # short lines of anything
            n = rng.randint(1, 30)
            t = [rng.choice("ACGTN" + SKIP) for _ in range(n)]
        n = len(t)
        # copies of the pattern: plain, with skipped bytes inside, cut by a skipped byte
        for _ in range(rng.choice([0, 0, 1, 1, 2])):
            if n < m + 8:
                break
            c = list(mutate(rng, core, rng.randint(0, tau + 1)))
            r = rng.random()
            if r < 0.45:
                for _ in range(rng.randint(1, 4)):
                    c.insert(rng.randrange(1, len(c)), rng.choice(SKIP))             # skipped bytes INSIDE the copy
            elif r < 0.55:
                c.insert(rng.randrange(len(c) + 1), rng.choice("\0" + SKIP))
            p = rng.randrange(0, max(1, n - len(c)))
            t[p:p + len(c)] = c
            t = t[:n]
        if rng.random() < 0.004 and n:
            t[rng.randrange(n)] = "\0"
        if kind < 0.30 and rng.random() < lower:
            t = [c.lower() if rng.random() < 0.7 else ("U" if c == "T" else c) for c in t]
        if align and rng.random() < 0.08:                                           # stretch: the NEXT line starts on a multiple of `align`
            room = (-(pos + len(t) + 1 + (1 if crlf else 0))) % align
            if room < 700:
                t += [rng.choice("ACGT" + (SKIP if rng.random() < 0.5 else "")) for _ in range(room)]
        if crlf:
            t.append("\r")
        lines.append("".join(t))
        pos += len(t) + 1
    buf = ("\n".join(lines) + ("\n" if b % 2 else "")).encode("latin-1")
    for forced in (None, "pair"):
        if forced:
            os.environ["SEEQ_FUSED_KERNEL"] = forced
        else:
            os.environ.pop("SEEQ_FUSED_KERNEL", None)
        if b % 4 == 3 or os.environ.get("IGNORE_FUZZ_SEGMENTS") == "1":             # (IGNORE_FUZZ_SEGMENTS=1: every buffer)
            os.environ["SEEQ_SEGMENT_BYTES"] = rng.choice(["65536", "65536", "16384", "131072"])   # lines and markers across segment seams
        else:
            os.environ.pop("SEEQ_SEGMENT_BYTES", None)
        p = dev.Pattern(pattern, tau)
        sc = dev.Scanner()
        FA = dev.SEEQDEV_FASTA if fasta else 0
        for nd in (dev.SQ_IGNORE, 0, dev.SQ_CONVERT):
            for mo in (SQ_BEST, SQ_ALL, SQ_FIRST):
                exp = o.buffer_scan(pattern, tau, buf, mo | nd, fasta=fasta)
                got = sc.scan_host(p, buf, mo | nd | FA, dev.WANT_RECORDS)
                kernels[sc.last_kernel()] = kernels.get(sc.last_kernel(), 0) + 1
                ctx = (seed0, b, pattern, tau, mo, nd, forced, os.environ.get("SEEQ_SEGMENT_BYTES"))
                if ONLY is not None and got["nlines"] != exp["nlines"]:
                    bad += 1
                    print("NLINES", ctx, sc.last_kernel(), "bytes", len(buf), "lines", got["nlines"], exp["nlines"], "last bytes", buf[-40:], flush=True)
                if ONLY is not None and not np.array_equal(got["records"].astype(np.uint64), exp["records"]):
                    print("DIFF", ctx, sc.last_kernel(), "bytes", len(buf), "lines", got["nlines"], exp["nlines"], "matching", got["nmatchlines"], exp["nmatchlines"], flush=True)
                    report(buf, got, exp, m)
                    bad += 1
                    continue
                assert got["nlines"] == exp["nlines"] and got["nmatchlines"] == exp["nmatchlines"], ("counts", ctx, got["nmatchlines"], exp["nmatchlines"], got["nlines"], exp["nlines"])
                if not np.array_equal(got["records"].astype(np.uint64), exp["records"]):
                    g, e = got["records"].astype(np.uint64), exp["records"]
                    k = 0
                    while k < min(len(g), len(e)) and np.array_equal(g[k], e[k]):
                        k += 1
                    raise AssertionError(("records", ctx, len(g), len(e), k, g[k:k + 3].tolist(), e[k:k + 3].tolist()))
                tot += 1
            expa = o.buffer_scan(pattern, tau, buf, SQ_ALL | nd, fasta=fasta)
            c1 = sc.scan_host(p, buf, nd | FA, dev.WANT_COUNTLINES)
            c2 = sc.scan_host(p, buf, nd | FA, dev.WANT_COUNTMATCH)
            assert c1["nmatchlines"] == expa["nmatchlines"] and c1["nlines"] == expa["nlines"], ("countlines", seed0, b, pattern, tau, nd, forced)
            assert c2["nhits"] == len(expa["records"]) and c2["nmatchlines"] == expa["nmatchlines"], ("countmatch", seed0, b, pattern, tau, nd, forced)
        # several patterns in one walk over the same buffer (seeqdevScanHostMulti): the pattern, a neighbour of it, a random one
        if forced is None and ONLY is None:
            others = [mutate(rng, core, 2)[:m] or core, "".join(rng.choice("ACGT") for _ in range(max(8, m - 4)))]
            pats = [p] + [dev.Pattern(q, min(tau, max(0, len(q) // 5))) for q in others]
            specs = [(pattern, tau)] + [(q, min(tau, max(0, len(q) // 5))) for q in others]
            for nd in (dev.SQ_IGNORE, 0, dev.SQ_CONVERT):
                res = sc.scan_host_multi(pats, buf, SQ_BEST | nd | FA, dev.WANT_RECORDS)
                for (q, tq), r in zip(specs, res):
                    exp = o.buffer_scan(q, tq, buf, SQ_BEST | nd, fasta=fasta)
                    assert r["nlines"] == exp["nlines"] and r["nmatchlines"] == exp["nmatchlines"], ("multi counts", seed0, b, q, tq, nd, sc.last_multi_one_pass(), r["nmatchlines"], exp["nmatchlines"])
                    assert np.array_equal(r["records"].astype(np.uint64), exp["records"]), ("multi records", seed0, b, q, tq, nd, sc.last_multi_one_pass())
                    tot += 1
                kernels["multi one pass" if sc.last_multi_one_pass() else "multi per pattern"] = kernels.get("multi one pass" if sc.last_multi_one_pass() else "multi per pattern", 0) + 1
            for q in pats[1:]:
                q.close()
        sc.close(); p.close()
if bad:
    print("ignore fuzz FAILED:", bad, "scans differ")
    sys.exit(1)
print("ignore fuzz OK:", tot, "record scans", kernels)
