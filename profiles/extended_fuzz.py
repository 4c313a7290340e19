import os, sys, random, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))

from oracle.pyoracle import Oracle, SQ_ALL, SQ_BEST, SQ_FIRST
from seeq_amd import device as dev
from test_gpu_parity import _mutate
o = Oracle()
LONG = len(sys.argv) > 1 and sys.argv[1] == "long"      # long lines: the LL variant of k_stream + the window walk of k_exact1
tot = 0; ks = {}
BASE = int(os.environ.get("FUZZ_SEED", "100"))          # FUZZ_SEED=<n>: another set of 12 seeds
FOREIGN = float(os.environ.get("FUZZ_FOREIGN", "0"))    # FUZZ_FOREIGN=<p>: every seed plants foreign bytes, in a fraction p of the lines (and a few NULs)
for seed in range(BASE, BASE + 12):
    rng = random.Random(seed)
    for it in range(25):
        m = rng.choice([4, 6, 9, 12, 15, 18, 20, 22, 25, 28, 30])
        parts, plain = [], []
        for _ in range(m):
            r = rng.random()
            if r < 0.06: parts.append("N"); plain.append("N")
            elif r < 0.15:
                cls = "".join(sorted(set(rng.choice("ACGT") for _ in range(rng.randint(1, 3))))); parts.append("[" + cls + "]"); plain.append(cls[0])
            else:
                c = rng.choice("ACGT"); parts.append(c); plain.append(c)
        pattern, core = "".join(parts), "".join(plain)
        tau = rng.randint(0, min(4, m - 1, 33 - m))
        lines = []
        for _ in range(120 if LONG else 3000):
            n = rng.choice([0, 151, 2000, 8191, 8192, 8300, 20000, 70000]) if LONG else rng.choice([0, 2, 19, 50, 100, 151, 151, 151, 260, 700])
            t = [rng.choice("ACGT") for _ in range(n)]
            for _rep in range(1 + (n // 900 if LONG else 0)):
                if n >= m and rng.random() < (0.8 if LONG else 0.35):
                    c = _mutate(rng, core.replace("N", "A"), rng.randint(0, tau + 2))
                    p = rng.randrange(0, n - len(c) + 1) if n >= len(c) else 0
                    t[p:p + len(c)] = list(c)
            if rng.random() < 0.03 and n: t[rng.randrange(n)] = "N"
            if seed % 3 == 0 and rng.random() < (0.3 if LONG else 0.01) and n: t[rng.randrange(n)] = rng.choice("!*+BJXZ.\t\r@>")
            if FOREIGN and n:
                for _rep in range(rng.choice([1, 1, 2, 5])):
                    if rng.random() < FOREIGN: t[rng.randrange(n)] = rng.choice("!*+BJXZH-.\t\r@")
                if rng.random() < FOREIGN / 20: t[rng.randrange(n)] = "\0"
            lines.append("".join(t)[:n])
        fasta = seed % 4 == 1
        if fasta:
            lines = [(">h%d " % i + l[:30]) if rng.random() < 0.3 else l for i, l in enumerate(lines)]
        buf = ("\n".join(lines) + ("\n" if it % 2 else "")).encode("latin-1")
        p = dev.Pattern(pattern, tau); sc = dev.Scanner()
        nd = [0, dev.SQ_CONVERT, dev.SQ_IGNORE][seed % 3] if not fasta else 0
        for mo in (SQ_FIRST, SQ_BEST, SQ_ALL):
            exp = o.buffer_scan(pattern, tau, buf, mo | nd, fasta=fasta)
            got = sc.scan_host(p, buf, mo | nd | (dev.SEEQDEV_FASTA if fasta else 0), dev.WANT_RECORDS)
            ks[sc.last_kernel()] = ks.get(sc.last_kernel(), 0) + 1
            assert got["nlines"] == exp["nlines"] and got["nmatchlines"] == exp["nmatchlines"], (seed, it, pattern, tau, mo, nd, fasta)
            assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (seed, it, pattern, tau, mo, nd, fasta)
            tot += 1
        expa = o.buffer_scan(pattern, tau, buf, SQ_ALL | nd, fasta=fasta)
        c1 = sc.scan_host(p, buf, nd | (dev.SEEQDEV_FASTA if fasta else 0), dev.WANT_COUNTLINES)
        c2 = sc.scan_host(p, buf, nd | (dev.SEEQDEV_FASTA if fasta else 0), dev.WANT_COUNTMATCH)
        assert c1["nmatchlines"] == expa["nmatchlines"] and c1["nlines"] == expa["nlines"], (seed, it, pattern, tau)
        assert c2["nhits"] == len(expa["records"]) and c2["nmatchlines"] == expa["nmatchlines"], (seed, it, pattern, tau)
        sc.close(); p.close()
print("extended fuzz OK:", tot, "record scans", ks)
