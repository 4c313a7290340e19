"""Stress of the scan context's bookkeeping: ONE Scanner reused over many buffers / patterns / option sets (so every workspace
array holds stale data of an earlier scan), small segments (several launches per buffer), workspace reserved absurdly small
or not at all (every overflow stage and re-run), short patterns (hits nearly everywhere) and long ones, read-length and long
lines, foreign bytes.  Every result against the oracle.  Usage: SEEQ_SEGMENT_BYTES=65536 python profiles/stress_reuse.py [seed]"""
import os, random, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from oracle.pyoracle import Oracle, SQ_ALL, SQ_BEST, SQ_FIRST
from seeq_amd import device as dev
from test_gpu_parity import _mutate

o = Oracle()
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
sc = dev.Scanner()
n_ok = 0
kernels = {}
for it in range(60):
    m = rng.choice([3, 5, 6, 8, 12, 20, 26, 40])
    pattern = "".join(rng.choice("ACGT") if rng.random() > 0.08 else rng.choice(["N", "[AC]", "[GT]"]) for _ in range(m))
    core = dev.plain_pattern(pattern).replace("N", "A")
    tau = rng.randint(0, min(5, m - 1, max(0, 33 - m) if m <= 30 else 5))
    long_lines = rng.random() < 0.3
    lines = []
    for _ in range(rng.choice([40, 400, 3000]) if not long_lines else 50):
        n = rng.choice([0, 1, 30, 100, 151, 151, 260]) if not long_lines else rng.choice([0, 151, 3000, 8192, 20000, 70000, 140000])
        t = [rng.choice("ACGT") for _ in range(n)]
        for _rep in range(1 + n // 700):
            if n >= m and rng.random() < 0.5:
                c = _mutate(rng, core, rng.randint(0, tau + 2))
                p = rng.randrange(0, n - len(c) + 1) if n >= len(c) else 0
                t[p:p + len(c)] = list(c)
        if rng.random() < 0.05 and n:
            t[rng.randrange(n)] = rng.choice("N!*XH-.\t")
        lines.append("".join(t)[:n])
    buf = ("\n".join(lines) + ("\n" if it % 2 else "")).encode("latin-1")
    pat = dev.Pattern(pattern, tau)
    nd = rng.choice([0, dev.SQ_CONVERT, dev.SQ_IGNORE])
    if rng.random() < 0.5:
        sc._lib.seeqdevScanReserve(sc._h, 0, rng.choice([1, 10, 1000]), rng.choice([1, 2, 100]), rng.choice([1, 5, 100]))
    for mo, want in ((SQ_FIRST, dev.WANT_RECORDS), (SQ_ALL, dev.WANT_RECORDS), (0, dev.WANT_COUNTLINES), (SQ_BEST, dev.WANT_RECORDS), (0, dev.WANT_COUNTMATCH)):
        exp = o.buffer_scan(pattern, tau, buf, (mo if want == dev.WANT_RECORDS else SQ_ALL) | nd)
        got = sc.scan_host(pat, buf, mo | nd, want)
        kernels[sc.last_kernel()] = kernels.get(sc.last_kernel(), 0) + 1
        tag = (it, pattern, tau, long_lines, nd, mo, want, len(buf))
        assert got["nlines"] == exp["nlines"] and got["nmatchlines"] == exp["nmatchlines"], tag
        if want == dev.WANT_RECORDS:
            assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), tag
        if want == dev.WANT_COUNTMATCH:
            assert got["nhits"] == len(exp["records"]), tag
        n_ok += 1
    pat.close()
print("stress OK:", n_ok, "scans", kernels)
