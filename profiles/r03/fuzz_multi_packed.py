"""Fresh-seed campaign for the two entries added in round 3: the one-walk multi-pattern scan (random barcode sets of 2..32
patterns, 6..14 positions, classes and N, distance 0..2, foreign bytes, SQ_FAIL / SQ_CONVERT, FASTA now and then) against the
oracle per pattern, and the packed scan (random pattern, distance, read length 1..256, N, every match option) against the
oracle over the same reads as text.  Usage (GPU box): python3 profiles/r03/fuzz_multi_packed.py [rounds] [seed]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch
from oracle.pyoracle import Oracle, SQ_ALL, SQ_BEST, SQ_FIRST
from seeq_amd import device as dev
from test_gpu_parity import _mutate

o = Oracle()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
print("seed0", seed0, flush=True)
one_walk = per_pattern = packed_ok = 0
for rd in range(rounds):
    rng = random.Random(seed0 + rd)
    # ---- multi ----
    npat = rng.choice([2, 3, 5, 8, 16, 16, 24, 32])
    pats, taus = [], []
    for _ in range(npat):
        m = rng.choice([6, 8, 8, 9, 10, 10, 11, 12, 14])
        parts = []
        for _i in range(m):
            r = rng.random()
            parts.append("N" if r < 0.04 else "[" + "".join(sorted(set(rng.choice("ACGT") for _ in range(2)))) + "]" if r < 0.10 else rng.choice("ACGT"))
        pats.append("".join(parts)); taus.append(rng.choice([0, 1, 1, 1, 2]))
    fasta = rng.random() < 0.2
    nd = rng.choice([0, 0, dev.SQ_CONVERT])
    foreign = rng.random() < 0.4
    lines = []
    for i in range(rng.choice([3000, 20000])):
        n = rng.choice([0, 30, 100, 150, 151, 300])
        t = [rng.choice("ACGT") for _ in range(n)]
        for _ in range(rng.choice([0, 1, 1, 2])):
            k = rng.randrange(npat)
            c = _mutate(rng, dev.plain_pattern(pats[k]).replace("N", "A"), rng.randint(0, taus[k] + 1))
            if n >= len(c):
                q = rng.choice([0, n - len(c), rng.randrange(n - len(c) + 1)]); t[q:q + len(c)] = list(c)
        if rng.random() < 0.03 and n: t[rng.randrange(n)] = "N"
        if foreign and rng.random() < 0.03 and n: t[rng.randrange(n)] = rng.choice("!*XZ-.\t")
        line = "".join(t)[:n]
        if fasta and rng.random() < 0.2: line = ">" + line[:40]
        lines.append(line)
    buf = ("\n".join(lines) + ("\n" if rng.random() < 0.5 else "")).encode()
    P = [dev.Pattern(b, t) for b, t in zip(pats, taus)]
    sc = dev.Scanner()
    fl = dev.SEEQDEV_FASTA if fasta else 0
    for opt, want in ((SQ_BEST, dev.WANT_RECORDS), (SQ_ALL, dev.WANT_RECORDS), (SQ_FIRST, dev.WANT_RECORDS), (0, dev.WANT_COUNTLINES), (0, dev.WANT_COUNTMATCH)):
        got = sc.scan_host_multi(P, buf, opt | nd | fl, want)
        if sc.last_multi_one_pass(): one_walk += 1
        else: per_pattern += 1
        for k in range(npat):
            exp = o.buffer_scan(pats[k], taus[k], buf, (opt if want == dev.WANT_RECORDS else SQ_ALL) | nd, fasta=fasta)
            assert got[k]["nlines"] == exp["nlines"] and got[k]["nmatchlines"] == exp["nmatchlines"], ("multi", seed0 + rd, k, opt, want, pats, taus)
            if want == dev.WANT_RECORDS:
                assert np.array_equal(got[k]["records"].astype(np.uint64), exp["records"]), ("multi rec", seed0 + rd, k, opt, pats, taus)
            if want == dev.WANT_COUNTMATCH:
                assert got[k]["nhits"] == len(exp["records"]), ("multi hits", seed0 + rd, k)
    sc.close()
    for p in P: p.close()
    # ---- packed ----
    m = rng.choice([4, 7, 10, 16, 20, 24, 31, 40])
    pattern = "".join("N" if rng.random() < 0.05 else rng.choice("ACGT") for _ in range(m))
    tau = rng.randint(0, min(5, m - 2))
    L = rng.choice([1, 3, 16, 37, 64, 100, 150, 151, 200, 255, 256])
    core = dev.plain_pattern(pattern).replace("N", "A")
    rl = []
    for i in range(4000):
        t = "".join(rng.choice("ACGT") for _ in range(L))
        if rng.random() < 0.4 and L >= len(core):
            c = _mutate(rng, core, rng.randint(0, tau + 2)); q = rng.randrange(max(1, L - len(c) + 1)); t = (t[:q] + c + t[q + len(c):])[:L]
        if rng.random() < 0.1: q = rng.randrange(L); t = t[:q] + "N" + t[q + 1:]
        rl.append(t)
    text = ("\n".join(rl) + "\n").encode()
    try:
        pat = dev.Pattern(pattern, tau)
        bases, nmask, n = dev.pack_reads(text, L)
        db, dn = torch.from_numpy(bases.copy()).cuda(), torch.from_numpy(nmask.copy()).cuda()
        sc = dev.Scanner()
        for mo in (SQ_FIRST, SQ_BEST, SQ_ALL):
            try:
                sc.run_packed(pat, db.data_ptr(), dn.data_ptr(), n, L, options=mo, want=dev.WANT_RECORDS)
            except dev.SeeqDeviceError:
                break                                      # (no pair automaton for this pattern: the entry says so)
            cnt = sc.fetch(); rec = sc.records(cnt["nrecords"])
            exp = o.buffer_scan(pattern, tau, text, mo)
            assert cnt["nmatchlines"] == exp["nmatchlines"] and np.array_equal(rec.astype(np.uint64), exp["records"]), ("packed", seed0 + rd, pattern, tau, L, mo)
            packed_ok += 1
        sc.close(); pat.close()
    except dev.SeeqDeviceError:
        pass
    print("round", rd, "ok", flush=True)
print("CAMPAIGN OK: multi scans on one walk %d, per pattern %d; packed scans %d" % (one_walk, per_pattern, packed_ok))
