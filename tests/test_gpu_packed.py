"""GPU (-m gpu): packed read batches (include/seeq_amd.h: seeqdev_packed_t; seeq_amd/csrc/seeq_packed.h) -- 2 bits per base,
one read per lane, no warm-up -- against the oracle over the same reads as ASCII text, one read per line: every match
option, both counts, read lengths from 1 to 256 (odd lengths, lengths that are no multiple of four), N through the mask,
patterns served by a prefix automaton and by a partition filter, workspace growth (every read a candidate)."""
import random
import sys

import numpy as np
import pytest

from conftest import GOLDEN
from oracle.pyoracle import SQ_ALL, SQ_BEST, SQ_FIRST

pytestmark = pytest.mark.gpu


def _packed_scan(dev, torch, pat, text, read_len, opt, want, with_n=True):
    bases, nmask, n = dev.pack_reads(text, read_len, with_n)
    db = torch.from_numpy(bases.copy()).cuda()
    dn = torch.from_numpy(nmask.copy()).cuda() if nmask is not None else None
    sc = dev.Scanner()
    sc.run_packed(pat, db.data_ptr(), dn.data_ptr() if dn is not None else None, n, read_len, options=opt, want=want)
    cnt = sc.fetch()
    res = dict(cnt)
    if want == dev.WANT_RECORDS:
        res["records"] = sc.records(cnt["nrecords"])
    res["kernel"] = sc.last_kernel()
    res["quad"] = sc.last_packed_quad()
    sc.close()
    return res


def test_packed_vs_oracle(gpu, capi, oracle):
    import torch
    from seeq_amd import device as dev
    sys.path.insert(0, GOLDEN)
    from make_golden import mutate, plain
    rng = random.Random(31)
    cases = [("GATGTAGCGCGATTAGCCTG", 3, 150), ("GATGTAGCGCGATTAGCCTG", 3, 149), ("GATTAGC", 1, 37), ("CACAGAT", 3, 50), ("ACGT", 1, 7),
             ("AC", 0, 1), ("ACNNGT[AC]TTG", 2, 100), ("GATG[TA]AGCNCGATTAGC[CG]TGAAAATGNGAGTAC[GAT]GCGCGA", 5, 250), ("GATGTAGCGCGATTAGCCTGAAAA", 3, 256),
             ("AAAAAAAAAAAAAAAAAAAA", 3, 151), ("GATGTAGCGCGATTAG", 4, 63)]
    nquad = 0
    for pattern, tau, L in cases:
        core = plain(pattern).replace("N", "A")
        lines = []
        for i in range(3000):
            t = "".join(rng.choice("ACGT") for _ in range(L))
            if i % 3 == 0 and L >= len(core):
                cp = mutate(rng, core, rng.randint(0, tau + 2))
                q = rng.choice([0, max(0, L - len(cp)), rng.randrange(max(1, L - len(cp) + 1))])
                t = (t[:q] + cp + t[q + len(cp):])[:L]
            if i % 11 == 0:
                q = rng.randrange(L)
                t = t[:q] + "N" + t[q + 1:]
            if i % 37 == 0:
                t = t.lower()
            lines.append(t)
        text = ("\n".join(lines) + ("\n" if rng.random() < 0.5 else "")).encode()
        pat = dev.Pattern(pattern, tau)
        import os
        for mo in (SQ_FIRST, SQ_BEST, SQ_ALL):
            exp = oracle.buffer_scan(pattern, tau, text, mo)
            got = _packed_scan(dev, torch, pat, text, L, mo, dev.WANT_RECORDS)
            assert got["kernel"] == "k_packed"
            assert got["nlines"] == exp["nlines"] == len(lines) and got["nmatchlines"] == exp["nmatchlines"], (pattern, tau, L, mo)
            assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (pattern, tau, L, mo)
            if got["quad"]:
                # four bases per table step (the quad table of a small partition filter): the same through the pair table
                nquad += mo == SQ_BEST
                os.environ["SEEQ_PACKED_QUAD"] = "0"
                try:
                    g2 = _packed_scan(dev, torch, pat, text, L, mo, dev.WANT_RECORDS)
                finally:
                    os.environ.pop("SEEQ_PACKED_QUAD", None)
                assert not g2["quad"] and np.array_equal(g2["records"].astype(np.uint64), exp["records"]), (pattern, tau, L, mo, "pair table")
        if (pattern, tau) == ("GATGTAGCGCGATTAGCCTG", 3):
            assert got["quad"], "the headline pattern has a 95-state two-part filter: the quad table serves it"
        expa = oracle.buffer_scan(pattern, tau, text, SQ_ALL)
        c1 = _packed_scan(dev, torch, pat, text, L, 0, dev.WANT_COUNTLINES)
        c2 = _packed_scan(dev, torch, pat, text, L, 0, dev.WANT_COUNTMATCH)
        assert c1["nmatchlines"] == expa["nmatchlines"] and c1["nlines"] == expa["nlines"], (pattern, tau, L)
        assert c2["nhits"] == len(expa["records"]) and c2["nmatchlines"] == expa["nmatchlines"], (pattern, tau, L)
        pat.close()
    assert nquad >= 3, nquad


def test_packed_every_read_a_candidate_and_no_nmask(gpu, capi, oracle):
    """Workspace growth (the optimistic hit-list capacity is one read in eight: here every read holds the pattern) and a
    batch without an N mask."""
    import torch
    from seeq_amd import device as dev
    rng = random.Random(5)
    pattern, tau, L = "GATTAGCCTG", 1, 60
    lines = []
    for _ in range(40000):
        q = rng.randrange(L - len(pattern) + 1)
        t = "".join(rng.choice("ACGT") for _ in range(L))
        lines.append(t[:q] + pattern + t[q + len(pattern):])
    text = ("\n".join(lines) + "\n").encode()
    pat = dev.Pattern(pattern, tau)
    exp = oracle.buffer_scan(pattern, tau, text, SQ_BEST)
    got = _packed_scan(dev, torch, pat, text, L, SQ_BEST, dev.WANT_RECORDS, with_n=False)
    assert got["nmatchlines"] == exp["nmatchlines"] == len(lines)
    assert np.array_equal(got["records"].astype(np.uint64), exp["records"])
    with pytest.raises(dev.SeeqDeviceError):
        dev.pack_reads(b"ACGN\n", 4, with_nmask=False)        # an N without a mask to put it in
    with pytest.raises(dev.SeeqDeviceError):
        dev.pack_reads(b"ACGT\nACG\n", 4)                     # a line of another length
    pat.close()


def test_pack_on_device_equals_host_packer_and_padded_strides(gpu, capi, oracle):
    """seeqdevPackReadsDevice against seeqdevPackReads byte for byte; a batch whose strides are larger than they need to be
    (reads padded to 16 bytes, masks to 8) gives the same records."""
    import torch
    from seeq_amd import device as dev
    rng = random.Random(77)
    for L in (1, 5, 8, 33, 150, 255):
        lines = ["".join(rng.choice("ACGTN" if rng.random() < 0.1 else "ACGT") for _ in range(L)) for _ in range(1000)]
        text = ("\n".join(lines) + "\n").encode()
        bases, nmask, n = dev.pack_reads(text, L)
        t = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
        db = torch.zeros(n * ((L + 3) // 4), dtype=torch.uint8, device="cuda:0")
        dn = torch.zeros(n * ((L + 7) // 8), dtype=torch.uint8, device="cuda:0")
        dev.pack_reads_device(t.data_ptr(), n, L, db.data_ptr(), dn.data_ptr())
        torch.cuda.synchronize()
        tail = (1 << (2 * ((4 - L % 4) % 4))) - 1                     # padding bits of a read's last byte are unspecified
        hb, gb = bases.reshape(n, -1).copy(), db.cpu().numpy().reshape(n, -1).copy()
        hb[:, -1] &= 0xFF ^ tail; gb[:, -1] &= 0xFF ^ tail
        assert np.array_equal(hb, gb), L
        ntail = (1 << ((8 - L % 8) % 8)) - 1
        hn, gn = nmask.reshape(n, -1).copy(), dn.cpu().numpy().reshape(n, -1).copy()
        hn[:, -1] &= 0xFF ^ ntail; gn[:, -1] &= 0xFF ^ ntail
        assert np.array_equal(hn, gn), L
    # padded strides
    pattern, tau, L = "GATTAGCCTG", 1, 50
    lines = []
    for i in range(5000):
        t_ = "".join(rng.choice("ACGT") for _ in range(L))
        if i % 4 == 0:
            q = rng.randrange(L - len(pattern) + 1); t_ = t_[:q] + pattern + t_[q + len(pattern):]
        if i % 13 == 0:
            q = rng.randrange(L); t_ = t_[:q] + "N" + t_[q + 1:]
        lines.append(t_)
    text = ("\n".join(lines) + "\n").encode()
    bases, nmask, n = dev.pack_reads(text, L)
    stride, nstride = 16, 8
    pb = np.full((n, stride), 0xA5, dtype=np.uint8); pb[:, :(L + 3) // 4] = bases.reshape(n, -1)
    pn = np.full((n, nstride), 0x5A, dtype=np.uint8); pn[:, :(L + 7) // 8] = nmask.reshape(n, -1)
    db, dn = torch.from_numpy(pb.reshape(-1)).cuda(), torch.from_numpy(pn.reshape(-1)).cuda()
    pat = dev.Pattern(pattern, tau)
    sc = dev.Scanner()
    sc.run_packed(pat, db.data_ptr(), dn.data_ptr(), n, L, stride=stride, nstride=nstride, options=SQ_ALL, want=dev.WANT_RECORDS)
    cnt = sc.fetch()
    exp = oracle.buffer_scan(pattern, tau, text, SQ_ALL)
    assert cnt["nmatchlines"] == exp["nmatchlines"] and np.array_equal(sc.records(cnt["nrecords"]).astype(np.uint64), exp["records"])
    sc.close(); pat.close()


def test_packed_two_segments_equal_the_ascii_scan(gpu, capi):
    """More reads than one packed segment holds (64 Mi by default; 8 Mi here: three segments): 20 M reads of 24 bases, the packed
    scan against the ASCII scan of the same reads on the GPU (the ASCII path is the one the oracle pins at this size elsewhere) --
    counts and every record."""
    import os
    import torch
    from seeq_amd import device as dev
    n, L = 20_000_000, 24
    os.environ["SEEQ_PACKED_SEG_READS"] = str(1 << 23)
    pattern, tau = "GATTAGCC", 1
    stream = torch.cuda.current_stream().cuda_stream
    text = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda:0")
    dev.synth_reads(text.data_ptr(), 0, n, L, pattern, tau, stream=stream)
    db = torch.empty(n * 6, dtype=torch.uint8, device="cuda:0"); dn = torch.empty(n * 3, dtype=torch.uint8, device="cuda:0")
    dev.pack_reads_device(text.data_ptr(), n, L, db.data_ptr(), dn.data_ptr(), stream=stream)
    pat = dev.Pattern(pattern, tau)
    try:
        sc = dev.Scanner(stream)
    finally:
        os.environ.pop("SEEQ_PACKED_SEG_READS", None)
    sc.run(pat, text.data_ptr(), text.numel(), SQ_BEST, dev.WANT_RECORDS)
    a = sc.fetch(); ra = sc.records(a["nrecords"])
    sc.run_packed(pat, db.data_ptr(), dn.data_ptr(), n, L, options=SQ_BEST, want=dev.WANT_RECORDS)
    b = sc.fetch(); rb = sc.records(b["nrecords"])
    assert sc.last_kernel() == "k_packed"
    assert a == b and a["nlines"] == n and a["nmatchlines"] > n // 50
    assert np.array_equal(ra, rb)
    assert int(rb[:, 0].max()) > (1 << 24)                      # records of the second segment are there
    sc.close(); pat.close()


def test_packed_patterns_the_walk_does_not_serve(gpu, capi, oracle):
    """No refusal (the reference takes any pattern, libseeq.c:43-138): a 70-position pattern (beyond the two-word column) and patterns
    without a pair automaton are served by unpacking the batch on the device and scanning that text -- same records as the oracle's."""
    import torch
    from seeq_amd import device as dev
    sys.path.insert(0, GOLDEN)
    from make_golden import mutate
    rng = random.Random(77)
    long_pat = "".join(rng.choice("ACGT") for _ in range(70))
    for pattern, tau, L in ((long_pat, 6, 150), ("ACG", 2, 40), ("ACGTA", 4, 33), (long_pat[:64] + "N[AC]", 3, 101)):
        core = pattern.replace("N", "A").replace("[AC]", "C")
        lines = []
        for i in range(1500):
            t = "".join(rng.choice("ACGT") for _ in range(L))
            if i % 3 == 0 and L >= len(core):
                cp = mutate(rng, core, rng.randint(0, tau + 2))
                q = rng.randrange(max(1, L - len(cp) + 1))
                t = (t[:q] + cp + t[q + len(cp):])[:L]
            if i % 9 == 0:
                q = rng.randrange(L)
                t = t[:q] + "N" + t[q + 1:]
            lines.append(t)
        text = ("\n".join(lines) + "\n").encode()
        pat = dev.Pattern(pattern, tau)
        for mo in (SQ_FIRST, SQ_BEST, SQ_ALL):
            exp = oracle.buffer_scan(pattern, tau, text, mo)
            got = _packed_scan(dev, torch, pat, text, L, mo, dev.WANT_RECORDS)
            assert got["kernel"] != "k_packed", (pattern, got["kernel"])                 # the fall-back ran
            assert got["nlines"] == exp["nlines"] == len(lines) and got["nmatchlines"] == exp["nmatchlines"], (pattern, tau, L, mo)
            assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (pattern, tau, L, mo)
        c2 = _packed_scan(dev, torch, pat, text, L, 0, dev.WANT_COUNTMATCH)
        expa = oracle.buffer_scan(pattern, tau, text, SQ_ALL)
        assert c2["nhits"] == len(expa["records"]) and c2["nmatchlines"] == expa["nmatchlines"], (pattern, tau, L)
        pat.close()


def test_packed_record_offsets_are_those_of_the_ascii_form(gpu, capi):
    """seeqdevScanCopyOffsets after a packed scan: per record the offset its read has in the ASCII form of the batch -- what the
    ASCII scan of the same reads reports -- not an offset into the scan's private staging text."""
    import torch
    from seeq_amd import device as dev
    n, L = 300_000, 75
    pattern, tau = "GATGTAGCGCGATTAGCCTG", 3
    stream = torch.cuda.current_stream().cuda_stream
    text = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda:0")
    dev.synth_reads(text.data_ptr(), 0, n, L, pattern, tau, stream=stream)
    db = torch.empty(n * ((L + 3) // 4), dtype=torch.uint8, device="cuda:0"); dn = torch.empty(n * ((L + 7) // 8), dtype=torch.uint8, device="cuda:0")
    dev.pack_reads_device(text.data_ptr(), n, L, db.data_ptr(), dn.data_ptr(), stream=stream)
    pat = dev.Pattern(pattern, tau)
    sc = dev.Scanner(stream)
    for mo in (SQ_BEST, SQ_ALL, SQ_FIRST):
        sc.run(pat, text.data_ptr(), text.numel(), mo, dev.WANT_RECORDS)
        a = sc.fetch(); ra = sc.records(a["nrecords"]); oa = sc.record_offsets(a["nrecords"])
        sc.run_packed(pat, db.data_ptr(), dn.data_ptr(), n, L, options=mo, want=dev.WANT_RECORDS)
        b = sc.fetch(); rb = sc.records(b["nrecords"]); ob = sc.record_offsets(b["nrecords"])
        assert sc.last_kernel() == "k_packed" and a == b and a["nrecords"] > 1000
        assert np.array_equal(ra, rb)
        assert np.array_equal(oa, ob), mo
        assert np.array_equal(ob, (rb[:, 0].astype(np.uint64) - 1) * (L + 1))
    sc.close(); pat.close()


def _packed_fuzz(dev, torch, oracle, seed, ncases, nreads=1500):
    """Random patterns (classes, N, 4 .. 44 positions, distance 0 .. 5), random read lengths, reads with planted mutated copies, N and lower
    case: the packed scan against the oracle for every match option and both counts.  Returns how many cases walked the quad table."""
    sys.path.insert(0, GOLDEN)
    from make_golden import mutate, plain
    rng = random.Random(seed)
    nquad = 0
    for case in range(ncases):
        m = rng.choice([4, 6, 8, 10, 12, 16, 20, 20, 20, 24, 31, 36, 44])
        pattern = "".join("N" if rng.random() < 0.04 else "[" + "".join(sorted(rng.sample("ACGT", 2))) + "]" if rng.random() < 0.06
                          else rng.choice("ACGT") for _ in range(m))
        if rng.random() < 0.15:
            unit = "".join(rng.choice("ACGT") for _ in range(rng.choice([1, 2, 3])))
            pattern = (unit * m)[:m]                         # periodic patterns: occurrences overlap, walks restart inside occurrences
        tau = rng.randint(0, min(5, m - 2))
        L = rng.choice([m, m + 1, 37, 50, 75, 100, 149, 150, 151, 200, 250, 256])
        L = max(L, 1)
        core = plain(pattern).replace("N", "A")
        lines = []
        for i in range(nreads):
            t = "".join(rng.choice("ACGT") for _ in range(L))
            if i % 3 == 0 and L >= len(core):
                cp = mutate(rng, core, rng.randint(0, tau + 2))
                q = rng.choice([0, max(0, L - len(cp)), rng.randrange(max(1, L - len(cp) + 1))])
                t = (t[:q] + cp + t[q + len(cp):])[:L]
                if rng.random() < 0.2 and L >= 2 * len(core):     # a second copy: several candidates per read
                    q2 = rng.randrange(L - len(core) + 1)
                    t = (t[:q2] + core + t[q2 + len(core):])[:L]
            if i % 13 == 0:
                q = rng.randrange(L)
                t = t[:q] + "N" + t[q + 1:]
            if i % 41 == 0:
                t = t.lower()
            lines.append(t)
        text = ("\n".join(lines) + "\n").encode()
        pat = dev.Pattern(pattern, tau)
        for mo in (SQ_FIRST, SQ_BEST, SQ_ALL):
            exp = oracle.buffer_scan(pattern, tau, text, mo)
            got = _packed_scan(dev, torch, pat, text, L, mo, dev.WANT_RECORDS)
            assert got["nmatchlines"] == exp["nmatchlines"], (seed, case, pattern, tau, L, mo, got["kernel"], got["quad"])
            assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (seed, case, pattern, tau, L, mo, got["kernel"], got["quad"])
        nquad += bool(got["quad"])
        expa = oracle.buffer_scan(pattern, tau, text, SQ_ALL)
        c2 = _packed_scan(dev, torch, pat, text, L, 0, dev.WANT_COUNTMATCH)
        assert c2["nhits"] == len(expa["records"]) and c2["nmatchlines"] == expa["nmatchlines"], (seed, case, pattern, tau, L)
        pat.close()
    return nquad


def test_packed_fuzz(gpu, capi, oracle):
    """The packed paths (pair table, quad table, device unpack for patterns neither serves) on random patterns and read lengths: one fixed
    seed and one fresh one per run (SEEQ_FUZZ_SEED replays it)."""
    import os
    import time
    import torch
    from seeq_amd import device as dev
    nq = _packed_fuzz(dev, torch, oracle, 20261004, 30)
    assert nq >= 4, nq                                       # (the quad table serves a share of them)
    env = os.environ.get("SEEQ_FUZZ_SEED")
    seed = int(env) if env else (int(time.time() * 1000) ^ os.getpid()) % 1_000_000_007
    print("SEEQ_FUZZ_SEED=%d" % seed)
    _packed_fuzz(dev, torch, oracle, seed, 40)


def test_text_alloc_picks_a_buffer_by_measurement(gpu, capi):
    """seeqdevTextAlloc (resident text placed by measurement): three candidates are probed, the returned buffer holds
    text like any other -- the same reads scanned in it and in a torch allocation give the same counts and records; a small request or
    candidates = 1 is a plain allocation."""
    import torch
    from seeq_amd import device as dev
    n, L = 2_000_000, 150
    pattern, tau = "GATGTAGCGCGATTAGCCTG", 3
    nb = n * (L + 1)
    buf = dev.TextBuffer(nb, candidates=3)
    assert buf.ptr and len(buf.probe_ms) == 3 and all(t > 0 for t in buf.probe_ms)
    # what the caller got (seeqdevTextAllocInfo): the candidate kept is the fastest, its block may be larger than asked (a power of two), the call's peak is reported
    assert buf.probe_ms[buf.chosen] == min(buf.probe_ms) and buf.allocated_bytes >= nb and buf.allocated_bytes in (nb, 1 << (nb - 1).bit_length())
    assert buf.probe_peak_bytes >= 3 * nb
    view = buf.tensor()
    assert view.data_ptr() == buf.ptr and view.numel() == nb and view.dtype == torch.uint8
    stream = torch.cuda.current_stream().cuda_stream
    ref = torch.empty(nb, dtype=torch.uint8, device="cuda:0")
    dev.synth_reads(ref.data_ptr(), 0, n, L, pattern, tau, stream=stream)
    dev.synth_reads(buf.ptr, 0, n, L, pattern, tau, stream=stream)
    torch.cuda.synchronize()
    pat = dev.Pattern(pattern, tau)
    sc = dev.Scanner(stream)
    sc.run(pat, ref.data_ptr(), nb, SQ_BEST, dev.WANT_RECORDS)
    a = sc.fetch(); ra = sc.records(a["nrecords"])
    sc.run(pat, buf.ptr, nb, SQ_BEST, dev.WANT_RECORDS)
    b = sc.fetch(); rb = sc.records(b["nrecords"])
    assert a == b and a["nmatchlines"] > n // 50 and np.array_equal(ra, rb)
    # seeqdevTextAllocFor (round 5): the candidates probed with the caller's own scan context -- the launch time is a property of the pair
    # text buffer / workspace -- which stays usable, profiling as it was, and scans the chosen buffer like any other
    sc.set_profiling(False)
    mine = dev.TextBuffer(nb, candidates=3, scanner=sc)
    assert mine.ptr and len(mine.probe_ms) == 3 and all(t > 0 for t in mine.probe_ms) and mine.probe_ms[mine.chosen] == min(mine.probe_ms)
    dev.synth_reads(mine.ptr, 0, n, L, pattern, tau, stream=stream)
    torch.cuda.synchronize()
    sc.run(pat, mine.ptr, nb, SQ_BEST, dev.WANT_RECORDS)
    c = sc.fetch(); rc_ = sc.records(c["nrecords"])
    assert a == c and np.array_equal(ra, rc_)
    mine.free()
    sc.close(); pat.close()
    buf.free()
    small = dev.TextBuffer(1 << 20, candidates=8)
    plain = dev.TextBuffer(nb, candidates=1)
    assert small.ptr and plain.ptr and small.probe_ms == [] and plain.probe_ms == []
    assert small.chosen == 0 and plain.allocated_bytes == nb
    small.free(); plain.free()


def test_packed_segment_where_every_read_is_a_candidate(gpu, capi, monkeypatch):
    """64 Mi reads per packed segment met a 32-bit limit of the ASCII staging text -- cap_hitlines x pitch above
    4 GiB was refused with E2BIG, at read_len 150 from 26.8 M hit-list entries, whether the staging text was used or not.  30 M reads that
    ALL carry the pattern in one segment: --best (the exact pass reads its windows from the batch: no staging text, no limit) and --all
    records (staging text in use: the segment is cut to what the text can address) both return one record per read, no E2BIG."""
    import torch
    from seeq_amd import device as dev
    monkeypatch.setenv("SEEQ_PACKED_SEG_READS", str(1 << 25))           # 33.5 M reads per segment: more than the 26.8 M lines a staging text can hold
    n, L = 30_000_000, 150
    pattern, tau = "GATGTAGCGCGATTAGCCTG", 3
    line = (pattern + "T" * (L - len(pattern))).encode() + b"\n"
    one = torch.frombuffer(bytearray(line), dtype=torch.uint8).to("cuda:0")
    text = one.repeat(n)
    stream = torch.cuda.current_stream().cuda_stream
    pb = torch.empty(n * ((L + 3) // 4), dtype=torch.uint8, device="cuda:0")
    pn = torch.empty(n * ((L + 7) // 8), dtype=torch.uint8, device="cuda:0")
    dev.pack_reads_device(text.data_ptr(), n, L, pb.data_ptr(), pn.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    del text
    pat = dev.Pattern(pattern, tau)
    sc = dev.Scanner(stream)
    for opt in (SQ_BEST, SQ_ALL):
        sc.run_packed(pat, pb.data_ptr(), pn.data_ptr(), n, L, options=opt, want=dev.WANT_RECORDS)
        c = sc.fetch()
        assert c["nlines"] == n and c["nmatchlines"] == n and c["nrecords"] == n, (opt, c)
        rec = sc.records(n)
        assert np.array_equal(rec[:, 0], np.arange(1, n + 1, dtype=rec.dtype))
        assert (rec[:, 1] == 0).all() and (rec[:, 2] == len(pattern)).all() and (rec[:, 3] == 0).all()
    sc.close(); pat.close()
