"""GPU (-m gpu): the HIP path, called through the C-ABI, against the oracle and the
golden vectors.  Bit-exact: (line, start, end, dist) and every count."""
import ctypes as C
import hashlib
import os
import random
import subprocess
import sys

import numpy as np
import pytest

import known_answers as KA
from conftest import GOLDEN, ROOT
from oracle.pyoracle import SQ_ALL, SQ_BEST, SQ_CONVERT, SQ_COUNT, SQ_FAIL, SQ_FIRST, SQ_IGNORE, SQ_STREAM

pytestmark = pytest.mark.gpu
MODE = dict(FIRST=SQ_FIRST, BEST=SQ_BEST, ALL=SQ_ALL)
FOPT = dict(ANY=0, MATCH=1, NOMATCH=2, COUNTLINES=3, COUNTMATCH=4)
PAT20 = "GATGTAGCGCGATTAGCCTG"
PAT40 = "GATG[TA]AGCNCGATTAGC[CG]TGAAAATGNGAGTAC[GAT]GCGCGA"


class SQ:
    """seeqNew/seeqStringMatch of the product library."""

    def __init__(self, capi, pattern, tau):
        self.L = capi.lib()
        self.sq = self.L.seeqNew(pattern.encode(), tau, 0)
        assert self.sq, capi.error_text()

    def match(self, text, opt):
        n = self.L.seeqStringMatch(text.encode("latin-1"), self.sq, opt)
        assert n >= 0
        m = self.sq.contents.match
        assert self.sq.contents.hits == n
        return [(m[i].start, m[i].end, m[i].dist) for i in range(n)]

    def close(self):
        self.L.seeqFree(self.sq)


def test_loaded_library_is_the_hip_build(gpu, capi):
    # the .so that serves these tests is the in-tree HIP library, and it sees the GPU
    maps = open("/proc/self/maps").read()
    assert capi.LIB_PATH in maps and "libamdhip64" in maps
    assert gpu >= 1


def test_seeqnew_struct_fields(gpu, capi):
    L = capi.lib()
    sq = L.seeqNew(b"ACTGA", 2, 0)                       # testset.c:768-785
    s = sq.contents
    assert (s.hits, s.stacksize, s.tau, s.wlen) == (0, 16, 2, 5)
    assert [s.keys[i][0] for i in range(5)] == [1, 2, 8, 4, 1]
    assert [s.rkeys[i][0] for i in range(5)] == [1, 4, 8, 2, 1]
    assert s.dfa and s.rdfa and s.match and not s.string
    L.seeqFree(sq)


def test_string_known_answers(gpu, capi):
    for pat, tau, text, mo, exp in KA.STRING_MATCH:
        s = SQ(capi, pat, tau)
        assert s.match(text, MODE[mo]) == exp, (pat, text, mo)
        s.close()


def test_golden_string_cases(gpu, capi, string_cases):
    cache = {}
    for c in string_cases:
        if "\0" in c["text"]:
            continue
        key = (c["pattern"], c["tau"])
        if key not in cache:
            cache[key] = SQ(capi, *key)
        got = cache[key].match(c["text"], c["options"])
        assert [list(h) for h in got] == c["hits"], c
    for s in cache.values():
        s.close()


def test_string_fuzz_vs_oracle(gpu, capi, oracle):
    sys.path.insert(0, GOLDEN)
    from make_golden import plain, rand_pattern, rand_text
    rng = random.Random(99)
    for _ in range(150):
        pat = rand_pattern(rng)
        m = len(plain(pat))
        tau = rng.randint(0, min(m - 1, rng.choice([0, 1, 2, 3, 3, 5, 8])))
        s = SQ(capi, pat, tau)
        for _ in range(4):
            text = rand_text(rng, pat, tau, rng.choice([0, 1, 5, 20, 60, 150, 250, 1000]))
            opt = rng.choice([SQ_FIRST, SQ_BEST, SQ_ALL, SQ_COUNT]) | rng.choice([SQ_FAIL, SQ_CONVERT, SQ_IGNORE]) \
                | rng.choice([0, 0, SQ_STREAM])
            assert s.match(text, opt) == oracle.string_match(pat, tau, text, opt), (pat, tau, text, opt)
        s.close()


def test_long_pattern_words(gpu, capi, oracle):
    rng = random.Random(5)
    for m in (33, 64, 65, 128, 129, 300, 512):
        pat = "".join(rng.choice("ACGT") for _ in range(m))
        tau = min(10, m - 1)
        text = "".join(rng.choice("ACGT") for _ in range(200)) + pat[:m // 2] + "T" + pat[m // 2 + 1:] \
            + "".join(rng.choice("ACGT") for _ in range(50))
        s = SQ(capi, pat, tau)
        for opt in (SQ_FIRST, SQ_BEST, SQ_ALL):
            assert s.match(text, opt) == oracle.string_match(pat, tau, text, opt), m
        s.close()
    assert not capi.lib().seeqNew(("A" * 513).encode(), 1, 0)     # documented limit: fails loudly


def _scan(capi, pattern, tau, buf, opt, want, fasta=False, path="auto", tile=None):
    """One batched scan through the device C-ABI.  path: 'generic' (newline index + k_forward<W>),
    'fused' = k_direct (one line per lane, text in registers), 'fused-stream' = k_stream (transition table in LDS:
    the pattern's complete automaton or a partition filter; k_direct when neither fits), 'fused-pair' = k_pair (two bytes per table step over the pattern's pair automaton, selective or not;
    k_stream / k_direct where it does not apply: SQ_IGNORE), or 'auto' (the library's own choice); the env knobs are read when the scan context is created."""
    from seeq_amd import device as dev
    if path != "auto":
        os.environ["SEEQ_FUSED_KERNEL"] = {"fused-stream": "stream", "fused-pair": "pair"}.get(path, "direct")
    if tile:
        os.environ["SEEQ_TILE_BYTES"] = str(tile)
    path = "fused" if path.startswith("fused-") else path
    os.environ["SEEQ_PATH"] = path
    try:
        pat = dev.Pattern(pattern, tau)
        sc = dev.Scanner()
        res = sc.scan_host(pat, bytes(buf), opt | (dev.SEEQDEV_FASTA if fasta else 0), want)
        res["path"] = sc.last_path()
        res["kernel"] = sc.last_kernel()
        res["filter"] = sc.last_filter()
        sc.close()
        pat.close()
    finally:
        os.environ.pop("SEEQ_PATH", None)
        os.environ.pop("SEEQ_TILE_BYTES", None)
        os.environ.pop("SEEQ_FUSED_KERNEL", None)
    return res


@pytest.mark.parametrize("path,tile", [("generic", None), ("fused", None), ("fused", 1024),
                                       ("fused-stream", None), ("fused-pair", None)])
@pytest.mark.parametrize("name,pattern,tau", [("reads_small.txt", PAT20, 3), ("fastq_small.txt", PAT20, 3),
                                              ("fasta_small.txt", PAT20, 3), ("reads250_small.txt", PAT40, 5),
                                              ("reads_small.txt", "GATTAGC", 1), ("testdata.txt", "CACAGAT", 3),
                                              ("reads250_small.txt", "GATGAAGCACGATTAGCCTGAAAATGAGAG", 5)])
def test_batch_scan_vs_oracle(gpu, capi, oracle, name, pattern, tau, path, tile):
    from seeq_amd import device as dev
    buf = open(os.path.join(GOLDEN, name), "rb").read()
    fasta = buf[:1] == b">"
    fusable = len(dev.plain_pattern(pattern)) <= 62
    for nd in (SQ_FAIL, SQ_CONVERT, SQ_IGNORE):
        for mo in (SQ_FIRST, SQ_BEST, SQ_ALL):
            exp = oracle.buffer_scan(pattern, tau, buf, mo | nd, fasta=fasta)
            got = _scan(capi, pattern, tau, buf, mo | nd, dev.WANT_RECORDS, fasta, path, tile)
            assert got["path"] == (path.split("-")[0] if fusable else "generic")     # the kernel under test really ran
            if path == "fused-stream" and nd == SQ_FAIL and not fasta and (pattern, tau) in ((PAT20, 3), ("GATTAGC", 1), ("CACAGAT", 3)):
                assert got["kernel"] == "k_stream" and not got["filter"]
            if path == "fused-stream" and tile is None and nd == SQ_FAIL and pattern == PAT40:
                assert got["kernel"] == "k_stream" and got["filter"]      # configs[4]: partition filter automaton
            if path == "fused":
                assert got["kernel"] == "k_direct"
            if path == "fused-pair" and nd != SQ_IGNORE and fusable:
                assert got["kernel"] == "k_pair" and got["filter"]      # every pattern here has a pair automaton
            assert got["nlines"] == exp["nlines"]
            assert got["nmatchlines"] == exp["nmatchlines"]
            assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (name, mo, nd)
        expa = oracle.buffer_scan(pattern, tau, buf, SQ_ALL | nd, fasta=fasta)
        c1 = _scan(capi, pattern, tau, buf, nd, dev.WANT_COUNTLINES, fasta, path, tile)
        c2 = _scan(capi, pattern, tau, buf, nd, dev.WANT_COUNTMATCH, fasta, path, tile)
        assert c1["nmatchlines"] == expa["nmatchlines"] and c1["nhits"] == expa["nmatchlines"]
        assert c1["nlines"] == expa["nlines"]
        assert c2["nhits"] == len(expa["records"]) and c2["nlines"] == expa["nlines"]


@pytest.mark.parametrize("path", ["generic", "fused", "fused-stream", "fused-pair", "auto"])
def test_edge_buffers(gpu, capi, oracle, path):
    """Empty / ragged / maximum-ish inputs: no trailing newline, empty lines, NUL and CR bytes, a line longer
    than the LDS window (fused: falls back to the HBM per-line scan), 70 k empty lines (fused: many passes
    per tile), every byte value, and lines that straddle tile boundaries at a hit."""
    from seeq_amd import device as dev
    rng = random.Random(3)
    ragged = b"".join((b"ACGT" * rng.randint(0, 60))[:rng.randint(0, 200)] + b"\n" for _ in range(3000))
    star = b"TTTT*ACGT\nACGT*TTTT\nAC*GT\n" * 50          # '*' (newline with the case bit set) must count as non-DNA
    cases = [star, b"", b"\n", b"\n\n\n", b"ACGT", b"ACGT\n", b"\nACGT", b"ACGT\n\nACGT\n", b"ACGT\0ACGT\nACGT",
             b"AC\rGT\r\nACGT\r\n", b"A" * 5000 + b"\n" + b"ACGT" * 3, b"\n" * 70000 + b"ACGT\n",
             bytes(range(256)) * 3, ragged, b"T" * 4090 + b"ACGT\nACGT" + b"T" * 4090 + b"AC\nGT\n",
             (b"ACGT" * 300 + b"\n") * 40,
             # whole tiles of four-byte hit lines, then ordinary lines; one 80 KB line; lines that end right behind a kilobyte
             b"ACG\n" * 40000 + b"ACGT\n" * 10, b"ACGT" * 20000 + b"\nACGT\n", (b"T" * 1020 + b"ACGT\n") * 130]
    for buf in cases:
        for opt in (SQ_ALL, SQ_ALL | SQ_CONVERT, SQ_BEST | SQ_IGNORE, SQ_FIRST):
            exp = oracle.buffer_scan("ACGT", 1, buf, opt)
            got = _scan(capi, "ACGT", 1, buf, opt, dev.WANT_RECORDS, False, path, {"fused": 1024}.get(path))
            assert got["nlines"] == exp["nlines"], (buf[:20], opt)
            assert got["nmatchlines"] == exp["nmatchlines"], (buf[:20], opt)
            assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (buf[:20], opt)
            cnt = _scan(capi, "ACGT", 1, buf, opt, dev.WANT_COUNTLINES, False, path, {"fused": 1024}.get(path))
            assert cnt["nmatchlines"] == exp["nmatchlines"] and cnt["nlines"] == exp["nlines"]


def test_segments_and_workspace_regrowth(gpu, capi, oracle):
    """Multi-segment scans (segment size forced small) and the overflow -> re-run path."""
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
from oracle.pyoracle import Oracle, SQ_ALL, SQ_BEST
from seeq_amd import device as dev
o = Oracle()
pat = "GATGTAGCGCGATTAGCCTG"
buf = o.synth_reads(0, 6000, 150, pat, 3).tobytes()
p = dev.Pattern(pat, 3)
for opt in (SQ_BEST, SQ_ALL):
    exp = o.buffer_scan(pat, 3, buf, opt)
    sc = dev.Scanner()
    sc.reserve(0, 10, 2, 1)                 # absurdly small: every capacity overflows
    got = sc.scan_host(p, buf, opt, dev.WANT_RECORDS)
    assert got["nlines"] == exp["nlines"] and got["nmatchlines"] == exp["nmatchlines"]
    assert np.array_equal(got["records"].astype(np.uint64), exp["records"])
print("OK")
''' % ROOT
    env = dict(os.environ, SEEQ_SEGMENT_BYTES="65536")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


def test_synth_generator_matches_oracle(gpu, capi, oracle):
    import torch
    from seeq_amd import device as dev
    for (pattern, tau, length, first, n) in [(PAT20, 3, 150, 0, 5000), (PAT20, 3, 150, 123456789, 3000),
                                             (dev.plain_pattern(PAT40), 5, 250, 7, 2000)]:
        t = torch.empty(n * (length + 1), dtype=torch.uint8, device="cuda")
        dev.synth_reads(t.data_ptr(), first, n, length, pattern, tau, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(t.cpu().numpy(), oracle.synth_reads(first, n, length, pattern, tau))


def test_device_resident_scan_large_properties(gpu, capi, oracle):
    """2 M reads resident in HBM: exact parity on a prefix + size-independent properties on the whole."""
    import torch
    from seeq_amd import device as dev
    n, length = 2_000_000, 150
    t = torch.empty(n * (length + 1), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    dev.synth_reads(t.data_ptr(), 0, n, length, PAT20, 3, stream=stream)
    torch.cuda.synchronize()                         # (the scanner's stream is ordered behind the null stream anyway)
    pat = dev.Pattern(PAT20, 3)
    sc = dev.Scanner(stream)
    whole = sc.scan_tensor(pat, t, SQ_BEST, dev.WANT_RECORDS)
    rec = sc.records(whole["nrecords"])
    assert whole["nlines"] == n and whole["nrecords"] == whole["nmatchlines"] == len(rec)
    # (1) prefix parity against the oracle
    k = 100_000
    exp = oracle.buffer_scan(PAT20, 3, t[:k * (length + 1)].cpu().numpy(), SQ_BEST)
    assert np.array_equal(rec[rec[:, 0] <= k].astype(np.uint64), exp["records"])
    # (2) additivity: counts of two halves add up; records of the second half are the tail shifted by line base
    h = (n // 2) * (length + 1)
    a = sc.scan_tensor(pat, t[:h], SQ_BEST, dev.WANT_RECORDS)
    b = sc.scan_tensor(pat, t[h:], SQ_BEST, dev.WANT_RECORDS)
    recb = sc.records(b["nrecords"])
    assert a["nmatchlines"] + b["nmatchlines"] == whole["nmatchlines"]
    tail = rec[rec[:, 0] > n // 2].copy()
    tail[:, 0] -= n // 2
    assert np.array_equal(tail, recb)
    # (3) ordering and bounds
    assert np.all(np.diff(rec[:, 0].astype(np.int64)) > 0)
    assert np.all(rec[:, 1] <= rec[:, 2]) and np.all(rec[:, 2] <= length) and np.all(rec[:, 3] <= 3)
    # (4) COUNTLINES == number of BEST records; COUNTMATCH >= COUNTLINES; monotone in tau
    c1 = sc.scan_tensor(pat, t, 0, dev.WANT_COUNTLINES)
    c2 = sc.scan_tensor(pat, t, 0, dev.WANT_COUNTMATCH)
    assert c1["nmatchlines"] == whole["nmatchlines"] and c2["nhits"] >= c1["nmatchlines"]
    pat2 = dev.Pattern(PAT20, 2)
    assert sc.scan_tensor(pat2, t, 0, dev.WANT_COUNTLINES)["nmatchlines"] <= c1["nmatchlines"]
    # (5) every record's distance is what the oracle says for that (line, window)
    sample = rec[:: max(1, len(rec) // 200)]
    host = t.cpu().numpy()
    for line, s, e, d in sample:
        txt = host[(line - 1) * (length + 1):(line - 1) * (length + 1) + length].tobytes().decode()
        assert oracle.string_match(PAT20, 3, txt, SQ_BEST) == [(s, e, d)]


def _file_match_sequence(capi, path, pattern, tau, calls, fasta=False):
    L = capi.lib()
    f = L.seeqOpen(path.encode())
    sq = L.seeqNew(pattern.encode(), tau, 0)
    assert f and sq
    out = []
    for (mo, fo) in calls:
        rv = L.seeqFileMatch(f, sq, mo, fo)
        s = sq.contents
        hits = []
        n = s.hits
        while True:
            m = L.seeqMatchIter(sq)
            if not m:
                break
            hits.append((m.contents.start, m.contents.end, m.contents.dist))
        string = C.string_at(s.string).decode("latin-1") if s.string else None
        out.append((rv, n, f.contents.line, string, hits))
    L.seeqClose(f)
    L.seeqFree(sq)
    return out


def test_filematch_known_answers(gpu, capi):
    path = os.path.join(GOLDEN, "testdata.txt")
    for pattern, tau, seq in KA.FILE_MATCH:
        got = _file_match_sequence(capi, path, pattern, tau, [(MODE[a], FOPT[b]) for a, b, *_ in seq])
        for (mo, fo, rv, nh, line, string, hits), g in zip(seq, got):
            assert g[0] == rv, (pattern, mo, fo, g)
            if nh is not None:
                assert g[1] == nh
            if line is not None:
                assert g[2] == line
            if string is not None:
                assert g[3] == string
            if hits is not None:
                assert g[4] == hits
    for pattern, tau, kind, exp in KA.FILE_COUNTS:
        got = _file_match_sequence(capi, path, pattern, tau, [(0, FOPT[kind])])
        assert got[0][0] == exp
    L = capi.lib()
    f = L.seeqOpen(path.encode())                          # testset.c:931-937
    sq = L.seeqNew(b"ATC", 0, 0)
    f.contents.fdi = None
    assert L.seeqFileMatch(f, sq, 0, 0) == -1 and capi.seeqerr() == 10
    L.seeqFree(sq)


def test_filematch_replay_vs_oracle_small_chunks(gpu, capi, oracle):
    """Every line returned one call at a time (SQ_ANY) with a tiny read-ahead chunk, so that lines
    straddle chunk boundaries; hits, line numbers and sq->string must match the oracle."""
    code = r'''
import os, sys, ctypes as C
sys.path.insert(0, %r)
from oracle.pyoracle import Oracle, SQ_ALL
from seeq_amd import _capi
o = Oracle(); L = _capi.lib()
for name, fasta in (("fastq_small.txt", False), ("fasta_small.txt", True)):
    path = os.path.join(%r, name)
    buf = open(path, "rb").read()
    exp = o.buffer_scan("GATGTAGCGCGATTAGCCTG", 3, buf, SQ_ALL, fasta=fasta)
    lines = [l for l in buf.decode().split("\n")]
    if lines and lines[-1] == "": lines.pop()
    if fasta: lines = [l for l in lines if not l.startswith(">")]
    f = L.seeqOpen(path.encode()); sq = L.seeqNew(b"GATGTAGCGCGATTAGCCTG", 3, 0)
    rec = []; n = 0
    while L.seeqFileMatch(f, sq, SQ_ALL, 0) > 0:
        n += 1
        assert f.contents.line == n
        assert C.string_at(sq.contents.string).decode() == lines[n-1], (n,)
        while True:
            m = L.seeqMatchIter(sq)
            if not m: break
            rec.append((n, m.contents.start, m.contents.end, m.contents.dist))
    assert n == exp["nlines"], (n, exp["nlines"])
    assert rec == [tuple(int(x) for x in r) for r in exp["records"]]
    L.seeqClose(f); L.seeqFree(sq)
print("OK")
''' % (ROOT, GOLDEN)
    env = dict(os.environ, SEEQ_CHUNK_BYTES="1000")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


def test_multi_pattern_scan_vs_independent_oracle_scans(gpu, capi, oracle):
    """Several patterns over one text (barcode demultiplexing, reference doc/response.tex:358-360): the text is staged once,
    every pattern's counts and records must be those of its own, independent oracle scan; the best-barcode assignment
    built on top is checked against the same rule applied to the oracle's records."""
    from seeq_amd import device as dev
    rng = random.Random(11)
    barcodes = ["ACGTTGCA", "TTGACCGA", "GGCATTAC", "CAGTGTCA", "ATATCGCG", "GATTACAG", PAT20, "TG[AC]CANNGT"]
    taus = [1, 1, 1, 2, 0, 1, 3, 1]
    lines = []
    for i in range(4000):
        n = rng.choice([40, 75, 150, 151])
        t = [rng.choice("ACGT") for _ in range(n)]
        for _ in range(rng.choice([0, 1, 1, 2])):
            k = rng.randrange(len(barcodes))
            core = dev.plain_pattern(barcodes[k]).replace("N", "A")
            c = _mutate(rng, core, rng.randint(0, taus[k] + 1))
            p = rng.randrange(0, max(1, n - len(c)))
            t[p:p + len(c)] = list(c)
        if rng.random() < 0.02:
            t[rng.randrange(n)] = "N"
        lines.append("".join(t)[:n])
    buf = ("\n".join(lines) + "\n").encode()
    pats = [dev.Pattern(b, t) for b, t in zip(barcodes, taus)]
    sc = dev.Scanner()
    for opt in (SQ_BEST, SQ_ALL):
        got = sc.scan_host_multi(pats, buf, opt, dev.WANT_RECORDS)
        assert sc.last_multi_one_pass()                     # the set has a union automaton: one walk for all eight (seeq_multi.h)
        exp = [oracle.buffer_scan(b, t, buf, opt) for b, t in zip(barcodes, taus)]
        for k in range(len(pats)):
            assert got[k]["nlines"] == exp[k]["nlines"] == len(lines)
            assert got[k]["nmatchlines"] == exp[k]["nmatchlines"], (k, opt)
            assert np.array_equal(got[k]["records"].astype(np.uint64), exp[k]["records"]), (k, opt)
        if opt == SQ_BEST:
            which, dist, start, end = dev.assign_best(got, len(lines))
            ew = [-1] * len(lines); ed = [None] * len(lines)
            for k in range(len(pats)):
                for ln, s, e, d in exp[k]["records"]:
                    if ed[int(ln) - 1] is None or int(d) < ed[int(ln) - 1]:
                        ew[int(ln) - 1], ed[int(ln) - 1] = k, int(d)
            assert which.tolist() == ew
            assert [None if x < 0 else int(x) for x in dist.tolist()] == ed
            assert (which >= 0).sum() > len(lines) // 4          # the test really assigns reads
    cnt = sc.scan_host_multi(pats, buf, 0, dev.WANT_COUNTLINES)
    for k in range(len(pats)):
        assert cnt[k]["nmatchlines"] == oracle.buffer_scan(barcodes[k], taus[k], buf, 0)["nmatchlines"]
    # the same over a device-resident tensor
    import torch
    t = torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()
    got = sc.scan_tensor_multi(pats, t, SQ_BEST, dev.WANT_RECORDS)
    for k in range(len(pats)):
        assert np.array_equal(got[k]["records"].astype(np.uint64), oracle.buffer_scan(barcodes[k], taus[k], buf, SQ_BEST)["records"]), k
    sc.close()
    for p in pats:
        p.close()


def test_cli_pipelined_ingest_lanes_and_devices(gpu, capi):
    """The ingest pipeline of seeqFileMatch: a reader thread cuts the file into chunks, successive chunks go to the
    lanes (two per device; SEEQ_DEVICES lists the devices -- here device 0 several times, which gives the scheduling of a
    multi-GPU run on the one GPU of the test box).  Whatever the chunk size, lane count and device list, stdout must be
    the bytes of the single-chunk run: records in file order, line numbers running on across chunks, FASTA headers of the
    right record."""
    cases = [("reads_small.txt", ["-d", "3", "-b", "-f", PAT20]), ("reads_small.txt", ["-d", "3", "-c", PAT20]),
             ("reads_small.txt", ["-d", "3", "-i", "-l", PAT20]), ("reads_small.txt", ["-d", "3", "-a", "-l", "-p", "-k", PAT20]),
             ("fasta_small.txt", ["-d", "3", "-b", PAT20]), ("fasta_small.txt", ["-d", "3", "-c", PAT20]),
             ("fasta_small.txt", ["-d", "3", "-l", "-p", PAT20]), ("fastq_small.txt", ["-d", "3", "-x", "1", "-a", "-f", PAT20])]
    from concurrent.futures import ThreadPoolExecutor
    variants = (("700", "1", None), ("4096", "2", None), ("1500", "2", "0,0"), ("3000", "4", "0,0,0"), ("65536", "3", "all"))

    def run(job):                                           # (four CLI processes at a time: each is mostly process and HIP start-up)
        (name, args), v = job
        env = dict(os.environ)
        if v:
            env.update(SEEQ_CHUNK_BYTES=v[0], SEEQ_LANES=v[1])
            if v[2]:
                env["SEEQ_DEVICES"] = v[2]
        return subprocess.run([capi.CLI_PATH] + args + [os.path.join(GOLDEN, name)], capture_output=True, env=env)
    jobs = [(c, v) for c in cases for v in (None,) + variants]
    with ThreadPoolExecutor(max_workers=4) as pool:
        res = dict(zip([(c[0], tuple(c[1]), v) for c, v in jobs], pool.map(run, jobs)))
    for name, args in cases:
        ref = res[(name, tuple(args), None)]
        assert ref.returncode == 0
        for v in variants:
            r = res[(name, tuple(args), v)]
            assert r.returncode == 0 and r.stdout == ref.stdout, (name, args, v, r.stderr[-500:])


def test_cli_pipeline_on_a_generated_file(gpu, capi, oracle, tmp_path):
    """A 6 MB generated file -- reads with planted hits, empty lines, lines of 20 ... 300 KB, foreign bytes -- through the CLI
    with chunk sizes from smaller than the longest line (the chunk has to grow) to larger than the file, 1-3 lanes and a
    device list: `-c` and the number of `-b -f` rows against the oracle, every other output byte-identical across the
    configurations (file order, running line numbers, growth of a chunk, the tail line carried from chunk to chunk)."""
    rng = random.Random(77)
    parts = []
    for i in range(30000):
        r = rng.random()
        n = 0 if r < 0.01 else rng.choice([20000, 100000, 300000]) if r < 0.0015 + 0.01 else rng.choice([50, 100, 151, 151, 151, 250])
        t = [rng.choice("ACGT") for _ in range(n)]
        for _ in range(1 + n // 5000):
            if n >= 30 and rng.random() < 0.3:
                c = _mutate(rng, PAT20, rng.randint(0, 5))
                p = rng.randrange(0, n - len(c) + 1)
                t[p:p + len(c)] = list(c)
        if n and rng.random() < 0.01:
            t[rng.randrange(n)] = rng.choice("N.-X")
        parts.append("".join(t)[:n])
    data = ("\n".join(parts) + "\n").encode()
    path = str(tmp_path / "generated.txt")
    open(path, "wb").write(data)
    exp = oracle.buffer_scan(PAT20, 3, data, SQ_BEST)
    outs = {}
    from concurrent.futures import ThreadPoolExecutor

    def run(job):                                           # (four CLI processes at a time: each is mostly process and HIP start-up)
        (chunk, lanes, devs), args = job
        env = dict(os.environ, SEEQ_CHUNK_BYTES=chunk, SEEQ_LANES=lanes)
        if devs:
            env["SEEQ_DEVICES"] = devs
        return subprocess.run([capi.CLI_PATH, "-d", "3"] + args + [PAT20, path], capture_output=True, env=env)
    jobs = [(cfg, args) for cfg in (("100000000", "2", None), ("4096", "1", None), ("65536", "3", "0,0"), ("1048576", "2", None), ("300001", "2", "0,0,0"))
            for args in (["-c"], ["-b", "-f"], ["-a", "-l", "-p", "-k"], ["-i", "-c"], ["-x", "2", "-b", "-l", "-m"])]
    with ThreadPoolExecutor(max_workers=4) as pool:
        results = list(pool.map(run, jobs))
    for ((chunk, lanes, devs), args), r in zip(jobs, results):
        assert r.returncode == 0, (chunk, lanes, devs, args, r.stderr[-500:])
        key = " ".join(args)
        if key in outs:
            assert r.stdout == outs[key], (chunk, lanes, devs, args)
        outs[key] = r.stdout
    assert int(outs["-c"]) == exp["nmatchlines"]
    rows = outs["-b -f"].decode().splitlines()
    assert len(rows) == len(exp["records"])
    assert rows[:50] == ["%d:%d-%d:%d" % (l, s_, e - 1, d) for l, s_, e, d in exp["records"][:50].tolist()]
    assert rows[-1] == "%d:%d-%d:%d" % tuple([int(exp["records"][-1][0]), int(exp["records"][-1][1]), int(exp["records"][-1][2]) - 1, int(exp["records"][-1][3])])


def test_cli_pipe_streams_lines_as_they_arrive(gpu, capi):
    """`producer | seeq PATTERN`: the reference works line by line (getline, seeq.c:361), so a match shows up as soon as its
    line has been written.  The chunked reader must not sit on a pipe until 64 MiB have arrived."""
    import select
    import time
    p = subprocess.Popen([capi.CLI_PATH, "-d", "1", "-l", "CACAGAT"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, bufsize=0)
    try:
        def expect(text, within=60.0):
            got = b""
            end = time.time() + within
            while len(got) < len(text) and time.time() < end:
                if select.select([p.stdout], [], [], 0.5)[0]:
                    piece = os.read(p.stdout.fileno(), 4096)
                    if not piece:
                        break
                    got += piece
            assert got == text, (got, text)
        p.stdin.write(b"TTTTCACAGATTTT\n"); p.stdin.flush()
        expect(b"1 TTTTCACAGATTTT\n")                     # (first chunk: includes HIP start-up)
        p.stdin.write(b"GGGG\nAAAA\n"); p.stdin.flush()
        time.sleep(0.3)
        p.stdin.write(b"CACAGTT"); p.stdin.flush()          # an unfinished line ...
        time.sleep(0.3)
        p.stdin.write(b"\nxCACAGATx\n"); p.stdin.flush()  # ... finished here; the next one has a non-DNA byte in front
        expect(b"4 CACAGTT\n")
        p.stdin.close()
        assert p.stdout.read() == b"" and p.wait(timeout=60) == 0
    finally:
        if p.poll() is None:
            p.kill()


def test_filematch_pattern_switch_mid_file(gpu, capi, oracle):
    """seeqFileMatch is resumable and takes the pattern per call (seeq.c:293): a caller that switches to another seeq_t (or
    other options) in the middle of a file gets that pattern's answers from the next line on -- the chunks scanned ahead
    for the old one are scanned again."""
    code = r'''
import os, sys, ctypes as C
sys.path.insert(0, %r)
from oracle.pyoracle import Oracle, SQ_ALL, SQ_BEST
from seeq_amd import _capi
o = Oracle(); L = _capi.lib()
path = os.path.join(%r, "reads_small.txt")
lines = open(path).read().split("\n")
if lines[-1] == "": lines.pop()
pats = [(b"GATGTAGCGCGATTAGCCTG", 3, SQ_ALL), (b"GATTAGC", 1, SQ_BEST)]
sqs = [L.seeqNew(p, t, 0) for p, t, _ in pats]
f = L.seeqOpen(path.encode())
n = 0
while True:
    which = (n // 7) %% 2                                   # change pattern and options every 7 lines
    if L.seeqFileMatch(f, sqs[which], pats[which][2], 0) <= 0: break
    n += 1
    assert f.contents.line == n
    got = []
    while True:
        m = L.seeqMatchIter(sqs[which])
        if not m: break
        got.append((m.contents.start, m.contents.end, m.contents.dist))
    assert got == o.string_match(pats[which][0].decode(), pats[which][1], lines[n - 1], pats[which][2])[::-1], (n, which, got)
assert n == len(lines), (n, len(lines))
L.seeqFree(sqs[0])                                          # (the engine goes before the file, as in the reference's seeq())
L.seeqClose(f); L.seeqFree(sqs[1])
print("OK")
''' % (ROOT, GOLDEN)
    for chunk in ("2000", "100000"):
        env = dict(os.environ, SEEQ_CHUNK_BYTES=chunk)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
        assert r.returncode == 0 and "OK" in r.stdout, (chunk, r.stdout[-300:], r.stderr[-2000:])


def test_cli_known_answers(gpu, capi):
    path = os.path.join(GOLDEN, "testdata.txt")
    for args, exp in KA.CLI:
        r = subprocess.run([capi.CLI_PATH] + args + [path], capture_output=True, text=True)
        assert r.stdout == exp, (args, r.stdout, r.stderr)
    r = subprocess.run([capi.CLI_PATH, "CACAG[AT", path], capture_output=True, text=True)
    assert r.returncode == 1 and "missing closing bracket" in r.stderr
    r = subprocess.run([capi.CLI_PATH, "-d", "7", "CACAGAT", path], capture_output=True, text=True)
    assert r.returncode == 1 and "larger than matching distance" in r.stderr
    r = subprocess.run([capi.CLI_PATH, "CACAGAT", "invented.txt"], capture_output=True, text=True)
    assert r.returncode == 1 and "seeqOpen" in r.stderr
    # stdin input
    r = subprocess.run([capi.CLI_PATH, "-f", "-d", "3", "CACAGAT"], input=open(path).read(), capture_output=True,
                       text=True)
    assert r.stdout == "1:8-14:0\n2:8-11:3\n"


def test_cli_golden_outputs(gpu, capi, cli_cases):
    """Byte-identical stdout with the reference CLI on the committed input files (135 invocations; four CLI processes at a time -- a
    run is 0.35 s of process and HIP start-up: the box allows six processes on its GPU)."""
    from concurrent.futures import ThreadPoolExecutor

    def run(c):
        return subprocess.run([capi.CLI_PATH] + c["args"] + [os.path.join(GOLDEN, c["file"])], capture_output=True)
    with ThreadPoolExecutor(max_workers=4) as pool:
        results = list(pool.map(run, cli_cases))
    for c, r in zip(cli_cases, results):
        out = r.stdout.decode("latin-1")
        if "stdout" in c:
            assert out == c["stdout"], (c["file"], c["args"])
        else:
            assert len(out) == c["nbytes"] and hashlib.sha256(out.encode()).hexdigest() == c["sha256"], \
                (c["file"], c["args"])


def test_python_module_surface(gpu, capi):
    import seeq_amd as seeq
    assert seeq.__version__ == "1.2"
    m = seeq.compile(KA.PY_PATTERN, KA.PY_TAU)
    assert (m.pattern, m.mismatches) == (KA.PY_PATTERN, KA.PY_TAU)
    E = KA.PY_EXPECT
    assert m.matchPrefix(KA.PY_NOMATCH, True) is None and m.matchPrefix(KA.PY_NOMATCH, False) is None
    assert m.matchSuffix(KA.PY_NOMATCH, True) is None and m.matchSuffix(KA.PY_NOMATCH, False) is None
    assert m.matchPrefix(KA.PY_MATCH, True) == E["prefix_true"] and m.matchPrefix(KA.PY_MATCH) == E["prefix_true"]
    assert m.matchPrefix(KA.PY_MATCH, False) == E["prefix_false"]
    assert m.matchSuffix(KA.PY_MATCH, True) == E["suffix_true"]
    assert m.matchSuffix(KA.PY_MATCH, False) == E["suffix_false"]
    r = m.match(KA.PY_MATCH)
    assert r.matchlist == E["matchlist"] and r.string == KA.PY_MATCH
    assert r.tokenize() == E["tokenize"] and r.split() == E["split"] and r.matches() == ("CGCTAATAATGGAAT",)
    assert m.match(KA.PY_NOMATCH) is None and m.matchBest(KA.PY_NOMATCH) is None and m.matchAll("") is None
    g = seeq.compile("GATC", 1)
    txt = "TGACTGATGACGTAGTCTACGATCGATCAGTCA"
    assert g.matchAll(txt).matchlist == [(1, 4, 1), (5, 9, 1), (8, 11, 1), (14, 17, 1), (20, 24, 0), (24, 28, 0),
                                         (29, 32, 1)]
    assert g.matchBest(txt).matchlist == [(20, 24, 0)]
    assert list(g.matchIter(txt)) == ["GAC", "GATG", "GAC", "GTC", "GATC", "GATC", "GTC"]
    with pytest.raises(seeq.libseeq_exception):
        seeq.compile("AC[GT", 1)
    # mode 0 converts non-DNA to N, mode 1 ignores it (seeqmodule.c:1079-1080)
    assert seeq.compile("CACAGAT", 0, 1).matchAll("RCACAGATCACAGATCACAGRATCAC").matches() == \
        ("CACAGAT", "CACAGAT", "CACAGRAT")
    assert seeq.compile("CACAGAT", 0, 0).matchAll("RCACAGATCACAGATCACAGRATCAC").matches() == ("CACAGAT", "CACAGAT")


def test_python_batched_api(gpu, capi, oracle):
    """matchBestBatch / matchAllBatch (one device scan for a list of strings) == per-string oracle results."""
    import seeq_amd as seeq
    sys.path.insert(0, GOLDEN)
    from make_golden import rand_text
    rng = random.Random(17)
    pat, tau = "GATGTAGCGCGATTAGCCTG", 3
    texts = [rand_text(rng, pat, tau, rng.choice([0, 5, 60, 150, 400])).replace("\n", "N").replace("\0", "N")
             for _ in range(3000)]
    m = seeq.compile(pat, tau)                       # mode 0 = SQ_CONVERT, as the reference module
    for fn, opt in ((m.matchBatch, SQ_FIRST), (m.matchBestBatch, SQ_BEST), (m.matchAllBatch, SQ_ALL)):
        got = fn(texts)
        assert len(got) == len(texts)
        for tx, g in zip(texts, got):
            assert g == oracle.string_match(pat, tau, tx, opt | SQ_CONVERT)[::-1], (tx, opt)
    assert m.matchBestBatch([]) == []
    with pytest.raises(ValueError):
        m.matchBestBatch(["AC\nGT"])


@pytest.mark.parametrize("m", [30, 31, 32, 33, 45, 62, 63])
def test_two_word_fused_path(gpu, capi, oracle, m):
    """Patterns of 31..62 positions take the two-word fused kernels (k_direct<4,2>, k_exact1<.,2>); 30 the
    one-word ones, 63 the generic path.  Random class/N patterns, planted mutated copies, all modes."""
    from seeq_amd import device as dev
    sys.path.insert(0, GOLDEN)
    from make_golden import mutate, plain
    rng = random.Random(1000 + m)
    parts = []
    for _ in range(m):
        x = rng.random()
        parts.append("N" if x < 0.05 else "[" + "".join(rng.sample("ACGT", 2)) + "]" if x < 0.12 else rng.choice("ACGT"))
    pattern = "".join(parts)
    tau = min(6, m - 1)
    lines = []
    for i in range(4000):
        t = "".join(rng.choice("ACGT") for _ in range(rng.choice([0, 20, 150, 250, 400])))
        if i % 5 == 0 and t:
            cp = mutate(rng, plain(pattern), rng.randint(0, tau + 2))
            q = rng.randrange(len(t) + 1)
            t = t[:q] + cp + t[q + len(cp):]
        if i % 97 == 0 and t:
            q = rng.randrange(len(t)); t = t[:q] + rng.choice("NRn*") + t[q + 1:]
        lines.append(t)
    buf = ("\n".join(lines) + "\n").encode()
    for opt in (SQ_FIRST, SQ_BEST | SQ_CONVERT, SQ_ALL | SQ_IGNORE, SQ_ALL):
        exp = oracle.buffer_scan(pattern, tau, buf, opt)
        got = _scan(capi, pattern, tau, buf, opt, dev.WANT_RECORDS, False, "auto")
        assert got["path"] == ("fused" if m <= 62 else "generic")
        assert got["nlines"] == exp["nlines"] and got["nmatchlines"] == exp["nmatchlines"]
        assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (m, opt)
        cm = _scan(capi, pattern, tau, buf, opt, dev.WANT_COUNTMATCH, False, "auto")
        assert cm["nhits"] == len(oracle.buffer_scan(pattern, tau, buf, (opt & ~3) | SQ_ALL)["records"])


def _mutate(rng, pat, nerr):
    """pat with nerr random edits (substitution / deletion / insertion)."""
    s = list(pat)
    for _ in range(nerr):
        i = rng.randrange(len(s))
        k = rng.randrange(3)
        if k == 0:
            s[i] = rng.choice("ACGT")
        elif k == 1 and len(s) > 1:
            del s[i]
        else:
            s.insert(i, rng.choice("ACGT"))
    return "".join(s)


@pytest.mark.parametrize("kernel", ["stream", "pair"])
def test_stream_chunk_and_tile_boundaries(gpu, capi, oracle, kernel):
    """k_stream gives lanes fixed chunks of the text, so hits, newlines and whole lines straddle chunk, tile
    (64 chunks) and segment boundaries in every possible way -- under k_stream, and under k_pair wherever it applies (the
    long lines of this text send it to k_stream's long-line variant after the first tile that says so): pattern copies planted at every offset around
    the boundaries, lines from 0 to 30 000 bytes with several hits each (one line reported by many lanes),
    non-DNA bytes before / after hits (dirty text: the filter's verdicts get verified), small segments."""
    code = r'''
import os, sys, random, numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, %r)
from oracle.pyoracle import Oracle, SQ_ALL, SQ_BEST, SQ_FIRST
from seeq_amd import device as dev
from test_gpu_parity import _mutate
o = Oracle()
pat = "GATGTAGCGCGATTAGCCTG"
rng = random.Random(77)
def dna(n): return "".join(rng.choice("ACGT") for _ in range(n))
tile = 64 * 128
for dirty in (False, True):
    parts = []
    # 1. copies ending at every offset around the first tile boundaries (one line per copy, lengths vary)
    for off in range(-40, 40):
        pre = dna(rng.randrange(0, 200))
        parts.append(pre + _mutate(rng, pat, rng.randrange(0, 5)) + dna(rng.randrange(0, 60)))
    # 2. short and empty lines, lines shorter than the pattern
    parts += ["", "A", dna(5), "", _mutate(rng, pat, 2), dna(19), pat, pat[:17]]
    # 3. long lines: several hits each, one of them longer than three tiles
    for n in (1000, tile - 7, tile, tile + 9, 30000):
        line = dna(n)
        for _ in range(1 + n // 700):
            p = rng.randrange(0, max(1, n - 30))
            c = _mutate(rng, pat, rng.randrange(0, 4))
            line = line[:p] + c + line[p + len(c):]
        parts.append(line[:n])
    # 4. N in the text, lower case
    parts += [pat[:8] + "N" + pat[9:], pat.lower(), dna(50) + pat[:5] + "NNN" + pat[8:] + dna(50)]
    if dirty:       # bytes that alias onto table columns, before and after would-be hits
        parts += [pat[:11] + "!" + pat[12:], dna(30) + "*" + pat + dna(10), pat + "\tX" + dna(40), "@read/1 " + pat,
                  "+", "IIIIIIIIIIIIIIIIIIII!!!!&&&&((((***", dna(40) + "B" + pat[1:], "\r" + pat + "\r"]
    rng.shuffle(parts)
    for tail in ("\n", ""):
        buf = ("\n".join(parts) + tail).encode()
        p = dev.Pattern(pat, 3)
        sc = dev.Scanner()
        for opt in (SQ_FIRST, SQ_BEST, SQ_ALL):
            exp = o.buffer_scan(pat, 3, buf, opt)
            got = sc.scan_host(p, buf, opt, dev.WANT_RECORDS)
            assert sc.last_kernel() in ("k_stream", "k_pair"), sc.last_kernel()
            assert got["nlines"] == exp["nlines"], (dirty, opt, got["nlines"], exp["nlines"])
            assert got["nmatchlines"] == exp["nmatchlines"], (dirty, opt, got["nmatchlines"], exp["nmatchlines"])
            assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (dirty, opt)
        expa = o.buffer_scan(pat, 3, buf, SQ_ALL)
        c1 = sc.scan_host(p, buf, 0, dev.WANT_COUNTLINES)
        c2 = sc.scan_host(p, buf, 0, dev.WANT_COUNTMATCH)
        assert c1["nmatchlines"] == expa["nmatchlines"] and c1["nlines"] == expa["nlines"]
        assert c2["nhits"] == len(expa["records"]) and c2["nmatchlines"] == expa["nmatchlines"]
        sc.close(); p.close()
        # SQ_CONVERT / SQ_IGNORE: k_stream walks a corrected copy of tiles with non-DNA bytes (default variant)
        for nd in (dev.SQ_CONVERT, dev.SQ_IGNORE):
            p = dev.Pattern(pat, 3)
            sc = dev.Scanner()
            for opt in (SQ_BEST, SQ_ALL):
                exp = o.buffer_scan(pat, 3, buf, opt | nd)
                got = sc.scan_host(p, buf, opt | nd, dev.WANT_RECORDS)
                # (the 128-byte-chunk variant walks a corrected copy under SQ_CONVERT; under SQ_IGNORE it does on read-length
                #  input -- this buffer's average line is long, so there a non-DNA byte sends the scan to a per-line kernel)
                if not dirty or nd == dev.SQ_CONVERT:
                    assert sc.last_kernel() in ("k_stream", "k_pair"), (dirty, nd, sc.last_kernel())
                assert got["nlines"] == exp["nlines"] and got["nmatchlines"] == exp["nmatchlines"], (dirty, nd, opt)
                assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (dirty, nd, opt)
            sc.close(); p.close()
# FASTA input: '>' header lines are not counted and never hit lines; headers next to tile / chunk boundaries, at the
# start of the buffer, in a row, as the last line, and headers that contain the pattern
hdrs = [">chr1 test", ">" + pat, ">", ">x " + dna(150) + pat + dna(30), ">seq|" + "N" * 40, "> " + pat.lower()]
fparts = [">first header " + pat]
for x in parts:
    if rng.random() < 0.45:
        fparts.append(rng.choice(hdrs))
        if rng.random() < 0.2:
            fparts.append(rng.choice(hdrs))
    fparts.append(x)
fparts.append(">last " + pat)
for tail in ("\n", ""):
    buf = ("\n".join(fparts) + tail).encode()
    p = dev.Pattern(pat, 3)
    sc = dev.Scanner()
    for opt in (SQ_FIRST, SQ_BEST, SQ_ALL):
        exp = o.buffer_scan(pat, 3, buf, opt, fasta=True)
        got = sc.scan_host(p, buf, opt | dev.SEEQDEV_FASTA, dev.WANT_RECORDS)
        assert sc.last_kernel() in ("k_stream", "k_pair"), sc.last_kernel()
        assert got["nlines"] == exp["nlines"], ("fasta", opt, got["nlines"], exp["nlines"])
        assert got["nmatchlines"] == exp["nmatchlines"], ("fasta", opt, got["nmatchlines"], exp["nmatchlines"])
        assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), ("fasta", opt)
    expa = o.buffer_scan(pat, 3, buf, SQ_ALL, fasta=True)
    c1 = sc.scan_host(p, buf, dev.SEEQDEV_FASTA, dev.WANT_COUNTLINES)
    c2 = sc.scan_host(p, buf, dev.SEEQDEV_FASTA, dev.WANT_COUNTMATCH)
    assert c1["nmatchlines"] == expa["nmatchlines"] and c1["nlines"] == expa["nlines"]
    assert c2["nhits"] == len(expa["records"]) and c2["nmatchlines"] == expa["nmatchlines"]
    sc.close(); p.close()
print("OK")
''' % (ROOT, os.path.join(ROOT, "tests"))
    for seg in ("65536", "0"):                     # many small segments; one segment
        env = dict(os.environ, SEEQ_FUSED_KERNEL=kernel)
        if seg != "0":
            env["SEEQ_SEGMENT_BYTES"] = seg
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
        assert r.returncode == 0 and "OK" in r.stdout, (seg, r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.parametrize("path", ["auto", "fused-stream", "fused-pair"])
def test_stream_fuzz_patterns(gpu, capi, oracle, path):
    """Random patterns (1..26 positions, N and [..] classes, every distance k_stream takes) over random reads with
    planted mutated copies, N, lower case and a few non-DNA bytes: FIRST/BEST/ALL records and both counts through
    the library's own kernel choice (k_pair while its candidates stay few, k_stream whenever the automaton fits), through
    k_stream alone and through k_pair wherever the pattern has a pair automaton, however unselective, against the oracle."""
    from seeq_amd import device as dev
    rng = random.Random(2025)
    kernels = {}
    for it in range(40):
        m = rng.choice([1, 2, 3, 5, 8, 12, 16, 20, 23, 26])
        parts, plain = [], []
        for _ in range(m):
            r = rng.random()
            if r < 0.08:
                parts.append("N"); plain.append("N")
            elif r < 0.18:
                cls = "".join(sorted(set(rng.choice("ACGT") for _ in range(rng.randint(1, 3)))))
                parts.append("[" + cls + "]"); plain.append(cls[0])
            else:
                c = rng.choice("ACGTacgt")
                parts.append(c); plain.append(c.upper())
        pattern, core = "".join(parts), "".join(plain)
        tau = rng.randint(0, min(5, m - 1, 33 - m))
        lines = []
        for _ in range(1500):
            n = rng.choice([0, 1, 7, 30, 60, 100, 151, 300])
            t = [rng.choice("ACGT") for _ in range(n)]
            if n >= m and rng.random() < 0.4:
                c = _mutate(rng, core.replace("N", "A"), rng.randint(0, tau + 2))
                p = rng.randrange(0, n - len(c) + 1) if n >= len(c) else 0
                t[p:p + len(c)] = list(c)
            if rng.random() < 0.05 and n:
                t[rng.randrange(n)] = "N"
            if rng.random() < 0.03 and n:
                t = [x.lower() for x in t]
            if it % 4 == 3 and rng.random() < 0.02 and n:
                t[rng.randrange(n)] = rng.choice("!*+BJXZ.\t\r@")
            lines.append("".join(t)[:n])
        buf = ("\n".join(lines) + ("\n" if it % 2 else "")).encode()
        for mo in (SQ_FIRST, SQ_BEST, SQ_ALL):
            exp = oracle.buffer_scan(pattern, tau, buf, mo)
            got = _scan(capi, pattern, tau, buf, mo, dev.WANT_RECORDS, False, path)
            kernels[got["kernel"]] = kernels.get(got["kernel"], 0) + 1
            assert got["nlines"] == exp["nlines"] and got["nmatchlines"] == exp["nmatchlines"], (pattern, tau, mo)
            assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (pattern, tau, mo)
        expa = oracle.buffer_scan(pattern, tau, buf, SQ_ALL)
        c1 = _scan(capi, pattern, tau, buf, 0, dev.WANT_COUNTLINES, False, path)
        c2 = _scan(capi, pattern, tau, buf, 0, dev.WANT_COUNTMATCH, False, path)
        assert c1["nmatchlines"] == expa["nmatchlines"] and c1["nlines"] == expa["nlines"], (pattern, tau)
        assert c2["nhits"] == len(expa["records"]) and c2["nmatchlines"] == expa["nmatchlines"], (pattern, tau)
    # the table walks really are what ran, mostly
    if path == "fused-pair":
        assert kernels.get("k_pair", 0) >= 60, kernels
    elif path == "fused-stream":
        assert kernels.get("k_stream", 0) >= 60 and not kernels.get("k_pair"), kernels
    else:
        assert kernels.get("k_stream", 0) + kernels.get("k_pair", 0) >= 60 and kernels.get("k_pair", 0) >= 10, kernels


def test_device_pointer_alignment(gpu, capi, oracle):
    """seeqdevScanRun on device pointers of any alignment (a tensor sliced at odd byte offsets): k_stream's 128-byte
    chunks then straddle memory lines; results must not change."""
    import torch
    from seeq_amd import device as dev
    n, length = 30_000, 150
    t = torch.empty(n * (length + 1) + 256, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    pat = dev.Pattern(PAT20, 3)
    sc = dev.Scanner(stream)
    for off in (1, 3, 17, 131):
        body = t[off:off + n * (length + 1)]
        dev.synth_reads(body.data_ptr(), 0, n, length, PAT20, 3, stream=stream)
        for cut in (0, 5):                          # also start in the middle of a line
            view = body[cut:]
            host = view.cpu().numpy()
            for mo in (SQ_BEST, SQ_ALL):
                exp = oracle.buffer_scan(PAT20, 3, host, mo)
                got = sc.scan_tensor(pat, view, mo, dev.WANT_RECORDS)
                assert sc.last_kernel() in ("k_stream", "k_pair")
                assert got["nlines"] == exp["nlines"] and got["nmatchlines"] == exp["nmatchlines"], (off, cut, mo)
                assert np.array_equal(sc.records(got["nrecords"]).astype(np.uint64), exp["records"]), (off, cut, mo)
    sc.close()
    pat.close()


def test_long_lines_window_walk(gpu, capi, oracle):
    """Chromosome-like input: a few lines of 0.3 - 2 MB with sparse planted hits (plus runs of adjacent hits, hits in
    the last bytes of a chunk and right behind a chunk boundary) between ordinary reads.  On such text the exact pass
    walks candidate windows instead of whole lines; records, counts and line numbers must still equal the oracle's."""
    from seeq_amd import device as dev
    rng = random.Random(99)
    pat, tau = PAT20, 3

    def dna(n):
        return "".join(rng.choice("ACGT") for _ in range(n))

    lines = []
    for n in (2_000_000, 150, 300_000, 151, 1_000_000, 150):
        t = list(dna(n))
        if n > 1000:
            spots = sorted(rng.sample(range(200, n - 200), 40))
            spots += [s + rng.choice([21, 25, 40, 100, 129]) for s in spots[:10]]          # neighbours: same / next chunk
            spots += [128 * rng.randrange(2, n // 128 - 2) + rng.choice([-30, -20, -2, 0, 3, 100, 120]) for _ in range(30)]
            for p in spots:
                c = _mutate(rng, pat, rng.randrange(0, tau + 2))
                t[p:p + len(c)] = list(c)
        elif rng.random() < 0.5:
            t[20:40] = list(pat)
        lines.append("".join(t)[:n])
    buf = ("\n".join(lines) + "\n").encode()
    for mo in (SQ_FIRST, SQ_BEST, SQ_ALL):
        exp = oracle.buffer_scan(pat, tau, buf, mo)
        got = _scan(capi, pat, tau, buf, mo, dev.WANT_RECORDS)
        assert got["kernel"] in ("k_stream", "k_pair")
        assert got["nlines"] == exp["nlines"] and got["nmatchlines"] == exp["nmatchlines"], mo
        assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), mo
    expa = oracle.buffer_scan(pat, tau, buf, SQ_ALL)
    assert len(expa["records"]) > 100
    c2 = _scan(capi, pat, tau, buf, 0, dev.WANT_COUNTMATCH)
    assert c2["nhits"] == len(expa["records"]) and c2["nmatchlines"] == expa["nmatchlines"]
    # the same lines as FASTA records (headers make the text "dirty": the walk has to prove every stretch it skips
    # clean), and with non-DNA bytes inside the long lines (SQ_FAIL: everything behind such a byte is dead)
    fa = []
    for i, ln in enumerate(lines):
        fa.append(">chr%d %s" % (i, pat if i % 2 else "len=%d" % len(ln)))
        fa.append(ln)
    dirty = list(lines)
    for i in (0, 2, 4):
        t = list(dirty[i])
        for p in rng.sample(range(len(t) // 3, len(t)), 3):
            t[p] = rng.choice("!*RYJ+.")
        dirty[i] = "".join(t)
    for text_lines, fasta in ((fa, True), (dirty, False)):
        b2 = ("\n".join(text_lines) + "\n").encode()
        for mo in (SQ_FIRST, SQ_BEST, SQ_ALL):
            exp = oracle.buffer_scan(pat, tau, b2, mo, fasta=fasta)
            got = _scan(capi, pat, tau, b2, mo, dev.WANT_RECORDS, fasta)
            assert got["kernel"] in ("k_stream", "k_pair")
            assert got["nlines"] == exp["nlines"] and got["nmatchlines"] == exp["nmatchlines"], (fasta, mo)
            assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (fasta, mo)
        expa = oracle.buffer_scan(pat, tau, b2, SQ_ALL, fasta=fasta)
        c1 = _scan(capi, pat, tau, b2, 0, dev.WANT_COUNTLINES, fasta)
        c2 = _scan(capi, pat, tau, b2, 0, dev.WANT_COUNTMATCH, fasta)
        assert c1["nmatchlines"] == expa["nmatchlines"] and c1["nlines"] == expa["nlines"], fasta
        assert c2["nhits"] == len(expa["records"]) and c2["nmatchlines"] == expa["nmatchlines"], fasta


def test_a_refused_reserve_leaves_the_context_usable(gpu, capi, oracle):
    """Round 5, the device boundary's failure contract on the real device (the CPU suite injects the same failures into the host code:
    tests/test_host_sanitizers.py): seeqdevScanReserve for more records than the card has memory is REFUSED -- -1, seeqerr = 0,
    errno = ENOMEM, seeqdevLastError names the call -- and the scan context goes on with the workspace it had: the next scans return
    the oracle's records (reference contract for its own allocation failures: libseeq.c:75-135, test/testset.c:150-155)."""
    import errno
    from seeq_amd import device as dev
    buf = open(os.path.join(GOLDEN, "reads_small.txt"), "rb").read()
    exp = oracle.buffer_scan(PAT20, 3, buf, SQ_BEST)
    pat = dev.Pattern(PAT20, 3)
    sc = dev.Scanner()
    L = capi.lib()
    got = sc.scan_host(pat, buf, SQ_BEST, dev.WANT_RECORDS)
    assert np.array_equal(got["records"].astype(np.uint64), exp["records"])
    for (nbytes, nlines, nhl, nrec) in ((0, 0, 0, 1 << 40), (0, 0, 1 << 38, 0), (0, 1 << 40, 0, 0)):      # 24 TB of records / 12 TB of hit-list arrays / 4 TB of line starts
        C.set_errno(0)
        rc = L.seeqdevScanReserve(sc._h, nbytes, nlines, nhl, nrec)
        assert rc == -1 and C.get_errno() == errno.ENOMEM and capi.seeqerr() == 0, (rc, C.get_errno(), capi.seeqerr())
        assert b"hipMalloc" in L.seeqdevLastError()
        for opt in (SQ_BEST, SQ_ALL):
            e2 = oracle.buffer_scan(PAT20, 3, buf, opt)
            g2 = sc.scan_host(pat, buf, opt, dev.WANT_RECORDS)
            assert g2["nlines"] == e2["nlines"] and np.array_equal(g2["records"].astype(np.uint64), e2["records"]), (nrec, opt)
    # a scan that needs MORE than the refused context has grows it by the ordinary re-run
    big = buf * 40
    e3 = oracle.buffer_scan(PAT20, 3, big, SQ_ALL)
    g3 = sc.scan_host(pat, big, SQ_ALL, dev.WANT_RECORDS)
    assert np.array_equal(g3["records"].astype(np.uint64), e3["records"])
    sc.close()
    pat.close()


def test_every_byte_value_alone(gpu, capi, oracle):
    """One foreign byte value at a time in otherwise clean text, right in front of a perfect copy of the pattern:
    k_stream's alphabet check must flag exactly the bytes that are not A C G T N (either case) or newline -- each of
    the 256 values gets its own scan, so no other byte can raise the flag for it."""
    from seeq_amd import device as dev
    pat = dev.Pattern(PAT20, 3)
    sc = dev.Scanner()
    # (a copy of the pattern right behind the byte, one with the byte inside it, one in front of it on the same line)
    for b in range(256):
        buf = (b"TTTTTTTT" + bytes([b]) + PAT20.encode() + b"TT\n" + b"ACGTACGTAC" + bytes([b]) + b"\n" +
               PAT20[:9].encode() + bytes([b]) + PAT20[10:].encode() + b"\n" + PAT20.encode() + b"AC" + bytes([b]) + b"GT\n") * 40
        for nd in (SQ_FAIL, SQ_CONVERT, SQ_IGNORE):
            for opt, want in ((0, dev.WANT_COUNTLINES), (SQ_ALL, dev.WANT_RECORDS), (SQ_BEST, dev.WANT_RECORDS)):
                exp = oracle.buffer_scan(PAT20, 3, buf, opt | nd)
                got = sc.scan_host(pat, buf, opt | nd, want)
                # SQ_FAIL: flagged tiles -> candidates verified; SQ_CONVERT: walked over a corrected copy; SQ_IGNORE: the same
                # with skip bytes, unless most lines become candidates (here every line holds the byte): then the per-line kernel
                if nd != SQ_IGNORE:
                    assert sc.last_kernel() in ("k_stream", "k_pair"), (b, nd)
                assert got["nlines"] == exp["nlines"] and got["nmatchlines"] == exp["nmatchlines"], (b, nd, opt)
                if want == dev.WANT_RECORDS:
                    assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (b, nd, opt)
    sc.close()
    pat.close()


def test_ignore_and_convert_with_foreign_bytes_inside_matches(gpu, capi, oracle):
    """SQ_IGNORE skips a byte that is neither a base nor a terminator (but counts it in coordinates), SQ_CONVERT reads it as
    N (reference libseeq.c:223-228, 265-270).  Pattern copies with 0..4 such bytes INSIDE them, at every offset relative
    to k_stream's 64-byte chains (so that skipped bytes fall into warm-up windows and matches straddle chunk and tile
    boundaries), plus NULs and lines made of foreign bytes only.  Read-length lines: k_stream serves both modes itself."""
    from seeq_amd import device as dev
    rng = random.Random(4242)
    foreign = "-.XH*+|!@8(\tZ"
    # (heavy: most lines hold foreign bytes -- SQ_IGNORE may hand the text to the per-line kernel; light: few do -- it must not)
    for (pattern, tau, heavy) in ((PAT20, 3, True), (PAT20, 3, False), ("GATTAGC", 1, False), (PAT40, 5, True), (PAT40, 5, False)):
        core = dev.plain_pattern(pattern).replace("N", "A")
        lines = []
        for i in range(3000):
            n = rng.choice([60, 100, 149, 150, 151, 152, 200])
            t = [rng.choice("ACGT") for _ in range(n)]
            if rng.random() < 0.6:
                c = list(_mutate(rng, core, rng.randint(0, tau + 1)))
                for _ in range(rng.choice([0, 1, 1, 2, 4]) if heavy or rng.random() < 0.1 else 0):
                    c.insert(rng.randrange(0, len(c) + 1), rng.choice(foreign))
                p = rng.randrange(0, max(1, n - len(c)))
                t[p:p + len(c)] = c
            if rng.random() < (0.1 if heavy else 0.02):
                t[rng.randrange(len(t))] = rng.choice(foreign)
            if rng.random() < 0.01:
                t[rng.randrange(len(t))] = "\0"
            if rng.random() < (0.01 if heavy else 0.003):
                t = [rng.choice(foreign) for _ in range(rng.choice([1, 5, 40]))]
            lines.append("".join(t))
        buf = ("\n".join(lines) + "\n").encode("latin-1")
        pat = dev.Pattern(pattern, tau)
        sc = dev.Scanner()
        for nd in (SQ_IGNORE, SQ_CONVERT, SQ_FAIL):
            for opt in (SQ_FIRST, SQ_BEST, SQ_ALL):
                exp = oracle.buffer_scan(pattern, tau, buf, opt | nd)
                got = sc.scan_host(pat, buf, opt | nd, dev.WANT_RECORDS)
                if nd != SQ_IGNORE or not heavy:
                    assert sc.last_kernel() in ("k_stream", "k_pair"), (pattern, nd, heavy, sc.last_kernel())
                assert got["nlines"] == exp["nlines"] and got["nmatchlines"] == exp["nmatchlines"], (pattern, nd, opt)
                assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (pattern, nd, opt)
            expa = oracle.buffer_scan(pattern, tau, buf, SQ_ALL | nd)
            c1 = sc.scan_host(pat, buf, nd, dev.WANT_COUNTLINES)
            c2 = sc.scan_host(pat, buf, nd, dev.WANT_COUNTMATCH)
            assert c1["nmatchlines"] == expa["nmatchlines"] and c2["nhits"] == len(expa["records"]), (pattern, nd)
        sc.close()
        pat.close()


@pytest.mark.parametrize("seg", [None, "65536"])
def test_fastq_records_on_the_pair_walk(gpu, capi, oracle, seg, monkeypatch):
    """Round 5: FASTQ-shaped text (four-line records, quality lines made of bytes that alias onto bases AND onto the newline
    column: '+', ':', ';', '*', 'J') stays on k_pair under SQ_FAIL, SQ_CONVERT and -- lines that hold a skipped byte named whole by
    markers, seeq_pair.h IG -- SQ_IGNORE (there the copies planted in quality lines behind foreign bytes ARE hits, with the bytes
    between their characters skipped).  Every tile fails the fast alphabet check
    there, remakes its newline masks from its registers, and k_verify looks at the bytes between a line's start and a
    candidate's window (reference libseeq.c:267-270: under SQ_FAIL such a byte ends the line).  Adversarial lines: a perfect
    copy of the pattern in a quality line BEHIND a foreign byte (no hit under SQ_FAIL, a hit under SQ_CONVERT), in front of the
    first one (a hit in both), a NUL in front of a copy (no hit in either), copies in header and '+' lines, reads with lower
    case, U, N; records of every match option, both counts, segments of 64 KiB (lines across seams)."""
    from seeq_amd import device as dev
    if seg:
        monkeypatch.setenv("SEEQ_SEGMENT_BYTES", seg)
    rng = random.Random(20255)
    qual = "".join(chr(c) for c in range(33, 75))
    for (pattern, tau, L) in ((PAT20, 3, 150), (PAT40, 5, 250), ("GATTAGC", 1, 36)):
        core = dev.plain_pattern(pattern).replace("N", "A")
        lines = []
        for i in range(6000):
            read = [rng.choice("ACGT") for _ in range(L)]
            if rng.random() < 0.3:
                c = _mutate(rng, core, rng.randint(0, tau + 2))
                p = rng.randrange(0, max(1, L - len(c)))
                read[p:p + len(c)] = list(c)
            r = rng.random()
            if r < 0.03: read[rng.randrange(L)] = rng.choice("Nn")
            elif r < 0.06: read = [ch.lower() if rng.random() < 0.5 else ch for ch in read]
            elif r < 0.08: read[rng.randrange(L)] = rng.choice("Uu")
            elif r < 0.09: read[rng.randrange(L)] = "\0"
            q = [rng.choice(qual) for _ in range(L)]
            r = rng.random()
            if r < 0.15:                                   # a copy somewhere in the quality line, foreign bytes before it
                p = rng.randrange(1, max(2, L - len(core)))
                q[p:p + len(core)] = list(core)
            elif r < 0.25:                                 # a copy at the very start: nothing ends the line before it
                q[0:len(core)] = list(core)
            elif r < 0.32:                                 # clean DNA up to a copy deep in the line, one foreign byte / NUL far in front of it
                q = [rng.choice("ACGT") for _ in range(L)]
                p = rng.randrange(L // 2, L - len(core))
                q[p:p + len(core)] = list(core)
                q[rng.randrange(0, max(1, p - len(core) - tau - 2))] = rng.choice("!+:J*\0;@")
            elif r < 0.36:                                 # the same without the foreign byte: a plain hit line among the quality lines
                q = [rng.choice("ACGT") for _ in range(L)]
                p = rng.randrange(0, L - len(core))
                q[p:p + len(core)] = list(core)
            hdr = "@r%07d" % i if rng.random() > 0.05 else "@" + core
            plus = "+" if rng.random() > 0.05 else "+" + core
            lines += [hdr, "".join(read), plus, "".join(q)]
        buf = ("\n".join(lines) + ("\n" if rng.random() < 0.5 else "")).encode("latin-1")
        pat = dev.Pattern(pattern, tau)
        sc = dev.Scanner()
        for nd in (SQ_FAIL, SQ_CONVERT, SQ_IGNORE):
            for opt in (SQ_FIRST, SQ_BEST, SQ_ALL):
                exp = oracle.buffer_scan(pattern, tau, buf, opt | nd)
                got = sc.scan_host(pat, buf, opt | nd, dev.WANT_RECORDS)
                if len(pattern) >= 20:
                    assert sc.last_kernel() == "k_pair", (pattern, nd, sc.last_kernel())      # the library's own choice: no knob is set
                assert got["nlines"] == exp["nlines"] and got["nmatchlines"] == exp["nmatchlines"], (pattern, nd, opt)
                assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (pattern, nd, opt)
            expa = oracle.buffer_scan(pattern, tau, buf, SQ_ALL | nd)
            c1 = sc.scan_host(pat, buf, nd, dev.WANT_COUNTLINES)
            c2 = sc.scan_host(pat, buf, nd, dev.WANT_COUNTMATCH)
            assert c1["nmatchlines"] == expa["nmatchlines"] and c2["nhits"] == len(expa["records"]), (pattern, nd)
        sc.close()
        pat.close()


def test_ignore_lines_that_begin_with_their_tile(gpu, capi, oracle):
    """Found by profiles/ignore_fuzz.py (round 5): under SQ_IGNORE on k_pair a line that begins with its 8 KiB tile -- the buffer's
    first line, or one whose first byte stands on a multiple of 8192 -- and runs past the tile's first lane (128 bytes) was left
    without its marker: the lanes behind the first did not know the line's start had been seen.  Here: such lines hold a copy of
    the pattern with skipped bytes INSIDE it (a hit only when they are skipped), at the buffer's start and at tiles 1, 2 and 5,
    lengths from one lane to three, with and without a skipped byte in the first lane."""
    from seeq_amd import device as dev
    rng = random.Random(50501)
    pattern, tau = PAT20, 3
    core = dev.plain_pattern(pattern)

    def special(n, where, skip_early):
        t = [rng.choice("ACGT") for _ in range(n)]
        c = list(core)
        for _ in range(7):                                  # (more than tau: the walk over aliased bytes cannot see the copy, only the marker names the line)
            c.insert(rng.randrange(1, len(c)), rng.choice("!+:J*;@5"))
        t[where:where + len(c)] = c
        if skip_early:
            t[rng.randrange(0, 100)] = "#"
        return "".join(t[:n])
    for n, where in ((150, 20), (150, 120), (300, 150), (300, 260), (129, 100), (400, 300)):
        for skip_early in (False, True):
            lines, pos = [], 0
            lines.append(special(n, where, skip_early)); pos += n + 1
            for tile in (1, 2, 5):
                while pos + 152 < tile * 8192:
                    lines.append("".join(rng.choice("ACGT") for _ in range(150))); pos += 151
                pad = tile * 8192 - pos - 1                # the filler line ends right in front of the tile
                lines.append("".join(rng.choice("ACGT") for _ in range(pad))); pos += pad + 1
                assert pos == tile * 8192
                lines.append(special(n, where, skip_early)); pos += n + 1
            for _ in range(40):
                lines.append("".join(rng.choice("ACGT") for _ in range(150)))
            buf = ("\n".join(lines) + "\n").encode("latin-1")
            pat = dev.Pattern(pattern, tau)
            sc = dev.Scanner()
            for opt in (SQ_BEST, SQ_ALL, SQ_FIRST):
                exp = oracle.buffer_scan(pattern, tau, buf, opt | SQ_IGNORE)
                got = sc.scan_host(pat, buf, opt | SQ_IGNORE, dev.WANT_RECORDS)
                assert sc.last_kernel() == "k_pair"
                assert exp["nmatchlines"] >= 4                                  # (the four special lines are hits under SQ_IGNORE)
                assert got["nlines"] == exp["nlines"] and got["nmatchlines"] == exp["nmatchlines"], (n, where, skip_early, opt, got["nmatchlines"], exp["nmatchlines"])
                assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (n, where, skip_early, opt)
            sc.close()
            pat.close()


def test_fasta_header_hit_behind_a_hit_line_on_long_line_plans(gpu, capi, oracle):
    """Found by profiles/ignore_fuzz.py (round 5): FASTA input under SQ_CONVERT on the long-line plans (a buffer that also holds a long
    line with candidates).  A header line carries the line number of the line before it, so a candidate inside a header is a
    "repeat" of that line, and the window walk of the exact pass jumped to it over the line's END when the stretch it skipped held
    nothing outside the alphabet -- the newline as its last byte, the header's '>' right behind it: a hit inside the header came out
    as a hit of the line before.  Here: hit lines of 150 bytes followed by headers that hold the pattern at offsets 1 .. 40, at
    eight random alignments each, so that every landing position of the jump occurs, '>' itself included."""
    from seeq_amd import device as dev
    rng = random.Random(50505)
    pattern, tau = "CAACCCCAACACCACAACCAAAAA", 4

    def dna(n):
        return "".join(rng.choice("ACGT") for _ in range(n))
    long_line = "".join((pattern if i % 7 == 0 else dna(40)) for i in range(700))      # ~27 KB with copies all over: the long-line plan
    lines = [long_line]
    for off in range(1, 41):
        for rep in range(8):                                # (the walk looks for its next candidate at block ends: filler lines of random length shift the line through every alignment)
            where = rng.choice((19, 19, 60, 100, 126))
            lines.append(dna(rng.randint(90, 220)))
            lines.append(dna(where) + pattern + dna(150 - where - len(pattern)))
            lines.append(">" + "r" * (off - 1) + pattern + " tail")
    buf = ("\n".join(lines) + "\n").encode("latin-1")
    pat = dev.Pattern(pattern, tau)
    sc = dev.Scanner()
    kernels = set()
    for nd in (SQ_CONVERT, SQ_FAIL):
        for opt in (SQ_ALL, SQ_BEST, SQ_FIRST):
            exp = oracle.buffer_scan(pattern, tau, buf, opt | nd, fasta=True)
            got = sc.scan_host(pat, buf, opt | nd | dev.SEEQDEV_FASTA, dev.WANT_RECORDS)
            kernels.add((nd, sc.last_kernel()))
            assert got["nlines"] == exp["nlines"] and got["nmatchlines"] == exp["nmatchlines"], (nd, opt, sc.last_kernel(), got["nmatchlines"], exp["nmatchlines"])
            assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (nd, opt, sc.last_kernel())
    assert (SQ_CONVERT, "k_myers") in kernels, kernels              # (the long-line plan of this text under SQ_CONVERT)
    sc.close()
    pat.close()


def test_ignore_line_across_a_segment_seam(gpu, capi, oracle, monkeypatch):
    """Found by profiles/ignore_fuzz.py (round 5): under SQ_IGNORE on k_pair a segment's last tile names the line that runs out of it
    by a marker made unseen; k_bounds2 drops the marker when the line holds no skipped byte -- but the next segment still took
    the line for covered by the segment before and dropped what it found in it as repeats: the hits of a line that starts in
    front of a seam and has them behind it were lost.  Lines of 300 to 1 200 bytes across a 64 KiB seam, copies of the pattern
    in front of the seam, behind it, on both sides; without a skipped byte, with one in front of the seam, with one behind it."""
    from seeq_amd import device as dev
    monkeypatch.setenv("SEEQ_SEGMENT_BYTES", "65536")
    rng = random.Random(50504)
    pattern, tau = PAT20, 2
    core = dev.plain_pattern(pattern)
    for n, before in ((1200, 460), (600, 300), (300, 150), (1200, 1100)):
        for copies in ("behind", "front", "both"):
            for skip in (None, "front", "behind"):
                lines, pos = [], 0
                while pos + 152 < 65536 - before - 200:
                    lines.append("".join(rng.choice("ACGT") for _ in range(150))); pos += 151
                pad = 65536 - before - pos - 1
                lines.append("".join(rng.choice("ACGT") for _ in range(pad))); pos += pad + 1
                t = [rng.choice("ACGT") for _ in range(n)]
                if copies in ("behind", "both"):
                    p = rng.randrange(before + 30, n - 25); t[p:p + 20] = list(core)
                if copies in ("front", "both"):
                    p = rng.randrange(0, before - 25); t[p:p + 20] = list(core)
                if skip == "front":
                    t[rng.randrange(0, before)] = "#"
                elif skip == "behind":
                    t[rng.randrange(before, n)] = "#"
                lines.append("".join(t))
                for _ in range(60):
                    lines.append("".join(rng.choice("ACGT") for _ in range(150)))
                buf = ("\n".join(lines) + "\n").encode("latin-1")
                pat = dev.Pattern(pattern, tau)
                sc = dev.Scanner()
                for opt in (SQ_BEST, SQ_ALL):
                    exp = oracle.buffer_scan(pattern, tau, buf, opt | SQ_IGNORE)
                    got = sc.scan_host(pat, buf, opt | SQ_IGNORE, dev.WANT_RECORDS)
                    assert exp["nmatchlines"] >= 1
                    assert got["nlines"] == exp["nlines"] and got["nmatchlines"] == exp["nmatchlines"], (n, before, copies, skip, opt, sc.last_kernel(), got["nmatchlines"], exp["nmatchlines"])
                    assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (n, before, copies, skip, opt, sc.last_kernel())
                sc.close()
                pat.close()


@pytest.mark.parametrize("kernel", [None, "stream", "direct", "generic"])
def test_u_and_lower_case_are_bases_on_every_walk(gpu, capi, oracle, monkeypatch, kernel):
    """Found by profiles/ignore_fuzz.py (round 5): U and u are T (reference seeqcore.h:89-111), but k_stream's alphabet check counted
    them as foreign -- harmless under SQ_FAIL (the exact pass looked), wrong under SQ_CONVERT and SQ_IGNORE, where the corrected copy it
    walks had turned them into N / a skipped byte: occurrences written with U were lost.  Reads in mixed case with U for T, planted
    copies written the same way, on read-length and on long lines, every walk, the three non-DNA modes."""
    from seeq_amd import device as dev
    if kernel == "generic":
        monkeypatch.setenv("SEEQ_PATH", "generic")
    elif kernel:
        monkeypatch.setenv("SEEQ_FUSED_KERNEL", kernel)
    rng = random.Random(50503)

    def rna(t):
        return "".join((c.lower() if rng.random() < 0.5 else c) if c != "T" or rng.random() < 0.3 else rng.choice("Uu") for c in t)
    for pattern, tau, lens in ((PAT20, 3, (150,)), ("TTCTTGTTAT", 1, (100, 151)), (PAT20, 2, (3000, 9000))):
        core = dev.plain_pattern(pattern)
        lines = []
        for i in range(1200 if lens[0] < 1000 else 60):
            n = rng.choice(lens)
            t = [rng.choice("ACGT") for _ in range(n)]
            for _ in range(1 + n // 1000):
                if rng.random() < 0.5:
                    c = _mutate(rng, core, rng.randint(0, tau + 1))
                    p = rng.randrange(0, n - len(c))
                    t[p:p + len(c)] = list(c)
            t = rna("".join(t))
            if rng.random() < 0.1:
                t = t[:5] + rng.choice("!;*") + t[6:]            # and now and then a byte that IS foreign
            lines.append(t)
        buf = ("\n".join(lines) + "\n").encode("latin-1")
        pat = dev.Pattern(pattern, tau)
        sc = dev.Scanner()
        for nd in (SQ_CONVERT, SQ_IGNORE, SQ_FAIL):
            for opt in (SQ_BEST, SQ_ALL):
                exp = oracle.buffer_scan(pattern, tau, buf, opt | nd)
                got = sc.scan_host(pat, buf, opt | nd, dev.WANT_RECORDS)
                assert exp["nmatchlines"] > 10
                assert got["nlines"] == exp["nlines"] and got["nmatchlines"] == exp["nmatchlines"], (kernel, pattern, lens, nd, opt, sc.last_kernel(), got["nmatchlines"], exp["nmatchlines"])
                assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (kernel, pattern, lens, nd, opt, sc.last_kernel())
        sc.close()
        pat.close()


def test_direct_regions_on_lines_of_259_bytes(gpu, capi, oracle, monkeypatch):
    """Found by profiles/ignore_fuzz.py (round 5): k_direct sizes its regions to 63.5 lines of the sampled average and reads a region
    in at most 16 rounds of 1 KiB -- lines averaging more than 258 bytes made regions of 16 KiB + 48 bytes, the last 48 bytes of
    which no round looked at: a newline there was not counted and every later record carried a line number one too small.  (The
    planner sends such text to k_direct by itself under SQ_IGNORE when it also holds long lines.)"""
    from seeq_amd import device as dev
    monkeypatch.setenv("SEEQ_FUSED_KERNEL", "direct")
    rng = random.Random(50502)
    pattern, tau = PAT20, 3
    core = dev.plain_pattern(pattern)
    ran_direct = 0
    for lens in ((258,), (257, 259), (240, 277), (200, 317)):      # (with the newline: 259 .. 259.5 bytes a line -- above 260 the planner leaves k_direct)
        lines = []
        for i in range(1500):
            n = rng.choice(lens)
            t = [rng.choice("ACGT") for _ in range(n)]
            if rng.random() < 0.3:
                c = _mutate(rng, core, rng.randint(0, tau + 1))
                p = rng.randrange(0, n - len(c))
                t[p:p + len(c)] = list(c)
            lines.append("".join(t))
        buf = ("\n".join(lines) + "\n").encode("latin-1")
        pat = dev.Pattern(pattern, tau)
        sc = dev.Scanner()
        for nd in (SQ_FAIL, SQ_IGNORE):
            exp = oracle.buffer_scan(pattern, tau, buf, SQ_BEST | nd)
            got = sc.scan_host(pat, buf, SQ_BEST | nd, dev.WANT_RECORDS)
            ran_direct += sc.last_kernel() == "k_direct"
            assert got["nlines"] == exp["nlines"] and got["nmatchlines"] == exp["nmatchlines"], (lens, nd, got["nlines"], exp["nlines"])
            assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (lens, nd)
        sc.close()
        pat.close()
    assert ran_direct >= 6


@pytest.mark.parametrize("seg", [None, "65536"])
def test_stream_fuzz_long_lines(gpu, capi, oracle, seg, monkeypatch):
    """Random patterns over long lines (up to 70 000 bytes, many planted hits per line, some with a non-DNA byte or as
    FASTA records): the long-line variant of k_stream and the window walk of the exact pass against the oracle.
    seg = 65536: the same with 64 KiB segments (lines longer than a segment; with the short patterns nearly every chain
    reports a hit, the first segment's hit list overflows the default workspace and the run is repeated -- the segments
    behind an overflowing one must not touch the hit arrays: this configuration once faulted in k_stream_bounds)."""
    from seeq_amd import device as dev
    if seg:
        monkeypatch.setenv("SEEQ_SEGMENT_BYTES", seg)         # (read when a scan context is created: _scan makes one per call)
    rng = random.Random(4242)
    seen = {}
    for it in range(12):
        m = rng.choice([6, 12, 18, 20, 25, 30])
        pattern = "".join(rng.choice("ACGT") if rng.random() > 0.1 else rng.choice(["N", "[AC]", "[GT]"]) for _ in range(m))
        core = dev.plain_pattern(pattern).replace("N", "A")
        tau = rng.randint(0, min(4, m - 1, 33 - m))
        fasta, dirty = it % 4 == 1, it % 4 == 2
        lines = []
        for _ in range(60):
            n = rng.choice([0, 151, 2000, 8191, 8192, 8300, 20000, 70000])
            t = [rng.choice("ACGT") for _ in range(n)]
            for _rep in range(1 + n // 900):
                if n >= m and rng.random() < 0.8:
                    c = _mutate(rng, core, rng.randint(0, tau + 2))
                    p = rng.randrange(0, n - len(c) + 1) if n >= len(c) else 0
                    t[p:p + len(c)] = list(c)
            if dirty and n and rng.random() < 0.4:
                t[rng.randrange(n)] = rng.choice("!*+BJXZ.\t\r@>")
            s = "".join(t)[:n]
            if fasta and rng.random() < 0.4:
                lines.append(">rec %d %s" % (len(lines), core))
            lines.append(s)
        buf = ("\n".join(lines) + ("\n" if it % 2 else "")).encode()
        for mo in (SQ_FIRST, SQ_BEST, SQ_ALL):
            exp = oracle.buffer_scan(pattern, tau, buf, mo, fasta=fasta)
            got = _scan(capi, pattern, tau, buf, mo, dev.WANT_RECORDS, fasta)
            seen[got["kernel"]] = seen.get(got["kernel"], 0) + 1
            assert got["nlines"] == exp["nlines"] and got["nmatchlines"] == exp["nmatchlines"], (it, pattern, tau, mo)
            assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (it, pattern, tau, mo)
        expa = oracle.buffer_scan(pattern, tau, buf, SQ_ALL, fasta=fasta)
        c2 = _scan(capi, pattern, tau, buf, 0, dev.WANT_COUNTMATCH, fasta)
        assert c2["nhits"] == len(expa["records"]) and c2["nmatchlines"] == expa["nmatchlines"], (it, pattern, tau)
    assert seen.get("k_stream", 0) + seen.get("k_myers", 0) + seen.get("k_pair", 0) >= 24 and seen.get("k_forward", 0) + seen.get("k_direct", 0) <= 12, seen      # (the long-line plans: round 5, k_pair's LL variant among them)


def test_long_string_match(gpu, capi, oracle):
    """seeqStringMatch on a chromosome-sized string (3 MB, sparse hits): served by the batched line scan (line 1 of
    the buffer) instead of the one-lane single-line kernels; same hits as the oracle, also with a '\\n' or a non-DNA
    byte in the middle (everything behind it is not part of the string's matchable prefix) and under SQ_CONVERT."""
    rng = random.Random(31)
    n = 3_000_000
    t = [rng.choice("ACGT") for _ in range(n)]
    for p in sorted(rng.sample(range(100, n - 100), 60)):
        c = _mutate(rng, PAT20, rng.randrange(0, 5))
        t[p:p + len(c)] = list(c)
    base = "".join(t)
    cut = n // 2
    variants = [(base, 0), (base[:cut] + "\n" + base[cut + 1:], 0), (base[:cut] + "R" + base[cut + 1:], 0),
                (base[:cut] + "R" + base[cut + 1:], SQ_CONVERT), (base.lower(), SQ_IGNORE)]
    s = SQ(capi, PAT20, 3)
    for text, nd in variants:
        for mo in (SQ_FIRST, SQ_BEST, SQ_ALL):
            assert s.match(text, mo | nd) == oracle.string_match(PAT20, 3, text, mo | nd), (mo, nd, text[cut - 2:cut + 2])
    s.close()


def test_string_match_every_size_class(gpu, capi, oracle):
    """seeqStringMatch across the size classes of k_string: read from host memory (<= 4 KiB), staged from device memory,
    positions shared out over the workgroup (<= 32 KiB, more than 64 KiB of LDS beyond 16 KiB), the batched scan beyond
    -- dense and sparse hits, terminators and N inside, a skipped byte (one-lane scan), one- and two-word patterns."""
    rng = random.Random(2718)
    for pat, tau in ((PAT20, 3), ("GATGAAGCACGATTAGCCTGAAAATGAGAG", 5), ("ACGT", 1)):
        core = pat
        s = SQ(capi, pat, tau)
        for n in (1, 19, 255, 256, 257, 4096, 4097, 8192, 8193, 16000, 20000, 32767, 32768, 32769, 40000):
            t = [rng.choice("ACGT") for _ in range(n)]
            for p in rng.sample(range(max(1, n - len(core))), min(12, max(1, n // 300))):
                c = _mutate(rng, core, rng.randrange(0, tau + 2))
                t[p:p + len(c)] = list(c)[:max(0, n - p)]
            base = "".join(t)[:n]
            texts = [base]
            if n > 40:
                texts.append(base[:n // 3] + "N" + base[n // 3 + 1:2 * n // 3] + "\n" + base[2 * n // 3 + 1:])
                texts.append(base[:n // 2] + "!" + base[n // 2 + 1:])
            for text in texts:
                for opt in (SQ_FIRST, SQ_BEST, SQ_ALL, SQ_ALL | SQ_CONVERT, SQ_BEST | SQ_IGNORE):
                    assert s.match(text, opt) == oracle.string_match(pat, tau, text, opt), (pat, n, opt, text[:30])
        s.close()


def test_mixed_reads_and_long_line(gpu, capi, oracle):
    """A file of reads (the sampled average line is short: the read-length kernels are chosen) with one 2 MB line
    that has hits: the scan notices, re-runs itself once with the long-line variant, and the results are the oracle's."""
    from seeq_amd import device as dev
    rng = random.Random(8)
    reads = ["".join(rng.choice("ACGT") for _ in range(150)) for _ in range(3000)]
    for i in range(0, 3000, 17):
        reads[i] = reads[i][:40] + _mutate(rng, PAT20, rng.randrange(0, 4)) + reads[i][60:]
    big = [rng.choice("ACGT") for _ in range(2_000_000)]
    for p in rng.sample(range(1000, 1_999_000), 25):
        big[p:p + 20] = list(_mutate(rng, PAT20, rng.randrange(0, 4)))[:20]
    lines = reads[:2000] + ["".join(big)] + reads[2000:]
    buf = ("\n".join(lines) + "\n").encode()
    for mo in (SQ_BEST, SQ_ALL):
        exp = oracle.buffer_scan(PAT20, 3, buf, mo)
        got = _scan(capi, PAT20, 3, buf, mo, dev.WANT_RECORDS)
        assert got["kernel"] in ("k_stream", "k_pair")
        assert got["nlines"] == exp["nlines"] and got["nmatchlines"] == exp["nmatchlines"], mo
        assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), mo


def test_scanner_reused_across_patterns(gpu, capi, oracle):
    """One Scanner, many patterns created and freed in turn (a freed pattern's address is often handed out again):
    the per-context EQ table cache is keyed on the pattern's generation id, so every scan uses its own pattern."""
    from seeq_amd import device as dev
    buf = open(os.path.join(GOLDEN, "reads_small.txt"), "rb").read()
    sc = dev.Scanner()
    rng = random.Random(5)
    for i in range(12):
        m = rng.choice([8, 12, 20, 28])
        pattern = "".join(rng.choice("ACGT") for _ in range(m)) if i % 3 else PAT20[:m]
        tau = rng.randint(1, 3)
        pat = dev.Pattern(pattern, tau)
        exp = oracle.buffer_scan(pattern, tau, buf, SQ_ALL)
        got = sc.scan_host(pat, buf, SQ_ALL, dev.WANT_RECORDS)
        assert got["nmatchlines"] == exp["nmatchlines"], (i, pattern, tau)
        assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (i, pattern, tau)
        pat.close()
    # the per-line kernels are the ones that read the EQ table in their scan loop
    os.environ["SEEQ_FUSED_KERNEL"] = "direct"
    try:
        sc2 = dev.Scanner()
        for i in range(8):
            pattern = "".join(rng.choice("ACGT") for _ in range(20))
            pat = dev.Pattern(pattern if i % 2 else PAT20, 3)
            exp = oracle.buffer_scan(pat.pattern, 3, buf, SQ_BEST)
            got = sc2.scan_host(pat, buf, SQ_BEST, dev.WANT_RECORDS)
            assert sc2.last_kernel() == "k_direct"
            assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (i, pat.pattern)
            pat.close()
        sc2.close()
    finally:
        os.environ.pop("SEEQ_FUSED_KERNEL", None)
    sc.close()


def test_filter_automaton_patterns(gpu, capi, oracle):
    """Patterns whose complete automaton does not fit LDS run k_stream over a partition FILTER automaton; every hit line
    it flags is verified by the exact pass.  configs[4]'s pattern plus random class/N patterns of 24..62 positions at
    distances 2..6, reads with planted copies at 0..tau+2 edits, N, lower case and (SQ_FAIL) a few non-DNA bytes:
    FIRST / BEST / ALL records and both counts against the oracle, and the filter really is what ran."""
    from seeq_amd import device as dev
    sys.path.insert(0, GOLDEN)
    from make_golden import mutate, plain
    rng = random.Random(404)
    cases = [(PAT40, 5), ("GATGTAGCGCGATTAGCCTGAAAA", 3), (PAT20, 4), ("GATGAAGCACGATTAGCCTGAAAATGAGAG", 5)]
    for _ in range(8):
        m = rng.choice([24, 31, 40, 50, 62])
        cases.append(("".join("N" if rng.random() < 0.05 else "[" + "".join(rng.sample("ACGT", 2)) + "]"
                              if rng.random() < 0.08 else rng.choice("ACGT") for _ in range(m)), rng.randint(2, 6)))
    nfilter = 0
    for ci, (pattern, tau) in enumerate(cases):
        core = plain(pattern)
        lines = []
        for i in range(3000):
            n = rng.choice([0, 20, 150, 250, 400])
            t = "".join(rng.choice("ACGT") for _ in range(n))
            if i % 4 == 0 and n >= len(core):
                cp = mutate(rng, core, rng.randint(0, tau + 2))
                q = rng.randrange(n - len(cp) + 1) if n >= len(cp) else 0
                t = (t[:q] + cp + t[q + len(cp):])[:n]
            if i % 41 == 0 and n:
                q = rng.randrange(n); t = t[:q] + "N" + t[q + 1:]
            if ci % 3 == 2 and i % 53 == 0 and n:
                q = rng.randrange(n); t = t[:q] + rng.choice("!*RJ+.\t") + t[q + 1:]
            lines.append(t.lower() if i % 29 == 0 else t)
        buf = ("\n".join(lines) + ("\n" if ci % 2 else "")).encode()
        for mo in (SQ_FIRST, SQ_BEST, SQ_ALL):
            exp = oracle.buffer_scan(pattern, tau, buf, mo)
            got = _scan(capi, pattern, tau, buf, mo, dev.WANT_RECORDS)
            nfilter += got["filter"]
            assert got["nlines"] == exp["nlines"] and got["nmatchlines"] == exp["nmatchlines"], (pattern, tau, mo)
            assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (pattern, tau, mo)
        expa = oracle.buffer_scan(pattern, tau, buf, SQ_ALL)
        c1 = _scan(capi, pattern, tau, buf, 0, dev.WANT_COUNTLINES)
        c2 = _scan(capi, pattern, tau, buf, 0, dev.WANT_COUNTMATCH)
        assert c1["nmatchlines"] == expa["nmatchlines"] and c1["nlines"] == expa["nlines"], (pattern, tau)
        assert c2["nhits"] == len(expa["records"]) and c2["nmatchlines"] == expa["nmatchlines"], (pattern, tau)
        if ci in (0, 1, 3):        # (the 20-mer at distance 4 only has a filter of 10-mers with 2 errors: not selective -> k_direct)
            assert got["kernel"] in ("k_stream", "k_pair") and got["filter"], (pattern, tau, got["kernel"])
    assert nfilter >= 15, nfilter


def test_all_mode_many_records_per_line(gpu, capi, oracle):
    """SQ_ALL with several / very many records per line: the exact pass keeps a line's first emission per hit line and
    parks the others in per-wave overflow lists (seeq_exact1.h); lines with hundreds of emissions (low-complexity text:
    one record per position) do not fit them and take the re-scan path.  Records in order, bit-exact, for a complete
    automaton, a filter automaton and the per-line kernel, also with FASTA headers and under SQ_IGNORE / SQ_CONVERT."""
    from seeq_amd import device as dev
    rng = random.Random(777)
    core40 = "GATGTAGCACGATTAGCCTGAAAATGAGAGTACGGCGCGA"
    for ci, (pattern, tau, unit) in enumerate([(PAT20, 3, PAT20), ("AAAAAAAA", 1, "A" * 40), (PAT40, 5, core40),
                                               ("ACACACACAC", 2, "AC" * 30)]):
        for dense in (False, True):
            lines = []
            for i in range(4000):
                n = rng.choice([60, 150, 250, 300])
                t = "".join(rng.choice("ACGT") for _ in range(n))
                k = (i % 7) if not dense else (i % 3) * 3
                if i % 2 == 0:
                    q = 0
                    for _ in range(k):                       # k copies of the unit, a few random characters apart
                        if q + len(unit) > n:
                            break
                        t = t[:q] + unit + t[q + len(unit):]
                        q += len(unit) + rng.randint(1, 12)
                if dense and i % 5 == 0:
                    t = (unit * 8)[:n]                       # an emission at (nearly) every position
                if ci == 1 and i % 31 == 0:
                    t = t[:n // 2] + "#!" + t[n // 2 + 2:]
                lines.append(t[:n])
            buf = ("\n".join(lines) + "\n").encode()
            for opt in (SQ_ALL, SQ_ALL | SQ_IGNORE, SQ_ALL | SQ_CONVERT):
                if opt != SQ_ALL and ci != 1:
                    continue
                exp = oracle.buffer_scan(pattern, tau, buf, opt)
                for path in ("auto", "fused"):
                    got = _scan(capi, pattern, tau, buf, opt, dev.WANT_RECORDS, False, path)
                    assert got["nlines"] == exp["nlines"] and got["nmatchlines"] == exp["nmatchlines"], (pattern, dense, opt, path)
                    assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (pattern, dense, opt, path)
                cm = _scan(capi, pattern, tau, buf, opt & ~3, dev.WANT_COUNTMATCH)
                assert cm["nhits"] == len(exp["records"]), (pattern, dense, opt)
            if ci == 0:
                fa = (">hdr one\n" + "\n".join(lines[:1500]) + "\n>hdr two GATGTAGCGCGATTAGCCTG\n" + "\n".join(lines[1500:3000]) + "\n").encode()
                exp = oracle.buffer_scan(pattern, tau, fa, SQ_ALL, fasta=True)
                got = _scan(capi, pattern, tau, fa, SQ_ALL, dev.WANT_RECORDS, True)
                assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (pattern, dense, "fasta")


MULTI16 = r"""
import os, sys, random, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
from oracle.pyoracle import Oracle, SQ_ALL, SQ_BEST, SQ_FIRST
from seeq_amd import device as dev
from test_gpu_parity import _mutate
o = Oracle()
rng = random.Random(%(seed)d)
lens = %(lens)r
barcodes = ["".join(rng.choice("ACGT") for _ in range(rng.choice(lens))) for _ in range(16)]
taus = [rng.choice(%(taus)r) for _ in range(16)]
lines = []
for i in range(%(nlines)d):
    n = rng.choice([150, 150, 150, 151, 100, 36, 0, 12])
    t = [rng.choice("ACGT") for _ in range(n)]
    for _ in range(rng.choice([0, 1, 1, 1, 2])):
        k = rng.randrange(16)
        c = _mutate(rng, barcodes[k], rng.randint(0, taus[k] + 1))
        if n >= len(c):
            q = rng.choice([0, 0, n - len(c), rng.randrange(n - len(c) + 1)])
            t[q:q + len(c)] = list(c)
    if rng.random() < 0.02 and n: t[rng.randrange(n)] = "N"
    if rng.random() < 0.01 and n: t = [x.lower() for x in t]
    if %(foreign)r and rng.random() < 0.01 and n: t[rng.randrange(n)] = rng.choice("!*XZ-.")
    lines.append("".join(t)[:n])
buf = ("\n".join(lines) + "\n").encode()
pats = [dev.Pattern(b, t) for b, t in zip(barcodes, taus)]
sc = dev.Scanner()
nd = %(nd)d
for opt, want in ((SQ_BEST, dev.WANT_RECORDS), (SQ_ALL, dev.WANT_RECORDS), (SQ_FIRST, dev.WANT_RECORDS), (0, dev.WANT_COUNTLINES), (0, dev.WANT_COUNTMATCH)):
    os.environ.pop("SEEQ_MULTI", None)
    one = sc.scan_host_multi(pats, buf, opt | nd, want)
    assert sc.last_multi_one_pass(), "the one-pass path did not run"
    os.environ["SEEQ_MULTI"] = "sequential"
    seq = sc.scan_host_multi(pats, buf, opt | nd, want)
    assert not sc.last_multi_one_pass()
    for k in range(16):
        for f in ("nlines", "nmatchlines", "nhits", "nrecords"):
            assert one[k][f] == seq[k][f], (k, f, opt, want, one[k], seq[k])
        if want == dev.WANT_RECORDS:
            assert np.array_equal(one[k]["records"], seq[k]["records"]), (k, opt)
    if want == dev.WANT_RECORDS:
        for k in (0, 5, 11, 15):
            exp = o.buffer_scan(barcodes[k], taus[k], buf, opt | nd)
            assert one[k]["nmatchlines"] == exp["nmatchlines"] and np.array_equal(one[k]["records"].astype(np.uint64), exp["records"]), (k, opt)
print("MULTI OK", sum(r["nmatchlines"] for r in one))
"""


@pytest.mark.parametrize("case", [dict(seed=5, lens=[10], taus=[1], nlines=60000, foreign=False, nd=0, env={}),
                                  dict(seed=6, lens=[8, 9, 11, 12], taus=[0, 1], nlines=60000, foreign=True, nd=0, env={}),
                                  dict(seed=7, lens=[8, 10, 12], taus=[1], nlines=40000, foreign=True, nd=4, env={}),
                                  dict(seed=8, lens=[9, 10], taus=[1], nlines=40000, foreign=False, nd=0, env={"SEEQ_SEGMENT_BYTES": "65536"})],
                         ids=["10mers", "mixed-foreign", "convert", "small-segments"])
def test_multi_pattern_one_pass_equals_sequential(gpu, capi, oracle, case):
    """Sixteen barcodes over one text (seeq_multi.h): the one walk for all of them against a scan per pattern
    (SEEQ_MULTI=sequential) -- counts and records of every pattern, every match option and `want` -- and four of the patterns
    against the oracle; reads with planted barcodes at either end, N, lower case, foreign bytes, SQ_CONVERT, 64 KiB segments."""
    code = MULTI16 % dict(root=ROOT, **{k: v for k, v in case.items() if k != "env"})
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **case["env"]), timeout=900)
    assert r.returncode == 0 and "MULTI OK" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def test_multi_pattern_edges_and_fallbacks(gpu, capi, oracle):
    """The one-walk multi-pattern scan at its edges -- two patterns, thirty-two, classes and N, FASTA input with candidates in
    header lines, empty lines, no newline at the end, a text of foreign bytes under SQ_FAIL -- and the sets / options it hands
    to the per-pattern scans (a pattern shorter than d + 2, SQ_IGNORE, one pattern): always the oracle's results per pattern."""
    from seeq_amd import device as dev
    rng = random.Random(404)

    def text_for(barcodes, taus, nlines, fasta=False, foreign=False, trailing=True, fasta_every=5):
        lines = []
        for i in range(nlines):
            n = rng.choice([0, 0, 20, 80, 150, 151])
            t = [rng.choice("ACGT") for _ in range(n)]
            for _ in range(rng.choice([0, 1, 2])):
                k = rng.randrange(len(barcodes))
                c = _mutate(rng, dev.plain_pattern(barcodes[k]).replace("N", "A"), rng.randint(0, taus[k] + 1))
                if n >= len(c):
                    q = rng.choice([0, n - len(c), rng.randrange(n - len(c) + 1)])
                    t[q:q + len(c)] = list(c)
            if foreign and rng.random() < 0.05 and n:
                t[rng.randrange(n)] = rng.choice("!*XZ-.\t")
            line = "".join(t)[:n]
            if fasta and i % fasta_every == 0:
                line = ">" + (dev.plain_pattern(barcodes[i % len(barcodes)]).replace("N", "A") + line)[:60]
            lines.append(line)
        return ("\n".join(lines) + ("\n" if trailing else "")).encode()

    def check(barcodes, taus, buf, options=0, fasta=False, expect_one_pass=True):
        pats = [dev.Pattern(b, t) for b, t in zip(barcodes, taus)]
        sc = dev.Scanner()
        fl = dev.SEEQDEV_FASTA if fasta else 0
        for opt, want in ((SQ_BEST, dev.WANT_RECORDS), (SQ_ALL, dev.WANT_RECORDS), (0, dev.WANT_COUNTLINES), (0, dev.WANT_COUNTMATCH)):
            got = sc.scan_host_multi(pats, buf, opt | options | fl, want)
            assert expect_one_pass is None or sc.last_multi_one_pass() == expect_one_pass, (barcodes, opt, want)
            for k, (b, t) in enumerate(zip(barcodes, taus)):
                exp = oracle.buffer_scan(b, t, buf, (opt if want == dev.WANT_RECORDS else SQ_ALL) | options, fasta=fasta)
                assert got[k]["nlines"] == exp["nlines"] and got[k]["nmatchlines"] == exp["nmatchlines"], (barcodes, k, opt, want)
                if want == dev.WANT_RECORDS:
                    assert np.array_equal(got[k]["records"].astype(np.uint64), exp["records"]), (barcodes, k, opt)
                if want == dev.WANT_COUNTMATCH:
                    assert got[k]["nhits"] == len(exp["records"]), (barcodes, k)
        sc.close()
        for p in pats:
            p.close()

    two = ["GATTACAGA", "TTGACCGAT"]
    check(two, [1, 1], text_for(two, [1, 1], 3000))
    check(two, [1, 1], text_for(two, [1, 1], 3000, trailing=False))
    thirty_two = ["".join(rng.choice("ACGT") for _ in range(10)) for _ in range(32)]
    check(thirty_two, [1] * 32, text_for(thirty_two, [1] * 32, 6000))
    classes = ["AC[GT]TNGCAT", "TTGAC[AC]GANN", "GGCATTAC", "NNCAGTGT"]
    check(classes, [1, 1, 0, 1], text_for(classes, [1, 1, 0, 1], 4000))
    check(two, [1, 1], text_for(two, [1, 1], 3000, fasta=True), fasta=True, expect_one_pass=None)      # (header lines every few lines: dirty, as above)
    check(two, [1, 1], text_for(two, [1, 1], 6000, fasta=True, fasta_every=300), fasta=True)               # sparse headers: the one walk, candidates inside header lines dropped
    # (a foreign byte in one line of twenty is more than one per 4 KB: the sampled text counts as dirty and k_pair -- hence the one
    #  walk -- stays out; the MULTI16 cases with one in a hundred lines run on it.  Either way: the oracle's results)
    check(two, [1, 1], text_for(two, [1, 1], 3000, foreign=True), expect_one_pass=None)      # SQ_FAIL: a foreign byte ends its line
    check(two, [1, 1], text_for(two, [1, 1], 3000, foreign=True), options=dev.SQ_CONVERT, expect_one_pass=None)
    # what the one walk does not take: a scan per pattern, same results
    check(two, [1, 1], text_for(two, [1, 1], 2000, foreign=True), options=dev.SQ_IGNORE, expect_one_pass=False)
    check(["ACG", "GATTACAGA"], [2, 1], text_for(["ACG", "GATTACAGA"], [2, 1], 1500), expect_one_pass=False)     # shorter than d + 2
    check(["GATTACAGA"], [1], text_for(["GATTACAGA"], [1], 1500), expect_one_pass=False)


def test_multi_pattern_workspace_growth(gpu, capi, oracle):
    """The one-walk multi-pattern scan with a workspace that is far too small for its candidates, its (line, pattern) pairs and
    its records (seeqdevScanReserve with tiny capacities): the device reports what it needs, the scan is run again, the
    results are the oracle's -- for one dominant pattern (its region of the shared workspace overflows first) and for a spread."""
    from seeq_amd import device as dev
    rng = random.Random(99)
    barcodes = ["GATTACAGAC", "TTGACCGATA", "CCATGGTACA", "AGAGTCTCTG"]
    taus = [1, 1, 1, 1]
    for weights in ([1, 1, 1, 1], [30, 1, 1, 1]):
        lines = []
        for _ in range(30000):
            n = 100
            t = [rng.choice("ACGT") for _ in range(n)]
            for _ in range(2):
                k = rng.choices(range(4), weights)[0]
                c = _mutate(rng, barcodes[k], rng.randint(0, 1))
                q = rng.randrange(n - len(c) + 1)
                t[q:q + len(c)] = list(c)
            lines.append("".join(t)[:n])
        buf = ("\n".join(lines) + "\n").encode()
        pats = [dev.Pattern(b, t) for b, t in zip(barcodes, taus)]
        for opt, want in ((SQ_ALL, dev.WANT_RECORDS), (SQ_BEST, dev.WANT_RECORDS), (0, dev.WANT_COUNTMATCH)):
            sc = dev.Scanner()
            sc.reserve(len(buf), len(lines) + 64, 2000, 1500)
            got = sc.scan_host_multi(pats, buf, opt, want)
            assert sc.last_multi_one_pass()
            for k, (b, t) in enumerate(zip(barcodes, taus)):
                exp = oracle.buffer_scan(b, t, buf, opt if want == dev.WANT_RECORDS else SQ_ALL)
                assert got[k]["nmatchlines"] == exp["nmatchlines"], (weights, k, opt, want)
                if want == dev.WANT_RECORDS:
                    assert np.array_equal(got[k]["records"].astype(np.uint64), exp["records"]), (weights, k, opt)
                else:
                    assert got[k]["nhits"] == len(exp["records"]), (weights, k)
            sc.close()
        for p in pats:
            p.close()


@pytest.mark.parametrize("m,k", [(42, 8), (42, 15), (34, 10), (27, 8), (20, 5), (20, 3), (27, 4), (34, 6), (42, 10)])
def test_published_sweep_cells_small(gpu, capi, oracle, m, k):
    """The reference's published sweep (doc/response.tex:209-232: chromosome lines, m = 20 / 27 / 34 / 42 x k) at a size the oracle
    finishes in seconds: six lines of 200-400 KB of random DNA with mutated copies of the pattern planted, `--all` with
    positions and both counts against the oracle.  m = 20, k = 3 has a complete automaton; the other cells a partition filter
    (taken on long lines when it flags fewer than ~3 positions per KB) or k_stream's Myers mode; the full-size cells are
    profiles/r04_chrom_sweep.txt."""
    from seeq_amd import device as dev
    rng = random.Random(1000 * m + k)
    pattern = "".join(rng.choice("ACGT") for _ in range(m))
    lines = []
    for i in range(6):
        n = rng.choice([200_000, 300_000, 400_000])
        t = [rng.choice("ACGT") for _ in range(n)]
        for _ in range(40):
            c = _mutate(rng, pattern, rng.randint(0, k + 2))
            q = rng.randrange(n - len(c))
            t[q:q + len(c)] = list(c)
        if i == 2:
            t[rng.randrange(n)] = "N"
        lines.append("".join(t))
    buf = ("\n".join(lines) + "\n").encode()
    pat = dev.Pattern(pattern, k)
    sc = dev.Scanner()
    exp = oracle.buffer_scan(pattern, k, buf, SQ_ALL)
    got = sc.scan_host(pat, buf, SQ_ALL, dev.WANT_RECORDS)
    # (20, 3): the complete automaton; cells whose partition filter makes fewer than ~3 candidates per KB (round 4: the filter on
    # long lines, the window walk running on m + k + 2 columns behind a candidate's chunk); the rest: k_stream's Myers mode
    # (round 5: where the pattern's pair automaton is selective enough, k_pair's long-line variant -- its candidates verified by the window walk)
    if (m, k) in ((20, 3), (27, 4), (34, 6)):
        assert sc.last_kernel() == "k_pair" and sc.last_filter(), (sc.last_kernel(), sc.last_filter())
    elif (m, k) == (42, 8):
        assert sc.last_kernel() == "k_stream" and sc.last_filter(), (sc.last_kernel(), sc.last_filter())
    elif (m, k) in ((42, 15), (34, 10), (27, 8)):
        assert sc.last_kernel() == "k_myers", sc.last_kernel()
    else:
        assert sc.last_kernel() in ("k_pair", "k_stream", "k_myers")
    assert got["nlines"] == exp["nlines"] == 6 and got["nmatchlines"] == exp["nmatchlines"]
    assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (m, k)
    expb = oracle.buffer_scan(pattern, k, buf, SQ_BEST)
    gotb = sc.scan_host(pat, buf, SQ_BEST, dev.WANT_RECORDS)
    assert np.array_equal(gotb["records"].astype(np.uint64), expb["records"]), (m, k)
    c2 = sc.scan_host(pat, buf, 0, dev.WANT_COUNTMATCH)
    assert c2["nhits"] == len(exp["records"]) and c2["nmatchlines"] == exp["nmatchlines"]
    sc.close(); pat.close()


LEADERS = r"""
import os, sys, random, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
from oracle.pyoracle import Oracle, SQ_ALL, SQ_BEST
from seeq_amd import device as dev
from test_gpu_parity import _mutate
o = Oracle()
rng = random.Random(%(seed)d)
cases = [("GATGTAGCGCGATTAGCCTG", 4, 2500, "random"), ("GATGTAGCGCGATTAGCCTGAAAATGCGAGTACGGCGCGAAT", 12, 1500, "random"),
         ("AAAAAAAAAAAAAAAAAAAA", 3, 300, "polyA"), ("ACACACACACACACACACACACAC", 4, 300, "periodic")]
for pattern, tau, nplant, kind in cases:
    lines = []
    for i in range(4):
        n = rng.choice([300_000, 500_000])
        t = [rng.choice("ACGT") for _ in range(n)]
        for _ in range(nplant):
            if kind == "random":
                c = _mutate(rng, pattern, rng.randint(0, tau + 1))
            elif kind == "polyA":
                c = "A" * rng.randint(10, 700)                       # runs far longer than a chunk: the walk goes on from chunk to chunk
            else:
                c = "AC" * rng.randint(5, 400)
            q = rng.randrange(n - len(c))
            t[q:q + len(c)] = list(c)
        lines.append("".join(t))
    buf = ("\n".join(lines) + "\n").encode()
    pat = dev.Pattern(pattern, tau)
    sc = dev.Scanner()
    exp = o.buffer_scan(pattern, tau, buf, SQ_ALL)
    got = sc.scan_host(pat, buf, SQ_ALL, dev.WANT_RECORDS)
    assert got["nlines"] == 4 and got["nmatchlines"] == exp["nmatchlines"], (pattern, got["nmatchlines"], exp["nmatchlines"])
    assert np.array_equal(got["records"].astype(np.uint64), exp["records"]), (pattern, kind, len(got["records"]), len(exp["records"]))
    c2 = sc.scan_host(pat, buf, 0, dev.WANT_COUNTMATCH)
    assert c2["nhits"] == len(exp["records"]) and c2["nmatchlines"] == exp["nmatchlines"], (pattern, kind)
    gb = sc.scan_host(pat, buf, SQ_BEST, dev.WANT_RECORDS)
    assert np.array_equal(gb["records"].astype(np.uint64), o.buffer_scan(pattern, tau, buf, SQ_BEST)["records"]), (pattern, kind)
    print(pattern[:12], kind, len(exp["records"]), "records", sc.last_kernel())
    sc.close(); pat.close()
print("LEADERS OK")
"""


@pytest.mark.parametrize("env", [{}, {"SEEQ_NO_LEADERS": "1"}, {"SEEQ_SEGMENT_BYTES": "1048576"}], ids=["leaders", "one-lane-per-line", "1MiB-segments"])
def test_long_lines_with_many_hits(gpu, capi, oracle, env):
    """Long lines whose candidates are counted in the thousands (the dense cells of the published sweep): candidates far
    behind the one before them are walked by lanes of their own (seeq_stream.h, leaders) -- random text with thousands of
    planted copies, and periodic text (poly-A runs, AC repeats of up to 800 bytes) where the walk of one candidate runs
    on into the next one's window: there the run is declared void and repeated with one lane per line.  Every record of
    `--all`, both counts and `--best` against the oracle; also with the leaders switched off and with 1 MiB segments."""
    code = LEADERS % dict(root=ROOT, seed=12)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env), timeout=900)
    assert r.returncode == 0 and "LEADERS OK" in r.stdout, (r.stdout[-600:], r.stderr[-3000:])
