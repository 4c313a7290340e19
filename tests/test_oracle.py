"""CPU: the oracle (oracle/seeq_oracle.c) against every golden vector we hold for the path."""
import hashlib
import os
import random

import numpy as np
import pytest

import known_answers as KA
from conftest import GOLDEN
from oracle.pyoracle import SQ_ALL, SQ_BEST, SQ_CONVERT, SQ_FIRST, SQ_IGNORE, SQ_STREAM

MODE = dict(FIRST=SQ_FIRST, BEST=SQ_BEST, ALL=SQ_ALL)


def test_parse_known_answers(oracle):
    for pat, keys in KA.PARSE_OK:
        assert oracle.parse(pat) == (keys, 0), pat
    for pat, err in KA.PARSE_ERR:
        assert oracle.parse(pat) == (None, err), pat


def test_translate_tables(oracle):
    # seeqcore.h:89-111 as probed in SURVEY section 8a3
    for conv in (False, True):
        for b in range(256):
            c = chr(b)
            exp = {"A": 0, "a": 0, "C": 1, "c": 1, "G": 2, "g": 2, "T": 3, "t": 3, "U": 3, "u": 3, "N": 4,
                   "n": 4, "\0": 5, "\n": 6}.get(c, 4 if conv else 7)
            assert oracle.translate(b, conv) == exp


def test_trace_known_answers(oracle):
    for pat, tau, text, dist, mtm in KA.TRACE:
        d, t = oracle.trace(pat, tau, text)
        assert d == dist and t == mtm, (pat, d, t)


def test_string_match_known_answers(oracle):
    for pat, tau, text, mode, exp in KA.STRING_MATCH:
        assert oracle.string_match(pat, tau, text, MODE[mode]) == exp, (pat, text, mode)


def test_golden_string_cases(oracle, string_cases):
    for c in string_cases:
        got = oracle.string_match(c["pattern"], c["tau"], c["text"], c["options"])
        assert [list(h) for h in got] == c["hits"], c


def _compact(rec):
    return "".join("%d:%d-%d:%d\n" % (r[0], r[1], r[2] - 1, r[3]) for r in rec)


def test_golden_cli_compact_and_count(oracle, cli_cases):
    """The reference CLI's -f / -c outputs == oracle.buffer_scan on the same files."""
    n = 0
    for c in cli_cases:
        a = c["args"]
        if "-f" not in a and "-c" not in a:
            continue
        with open(os.path.join(GOLDEN, c["file"]), "rb") as f:
            buf = f.read()
        pat, d = a[-1], int(a[a.index("-d") + 1])
        opt = {"0": 0, "1": SQ_CONVERT, "2": SQ_IGNORE}[a[a.index("-x") + 1]]
        opt |= SQ_BEST if "-b" in a else SQ_ALL if "-a" in a else SQ_FIRST
        res = oracle.buffer_scan(pat, d, buf, opt, fasta=buf[:1] == b">")
        out = "%d\n" % res["nmatchlines"] if "-c" in a else _compact(res["records"])
        if "stdout" in c:
            assert out == c["stdout"], (c["file"], a)
        else:
            assert len(out) == c["nbytes"] and hashlib.sha256(out.encode()).hexdigest() == c["sha256"]
        n += 1
    assert n >= 40


def test_file_known_answers(oracle):
    buf = ("\n".join(KA.TESTDATA_LINES) + "\n").encode()
    with open(os.path.join(GOLDEN, "testdata.txt"), "rb") as f:
        assert f.read() == buf
    for pat, tau, kind, exp in KA.FILE_COUNTS:
        if kind == "COUNTLINES":
            assert oracle.buffer_scan(pat, tau, buf, SQ_FIRST)["nmatchlines"] == exp
        else:
            assert len(oracle.buffer_scan(pat, tau, buf, SQ_ALL)["records"]) == exp


def test_stream_option(oracle):
    # newline counted in coordinates under SQ_STREAM
    pat = "GATGTAGCGCGATTAGCCTG"
    text = "ACGTGATGTAGC\nGCGATTAGCCTGAAA\nTTT"
    assert oracle.string_match(pat, 3, text, SQ_STREAM | SQ_ALL) == [(4, 25, 0)]
    assert oracle.string_match(pat, 3, text, SQ_ALL) == []


def test_oracle_vs_reference_fuzz(oracle, reference):
    """Differential fuzz against the reference itself (only where oracle/_ref is built)."""
    import sys
    sys.path.insert(0, GOLDEN)
    from make_golden import plain, rand_pattern, rand_text
    rng = random.Random(77)
    for _ in range(400):
        pat = rand_pattern(rng)
        m = len(plain(pat))
        tau = rng.randint(0, min(m - 1, rng.choice([0, 1, 2, 3, 5, 8])))
        text = rand_text(rng, pat, tau, rng.choice([0, 1, 20, 150, 250]))
        for opt in (0, 1, 2, 3, 4 | 1, 8 | 2, 0x10 | 2, 0x10 | 8 | 1):
            assert oracle.string_match(pat, tau, text, opt) == reference.string_match(pat, tau, text, opt)


def test_synth_reads_shape(oracle):
    a = oracle.synth_reads(0, 1000, 150, "GATGTAGCGCGATTAGCCTG", 3)
    assert a.size == 1000 * 151 and np.all(a[150::151] == 10)
    b = oracle.synth_reads(500, 500, 150, "GATGTAGCGCGATTAGCCTG", 3)
    assert np.array_equal(a[500 * 151:], b)          # counter-based: any shard reproducible
    assert set(np.unique(a)) <= set(b"ACGTN\n")
