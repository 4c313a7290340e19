"""CPU: the per-line DEVICE functions (seeq_kernel_core.h), compiled for the host by
tests/host_harness.cpp, fuzzed against the oracle.  This checks the Myers column,
the acceptance rules and the reverse scan before any GPU time is spent; the GPU
tests then check the kernels that call these functions."""
import random
import sys

import numpy as np

import known_answers as KA
from conftest import GOLDEN
from oracle.pyoracle import SQ_ALL, SQ_BEST, SQ_CONVERT, SQ_COUNT, SQ_FAIL, SQ_FIRST, SQ_IGNORE, SQ_STREAM

ANY, COUNT, EMIT = 0, 1, 2


def run(H, oracle, pat, tau, text, opt, mode=EMIT, wforce=0):
    keys, err = oracle.parse(pat)
    assert keys is not None
    tb = text.encode("latin-1")
    out = np.zeros(3 * 4096, dtype=np.uint32)
    n = H.harness_scan(tb, len(tb), bytes(keys), len(keys), tau, opt, mode, wforce, out.ctypes.data, 4096)
    if mode != EMIT:
        return n
    return [tuple(int(x) for x in out[3 * k:3 * k + 3]) for k in range(n)]


def test_pattern_compiler(harness, oracle):
    import ctypes as C
    for pat, keys in KA.PARSE_OK:
        kb = C.create_string_buffer(len(pat) + 1)
        err = C.c_int(0)
        assert harness.harness_compile(pat.encode(), kb, C.byref(err)) == len(keys)
        assert list(kb.raw[:len(keys)]) == keys
    for pat, e in KA.PARSE_ERR + [(p, e) for p, t, e in KA.SEEQNEW_ERR if e in (2, 3, 4, 5)]:
        kb = C.create_string_buffer(len(pat) + 1)
        err = C.c_int(0)
        assert harness.harness_compile(pat.encode(), kb, C.byref(err)) == -1 and err.value == e, pat


def test_known_answers(harness, oracle):
    mode = dict(FIRST=SQ_FIRST, BEST=SQ_BEST, ALL=SQ_ALL)
    for pat, tau, text, mo, exp in KA.STRING_MATCH:
        assert run(harness, oracle, pat, tau, text, mode[mo]) == exp[::-1]


def test_golden_string_cases(harness, oracle, string_cases):
    for c in string_cases:
        got = run(harness, oracle, c["pattern"], c["tau"], c["text"], c["options"])
        assert [list(h) for h in got] == c["hits"][::-1], c


def test_fuzz_vs_oracle(harness, oracle):
    sys.path.insert(0, GOLDEN)
    from make_golden import plain, rand_pattern, rand_text
    rng = random.Random(4242)
    for _ in range(1500):
        pat = rand_pattern(rng)
        m = len(plain(pat))
        tau = rng.randint(0, min(m - 1, rng.choice([0, 1, 2, 3, 3, 5, 8])))
        text = rand_text(rng, pat, tau, rng.choice([0, 1, 5, 20, 60, 150, 250]))
        for mo in (SQ_FIRST, SQ_BEST, SQ_ALL, SQ_COUNT):
            nd = rng.choice([SQ_FAIL, SQ_CONVERT, SQ_IGNORE])
            st = rng.choice([0, 0, SQ_STREAM])
            opt = mo | nd | st
            exp = oracle.string_match(pat, tau, text, opt)[::-1]
            wf = rng.choice([0] + [w for w in (2, 4, 8, 16) if 32 * w >= m])
            assert run(harness, oracle, pat, tau, text, opt, EMIT, wf) == exp, (pat, tau, text, opt, wf)
        allh = oracle.string_match(pat, tau, text, (opt & ~3) | SQ_ALL)
        assert run(harness, oracle, pat, tau, text, opt, ANY) == (1 if allh else 0)
        assert run(harness, oracle, pat, tau, text, opt, COUNT) == len(allh)


def test_string_positions_in_blocks_vs_oracle(harness, oracle, string_cases):
    """Host side of k_string: the positions of a string shared out in blocks (block sizes 1, 2, 7 and 'whole string'),
    every block from a fresh column through sq_emit_window -- the function the kernel's threads call -- with the
    acceptance rules evaluated per position from three consecutive scores: the same hits as the oracle's line-long scan,
    for the golden string cases and random patterns / texts / options.  Strings with a skipped byte before the
    terminator (SQ_IGNORE, SQ_STREAM) are left to the one-lane scan by the kernel (-1 here)."""
    import ctypes as C
    H = harness
    H.harness_string_par.restype = C.c_long
    H.harness_string_par.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_size_t]

    def par(pat, tau, text, opt, block, wforce=0):
        keys, _ = oracle.parse(pat)
        tb = text.encode("latin-1")
        out = np.zeros(3 * 4096, dtype=np.uint32)
        n = H.harness_string_par(tb, len(tb), bytes(keys), len(keys), tau, opt, block, wforce, out.ctypes.data, 4096)
        return None if n < 0 else [tuple(int(x) for x in out[3 * k:3 * k + 3]) for k in range(n)]

    npar = 0
    for c in string_cases:
        for block in (1, 3, 1 << 20):
            got = par(c["pattern"], c["tau"], c["text"], c["options"], block)
            if got is not None:
                npar += 1
                assert [list(h) for h in got] == c["hits"][::-1], (c, block)
    sys.path.insert(0, GOLDEN)
    from make_golden import plain, rand_pattern, rand_text
    rng = random.Random(77)
    for _ in range(1200):
        pat = rand_pattern(rng)
        m = len(plain(pat))
        tau = rng.randint(0, min(m - 1, rng.choice([0, 1, 2, 3, 3, 5, 8])))
        text = rand_text(rng, pat, tau, rng.choice([0, 1, 5, 20, 60, 150, 250, 700]))
        for mo in (SQ_FIRST, SQ_BEST, SQ_ALL, SQ_COUNT):
            opt = mo | rng.choice([SQ_FAIL, SQ_CONVERT, SQ_CONVERT, SQ_IGNORE]) | rng.choice([0, 0, 0, SQ_STREAM])
            wf = rng.choice([0] + [w for w in (2, 4, 8, 16) if 32 * w >= m])
            got = par(pat, tau, text, opt, rng.choice([1, 2, 7, 64, 1 << 20]), wf)
            if got is None:
                continue
            npar += 1
            assert got == oracle.string_match(pat, tau, text, opt)[::-1], (pat, tau, text, opt, wf)
    assert npar > 3000


def test_stream_automaton_vs_oracle(harness, oracle):
    """Host side of k_stream: the complete Levenshtein automaton (seeq_dfa.h) walked chunk by chunk with a warm-up,
    exactly as the kernel decomposes the text, reports a first-hit event in a line iff the oracle finds a hit in it
    (clean text: A C G T N, both cases, newlines) -- for chunk sizes 128 / 64 / 16 and the smallest legal warm-up."""
    import ctypes as C
    harness.harness_dfa_stream.restype = C.c_long
    harness.harness_dfa_stream.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
    rng = random.Random(11)
    cases = [("GATGTAGCGCGATTAGCCTG", 3), ("GATTAGC", 1), ("CACAGAT", 3), ("ACGT", 1), ("A", 0), ("ACNNGT[AC]TTG", 2),
             ("GATGTAGCGCGATTAGCCTGAAAATG", 2), ("TTTTTTTT", 2), ("GATGTAGCGCGATTAG", 4)]
    for pat, tau in cases:
        keys, _ = oracle.parse(pat)
        m = len(keys)
        core = "".join("ACGT"[(k & -k).bit_length() - 1] if k != 0x1F else "N" for k in keys)
        lines = []
        for _ in range(400):
            n = rng.choice([0, 3, 20, 60, 150, 151, 400])
            t = [rng.choice("ACGT") for _ in range(n)]
            if n >= m and rng.random() < 0.5:
                c = list(core.replace("N", "A"))
                for _e in range(rng.randint(0, tau + 2)):
                    i = rng.randrange(len(c))
                    k = rng.randrange(3)
                    if k == 0:
                        c[i] = rng.choice("ACGT")
                    elif k == 1 and len(c) > 1:
                        del c[i]
                    else:
                        c.insert(i, rng.choice("ACGT"))
                p = rng.randrange(0, max(1, n - len(c) + 1))
                t[p:p + len(c)] = c
            if n and rng.random() < 0.1:
                t[rng.randrange(len(t))] = "N"
            s = "".join(t)[:n]
            lines.append(s.lower() if rng.random() < 0.05 else s)
        buf = ("\n".join(lines) + "\n").encode()
        exp = oracle.buffer_scan(pat, tau, buf, SQ_FIRST)
        want = sorted(set(int(x) for x in exp["records"][:, 0]))
        starts = np.cumsum([0] + [len(x) + 1 for x in lines])            # byte offset of every line
        for chunk in (128, 64, 16):
            warm = m + tau - 1
            out = np.zeros(1 << 16, dtype=np.uint64)
            ns = C.c_uint32(0)
            ne = harness.harness_dfa_stream(buf, len(buf), bytes(keys), m, tau, chunk, warm, out.ctypes.data, out.size,
                                            C.byref(ns))
            assert ne >= 0, (pat, tau)
            got = sorted(set(int(np.searchsorted(starts, int(p), side="right")) for p in out[:ne]))   # 1-based line numbers
            assert got == want, (pat, tau, chunk)
        if (pat, tau) == ("GATGTAGCGCGATTAGCCTG", 3):
            assert ns.value == 3342


def test_filter_automaton_is_a_superset(harness, oracle):
    """Host side of k_stream for patterns whose complete automaton does not fit: the partition filter automaton
    (seeq_dfa.h) walked chunk by chunk with its own warm-up flags EVERY line the oracle finds a hit in (and few
    others) -- BASELINE configs[4]'s pattern and random class/N patterns of 24..62 positions, distances up to 6."""
    import ctypes as C
    sys.path.insert(0, GOLDEN)
    from make_golden import mutate, plain
    harness.harness_dfa_filter.restype = C.c_long
    harness.harness_dfa_filter.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
    rng = random.Random(77)
    cases = [("GATG[TA]AGCNCGATTAGC[CG]TGAAAATGNGAGTAC[GAT]GCGCGA", 5, 0), ("GATGTAGCGCGATTAGCCTGAAAA", 3, 0),
             ("GATGTAGCGCGATTAGCCTG", 4, 0), ("GATGTAGCGCGATTAGCCTG", 3, 2), ("GATGTAGCGCGATTAGCCTG", 3, 4),
             ("GATGAAGCACGATTAGCCTGAAAATGAGAG", 5, 0)]
    for _ in range(6):
        m = rng.choice([24, 31, 40, 50, 62])
        pat = "".join("N" if rng.random() < 0.05 else "[" + "".join(rng.sample("ACGT", 2)) + "]" if rng.random() < 0.08
                      else rng.choice("ACGT") for _ in range(m))
        cases.append((pat, rng.randint(2, 6), 0))
    seen_filter = 0
    for pat, tau, parts in cases:
        keys, _ = oracle.parse(pat)
        m = len(keys)
        core = plain(pat)
        lines = []
        for i in range(500):
            n = rng.choice([0, 20, 150, 250, 400])
            t = "".join(rng.choice("ACGT") for _ in range(n))
            if i % 3 == 0 and n >= m:
                cp = mutate(rng, core, rng.randint(0, tau + 2))
                q = rng.randrange(n - len(cp) + 1) if n >= len(cp) else 0
                t = (t[:q] + cp + t[q + len(cp):])[:n]
            if i % 41 == 0 and n:
                q = rng.randrange(n)
                t = t[:q] + "N" + t[q + 1:]
            lines.append(t.lower() if i % 29 == 0 else t)
        buf = ("\n".join(lines) + "\n").encode()
        want = set(int(x) for x in oracle.buffer_scan(pat, tau, buf, SQ_FIRST)["records"][:, 0])
        assert len(want) > 20
        starts = np.cumsum([0] + [len(x) + 1 for x in lines])
        for chunk in (128, 64):
            out = np.zeros(1 << 16, dtype=np.uint64)
            info = (C.c_uint32 * 4)()
            ne = harness.harness_dfa_filter(buf, len(buf), bytes(keys), m, tau, parts, chunk, out.ctypes.data, out.size, info)
            if ne < 0:
                continue                                   # nothing fits: the per-line kernel serves this pattern
            got = set(int(np.searchsorted(starts, int(p), side="right")) for p in out[:ne])
            assert want <= got, (pat, tau, parts, chunk, sorted(want - got)[:5])
            assert info[2] <= 32 and info[0] <= 4000
            if info[1] > 1:
                seen_filter += 1
                # selectivity: beyond the planted near misses (copies with tau+1 / tau+2 edits, at most 1 line in 3 holds a
                # copy at all) the false candidates are what the library's accept-rate estimate predicts for random text
                extra = len(got - want)
                assert extra <= len(lines) // 3 - len(want) + 5 + 6 * (info[3] * 1e-9) * len(buf), (pat, tau, extra, info[3])
    assert seen_filter >= 8


def test_restart_table_names_every_occurrence_on_long_lines(harness, oracle):
    """Round 5, the long-line filter's restart table (seeq_dfa_restart_variant) under the kernel's chunked walk: EVERY occurrence the oracle
    reports on a long line (SQ_ALL records) holds a candidate -- a position s <= c <= e -- so that the exact pass may scan m + tau either side
    of a candidate and nothing else (seeq_exact1.h, ScanArgs.ll_restart).  The absorbing table of round 4 does not have the property (one
    candidate per chain and line): checked too, so that the test would notice if it were handed the wrong table."""
    import ctypes as C
    sys.path.insert(0, GOLDEN)
    from make_golden import mutate, plain
    harness.harness_dfa_filter_restart.restype = C.c_long
    harness.harness_dfa_filter_restart.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                   C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
    harness.harness_dfa_filter.restype = C.c_long
    harness.harness_dfa_filter.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
    rng = random.Random(505)
    full = "GATGTAGCGCGATTAGCCTGAAAATGCGAGTACGGCGCGAAT"
    cases = [(full[:34], 6, 0), (full[:34], 8, 0), (full[:27], 4, 0), (full[:27], 5, 0), (full, 8, 0), ("GATGTAGCGCGATTAGCCTGAAAA", 3, 0),
             ("GATG[TA]AGCNCGATTAGC[CG]TGAAAATGNGAGTAC[GAT]GCGCGA", 5, 0), (full[:34], 6, 2), (full[:30], 5, 3)]
    checked = absorbing_misses = 0
    for pat, tau, parts in cases:
        keys, _ = oracle.parse(pat)
        m = len(keys)
        core = plain(pat)
        lines = []
        for _ in range(3):                                  # long lines with copies all over: clean hits, near misses, tandem copies
            n = rng.choice([20000, 40000])
            t = [rng.choice("ACGT") for _ in range(n)]
            q = rng.randrange(0, 300)
            while q + 2 * m + 20 < n:
                cp = mutate(rng, core.replace("N", "A"), rng.randint(0, tau + 2))
                t[q:q + len(cp)] = list(cp)
                if rng.random() < 0.2:                      # a tandem copy right behind
                    cp2 = mutate(rng, core.replace("N", "A"), rng.randint(0, tau))
                    t[q + len(cp):q + len(cp) + len(cp2)] = list(cp2)
                q += rng.choice([m + 3, 150, 400, 1000])
            lines.append("".join(t[:n]))
        buf = ("\n".join(lines) + "\n").encode()
        starts = np.cumsum([0] + [len(x) + 1 for x in lines])
        rec = oracle.buffer_scan(pat, tau, buf, SQ_ALL)["records"]
        assert len(rec) > 50
        occ = [(int(starts[int(l) - 1]) + int(s_), int(starts[int(l) - 1]) + int(e) - 1) for l, s_, e, _d in rec.tolist()]     # absolute [s, e]
        for chain in (64, 128):
            out = np.zeros(1 << 18, dtype=np.uint64)
            info = (C.c_uint32 * 4)()
            ne = harness.harness_dfa_filter_restart(buf, len(buf), bytes(keys), m, tau, parts, chain, 0, out.ctypes.data, out.size, info)
            if ne < 0:
                continue
            cand = np.sort(out[:ne].astype(np.int64))
            for s_, e in occ:
                k = int(np.searchsorted(cand, s_, side="left"))
                assert k < len(cand) and cand[k] <= e, (pat, tau, parts, chain, s_, e, cand[max(0, k - 1):k + 2].tolist())
                checked += 1
            ne2 = harness.harness_dfa_filter(buf, len(buf), bytes(keys), m, tau, parts, chain, out.ctypes.data, out.size, info)
            cand2 = np.sort(out[:max(ne2, 0)].astype(np.int64))
            for s_, e in occ:
                k = int(np.searchsorted(cand2, s_, side="left"))
                absorbing_misses += not (k < len(cand2) and cand2[k] <= e)
    assert checked > 2000 and absorbing_misses > 100, (checked, absorbing_misses)


def _pair_events(harness, buf, keys, m, tau, chain=64, warm=0):
    import ctypes as C
    harness.harness_pair_walk.restype = C.c_long
    harness.harness_pair_walk.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
    out = np.zeros(1 << 17, dtype=np.uint64)
    info = (C.c_uint32 * 6)()
    ne = harness.harness_pair_walk(buf, len(buf), bytes(keys), m, tau, chain, warm, out.ctypes.data, out.size, info)
    return ne, out[:max(ne, 0)], list(info)


def test_pair_automaton_is_a_superset_and_its_first_candidate_bounds_the_scan(harness, oracle):
    """Host side of k_pair (seeq_dfa.h section 3): the pair automaton of the pattern's longest prefix that fits, walked
    two bytes per step in 64-byte chains with the kernel's warm-up and restart-on-accept, over text where every byte
    outside A C G T aliases onto a base (newline -> C, N -> G):
      (1) every line the oracle finds a hit in gets a candidate (position of a pair's second byte; the line is the one
          that byte lies in, a newline belonging to the line it ends);
      (2) the exact pass may start m + tau columns before a line's FIRST candidate: the oracle over that suffix reports
          the same hits (shifted) as over the whole line -- for SQ_ALL, hence for every match option."""
    sys.path.insert(0, GOLDEN)
    from make_golden import mutate, plain
    rng = random.Random(2026)
    cases = [("GATGTAGCGCGATTAGCCTG", 3), ("GATTAGC", 1), ("CACAGAT", 3), ("ACGT", 1), ("AC", 0), ("ACNNGT[AC]TTG", 2),
             ("GATGTAGCGCGATTAGCCTGAAAATG", 2), ("TTTTTTTT", 2), ("GATGTAGCGCGATTAG", 4), ("GATGTAGCGCGATTAGCCTGAAAA", 3),
             ("GATG[TA]AGCNCGATTAGC[CG]TGAAAATGNGAGTAC[GAT]GCGCGA", 5), ("AAAAAAAAAAAAAAAAAAAA", 3), ("ACGTACGTACGTACGT", 2)]
    for _ in range(8):
        m = rng.choice([6, 12, 20, 31, 45])
        pat = "".join("N" if rng.random() < 0.05 else "[" + "".join(rng.sample("ACGT", 2)) + "]" if rng.random() < 0.08
                      else rng.choice("ACGT") for _ in range(m))
        cases.append((pat, rng.randint(0, min(5, m - 2))))
    nwalked = 0
    for pat, tau in cases:
        keys, _ = oracle.parse(pat)
        m = len(keys)
        core = plain(pat)
        lines = []
        for i in range(500):
            n = rng.choice([0, 1, 3, 20, 60, 150, 151, 400])
            t = "".join(rng.choice("ACGT") for _ in range(n))
            if i % 2 == 0 and n >= m:
                cp = mutate(rng, core.replace("N", "A"), rng.randint(0, tau + 2))
                q = rng.choice([0, 0, max(0, n - len(cp)), rng.randrange(max(1, n - len(cp) + 1))])   # also flush with either end of the line
                t = (t[:q] + cp + t[q + len(cp):])[:n]
            if i % 17 == 0 and n:
                q = rng.randrange(n)
                t = t[:q] + "N" + t[q + 1:]
            lines.append(t.lower() if i % 29 == 0 else t)
        if len(set(core)) <= 2:
            # periodic patterns: occurrences overlap and follow each other across line ends, so walks restart inside their
            # warm-up windows -- runs of the pattern's own bases, m - tau - 1 .. m + 2 long, at either end of short lines
            lines = []
            for i in range(3000):
                unit = core[:2] if len(set(core)) == 2 else core[:1]
                run = (unit * m)[:rng.randint(max(1, m - tau - 1), m + 2)]
                junk = lambda k: "".join(rng.choice("ACGT") for _ in range(k))
                lines.append(rng.choice([run + junk(rng.randint(0, 50)), junk(rng.randint(0, 50)) + run,
                                         run, junk(rng.randint(0, 30)) + run + junk(rng.randint(0, 30))]))
        buf = ("\n".join(lines) + ("\n" if rng.random() < 0.5 else "")).encode()
        exp = oracle.buffer_scan(pat, tau, buf, SQ_ALL)
        want = {}
        for ln, st, en, di in exp["records"]:
            want.setdefault(int(ln), []).append((int(st), int(en), int(di)))
        starts = np.cumsum([0] + [len(x) + 1 for x in lines])            # byte offset of every line
        for chain, warm in ((64, 0), (64, 32), (16, 0)):
            ne, ev, info = _pair_events(harness, buf, keys, m, tau, chain, warm)
            if ne < 0:
                continue                                   # no prefix of >= tau + 2 positions fits: another kernel serves the pattern
            nwalked += 1
            assert info[0] + 5 <= 2047 and info[0] <= info[1] and tau + 2 <= info[2] <= m and info[3] <= 32
            assert info[5] > 1 or info[3] == info[2] + tau - 1
            first = {}
            for p in ev:
                p = int(p)
                ln = int(np.searchsorted(starts, p, side="right"))        # 1-based; a newline at p belongs to the line it ends
                first.setdefault(ln, p - int(starts[ln - 1]))
            missing = sorted(set(want) - set(first))
            assert not missing, (pat, tau, chain, warm, missing[:5])
            for ln, hits in want.items():
                col = first[ln]
                pos = max(0, col - (m + tau))
                line = lines[ln - 1]
                sub = oracle.string_match(pat, tau, line[pos:], SQ_ALL)[::-1]
                assert [(s + pos, e + pos, d) for s, e, d in sub] == hits, (pat, tau, chain, ln, col, pos)
        if (pat, tau) == ("GATGTAGCGCGATTAGCCTG", 3):
            assert info[2] == 17 and info[1] == 1839, info
    assert nwalked >= 40


def test_pair_automaton_names_every_occurrence_on_long_lines(harness, oracle):
    """Round 5, k_pair's long-line variant: with every flag of a chain kept, EVERY occurrence the oracle reports on a long line holds a
    candidate or ends on the byte before one (s <= c <= e + 1: a candidate is the second byte of a pair) -- what the window walk's
    m + tau + 1 columns either side of a candidate rest on (seeq_pair.h LL, seeq_exact1.h ll_restart)."""
    sys.path.insert(0, GOLDEN)
    from make_golden import mutate, plain
    rng = random.Random(506)
    full = "GATGTAGCGCGATTAGCCTGAAAATGCGAGTACGGCGCGAAT"
    checked = 0
    for pat, tau in ((full[:20], 3), (full[:27], 3), (full[:27], 5), (full[:34], 5), (full[:34], 7), ("GATG[TA]AGCNCGATTAGC[CG]TGAAAATGNGAGTAC[GAT]GCGCGA", 5)):
        keys, _ = oracle.parse(pat)
        m = len(keys)
        core = plain(pat)
        lines = []
        for _ in range(3):
            n = rng.choice([20000, 40000])
            t = [rng.choice("ACGT") for _ in range(n)]
            q = rng.randrange(0, 300)
            while q + 2 * m + 20 < n:
                cp = mutate(rng, core.replace("N", "A"), rng.randint(0, tau + 2))
                t[q:q + len(cp)] = list(cp)
                if rng.random() < 0.2:
                    cp2 = mutate(rng, core.replace("N", "A"), rng.randint(0, tau))
                    t[q + len(cp):q + len(cp) + len(cp2)] = list(cp2)
                q += rng.choice([m + 3, 150, 400, 1000])
            lines.append("".join(t[:n]))
        buf = ("\n".join(lines) + "\n").encode()
        starts = np.cumsum([0] + [len(x) + 1 for x in lines])
        rec = oracle.buffer_scan(pat, tau, buf, SQ_ALL)["records"]
        assert len(rec) > 50
        ne, ev, info = _pair_events(harness, buf, keys, m, tau)
        assert ne > 0, (pat, tau)
        cand = np.sort(ev.astype(np.int64))
        for l, s_, e, _d in rec.tolist():
            a, b = int(starts[int(l) - 1]) + int(s_), int(starts[int(l) - 1]) + int(e) - 1      # the occurrence, absolute [a, b]
            k = int(np.searchsorted(cand, a, side="left"))
            assert k < len(cand) and cand[k] <= b + 1, (pat, tau, a, b, cand[max(0, k - 1):k + 2].tolist())
            checked += 1
    assert checked > 1500, checked


def test_quad_automaton_flags_every_hit_read_and_bounds_its_window(harness, oracle):
    """Host side of the packed walk's four-bases-per-step table (seeq_dfa.h section 3b): every line is a read walked from the root,
    a packed byte per step.  (1) Every read the oracle finds a hit in gets a candidate; (2) the exact pass over the window
    [first candidate - (m + tau), last candidate + m + tau + 2) of the read reports the read's own hits (SQ_ALL, hence every
    option) -- what k_verify_packed scans; (3) the headline pattern is served by its two-part filter: 95 states."""
    import ctypes as C
    sys.path.insert(0, GOLDEN)
    from make_golden import mutate, plain
    rng = random.Random(4242)
    harness.harness_quad_walk.restype = C.c_long
    harness.harness_quad_walk.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
    cases = [("GATGTAGCGCGATTAGCCTG", 3), ("GATTAGC", 1), ("CACAGAT", 3), ("ACGT", 1), ("ACNNGT[AC]TTG", 2), ("GATGTAGCGCGATTAGCCTGAAAATG", 2),
             ("TTTTTTTT", 2), ("GATGTAGCGCGATTAG", 4), ("GATG[TA]AGCNCGATTAGC[CG]TGAAAATGNGAGTAC[GAT]GCGCGA", 5), ("AAAAAAAAAAAAAAAAAAAA", 3),
             ("ACGTACGTACGTACGT", 2), ("GATGTAGCGCGATTAGCCTG", 1), ("GATGTAGCGCGATTAGCCTG", 5)]
    for _ in range(8):
        m = rng.choice([6, 12, 20, 31, 45])
        pat = "".join("N" if rng.random() < 0.05 else "[" + "".join(rng.sample("ACGT", 2)) + "]" if rng.random() < 0.08
                      else rng.choice("ACGT") for _ in range(m))
        cases.append((pat, rng.randint(0, min(5, m - 2))))
    nwalked = 0
    for pat, tau in cases:
        keys, _ = oracle.parse(pat)
        m = len(keys)
        core = plain(pat)
        lines = []
        for i in range(600):
            n = rng.choice([1, 3, 20, 60, 150, 151, 250])
            t = "".join(rng.choice("ACGT") for _ in range(n))
            if i % 2 == 0 and n >= m:
                cp = mutate(rng, core.replace("N", "A"), rng.randint(0, tau + 2))
                q = rng.choice([0, 0, max(0, n - len(cp)), rng.randrange(max(1, n - len(cp) + 1))])
                t = (t[:q] + cp + t[q + len(cp):])[:n]
            if len(set(core)) <= 2 and i % 3 == 0:
                unit = core[:2] if len(set(core)) == 2 else core[:1]
                t = ((unit * m)[:rng.randint(max(1, m - tau - 1), m + 2)] + t)[:max(n, 1)]
            if i % 17 == 0:
                q = rng.randrange(n)
                t = t[:q] + "N" + t[q + 1:]                  # (an N aliases onto G in the walk: a superset)
            lines.append(t)
        buf = ("\n".join(lines) + "\n").encode()
        exp = oracle.buffer_scan(pat, tau, buf, SQ_ALL)
        want = {}
        for ln, st, en, di in exp["records"]:
            want.setdefault(int(ln), []).append((int(st), int(en), int(di)))
        out = np.zeros(1 << 18, dtype=np.uint64)
        info = (C.c_uint32 * 5)()
        ne = harness.harness_quad_walk(buf, len(buf), bytes(keys), m, tau, out.ctypes.data, out.size, info)
        if ne < 0:
            continue
        assert ne <= out.size
        nwalked += 1
        assert 1 <= info[0] <= 127 and info[0] <= info[1], list(info)
        starts = np.cumsum([0] + [len(x) + 1 for x in lines])
        first, last = {}, {}
        for p in out[:ne]:
            p = int(p)
            ln = int(np.searchsorted(starts, p, side="right"))
            col = p - int(starts[ln - 1])
            first.setdefault(ln, col)
            last[ln] = col
        missing = sorted(set(want) - set(first))
        assert not missing, (pat, tau, missing[:5])
        for ln, hits in want.items():
            line = lines[ln - 1]
            pos = max(0, first[ln] - (m + tau))
            stop = min(len(line), last[ln] + m + tau + 2)
            sub = oracle.string_match(pat, tau, line[pos:stop], SQ_ALL)[::-1]
            assert [(s + pos, e + pos, d) for s, e, d in sub] == hits, (pat, tau, ln, first[ln], last[ln])
        if (pat, tau) == ("GATGTAGCGCGATTAGCCTG", 3):
            assert info[0] == 95 and info[3] == 2, list(info)
    assert nwalked >= 12, nwalked


def test_quad_automaton_under_the_chunked_walk(harness, oracle):
    """The quad table under the ASCII walk of seeq_pair.h (QD): 64-byte chains, each warmed up over the automaton's own warm-up (whole words)
    and restarted when it accepts, newlines and N aliased onto bases.  (1) every line with a hit gets a candidate (the line of the
    candidate's byte, a newline belonging to the line it ends); (2) the oracle over the line from m + tau columns before its FIRST candidate
    reports the line's own hits -- what the exact pass scans."""
    import ctypes as C
    sys.path.insert(0, GOLDEN)
    from make_golden import mutate, plain
    rng = random.Random(777001)
    harness.harness_quad_chain_walk.restype = C.c_long
    harness.harness_quad_chain_walk.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                                C.POINTER(C.c_uint32)]
    cases = [("GATGTAGCGCGATTAGCCTG", 3), ("GATTAGC", 1), ("CACAGAT", 3), ("ACNNGT[AC]TTG", 2), ("GATGTAGCGCGATTAGCCTGAAAATG", 2), ("TTTTTTTT", 2),
             ("GATGTAGCGCGATTAG", 4), ("AAAAAAAAAAAAAAAAAAAA", 3), ("ACGTACGTACGTACGT", 2), ("GATGTAGCGCGATTAGCCTG", 1), ("GATGTAGCGCGATTAGCCTG", 5)]
    for _ in range(6):
        m = rng.choice([6, 12, 20, 31])
        pat = "".join("N" if rng.random() < 0.05 else "[" + "".join(rng.sample("ACGT", 2)) + "]" if rng.random() < 0.08
                      else rng.choice("ACGT") for _ in range(m))
        cases.append((pat, rng.randint(0, min(4, m - 2))))
    nwalked = 0
    for pat, tau in cases:
        keys, _ = oracle.parse(pat)
        m = len(keys)
        core = plain(pat)
        lines = []
        for i in range(500):
            n = rng.choice([0, 1, 3, 20, 60, 150, 151, 400])
            t = "".join(rng.choice("ACGT") for _ in range(n))
            if i % 2 == 0 and n >= m:
                cp = mutate(rng, core.replace("N", "A"), rng.randint(0, tau + 2))
                q = rng.choice([0, 0, max(0, n - len(cp)), rng.randrange(max(1, n - len(cp) + 1))])
                t = (t[:q] + cp + t[q + len(cp):])[:n]
            if i % 17 == 0 and n:
                q = rng.randrange(n)
                t = t[:q] + "N" + t[q + 1:]
            lines.append(t.lower() if i % 29 == 0 else t)
        if len(set(core)) <= 2:
            lines = []
            for i in range(3000):
                unit = core[:2] if len(set(core)) == 2 else core[:1]
                run = (unit * m)[:rng.randint(max(1, m - tau - 1), m + 2)]
                junk = lambda k: "".join(rng.choice("ACGT") for _ in range(k))
                lines.append(rng.choice([run + junk(rng.randint(0, 50)), junk(rng.randint(0, 50)) + run, run, junk(rng.randint(0, 30)) + run + junk(rng.randint(0, 30))]))
        buf = ("\n".join(lines) + ("\n" if rng.random() < 0.5 else "")).encode()
        exp = oracle.buffer_scan(pat, tau, buf, SQ_ALL)
        want = {}
        for ln, st, en, di in exp["records"]:
            want.setdefault(int(ln), []).append((int(st), int(en), int(di)))
        starts = np.cumsum([0] + [len(x) + 1 for x in lines])
        for chain, warm in ((64, 0), (64, 32), (16, 0)):
            out = np.zeros(1 << 18, dtype=np.uint64)
            info = (C.c_uint32 * 6)()
            ne = harness.harness_quad_chain_walk(buf, len(buf), bytes(keys), m, tau, chain, warm, out.ctypes.data, out.size, info)
            if ne < 0:
                continue
            assert ne <= out.size
            nwalked += 1
            first = {}
            for p in out[:ne]:
                p = int(p)
                ln = int(np.searchsorted(starts, p, side="right"))
                first.setdefault(ln, p - int(starts[ln - 1]))
            missing = sorted(set(want) - set(first))
            assert not missing, (pat, tau, chain, warm, missing[:5])
            for ln, hits in want.items():
                pos = max(0, first[ln] - (m + tau))
                line = lines[ln - 1]
                sub = oracle.string_match(pat, tau, line[pos:], SQ_ALL)[::-1]
                assert [(s + pos, e + pos, d) for s, e, d in sub] == hits, (pat, tau, chain, ln, first[ln], pos)
    assert nwalked >= 24, nwalked


def test_multi_pattern_automata_cover_every_pattern_of_every_line(harness, oracle):
    """Host side of the one-pass multi-pattern scan (seeq_dfa.h section 4): the UNION pair automaton of a barcode set walked
    in 64-byte chains as k_pair walks it, then the resolve automaton over each candidate line's window
    [first candidate - maxspan, last candidate + maxspan + 2):
      (1) every line in which the oracle finds ANY of the patterns gets a candidate;
      (2) the window's pattern mask holds every pattern the oracle finds in the line (exactly those when the automata carry
          the whole patterns);
      (3) each such pattern scanned over the window alone gives the line's own records (SQ_ALL, hence every option)."""
    import ctypes as C
    sys.path.insert(0, GOLDEN)
    from make_golden import mutate, plain
    rng = random.Random(77)
    harness.harness_multi_walk.restype = C.c_long
    harness.harness_multi_walk.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int,
                                           C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
    harness.harness_multi_resolve.restype = C.c_int
    harness.harness_multi_resolve.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_size_t, C.c_void_p]
    sets = [(["ACGTTGCA", "TTGACCGA", "GGCATTAC", "CAGTGTCA", "ATATCGCG", "GATTACAG"], [1, 1, 1, 1, 0, 1]),
            (["".join(rng.choice("ACGT") for _ in range(10)) for _ in range(16)], [1] * 16),
            (["".join(rng.choice("ACGT") for _ in range(rng.choice([8, 9, 11, 12]))) for _ in range(16)], [rng.choice([0, 1]) for _ in range(16)]),
            (["ACGTTGCA", "TG[AC]CANNGT", "GATGTAGCGCGATTAGCCTG", "AAAAAAAA"], [1, 1, 3, 2]),
            (["ACACACAC", "CACACACA", "ACACACACAC"], [1, 1, 2])]
    nsets = 0
    for pats, taus in sets:
        keysl = [oracle.parse(p)[0] for p in pats]
        ms = [len(k) for k in keysl]
        cat = bytes(b for k in keysl for b in k)
        npat = len(pats)
        cm = (C.c_int * npat)(*ms); ct = (C.c_int * npat)(*taus)
        lines = []
        for i in range(700):
            n = rng.choice([0, 5, 30, 60, 150, 151, 300])
            t = [rng.choice("ACGT") for _ in range(n)]
            for _ in range(rng.choice([0, 1, 1, 2, 3])):
                k = rng.randrange(npat)
                cp = mutate(rng, plain(pats[k]).replace("N", "A"), rng.randint(0, taus[k] + 1))
                if n >= len(cp):
                    q = rng.choice([0, n - len(cp), rng.randrange(n - len(cp) + 1)])
                    t[q:q + len(cp)] = list(cp)
            if i % 19 == 0 and n:
                t[rng.randrange(n)] = "N"
            t = "".join(t)[:n]
            lines.append(t.lower() if i % 31 == 0 else t)
        if len(set("".join(pats))) <= 2:
            lines += ["AC" * rng.randint(2, 9) + "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 20))) for _ in range(300)]
        buf = ("\n".join(lines) + "\n").encode()
        starts = np.cumsum([0] + [len(x) + 1 for x in lines])
        out = np.zeros(1 << 18, dtype=np.uint64)
        info = (C.c_uint32 * 8)()
        ne = harness.harness_multi_walk(buf, len(buf), cat, cm, ct, npat, 64, out.ctypes.data, out.size, info)
        assert ne >= 0, pats
        nsets += 1
        maxspan = int(info[6])
        assert maxspan == max(m + t for m, t in zip(ms, taus)) and info[0] + 5 <= 2047 and info[3] <= 32
        first, last = {}, {}
        for p in out[:ne]:
            p = int(p)
            ln = int(np.searchsorted(starts, p, side="right"))
            first.setdefault(ln, p - int(starts[ln - 1]))
            last[ln] = p - int(starts[ln - 1])
        want = [{} for _ in pats]
        for k, (pt, tau) in enumerate(zip(pats, taus)):
            for ln, st, en, di in oracle.buffer_scan(pt, tau, buf, SQ_ALL)["records"]:
                want[k].setdefault(int(ln), []).append((int(st), int(en), int(di)))
        hit_lines = set().union(*[set(w) for w in want])
        assert not (hit_lines - set(first)), (pats, sorted(hit_lines - set(first))[:5])
        cand = sorted(first)
        lo = np.array([starts[ln - 1] + max(0, first[ln] - maxspan) for ln in cand], dtype=np.uint64)
        hi = np.array([starts[ln - 1] + min(len(lines[ln - 1]), last[ln] + maxspan + 2) for ln in cand], dtype=np.uint64)
        masks = np.zeros(len(cand), dtype=np.uint32)
        assert harness.harness_multi_resolve(buf, cat, cm, ct, npat, lo.ctypes.data, hi.ctypes.data, len(cand), masks.ctypes.data) == 0
        for ln, a, b, mk in zip(cand, lo, hi, masks):
            truth = sum(1 << k for k in range(npat) if ln in want[k])
            assert (int(mk) & truth) == truth, (pats, ln, bin(int(mk)), bin(truth))
            if info[5] and "N" not in lines[ln - 1].upper():
                assert int(mk) == truth, (pats, ln, bin(int(mk)), bin(truth))
            a, b = int(a - starts[ln - 1]), int(b - starts[ln - 1])
            for k in range(npat):
                if truth >> k & 1:
                    sub = oracle.string_match(pats[k], taus[k], lines[ln - 1][a:b], SQ_ALL)[::-1]
                    assert [(s + a, e + a, d) for s, e, d in sub] == want[k][ln], (pats[k], ln, a, b)
    assert nsets == len(sets)


def test_scan_plans_of_the_baseline_configurations(harness):
    """The planner (seeq_amd/csrc/seeq_plan.h) is a pure host function -- run_segments only executes what it returns -- so which
    kernels serve the BASELINE configurations, the published sweep's cells and the FASTQ shape can be pinned without a GPU."""
    import ctypes as C
    H = harness
    H.harness_plan.restype = C.c_int
    H.harness_plan.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_int)]
    names = ["rc", "path", "fw", "use_stream", "use_pair", "use_myers", "filter", "stream_ll", "stream_sub", "stream_wu", "verify", "order2",
             "leaders", "window_ok", "ll_filter", "skip_back"]

    def plan(expr, tau, options, want, avg_line, flags=0, kernel=0):
        keys = C.create_string_buffer(2048)
        err = C.c_int(0)
        m = H.harness_compile(expr.encode(), keys, C.byref(err))
        assert m > 0
        out = (C.c_int * 16)()
        assert H.harness_plan(keys, m, tau, options, want, avg_line, flags, kernel, out) == 0
        return dict(zip(names, list(out)))
    SQ_BEST, SQ_ALL, SQ_CONVERT, SQ_IGNORE, FASTA = 1, 2, 4, 8, 0x100
    COUNTLINES, COUNTMATCH, RECORDS = 0, 1, 2
    head = "GATGTAGCGCGATTAGCCTG"
    # configs[1] / [2]: 150 bp reads, 20-mer, d = 3 -- k_pair over the prefix automaton, candidate windows, k_verify, three-launch ordering
    for opt, want in ((0, COUNTLINES), (SQ_BEST, RECORDS), (SQ_ALL, RECORDS), (0, COUNTMATCH)):
        p = plan(head, 3, opt, want, 151.0)
        assert (p["rc"], p["path"], p["fw"], p["use_pair"], p["filter"], p["stream_ll"]) == (0, 6, 1, 1, 1, 0), p
        assert p["verify"] == 1 and p["order2"] == 1 and p["window_ok"] == 1 and p["leaders"] == 0 and p["skip_back"] == 23, p
    # configs[4]: 40 positions, d = 5, 250 bp reads, --all: two-word column, a partition filter on k_pair
    p = plan("GATG[TA]AGCNCGATTAGC[CG]TGAAAATGNGAGTAC[GAT]GCGCGA", 5, SQ_ALL, RECORDS, 251.0)
    assert (p["path"], p["fw"], p["use_pair"], p["filter"], p["verify"], p["order2"]) == (6, 2, 1, 1, 1, 1), p
    # SQ_IGNORE on read-length lines: k_pair too since round 5 (a line that holds a skipped byte is named whole by a marker: seeq_pair.h IG), with
    # k_exact1 -- which knows how to skip -- behind it instead of k_verify; k_stream's skip variant where k_pair is told off or the input is FASTA
    p = plan(head, 3, SQ_BEST | SQ_IGNORE, RECORDS, 79.0)
    assert (p["path"], p["use_pair"], p["stream_sub"], p["filter"], p["verify"], p["order2"], p["window_ok"]) == (6, 1, 0, 1, 0, 1, 1), p
    p = plan(head, 3, SQ_BEST | SQ_IGNORE, RECORDS, 79.0, kernel=1)
    assert (p["path"], p["use_pair"], p["stream_sub"], p["filter"], p["verify"], p["order2"]) == (5, 0, 2, 0, 0, 1), p
    # FASTQ-shaped sample (foreign bytes), SQ_FAIL / SQ_CONVERT: k_pair since round 5 (dirty tiles remake their newline masks from registers,
    # k_verify looks at the bytes before a window); FASTA records with such a sample stay with k_stream
    for opt in (SQ_BEST, SQ_BEST | SQ_CONVERT):
        p = plan(head, 3, opt, RECORDS, 79.0, flags=16)
        assert (p["path"], p["use_pair"], p["stream_sub"], p["verify"], p["window_ok"]) == (6, 1, 0, 1, 1), p
    p = plan(head, 3, SQ_BEST | FASTA, RECORDS, 79.0, flags=16)
    assert (p["path"], p["use_pair"]) == (5, 0), p
    # the published sweep (chromosome lines): complete automaton / partition filter on long lines / Myers mode
    # (round 5: where the pair automaton is selective enough k_pair's long-line variant walks, its candidates go through the window walk -- ll_filter 2:
    #  windows of the restart walk; forced onto k_stream the plans of round 4 stay: the complete automaton, the filters' absorbing / restart tables)
    full = "GATGTAGCGCGATTAGCCTGAAAATGCGAGTACGGCGCGAAT"
    p = plan(head, 3, SQ_ALL, RECORDS, 1.3e8)
    assert (p["path"], p["use_pair"], p["stream_ll"], p["filter"], p["ll_filter"], p["leaders"], p["verify"], p["order2"], p["window_ok"]) == (6, 1, 1, 1, 2, 1, 0, 0, 0), p
    p = plan(head, 3, SQ_ALL, RECORDS, 1.3e8, kernel=1)
    assert (p["path"], p["stream_ll"], p["filter"], p["ll_filter"], p["leaders"], p["verify"], p["order2"]) == (5, 1, 0, 0, 1, 0, 0), p
    p = plan(full[:27], 3, SQ_ALL, RECORDS, 1.3e8, kernel=1)          # a selective filter: the absorbing table
    assert (p["path"], p["stream_ll"], p["filter"], p["ll_filter"], p["leaders"]) == (5, 1, 1, 1, 1), p
    p = plan(full[:27], 4, SQ_ALL, RECORDS, 1.3e8, kernel=1)          # filters that flag more than a position in 20 KB walk their restart table
    assert (p["path"], p["stream_ll"], p["filter"], p["ll_filter"], p["leaders"]) == (5, 1, 1, 2, 1), p
    p = plan(full[:34], 7, SQ_ALL, RECORDS, 1.3e8)
    assert (p["path"], p["fw"], p["stream_ll"], p["filter"], p["ll_filter"], p["leaders"]) == (6, 2, 1, 1, 2, 1), p
    p = plan(full[:34], 7, SQ_ALL, RECORDS, 1.3e8, kernel=1)
    assert (p["path"], p["fw"], p["stream_ll"], p["filter"], p["ll_filter"], p["leaders"]) == (5, 2, 1, 1, 2, 1), p
    p = plan(full, 8, SQ_ALL, RECORDS, 1.3e8)                          # (its pair automaton flags 2.4 positions per KB: k_stream's filter, 0.035)
    assert (p["path"], p["stream_ll"], p["filter"]) == (5, 1, 1), p
    p = plan(full, 15, SQ_ALL, RECORDS, 1.3e8)
    assert (p["path"], p["use_myers"], p["fw"], p["stream_ll"], p["filter"]) == (7, 1, 2, 1, 0), p
    # patterns beyond the two-word column, SQ_STREAM input: the generic path; a multi-pattern scan that is not k_pair's: a scan per pattern
    assert plan("ACGT" * 20, 4, 0, COUNTLINES, 151.0)["path"] == 1
    assert plan(head, 3, 0x10, COUNTLINES, 151.0)["path"] in (1, 3)
    assert plan(head, 3, SQ_IGNORE, COUNTLINES, 151.0, flags=32)["rc"] == -2
    # a context that has met a line of a whole tile (force_ll) goes to the long-line variants: k_pair's own (round 5), k_stream's under SQ_IGNORE
    p = plan(head, 3, SQ_BEST, RECORDS, 151.0, flags=1)
    assert (p["path"], p["use_pair"], p["stream_ll"], p["window_ok"], p["order2"]) == (6, 1, 1, 0, 0), p
    p = plan(head, 3, SQ_BEST | SQ_IGNORE, RECORDS, 151.0, flags=1)
    assert (p["use_pair"], p["stream_ll"]) == (0, 1) or p["path"] in (1, 3), p
