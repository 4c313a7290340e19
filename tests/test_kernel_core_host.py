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
