"""CPU: the C-ABI library loads, exports every symbol the public headers declare,
does the host-only work (pattern errors, seeq_t layout) and FAILS LOUDLY without a GPU."""
import ctypes as C
import errno
import os
import re
import subprocess

import pytest

import known_answers as KA
from conftest import ROOT


def test_exports_every_declared_symbol(capi):
    L = capi.lib()
    for name in capi.EXPORTS:
        assert hasattr(L, name), name
    # and the list itself covers every prototype in include/*.h
    declared = set()
    for h in ("libseeq.h", "seeq.h", "seeq_amd.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        declared |= set(re.findall(r"\b(seeq\w*|stackNew|stackAddMatch|recursive_merge)\s*\(", src))
    declared -= {"seeqarg_t", "seeqfile_t", "seeq_t"}
    assert declared <= set(capi.EXPORTS), declared - set(capi.EXPORTS)


def test_struct_layouts(capi):
    # reference libseeq.h:62-80 on LP64: match_t 24 B, seeq_t 80 B
    assert C.sizeof(capi.match_t) == 24
    assert C.sizeof(capi.seeq_t) == 80
    assert C.sizeof(capi.seeqdev_hit_t) == 16


def test_seeqnew_errors_are_host_side(capi):
    L = capi.lib()
    for pat, tau, err in KA.SEEQNEW_ERR:
        assert not L.seeqNew(pat.encode(), tau, 0)
        assert capi.seeqerr() == err, (pat, tau)
    for pat, err in KA.PARSE_ERR:
        assert not L.seeqNew(pat.encode(), 0, 0)
        assert capi.seeqerr() == err
    assert not L.seeqNew(b"", 0, 0) and capi.seeqerr() == 9


def test_error_strings(capi):
    L = capi.lib()
    L.seeqNew(b"ACG[AT]", 4, 0)
    assert L.seeqPrintError() == b"Pattern length must be larger than matching distance"
    L.seeqNew(b"Z", 0, 0)
    assert L.seeqPrintError() == b"Incorrect pattern (illegal character)"


def test_no_gpu_fails_loudly(capi):
    """Without a HIP device there is no matcher at all: seeqNew -> NULL, errno ENODEV."""
    L = capi.lib()
    if L.seeqdevDeviceCount() > 0:
        pytest.skip("a GPU is visible")
    C.set_errno(0)
    lib = C.CDLL(capi.LIB_PATH, use_errno=True)
    lib.seeqNew.restype = C.c_void_p
    lib.seeqNew.argtypes = [C.c_char_p, C.c_int, C.c_size_t]
    assert lib.seeqNew(b"ACGT", 1, 0) is None
    assert C.get_errno() == errno.ENODEV and capi.seeqerr() == 0
    assert b"no HIP device" in L.seeqdevLastError()
    lib.seeqdevScanNew.restype = C.c_void_p
    lib.seeqdevScanNew.argtypes = [C.c_void_p]
    assert lib.seeqdevScanNew(None) is None
    r = subprocess.run([capi.CLI_PATH, "-c", "CACAGAT", os.path.join(ROOT, "tests/golden/testdata.txt")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "error in 'seeqNew()'" in r.stderr and r.stdout == ""
    import seeq_amd
    with pytest.raises(seeq_amd.libseeq_exception):
        seeq_amd.compile("ACGT", 1)


def test_seeqopen_errors(capi):
    L = capi.lib()
    assert not L.seeqOpen(b"/nonexistent/invented.txt")
    assert capi.seeqerr() == errno.ENOENT            # raw errno, testset.c:1222 expects 2
    f = L.seeqOpen(os.path.join(ROOT, "tests/golden/fasta_small.txt").encode())
    assert f and f.contents.flags == 1 and f.contents.line == 0
    assert L.seeqClose(f) == 0
    f = L.seeqOpen(os.path.join(ROOT, "tests/golden/testdata.txt").encode())
    assert f and f.contents.flags == 0
    assert L.seeqClose(f) == 0


def test_cli_argument_errors(capi):
    def run(*a):
        return subprocess.run([capi.CLI_PATH] + list(a), capture_output=True, text=True)
    r = run()
    assert r.returncode == 0 and r.stderr.startswith("seeq-1.2\nUsage:")
    assert run("-v").stderr == "seeq-1.2\n"
    r = run("-d", "1", "-d", "2", "ACGT")
    assert r.returncode == 1 and "distance option set more than once" in r.stderr
    r = run("-x", "3", "ACGT")
    assert r.returncode == 1 and "nondna value must be either 0, 1 or 2" in r.stderr
    r = run("-n", "ACGT")
    assert r.returncode == 1 and "No output will be generated" in r.stderr
    r = run("-c")
    assert r.returncode == 1 and "not enough arguments" in r.stderr


def test_match_stack_utilities_vs_reference_vectors(capi):
    """stackNew / stackAddMatch / recursive_merge (libseeq.h; reference libseeq.c:355-424): host-only utilities that a
    caller of the drop-in may use although the reference itself no longer does.  300 random stacks per distance, merged by
    the reference (tests/golden/make_stack_golden.py): the same sq->match, in the same order, and the same leftovers."""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_stack_golden as G
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_stack_cases.json")))
    assert len(cases) == 300
    L = G.bind(C.CDLL(capi.LIB_PATH))
    libc = C.CDLL(None)
    nm = 0
    for c in cases:
        merged, left = G.run_case(L, libc, c)
        assert merged == c["merged"] and left == c["left"], c
        nm += len(merged)
    assert nm > 1000


def test_pack_reads_layout_on_the_host(capi):
    """seeqdevPackReads needs no GPU: the packed layout of include/seeq_amd.h bit by bit -- four bases per byte, the first in
    bits 7-6, code = (ASCII >> 1) & 3 (A 0, C 1, T/U 2, G 3), N through the mask (first base of a byte = bit 7), either case;
    the error returns."""
    import ctypes as C
    import numpy as np
    L = C.CDLL(capi.LIB_PATH)
    L.seeqdevPackReads.restype = C.c_long
    L.seeqdevPackReads.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
    text = b"ACGTN\nacgun\nTTTTT"                       # no newline behind the last read
    bases = np.zeros(3 * 2, dtype=np.uint8); nmask = np.zeros(3, dtype=np.uint8)
    assert L.seeqdevPackReads(text, len(text), 5, bases.ctypes.data, nmask.ctypes.data, 2, 1) == 3
    code = {"A": 0, "C": 1, "T": 2, "U": 2, "G": 3}
    for r, line in enumerate(["ACGTN", "ACGUN", "TTTTT"]):
        for i, ch in enumerate(line):
            got = (int(bases[2 * r + i // 4]) >> (6 - 2 * (i % 4))) & 3
            isn = (int(nmask[r]) >> (7 - i)) & 1
            assert isn == (ch == "N"), (r, i)
            if ch != "N":
                assert got == code[ch], (r, i)
    assert L.seeqdevPackReads(b"ACGT\nACG\n", 9, 4, bases.ctypes.data, nmask.ctypes.data, 1, 1) == -1      # a line of another length
    assert L.seeqdevPackReads(b"ACXT\n", 5, 4, bases.ctypes.data, nmask.ctypes.data, 1, 1) == -1           # not a base
    assert L.seeqdevPackReads(b"ACNT\n", 5, 4, bases.ctypes.data, None, 1, 0) == -1                        # an N and no mask
    assert L.seeqdevPackReads(b"", 0, 4, bases.ctypes.data, nmask.ctypes.data, 1, 1) == 0
