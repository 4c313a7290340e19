"""ctypes bindings for the parity checker.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  `Oracle` wraps oracle/liboracle.so (own CPU restatement);
`Reference` wraps oracle/_ref/libseeq_ref.so (the reference itself, compiled
from /root/reference by oracle/Makefile) when that file is present.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
REF_LIB = os.path.join(HERE, "_ref", "libseeq_ref.so")
REF_BIN = os.path.join(HERE, "_ref", "seeq_ref")

SQ_FIRST, SQ_BEST, SQ_ALL, SQ_COUNT = 0, 1, 2, 3
SQ_FAIL, SQ_CONVERT, SQ_IGNORE = 0, 4, 8
SQ_LINES, SQ_STREAM = 0, 0x10


def build(ref=True):
    """Compile liboracle.so (always) and _ref/ (when /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    if ref and os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


class _Match(C.Structure):
    _fields_ = [("start", C.c_size_t), ("end", C.c_size_t), ("dist", C.c_size_t)]


class Oracle:
    def __init__(self):
        if not os.path.exists(LIB):
            build(ref=False)
        L = C.CDLL(LIB)
        L.orc_parse.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
        L.orc_parse.restype = C.c_int
        L.orc_translate.argtypes = [C.c_ubyte, C.c_int]
        L.orc_translate.restype = C.c_int
        L.orc_string_match.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                       C.POINTER(_Match), C.c_size_t]
        L.orc_string_match.restype = C.c_long
        L.orc_trace.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_int, C.c_int,
                                C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_trace.restype = None
        L.orc_buffer_scan.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                      C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_buffer_scan.restype = C.c_long
        L.orc_synth_reads.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_char_p,
                                      C.c_int, C.c_int, C.c_uint64]
        L.orc_synth_reads.restype = None
        self.L = L

    def parse(self, expr):
        """-> (keys list, 0) or (None, seeqerr)."""
        if isinstance(expr, str):
            expr = expr.encode()
        keys = C.create_string_buffer(max(len(expr), 1))
        err = C.c_int(0)
        m = self.L.orc_parse(expr, keys, C.byref(err))
        if m < 0:
            return None, err.value
        return [keys.raw[i] for i in range(m)], 0

    def translate(self, byte, convert=False):
        return self.L.orc_translate(byte, 1 if convert else 0)

    def _keys(self, pattern):
        keys, err = self.parse(pattern)
        if keys is None:
            raise ValueError("pattern error %d" % err)
        return bytes(keys), len(keys)

    def string_match(self, pattern, tau, data, options=0):
        """-> list of (start, end, dist) in sq->match[] order (last hit first)."""
        keys, m = self._keys(pattern)
        if isinstance(data, str):
            data = data.encode("latin-1")
        cap = 64
        while True:
            out = (_Match * cap)()
            n = self.L.orc_string_match(data, keys, m, tau, options, out, cap)
            if n < 0:
                raise MemoryError
            if n <= cap:
                return [(out[i].start, out[i].end, out[i].dist) for i in range(n)]
            cap = n

    def trace(self, pattern, tau, data):
        keys, m = self._keys(pattern)
        if isinstance(data, str):
            data = data.encode()
        n = len(data)
        d = (C.c_int * n)()
        t = (C.c_int * n)()
        self.L.orc_trace(data, n, keys, m, tau, d, t)
        return list(d), list(t)

    def buffer_scan(self, pattern, tau, buf, options=0, fasta=False):
        """-> dict(records=ndarray[n,4] u64 (line,start,end,dist), line_nhits, nlines, nmatchlines)."""
        keys, m = self._keys(pattern)
        if isinstance(buf, (bytes, bytearray)):
            arr = np.frombuffer(bytes(buf), dtype=np.uint8)
        else:
            arr = np.ascontiguousarray(buf, dtype=np.uint8)
        nb = arr.size
        line_cap = int(np.count_nonzero(arr == 10)) + 1
        line_nhits = np.zeros(line_cap, dtype=np.uint32)
        nl = C.c_uint64(0)
        nm = C.c_uint64(0)
        cap = 1024
        while True:
            rec = np.zeros((cap, 4), dtype=np.uint64)
            n = self.L.orc_buffer_scan(arr.ctypes.data if nb else None, nb, keys, m, tau, options,
                                       1 if fasta else 0, rec.ctypes.data, cap,
                                       line_nhits.ctypes.data, line_cap, C.byref(nl), C.byref(nm))
            if n < 0:
                raise MemoryError
            if n <= cap:
                break
            cap = n
        return dict(records=rec[:n].copy(), line_nhits=line_nhits[:nl.value].copy(),
                    nlines=nl.value, nmatchlines=nm.value)

    def synth_reads(self, first, n, length, pattern_plain, tau, seed=0x5EE92025):
        if isinstance(pattern_plain, str):
            pattern_plain = pattern_plain.encode()
        out = np.empty(n * (length + 1), dtype=np.uint8)
        self.L.orc_synth_reads(out.ctypes.data, first, n, length, pattern_plain,
                               len(pattern_plain), tau, seed)
        return out


class _SeeqT(C.Structure):
    # libseeq.h:68-80 (public, field-accessed by callers).
    _fields_ = [("hits", C.c_size_t), ("stacksize", C.c_size_t), ("match", C.POINTER(_Match)),
                ("bufsz", C.c_size_t), ("string", C.c_char_p), ("tau", C.c_int), ("wlen", C.c_int),
                ("keys", C.POINTER(C.c_char)), ("rkeys", C.POINTER(C.c_char)),
                ("dfa", C.c_void_p), ("rdfa", C.c_void_p)]


class Reference:
    """The reference's libseeq, when oracle/_ref/libseeq_ref.so exists."""

    @staticmethod
    def available():
        return os.path.exists(REF_LIB)

    def __init__(self):
        L = C.CDLL(REF_LIB)
        L.seeqNew.argtypes = [C.c_char_p, C.c_int, C.c_size_t]
        L.seeqNew.restype = C.POINTER(_SeeqT)
        L.seeqFree.argtypes = [C.POINTER(_SeeqT)]
        L.seeqFree.restype = None
        L.seeqStringMatch.argtypes = [C.c_char_p, C.POINTER(_SeeqT), C.c_int]
        L.seeqStringMatch.restype = C.c_long
        self.L = L
        self._cache = {}

    def seeqerr(self):
        return C.c_int.in_dll(self.L, "seeqerr").value

    def new(self, pattern, tau, mem=0):
        if isinstance(pattern, str):
            pattern = pattern.encode()
        return self.L.seeqNew(pattern, tau, mem)

    def string_match(self, pattern, tau, data, options=0):
        key = (pattern, tau)
        sq = self._cache.get(key)
        if sq is None:
            sq = self.new(pattern, tau)
            if not sq:
                raise ValueError("seeqNew failed: seeqerr=%d" % self.seeqerr())
            if len(self._cache) > 64:
                for v in self._cache.values():
                    self.L.seeqFree(v)
                self._cache.clear()
            self._cache[key] = sq
        if isinstance(data, str):
            data = data.encode("latin-1")
        n = self.L.seeqStringMatch(data, sq, options)
        if n < 0:
            raise RuntimeError("seeqStringMatch failed")
        m = sq.contents.match
        return [(m[i].start, m[i].end, m[i].dist) for i in range(n)]
