"""Python surface of the reference's `seeq` module, on top of the GPU library.

Mirrors reference src/seeqmodule.c: `compile()` (:1060-1094) returns a
SeeqObject whose match/matchBest/matchAll/matchIter/matchPrefix/matchSuffix
(:736-976) call seeqStringMatch of libseeq_amd.so (HIP kernels) and wrap the
hits in SeeqMatch (:324-623) / SeeqIter (:89-221).  Same names, argument
meaning, return values (None when nothing matches) and exception classes.
"""
import ctypes as C

from . import _capi

__version__ = "1.2"          # reference setup.py:4-5, seeqmodule.c:1139


class exception(Exception):
    """seeq.exception (reference seeqmodule.c:1142-1144)."""


class libseeq_exception(Exception):
    """libseeq.exception (reference seeqmodule.c:1146-1148)."""


def _utf8(s):
    if not isinstance(s, str):
        raise TypeError("a str is required")
    b = s.encode("utf-8")
    if b"\0" in b:
        raise ValueError("embedded null character")
    return b


class SeeqMatch:
    """Hits of one string: `matchlist` = [(start, end, dist)], `string`."""

    def __init__(self, string, matchlist):
        if not string:
            raise exception("Empty string")
        self.string = string
        self.matchlist = matchlist
        self._b = string.encode("utf-8")

    def _tok(self, lo, hi):
        return self._b[lo:hi].decode("utf-8", errors="replace")

    def tokenize(self):
        """Prefix, match, prefix, match, ..., suffix (reference seeqmodule.c:349-444)."""
        if not self._b or not self.matchlist:
            return None
        out, pos = [], 0
        for (s, e, _d) in self.matchlist:
            if s - pos >= 0:
                out.append(self._tok(pos, s))
            if e - s > 0:
                out.append(self._tok(s, e))
            pos = e
        if len(self._b) - pos >= 0:
            out.append(self._tok(pos, len(self._b)))
        return tuple(out)

    def split(self):
        """Non-empty fragments between the matches (reference seeqmodule.c:446-529)."""
        if not self._b or not self.matchlist:
            return None
        out, pos = [], 0
        for (s, e, _d) in self.matchlist:
            if s - pos > 0:
                out.append(self._tok(pos, s))
            pos = e
        if len(self._b) - pos > 0:
            out.append(self._tok(pos, len(self._b)))
        return tuple(out)

    def matches(self):
        """The matched substrings (reference seeqmodule.c:531-603)."""
        if not self._b or not self.matchlist:
            return None
        return tuple(self._tok(s, e) for (s, e, _d) in self.matchlist if e - s > 0)


class SeeqIter:
    """Iterator over the matched parts of a string (reference seeqmodule.c:89-214)."""

    def __init__(self, sqobj, string, match_iter=1):
        self.string = string
        self._b = _utf8(string)
        self._match_iter = match_iter
        self._last = 0
        hits = sqobj._run(self._b, _capi.SQ_ALL)      # left to right, as seeqMatchIter pops them
        self._hits = list(hits)
        self._i = 0

    def __iter__(self):
        return self

    def __next__(self):
        while True:
            if self._i < len(self._hits):
                s, e, _d = self._hits[self._i]
                self._i += 1
                last, self._last = self._last, e
                if self._match_iter == 1:
                    return self._b[s:e].decode("utf-8", errors="replace")
                if s - last > 0:
                    return self._b[last:s].decode("utf-8", errors="replace")
                continue
            if self._match_iter == 1:
                raise StopIteration
            if len(self._b) - self._last > 0:
                tail = self._b[self._last:].decode("utf-8", errors="replace")
                self._last = len(self._b)
                return tail
            raise StopIteration


class SeeqObject:
    """A compiled pattern (reference seeqmodule.c:673-1057)."""

    def __init__(self, pattern, mismatches, sq, options):
        if not pattern:
            raise exception("Empty pattern")
        if mismatches < 0:
            raise exception("Mismatches must be a non-negative integer")
        if not sq:
            raise exception("NULL reference to DFA pointer")
        self.pattern = pattern
        self.mismatches = mismatches
        self._sq = sq
        self._options = options
        self._lib = _capi.lib()

    def __del__(self):
        sq, self._sq = getattr(self, "_sq", None), None
        if sq:
            try:
                self._lib.seeqFree(sq)
            except Exception:
                pass

    def _run(self, data, match_opt):
        """seeqStringMatch + drain seeqMatchIter -> hits left to right."""
        n = self._lib.seeqStringMatch(data, self._sq, match_opt | self._options)
        if n < 0:
            raise libseeq_exception(_capi.error_text())
        hits = []
        while True:
            m = self._lib.seeqMatchIter(self._sq)
            if not m:
                break
            hits.append((int(m.contents.start), int(m.contents.end), int(m.contents.dist)))
        return hits

    def _seeqmatch(self, string, match_opt):
        data = _utf8(string)
        hits = self._run(data, match_opt)
        if not hits:
            return None
        return SeeqMatch(string, hits)

    def match(self, string):
        return self._seeqmatch(string, _capi.SQ_FIRST)

    def matchBest(self, string):
        return self._seeqmatch(string, _capi.SQ_BEST)

    def matchAll(self, string):
        return self._seeqmatch(string, _capi.SQ_ALL)

    def matchIter(self, string):
        return SeeqIter(self, string, 1)

    # ---- batched extension (not in the reference module): one GPU scan for many strings ----
    def _batch(self, strings, match_opt):
        """strings: iterable of str without newlines/NULs.  Returns a list with, per string, the list of
        (start, end, dist) hits in left-to-right order ([] when nothing matches).  The whole batch is one
        device scan (seeqdevScanHost over the newline-joined strings) instead of one GPU round trip per
        string."""
        from . import device as dev
        strings = list(strings)
        if not strings:
            return []
        try:                                      # one join + one encode for the whole batch (per-string work is what
            buf = ("\n".join(strings) + "\n").encode("utf-8")      # costs here: the device scan takes a fraction of a ms)
        except TypeError:
            raise TypeError("a str is required") from None
        if b"\0" in buf:
            raise ValueError("embedded null character")
        if buf.count(b"\n") != len(strings):
            raise ValueError("batched matching takes one string per line: no embedded newlines")
        data = strings
        if not hasattr(self, "_scanner"):
            self._scanner = dev.Scanner()
            self._devpat = type("P", (), {"handle": self._lib.seeqdevPatternOf(self._sq)})()
        res = self._scanner.scan_host(self._devpat, buf, match_opt | self._options, dev.WANT_RECORDS)
        out = [[] for _ in data]
        for line, s, e, d in res["records"]:
            out[int(line) - 1].append((int(s), int(e), int(d)))
        return out

    def matchBatch(self, strings):
        return self._batch(strings, _capi.SQ_FIRST)

    def matchBestBatch(self, strings):
        return self._batch(strings, _capi.SQ_BEST)

    def matchAllBatch(self, strings):
        return self._batch(strings, _capi.SQ_ALL)

    def _best0(self, string):
        data = _utf8(string)
        n = self._lib.seeqStringMatch(data, self._sq, _capi.SQ_BEST | self._options)
        if n < 0:
            raise libseeq_exception(_capi.error_text())
        if n == 0:
            return data, None
        m = self._sq.contents.match[0]            # read directly, like seeqmodule.c:778-829
        return data, (int(m.start), int(m.end))

    def matchPrefix(self, string, include_match=True):
        data, m = self._best0(string)
        if m is None:
            return None
        cut = m[1] if include_match else m[0]
        return data[:cut].decode("utf-8", errors="replace")

    def matchSuffix(self, string, include_match=True):
        data, m = self._best0(string)
        if m is None:
            return None
        cut = m[0] if include_match else m[1]
        return data[cut:].decode("utf-8", errors="replace")


def compile(pattern, mismatches, mode=0, memory=0):   # noqa: A001  (same name as the reference)
    """compile(pattern, distance[, mode=0[, memoryMB=0]]) (reference seeqmodule.c:1060-1094)."""
    L = _capi.lib()
    options = _capi.SQ_CONVERT if mode == 0 else _capi.SQ_IGNORE     # seeqmodule.c:1079-1080
    sq = L.seeqNew(_utf8(pattern), int(mismatches), int(memory) * 1024 * 1024)
    if not sq:
        raise libseeq_exception(_capi.error_text())
    try:
        return SeeqObject(pattern, int(mismatches), sq, options)
    except Exception:
        L.seeqFree(sq)
        raise
