"""Batched device-level scans (include/seeq_amd.h) for Python callers.

`Pattern` is a compiled pattern living in HBM; `Scanner` owns a HIP stream +
workspace and runs the whole per-file hot path over a text buffer that is
already in HBM (a torch uint8 CUDA tensor, or any device pointer).
torch is used only for memory and streams; all compute is in libseeq_amd.so.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import (SQ_ALL, SQ_BEST, SQ_CONVERT, SQ_FAIL, SQ_FIRST, SQ_IGNORE, SEEQDEV_FASTA,  # noqa: F401
                    WANT_COUNTLINES, WANT_COUNTMATCH, WANT_RECORDS)


class SeeqDeviceError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise SeeqDeviceError(_capi.error_text())


def pack_reads(text, read_len, with_nmask=True):
    """ASCII reads (bytes, one per line, each exactly read_len bases) -> (bases, nmask, nreads): numpy uint8 arrays in the
    packed layout of seeq_amd.h (stride ceil(read_len / 4), nstride ceil(read_len / 8)); nmask is None without with_nmask."""
    stride, nstride = (read_len + 3) // 4, (read_len + 7) // 8
    nmax = len(text) // read_len + 1
    bases = np.zeros(nmax * stride, dtype=np.uint8)
    nmask = np.zeros(nmax * nstride, dtype=np.uint8) if with_nmask else None
    n = _capi.lib().seeqdevPackReads(text, len(text), read_len, bases.ctypes.data, nmask.ctypes.data if with_nmask else None, stride, nstride)
    if n < 0:
        raise SeeqDeviceError("seeqdevPackReads: a line of another length, or a byte that is not A C G T U N")
    return bases[:n * stride], (nmask[:n * nstride] if with_nmask else None), int(n)


class TextBuffer:
    """Device memory for resident text chosen by measurement (seeqdevTextAllocInfo): the scan kernel's speed follows the physical pages a
    buffer gets, so up to `candidates` allocations are probed and the fastest kept.  `ptr` is the device address, `probe_ms` the candidates'
    scan-kernel times (empty when nothing was probed; candidate 0 is the plain allocation), `chosen` the index of the one kept,
    `allocated_bytes` the size of the allocation behind it (a power-of-two block may be up to twice `nbytes`), `probe_peak_bytes` what the
    call held on the device at its peak.  Contents undefined; free() or the garbage collector releases it.  `tensor()`: a torch uint8
    view of the first `nbytes` bytes (plumbing for callers that slice / copy with torch; the buffer must outlive the view)."""

    def __init__(self, nbytes, candidates=12, scanner=None):
        """scanner: the Scanner that will scan the text (seeqdevTextAllocFor: the candidates are probed with ITS workspace -- reserve() it first);
        None: a context made for the probe."""
        info = _capi.seeqdev_textinfo_t()
        if scanner is not None:
            p = _capi.lib().seeqdevTextAllocFor(scanner._h, int(nbytes), int(candidates), C.byref(info))
        else:
            p = _capi.lib().seeqdevTextAllocInfo(int(nbytes), int(candidates), C.byref(info))
        if not p:
            raise SeeqDeviceError(_capi.error_text())
        self.ptr, self.nbytes = int(p), int(nbytes)
        self.probe_ms = [float(info.probe_ms[i]) for i in range(info.nprobed)]
        self.chosen = int(info.chosen)
        self.allocated_bytes = int(info.allocated_bytes)
        self.probe_peak_bytes = int(info.probe_peak_bytes)

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2, "strides": None}

    def tensor(self, device=None):
        import torch
        return torch.as_tensor(self, device=device if device is not None else "cuda")

    def free(self):
        if self.ptr:
            _capi.lib().seeqdevTextFree(C.c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def pack_reads_device(text_ptr, nreads, read_len, bases_ptr, nmask_ptr=None, stream=None):
    """ASCII reads in HBM (nreads lines of read_len bases + newline) -> the packed layout in HBM (device pointers)."""
    _check(_capi.lib().seeqdevPackReadsDevice(C.c_void_p(text_ptr), nreads, read_len, C.c_void_p(bases_ptr), C.c_void_p(nmask_ptr) if nmask_ptr else None,
                                              (read_len + 3) // 4, (read_len + 7) // 8, C.c_void_p(stream) if stream else None))


def device_count():
    return _capi.lib().seeqdevDeviceCount()


def plain_pattern(pattern):
    """One concrete base per pattern position (first member of a class, 'A' for N):
    the string the synthetic-read generator plants."""
    out, i = [], 0
    while i < len(pattern):
        c = pattern[i]
        if c == '[':
            j = pattern.index(']', i)
            if j > i + 1:
                out.append(pattern[i + 1].upper().replace('U', 'T').replace('N', 'A'))
            i = j + 1
        else:
            out.append('A' if c in 'Nn' else c.upper().replace('U', 'T'))
            i += 1
    return ''.join(out)


class Pattern:
    def __init__(self, pattern, tau):
        self._lib = _capi.lib()
        self.pattern, self.tau = pattern, tau
        self._sq = self._lib.seeqNew(pattern.encode(), int(tau), 0)
        if not self._sq:
            raise SeeqDeviceError("seeqNew(%r, %d): %s" % (pattern, tau, _capi.error_text()))
        self.wlen = self._sq.contents.wlen
        self.handle = self._lib.seeqdevPatternOf(self._sq)

    def close(self):
        sq, self._sq = self._sq, None
        if sq:
            self._lib.seeqFree(sq)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scanner:
    def __init__(self, stream=None):
        """stream: a hipStream_t as int (e.g. torch.cuda.current_stream().cuda_stream) or None.  None and 0 (torch's
        default stream is the legacy null stream, handle 0) both give the scanner a private stream of the default,
        "blocking" kind: HIP orders it after work already queued on the null stream and the null stream after it."""
        self._lib = _capi.lib()
        self._h = self._lib.seeqdevScanNew(C.c_void_p(stream) if stream else None)
        if not self._h:
            raise SeeqDeviceError("seeqdevScanNew: " + _capi.error_text())

    def close(self):
        h, self._h = self._h, None
        if h:
            self._lib.seeqdevScanFree(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reserve(self, max_bytes=0, max_lines=0, max_hitlines=0, max_records=0):
        _check(self._lib.seeqdevScanReserve(self._h, max_bytes, max_lines, max_hitlines, max_records))

    def set_profiling(self, on=True):
        _check(self._lib.seeqdevScanSetProfiling(self._h, 1 if on else 0))

    def set_line_hint(self, avg_bytes_per_line):
        _check(self._lib.seeqdevScanSetLineHint(self._h, float(avg_bytes_per_line)))

    def last_path(self):
        return {1: "generic", 3: "fused", 5: "fused", 6: "fused", 7: "fused", 8: "packed"}.get(self._lib.seeqdevScanLastPath(self._h), "none")

    def last_kernel(self):
        return {1: "k_forward", 3: "k_direct", 5: "k_stream", 6: "k_pair", 7: "k_myers", 8: "k_packed"}.get(self._lib.seeqdevScanLastPath(self._h), "none")

    def last_filter(self):
        """True when the last k_stream run walked a partition filter automaton (candidates verified by the exact pass)."""
        return bool(self._lib.seeqdevScanLastFilter(self._h))

    def last_packed_quad(self):
        """True when the last packed run walked the quad table (four bases per table step)."""
        return bool(self._lib.seeqdevScanLastPackedQuad(self._h))

    def last_times_ms(self):
        ms = (C.c_float * 4)()
        _check(self._lib.seeqdevScanLastTimes(self._h, ms))
        return dict(index=ms[0], forward=ms[1], exact=ms[2], total=ms[3],
                    forward_launches=self._lib.seeqdevScanLastLaunches(self._h))

    def last_clock_mhz(self):
        """Core clock the last run's scan launches ran at (k_pair's own clock readings; profiling on), 0 when not measured."""
        return float(self._lib.seeqdevScanLastClockMHz(self._h))

    def last_launch_times_ms(self):
        """Duration of every forward-scan launch of the last fetched scan (profiling on), in launch order."""
        n = self._lib.seeqdevScanLastLaunches(self._h)
        ms = (C.c_float * max(1, n))()
        got = self._lib.seeqdevScanLastLaunchTimes(self._h, ms, n)
        return [float(ms[i]) for i in range(max(0, min(got, n)))]

    def run(self, pattern, d_ptr, nbytes, options=0, want=WANT_COUNTLINES):
        """Enqueue the scan (asynchronous)."""
        _check(self._lib.seeqdevScanRun(self._h, pattern.handle, C.c_void_p(d_ptr), nbytes, options, want))

    def fetch(self):
        cnt = _capi.seeqdev_counts_t()
        _check(self._lib.seeqdevScanFetch(self._h, C.byref(cnt)))
        return dict(nlines=cnt.nlines, nmatchlines=cnt.nmatchlines, nhits=cnt.nhits, nrecords=cnt.nrecords,
                    nheaders=cnt.nheaders)

    def records(self, n=None, first=0):
        """Copy hit records to the host -> ndarray [n,4] u32 (line,start,end,dist)."""
        if n is None:
            raise ValueError("n required")
        out = np.zeros((n, 4), dtype=np.uint32)
        if n:
            _check(self._lib.seeqdevScanCopyRecords(self._h, out.ctypes.data, first, n))
        return out

    def record_offsets(self, n, first=0):
        """Per record: byte offset of its line in the scanned buffer -> ndarray [n] u64."""
        out = np.zeros(n, dtype=np.uint64)
        if n:
            _check(self._lib.seeqdevScanCopyOffsets(self._h, out.ctypes.data, first, n))
        return out

    def records_device_ptr(self):
        return self._lib.seeqdevScanRecordsDevice(self._h)

    def scan_tensor(self, pattern, t, options=0, want=WANT_COUNTLINES):
        """t: torch uint8 CUDA tensor (contiguous).  Runs and fetches."""
        self.run(pattern, t.data_ptr(), t.numel(), options, want)
        return self.fetch()

    def run_packed(self, pattern, bases_ptr, nmask_ptr, nreads, read_len, stride=None, nstride=None, options=0, want=WANT_COUNTLINES):
        """Enqueue the scan of a packed read batch resident in HBM (seeq_amd.h: seeqdev_packed_t); fetch() waits."""
        b = _capi.seeqdev_packed_t(bases_ptr, nmask_ptr or None, nreads, read_len, stride or (read_len + 3) // 4,
                                   nstride or (read_len + 7) // 8)
        _check(self._lib.seeqdevScanPacked(self._h, pattern.handle, C.byref(b), options, want))

    def scan_host(self, pattern, data, options=0, want=WANT_COUNTLINES):
        """data: bytes.  H2D + scan + fetch (+ records when want == WANT_RECORDS)."""
        cnt = _capi.seeqdev_counts_t()
        _check(self._lib.seeqdevScanHost(self._h, pattern.handle, data, len(data), options, want, C.byref(cnt)))
        res = dict(nlines=cnt.nlines, nmatchlines=cnt.nmatchlines, nhits=cnt.nhits, nrecords=cnt.nrecords,
                   nheaders=cnt.nheaders)
        if want == WANT_RECORDS:
            res["records"] = self.records(cnt.nrecords)
        return res


    # ---- several patterns, one text (include/seeq_amd.h: seeqdevScanRunMulti / seeqdevScanHostMulti) ----
    def _multi(self, patterns, call, want, copy=True):
        n = len(patterns)
        arr = (C.c_void_p * n)(*[C.cast(p.handle, C.c_void_p) for p in patterns])
        cnts = (_capi.seeqdev_counts_t * n)()
        _check(call(arr, n, C.cast(cnts, C.c_void_p)))
        out = []
        for k in range(n):
            c = cnts[k]
            res = dict(nlines=c.nlines, nmatchlines=c.nmatchlines, nhits=c.nhits, nrecords=c.nrecords, nheaders=c.nheaders)
            if want == WANT_RECORDS:
                ptr, m = C.c_void_p(), C.c_size_t()
                _check(self._lib.seeqdevScanMultiRecords(self._h, k, C.byref(ptr), C.byref(m)))
                if copy:
                    rec = np.zeros((m.value, 4), dtype=np.uint32)
                    if m.value:
                        C.memmove(rec.ctypes.data, ptr.value, m.value * 16)
                elif m.value:
                    # a view of the context's page-locked buffer: valid until the next multi scan of this Scanner
                    rec = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(m.value, 4))
                else:
                    rec = np.zeros((0, 4), dtype=np.uint32)
                res["records"] = rec
            out.append(res)
        return out

    def last_multi_one_pass(self):
        """True when the last multi-pattern scan walked the text once for all its patterns (seeq_multi.h)."""
        return bool(self._lib.seeqdevScanLastMulti(self._h))

    def scan_host_multi(self, patterns, data, options=0, want=WANT_COUNTLINES, copy=True):
        """data: bytes, staged ONCE.  -> one result dict per pattern (copy=False: records are views of the Scanner's buffer,
        valid until its next multi scan)."""
        return self._multi(patterns, lambda arr, n, cnts: self._lib.seeqdevScanHostMulti(self._h, arr, n, data, len(data), options, want, cnts), want, copy)

    def scan_tensor_multi(self, patterns, t, options=0, want=WANT_COUNTLINES, copy=True):
        """t: torch uint8 CUDA tensor (contiguous), resident.  -> one result dict per pattern (copy: see scan_host_multi)."""
        return self._multi(patterns, lambda arr, n, cnts: self._lib.seeqdevScanRunMulti(self._h, arr, n, C.c_void_p(t.data_ptr()), t.numel(),
                                                                                     options, want, cnts), want, copy)


def assign_best(results, nlines):
    """Demultiplexing rule on top of a multi-pattern SQ_BEST scan: per line the pattern with the smallest distance
    (ties: the first pattern in the list).  results: what scan_*_multi(..., SQ_BEST, WANT_RECORDS) returned.
    -> (which [nlines] int32, -1 = no pattern matched; dist [nlines] int32; start, end [nlines] int64)."""
    which = np.full(nlines, -1, dtype=np.int32)
    dist = np.full(nlines, np.iinfo(np.int32).max, dtype=np.int32)
    start = np.zeros(nlines, dtype=np.int64)
    end = np.zeros(nlines, dtype=np.int64)
    for k, r in enumerate(results):
        rec = r["records"]
        if not len(rec):
            continue
        ln = rec[:, 0].astype(np.int64) - 1
        better = rec[:, 3].astype(np.int32) < dist[ln]           # strict: an earlier pattern keeps a tie
        ln = ln[better]
        which[ln] = k
        dist[ln] = rec[better, 3]
        start[ln] = rec[better, 1]
        end[ln] = rec[better, 2]
    dist[which < 0] = -1
    return which, dist, start, end


def synth_reads(d_ptr, first, n, length, pattern_plain, tau, seed=0x5EE92025, stream=None):
    """Fill device memory with n synthetic reads (length bases + newline each)."""
    p = pattern_plain.encode() if isinstance(pattern_plain, str) else pattern_plain
    _check(_capi.lib().seeqdevSynthReads(C.c_void_p(d_ptr), first, n, length, p, len(p), tau, seed,
                                         C.c_void_p(stream) if stream else None))
