"""ctypes binding of the C-ABI library seeq_amd/lib/libseeq_amd.so.

The declarations mirror include/libseeq.h, include/seeq.h and
include/seeq_amd.h one to one.  There is no fallback: if the library is not
built, or it finds no GPU, the calls raise.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libseeq_amd.so")
CLI_PATH = os.path.join(HERE, "bin", "seeq")
CSRC = os.path.join(HERE, "csrc")

# libseeq.h option bits
SQ_FIRST, SQ_BEST, SQ_ALL, SQ_COUNT = 0x00, 0x01, 0x02, 0x03
SQ_FAIL, SQ_CONVERT, SQ_IGNORE = 0x00, 0x04, 0x08
SQ_LINES, SQ_STREAM = 0x00, 0x10
# seeq.h file options
SQ_ANY, SQ_MATCH, SQ_NOMATCH, SQ_COUNTLINES, SQ_COUNTMATCH = 0, 1, 2, 3, 4
# seeq_amd.h
WANT_COUNTLINES, WANT_COUNTMATCH, WANT_RECORDS = 0, 1, 2
SEEQDEV_FASTA, SEEQDEV_SINGLELINE = 0x100, 0x200


class match_t(C.Structure):
    _fields_ = [("start", C.c_size_t), ("end", C.c_size_t), ("dist", C.c_size_t)]


class seeq_t(C.Structure):
    _fields_ = [("hits", C.c_size_t), ("stacksize", C.c_size_t), ("match", C.POINTER(match_t)),
                ("bufsz", C.c_size_t), ("string", C.c_void_p), ("tau", C.c_int), ("wlen", C.c_int),
                ("keys", C.POINTER(C.c_char)), ("rkeys", C.POINTER(C.c_char)),
                ("dfa", C.c_void_p), ("rdfa", C.c_void_p)]


class seeqdev_packed_t(C.Structure):
    _fields_ = [("bases", C.c_void_p), ("nmask", C.c_void_p), ("nreads", C.c_uint64), ("read_len", C.c_uint32),
                ("stride", C.c_uint32), ("nstride", C.c_uint32)]


class seeqfile_t(C.Structure):
    _fields_ = [("flags", C.c_int), ("line", C.c_size_t), ("info", C.c_char_p), ("fdi", C.c_void_p)]


class seeqdev_hit_t(C.Structure):
    _fields_ = [("line", C.c_uint32), ("start", C.c_uint32), ("end", C.c_uint32), ("dist", C.c_uint32)]


class seeqdev_counts_t(C.Structure):
    _fields_ = [("nlines", C.c_uint64), ("nmatchlines", C.c_uint64), ("nhits", C.c_uint64),
                ("nrecords", C.c_uint64), ("nheaders", C.c_uint64)]


class seeqdev_textinfo_t(C.Structure):
    _fields_ = [("probe_ms", C.c_float * 12), ("nprobed", C.c_int), ("chosen", C.c_int), ("allocated_bytes", C.c_size_t),
                ("probe_peak_bytes", C.c_size_t)]


# Every symbol the three public headers declare (tests check the .so exports all of them).
EXPORTS = [
    # libseeq.h
    "seeqNew", "seeqFree", "seeqMatchIter", "seeqGetString", "seeqStringMatch", "seeqPrintError",
    "seeqAddMatch", "stackNew", "stackAddMatch", "recursive_merge", "seeqerr",
    # seeq.h
    "seeq", "seeqFileMatch", "seeqOpen", "seeqClose",
    # seeq_amd.h
    "seeqdevDeviceCount", "seeqdevSetDevice", "seeqdevLastError", "seeqdevPatternNew", "seeqdevPatternFree",
    "seeqdevPatternOf", "seeqdevScanNew", "seeqdevScanFree", "seeqdevScanReserve", "seeqdevScanRun",
    "seeqdevScanFetch", "seeqdevScanRecordsDevice", "seeqdevScanCopyRecords", "seeqdevScanHost",
    "seeqdevScanSetProfiling", "seeqdevScanLastTimes", "seeqdevScanLastLaunches", "seeqdevScanLastLaunchTimes", "seeqdevScanLastClockMHz", "seeqdevSynthReads",
    "seeqdevScanSetLineHint", "seeqdevScanLastPath", "seeqdevScanLastFilter", "seeqdevScanLastPackedQuad", "seeqdevScanCopyOffsets", "seeqdevHostAlloc", "seeqdevTextAlloc", "seeqdevTextAllocInfo", "seeqdevTextAllocFor", "seeqdevTextFree",
    "seeqdevHostFree", "seeqdevStringMatch", "seeqdevScanHostBegin", "seeqdevScanLastCopyMs", "seeqdevPatternDevice",
    "seeqdevScanRunMulti", "seeqdevScanHostMulti", "seeqdevScanMultiRecords", "seeqdevScanLastMulti", "seeqdevScanPacked", "seeqdevPackReads", "seeqdevPackReadsDevice",
]


def build(verbose=False):
    """Compile the library and the CLI in-tree (hipcc --offload-arch=gfx950)."""
    cmd = ["make", "-C", CSRC, "all"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)


_lib = None


def _share_torch_hip_runtime():
    """torch wheels bundle their own libamdhip64.so.7; /opt/rocm has another with the same
    soname.  One process must use ONE HIP runtime, so when torch is installed its copy is
    loaded first (without importing torch) and libseeq_amd.so binds to it; C callers such as
    seeq_amd/bin/seeq simply use /opt/rocm's.  SEEQ_AMD_SYSTEM_HIP=1 disables this."""
    if os.environ.get("SEEQ_AMD_SYSTEM_HIP") == "1":
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def lib():
    """Load libseeq_amd.so (raises OSError with a clear message if it is not built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("%s is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "or `make -C seeq_amd/csrc`; seeq_amd has no fallback matcher" % LIB_PATH)
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH, use_errno=True)      # (C.get_errno(): the errno a failed call left -- device failures are errno's, seeqerr = 0)
    P = C.POINTER
    L.seeqNew.argtypes = [C.c_char_p, C.c_int, C.c_size_t]
    L.seeqNew.restype = P(seeq_t)
    L.seeqFree.argtypes = [P(seeq_t)]
    L.seeqFree.restype = None
    L.seeqMatchIter.argtypes = [P(seeq_t)]
    L.seeqMatchIter.restype = P(match_t)
    L.seeqGetString.argtypes = [P(seeq_t)]
    L.seeqGetString.restype = C.c_char_p
    L.seeqStringMatch.argtypes = [C.c_char_p, P(seeq_t), C.c_int]
    L.seeqStringMatch.restype = C.c_long
    L.seeqPrintError.argtypes = []
    L.seeqPrintError.restype = C.c_char_p
    L.seeqAddMatch.argtypes = [P(seeq_t), match_t]
    L.seeqAddMatch.restype = C.c_int
    L.seeqOpen.argtypes = [C.c_char_p]
    L.seeqOpen.restype = P(seeqfile_t)
    L.seeqClose.argtypes = [P(seeqfile_t)]
    L.seeqClose.restype = C.c_int
    L.seeqFileMatch.argtypes = [P(seeqfile_t), P(seeq_t), C.c_int, C.c_int]
    L.seeqFileMatch.restype = C.c_long
    L.seeqdevDeviceCount.argtypes = []
    L.seeqdevDeviceCount.restype = C.c_int
    L.seeqdevSetDevice.argtypes = [C.c_int]
    L.seeqdevSetDevice.restype = C.c_int
    L.seeqdevLastError.argtypes = []
    L.seeqdevLastError.restype = C.c_char_p
    L.seeqdevPatternNew.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.seeqdevPatternNew.restype = C.c_void_p
    L.seeqdevPatternFree.argtypes = [C.c_void_p]
    L.seeqdevPatternFree.restype = None
    L.seeqdevPatternOf.argtypes = [P(seeq_t)]
    L.seeqdevPatternOf.restype = C.c_void_p
    L.seeqdevScanNew.argtypes = [C.c_void_p]
    L.seeqdevScanNew.restype = C.c_void_p
    L.seeqdevScanFree.argtypes = [C.c_void_p]
    L.seeqdevScanFree.restype = None
    L.seeqdevScanReserve.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]
    L.seeqdevScanReserve.restype = C.c_int
    L.seeqdevScanRun.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
    L.seeqdevScanRun.restype = C.c_int
    L.seeqdevScanFetch.argtypes = [C.c_void_p, P(seeqdev_counts_t)]
    L.seeqdevScanFetch.restype = C.c_int
    L.seeqdevScanRecordsDevice.argtypes = [C.c_void_p]
    L.seeqdevScanRecordsDevice.restype = C.c_void_p
    L.seeqdevScanCopyRecords.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
    L.seeqdevScanCopyRecords.restype = C.c_int
    L.seeqdevScanCopyOffsets.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
    L.seeqdevScanCopyOffsets.restype = C.c_int
    L.seeqdevScanHost.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_int,
                                  P(seeqdev_counts_t)]
    L.seeqdevScanHost.restype = C.c_int
    L.seeqdevScanSetProfiling.argtypes = [C.c_void_p, C.c_int]
    L.seeqdevScanSetProfiling.restype = C.c_int
    L.seeqdevScanLastTimes.argtypes = [C.c_void_p, P(C.c_float)]
    L.seeqdevScanLastTimes.restype = C.c_int
    L.seeqdevScanLastLaunches.argtypes = [C.c_void_p]
    L.seeqdevScanLastLaunches.restype = C.c_int
    L.seeqdevScanLastLaunchTimes.argtypes = [C.c_void_p, P(C.c_float), C.c_int]
    L.seeqdevScanLastLaunchTimes.restype = C.c_int
    L.seeqdevScanLastClockMHz.argtypes = [C.c_void_p]
    L.seeqdevScanLastClockMHz.restype = C.c_float
    L.seeqdevScanSetLineHint.argtypes = [C.c_void_p, C.c_double]
    L.seeqdevScanSetLineHint.restype = C.c_int
    L.seeqdevScanLastPath.argtypes = [C.c_void_p]
    L.seeqdevScanLastPath.restype = C.c_int
    L.seeqdevScanLastFilter.argtypes = [C.c_void_p]
    L.seeqdevScanLastFilter.restype = C.c_int
    L.seeqdevScanLastPackedQuad.argtypes = [C.c_void_p]
    L.seeqdevScanLastPackedQuad.restype = C.c_int
    L.seeqdevScanRunMulti.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    L.seeqdevScanRunMulti.restype = C.c_int
    L.seeqdevScanHostMulti.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    L.seeqdevScanHostMulti.restype = C.c_int
    L.seeqdevScanMultiRecords.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.seeqdevScanMultiRecords.restype = C.c_int
    L.seeqdevScanLastMulti.argtypes = [C.c_void_p]
    L.seeqdevScanLastMulti.restype = C.c_int
    L.seeqdevScanPacked.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(seeqdev_packed_t), C.c_int, C.c_int]
    L.seeqdevScanPacked.restype = C.c_int
    L.seeqdevPackReads.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
    L.seeqdevPackReads.restype = C.c_long
    L.seeqdevPackReadsDevice.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.seeqdevPackReadsDevice.restype = C.c_int
    L.seeqdevHostAlloc.argtypes = [C.c_size_t]
    L.seeqdevHostAlloc.restype = C.c_void_p
    L.seeqdevHostFree.argtypes = [C.c_void_p]
    L.seeqdevHostFree.restype = None
    L.seeqdevTextAlloc.argtypes = [C.c_size_t, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.seeqdevTextAlloc.restype = C.c_void_p
    L.seeqdevTextAllocInfo.argtypes = [C.c_size_t, C.c_int, C.POINTER(seeqdev_textinfo_t)]
    L.seeqdevTextAllocInfo.restype = C.c_void_p
    L.seeqdevTextAllocFor.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(seeqdev_textinfo_t)]
    L.seeqdevTextAllocFor.restype = C.c_void_p
    L.seeqdevTextFree.argtypes = [C.c_void_p]
    L.seeqdevTextFree.restype = None
    L.seeqdevSynthReads.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_char_p, C.c_int, C.c_int,
                                    C.c_uint64, C.c_void_p]
    L.seeqdevSynthReads.restype = C.c_int
    _lib = L
    return L


def seeqerr():
    return C.c_int.in_dll(lib(), "seeqerr").value


def error_text():
    L = lib()
    msg = L.seeqPrintError().decode(errors="replace")
    dev = L.seeqdevLastError().decode(errors="replace")
    return msg + (" [" + dev + "]" if dev and seeqerr() == 0 else "")
