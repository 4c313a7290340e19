/*
 * include/seeq_amd.h -- batched, device-level C-ABI of seeq-mi355x.
 *
 * This is the boundary the HIP kernels sit behind.  Plain pointers and sizes
 * only (no torch / C++ types).  What it replaces in the reference is the
 * per-file hot loop: one `getline` + one `seeqStringMatch` per line
 * (reference seeq.c:361-387 calling libseeq.c:171-352).  One `seeqdevScanRun`
 * call performs that whole loop for every line of a text buffer that is
 * already resident in HBM, and leaves counts + ordered hit records in HBM.
 *
 * The libseeq.h / seeq.h entry points (seeqStringMatch, seeqFileMatch) are
 * implemented on top of these calls; bench.py and the Python module call
 * them directly through ctypes with device pointers.
 *
 * Error convention (same as libseeq.h): functions return NULL / -1, set
 * `seeqerr = 0` and `errno` (ENODEV: no usable GPU, ENOMEM: device or host
 * allocation failed, EIO: a HIP call failed, EINVAL/E2BIG: bad arguments);
 * seeqdevLastError() returns the HIP error text.
 */
#ifndef SEEQ_AMD_H_
#define SEEQ_AMD_H_

#include <stddef.h>
#include <stdint.h>

#include "libseeq.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SEEQ_AMD_VERSION "seeq-mi355x-0.1"

/* Longest pattern (positions) the kernels are instantiated for. */
#define SEEQDEV_MAX_WLEN 512

/* What a scan must produce (`want`): mirrors the file options of reference
 * seeq.h:64-68.  Counts are always produced; RECORDS adds hit records. */
#define SEEQDEV_WANT_COUNTLINES 0   /* SQ_COUNTLINES: #lines with >=1 hit (seeq.c:349 forces FIRST)   */
#define SEEQDEV_WANT_COUNTMATCH 1   /* SQ_COUNTMATCH: total hits under SQ_ALL (seeq.c:348)             */
#define SEEQDEV_WANT_RECORDS    2   /* (line,start,end,dist) per hit under match_opt FIRST/BEST/ALL    */

/* Extra scan flags, OR-ed into `options` above the libseeq.h option bits. */
#define SEEQDEV_FASTA       0x100   /* lines starting with '>' are headers: skipped, not counted (seeq.c:367-374) */
#define SEEQDEV_SINGLELINE  0x200   /* the buffer is ONE string (seeqStringMatch semantics, libseeq.c:171): no
                                       newline index; with SQ_STREAM newlines are skipped (libseeq.c:265)  */

/* One hit.  `line` is the 1-based index among counted lines of the scanned
 * buffer (reference seeq.c:377); `end` is exclusive (libseeq.h:62-66).
 * 16 bytes: the "algorithmic bytes per hit" of SURVEY section 8d. */
typedef struct {
   uint32_t line;
   uint32_t start;
   uint32_t end;
   uint32_t dist;
} seeqdev_hit_t;

typedef struct {
   uint64_t nlines;       /* counted lines (FASTA headers excluded)            */
   uint64_t nmatchlines;  /* lines with at least one hit                       */
   uint64_t nhits;        /* hits under the requested match mode               */
   uint64_t nrecords;     /* records stored (== nhits for WANT_RECORDS else 0) */
   uint64_t nheaders;     /* FASTA header lines skipped                        */
} seeqdev_counts_t;

typedef struct seeqdev_pattern seeqdev_pattern_t;   /* pattern tables in HBM */
typedef struct seeqdev_scan    seeqdev_scan_t;      /* stream + workspace    */

/* Number of usable HIP devices (0 if none / no runtime).  Never fails. */
int seeqdevDeviceCount(void);
/* Select the device used by subsequent seeqdevPatternNew / seeqdevScanNew calls of this thread (hipSetDevice).
 * Patterns and scan contexts remember the device they were created on; every call on them runs there. */
int seeqdevSetDevice(int device);
const char * seeqdevLastError(void);

/* Upload the compiled pattern.  `keys` is the output of the pattern
 * compiler (one byte per position, reference libseeq.c:517-543), wlen <=
 * SEEQDEV_MAX_WLEN, 0 <= tau < wlen (reference libseeq.c:69-72,92-96). */
seeqdev_pattern_t * seeqdevPatternNew(const char * keys, int wlen, int tau);
void                seeqdevPatternFree(seeqdev_pattern_t * pat);
/* The device pattern behind a seeq_t made by seeqNew() (sq->dfa). */
seeqdev_pattern_t * seeqdevPatternOf(const seeq_t * sq);
/* The HIP device a pattern lives on (the device that was current in seeqdevPatternNew); -1 for NULL.  A pattern and
 * the scan contexts that use it must live on the same device: a caller that spreads chunks over several GPUs
 * (seeqFileMatch with SEEQ_DEVICES) makes one pattern per device. */
int seeqdevPatternDevice(const seeqdev_pattern_t * pat);

/* A scan context owns its workspace in HBM and runs on `hip_stream`
 * (a hipStream_t passed as void*; NULL => a private stream). */
seeqdev_scan_t * seeqdevScanNew(void * hip_stream);
void             seeqdevScanFree(seeqdev_scan_t * scan);
/* Pre-size the workspace so that seeqdevScanRun never allocates (so it can
 * sit inside a timed region / graph).  Any argument may be 0 = keep. */
int seeqdevScanReserve(seeqdev_scan_t * scan, size_t max_bytes, size_t max_lines, size_t max_hitlines,
                       size_t max_records);

/* Optional: average bytes per line (newline included) of the buffers to come; sizes the text
 * tiles of the fused kernel.  0 (default) = estimate it from a 64 KiB sample of each new buffer. */
int seeqdevScanSetLineHint(seeqdev_scan_t * scan, double avg_bytes_per_line);
/* Which device path served the last run: 1 = generic (newline index + k_forward<W>),
 * 3 = one-pass per-line bit-vector kernel, text in registers (k_direct),
 * 5 = one-pass table-driven line-agnostic kernel (k_stream: every lane walks a fixed chunk of the text
 *     through the pattern's Levenshtein automaton held in LDS),
 * 6 = the same walk, two text bytes per table step, over the pattern's pair automaton (k_pair: candidates, verified),
 * 7 = the same frame with the bit-vector column instead of a table (k_stream's Myers mode: patterns without an automaton on
 *     long lines), 8 = a packed read batch (seeqdevScanPacked).
 * The one-pass kernels serve patterns <= 62 positions in ONE pass over the text. */
int seeqdevScanLastPath(const seeqdev_scan_t * scan);
/* 1 when the last run's k_stream walked a partition FILTER automaton (candidates verified by the exact pass)
 * instead of the pattern's complete automaton. */
int seeqdevScanLastFilter(const seeqdev_scan_t * scan);
/* 1 when the last packed run (path 8) walked the pattern's QUAD table -- four bases, one packed byte, per table step over a small
 * partition-filter automaton (seeq_dfa.h section 3b) -- instead of the pair table (two bases per step). */
int seeqdevScanLastPackedQuad(const seeqdev_scan_t * scan);

/* Enqueue (asynchronously, on the context's stream) the whole hot path over
 * d_text[0..nbytes): newline index -> per-line forward scan -> hit-line
 * compaction -> exact pass (acceptance rules + reverse start recovery) ->
 * ordered records.  `options` = libseeq.h match/non-DNA/input bits |
 * SEEQDEV_* flags.  d_text must stay valid until seeqdevScanFetch returns. */
int seeqdevScanRun(seeqdev_scan_t * scan, const seeqdev_pattern_t * pat, const void * d_text, size_t nbytes,
                   int options, int want);

/* Wait for the scan and return its counts.  If the workspace was too small
 * the scan is transparently re-run with a larger one (never happens after a
 * sufficient seeqdevScanReserve). */
int seeqdevScanFetch(seeqdev_scan_t * scan, seeqdev_counts_t * counts);

/* Hit records of the last fetched scan: device pointer / copy to host. */
const seeqdev_hit_t * seeqdevScanRecordsDevice(const seeqdev_scan_t * scan);
int seeqdevScanCopyRecords(seeqdev_scan_t * scan, seeqdev_hit_t * host_out, size_t first, size_t n);
/* Per record, the byte offset (within the scanned buffer) of the first byte of its line: lets a caller that
 * holds the text go from hit to hit without walking the lines in between. */
int seeqdevScanCopyOffsets(seeqdev_scan_t * scan, uint64_t * host_out, size_t first, size_t n);

/* Page-locked host memory (hipHostMalloc / hipHostFree) for buffers handed to seeqdevScanHost.  May be called from any thread (seeqFileMatch's
 * reader thread does): NULL + errno = ENOMEM on failure, and -- alone among these entries -- seeqerr is left untouched (it is the reference's plain
 * global, libseeq.h:38, and belongs to the thread that calls the seeq API). */
void * seeqdevHostAlloc(size_t bytes);
void   seeqdevHostFree(void * p);

/* Device memory for RESIDENT text, chosen by measurement: the scan kernel's time per 3.75 GiB follows the physical pages a buffer gets
 * from the driver (0.77 / 0.87 / 0.92 ms for the same text, stable for the life of the allocation), so a caller that
 * keeps text buffers resident picks each once.  Up to `candidates` (<= 12) allocations -- the plain one, then power-of-two blocks, which are
 * fast far more often -- are filled with synthetic reads and scanned; the fastest is returned, the others freed.  candidates < 2 (or a
 * buffer under 64 MiB): a plain allocation.  probe_ms (room for 12 floats) / nprobed (may be NULL): the candidates' scan-kernel times.
 * Contents undefined.
 * NULL + errno on failure.  (The reference has no device memory: an addition of this boundary, like seeqdevHostAlloc.) */
void * seeqdevTextAlloc(size_t bytes, int candidates, float * probe_ms, int * nprobed);
/* The same, telling the caller what it got: the block returned may be LARGER than `bytes` (a power-of-two block: up to 2 x bytes stay
 * allocated until seeqdevTextFree), and while the candidates are probed up to candidates x 2 x bytes + the probing scan's workspace
 * (about half a segment's bytes + 256 MiB) are held on the device -- probe_peak_bytes; other allocations of the caller may fail meanwhile. */
typedef struct seeqdev_textinfo {
   float  probe_ms[12];        /* the candidates' scan-kernel times (nprobed of them; candidate 0 = the plain allocation) */
   int    nprobed;             /* 0: nothing was probed (one candidate, a small buffer, or the probe failed: the plain allocation) */
   int    chosen;              /* index of the candidate returned */
   size_t allocated_bytes;     /* size of the allocation behind the returned pointer (>= bytes) */
   size_t probe_peak_bytes;    /* device memory held at the peak of the call */
} seeqdev_textinfo_t;
void * seeqdevTextAllocInfo(size_t bytes, int candidates, seeqdev_textinfo_t * info);
/* The same with the candidates probed by the CALLER's scan context.  The scan kernel's time is a property of the pair (text buffer, scan context's
 * workspace): one text runs at 0.72 or at 0.84 ms per 3.75 GiB with two contexts of one process, reproducibly (profiles/r05/workspace_probe.txt) --
 * so the candidate that is fastest with a context made for the probe need not be the fastest with the context that will scan the text.  `scan`: the
 * context that will (seeqdevScanReserve it first, so that its workspace is the one that stays); NULL: as seeqdevTextAllocInfo.  The context's
 * last results are those of the probe's last scan. */
void * seeqdevTextAllocFor(seeqdev_scan_t * scan, size_t bytes, int candidates, seeqdev_textinfo_t * info);
void   seeqdevTextFree(void * d_text);

/* Convenience: host buffer in, counts (+ records) out.  Stages through the
 * context's pinned buffer, H2D, ScanRun, ScanFetch. */
int seeqdevScanHost(seeqdev_scan_t * scan, const seeqdev_pattern_t * pat, const char * host_text, size_t nbytes,
                    int options, int want, seeqdev_counts_t * counts);

/* One string in ONE kernel launch -- what seeqStringMatch (reference libseeq.c:171-352, called once per string by
 * seeqmodule.c:858) runs on: data[0..n) is staged (short strings are read by the kernel straight from page-locked
 * host memory), scanned with `options` (libseeq.h match / non-DNA / input bits; SINGLELINE semantics) and the hits
 * land in page-locked host memory: *rec (left to right, valid until the next call on this context), *nrec. */
int seeqdevStringMatch(seeqdev_scan_t * scan, const seeqdev_pattern_t * pat, const char * data, size_t n, int options,
                       const seeqdev_hit_t ** rec, size_t * nrec);

/* The asynchronous half of seeqdevScanHost: stage (H2D on the context's stream) and enqueue the scan, return at once;
 * seeqdevScanFetch() then waits for it.  host_text must stay valid and unchanged until the fetch returns.  A reader
 * that fills the next chunk meanwhile, and one context per GPU, is how seeqFileMatch pipelines its ingest. */
int seeqdevScanHostBegin(seeqdev_scan_t * scan, const seeqdev_pattern_t * pat, const char * host_text, size_t nbytes,
                         int options, int want);
/* SEVERAL PATTERNS, ONE TEXT (barcode demultiplexing; the reference names the multi-pattern search as the place where its
 * algorithm has parallel work, doc/response.tex:358-360, and runs one pattern per scan, seeq.c:307-437).  The text is staged
 * once (seeqdevScanHostMulti) or is already resident (seeqdevScanRunMulti).  A set of 2 .. 32 patterns that has a union
 * automaton (seeq_multi.h: barcodes of 8 .. 12 positions at distance <= 1 do, sixteen at a time) is scanned in ONE walk over
 * the text -- read-length lines, SQ_FAIL or SQ_CONVERT -- that finds the candidate (line, pattern) pairs; the exact pass
 * verifies each pair.  Every other set / text / option, and any set under SEEQ_MULTI=sequential, gets a scan per pattern over
 * the resident text, back to back on the context's stream (seeqdevScanLastMulti tells which it was).  Synchronous.
 * counts[k] (may be NULL) = pattern k's counts; with SEEQDEV_WANT_RECORDS pattern k's ordered records are kept on the host
 * (seeqdevScanMultiRecords: context-owned, valid until the next multi scan).  All patterns must live on the context's
 * device.  Results per pattern are those of a seeqdevScanRun of its own -- whichever way the set was scanned. */
int seeqdevScanRunMulti(seeqdev_scan_t * scan, const seeqdev_pattern_t * const * pats, int npat, const void * d_text, size_t nbytes,
                        int options, int want, seeqdev_counts_t * counts);
int seeqdevScanHostMulti(seeqdev_scan_t * scan, const seeqdev_pattern_t * const * pats, int npat, const char * host_text, size_t nbytes,
                         int options, int want, seeqdev_counts_t * counts);
int seeqdevScanMultiRecords(const seeqdev_scan_t * scan, int k, const seeqdev_hit_t ** rec, size_t * nrec);
int seeqdevScanLastMulti(const seeqdev_scan_t * scan);     /* 1: one walk for all patterns; 0: a scan per pattern */

/* PACKED READ BATCHES -- 2 bits per base instead of a byte: a quarter of the HBM (and PCIe) traffic of the ASCII scan for
 * read sets that are kept packed anyway (BAM, .2bit, a sequencer's own format).  Layout, all device pointers:
 *   bases : four bases per byte, the FIRST base of a byte in its bits 7-6; code = (ASCII >> 1) & 3, i.e. A 0, C 1, T/U 2, G 3;
 *           read r starts at bases + r * stride (stride >= ceil(read_len / 4); padding bits are ignored);
 *   nmask : optional (NULL: no N anywhere): one bit per base, first base of a byte in bit 7, set where the base is N (the
 *           2-bit code of such a base is ignored); read r at nmask + r * nstride (nstride >= ceil(read_len / 8));
 *   every read has read_len bases (1 .. 256).
 * The result is that of seeqdevScanRun over the same reads as ASCII text, one read per line ('line' of a record = read
 * index + 1), for every match option and every `want` -- the scan kernel walks the packed bytes (one read per lane, no warm-
 * up, seeq_packed.h), candidate reads are unpacked and verified by the exact pass.  Asynchronous: seeqdevScanFetch waits.
 * Any pattern: those of more than 62 positions, or without a pair automaton, are served by unpacking the batch on the device
 * (nreads * (read_len + 1) bytes of scratch) and scanning that text.  seeqdevScanCopyOffsets after a packed scan reports, per
 * record, the offset its read has in the ASCII form of the batch: (line - 1) * (read_len + 1).
 * What replaces what: the reference has no packed input; this is the boundary's batch entry for callers that do. */
typedef struct {
   const void * bases;
   const void * nmask;
   uint64_t     nreads;
   uint32_t     read_len;
   uint32_t     stride;
   uint32_t     nstride;
} seeqdev_packed_t;
int  seeqdevScanPacked(seeqdev_scan_t * scan, const seeqdev_pattern_t * pat, const seeqdev_packed_t * batch, int options, int want);
/* Host helper: ASCII reads (one per line, each exactly read_len bases of A C G T U N in either case) -> that layout, into
 * caller-provided host buffers (nmask_out may be NULL when the text holds no N: an N is then an error).  Returns the number
 * of reads, or -1 (errno EINVAL: another byte, a line of another length). */
long seeqdevPackReads(const char * text, size_t nbytes, uint32_t read_len, void * bases_out, void * nmask_out, uint32_t stride, uint32_t nstride);
/* The same device to device: d_text holds nreads lines of read_len bases + '\n' in HBM (a byte that is no base counts as N with
 * a mask, as A without).  Asynchronous on hip_stream (a hipStream_t; NULL = the null stream). */
int  seeqdevPackReadsDevice(const void * d_text, uint64_t nreads, uint32_t read_len, void * d_bases, void * d_nmask, uint32_t stride, uint32_t nstride,
                            void * hip_stream);

/* Time (ms) of that H2D copy for the last fetched scan (profiling on), from HIP events on the context's stream. */
int seeqdevScanLastCopyMs(const seeqdev_scan_t * scan, float * h2d_ms);

/* Device time (ms) of the last fetched scan, measured with HIP events recorded
 * on the scan's stream around each phase of each segment (no extra
 * synchronisation): [0] newline index, [1] forward scan = the k_forward
 * launches (the dominant kernel), [2] compaction + exact pass + records,
 * [3] total.  Sums over the segments; seeqdevScanLastLaunches() returns how
 * many k_forward launches [1] covers.  Only filled when profiling was enabled
 * with seeqdevScanSetProfiling(1) before the run. */
int seeqdevScanSetProfiling(seeqdev_scan_t * scan, int on);
int seeqdevScanLastTimes(const seeqdev_scan_t * scan, float ms[4]);
int seeqdevScanLastLaunches(const seeqdev_scan_t * scan);
/* The same per launch: the duration (ms) of each of the last run's forward-scan
 * launches, in launch order, up to `cap` of them; returns how many there were. */
int seeqdevScanLastLaunchTimes(const seeqdev_scan_t * scan, float * ms, int cap);
/* The core clock (MHz) the last run's scan launches actually ran at: the kernel's first wave reads the shader clock
 * and the constant 100 MHz counter before and after its tiles (k_pair, profiling on); 0 when not measured. */
float seeqdevScanLastClockMHz(const seeqdev_scan_t * scan);

/* Synthetic shape-R reads written straight into HBM (bench/test input; spec
 * in SURVEY.md section 8d, CPU twin in oracle/seeq_oracle.c): n lines of
 * `len` bases + '\n' for read indices [first, first+n). */
int seeqdevSynthReads(void * d_out, uint64_t first, uint64_t n, int len, const char * pattern_plain, int plen,
                      int tau, uint64_t seed, void * hip_stream);

#ifdef __cplusplus
}
#endif
#endif
