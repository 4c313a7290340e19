/*
 * tests/fake_seeqdev.c -- TEST DOUBLE of the device C-ABI (include/seeq_amd.h), for the host-side sanitizer builds.
 *
 * The product's host code (seeq_amd/csrc/libseeq_api.c, seeq_file.c, seeq_main.c: the libseeq.h / seeq.h entry points,
 * the ingest pipeline with its reader thread and lanes, the replay and the output formatter) is plain C that only talks
 * to the GPU through seeq_amd.h.  This file implements that interface on the CPU with the oracle (oracle/seeq_oracle.c)
 * so that the host code can run under -fsanitize=address,undefined and -fsanitize=thread in the CPU test suite, where
 * there is no GPU.  It is linked into test executables only (tests/test_host_sanitizers.py); the product library never
 * sees it -- without a HIP device the product fails loudly (tests/test_capi_host.py::test_no_gpu_fails_loudly).
 *
 * A scan "enqueued" with seeqdevScanHostBegin runs on a worker thread and is joined by seeqdevScanFetch, so the
 * caller's buffers are really read while the caller goes on -- an early reuse or free shows up under the sanitizers.
 * FAKE_SEEQ_DEVICES=n pretends to have n devices (seeqFileMatch's SEEQ_DEVICES spreading).
 *
 * FAULT INJECTION (round 5; the reference tests every allocation failure of its own, test/faultymalloc.c:20-56 -- here the failures
 * that matter are the device boundary's): FAKE_SEEQDEV_FAIL="begin@3,fetch@7,copy@2" makes the N-th call (1-based, counted over the
 * process) of the named entry point fail the way the HIP library does -- -1 / NULL, seeqerr = 0, errno set (EIO; ENOMEM for the
 * allocating ones):  new = seeqdevScanNew, pattern = seeqdevPatternNew, hostalloc = seeqdevHostAlloc, setdevice = seeqdevSetDevice,
 * begin = seeqdevScanHostBegin, fetch = seeqdevScanFetch (the job has run: the failure is the device's), copy = seeqdevScanCopyRecords,
 * offsets = seeqdevScanCopyOffsets, string = seeqdevStringMatch.  N may be a range "fetch@3-5" or open "fetch@3-" (every call from the third).
 */
#define _GNU_SOURCE
#include <errno.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "libseeq.h"
#include "seeq_amd.h"
#include "../oracle/seeq_oracle.h"

struct seeqdev_pattern { int device, wlen, tau; char *keys; };

struct seeqdev_scan {
   int device;
   int prof;
   pthread_t th;
   int running, ran;
   /* job */
   const struct seeqdev_pattern *pat;
   const char *text; size_t nbytes; int options, want;
   /* result */
   seeqdev_counts_t cnt;
   seeqdev_hit_t *rec; uint64_t *off; size_t cap;
   int failed;
   /* seeqdevStringMatch */
   seeqdev_hit_t *srec; size_t scap;
};

static __thread int t_device = 0;

enum { FP_NEW, FP_PATTERN, FP_HOSTALLOC, FP_SETDEVICE, FP_BEGIN, FP_FETCH, FP_COPY, FP_OFFSETS, FP_STRING, FP_COUNT };
static const char *const fp_name[FP_COUNT] = {"new", "pattern", "hostalloc", "setdevice", "begin", "fetch", "copy", "offsets", "string"};
static unsigned long fp_calls[FP_COUNT];

/* 1 when this call of entry point `fp` is to fail (errno set, seeqerr cleared) */
static int fake_fail(int fp, int err_no)
{
   const unsigned long n = __atomic_add_fetch(&fp_calls[fp], 1ul, __ATOMIC_RELAXED);
   const char *e = getenv("FAKE_SEEQDEV_FAIL");
   if (!e) return 0;
   const size_t ln = strlen(fp_name[fp]);
   while (*e) {
      if (!strncmp(e, fp_name[fp], ln) && e[ln] == '@') {
         char *end;
         unsigned long lo = strtoul(e + ln + 1, &end, 10), hi = lo;
         if (*end == '-') { hi = end[1] >= '0' && end[1] <= '9' ? strtoul(end + 1, &end, 10) : ~0ul; }
         if (n >= lo && n <= hi) { if (fp != FP_HOSTALLOC) seeqerr = 0; errno = err_no; return 1; }   /* (HostAlloc runs on the reader thread: errno only, as the HIP library's) */
      }
      while (*e && *e != ',') e++;
      if (*e == ',') e++;
   }
   return 0;
}

const char *seeqdevLastError(void) { return "fake device layer (tests)"; }

int seeqdevDeviceCount(void)
{
   const char *e = getenv("FAKE_SEEQ_DEVICES");
   const int n = e ? atoi(e) : 1;
   return n < 0 ? 0 : n;
}

int seeqdevSetDevice(int device)
{
   if (device < 0 || device >= seeqdevDeviceCount()) { seeqerr = 0; errno = ENODEV; return -1; }
   if (fake_fail(FP_SETDEVICE, ENODEV)) return -1;
   t_device = device;
   return 0;
}

seeqdev_pattern_t *seeqdevPatternNew(const char *keys, int wlen, int tau)
{
   seeqerr = 0;
   if (!keys || wlen < 1 || tau < 0 || tau >= wlen) { errno = EINVAL; return NULL; }
   if (wlen > SEEQDEV_MAX_WLEN) { errno = E2BIG; return NULL; }
   if (seeqdevDeviceCount() < 1) { errno = ENODEV; return NULL; }
   if (fake_fail(FP_PATTERN, ENOMEM)) return NULL;
   struct seeqdev_pattern *p = calloc(1, sizeof *p);
   if (!p) return NULL;
   p->device = t_device; p->wlen = wlen; p->tau = tau;
   p->keys = malloc((size_t)wlen);
   if (!p->keys) { free(p); return NULL; }
   memcpy(p->keys, keys, (size_t)wlen);
   return p;
}

void seeqdevPatternFree(seeqdev_pattern_t *p) { if (p) { free(p->keys); free(p); } }
int seeqdevPatternDevice(const seeqdev_pattern_t *p) { return p ? p->device : -1; }

seeqdev_scan_t *seeqdevScanNew(void *hip_stream)
{
   (void)hip_stream;
   seeqerr = 0;
   if (seeqdevDeviceCount() < 1) { errno = ENODEV; return NULL; }
   if (fake_fail(FP_NEW, ENOMEM)) return NULL;
   struct seeqdev_scan *s = calloc(1, sizeof *s);
   if (s) s->device = t_device;
   return s;
}

static void join_job(struct seeqdev_scan *s)
{
   if (s->running) { pthread_join(s->th, NULL); s->running = 0; }
}

void seeqdevScanFree(seeqdev_scan_t *s)
{
   if (!s) return;
   join_job(s);
   free(s->rec); free(s->off); free(s->srec);
   free(s);
}

int seeqdevScanSetProfiling(seeqdev_scan_t *s, int on) { if (!s) { errno = EINVAL; return -1; } s->prof = on; return 0; }
int seeqdevScanLastTimes(const seeqdev_scan_t *s, float ms[4]) { (void)s; ms[0] = ms[1] = ms[2] = ms[3] = 0.f; return 0; }
int seeqdevScanLastCopyMs(const seeqdev_scan_t *s, float *ms) { (void)s; *ms = 0.f; return 0; }

/* The whole scan, as the device would deliver it: counts, ordered records, per record the offset of its line. */
static void *job_main(void *arg)
{
   struct seeqdev_scan *s = arg;
   const struct seeqdev_pattern *p = s->pat;
   const int fasta = (s->options & SEEQDEV_FASTA) != 0;
   int opt = s->options & 0x0F;                          /* match mode + non-DNA handling (line mode) */
   if (s->want == SEEQDEV_WANT_COUNTLINES) opt = (opt & ~3) | ORC_FIRST;
   if (s->want == SEEQDEV_WANT_COUNTMATCH) opt = (opt & ~3) | ORC_ALL;
   uint64_t nlines = 0, nmatch = 0;
   size_t cap = 1024;
   uint64_t *quad = NULL;
   long n;
   for (;;) {
      uint64_t *g = realloc(quad, cap * 4 * sizeof *g);
      if (!g) { free(quad); s->failed = 1; return NULL; }
      quad = g;
      n = orc_buffer_scan(s->text, s->nbytes, p->keys, p->wlen, p->tau, opt, fasta, quad, cap, NULL, 0, &nlines, &nmatch);
      if (n < 0) { free(quad); s->failed = 1; return NULL; }
      if ((size_t)n <= cap) break;
      cap = (size_t)n + 16;
   }
   memset(&s->cnt, 0, sizeof s->cnt);
   s->cnt.nlines = nlines;
   s->cnt.nmatchlines = nmatch;
   s->cnt.nhits = s->want == SEEQDEV_WANT_COUNTLINES ? nmatch : (uint64_t)n;
   /* headers skipped (FASTA) */
   if (fasta) {
      size_t i = 0;
      while (i < s->nbytes) {
         if (s->text[i] == '>') s->cnt.nheaders++;
         const char *nl = memchr(s->text + i, '\n', s->nbytes - i);
         if (!nl) break;
         i = (size_t)(nl - s->text) + 1;
      }
   }
   if (s->want == SEEQDEV_WANT_RECORDS) {
      if ((size_t)n > s->cap) {
         free(s->rec); free(s->off);
         s->rec = malloc(((size_t)n + 1) * sizeof *s->rec);
         s->off = malloc(((size_t)n + 1) * sizeof *s->off);
         s->cap = (size_t)n;
         if (!s->rec || !s->off) { free(quad); s->failed = 1; return NULL; }
      }
      /* walk the counted lines once to find each record's line offset */
      size_t pos = 0, k = 0;
      uint64_t line = 0;
      while (pos < s->nbytes && k < (size_t)n) {
         const char *nl = memchr(s->text + pos, '\n', s->nbytes - pos);
         const size_t end = nl ? (size_t)(nl - s->text) : s->nbytes;
         if (!(fasta && s->text[pos] == '>')) {
            line++;
            while (k < (size_t)n && quad[4 * k] == line) {
               s->rec[k].line = (uint32_t)quad[4 * k]; s->rec[k].start = (uint32_t)quad[4 * k + 1];
               s->rec[k].end = (uint32_t)quad[4 * k + 2]; s->rec[k].dist = (uint32_t)quad[4 * k + 3];
               s->off[k] = pos;
               k++;
            }
         }
         pos = end + 1;
      }
      s->cnt.nrecords = (uint64_t)n;
   }
   free(quad);
   return NULL;
}

int seeqdevScanHostBegin(seeqdev_scan_t *s, const seeqdev_pattern_t *pat, const char *host_text, size_t nbytes, int options, int want)
{
   seeqerr = 0;
   if (!s || !pat || (!host_text && nbytes) || pat->device != s->device) { errno = EINVAL; return -1; }
   if (fake_fail(FP_BEGIN, EIO)) return -1;
   join_job(s);
   s->pat = pat; s->text = host_text; s->nbytes = nbytes; s->options = options; s->want = want;
   s->failed = 0; s->ran = 0;
   if (pthread_create(&s->th, NULL, job_main, s)) { errno = EAGAIN; return -1; }
   s->running = 1;
   return 0;
}

int seeqdevScanFetch(seeqdev_scan_t *s, seeqdev_counts_t *counts)
{
   seeqerr = 0;
   if (!s || (!s->running && !s->ran)) { errno = EINVAL; return -1; }
   join_job(s);
   s->ran = 1;
   if (fake_fail(FP_FETCH, EIO)) return -1;
   if (s->failed) { errno = ENOMEM; return -1; }
   if (counts) *counts = s->cnt;
   return 0;
}

int seeqdevScanHost(seeqdev_scan_t *s, const seeqdev_pattern_t *pat, const char *host_text, size_t nbytes, int options, int want,
                    seeqdev_counts_t *counts)
{
   if (seeqdevScanHostBegin(s, pat, host_text, nbytes, options, want)) return -1;
   return seeqdevScanFetch(s, counts);
}

int seeqdevScanCopyRecords(seeqdev_scan_t *s, seeqdev_hit_t *out, size_t first, size_t n)
{
   if (!s || first + n > s->cnt.nrecords) { errno = EINVAL; return -1; }
   if (fake_fail(FP_COPY, EIO)) return -1;
   if (n) memcpy(out, s->rec + first, n * sizeof *out);
   return 0;
}

int seeqdevScanCopyOffsets(seeqdev_scan_t *s, uint64_t *out, size_t first, size_t n)
{
   if (!s || first + n > s->cnt.nrecords) { errno = EINVAL; return -1; }
   if (fake_fail(FP_OFFSETS, EIO)) return -1;
   if (n) memcpy(out, s->off + first, n * sizeof *out);
   return 0;
}

void *seeqdevHostAlloc(size_t bytes) { if (fake_fail(FP_HOSTALLOC, ENOMEM)) return NULL; return malloc(bytes ? bytes : 1); }
void seeqdevHostFree(void *p) { free(p); }

int seeqdevStringMatch(seeqdev_scan_t *s, const seeqdev_pattern_t *pat, const char *data, size_t n, int options,
                       const seeqdev_hit_t **rec, size_t *nrec)
{
   seeqerr = 0;
   if (!s || !pat || !rec || !nrec) { errno = EINVAL; return -1; }
   if (fake_fail(FP_STRING, EIO)) return -1;
   char *z = malloc(n + 1);                               /* the oracle wants a NUL-terminated string */
   if (!z) return -1;
   memcpy(z, data, n); z[n] = 0;
   size_t cap = 64;
   orc_match_t *m = NULL;
   long k;
   for (;;) {
      orc_match_t *g = realloc(m, cap * sizeof *g);
      if (!g) { free(m); free(z); return -1; }
      m = g;
      k = orc_string_match(z, pat->keys, pat->wlen, pat->tau, options, m, cap);
      if (k < 0) { free(m); free(z); return -1; }
      if ((size_t)k <= cap) break;
      cap = (size_t)k;
   }
   if ((size_t)k > s->scap) {
      free(s->srec);
      s->srec = malloc(((size_t)k + 1) * sizeof *s->srec);
      s->scap = (size_t)k;
      if (!s->srec) { free(m); free(z); return -1; }
   }
   for (long i = 0; i < k; i++) {                         /* the oracle leaves them last hit first; the device left to right */
      const orc_match_t *q = &m[k - 1 - i];
      s->srec[i].line = 1; s->srec[i].start = (uint32_t)q->start; s->srec[i].end = (uint32_t)q->end; s->srec[i].dist = (uint32_t)q->dist;
   }
   *rec = s->srec; *nrec = (size_t)k;
   free(m); free(z);
   return 0;
}
