/*
 * tests/host_fault_driver.c -- the product's host code (libseeq_api.c + seeq_file.c over tests/fake_seeqdev.c) with the DEVICE
 * BOUNDARY FAILING under it (FAKE_SEEQDEV_FAIL, see fake_seeqdev.c), under the sanitizers.  The reference exercises every
 * allocation failure of its own code (test/faultymalloc.c:20-56, test/testset.c:150-155,443-457,1229-1242) and promises
 * -1 / NULL with the cause in seeqerr or -- seeqerr = 0 -- in errno (src/libseeq.c:75-135,505); the device errors of this
 * build come out the same way (HIP errors -> -1, seeqerr = 0, errno = ENOMEM / EIO).  Checked here, for
 * whatever failure the environment injects:
 *
 *   - seeqNew: NULL or a usable engine;  seeqStringMatch: -1 with seeqerr = 0 and errno set, and the NEXT call is served;
 *   - seeqFileMatch (SQ_MATCH loop, as the CLI's, seeq.c:131): every line it returns before the failure is the oracle's next
 *     matching line with the oracle's hits (no wrong, skipped or repeated line), the failure is -1 with seeqerr = 0 and errno
 *     set, a call after the failure returns (no hang) -1 or 0 and delivers no line;
 *   - seeqClose / seeqFree in either order afterwards (argv[3] = "close-first" / "free-first"): no crash, no hang; the
 *     sanitizers watch for leaks, races and use after free.
 *
 * argv: file, mode (file | string | new), order.  Prints "OK lines=<n> rv=<last return> errno=<e>" or "FAIL ...".
 */
#define _GNU_SOURCE
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "libseeq.h"
#include "seeq.h"
#include "../oracle/seeq_oracle.h"

static int fail(const char *what, long a, long b) { printf("FAIL %s (%ld, %ld)\n", what, a, b); return 1; }

int main(int argc, char **argv)
{
   if (argc < 4) return 2;
   const char *path = argv[1], *mode = argv[2];
   const int close_first = !strcmp(argv[3], "close-first");
   const char *pat = "GATGTAGCGCGATTAGCCTG";
   const int tau = 3, opt = SQ_ALL;
   char keys[64];
   int err;
   const int m = orc_parse(pat, keys, &err);

   if (!strcmp(mode, "new")) {
      /* a failing seeqNew: NULL, no crash; the next one (the failure was the N-th call only) works */
      errno = 0;
      seeq_t *a = seeqNew(pat, tau, 0);
      const int e1 = errno, s1 = seeqerr;
      seeq_t *b = seeqNew(pat, tau, 0);
      if (!a && s1 != 0) return fail("seeqNew failed with seeqerr set", s1, e1);
      if (!a && e1 == 0) return fail("seeqNew failed without errno", s1, e1);
      if (b) {                                               /* (the engine's scan context is made by its first match: that call may be the one that fails) */
         long rv = seeqStringMatch("TTGATGTAGCGCGATTAGCCTGTT", b, SQ_BEST);
         if (rv < 0 && (seeqerr != 0 || errno == 0)) { seeqFree(b); if (a) seeqFree(a); return fail("failed match without errno / with seeqerr", seeqerr, errno); }
         if (rv < 0) rv = seeqStringMatch("TTGATGTAGCGCGATTAGCCTGTT", b, SQ_BEST);
         if (rv != 1) { seeqFree(b); if (a) seeqFree(a); return fail("engine after a failed call", rv, errno); }
      }
      if (a) seeqFree(a);
      if (b) seeqFree(b);
      printf("OK new first=%s second=%s errno=%d\n", a ? "ok" : "NULL", b ? "ok" : "NULL", e1);
      return 0;
   }

   seeq_t *sq = seeqNew(pat, tau, 0);
   if (!sq) { printf("OK seeqNew=NULL errno=%d\n", errno); return 0; }      /* (the injected failure hit seeqNew itself) */

   if (!strcmp(mode, "string")) {
      long served = 0, failed = 0;
      const char *texts[3] = {"TTGATGTAGCGCGATTAGCCTGTT", "ACGTACGTACGT", "GATGTAGCGCGATTAGCCTGGATGTAGCGCGATTAGCCTG"};
      for (int i = 0; i < 12; i++) {
         errno = 0;
         const long rv = seeqStringMatch(texts[i % 3], sq, opt);
         if (rv < 0) {
            if (seeqerr != 0 || errno == 0) return fail("seeqStringMatch failure without errno / with seeqerr", seeqerr, errno);
            failed++;
            continue;
         }
         orc_match_t exp[16];
         const long ne = orc_string_match(texts[i % 3], keys, m, tau, opt, exp, 16);
         if (rv != ne || (long)sq->hits != ne) return fail("hits after / before a failed call", rv, ne);
         for (long k = 0; k < ne; k++)
            if (sq->match[k].start != exp[k].start || sq->match[k].end != exp[k].end || sq->match[k].dist != exp[k].dist) return fail("match", i, k);
         served++;
      }
      seeqFree(sq);
      printf("OK string served=%ld failed=%ld\n", served, failed);
      return 0;
   }

   /* mode file: the CLI's loop */
   FILE *fp = fopen(path, "r");
   if (!fp) return 2;
   char **lines = NULL; size_t nlines = 0, cap = 0;
   char *buf = NULL; size_t bsz = 0; ssize_t r;
   while ((r = getline(&buf, &bsz, fp)) >= 0) {
      if (r && buf[r - 1] == '\n') buf[r - 1] = 0;
      if (nlines == cap) { cap = cap ? 2 * cap : 1024; lines = realloc(lines, cap * sizeof *lines); }
      lines[nlines++] = strdup(buf);
   }
   fclose(fp);
   free(buf);
   seeqfile_t *f = seeqOpen(path);
   if (!f) return fail("seeqOpen", 0, seeqerr);
   size_t cursor = 0;                                       /* lines of the file consumed so far (by the oracle's account) */
   long delivered = 0, rv;
   int e_fail = 0;
   for (;;) {
      errno = 0;
      rv = seeqFileMatch(f, sq, opt, SQ_MATCH);
      if (rv <= 0) { e_fail = errno; break; }
      /* the oracle's next matching line */
      orc_match_t exp[256];
      long ne = 0;
      while (cursor < nlines && (ne = orc_string_match(lines[cursor], keys, m, tau, opt, exp, 256)) == 0) cursor++;
      if (cursor >= nlines) return fail("a line beyond the oracle's last match", (long)f->line, (long)cursor);
      cursor++;
      if (f->line != cursor) return fail("line number", (long)f->line, (long)cursor);
      if (strcmp(sq->string, lines[cursor - 1])) return fail("sq->string", (long)cursor, 0);
      if ((long)sq->hits != ne) return fail("hits", (long)sq->hits, ne);
      for (long k = 0; k < ne; k++)
         if (sq->match[k].start != exp[k].start || sq->match[k].end != exp[k].end || sq->match[k].dist != exp[k].dist) return fail("match", (long)cursor, k);
      delivered++;
   }
   if (rv == -1) {
      if (seeqerr != 0) return fail("seeqFileMatch failed with seeqerr set (a device error is an errno)", seeqerr, e_fail);
      if (e_fail == 0) return fail("seeqFileMatch failed without errno", 0, 0);
      /* after the failure: no hang, no line out of nowhere */
      const long again = seeqFileMatch(f, sq, opt, SQ_MATCH);
      if (again > 0) return fail("a line after the failure", again, (long)f->line);
   } else {
      /* no failure reached this run (the N-th call never happened): the whole file must have been served */
      long ne = 0;
      orc_match_t exp[256];
      while (cursor < nlines && (ne = orc_string_match(lines[cursor], keys, m, tau, opt, exp, 256)) == 0) cursor++;
      if (cursor < nlines) return fail("end of input before the oracle's", (long)cursor, (long)nlines);
   }
   if (close_first) { if (seeqClose(f)) return fail("seeqClose", seeqerr, 0); seeqFree(sq); }
   else { seeqFree(sq); if (seeqClose(f)) return fail("seeqClose", seeqerr, 0); }
   for (size_t i = 0; i < nlines; i++) free(lines[i]);
   free(lines);
   printf("OK file lines=%ld rv=%ld errno=%d\n", delivered, rv, e_fail);
   return 0;
}
