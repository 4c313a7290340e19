/*
 * tests/host_driver.c -- drives the product's host code (libseeq_api.c + seeq_file.c over tests/fake_seeqdev.c) through
 * call sequences the CLI does not make, under the sanitizers: a pattern / option switch in the middle of a file with the
 * engine freed before the file is closed (reference usage seeq.c:293-392, seeq.c:195-196), every file option, a file that
 * is closed while read-ahead scans are in flight.  Prints "OK" or the first difference.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "libseeq.h"
#include "seeq.h"
#include "../oracle/seeq_oracle.h"

static int fail(const char *what, long a, long b) { printf("FAIL %s (%ld, %ld)\n", what, a, b); return 1; }

int main(int argc, char **argv)
{
   if (argc < 2) return 2;
   const char *path = argv[1];
   /* the file's lines */
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
   const char *pats[2] = {"GATGTAGCGCGATTAGCCTG", "GATTAGC"};
   const int taus[2] = {3, 1}, opts[2] = {SQ_ALL, SQ_BEST | SQ_CONVERT};
   seeq_t *sq[2];
   char keys[2][64]; int m[2];
   for (int i = 0; i < 2; i++) {
      sq[i] = seeqNew(pats[i], taus[i], 0);
      if (!sq[i]) return fail("seeqNew", i, seeqerr);
      int err; m[i] = orc_parse(pats[i], keys[i], &err);
   }
   /* 1. SQ_ANY line by line, switching pattern and options every 7 lines */
   seeqfile_t *f = seeqOpen(path);
   if (!f) return fail("seeqOpen", 0, seeqerr);
   size_t n = 0;
   for (;;) {
      const int w = (int)((n / 7) % 2);
      const long rv = seeqFileMatch(f, sq[w], opts[w], SQ_ANY);
      if (rv <= 0) break;
      n++;
      if (f->line != n) return fail("line number", (long)f->line, (long)n);
      if (strcmp(sq[w]->string, lines[n - 1])) return fail("sq->string", (long)n, 0);
      orc_match_t exp[256];
      const long ne = orc_string_match(lines[n - 1], keys[w], m[w], taus[w], opts[w], exp, 256);
      if ((long)sq[w]->hits != ne) return fail("hits", (long)sq[w]->hits, ne);
      for (long k = 0; k < ne; k++)
         if (sq[w]->match[k].start != exp[k].start || sq[w]->match[k].end != exp[k].end || sq[w]->match[k].dist != exp[k].dist)
            return fail("match", (long)n, k);
   }
   if (n != nlines) return fail("lines replayed", (long)n, (long)nlines);
   seeqFree(sq[0]);                                      /* the engine goes before the file, as in the reference's seeq() */
   seeqClose(f);
   /* 2. SQ_MATCH / SQ_NOMATCH / counts with the other pattern */
   long nmatch = 0, nhits_all = 0;
   for (size_t i = 0; i < nlines; i++) {
      orc_match_t e[256];
      const long k = orc_string_match(lines[i], keys[1], m[1], taus[1], SQ_ALL, e, 256);
      nmatch += k > 0; nhits_all += k;
   }
   f = seeqOpen(path);
   long got = 0, rv;
   while ((rv = seeqFileMatch(f, sq[1], SQ_FIRST, SQ_MATCH)) > 0) got++;
   if (got != nmatch) return fail("SQ_MATCH lines", got, nmatch);
   seeqClose(f);
   f = seeqOpen(path); got = 0;
   while ((rv = seeqFileMatch(f, sq[1], SQ_FIRST, SQ_NOMATCH)) > 0) got++;
   if (got != (long)nlines - nmatch) return fail("SQ_NOMATCH lines", got, (long)nlines - nmatch);
   seeqClose(f);
   f = seeqOpen(path);
   if ((rv = seeqFileMatch(f, sq[1], 0, SQ_COUNTLINES)) != nmatch) return fail("SQ_COUNTLINES", rv, nmatch);
   seeqClose(f);
   f = seeqOpen(path);
   if ((rv = seeqFileMatch(f, sq[1], 0, SQ_COUNTMATCH)) != nhits_all) return fail("SQ_COUNTMATCH", rv, nhits_all);
   seeqClose(f);
   /* 2b. the SAME engine, match option switched every 7 lines (SQ_BEST / SQ_ALL) while read-ahead scans made for the other
      option are in flight on every lane; then the file option switched (SQ_MATCH -> SQ_ANY -> SQ_COUNTLINES: `want` of the
      scans changes) in the middle of the file */
   f = seeqOpen(path);
   if (!f) return fail("seeqOpen", 1, seeqerr);
   n = 0;
   for (;;) {
      const int mo = (n / 7) % 2 ? SQ_ALL : SQ_BEST;
      const long rv2 = seeqFileMatch(f, sq[1], mo, SQ_ANY);
      if (rv2 <= 0) break;
      n++;
      if (f->line != n) return fail("2b line number", (long)f->line, (long)n);
      orc_match_t exp[256];
      const long ne = orc_string_match(lines[n - 1], keys[1], m[1], taus[1], mo, exp, 256);
      if ((long)sq[1]->hits != ne) return fail("2b hits", (long)sq[1]->hits, ne);
      for (long k = 0; k < ne; k++)
         if (sq[1]->match[k].start != exp[k].start || sq[1]->match[k].end != exp[k].end || sq[1]->match[k].dist != exp[k].dist)
            return fail("2b match", (long)n, k);
   }
   if (n != nlines) return fail("2b lines replayed", (long)n, (long)nlines);
   seeqClose(f);
   f = seeqOpen(path);
   {
      size_t at = 0;                                     /* lines consumed so far */
      long first_match = -1;
      for (size_t i = 0; i < nlines && first_match < 0; i++) {
         orc_match_t e[256];
         if (orc_string_match(lines[i], keys[1], m[1], taus[1], SQ_FIRST, e, 256) > 0) first_match = (long)i;
      }
      if (first_match >= 0) {
         if (seeqFileMatch(f, sq[1], SQ_FIRST, SQ_MATCH) != 1) return fail("2b SQ_MATCH", 0, 0);
         if ((long)f->line != first_match + 1) return fail("2b SQ_MATCH line", (long)f->line, first_match + 1);
         at = (size_t)first_match + 1;
      }
      for (int i = 0; i < 3 && at < nlines; i++, at++)
         if (seeqFileMatch(f, sq[1], SQ_ALL, SQ_ANY) != 1 || f->line != at + 1) return fail("2b SQ_ANY after SQ_MATCH", (long)f->line, (long)at + 1);
      long rest = 0;
      for (size_t i = at; i < nlines; i++) {
         orc_match_t e[256];
         rest += orc_string_match(lines[i], keys[1], m[1], taus[1], SQ_FIRST, e, 256) > 0;
      }
      if ((rv = seeqFileMatch(f, sq[1], 0, SQ_COUNTLINES)) != rest) return fail("2b SQ_COUNTLINES after SQ_ANY", rv, rest);
   }
   seeqClose(f);
   /* 3. close in the middle: read-ahead scans are in flight */
   f = seeqOpen(path);
   for (int i = 0; i < 5; i++) (void)seeqFileMatch(f, sq[1], SQ_ALL, SQ_ANY);
   seeqClose(f);
   seeqFree(sq[1]);
   for (size_t i = 0; i < nlines; i++) free(lines[i]);
   free(lines);
   printf("OK\n");
   return 0;
}
