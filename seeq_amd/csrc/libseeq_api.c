/*
 * libseeq_api.c -- the libseeq.h entry points of seeq-mi355x (host side, C).
 *
 * Same names, argument meaning and error behaviour as the reference's
 * src/libseeq.c:43-507, but every match is computed on the GPU through the
 * device C-ABI of include/seeq_amd.h.  There is no CPU matcher here: a
 * seeq_t cannot even be created without a HIP device (seeqNew -> NULL,
 * seeqerr = 0, errno = ENODEV).
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <errno.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "libseeq.h"
#include "seeq_amd.h"
#include "seeq_internal.h"
#include "seeq_pattern.h"

int seeqerr = 0;

/* Same messages, same indices as the reference (libseeq.c:28-41): callers
 * print them (reference seeq.c:79,86,106,179). */
static const char *const seeq_messages[12] = {
   "Check errno",
   "Illegal matching distance value",
   "Incorrect pattern (double opening brackets)",
   "Incorrect pattern (double closing brackets)",
   "Incorrect pattern (illegal character)",
   "Incorrect pattern (missing closing bracket)",
   "Illegal path value passed to 'trie_search'",
   "Illegal nodeid passed to 'trie_getrow' (node is not a leaf).\n",
   "Illegal path value passed to 'trie_insert'",
   "Pattern length must be larger than matching distance",
   "Passed seeq_t struct does not contain a valid file pointer",
   "End of line reached."};

static unsigned long g_engine_ids = 0;

/* reference libseeq.c:43-138 */
seeq_t *seeqNew(const char *pattern, int mismatches, size_t maxmemory)
{
   (void)maxmemory;   /* DFA memory cap (reference libseeq.c:919-932): there is no DFA cache to cap */
   if (mismatches < 0) {                       /* reference libseeq.c:69-72 */
      seeqerr = SEEQ_ERR_DIST;
      return NULL;
   }
   const size_t plen = strlen(pattern);
   char *keys = malloc(plen ? plen : 1);
   char *rkeys = malloc(plen ? plen : 1);
   seeq_engine_t *eng = calloc(1, sizeof *eng);
   seeq_t *sq = calloc(1, sizeof *sq);
   match_t *stack = malloc(INITIAL_MATCH_STACK_SIZE * sizeof(match_t));
   if (!keys || !rkeys || !eng || !sq || !stack) {
      seeqerr = 0;
      goto fail;
   }
   int err = 0;
   const int wlen = seeq_compile_pattern(pattern, keys, &err);
   if (wlen < 0) {                             /* reference libseeq.c:84-88 */
      seeqerr = err;
      goto fail;
   }
   if (mismatches >= wlen) {                   /* reference libseeq.c:92-96 */
      seeqerr = SEEQ_ERR_DIST_GE_LEN;
      goto fail;
   }
   for (int i = 0; i < wlen; i++) rkeys[i] = keys[wlen - 1 - i];   /* reference libseeq.c:89 */

   seeqerr = 0;
   eng->magic = SEEQ_ENGINE_MAGIC;
   eng->id = ++g_engine_ids;
   /* What the reference's `seeq -z` reads through sq->dfa / sq->rdfa
    * (reference seeq.c:184-189): word 0 = number of states, word 4 -> a size_t. */
   eng->compat_f[4] = (size_t)&eng->compat_zero;
   eng->compat_r[4] = (size_t)&eng->compat_zero;
   eng->pat = seeqdevPatternNew(keys, wlen, mismatches);
   if (!eng->pat) goto fail;                   /* seeqerr = 0, errno set (ENODEV, E2BIG, ENOMEM, EIO) */

   sq->hits = 0;
   sq->stacksize = INITIAL_MATCH_STACK_SIZE;
   sq->match = stack;
   sq->bufsz = 0;
   sq->string = NULL;
   sq->tau = mismatches;
   sq->wlen = wlen;
   sq->keys = keys;
   sq->rkeys = rkeys;
   sq->dfa = eng;                  /* forward half of the handle  */
   sq->rdfa = eng->compat_r;       /* reverse half (same object)  */
   return sq;

fail:
   free(keys); free(rkeys); free(eng); free(sq); free(stack);
   return NULL;
}

seeq_engine_t *seeq_engine_of(const seeq_t *sq)
{
   if (!sq || !sq->dfa) return NULL;
   seeq_engine_t *eng = (seeq_engine_t *)sq->dfa;
   return eng->magic == SEEQ_ENGINE_MAGIC ? eng : NULL;
}

seeqdev_pattern_t *seeqdevPatternOf(const seeq_t *sq)
{
   seeq_engine_t *eng = seeq_engine_of(sq);
   return eng ? eng->pat : NULL;
}

seeqdev_scan_t *seeq_engine_scan(seeq_engine_t *eng)
{
   if (!eng->scan) eng->scan = seeqdevScanNew(NULL);
   return eng->scan;
}

/* reference libseeq.c:140-168 */
void seeqFree(seeq_t *sq)
{
   if (!sq) return;
   seeq_engine_t *eng = seeq_engine_of(sq);
   if (eng) {
      seeq_file_forget_engine(eng->id);      /* read-ahead scans of an open file may still use the pattern */
      if (eng->scan) seeqdevScanFree(eng->scan);
      if (eng->pat) seeqdevPatternFree(eng->pat);
      eng->magic = 0;
      free(eng->rec);
      free(eng);
   }
   free(sq->string);
   free(sq->match);
   free(sq->keys);
   free(sq->rkeys);
   free(sq);
}

/* Append the records of one line to sq->match[] in the order the reference
 * leaves them: last hit first (reference libseeq.c:345-349). */
int seeq_store_hits(seeq_t *sq, const seeqdev_hit_t *rec, size_t n)
{
   sq->hits = 0;
   for (size_t k = n; k-- > 0;) {
      match_t m = {rec[k].start, rec[k].end, rec[k].dist};
      if (seeqAddMatch(sq, m)) return -1;
   }
   return 0;
}

/* reference libseeq.c:171-352: one string, on the GPU, in one launch (seeqdevStringMatch). */
#define SEEQ_LONG_STRING ((size_t)1 << 15)                  /* 32 KiB */

long seeqStringMatch(const char *data, seeq_t *sq, int options)
{
   seeqerr = 0;
   seeq_engine_t *eng = seeq_engine_of(sq);
   if (!eng || !data) { errno = EINVAL; return -1; }
   sq->hits = 0;                                               /* reference libseeq.c:237 */
   seeqdev_scan_t *scan = seeq_engine_scan(eng);
   if (!scan) return -1;
   const size_t n = strlen(data);                              /* reference libseeq.c:245 */
   /* A long string (a chromosome handed to seeqStringMatch / the Python module) goes through the batched line
      scan instead of the one-lane string kernel: in line mode a '\n' ends the string just as it does here
      (libseeq.c:267-270), so the string's hits are exactly the records of line 1. */
   const int as_lines = n >= SEEQ_LONG_STRING && !(options & MASK_INPUT) && sq->wlen <= 62;
   const int dev_opt = options & (MASK_MATCH | MASK_NONDNA | MASK_INPUT);
   if (!as_lines) {
      const seeqdev_hit_t *rec = NULL;
      size_t nrec = 0;
      if (seeqdevStringMatch(scan, eng->pat, data, n, dev_opt, &rec, &nrec)) return -1;
      if (seeq_store_hits(sq, rec, nrec)) return -1;
      return (long)sq->hits;                                   /* reference libseeq.c:351 */
   }
   seeqdev_counts_t cnt;
   if (seeqdevScanHost(scan, eng->pat, data, n, dev_opt, SEEQDEV_WANT_RECORDS, &cnt)) return -1;
   if (cnt.nrecords > eng->rec_cap) {
      seeqdev_hit_t *r = realloc(eng->rec, cnt.nrecords * sizeof *r);
      if (!r) { seeqerr = 0; return -1; }
      eng->rec = r;
      eng->rec_cap = cnt.nrecords;
   }
   if (seeqdevScanCopyRecords(scan, eng->rec, 0, cnt.nrecords)) return -1;
   uint64_t k = 0;                                             /* records are ordered by line: keep line 1 */
   while (k < cnt.nrecords && eng->rec[k].line == 1) k++;
   if (seeq_store_hits(sq, eng->rec, k)) return -1;
   return (long)sq->hits;                                      /* reference libseeq.c:351 */
}

/* reference libseeq.c:427-443 */
int seeqAddMatch(seeq_t *sq, match_t match)
{
   if (sq->hits >= sq->stacksize) {
      const size_t newsize = sq->stacksize > 0 ? 2 * sq->stacksize : 1;
      match_t *grown = realloc(sq->match, newsize * sizeof(match_t));
      if (!grown) return -1;
      sq->match = grown;
      sq->stacksize = newsize;
   }
   sq->match[sq->hits++] = match;
   return 0;
}

/* reference libseeq.c:446-465: pops from the end, i.e. left-to-right hits. */
match_t *seeqMatchIter(seeq_t *sq)
{
   if (sq->hits == 0) return NULL;
   sq->hits--;
   return &sq->match[sq->hits];
}

/* reference libseeq.c:467-486 */
char *seeqGetString(seeq_t *sq) { return sq->string; }

/* reference libseeq.c:488-507 */
const char *seeqPrintError(void)
{
   if (seeqerr <= 0 || seeqerr >= 12) return strerror(seeqerr > 0 ? seeqerr : errno);
   return seeq_messages[seeqerr];
}

/* ---- the match-stack utilities of the reference (libseeq.c:355-424).  The reference itself no longer calls them (its
 * only call site is commented out, libseeq.c:340), but libseeq.h declares them, so a caller of the drop-in may: they
 * work as the reference's do (tests/test_capi_host.py compares them with the reference's on random stacks). ---- */
mstack_t *stackNew(size_t size)
{
   if (size < 1) size = 1;
   mstack_t *st = malloc(sizeof(mstack_t) + size * sizeof(match_t));
   if (!st) return NULL;
   st->size = size;
   st->pos = 0;
   return st;
}

int stackAddMatch(mstack_t **stackp, match_t match)
{
   mstack_t *st = *stackp;
   if (st->pos >= st->size) {
      const size_t newsize = 2 * st->size;
      mstack_t *grown = realloc(st, sizeof(mstack_t) + newsize * sizeof(match_t));
      if (!grown) return -1;
      *stackp = st = grown;
      st->size = newsize;
   }
   st->match[st->pos++] = match;
   return 0;
}

/* One stack per distance (stackp[0 .. sq->tau], each holding matches in text order): moves into sq->match, right to left,
 * the matches inside [start, end) that do not overlap a match of a smaller distance -- level `tau` first, the gaps
 * between its matches handed to level tau + 1 (reference libseeq.c:355-390). */
int recursive_merge(size_t start, size_t end, int tau, seeq_t *sq, mstack_t **stackp)
{
   if (tau > sq->tau) return 0;
   mstack_t *st = stackp[tau];
   size_t hi = end;
   /* matches that reach beyond the interval are dropped.  (The reference reads match[pos-1] once more after its pos has
      reached 0 -- one element in front of the array; here the loop stops at the bottom.) */
   while (st->pos > 0 && st->match[st->pos - 1].end > end) st->pos--;
   while (st->pos > 0) {
      const match_t top = st->match[st->pos - 1];
      if (start > top.start) break;
      if (recursive_merge(top.end, hi, tau + 1, sq, stackp)) return -1;      /* the gap to the right of it */
      hi = top.start;
      st->pos--;
      if (seeqAddMatch(sq, top)) return -1;
   }
   return recursive_merge(start, hi, tau + 1, sq, stackp);
}
