/*
 * seeq_file.c -- file-level entry points of seeq-mi355x (host side, C):
 * seeqOpen / seeqClose / seeqFileMatch / seeq.
 *
 * seeqFileMatch keeps the reference's contract (src/seeq.c:293-392): each call
 * continues where the previous one stopped, returns after the line the file
 * option asks for, and leaves that line in sq->string, its hits in sq->match
 * and its number in sqfile->line.  What changes is how the hits are found:
 * the file is read ahead in large chunks, each chunk is scanned ONCE on the GPU
 * (include/seeq_amd.h: newline index, one line per lane, ordered hit records)
 * and the calls replay the records.  No matching happens on the host.
 *
 * Divergence from the reference, by construction of the read-ahead: the FILE*
 * position runs ahead of the line last returned.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <pthread.h>
#include <sys/stat.h>

#include "seeq.h"
#include "seeq_amd.h"
#include "seeq_internal.h"

/* ------------------------------------------------------------------------ */
/* Read-ahead state, one per open seeqfile_t (kept beside the public struct  */
/* so that its layout, reference seeq.h:55-60, is untouched).               */
/* ------------------------------------------------------------------------ */
typedef struct fstate_t {
   struct fstate_t *next;
   seeqfile_t      *key;
   char            *buf;       /* chunk */
   size_t           cap, len;  /* capacity, bytes read */
   size_t           avail;     /* bytes that belong to complete lines (all of len at EOF) */
   size_t           pos;       /* next unread byte */
   int              eof;
   /* cached GPU scan of buf[scan_from, avail) */
   int              have;
   unsigned long    eng_id;
   int              opt_key;
   size_t           counted;   /* counted lines replayed since scan_from */
   seeqdev_hit_t   *rec;
   uint64_t        *rec_off;   /* per record: offset of its line, relative to scan_from */
   size_t           rec_cap, nrec, rec_pos;
   size_t           scan_from; /* where in buf the cached scan started */
   size_t           scan_lines;/* counted lines in the cached scan */
} fstate_t;

static fstate_t *g_states = NULL;

static size_t chunk_bytes(void)
{
   const char *env = getenv("SEEQ_CHUNK_BYTES");
   if (env) {
      long long v = atoll(env);
      if (v >= 64) return (size_t)v;
   }
   return (size_t)64 << 20;
}

static fstate_t *state_of(seeqfile_t *f, int create)
{
   for (fstate_t *s = g_states; s; s = s->next)
      if (s->key == f) return s;
   if (!create) return NULL;
   fstate_t *s = calloc(1, sizeof *s);
   if (!s) return NULL;
   s->key = f;
   s->next = g_states;
   g_states = s;
   return s;
}

static void state_drop(seeqfile_t *f)
{
   for (fstate_t **pp = &g_states; *pp; pp = &(*pp)->next) {
      if ((*pp)->key == f) {
         fstate_t *s = *pp;
         *pp = s->next;
         seeqdevHostFree(s->buf);
         free(s->rec);
         free(s->rec_off);
         free(s);
         return;
      }
   }
}

/* fread() for big reads of a regular file: the range is split over a few threads that pread() it straight into the
 * (page-locked) chunk buffer -- one thread copies out of the page cache at ~5-10 GB/s, four come close to the link
 * speed the GPU side can take.  Pipes, small reads and anything unusual go through fread(). */
typedef struct { int fd; char *dst; size_t n; off_t off; size_t got; } rd_job_t;

static void *rd_worker(void *p)
{
   rd_job_t *j = p;
   size_t g = 0;
   while (g < j->n) {
      const ssize_t r = pread(j->fd, j->dst + g, j->n - g, j->off + (off_t)g);
      if (r < 0 && errno == EINTR) continue;
      if (r <= 0) break;
      g += (size_t)r;
   }
   j->got = g;
   return NULL;
}

static size_t chunk_read(FILE *fdi, char *dst, size_t want)
{
   enum { NT = 4 };
   const size_t big = (size_t)8 << 20;
   struct stat st;
   const int fd = fileno(fdi);
   off_t off;
   if (want >= big && fd >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && (off = ftello(fdi)) >= 0 &&
       st.st_size > off && (size_t)(st.st_size - off) >= big) {
      const size_t n = (size_t)(st.st_size - off) < want ? (size_t)(st.st_size - off) : want;
      const size_t per = ((n + NT - 1) / NT + 4095) & ~(size_t)4095;
      rd_job_t job[NT];
      pthread_t th[NT];
      int started[NT] = {0};
      for (int i = 0; i < NT; i++) {
         const size_t lo = (size_t)i * per < n ? (size_t)i * per : n;
         const size_t hi = lo + per < n ? lo + per : n;
         job[i] = (rd_job_t){fd, dst + lo, hi - lo, off + (off_t)lo, 0};
      }
      for (int i = 1; i < NT; i++) started[i] = job[i].n && pthread_create(&th[i], NULL, rd_worker, &job[i]) == 0;
      rd_worker(&job[0]);
      for (int i = 1; i < NT; i++) {
         if (started[i]) pthread_join(th[i], NULL);
         else if (job[i].n) rd_worker(&job[i]);
      }
      size_t total = 0;                                   /* the contiguous prefix that really arrived */
      for (int i = 0; i < NT; i++) {
         total += job[i].got;
         if (job[i].got < job[i].n) break;
      }
      if (fseeko(fdi, off + (off_t)total, SEEK_SET) == 0) return total;
      return total;
   }
   return fread(dst, 1, want, fdi);
}

/* Bring the next chunk in: keep the unfinished tail line, read on, and cut
 * at the last newline (everything, at EOF).  Returns -1 on allocation failure. */
static int refill(fstate_t *s, FILE *fdi)
{
   const size_t tail = s->len - s->avail;
   if (tail && s->avail) memmove(s->buf, s->buf + s->avail, tail);
   s->len = tail;
   s->avail = 0;
   s->pos = 0;
   s->have = 0;
   if (!s->buf) {
      s->cap = chunk_bytes();
      s->buf = seeqdevHostAlloc(s->cap);            /* page-locked: H2D at link speed */
      if (!s->buf) return -1;
   }
   for (;;) {
      if (s->len == s->cap) {           /* one line longer than the chunk: grow */
         char *g = seeqdevHostAlloc(2 * s->cap);
         if (!g) return -1;
         memcpy(g, s->buf, s->len);
         seeqdevHostFree(s->buf);
         s->buf = g;
         s->cap *= 2;
      }
      const size_t got = chunk_read(fdi, s->buf + s->len, s->cap - s->len);
      const size_t scan_from = s->len;
      s->len += got;
      if (got == 0) {
         s->eof = 1;
         s->avail = s->len;
         return 0;
      }
      /* last newline: search only what is new when nothing older had one */
      (void)scan_from;
      const char *nl = memrchr(s->buf, '\n', s->len);
      if (nl) {
         s->avail = (size_t)(nl - s->buf) + 1;
         return 0;
      }
   }
}

/* getline-like: sq->string holds the line, NUL-terminated (seeq.c:361-364). */
static int set_string(seeq_t *sq, const char *line, size_t n)
{
   if (sq->bufsz < n + 2 || !sq->string) {
      size_t want = sq->bufsz ? sq->bufsz : 120;
      while (want < n + 2) want *= 2;
      char *g = realloc(sq->string, want);
      if (!g) return -1;
      sq->string = g;
      sq->bufsz = want;
   }
   memcpy(sq->string, line, n);
   sq->string[n] = 0;
   return 0;
}

/* reference seeq.c:201-256 */
seeqfile_t *seeqOpen(const char *file)
{
   seeqerr = 0;
   seeqfile_t *f = calloc(1, sizeof *f);
   if (!f) { seeqerr = errno; return NULL; }
   FILE *fdi = file ? fopen(file, "r") : stdin;
   if (!fdi) {
      seeqerr = errno;                    /* raw errno, reference seeq.c:234 */
      free(f);
      return NULL;
   }
   f->fdi = fdi;
   f->line = 0;
   const int c = getc(fdi);               /* FASTA sniff, reference seeq.c:243-253 */
   if (c == '>') {
      f->flags = 1;
      f->info = calloc(32, 1);
      if (!f->info) {
         seeqerr = errno;
         if (fdi != stdin) fclose(fdi);
         free(f);
         return NULL;
      }
   }
   if (c != EOF) ungetc(c, fdi);
   return f;
}

/* reference seeq.c:258-291 */
int seeqClose(seeqfile_t *f)
{
   seeqerr = 0;
   FILE *fdi = f->fdi;
   state_drop(f);
   free(f->info);
   f->info = NULL;
   free(f);
   if (fdi && fdi != stdin && fclose(fdi) != 0) {
      seeqerr = errno;
      return -1;
   }
   return 0;
}

/* Scan buf[pos, avail) once for the given options and cache the records. */
static int scan_chunk(fstate_t *s, seeq_engine_t *eng, int dev_opt, int want, seeqdev_counts_t *cnt)
{
   seeqdev_scan_t *scan = seeq_engine_scan(eng);
   if (!scan) return -1;
   if (seeqdevScanHost(scan, eng->pat, s->buf + s->pos, s->avail - s->pos, dev_opt, want, cnt)) return -1;
   if (want != SEEQDEV_WANT_RECORDS) return 0;
   if (cnt->nrecords > s->rec_cap) {
      seeqdev_hit_t *g = realloc(s->rec, cnt->nrecords * sizeof *g);
      if (!g) { seeqerr = 0; return -1; }
      s->rec = g;
      uint64_t *o = realloc(s->rec_off, cnt->nrecords * sizeof *o);
      if (!o) { seeqerr = 0; return -1; }
      s->rec_off = o;
      s->rec_cap = cnt->nrecords;
   }
   if (seeqdevScanCopyRecords(scan, s->rec, 0, cnt->nrecords)) return -1;
   if (seeqdevScanCopyOffsets(scan, s->rec_off, 0, cnt->nrecords)) return -1;
   s->scan_from = s->pos;
   s->scan_lines = cnt->nlines;
   s->nrec = cnt->nrecords;
   s->rec_pos = 0;
   s->counted = 0;
   return 0;
}

/* FASTA: remember the last header line ('>' first) that starts inside buf[lo, hi) (seeq.c:367-374). */
static int last_header(seeqfile_t *f, const char *buf, size_t lo, size_t hi)
{
   size_t q = hi;
   while (q > lo) {
      const char *g = memrchr(buf + lo, '>', q - lo);
      if (!g) return 0;
      const size_t at = (size_t)(g - buf);
      if (at == 0 || buf[at - 1] == '\n') {                       /* a line starts here */
         const char *nl = memchr(g, '\n', hi - at);
         const size_t n = nl ? (size_t)(nl - g) : hi - at;
         char *dup = strndup(g, n);
         if (!dup) { seeqerr = 666; return -1; }
         f->info = dup;                                            /* the reference does not free the old one either */
         return 0;
      }
      q = at;
   }
   return 0;
}

/* reference seeq.c:293-392 */
long seeqFileMatch(seeqfile_t *sqfile, seeq_t *sq, int match_opt, int file_opt)
{
   seeqerr = 0;
   const int fasta = sqfile->flags & 0x1;
   if (file_opt == SQ_COUNTMATCH) match_opt = (match_opt & ~MASK_MATCH) | SQ_ALL;         /* seeq.c:348 */
   else if (file_opt == SQ_COUNTLINES) match_opt = (match_opt & ~MASK_MATCH) | SQ_FIRST;  /* seeq.c:349 */
   if (sqfile->fdi == NULL) {                                                             /* seeq.c:351-354 */
      seeqerr = 10;
      return -1;
   }
   seeq_engine_t *eng = seeq_engine_of(sq);
   if (!eng) { errno = EINVAL; return -1; }
   fstate_t *s = state_of(sqfile, 1);
   if (!s) return -1;

   /* SQ_STREAM has no meaning per file line (getline has already split at '\n'). */
   const int dev_opt = (match_opt & (MASK_MATCH | MASK_NONDNA)) | (fasta ? SEEQDEV_FASTA : 0);
   const int counting = file_opt == SQ_COUNTLINES || file_opt == SQ_COUNTMATCH;
   long count = 0;
   const size_t startline = sqfile->line;

   for (;;) {
      if (s->pos >= s->avail) {
         if (s->eof) break;
         if (refill(s, sqfile->fdi)) { seeqerr = 0; return -1; }
         if (s->avail == 0) continue;          /* eof with nothing left -> break above */
      }
      if (counting) {
         /* Whole chunk in one go: only the counts travel back (seeq.c:104-107). */
         seeqdev_counts_t cnt;
         const int want = file_opt == SQ_COUNTLINES ? SEEQDEV_WANT_COUNTLINES : SEEQDEV_WANT_COUNTMATCH;
         if (scan_chunk(s, eng, dev_opt, want, &cnt)) return -1;
         count += (long)(file_opt == SQ_COUNTLINES ? cnt.nmatchlines : cnt.nhits);
         sqfile->line += cnt.nlines;
         /* side effects the reference leaves behind: last line in sq->string, last header in info */
         size_t e = s->avail;
         if (e > s->pos && s->buf[e - 1] == '\n') e--;
         const char *b = e > s->pos ? memrchr(s->buf + s->pos, '\n', e - s->pos) : NULL;
         const size_t ls = b ? (size_t)(b - s->buf) + 1 : s->pos;
         if (set_string(sq, s->buf + ls, e - ls)) { seeqerr = 0; return -1; }
         sq->hits = 0;
         if (fasta && cnt.nheaders) {
            size_t q = s->avail;
            while (q > s->pos) {               /* last line of the chunk that starts with '>' */
               size_t le = q;
               if (s->buf[le - 1] == '\n') le--;
               const char *p = le > s->pos ? memrchr(s->buf + s->pos, '\n', le - s->pos) : NULL;
               const size_t lb = p ? (size_t)(p - s->buf) + 1 : s->pos;
               if (s->buf[lb] == '>' && le > lb) {
                  char *dup = strndup(s->buf + lb, le - lb);
                  if (!dup) { seeqerr = 666; return -1; }
                  sqfile->info = dup;          /* the reference does not free the old one either (seeq.c:368) */
                  break;
               }
               q = lb;
            }
         }
         s->pos = s->avail;
         continue;
      }

      const int key = dev_opt;
      if (!s->have || s->eng_id != eng->id || s->opt_key != key) {
         seeqdev_counts_t cnt;
         if (scan_chunk(s, eng, dev_opt, SEEQDEV_WANT_RECORDS, &cnt)) return -1;
         s->have = 1;
         s->eng_id = eng->id;
         s->opt_key = key;
      }
      if (file_opt == SQ_MATCH) {
         /* Jump from hit to hit: every line in between has 0 hits, so the reference's loop (seeq.c:361-386)
            would just count it (and remember FASTA headers).  The device gave us each record's line offset. */
         if (s->rec_pos < s->nrec) {
            const seeqdev_hit_t *r = s->rec + s->rec_pos;
            const size_t lo = s->scan_from + (size_t)s->rec_off[s->rec_pos];
            size_t k = 1;
            while (s->rec_pos + k < s->nrec && r[k].line == r->line) k++;
            const char *line = s->buf + lo;
            const char *nl = memchr(line, '\n', s->avail - lo);
            const size_t n = nl ? (size_t)(nl - line) : s->avail - lo;
            if (fasta && last_header(sqfile, s->buf, s->pos, lo)) return -1;
            sqfile->line += r->line - s->counted;                  /* seeq.c:377, for all the lines skipped */
            s->counted = r->line;
            s->pos = lo + n + (nl ? 1 : 0);
            if (set_string(sq, line, n)) { seeqerr = 0; return -1; }
            if (seeq_store_hits(sq, r, k)) return -1;
            s->rec_pos += k;
            return 1;                                               /* count = k > 0: seeq.c:385-386 */
         }
         /* no hit left in this chunk: consume the rest */
         if (fasta && last_header(sqfile, s->buf, s->pos, s->avail)) return -1;
         sqfile->line += s->scan_lines - s->counted;
         s->counted = s->scan_lines;
         if (s->avail > s->pos) {                                   /* the last line read stays in sq->string */
            size_t e = s->avail;
            if (s->buf[e - 1] == '\n') e--;
            const char *b = e > s->pos ? memrchr(s->buf + s->pos, '\n', e - s->pos) : NULL;
            const size_t ls = b ? (size_t)(b - s->buf) + 1 : s->pos;
            if (set_string(sq, s->buf + ls, e - ls)) { seeqerr = 0; return -1; }   /* headers too: getline put them there */
            sq->hits = 0;
         }
         s->pos = s->avail;
         continue;
      }
      /* SQ_ANY / SQ_NOMATCH: replay line by line (seeq.c:361-386). */
      const char *line = s->buf + s->pos;
      const char *nl = memchr(line, '\n', s->avail - s->pos);
      const size_t n = nl ? (size_t)(nl - line) : s->avail - s->pos;
      s->pos += n + (nl ? 1 : 0);
      if (fasta && n > 0 && line[0] == '>') {                     /* seeq.c:367-374 */
         if (set_string(sq, line, n)) { seeqerr = 0; return -1; }
         sqfile->info = strdup(sq->string);
         if (!sqfile->info) { seeqerr = 666; return -1; }
         continue;
      }
      sqfile->line++;                                             /* seeq.c:377 */
      s->counted++;
      size_t k = 0;
      while (s->rec_pos + k < s->nrec && s->rec[s->rec_pos + k].line == s->counted) k++;
      const long rval = (long)k;
      const int stop = file_opt == SQ_ANY || (rval == 0 && file_opt == SQ_NOMATCH);
      /* sq->string / sq->match only matter for the line a call returns on, or the last line of the file */
      if (stop || s->pos >= s->avail) {
         if (set_string(sq, line, n)) { seeqerr = 0; return -1; }
         if (seeq_store_hits(sq, s->rec + s->rec_pos, k)) return -1;
      }
      s->rec_pos += k;
      count += rval;
      if (stop) return 1;                                         /* seeq.c:385-386 */
   }
   return sqfile->line == startline ? 0 : count;                  /* seeq.c:390-391 */
}

/* ------------------------------------------------------------------------ */
/* seeq(): open, match, print.  Output formats of reference seeq.c:104-176.   */
/* ------------------------------------------------------------------------ */
/* The formatter writes with the *_unlocked calls (the reference is single-threaded) and formats its integers
   itself: hit-rich or inverted outputs print tens of millions of short lines.  (The CLI gives stdout a 1 MiB buffer.) */
static void put_range(const char *s, size_t from, size_t to) { fwrite_unlocked(s + from, 1, to - from, stdout); }
static void put_str(const char *s) { fwrite_unlocked(s, 1, strlen(s), stdout); }
static void put_long(long v, char trail)            /* "%ld" followed by `trail` (0 = nothing) */
{
   char b[24];
   int i = 23;
   unsigned long u = v < 0 ? 0ul - (unsigned long)v : (unsigned long)v;
   if (trail) b[i--] = trail;
   do { b[i--] = (char)('0' + u % 10); u /= 10; } while (u);
   if (v < 0) b[i--] = '-';
   fwrite_unlocked(b + i + 1, 1, (size_t)(23 - i), stdout);
}

static void print_hit(const struct seeqarg_t *a, const seeqfile_t *f, const seeq_t *sq, const match_t *m,
                      int fasta_header, int color)
{
   const char *str = sq->string;
   if (a->compact) {                                              /* seeq.c:134 */
      put_long((long)f->line, ':'); put_long((long)m->start, '-'); put_long((long)m->end - 1, ':');
      put_long((long)m->dist, '\n');
      return;
   }
   if (a->showline) put_long((long)f->line, ' ');
   if (a->showpos)  { put_long((long)m->start, '-'); put_long((long)m->end - 1, ' '); }
   if (a->showdist) put_long((long)m->dist, ' ');
   if (fasta_header) { put_str(f->info); putc_unlocked('\n', stdout); }
   const size_t len = strlen(str);
   if (a->matchonly) {
      put_range(str, m->start, m->end < len ? m->end : len);
   } else if (a->prefix) {
      put_range(str, 0, m->start < len ? m->start : len);
   } else if (a->endline) {
      if (m->end < len) put_range(str, m->end, len);
   } else if (a->split) {
      put_range(str, 0, m->start); putc_unlocked('\t', stdout);
      put_range(str, m->start, m->end); putc_unlocked('\t', stdout);
      if (m->end < len) put_range(str, m->end, len);
   } else if (a->printline) {
      if (color) {
         put_range(str, 0, m->start);
         put_str(m->dist ? BOLDRED : BOLDGREEN);
         put_range(str, m->start, m->end);
         put_str(RESET);
         if (m->end < len) put_range(str, m->end, len);
      } else {
         put_range(str, 0, len);
      }
   }
   putc_unlocked('\n', stdout);
}

int seeq(char *expression, char *input, struct seeqarg_t args)
{
   seeq_t *sq = seeqNew(expression, args.dist, args.memory);
   if (!sq) {
      fprintf(stderr, "error in 'seeqNew()'; %s\n:", seeqPrintError());   /* sic, seeq.c:79 */
      return EXIT_FAILURE;
   }
   if (args.verbose) fprintf(stderr, "opening input file... ");
   seeqfile_t *f = seeqOpen(input);
   if (!f) {
      fprintf(stderr, "error in 'seeqOpen()': %s\n", seeqPrintError());
      seeqFree(sq);
      return EXIT_FAILURE;
   }
   const int is_fasta = f->flags & 0x1;
   struct timespec t0 = {0, 0}, t1;
   if (args.verbose) {
      fprintf(stderr, "\nmatching...\n");
      clock_gettime(CLOCK_MONOTONIC, &t0);
   }
   int opt = 0;
   if (args.non_dna == 1) opt |= SQ_CONVERT;
   else if (args.non_dna == 2) opt |= SQ_IGNORE;

   if (args.count) {
      const long n = seeqFileMatch(f, sq, opt, SQ_COUNTLINES);
      if (n < 0) fprintf(stderr, "error in 'seeqFileMatch()': %s\n", seeqPrintError());
      else fprintf(stdout, "%ld\n", n);
   } else {
      if (args.all) { opt |= SQ_ALL; args.matchonly = 1; }            /* seeq.c:109-112 */
      else if (args.best) opt |= SQ_BEST;
      const int fasta_header = is_fasta && !args.split && !args.showline && !args.showpos && !args.showdist;
      const int color = COLOR_TERMINAL && isatty(fileno(stdout));
      long rv;
      if (args.invert) {
         while ((rv = seeqFileMatch(f, sq, opt, SQ_NOMATCH)) > 0) {    /* seeq.c:125-129 */
            if (args.showline) put_long((long)f->line, ' ');
            if (fasta_header) { put_str(f->info); putc_unlocked('\n', stdout); }
            put_str(sq->string); putc_unlocked('\n', stdout);
         }
      } else {
         while ((rv = seeqFileMatch(f, sq, opt, SQ_MATCH)) > 0) {      /* seeq.c:131-176 */
            const match_t *m;
            while ((m = seeqMatchIter(sq)) != NULL) print_hit(&args, f, sq, m, fasta_header, color);
         }
      }
      if (rv == -1) fprintf(stderr, "error in 'seeqFileMatch()': %s\n", seeqPrintError());
   }
   if (args.verbose) {
      clock_gettime(CLOCK_MONOTONIC, &t1);
      fprintf(stderr, "engine: %s (HIP, no DFA cache)\n", SEEQ_AMD_VERSION);
      fprintf(stderr, "done in %.3fs\n", (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec));
   }
   seeqFree(sq);
   seeqClose(f);
   return EXIT_SUCCESS;
}
