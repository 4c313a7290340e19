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
#include <poll.h>
#include <pthread.h>
#include <sys/stat.h>

#include "seeq.h"
#include "seeq_amd.h"
#include "seeq_internal.h"

/* ------------------------------------------------------------------------ */
/* The ingest pipeline, one per open seeqfile_t (kept beside the public       */
/* struct so that its layout, reference seeq.h:55-60, is untouched).          */
/*                                                                            */
/*   reader thread : file -> page-locked chunks (cut at the last '\n', the    */
/*                   unfinished tail line carried into the next chunk)        */
/*   GPU lanes     : chunk k+1 is staged (H2D) and scanned on its own stream  */
/*                   while chunk k is still being scanned / replayed; two     */
/*                   lanes per device, SEEQ_DEVICES=0,1,.. spreads successive */
/*                   chunks over several GPUs (lines are independent: the     */
/*                   host just adds the counts and runs the line number on)   */
/*   caller        : replays the records of the oldest chunk call by call     */
/* ------------------------------------------------------------------------ */
#define MAX_SLOTS 12
#define MAX_LANES 16
enum { SL_FREE = 0, SL_FILLING, SL_FILLED, SL_SCANNING, SL_READY };

typedef struct lane_t {
   int                device;
   seeqdev_pattern_t *pat;       /* the engine's own pattern on its device, a private copy on any other device */
   int                own_pat;
   seeqdev_scan_t    *scan;
   int                slot;      /* index of the chunk whose scan is in flight, -1: idle */
} lane_t;

typedef struct slot_t {
   char            *buf;         /* page-locked chunk */
   size_t           cap, len;    /* capacity, bytes read */
   size_t           avail;       /* bytes that belong to complete lines (all of len at EOF) */
   int              eof;         /* the input ended with this chunk */
   int              state;
   /* the scan of buf[scan_from, avail) */
   int              lane;
   unsigned long    eng_id;
   int              opt_key, want;
   size_t           scan_from;
   seeqdev_counts_t cnt;
   seeqdev_hit_t   *rec;
   uint64_t        *rec_off;     /* per record: offset of its line, relative to scan_from */
   size_t           rec_cap;
   size_t           counted;     /* counted lines replayed since scan_from */
   size_t           rec_pos;
} slot_t;

typedef struct fstate_t {
   struct fstate_t *next;
   seeqfile_t      *key;
   FILE            *fdi;
   int              raw_fd;      /* not a regular file: the stream is unbuffered and read with read(2), chunks are handed over as lines arrive */
   slot_t           slot[MAX_SLOTS];
   int              nslots;
   unsigned long    head;        /* sequence number of the chunk being replayed (slot = seq % nslots) */
   unsigned long    filled;      /* chunks the reader has published */
   size_t           pos;         /* next unread byte of the head chunk */
   pthread_t        reader;
   pthread_mutex_t  mu;
   pthread_cond_t   cv;
   int              sync_init, reader_on, reader_stop, reader_done, reader_err, waiting;
   int              dead;          /* errno of the device / reader failure that ended this file's scanning: every later seeqFileMatch fails with it (no call into a
                                      device context whose state is unknown, no line out of order after a lost chunk) */
   lane_t           lane[MAX_LANES];
   int              nlanes, next_lane;
   unsigned long    lanes_eng;
   seeqdev_pattern_t *dev_pat[MAX_LANES];   /* private pattern copies, one per foreign device */
   /* -z / --verbose */
   double           t_read, t_wait_reader, t_wait_gpu, ms_h2d, ms_kernels;
   size_t           bytes, chunks;
} fstate_t;

static fstate_t *g_states = NULL;
static int g_profile = 0;          /* seeq -z: scan contexts record HIP events (H2D, kernels) */

static double now_s(void)
{
   struct timespec t;
   clock_gettime(CLOCK_MONOTONIC, &t);
   return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static size_t chunk_bytes(void)
{
   const char *env = getenv("SEEQ_CHUNK_BYTES");
   if (env) {
      long long v = atoll(env);
      if (v >= 64) return (size_t)v;
   }
   return (size_t)32 << 20;        /* page-locking a chunk costs ~0.18 ms per MiB, a chunk's scan ~0.2 ms of fixed work */
}

static fstate_t *state_of(seeqfile_t *f, int create)
{
   for (fstate_t *s = g_states; s; s = s->next)
      if (s->key == f) return s;
   if (!create) return NULL;
   fstate_t *s = calloc(1, sizeof *s);
   if (!s) return NULL;
   s->key = f;
   s->fdi = f->fdi;
   s->next = g_states;
   g_states = s;
   return s;
}

/* Wait for the scans in flight and forget every cached result (the chunks stay). */
static void quiesce(fstate_t *s)
{
   for (int i = 0; i < s->nlanes; i++) {
      lane_t *ln = &s->lane[i];
      if (ln->slot >= 0) {
         seeqdev_counts_t c;
         (void)seeqdevScanFetch(ln->scan, &c);
         ln->slot = -1;
      }
   }
   if (s->sync_init) pthread_mutex_lock(&s->mu);
   for (int i = 0; i < s->nslots; i++)
      if (s->slot[i].state == SL_SCANNING || s->slot[i].state == SL_READY) s->slot[i].state = SL_FILLED;
   if (s->sync_init) pthread_mutex_unlock(&s->mu);
}

static void drop_lanes(fstate_t *s)
{
   quiesce(s);
   for (int i = 0; i < s->nlanes; i++)
      if (s->lane[i].scan) seeqdevScanFree(s->lane[i].scan);
   for (int i = 0; i < MAX_LANES; i++)
      if (s->dev_pat[i]) { seeqdevPatternFree(s->dev_pat[i]); s->dev_pat[i] = NULL; }
   s->nlanes = 0;
   s->lanes_eng = 0;
}

/* seeqFree() of an engine whose pattern the lanes of an open file still use (the CLI frees the seeq_t first). */
void seeq_file_forget_engine(unsigned long eng_id)
{
   for (fstate_t *s = g_states; s; s = s->next)
      if (s->nlanes && s->lanes_eng == eng_id) drop_lanes(s);
}

static void state_drop(seeqfile_t *f)
{
   for (fstate_t **pp = &g_states; *pp; pp = &(*pp)->next) {
      if ((*pp)->key == f) {
         fstate_t *s = *pp;
         *pp = s->next;
         if (s->reader_on) {                              /* it may sit in read(2) on a pipe: cancel, then join */
            pthread_mutex_lock(&s->mu);
            s->reader_stop = 1;
            pthread_cond_broadcast(&s->cv);
            pthread_mutex_unlock(&s->mu);
            pthread_cancel(s->reader);
            pthread_join(s->reader, NULL);
         }
         drop_lanes(s);
         for (int i = 0; i < MAX_SLOTS; i++) {
            seeqdevHostFree(s->slot[i].buf);
            free(s->slot[i].rec);
            free(s->slot[i].rec_off);
         }
         if (s->sync_init) { pthread_mutex_destroy(&s->mu); pthread_cond_destroy(&s->cv); }
         free(s);
         return;
      }
   }
}

/* fread() for big reads of a regular file: the range is split over a few threads that pread() it straight into the
 * (page-locked) chunk buffer -- one thread copies out of the page cache at ~5-10 GB/s, four reach ~25 GB/s (measured,
 * profiles/r02_cli_wallclock.txt), eight come close to the link speed the GPU side can take. */
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
   enum { NTMAX = 16 };
   static int nt_env = 0;                                  /* SEEQ_READ_THREADS (1..16), default 8 */
   if (!nt_env) {
      const char *e = getenv("SEEQ_READ_THREADS");
      nt_env = e && atoi(e) >= 1 && atoi(e) <= NTMAX ? atoi(e) : 8;
   }
   const int NT = nt_env;
   const size_t big = (size_t)8 << 20;
   struct stat st;
   const int fd = fileno(fdi);
   off_t off;
   if (want >= big && fd >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && (off = ftello(fdi)) >= 0 &&
       st.st_size > off && (size_t)(st.st_size - off) >= big) {
      const size_t n = (size_t)(st.st_size - off) < want ? (size_t)(st.st_size - off) : want;
      const size_t per = ((n + NT - 1) / NT + 4095) & ~(size_t)4095;
      rd_job_t job[NTMAX];
      pthread_t th[NTMAX];
      int started[NTMAX] = {0};
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

/* A chunk buffer of at least `want` bytes, contents kept. */
static int slot_reserve(slot_t *sl, size_t want)
{
   if (sl->cap >= want && sl->buf) return 0;
   size_t cap = sl->cap ? sl->cap : chunk_bytes();
   while (cap < want) cap *= 2;
   char *g = seeqdevHostAlloc(cap);                      /* page-locked: H2D at link speed */
   if (!g) return -1;
   if (sl->buf) {
      memcpy(g, sl->buf, sl->len);
      seeqdevHostFree(sl->buf);
   }
   sl->buf = g;
   sl->cap = cap;
   return 0;
}

/* The reader thread: chunk after chunk, each cut at its last newline (everything, at EOF); the unfinished tail
 * line opens the next chunk.  (reference seeq.c:361: getline, one line at a time.) */
static void *reader_main(void *arg)
{
   fstate_t *s = arg;
   int unused;
   pthread_setcancelstate(PTHREAD_CANCEL_DISABLE, &unused);
   const char *carry = NULL;
   size_t ncarry = 0;
   int first_raw = s->raw_fd;
   for (unsigned long seq = 0;; seq++) {
      slot_t *sl = &s->slot[seq % (unsigned long)s->nslots];
      pthread_mutex_lock(&s->mu);
      while (sl->state != SL_FREE && !s->reader_stop) pthread_cond_wait(&s->cv, &s->mu);
      if (s->reader_stop) { pthread_mutex_unlock(&s->mu); break; }
      sl->state = SL_FILLING;
      pthread_mutex_unlock(&s->mu);
      int err = 0, eof = 0;
      sl->len = 0;
      if (slot_reserve(sl, ncarry + 4096)) err = errno ? errno : ENOMEM;
      if (!err && ncarry) { memmove(sl->buf, carry, ncarry); sl->len = ncarry; }
      size_t nl_end = 0;                                  /* offset just behind the last newline seen in this chunk */
      while (!err) {
         if (sl->len == sl->cap && slot_reserve(sl, 2 * sl->cap)) { err = errno ? errno : ENOMEM; break; }   /* one line longer than the chunk */
         const double t0 = now_s();
         size_t got = 0;
         if (s->raw_fd) {
            if (first_raw) {                              /* the byte seeqOpen sniffed and pushed back */
               first_raw = 0;
               const int c = getc(s->fdi);
               if (c != EOF) { sl->buf[sl->len] = (char)c; got = 1; }
            }
            if (!got) {
               pthread_setcancelstate(PTHREAD_CANCEL_ENABLE, &unused);
               ssize_t r;
               do r = read(fileno(s->fdi), sl->buf + sl->len, sl->cap - sl->len); while (r < 0 && errno == EINTR);
               pthread_setcancelstate(PTHREAD_CANCEL_DISABLE, &unused);
               if (r < 0) { err = errno; break; }
               got = (size_t)r;
            }
         } else {
            /* (not cancellable: chunk_read joins its pread threads, and a regular file never blocks for long --
               seeqClose waits for this chunk and the reader then sees reader_stop) */
            got = chunk_read(s->fdi, sl->buf + sl->len, sl->cap - sl->len);
         }
         s->t_read += now_s() - t0;
         if (got == 0) { eof = 1; break; }
         const char *nl = memrchr(sl->buf + sl->len, '\n', got);
         if (nl) nl_end = (size_t)(nl - sl->buf) + 1;
         sl->len += got;
         if (!nl_end) continue;                           /* no complete line yet */
         if (sl->len == sl->cap) break;
         if (s->raw_fd) {                                 /* a pipe: hand the lines over as soon as somebody waits for them */
            pthread_mutex_lock(&s->mu);
            const int waiting = s->waiting;
            pthread_mutex_unlock(&s->mu);
            if (waiting) break;
            /* nobody waits yet -- but the caller may start to while this thread sleeps in read(2) with complete lines in
               hand (then nobody would ever hand them over): read on only while the producer keeps delivering */
            struct pollfd pf = { fileno(s->fdi), POLLIN, 0 };
            if (poll(&pf, 1, 2) == 0) break;
         }
      }
      sl->avail = eof ? sl->len : nl_end;
      sl->eof = eof;
      carry = sl->buf + sl->avail;
      ncarry = sl->len - sl->avail;
      pthread_mutex_lock(&s->mu);
      if (err) s->reader_err = err;
      else { sl->state = SL_FILLED; s->filled = seq + 1; s->bytes += sl->avail; }
      if (err || eof) s->reader_done = 1;
      pthread_cond_broadcast(&s->cv);
      pthread_mutex_unlock(&s->mu);
      if (err || eof) break;
   }
   return NULL;
}

/* SEEQ_DEVICES="0,2,5" / "0-7" / "all": the devices successive chunks go to.  Default: the engine's own device. */
static int parse_devices(int *dev, int maxn, int own)
{
   const char *env = getenv("SEEQ_DEVICES");
   const int have = seeqdevDeviceCount();
   int n = 0;
   if (env && *env) {
      if (!strcmp(env, "all")) {
         for (int d = 0; d < have && n < maxn; d++) dev[n++] = d;
      } else {
         const char *p = env;
         while (*p && n < maxn) {
            char *e;
            long a = strtol(p, &e, 10), b = a;
            if (e == p) break;
            if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); if (e == p) break; }
            for (long d = a; d <= b && n < maxn; d++)
               if (d >= 0 && d < have) dev[n++] = (int)d;
            p = *e == ',' ? e + 1 : e;
            if (*e && *e != ',') break;
         }
      }
   }
   if (n == 0) dev[n++] = own;
   return n;
}

static int ensure_lanes(fstate_t *s, seeq_engine_t *eng, const seeq_t *sq)
{
   if (s->nlanes && s->lanes_eng == eng->id) return 0;
   if (s->nlanes) drop_lanes(s);
   const int own = seeqdevPatternDevice(eng->pat);
   int dev[MAX_LANES / 2];
   const int ndev = parse_devices(dev, MAX_LANES / 2, own);
   int per = 2;                                           /* lanes per device: the H2D of one overlaps the kernels of the other */
   const char *env = getenv("SEEQ_LANES");
   if (env && atoi(env) >= 1 && atoi(env) <= 4) per = atoi(env);
   while (ndev * per > MAX_LANES) per--;
   int n = 0, rc = 0;
   for (int k = 0; k < ndev * per && !rc; k++) {
      const int d = dev[k % ndev];                        /* consecutive lanes sit on different devices */
      lane_t *ln = &s->lane[n];
      memset(ln, 0, sizeof *ln);
      ln->device = d;
      ln->slot = -1;
      if (seeqdevSetDevice(d)) { rc = -1; break; }
      if (d == own) ln->pat = eng->pat;
      else {
         if (!s->dev_pat[k % ndev]) s->dev_pat[k % ndev] = seeqdevPatternNew(sq->keys, sq->wlen, sq->tau);
         ln->pat = s->dev_pat[k % ndev];
      }
      if (ln->pat) ln->scan = seeqdevScanNew(NULL);
      if (!ln->pat || !ln->scan) { rc = -1; break; }
      if (g_profile) (void)seeqdevScanSetProfiling(ln->scan, 1);
      n++;
   }
   s->nlanes = n;
   (void)seeqdevSetDevice(own);
   if (rc) { drop_lanes(s); return -1; }
   s->lanes_eng = eng->id;
   s->next_lane = 0;
   return 0;
}

static int ensure_reader(fstate_t *s)
{
   if (s->reader_on) return 0;
   if (!s->sync_init) {
      pthread_mutex_init(&s->mu, NULL);
      pthread_cond_init(&s->cv, NULL);
      s->sync_init = 1;
   }
   s->nslots = s->nlanes + 2;                             /* one being filled, one being replayed, the rest on the GPUs */
   if (s->nslots > MAX_SLOTS) s->nslots = MAX_SLOTS;
   if (s->nslots < 3) s->nslots = 3;
   if (pthread_create(&s->reader, NULL, reader_main, s)) return -1;
   s->reader_on = 1;
   return 0;
}

/* Chunk states change under the mutex: the reader polls the state of the chunk it wants next. */
static void set_state(fstate_t *s, slot_t *sl, int st)
{
   pthread_mutex_lock(&s->mu);
   sl->state = st;
   pthread_mutex_unlock(&s->mu);
}

/* Enqueue the scan of chunk `sl` on an idle lane (asynchronous); 0 when none is idle. */
static int begin_scan(fstate_t *s, slot_t *sl, int idx, size_t from, seeq_engine_t *eng, int dev_opt, int want)
{
   lane_t *ln = NULL;
   int li = -1;
   for (int k = 0; k < s->nlanes; k++) {
      const int c = (s->next_lane + k) % s->nlanes;
      if (s->lane[c].slot < 0) { ln = &s->lane[c]; li = c; break; }
   }
   if (!ln) return 0;
   s->next_lane = (li + 1) % s->nlanes;
   sl->scan_from = from;
   sl->eng_id = eng->id;
   sl->opt_key = dev_opt;
   sl->want = want;
   sl->lane = li;
   sl->counted = 0;
   sl->rec_pos = 0;
   if (seeqdevScanHostBegin(ln->scan, ln->pat, sl->buf + from, sl->avail - from, dev_opt, want)) return -1;
   ln->slot = idx;
   set_state(s, sl, SL_SCANNING);
   s->chunks++;
   return 1;
}

static int finish_scan(fstate_t *s, slot_t *sl)
{
   lane_t *ln = &s->lane[sl->lane];
   const double t0 = now_s();
   const int rc = seeqdevScanFetch(ln->scan, &sl->cnt);
   s->t_wait_gpu += now_s() - t0;
   ln->slot = -1;
   if (rc) return -1;
   if (g_profile) {
      float ms[4] = {0, 0, 0, 0}, h2d = 0;
      (void)seeqdevScanLastTimes(ln->scan, ms);
      (void)seeqdevScanLastCopyMs(ln->scan, &h2d);
      s->ms_kernels += ms[3];
      s->ms_h2d += h2d;
   }
   if (sl->want != SEEQDEV_WANT_RECORDS) return 0;
   const size_t n = sl->cnt.nrecords;
   if (n > sl->rec_cap) {
      seeqdev_hit_t *g = realloc(sl->rec, n * sizeof *g);
      if (!g) { seeqerr = 0; return -1; }
      sl->rec = g;
      uint64_t *o = realloc(sl->rec_off, n * sizeof *o);
      if (!o) { seeqerr = 0; return -1; }
      sl->rec_off = o;
      sl->rec_cap = n;
   }
   if (seeqdevScanCopyRecords(ln->scan, sl->rec, 0, n)) return -1;
   if (seeqdevScanCopyOffsets(ln->scan, sl->rec_off, 0, n)) return -1;
   return 0;
}

/* The oldest chunk, scanned for (engine, options, want) from s->pos on; *out = NULL at the end of the input.
 * Side effect: every published chunk behind it that finds an idle lane gets its scan enqueued (read-ahead). */
static int pump(fstate_t *s, seeq_engine_t *eng, const seeq_t *sq, int dev_opt, int want, slot_t **out)
{
   *out = NULL;
   if (ensure_lanes(s, eng, sq)) return -1;
   if (ensure_reader(s)) { seeqerr = 0; return -1; }
   for (;;) {
      pthread_mutex_lock(&s->mu);
      const unsigned long filled = s->filled;
      pthread_mutex_unlock(&s->mu);
      /* (only this thread moves chunks out of FILLED and back; the reader only touches FREE chunks) */
      slot_t *head = &s->slot[s->head % (unsigned long)s->nslots];
      /* The head chunk was scanned (or is being scanned) for another pattern / option set / kind of result: so were the
         chunks read ahead behind it.  Wait for the scans in flight and forget all of them -- resetting the head alone left
         it FILLED with every lane busy on chunks nobody would ever collect, and this thread asleep for good (found by the
         advisor, round 2; tests/host_driver.c section 2b). */
      if (s->head < filled && (head->state == SL_READY || head->state == SL_SCANNING) &&
          (head->eng_id != eng->id || head->opt_key != dev_opt || head->want != want)) quiesce(s);
      for (unsigned long seq = s->head; seq < filled; seq++) {
         const int idx = (int)(seq % (unsigned long)s->nslots);
         slot_t *sl = &s->slot[idx];
         if (sl->state != SL_FILLED) continue;
         const size_t from = seq == s->head ? s->pos : 0;
         if (from >= sl->avail) {                         /* nothing (left) to scan */
            memset(&sl->cnt, 0, sizeof sl->cnt);
            sl->scan_from = from; sl->eng_id = eng->id; sl->opt_key = dev_opt; sl->want = want;
            sl->counted = 0; sl->rec_pos = 0;
            set_state(s, sl, SL_READY);
            continue;
         }
         const int r = begin_scan(s, sl, idx, from, eng, dev_opt, want);
         if (r < 0) return -1;
         if (r == 0) break;                               /* every lane is busy */
      }
      if (s->head < filled) {
         if (head->state == SL_SCANNING) {
            if (finish_scan(s, head)) return -1;
            set_state(s, head, SL_READY);
            continue;                                     /* its lane is idle again: feed it before replaying */
         }
         if (head->state == SL_READY) { *out = head; return 0; }
      }
      pthread_mutex_lock(&s->mu);
      if (s->reader_err) { errno = s->reader_err; seeqerr = 0; pthread_mutex_unlock(&s->mu); return -1; }
      if (s->filled == filled) {
         if (s->reader_done && s->head >= s->filled) { pthread_mutex_unlock(&s->mu); return 0; }   /* end of the input */
         const double t0 = now_s();
         s->waiting = 1;
         /* about to sleep until a pipe delivers more: what the caller has printed so far should be visible by then
            (`producer | seeq ... | consumer` behaves line by line with the reference's getline loop) */
         if (s->raw_fd) { pthread_mutex_unlock(&s->mu); fflush(stdout); pthread_mutex_lock(&s->mu); if (s->filled != filled) { s->waiting = 0; pthread_mutex_unlock(&s->mu); continue; } }
         pthread_cond_wait(&s->cv, &s->mu);
         s->waiting = 0;
         s->t_wait_reader += now_s() - t0;
      }
      pthread_mutex_unlock(&s->mu);
   }
}

/* The head chunk is replayed: give it back to the reader. */
static void advance(fstate_t *s)
{
   pthread_mutex_lock(&s->mu);
   s->slot[s->head % (unsigned long)s->nslots].state = SL_FREE;
   s->head++;
   s->pos = 0;
   pthread_cond_broadcast(&s->cv);
   pthread_mutex_unlock(&s->mu);
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
   /* Not a regular file (a pipe, a terminal): read with read(2) from the reader thread so that lines can be handed on as
      they arrive instead of when 64 MiB are full; the stream is made unbuffered first, so that nothing but the byte
      sniffed below is ever held back in the FILE. */
   struct stat st;
   int raw = 0;
   if (fstat(fileno(fdi), &st) != 0 || !S_ISREG(st.st_mode)) raw = setvbuf(fdi, NULL, _IONBF, 0) == 0;
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
   if (raw) {
      fstate_t *s = state_of(f, 1);
      if (s) s->raw_fd = 1;
   }
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
   if (!s->reader_on) s->fdi = sqfile->fdi;          /* (the reader thread owns the stream from its start on) */

   /* SQ_STREAM has no meaning per file line (getline has already split at '\n'). */
   const int dev_opt = (match_opt & (MASK_MATCH | MASK_NONDNA)) | (fasta ? SEEQDEV_FASTA : 0);
   const int counting = file_opt == SQ_COUNTLINES || file_opt == SQ_COUNTMATCH;
   const int want = file_opt == SQ_COUNTLINES ? SEEQDEV_WANT_COUNTLINES
                  : file_opt == SQ_COUNTMATCH ? SEEQDEV_WANT_COUNTMATCH : SEEQDEV_WANT_RECORDS;
   long count = 0;
   const size_t startline = sqfile->line;

   for (;;) {
      slot_t *c;
      /* A failure of the device boundary (HIP: -1, seeqerr = 0, errno = ENOMEM / EIO / ENODEV) or of the reader ends the scanning of this file: the
         chunk it happened on is lost and a device context may be in any state, so every later call reports the same failure (the reference, with
         nothing but malloc to fail, loses the line and goes on: seeq.c:361-371; its callers stop at the first -1: seeq.c:131,176) */
      if (s->dead) { seeqerr = 0; errno = s->dead; return -1; }
      if (pump(s, eng, sq, dev_opt, want, &c)) { if (seeqerr == 0) s->dead = errno ? errno : EIO; return -1; }
      if (!c) break;                           /* end of the input, everything replayed */
      if (s->pos >= c->avail) {                /* this chunk is done: the reader may have it back */
         advance(s);
         continue;
      }
      if (counting) {
         /* Whole chunk in one go: only the counts travel back (seeq.c:104-107). */
         const seeqdev_counts_t cnt = c->cnt;
         count += (long)(file_opt == SQ_COUNTLINES ? cnt.nmatchlines : cnt.nhits);
         sqfile->line += cnt.nlines;
         /* side effects the reference leaves behind: last line in sq->string, last header in info */
         {
            size_t e = c->avail;
            if (e > s->pos && c->buf[e - 1] == '\n') e--;
            const char *b = e > s->pos ? memrchr(c->buf + s->pos, '\n', e - s->pos) : NULL;
            const size_t ls = b ? (size_t)(b - c->buf) + 1 : s->pos;
            if (set_string(sq, c->buf + ls, e - ls)) { seeqerr = 0; return -1; }
         }
         sq->hits = 0;
         if (fasta && cnt.nheaders) {
            size_t q = c->avail;
            while (q > s->pos) {               /* last line of the chunk that starts with '>' */
               size_t le = q;
               if (c->buf[le - 1] == '\n') le--;
               const char *p = le > s->pos ? memrchr(c->buf + s->pos, '\n', le - s->pos) : NULL;
               const size_t lb = p ? (size_t)(p - c->buf) + 1 : s->pos;
               if (c->buf[lb] == '>' && le > lb) {
                  char *dup = strndup(c->buf + lb, le - lb);
                  if (!dup) { seeqerr = 666; return -1; }
                  sqfile->info = dup;          /* the reference does not free the old one either (seeq.c:368) */
                  break;
               }
               q = lb;
            }
         }
         s->pos = c->avail;
         continue;
      }

      const size_t nrec = c->cnt.nrecords;
      if (file_opt == SQ_MATCH) {
         /* Jump from hit to hit: every line in between has 0 hits, so the reference's loop (seeq.c:361-386)
            would just count it (and remember FASTA headers).  The device gave us each record's line offset. */
         if (c->rec_pos < nrec) {
            const seeqdev_hit_t *r = c->rec + c->rec_pos;
            const size_t lo = c->scan_from + (size_t)c->rec_off[c->rec_pos];
            size_t k = 1;
            while (c->rec_pos + k < nrec && r[k].line == r->line) k++;
            const char *line = c->buf + lo;
            const char *nl = memchr(line, '\n', c->avail - lo);
            const size_t n = nl ? (size_t)(nl - line) : c->avail - lo;
            if (fasta && last_header(sqfile, c->buf, s->pos, lo)) return -1;
            sqfile->line += r->line - c->counted;                  /* seeq.c:377, for all the lines skipped */
            c->counted = r->line;
            s->pos = lo + n + (nl ? 1 : 0);
            if (set_string(sq, line, n)) { seeqerr = 0; return -1; }
            if (seeq_store_hits(sq, r, k)) return -1;
            c->rec_pos += k;
            return 1;                                               /* count = k > 0: seeq.c:385-386 */
         }
         /* no hit left in this chunk: consume the rest */
         if (fasta && last_header(sqfile, c->buf, s->pos, c->avail)) return -1;
         sqfile->line += c->cnt.nlines - c->counted;
         c->counted = c->cnt.nlines;
         if (c->avail > s->pos) {                                   /* the last line read stays in sq->string */
            size_t e = c->avail;
            if (c->buf[e - 1] == '\n') e--;
            const char *b = e > s->pos ? memrchr(c->buf + s->pos, '\n', e - s->pos) : NULL;
            const size_t ls = b ? (size_t)(b - c->buf) + 1 : s->pos;
            if (set_string(sq, c->buf + ls, e - ls)) { seeqerr = 0; return -1; }   /* headers too: getline put them there */
            sq->hits = 0;
         }
         s->pos = c->avail;
         continue;
      }
      /* SQ_ANY / SQ_NOMATCH: replay line by line (seeq.c:361-386). */
      const char *line = c->buf + s->pos;
      const char *nl = memchr(line, '\n', c->avail - s->pos);
      const size_t n = nl ? (size_t)(nl - line) : c->avail - s->pos;
      s->pos += n + (nl ? 1 : 0);
      if (fasta && n > 0 && line[0] == '>') {                     /* seeq.c:367-374 */
         if (set_string(sq, line, n)) { seeqerr = 0; return -1; }
         sqfile->info = strdup(sq->string);
         if (!sqfile->info) { seeqerr = 666; return -1; }
         continue;
      }
      sqfile->line++;                                             /* seeq.c:377 */
      c->counted++;
      size_t k = 0;
      while (c->rec_pos + k < nrec && c->rec[c->rec_pos + k].line == c->counted) k++;
      const long rval = (long)k;
      const int stop = file_opt == SQ_ANY || (rval == 0 && file_opt == SQ_NOMATCH);
      /* sq->string / sq->match only matter for the line a call returns on, or the last line of the file */
      if (stop || s->pos >= c->avail) {
         if (set_string(sq, line, n)) { seeqerr = 0; return -1; }
         if (seeq_store_hits(sq, c->rec + c->rec_pos, k)) return -1;
      }
      c->rec_pos += k;
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
   const double t_new0 = now_s();
   seeq_t *sq = seeqNew(expression, args.dist, args.memory);
   const double t_new = now_s() - t_new0;     /* first HIP call of the process: runtime + device initialisation */
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
      g_profile = 1;                          /* HIP events around the H2D copies and the kernels of every chunk */
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
      const double wall = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
      fprintf(stderr, "engine: %s (HIP, no DFA cache); seeqNew incl. HIP start-up %.3fs\n", SEEQ_AMD_VERSION, t_new);
      const fstate_t *s = state_of(f, 0);
      if (s && s->chunks) {
         const double gb = (double)s->bytes / 1e9;
         fprintf(stderr, "ingest: %zu chunk(s), %.3f GB, %d lane(s); reader %.3fs (%.2f GB/s, own thread); H2D %.1f ms (%.1f GB/s); kernels %.1f ms (%.1f GB/s);\n"
                         "        caller waited %.3fs for the reader, %.3fs for the GPU; replay + output %.3fs\n",
                 s->chunks, gb, s->nlanes, s->t_read, s->t_read > 0 ? gb / s->t_read : 0.0, s->ms_h2d, s->ms_h2d > 0 ? gb / (s->ms_h2d * 1e-3) : 0.0,
                 s->ms_kernels, s->ms_kernels > 0 ? gb / (s->ms_kernels * 1e-3) : 0.0, s->t_wait_reader, s->t_wait_gpu,
                 wall - s->t_wait_reader - s->t_wait_gpu);
      }
      fprintf(stderr, "done in %.3fs\n", wall);
   }
   seeqFree(sq);
   seeqClose(f);
   return EXIT_SUCCESS;
}
