/*
 * oracle/seeq_oracle.c -- CPU restatement of seeq's per-line approximate match.
 * TEST INFRASTRUCTURE ONLY (see seeq_oracle.h).  Parity status: PINNED against
 * the reference's golden vectors and the reference built under oracle/_ref/.
 *
 * Every function cites the reference lines (under /root/reference/src) whose
 * behaviour it restates.  No reference source text is reproduced: the DFA,
 * trie and path codec are replaced by a direct column update per character.
 */
#include "seeq_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* Pattern compiler: libseeq.c:511-603                                       */
/* ------------------------------------------------------------------------ */
int orc_parse(const char *expr, char *keys, int *err)
{
   size_t n = strlen(expr);
   int in_class = 0;      /* libseeq.c:565 'add' */
   int pos = 0;           /* libseeq.c:564 'l'   */
   char last = 0;         /* libseeq.c:566 'lc'  */
   *err = 0;
   memset(keys, 0, n);
   for (size_t i = 0; i < n && (size_t)pos < n; i++) {
      char c = expr[i];
      switch (c) {
      case 'A': case 'a': keys[pos] |= 0x01; break;             /* :568 */
      case 'C': case 'c': keys[pos] |= 0x02; break;             /* :569 */
      case 'G': case 'g': keys[pos] |= 0x04; break;             /* :570 */
      case 'T': case 't': case 'U': case 'u': keys[pos] |= 0x08; break; /* :571 */
      case 'N': case 'n': keys[pos] |= 0x1F; break;             /* :572 */
      case '[':
         if (in_class) { *err = 2; return -1; }                 /* :574-577 */
         in_class = 1;
         break;
      case ']':
         if (!in_class) { *err = 3; return -1; }                /* :581-584 */
         if (last == '[') pos--;         /* empty class adds nothing, :585 */
         in_class = 0;
         break;
      default:
         *err = 4; return -1;                                   /* :588-591 */
      }
      if (!in_class) pos++;                                     /* :593 */
      last = c;
   }
   if (in_class) { *err = 5; return -1; }                       /* :598-601 */
   return pos;
}

/* ------------------------------------------------------------------------ */
/* Byte classes: seeqcore.h:89-111                                           */
/* ------------------------------------------------------------------------ */
int orc_translate(unsigned char b, int convert)
{
   switch (b) {
   case 'A': case 'a': return 0;
   case 'C': case 'c': return 1;
   case 'G': case 'g': return 2;
   case 'T': case 't': case 'U': case 'u': return 3;
   case 'N': case 'n': return 4;
   case 0:    return 5;
   case '\n': return 6;
   default:   return convert ? 4 : 7;
   }
}

/* ------------------------------------------------------------------------ */
/* One column of the saturated edit-distance matrix: libseeq.c:767-789        */
/* col[0..m], col[i] = min(tau+1, D[i][j]); D[0][j] = 0 (free start).         */
/* Initial column (libseeq.c:681-682 + path_to_align :1177-1187):            */
/*   col[i] = min(i, tau+1).                                                 */
/* ------------------------------------------------------------------------ */
static void col_init(int *col, int m, int tau)
{
   for (int i = 0; i <= m; i++) col[i] = i <= tau ? i : tau + 1;
}

static int col_step(int *col, const char *keys, int m, int tau, int code,
                    int *min_to_match)
{
   const int bit = 1 << code;            /* :737 */
   int diag = col[0];                    /* :767 'old' */
   int up = 0;                           /* :768 'prev' */
   int last_active = 1;                  /* :769 (yes, 1) */
   col[0] = 0;
   for (int i = 1; i <= m; i++) {
      int left = col[i];
      int sub = diag + ((bit & keys[i - 1]) == 0);
      int gap = (up < left ? up : left) + 1;
      int v = sub < gap ? sub : gap;
      if (v > tau + 1) v = tau + 1;      /* :781 */
      if (v <= tau) last_active = i;     /* :782 */
      col[i] = v;
      up = v;
      diag = left;
   }
   if (min_to_match) *min_to_match = m - last_active;  /* :789 */
   return up;
}

void orc_trace(const char *data, size_t n, const char *keys, int m, int tau,
               int *dist, int *mtm)
{
   int *col = malloc((size_t)(m + 1) * sizeof(int));
   col_init(col, m, tau);
   for (size_t i = 0; i < n; i++) {
      int c = orc_translate((unsigned char)data[i], 0);
      if (c < 5) dist[i] = col_step(col, keys, m, tau, c, &mtm[i]);
      else { dist[i] = -1; mtm[i] = -1; }
   }
   free(col);
}

/* ------------------------------------------------------------------------ */
/* seeqStringMatch: libseeq.c:171-352                                        */
/* ------------------------------------------------------------------------ */
long orc_string_match(const char *data, const char *keys, int m, int tau,
                      int options, orc_match_t *out, size_t cap)
{
   const int match_opt = options & 0x03;                 /* :219 */
   const int opt_best = match_opt == ORC_BEST;           /* :220 */
   const int all_match = match_opt == ORC_ALL || opt_best; /* :221 */
   const int nondna = options & 0x0C;                    /* :223 */
   const int opt_ignore = nondna == ORC_IGNORE;          /* :224 */
   const int convert = nondna == ORC_CONVERT;            /* :226 */
   const int stream = options & ORC_STREAM;              /* :228 */

   char *rkeys = malloc((size_t)m);
   int *col = malloc((size_t)(m + 1) * sizeof(int));
   int *rcol = malloc((size_t)(m + 1) * sizeof(int));
   if (!rkeys || !col || !rcol) { free(rkeys); free(col); free(rcol); return -1; }
   for (int i = 0; i < m; i++) rkeys[i] = keys[m - 1 - i];   /* :89 */

   size_t hits = 0;
   int best_d = tau + 1;          /* :240 */
   int streak = tau + 1;          /* :242 */
   int latch = 0;                 /* :243 'match' */
   int slen = (int)strlen(data);  /* :245 */
   int end = 0;
   col_init(col, m, tau);

   for (int i = 0; i <= slen; i++) {                     /* :250 */
      int code = orc_translate((unsigned char)data[i], convert);
      int cur = tau + 1;
      int min_to_match = 0;
      if (code < 5) {
         cur = col_step(col, keys, m, tau, code, &min_to_match);  /* :255-264 */
      }
      else if (code == 6 && stream) continue;            /* :265 */
      else if (code == 7 && opt_ignore) continue;        /* :266 */
      else { cur = tau + 1; end = 1; }                   /* :267-270 */

      if (slen - i - 1 < min_to_match) { cur = tau + 1; end = 1; }  /* :272-275 */

      if (streak >= cur) latch = 0;                      /* :278 */
      int perfect = streak == 0;                         /* :286 */
      int stop = streak <= tau && streak < cur;          /* :287 */
      if ((perfect || stop) && !latch && (!opt_best || streak < best_d)) { /* :288 */
         latch = 1;
         /* Reverse scan for the start: :290-316. */
         int j = 0, d = tau + 1, last_d, ignores = 0;
         col_init(rcol, m, tau);
         do {
            int c = orc_translate((unsigned char)data[i - ++j], convert);
            last_d = d;
            if (c < 5) {
               ignores = 0;
               d = col_step(rcol, rkeys, m, tau, c, NULL);
            } else {
               ignores++;      /* 'continue' in a do-while re-tests the condition */
            }
         } while (d > streak && j < i);                  /* :313 */
         j = (last_d < d ? j - 1 : j) - ignores;         /* :315 */
         orc_match_t hit = { (size_t)(i - j), (size_t)i, (size_t)streak };
         if (opt_best) {                                 /* :321-325 */
            hits = 1;
            if (cap > 0) out[0] = hit;
            best_d = streak;
         } else {                                        /* :327 */
            if (hits < cap) out[hits] = hit;
            hits++;
         }
         if (!all_match) end = 1;                        /* :330 */
      }
      if (end) break;                                    /* :334 */
      streak = cur;                                      /* :337 */
   }
   /* Array reversal: :345-349. */
   size_t stored = hits < cap ? hits : cap;
   if (hits <= cap) {
      for (size_t a = 0; a < stored / 2; a++) {
         orc_match_t t = out[a]; out[a] = out[stored - 1 - a]; out[stored - 1 - a] = t;
      }
   }
   free(rkeys); free(col); free(rcol);
   return (long)hits;
}

/* ------------------------------------------------------------------------ */
/* seeqFileMatch's line loop over a memory buffer: seeq.c:361-387             */
/* ------------------------------------------------------------------------ */
long orc_buffer_scan(const char *buf, size_t nbytes, const char *keys, int m,
                     int tau, int options, int fasta,
                     uint64_t *rec, size_t rec_cap,
                     uint32_t *line_nhits, size_t line_cap,
                     uint64_t *nlines, uint64_t *nmatchlines)
{
   size_t pos = 0, line_buf_cap = 256;
   char *line = malloc(line_buf_cap);
   size_t hit_cap = 64;
   orc_match_t *hits = malloc(hit_cap * sizeof(orc_match_t));
   uint64_t lineno = 0, matchlines = 0;
   long nrec = 0;
   while (pos < nbytes) {
      /* getline: up to and including '\n' (seeq.c:361), newline stripped (:364). */
      const char *nl = memchr(buf + pos, '\n', nbytes - pos);
      size_t len = nl ? (size_t)(nl - (buf + pos)) : nbytes - pos;
      if (len + 1 > line_buf_cap) { line_buf_cap = 2 * (len + 1); line = realloc(line, line_buf_cap); }
      memcpy(line, buf + pos, len);
      line[len] = 0;
      pos += len + (nl ? 1 : 0);
      if (fasta && line[0] == '>') continue;             /* :367-374 */
      lineno++;                                          /* :377 */
      long h = orc_string_match(line, keys, m, tau, options, hits, hit_cap);
      if (h < 0) { nrec = -1; break; }
      if ((size_t)h > hit_cap) {
         hit_cap = (size_t)h;
         hits = realloc(hits, hit_cap * sizeof(orc_match_t));
         h = orc_string_match(line, keys, m, tau, options, hits, hit_cap);
      }
      if (line_nhits && lineno - 1 < line_cap) line_nhits[lineno - 1] = (uint32_t)h;
      if (h > 0) matchlines++;
      /* out[] is in sq->match[] order (last hit first); emit left-to-right. */
      for (long k = h - 1; k >= 0; k--) {
         if ((size_t)nrec < rec_cap) {
            rec[4 * nrec + 0] = lineno;
            rec[4 * nrec + 1] = hits[k].start;
            rec[4 * nrec + 2] = hits[k].end;
            rec[4 * nrec + 3] = hits[k].dist;
         }
         nrec++;
      }
   }
   if (nlines) *nlines = lineno;
   if (nmatchlines) *nmatchlines = matchlines;
   free(line); free(hits);
   return nrec;
}

/* ------------------------------------------------------------------------ */
/* Synthetic reads, counter-based.                             */
/* The HIP generator in seeq_amd/csrc/synth.hip implements the same spec;    */
/* tests compare the two byte for byte.                                      */
/* ------------------------------------------------------------------------ */
static uint64_t splitmix64(uint64_t x)
{
   x += 0x9E3779B97F4A7C15ULL;
   uint64_t z = x;
   z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
   z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
   return z ^ (z >> 31);
}

#define SYNTH_MAXP 96

void orc_synth_reads(char *out, uint64_t first, uint64_t n, int len,
                     const char *pattern_plain, int plen, int tau,
                     uint64_t seed)
{
   static const char B[4] = { 'A', 'C', 'G', 'T' };
   for (uint64_t k = 0; k < n; k++) {
      uint64_t r = first + k;
      char *line = out + k * (uint64_t)(len + 1);
      for (int p = 0; p < len; p++)
         line[p] = B[splitmix64(seed ^ (r * 256 + (uint64_t)p)) >> 62];
      line[len] = '\n';
      /* 1/16 of reads carry a mutated copy of the pattern. */
      uint64_t hr = splitmix64(seed ^ 0xA5A5A5A5DEADBEEFULL ^ (r * 0x100000001B3ULL));
      if ((hr & 15) == 0 && plen > 0 && plen + tau + 2 <= SYNTH_MAXP && plen + tau + 2 <= len) {
         char s[SYNTH_MAXP];
         int cur = plen;
         memcpy(s, pattern_plain, (size_t)plen);
         int e = (int)((hr >> 4) % (uint64_t)(tau + 3));     /* 0..tau+2 edits */
         for (int q = 0; q < e; q++) {
            uint64_t hk = splitmix64(hr + (uint64_t)q + 1);
            int type = (int)(hk % 3);
            int pos = (int)((hk >> 8) % (uint64_t)cur);
            char b = B[(hk >> 40) & 3];
            if (type == 0) s[pos] = b;                        /* substitution */
            else if (type == 1) {                             /* insertion */
               for (int t = cur; t > pos; t--) s[t] = s[t - 1];
               s[pos] = b; cur++;
            } else if (cur > 1) {                             /* deletion */
               for (int t = pos; t < cur - 1; t++) s[t] = s[t + 1];
               cur--;
            }
         }
         int off = (int)((hr >> 20) % (uint64_t)(len - cur + 1));
         memcpy(line + off, s, (size_t)cur);
      }
      /* 1/256 of reads carry one N. */
      uint64_t hn = splitmix64(seed ^ 0x5BD1E9955BD1E995ULL ^ (r * 0x9E3779B1ULL));
      if ((hn & 255) == 0) line[(hn >> 8) % (uint64_t)len] = 'N';
   }
}
