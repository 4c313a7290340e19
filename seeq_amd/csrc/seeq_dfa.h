/*
 * seeq_dfa.h -- host-side construction of the automata k_stream walks (seeq_stream.h).
 *
 * (1) The COMPLETE Levenshtein DFA of a pattern.  This is the reference's own idea (a DFA whose states are the
 * saturated Needleman-Wunsch columns, reference libseeq.c:698-842, doc/document.tex:92-136) taken to its conclusion
 * for the GPU: instead of growing the automaton lazily while scanning, the WHOLE reachable automaton is built up
 * front, breadth first, with every accepting state (D[m] <= tau) folded into one absorbing state -- the scan kernel
 * only has to know whether a line ever reaches it.  For the headline pattern (20 positions, tau = 3) that is 3 342
 * states; the table (8 columns x u16 per state) is 53 KB and lives in LDS.
 *
 * (2) A partition FILTER automaton, for patterns whose complete automaton does not fit.  Cut the pattern into k
 * contiguous parts: an occurrence with <= tau errors spends <= floor(tau / k) of them in at least one part (pigeon
 * hole), so it CONTAINS an occurrence of that part with <= t = floor(tau / k) errors.  The filter is the automaton
 * whose state is the tuple of the k parts' saturated columns (threshold t), advanced together; it accepts when any
 * part does.  Reachable tuples are few (configs[4]: 40 positions, tau = 5 -> k = 3, t = 1: 580 states).  Lines it
 * flags are CANDIDATES -- a superset of the hit lines -- and the exact pass verifies every one of them with the
 * bit-vector column; lines it does not flag hold no hit.  `p_accept` (stationary probability that a random DNA
 * character completes a part) tells the caller how many false candidates to expect.
 *
 * Row layout (16 bytes per state, entries are ROW BYTE OFFSETS so that the kernel's address is
 * `state ^ (byte & 0xE)`):   column c = (byte >> 1) & 7
 *      0 'A'  1 'C'  2 'T','U'  3 'G'  5 '\n'  7 'N'     (same for lower case);  4, 6: no DNA byte
 * Bytes that are not DNA alias onto these columns.  Under SQ_FAIL such a byte ends the line
 * (reference libseeq.c:267-270), so whatever the automaton does after it can only ADD spurious hit
 * lines, never lose one: the filter is a superset and the exact pass verifies every flagged line.
 */
#ifndef SEEQ_DFA_H_
#define SEEQ_DFA_H_

#include <stdbool.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef SEEQ_DFA_MAX_STATES
#define SEEQ_DFA_MAX_STATES 4000        /* (states + 3 special rows) * 16 B must fit 16-bit row offsets */
#endif
#define SEEQ_DFA_MAX_PARTS  8

typedef struct {
   uint32_t  nstates;        /* BFS states incl. the absorbing accepting one */
   uint32_t  acc_final;      /* state value of ACC_NEW */
   uint32_t  dead_final;
   uint32_t  final_base;     /* state value of ROOT_NL */
   uint32_t  nrows;          /* nstates + 3 */
   int       nparts;         /* 1 = complete automaton (exact); > 1 = partition filter */
   int       warm;           /* text bytes a walk started at the root needs before its verdicts are those of the line-long walk */
   double    p_accept;       /* filter: stationary probability that a uniformly random A/C/G/T completes a candidate */
   uint16_t *table;          /* nrows * 8 entries */
} seeq_dfa_t;

static inline void seeq_dfa_free(seeq_dfa_t *d) { if (d) { free(d->table); free(d); } }

/* Breadth-first construction of the reachable automaton of `nparts` pattern slices advanced together, each with its
 * own saturated column (threshold t).  keys: one byte per pattern position (bit0 A, bit1 C, bit2 G, bit3 T, N = 0x1F);
 * part p covers positions [cut[p], cut[p+1]).  nparts = 1, t = tau is the complete automaton of the pattern
 * (reference libseeq.c:767-786 per column).  On success returns the number of states n (state 0 = root, state 1 =
 * the absorbing accepting state) and *next_out = n * 5 transitions (classes A C G T N), to be free()d.
 * Returns 0 when the automaton has more than SEEQ_DFA_MAX_STATES states (or on allocation failure). */
/* The general form: a threshold per part (tp[p]), up to SEEQ_MULTI_MAX_PARTS parts (several PATTERNS advanced together:
 * section 4), `cap` states at most; absorb = 1: as above (the first acceptance of any part leads to the absorbing state 1);
 * absorb = 0: the walk goes on after an acceptance -- there is no state 1 then (state numbers are 0 .. n - 1, all real),
 * and *mask_out (n words, to be free()d) holds per state the set of parts whose column ends at or below its threshold there. */
#define SEEQ_MULTI_MAX_PARTS 32
static inline uint32_t seeq_dfa_bfs_parts_ex(const char *keys, const int *cut, int nparts, const int *tp, int absorb, int cap,
                                             uint32_t **next_out, uint32_t **mask_out)
{
   *next_out = NULL;
   if (mask_out) *mask_out = NULL;
   const int m = cut[nparts];
   if (nparts < 1 || nparts > SEEQ_MULTI_MAX_PARTS || m < 1 || m > 2048 || cap < 4) return 0;
   for (int p = 0; p < nparts; p++) if (tp[p] < 0 || cut[p + 1] - cut[p] <= tp[p]) return 0;       /* a part must be longer than its threshold */
   const size_t colsz = (size_t)m + (size_t)nparts;                /* every part has its own row 0 */
   uint8_t *cols = (uint8_t *)malloc((size_t)(cap + 1) * colsz);          /* state -> columns, concatenated */
   uint32_t *next = (uint32_t *)malloc((size_t)(cap + 1) * 5 * sizeof(uint32_t));
   uint32_t HSZ = 16384;                                           /* open addressing, power of two, <= 1/4 full */
   while (HSZ < 4u * (uint32_t)cap) HSZ <<= 1;
   int32_t *hash = (int32_t *)malloc(HSZ * sizeof(int32_t));
   uint8_t *tmp = (uint8_t *)malloc(colsz);
   uint32_t *mask = (!absorb && mask_out) ? (uint32_t *)calloc((size_t)cap + 1, sizeof(uint32_t)) : NULL;
   uint32_t n = absorb ? 2 : 1;
   bool ok = false;
   if (!cols || !next || !hash || !tmp || (!absorb && mask_out && !mask)) goto done;
   memset(hash, 0xFF, HSZ * sizeof(int32_t));
   /* state 0 = root columns min(i, t+1) (reference libseeq.c:681-682); state 1 = ACC (absorbing) */
   {
      size_t o = 0;
      for (int p = 0; p < nparts; p++)
         for (int i = 0; i <= cut[p + 1] - cut[p]; i++) cols[o++] = (uint8_t)(i <= tp[p] ? i : tp[p] + 1);
   }
   if (absorb) memset(cols + colsz, 0xFF, colsz);
   {
      uint32_t h = 2166136261u;
      for (size_t i = 0; i < colsz; i++) h = (h ^ cols[i]) * 16777619u;
      hash[h & (HSZ - 1)] = 0;
   }
   if (absorb) for (int c = 0; c < 5; c++) next[5 + c] = 1;     /* ACC stays ACC */
   for (uint32_t s = 0; s < n; s++) {
      if (absorb && s == 1) continue;
      const uint8_t *col = cols + (size_t)s * colsz;
      for (int c = 0; c < 5; c++) {
         /* one column of the saturated matrix per part: reference libseeq.c:767-786 */
         const int bit = 1 << c;
         bool acc = false;
         uint32_t accmask = 0;
         size_t o = 0;
         for (int p = 0; p < nparts; p++) {
            const int len = cut[p + 1] - cut[p];
            const char *pk = keys + cut[p];
            const int t = tp[p];
            int diag = col[o], up = 0;
            tmp[o] = 0;
            for (int i = 1; i <= len; i++) {
               const int left = col[o + i];
               int v = diag + ((pk[i - 1] & bit) == 0);
               const int g = (up < left ? up : left) + 1;
               if (g < v) v = g;
               if (v > t + 1) v = t + 1;
               tmp[o + i] = (uint8_t)v;
               up = v;
               diag = left;
            }
            if (tmp[o + len] <= t) { acc = true; accmask |= 1u << p; }
            o += (size_t)len + 1;
         }
         uint32_t tgt;
         if (acc && absorb) {
            tgt = 1;                                             /* accepting: absorbed */
         } else {
            uint32_t h = 2166136261u;
            for (size_t i = 0; i < colsz; i++) h = (h ^ tmp[i]) * 16777619u;
            uint32_t slot = h & (HSZ - 1);
            for (;;) {
               const int32_t e = hash[slot];
               if (e < 0) {
                  if ((int)n >= cap) goto done;                  /* too large for the LDS table */
                  memcpy(cols + (size_t)n * colsz, tmp, colsz);
                  hash[slot] = (int32_t)n;
                  if (mask) mask[n] = accmask;                   /* (a function of the columns: the same for every way into the state) */
                  tgt = n++;
                  break;
               }
               if (memcmp(cols + (size_t)e * colsz, tmp, colsz) == 0) { tgt = (uint32_t)e; break; }
               slot = (slot + 1) & (HSZ - 1);
            }
         }
         next[(size_t)s * 5 + c] = tgt;
      }
   }
   ok = true;
done:
   free(cols); free(hash); free(tmp);
   if (!ok) { free(next); free(mask); return 0; }
   *next_out = next;
   if (mask_out) *mask_out = mask; else free(mask);
   return n;
}

static inline uint32_t seeq_dfa_bfs_parts(const char *keys, const int *cut, int nparts, int t, uint32_t **next_out)
{
   int tp[SEEQ_DFA_MAX_PARTS];
   *next_out = NULL;
   if (nparts < 1 || nparts > SEEQ_DFA_MAX_PARTS || cut[nparts] > 62 || t < 0) return 0;
   for (int p = 0; p < nparts; p++) tp[p] = t;
   return seeq_dfa_bfs_parts_ex(keys, cut, nparts, tp, 1, SEEQ_DFA_MAX_STATES, next_out, NULL);
}

static inline uint32_t seeq_dfa_bfs(const char *keys, int m, int tau, uint32_t **next_out)
{
   const int cut[2] = {0, m};
   return seeq_dfa_bfs_parts(keys, cut, 1, tau, next_out);
}

/* Stationary probability that one uniformly random A/C/G/T takes the walk into ACC, the walk restarting at the root
 * after every acceptance (power iteration; the chain forgets its start within a few pattern lengths). */
static inline double seeq_dfa_accept_rate(const uint32_t *next, uint32_t n, int rounds)
{
   double *cur = (double *)calloc(n, sizeof(double)), *nxt = (double *)calloc(n, sizeof(double));
   double rate = 1.0;
   if (cur && nxt) {
      cur[0] = 1.0;
      for (int r = 0; r < rounds; r++) {
         memset(nxt, 0, n * sizeof(double));
         double acc = 0.0;
         for (uint32_t s = 0; s < n; s++) {
            if (s == 1 || cur[s] == 0.0) continue;
            const double q = cur[s] * 0.25;
            for (int c = 0; c < 4; c++) {
               const uint32_t tgt = next[(size_t)s * 5 + c];
               if (tgt == 1) acc += q; else nxt[tgt] += q;
            }
         }
         nxt[0] += acc;
         rate = acc;
         double *sw = cur; cur = nxt; nxt = sw;
      }
   }
   free(cur); free(nxt);
   return rate;
}

static const int seeq_dfa_colof[5] = {0, 1, 3, 2, 7};           /* table column of a class: A C G T N */

/* Streaming automaton (k_stream): runs across line ends.  Special rows, as row byte offsets:
 *    16           ACC_OLD   the line already has a hit (absorbing until the line ends)
 *    acc_final    ACC_NEW   same row, entered by the transition that completes the FIRST hit of the line
 *    dead_final   DEAD      a non-DNA byte ended the line (SQ_FAIL, reference libseeq.c:267-270); waits for '\n'
 *    final_base+32 ROOT_NL  copy of the root row: the state right after a '\n'
 * so "state == ACC_NEW" marks the text position where a line gets its first hit. */
static inline seeq_dfa_t *seeq_dfa_layout_stream(uint32_t *next, uint32_t n)
{
   seeq_dfa_t *d = (seeq_dfa_t *)calloc(1, sizeof *d);
   if (d) {
      d->nstates = n;
      d->nrows = n + 3;
      d->final_base = n * 16;
      d->acc_final = n * 16;                                      /* ACC_NEW */
      d->dead_final = (n + 1) * 16;                               /* DEAD */
      d->table = (uint16_t *)malloc((size_t)d->nrows * 8 * sizeof(uint16_t));
      if (!d->table) { free(d); d = NULL; }
   }
   if (d) {
      /* Bank spreading: the DNA columns are the first 8 bytes of a row, i.e. 2 of the 4 LDS banks a row
         covers (32 banks x 4 bytes: 8 rows per bank cycle).  Every second group of 8 rows is stored rotated
         by 8 bytes and the state VALUE of such a row carries bit 3, so that the kernel's address
         "state ^ column" lands in the other half: 16 consecutive rows put their DNA entries in 32 different
         banks.  (Measured / simulated: 8.0 -> 5.6 LDS cycles per 64-lane gather; profiles/microbench.) */
#define SEEQ_ROT(row) ((((row) >> 3) & 1) != 0)
#define SEEQ_SV(row) ((uint16_t)((row) * 16 + (SEEQ_ROT(row) ? 8 : 0)))
      const uint32_t r_accnew = n, r_dead = n + 1, r_rootnl = n + 2;
      const uint16_t acc_new = SEEQ_SV(r_accnew), dead = SEEQ_SV(r_dead), root_nl = SEEQ_SV(r_rootnl);
      uint16_t lrow[8];                                           /* logical row: entry per column */
      for (uint32_t s = 0; s < d->nrows; s++) {
         const uint32_t src = s == r_accnew ? 1 : (s == r_rootnl ? 0 : s);     /* ACC_NEW = ACC_OLD row, ROOT_NL = root row */
         for (int k = 0; k < 8; k++) lrow[k] = dead;              /* columns no DNA byte maps to */
         lrow[5] = root_nl;                                       /* '\n' */
         if (s != r_dead) {
            for (int c = 0; c < 5; c++) {
               const uint32_t t = next[(size_t)src * 5 + c];
               lrow[seeq_dfa_colof[c]] = t == 1 ? (src == 1 ? SEEQ_SV(1) : acc_new) : SEEQ_SV(t);
            }
         }
         uint16_t *row = d->table + (size_t)s * 8;
         for (int k = 0; k < 8; k++) row[k ^ (SEEQ_ROT(s) ? 4 : 0)] = lrow[k];    /* 8-byte rotation = column ^ 4 */
      }
      d->acc_final = acc_new;                                     /* state VALUES (rotation bit included) */
      d->dead_final = dead;
      d->final_base = root_nl;
#undef SEEQ_SV
#undef SEEQ_ROT
   }
   free(next);
   return d;
}

/* The SKIP variant of a laid-out table (k_stream under SQ_IGNORE, reference libseeq.c:265-266: a byte that is neither a
 * base nor a terminator is skipped): column 4 -- no DNA byte maps to it; k_stream replaces every byte to skip by 'H' --
 * leads every state onto itself (ACC_NEW onto ACC_OLD: the line's first hit is reported once).  Same rows, same state
 * values as the table it is made from.  Returns a malloc'ed copy of the table, or NULL. */
static inline uint16_t *seeq_dfa_skip_variant(const seeq_dfa_t *d)
{
   uint16_t *t = (uint16_t *)malloc((size_t)d->nrows * 8 * sizeof(uint16_t));
   if (!t) return NULL;
   memcpy(t, d->table, (size_t)d->nrows * 8 * sizeof(uint16_t));
   for (uint32_t s = 0; s < d->nrows; s++) {
      const int rot = (int)((s >> 3) & 1);
      const uint32_t self = s == d->nstates ? 1u : s;                  /* row nstates is ACC_NEW: on to ACC_OLD (row 1) */
      t[(size_t)s * 8 + (size_t)(4 ^ (rot ? 4 : 0))] = (uint16_t)(self * 16 + (((self >> 3) & 1) ? 8 : 0));
   }
   return t;
}

/* The RESTART variant of a laid-out FILTER table (k_stream's long-line variant, round 5): acceptance does not absorb -- the row of ACC_NEW
 * (row nstates: entered by the transition that completes a part occurrence, which is what the kernel flags) is a copy of the ROOT's row, so
 * the walk goes on from the root and flags EVERY part occurrence, as k_pair's automata do (seeq_pair.h): the exact pass then scans
 * m + tau columns either side of a candidate instead of the rest of its chunk.  Same rows, same state values.  malloc'ed copy, or NULL. */
static inline uint16_t *seeq_dfa_restart_variant(const seeq_dfa_t *d)
{
   uint16_t *t = (uint16_t *)malloc((size_t)d->nrows * 8 * sizeof(uint16_t));
   if (!t) return NULL;
   memcpy(t, d->table, (size_t)d->nrows * 8 * sizeof(uint16_t));
   const uint32_t r = d->nstates;                                     /* ACC_NEW */
   const int rot_r = (int)((r >> 3) & 1);                             /* (the root, row 0, is not rotated) */
   for (int k = 0; k < 8; k++) t[(size_t)r * 8 + (size_t)(k ^ (rot_r ? 4 : 0))] = d->table[k];
   return t;
}

/* The complete automaton of the pattern, or NULL when it has more than SEEQ_DFA_MAX_STATES states. */
static inline seeq_dfa_t *seeq_dfa_build_stream(const char *keys, int m, int tau)
{
   uint32_t *next = NULL;
   const uint32_t n = seeq_dfa_bfs(keys, m, tau, &next);
   if (!n) return NULL;
   seeq_dfa_t *d = seeq_dfa_layout_stream(next, n);
   if (d) { d->nparts = 1; d->warm = m + tau - 1; d->p_accept = 0.0; }
   return d;
}

/* The filter of k equal parts (sizes differ by at most one), threshold floor(tau / k); NULL when it does not fit. */
static inline seeq_dfa_t *seeq_dfa_build_filter(const char *keys, int m, int tau, int k)
{
   int cut[SEEQ_DFA_MAX_PARTS + 1];
   if (k < 2 || k > SEEQ_DFA_MAX_PARTS || k > m) return NULL;
   for (int p = 0; p <= k; p++) cut[p] = (int)((long)p * m / k);
   const int t = tau / k;
   int longest = 0;
   for (int p = 0; p < k; p++) if (cut[p + 1] - cut[p] > longest) longest = cut[p + 1] - cut[p];
   uint32_t *next = NULL;
   const uint32_t n = seeq_dfa_bfs_parts(keys, cut, k, t, &next);
   if (!n) return NULL;
   const double rate = seeq_dfa_accept_rate(next, n, 4 * m + 64);
   seeq_dfa_t *d = seeq_dfa_layout_stream(next, n);
   if (d) { d->nparts = k; d->warm = longest + t - 1; d->p_accept = rate; }
   return d;
}

/* What k_stream should walk for this pattern: the complete automaton when it fits and warms up within 32 bytes
 * (exact verdicts), else the filter with the fewest parts that does (most selective first); NULL: none. */
static inline seeq_dfa_t *seeq_dfa_plan_stream(const char *keys, int m, int tau, int complete_only)
{
   if (m + tau - 1 <= 32) {
      seeq_dfa_t *d = seeq_dfa_build_stream(keys, m, tau);
      if (d) return d;
   }
   if (complete_only) return NULL;
   for (int k = 2; k <= SEEQ_DFA_MAX_PARTS && k <= tau + 1; k++) {
      seeq_dfa_t *d = seeq_dfa_build_filter(keys, m, tau, k);
      if (!d) continue;
      if (d->warm <= 32) return d;
      seeq_dfa_free(d);
   }
   return NULL;
}

/* ======================================================================================================================
 * (3) The PAIR automaton (k_pair, seeq_pair.h): two text bytes per table step.
 *
 * k_stream pays one LDS gather per text byte (1.375 with its warm-up), and that gather unit is what holds it.  A table
 * indexed by (state, PAIR of bases) halves the gathers -- 16 columns of u16 per state, 32 bytes a row -- but row offsets
 * must stay 16 bit: <= 2 047 rows.  So the walk does not carry the whole pattern but its longest PREFIX whose automaton
 * fits (headline pattern: 17 of 20 positions, 1 840 states, 59 KB): an occurrence of the pattern with <= tau errors
 * starts with an occurrence of the prefix with <= tau errors, so every hit line is flagged; the lines flagged are
 * CANDIDATES (random text completes such a prefix once in 10^5 bytes) and the exact pass verifies every one -- which it
 * does anyway whenever records are wanted.
 *
 * The text is taken as 2-bit codes, bits 1-2 of the byte: A 0, C 1, T/U 2, G 3 in either case.  Every other byte
 * aliases onto one of them ('\n' -> C, N -> G): an alias can only turn a mismatch into a match, so the walk stays a
 * superset; and there is no newline column -- the walk runs across line ends (an occurrence that spans one is one more
 * false candidate, at the start of a line).  Acceptance RESTARTS the walk at the root instead of entering an absorbing
 * state (which only a newline could leave): the transition that completes an occurrence leads to a flagged copy of the
 * root row (hit at the second byte of the pair) or of the row the root reaches over the second byte (hit at the first).
 * Flagged rows have ODD state values (they are stored one byte late): bit 0 of the state marks a pair in which the walk
 * accepted -- one v_alignbit per step shifts it into the chain's mask; position = the pair's second byte.
 * Superset at line level, chunked walks with a warm-up of mp + tau - 1 bytes included: see seeq_pair.h.
 *
 * The automaton is minimised first (partition refinement on {hit flags, successor classes} per base): the saturated
 * columns are not a minimal automaton -- headline pattern 3 342 -> 3 198 states over A C G T -- which buys a longer
 * prefix now and then.
 */
#define SEEQ_PAIR_MAX_ROWS 2047         /* row byte offsets (32 B rows) in 16 bits */

typedef struct {
   uint32_t  nstates;        /* walk states after minimisation (accepting columns are not states: restart) */
   uint32_t  nstates_raw;    /* reachable non-accepting columns before minimisation */
   uint32_t  nrows;          /* nstates + 5 flagged rows */
   uint32_t  table_bytes;    /* nrows * 32 + 16 (the flagged rows sit one byte late), a multiple of 16 */
   int       mp;             /* positions of the pattern the walk carries (prefix) */
   int       nparts;         /* 1: a prefix with the pattern's own threshold; > 1: partition filter (threshold floor(tau / nparts) per part) */
   int       warm;           /* text bytes a walk started at the root needs before it sees what the line-long walk sees */
   double    p_accept;       /* probability that a uniformly random A/C/G/T completes a candidate */
   uint8_t  *table;          /* state VALUE = byte offset of the state's row: 32 * row for a walk state, 32 * row + 1 for a
                                flagged row (bit 0 of a state value IS the flag; 16-bit reads at odd LDS addresses are
                                byte-exact on gfx950: profiles/microbench/lds_unaligned_u16.hip).  The u16 at
                                table[value + 2 * (4 * first_code + second_code)] is the next state value. */
} seeq_pair_t;

static inline void seeq_pair_free(seeq_pair_t *d) { if (d) { free(d->table); free(d); } }

static const int seeq_pair_class_of_code[4] = {0, 1, 3, 2};     /* 2-bit text code (A C T G) -> class of seeq_dfa_bfs (A C G T) */

/* Mealy minimisation of the restart automaton over the four bases.  next: n * 5 (seeq_dfa_bfs; state 1 = ACC).  Writes
 * cls[s] (class of state s; cls[1] is meaningless) and returns the number of classes of the states != 1; the root's
 * class is 0. */
static inline uint32_t seeq_dfa_minimise_restart(const uint32_t *next, uint32_t n, uint32_t *cls)
{
   uint32_t *nc = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
   uint32_t HSZ = 1024;
   while (HSZ < 4u * n) HSZ <<= 1;
   int32_t *hash = (int32_t *)malloc((size_t)HSZ * sizeof(int32_t));
   uint32_t *sig = (uint32_t *)malloc((size_t)n * 5 * sizeof(uint32_t));
   uint32_t ncls = 1;
   if (!nc || !hash || !sig) { free(nc); free(hash); free(sig); for (uint32_t s = 0; s < n; s++) cls[s] = s; return n; }
   for (uint32_t s = 0; s < n; s++) cls[s] = 0;
   for (;;) {
      /* signature of s: its class, and per base the class of the target with the hit flag in bit 31 */
      for (uint32_t s = 0; s < n; s++) {
         uint32_t *g = sig + (size_t)s * 5;
         g[0] = cls[s];
         for (int c = 0; c < 4; c++) {
            const uint32_t t = next[(size_t)s * 5 + seeq_pair_class_of_code[c]];
            g[1 + c] = t == 1 ? (0x80000000u | cls[0]) : cls[t];
         }
      }
      memset(hash, 0xFF, (size_t)HSZ * sizeof(int32_t));
      uint32_t k = 0;
      for (uint32_t s = 0; s < n; s++) {                  /* classes numbered in order of first appearance: the root stays 0 */
         if (s == 1) { nc[s] = 0; continue; }
         const uint32_t *g = sig + (size_t)s * 5;
         uint32_t h = 2166136261u;
         for (int i = 0; i < 5; i++) h = (h ^ g[i]) * 16777619u;
         uint32_t slot = h & (HSZ - 1);
         for (;;) {
            const int32_t e = hash[slot];
            if (e < 0) { hash[slot] = (int32_t)s; nc[s] = k++; break; }
            if (memcmp(sig + (size_t)e * 5, g, 5 * sizeof(uint32_t)) == 0) { nc[s] = nc[e]; break; }
            slot = (slot + 1) & (HSZ - 1);
         }
      }
      memcpy(cls, nc, (size_t)n * sizeof(uint32_t));
      if (k == ncls) break;
      ncls = k;
   }
   free(nc); free(hash); free(sig);
   return ncls;
}

/* The pair table of a restart automaton given as seeq_dfa_bfs_parts() output (n states, state 1 = ACC; consumed), or
 * NULL when it has more than SEEQ_PAIR_MAX_ROWS rows after minimisation.  `shortest` = the shortest part: the byte after
 * a restart must not accept on its own, i.e. shortest - t >= 2. */
static inline seeq_pair_t *seeq_pair_from_next(uint32_t *next, uint32_t n, int rounds)
{
   seeq_pair_t *d = NULL;
   uint32_t *cls = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
   uint32_t *rep = NULL;
   if (cls) {
      const uint32_t k = seeq_dfa_minimise_restart(next, n, cls);
      if (k + 5 <= SEEQ_PAIR_MAX_ROWS) {
         d = (seeq_pair_t *)calloc(1, sizeof *d);
         rep = (uint32_t *)malloc((size_t)k * sizeof(uint32_t));          /* class -> a state of it */
         if (d) { d->table_bytes = (k + 5) * 32 + 16; d->table = (uint8_t *)calloc(d->table_bytes, 1); }
         if (!d || !rep || !d->table) { if (d) free(d->table); free(d); d = NULL; }
      }
      if (d) {
         d->nstates = k; d->nstates_raw = n - 1; d->nrows = k + 5;
         d->p_accept = seeq_dfa_accept_rate(next, n, rounds);
         for (uint32_t s = n; s-- > 0;) if (s != 1) rep[cls[s]] = s;
         /* one step of the restart automaton from class q over code c: *hit set when it accepts (then back at the root) */
#define SEEQ_PAIR_STEP(q, c, hit) (next[(size_t)rep[q] * 5 + seeq_pair_class_of_code[c]] == 1 ? ((hit) = 1, 0u) : cls[next[(size_t)rep[q] * 5 + seeq_pair_class_of_code[c]]])
#define SEEQ_PAIR_VALUE(row) ((row) < k ? (row) * 32u : (row) * 32u + 1u)
         for (uint32_t r = 0; r < d->nrows; r++) {
            /* rows k .. k+3: flagged copies of the row the root reaches over code r - k; row k+4: flagged copy of the root row */
            int dummy = 0;
            const uint32_t q = r < k ? r : (r < k + 4 ? SEEQ_PAIR_STEP(0u, (int)(r - k), dummy) : 0u);
            (void)dummy;                                     /* (every part is >= t + 2 long: one byte from the root never accepts) */
            for (int c1 = 0; c1 < 4; c1++)
               for (int c2 = 0; c2 < 4; c2++) {
                  int h1 = 0, h2 = 0;
                  const uint32_t s1 = SEEQ_PAIR_STEP(q, c1, h1);
                  const uint32_t s2 = SEEQ_PAIR_STEP(s1, c2, h2);
                  const uint32_t row = h2 ? k + 4 : (h1 ? k + (uint32_t)c2 : s2);
                  const uint16_t val = (uint16_t)SEEQ_PAIR_VALUE(row);
                  memcpy(d->table + SEEQ_PAIR_VALUE(r) + 2u * (uint32_t)(c1 * 4 + c2), &val, 2);
               }
         }
#undef SEEQ_PAIR_VALUE
#undef SEEQ_PAIR_STEP
      }
   }
   free(rep); free(cls); free(next);
   return d;
}

/* The pair automaton of the first `mp` positions (threshold tau), or NULL when it does not fit. */
static inline seeq_pair_t *seeq_pair_build_prefix(const char *keys, int mp, int tau)
{
   if (mp < tau + 2) return NULL;
   uint32_t *next = NULL;
   const uint32_t n = seeq_dfa_bfs(keys, mp, tau, &next);
   if (!n) return NULL;
   seeq_pair_t *d = seeq_pair_from_next(next, n, 4 * mp + 64);
   if (d) { d->mp = mp; d->nparts = 1; d->warm = mp + tau - 1; }
   return d;
}

/* The pair automaton of the partition filter of k parts (seeq_dfa_build_filter), or NULL. */
static inline seeq_pair_t *seeq_pair_build_filter(const char *keys, int m, int tau, int k)
{
   int cut[SEEQ_DFA_MAX_PARTS + 1];
   if (k < 2 || k > SEEQ_DFA_MAX_PARTS || k > m) return NULL;
   for (int p = 0; p <= k; p++) cut[p] = (int)((long)p * m / k);
   const int t = tau / k;
   int longest = 0, shortest = m;
   for (int p = 0; p < k; p++) {
      const int len = cut[p + 1] - cut[p];
      if (len > longest) longest = len;
      if (len < shortest) shortest = len;
   }
   if (shortest < t + 2) return NULL;
   uint32_t *next = NULL;
   const uint32_t n = seeq_dfa_bfs_parts(keys, cut, k, t, &next);
   if (!n) return NULL;
   seeq_pair_t *d = seeq_pair_from_next(next, n, 4 * m + 64);
   if (d) { d->mp = m; d->nparts = k; d->warm = longest + t - 1; }
   return d;
}

/* What k_pair should walk for this pattern: among the longest prefix (>= tau + 2 positions) whose pair automaton fits
 * and the partition filters of 2 .. tau + 1 parts, the one that makes the fewest false candidates, warm-up within 32
 * bytes; NULL: none. */
static inline seeq_pair_t *seeq_pair_plan(const char *keys, int m, int tau)
{
   seeq_pair_t *best = NULL;
   if (m > 62) return NULL;
   for (int mp = tau + 2; mp <= m && mp + tau - 1 <= 32; mp++) {
      seeq_pair_t *d = seeq_pair_build_prefix(keys, mp, tau);
      if (!d) break;                                        /* (sizes grow with the prefix: the first that does not fit ends the search) */
      seeq_pair_free(best);
      best = d;
   }
   for (int k = 2; k <= SEEQ_DFA_MAX_PARTS && k <= tau + 1; k++) {
      seeq_pair_t *d = seeq_pair_build_filter(keys, m, tau, k);
      if (!d) continue;
      if (d->warm <= 32 && (!best || d->p_accept < best->p_accept)) { seeq_pair_free(best); best = d; }
      else seeq_pair_free(d);
   }
   return best;
}

/* ======================================================================================================================
 * (3b) The QUAD automaton (k_packed_walk<true>, seeq_packed.h): FOUR bases per table step -- one byte of a packed read.
 *
 * The packed walk is held by the LDS gather unit alone (no warm-up, no checks, no newlines: 2.25 VALU per gather), so only
 * fewer gathers per base move it.  A table indexed by (state, packed byte) has 256 columns of u16 per state, 512 bytes a row:
 * a small automaton only -- at most 127 rows (64 KB).  The longest prefix that fits in so few states is useless (8 positions
 * of the headline pattern: a candidate in every read), but a PARTITION FILTER is small by nature: the headline 20-mer at
 * distance 3 as two 10-mers with one error each has 95 states and completes by chance once in 8 600 positions (one read in
 * 57 becomes a false candidate; the pair automaton's 17-position prefix: one in 700) -- the exact pass verifies 30 % more
 * windows and the walk makes half the gathers.
 *
 * Same restart automaton as the pair table (accepting = back at the root, any part), composed four bases at a time.  An entry is
 *    bits 9-15  the row of the state after the four bases (its byte offset: & 0xFE00)
 *    bits 0-3   bit i: the walk accepted on base i of the four
 * so one v_alignbit_b32(entry, mask, 4) per step shifts the four position flags into a per-base mask (exact positions: the pair
 * table knows the pair only), and one v_bfi_b32 puts the next index under the row offset.
 * ====================================================================================================================== */
#define SEEQ_QUAD_MAX_ROWS 127

typedef struct {
   uint32_t  nstates;        /* rows: walk states after minimisation */
   uint32_t  nstates_raw;
   uint32_t  table_bytes;    /* nstates * 512 */
   int       mp, nparts;     /* as seeq_pair_t */
   int       warm;           /* text bytes a walk started at the root needs before it sees what the line-long walk sees (the chunked ASCII walk) */
   double    p_accept;
   uint16_t *table;          /* [nstates][256]: index = the packed byte (first base in bits 7-6; codes A 0, C 1, T 2, G 3) */
} seeq_quad_t;

static inline void seeq_quad_free(seeq_quad_t *d) { if (d) { free(d->table); free(d); } }

/* The quad table of a restart automaton given as seeq_dfa_bfs_parts() output (consumed), or NULL when it has more than
 * SEEQ_QUAD_MAX_ROWS states after minimisation. */
static inline seeq_quad_t *seeq_quad_from_next(uint32_t *next, uint32_t n, int rounds)
{
   seeq_quad_t *d = NULL;
   uint32_t *cls = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
   uint32_t *rep = NULL;
   if (cls) {
      const uint32_t k = seeq_dfa_minimise_restart(next, n, cls);
      if (k <= SEEQ_QUAD_MAX_ROWS) {
         d = (seeq_quad_t *)calloc(1, sizeof *d);
         rep = (uint32_t *)malloc((size_t)k * sizeof(uint32_t));
         if (d) { d->table_bytes = k * 512u; d->table = (uint16_t *)calloc(d->table_bytes, 1); }
         if (!d || !rep || !d->table) { if (d) free(d->table); free(d); d = NULL; }
      }
      if (d) {
         d->nstates = k; d->nstates_raw = n - 1;
         d->p_accept = seeq_dfa_accept_rate(next, n, rounds);
         for (uint32_t s = n; s-- > 0;) if (s != 1) rep[cls[s]] = s;
         for (uint32_t r = 0; r < k; r++)
            for (uint32_t b = 0; b < 256; b++) {
               uint32_t q = r, flags = 0;
               for (int i = 0; i < 4; i++) {
                  const int c = (int)((b >> (6 - 2 * i)) & 3u);
                  const uint32_t t = next[(size_t)rep[q] * 5 + seeq_pair_class_of_code[c]];
                  if (t == 1) { flags |= 1u << i; q = 0; }     /* accepted: back at the root */
                  else q = cls[t];
               }
               d->table[(size_t)r * 256 + b] = (uint16_t)((q << 9) | flags);
            }
      }
   }
   free(rep); free(cls); free(next);
   return d;
}

/* What the quad walk should carry for this pattern: among the prefixes (>= tau + 2 positions) and the partition filters of
 * 2 .. tau + 1 parts whose automata fit, the one that makes the fewest false candidates; NULL: none fits. */
static inline seeq_quad_t *seeq_quad_plan(const char *keys, int m, int tau)
{
   seeq_quad_t *best = NULL;
   if (m > 62) return NULL;
   for (int mp = tau + 2; mp <= m; mp++) {
      uint32_t *next = NULL;
      const uint32_t n = seeq_dfa_bfs(keys, mp, tau, &next);
      if (!n) break;
      seeq_quad_t *d = seeq_quad_from_next(next, n, 4 * mp + 64);
      if (!d) break;                                        /* (sizes grow with the prefix) */
      d->mp = mp; d->nparts = 1; d->warm = mp + tau - 1;
      seeq_quad_free(best);
      best = d;
   }
   for (int k = 2; k <= SEEQ_DFA_MAX_PARTS && k <= tau + 1 && k <= m; k++) {
      int cut[SEEQ_DFA_MAX_PARTS + 1];
      for (int p = 0; p <= k; p++) cut[p] = (int)((long)p * m / k);
      const int t = tau / k;
      int shortest = m, longest = 0;
      for (int p = 0; p < k; p++) {
         if (cut[p + 1] - cut[p] < shortest) shortest = cut[p + 1] - cut[p];
         if (cut[p + 1] - cut[p] > longest) longest = cut[p + 1] - cut[p];
      }
      if (shortest < t + 2) continue;
      uint32_t *next = NULL;
      const uint32_t n = seeq_dfa_bfs_parts(keys, cut, k, t, &next);
      if (!n) continue;
      seeq_quad_t *d = seeq_quad_from_next(next, n, 4 * m + 64);
      if (!d) continue;
      d->mp = m; d->nparts = k; d->warm = longest + t - 1;
      if (!best || d->p_accept < best->p_accept) { seeq_quad_free(best); best = d; }
      else seeq_quad_free(d);
   }
   return best;
}

/* ==========================================================================================================================
 * (4) SEVERAL PATTERNS, ONE WALK (barcode sets; reference doc/response.tex:358-360 names the multi-pattern search as the
 *     place where parallel work exists).  Two automata over the same prefixes lp[p] <= m[p] of the patterns:
 *
 *   pair    the UNION walk of k_pair: all prefixes advanced together (parts = patterns, each with its own threshold),
 *           restart at the root when ANY of them accepts, minimised, two bases per step.  Superset per pattern as in
 *           seeq_pair.h: every occurrence of pattern p (<= tau[p] errors) holds an occurrence [s, j] of its prefix, the
 *           chain flags j unless it restarted (for whatever pattern) at j1 in [s, j) -- then j1 is flagged.  So every
 *           occurrence of every pattern has a candidate inside it or on the byte after it, the first candidate c of a
 *           line is <= start + maxspan for every occurrence, the last one l >= start + 1: every occurrence of every
 *           pattern lies in [c - maxspan, l + maxspan], maxspan = max(m + tau).
 *   res     the same prefixes advanced together WITHOUT restart or absorption, byte by byte (classes A C G T N): per state
 *           the set of patterns whose prefix has an alignment within its threshold ending there.  Walked from the root
 *           over the window above it sees every such alignment of the window -- the union of the masks is a superset of
 *           the patterns that occur in the line, and (prefix = pattern) exactly that set.
 *
 * The exact pass then verifies (line, pattern) pairs only.  NULL when a pattern is shorter than tau + 2, the union does not
 * fit the pair table with prefixes of at least min(m, tau + 5) positions, or `res` needs more than 65 535 states.
 * ========================================================================================================================== */
typedef struct {
   int          npat;
   int          m[SEEQ_MULTI_MAX_PARTS], tau[SEEQ_MULTI_MAX_PARTS], lp[SEEQ_MULTI_MAX_PARTS];
   int          maxspan;        /* max over patterns of m + tau */
   seeq_pair_t *pair;           /* mp = longest prefix, warm = max(lp + tau - 1) */
   uint32_t     res_states;
   uint16_t    *res_next;       /* res_states * 8: classes A C G T N, then three unused entries (a 16-byte row) */
   uint32_t    *res_mask;       /* res_states */
   int          res_exact;      /* every prefix is its whole pattern */
} seeq_multi_t;

static inline void seeq_multi_free(seeq_multi_t *d)
{
   if (d) { seeq_pair_free(d->pair); free(d->res_next); free(d->res_mask); free(d); }
}

/* keys[p]: m[p] key bytes of pattern p (bit0 A .. bit3 T, N = 0x1F). */
static inline seeq_multi_t *seeq_multi_build(const char *const *keys, const int *m, const int *tau, int npat)
{
   if (npat < 1 || npat > SEEQ_MULTI_MAX_PARTS) return NULL;
   int maxm = 0, total = 0;
   for (int p = 0; p < npat; p++) {
      if (m[p] < tau[p] + 2 || m[p] > 62 || tau[p] < 0) return NULL;
      if (m[p] > maxm) maxm = m[p];
      total += m[p];
   }
   seeq_multi_t *d = (seeq_multi_t *)calloc(1, sizeof *d);
   char *cat = (char *)malloc((size_t)total);
   int cut[SEEQ_MULTI_MAX_PARTS + 1];
   if (!d || !cat) { free(d); free(cat); return NULL; }
   d->npat = npat;
   for (int p = 0; p < npat; p++) { d->m[p] = m[p]; d->tau[p] = tau[p]; if (m[p] + tau[p] > d->maxspan) d->maxspan = m[p] + tau[p]; }
   for (int cap_len = maxm; cap_len >= 3 && !d->pair; cap_len--) {
      int o = 0, warm = 0, longest = 0, ok = 1;
      for (int p = 0; p < npat; p++) {
         int lp = m[p] < cap_len ? m[p] : cap_len;
         const int least = m[p] < tau[p] + 5 ? m[p] : tau[p] + 5;       /* shorter prefixes flag most of a random text */
         if (lp < least) lp = least;
         if (lp < tau[p] + 2) ok = 0;
         d->lp[p] = lp;
         cut[p] = o;
         memcpy(cat + o, keys[p], (size_t)lp);
         o += lp;
         if (lp + tau[p] - 1 > warm) warm = lp + tau[p] - 1;
         if (lp > longest) longest = lp;
      }
      cut[npat] = o;
      if (!ok) break;
      if (warm <= 32) {
         uint32_t *next = NULL;
         const uint32_t n = seeq_dfa_bfs_parts_ex(cat, cut, npat, tau, 1, 24000, &next, NULL);
         if (n) {
            seeq_pair_t *pr = seeq_pair_from_next(next, n, 4 * longest + 64);      /* (consumes next) */
            if (pr) { pr->mp = longest; pr->nparts = npat; pr->warm = warm; d->pair = pr; }
         }
      }
      /* every prefix at its floor already: shorter caps change nothing */
      int at_floor = 1;
      for (int p = 0; p < npat; p++) { const int least = m[p] < tau[p] + 5 ? m[p] : tau[p] + 5; if (d->lp[p] > least) at_floor = 0; }
      if (!d->pair && at_floor) break;
   }
   if (d->pair) {
      /* the resolve automaton is not bound by the pair table's 2 047 rows: whole patterns when that stays below 65 536
         states (its sets are exact then), else the prefixes of the union walk */
      uint32_t *next = NULL, *mask = NULL;
      uint32_t n = 0;
      int whole = 1;
      for (int p = 0; p < npat; p++) if (d->lp[p] != m[p]) whole = 0;
      if (!whole) {
         char *full = (char *)malloc((size_t)total);
         int fcut[SEEQ_MULTI_MAX_PARTS + 1];
         if (full) {
            int o = 0;
            for (int p = 0; p < npat; p++) { fcut[p] = o; memcpy(full + o, keys[p], (size_t)m[p]); o += m[p]; }
            fcut[npat] = o;
            n = seeq_dfa_bfs_parts_ex(full, fcut, npat, tau, 0, 65535, &next, &mask);
            if (n) whole = 1;
            free(full);
         }
      }
      if (!n) n = seeq_dfa_bfs_parts_ex(cat, cut, npat, tau, 0, 65535, &next, &mask);
      if (n) {
         d->res_next = (uint16_t *)calloc((size_t)n * 8, sizeof(uint16_t));
         if (d->res_next) {
            for (uint32_t q = 0; q < n; q++) for (int c = 0; c < 5; c++) d->res_next[(size_t)q * 8 + c] = (uint16_t)next[(size_t)q * 5 + c];
            d->res_states = n; d->res_mask = mask; mask = NULL;
            d->res_exact = whole;
         }
      }
      free(next); free(mask);
   }
   free(cat);
   if (!d->pair || !d->res_next) { seeq_multi_free(d); return NULL; }
   return d;
}

#endif
