/*
 * oracle/seeq_oracle.h -- CPU restatement of the seeq per-line approximate
 * match path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity checker for the HIP path.  Nothing in the product
 * (seeq_amd/, include/) may include, link or call it; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement
 * against the reference's own known answers (test/testset.c) and
 * tests/test_oracle_vs_ref.py fuzzes it against the reference itself
 * (oracle/_ref/libseeq_ref.so, built from /root/reference by oracle/Makefile).
 *
 * The algorithm here is deliberately NOT the one the GPU uses: it is the
 * reference's saturated Needleman-Wunsch column update (libseeq.c:767-786)
 * applied at every character, without the DFA/trie memoisation
 * (libseeq.c:606-695, 845-1241), which has no observable effect on results.
 */
#ifndef SEEQ_ORACLE_H_
#define SEEQ_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Option bits: same values as the reference (libseeq.h:34-48). */
#define ORC_FIRST   0x00
#define ORC_BEST    0x01
#define ORC_ALL     0x02
#define ORC_COUNT   0x03
#define ORC_FAIL    0x00
#define ORC_CONVERT 0x04
#define ORC_IGNORE  0x08
#define ORC_LINES   0x00
#define ORC_STREAM  0x10

typedef struct {
   size_t start;
   size_t end;   /* exclusive */
   size_t dist;
} orc_match_t;

/* libseeq.c:511-603.  keys must hold strlen(expr) bytes.  Returns the number
 * of pattern positions or -1 with *err set to the reference's seeqerr code. */
int  orc_parse(const char *expr, char *keys, int *err);

/* Byte -> code (seeqcore.h:89-111).  convert != 0 selects translate_convert.
 * Bytes >= 0x80 are defined as non-DNA (the reference indexes the table with
 * a signed char there: undefined behaviour). */
int  orc_translate(unsigned char byte, int convert);

/* libseeq.c:171-352.  Scans the NUL-terminated string `data`.
 * Hits are written to out[0..cap) in the order the reference leaves them in
 * sq->match[] (i.e. AFTER the final array reversal, libseeq.c:345-349).
 * Returns the number of hits (may exceed cap; only cap are stored) or -1. */
long orc_string_match(const char *data, const char *keys, int m, int tau,
                      int options, orc_match_t *out, size_t cap);

/* Per-character trace of (capped distance, min_to_match) exactly as
 * seeqStringMatch reads them from the DFA vertex (libseeq.c:261-263) for the
 * first n text characters; used against testset.c:546-685.  Characters that
 * are not bases get dist = -1. */
void orc_trace(const char *data, size_t n, const char *keys, int m, int tau,
               int *dist, int *mtm);

/* Line-buffer scan mirroring the loop of seeqFileMatch (seeq.c:361-387) over
 * an in-memory buffer holding '\n'-separated lines (last line may lack the
 * newline).  fasta != 0 skips lines starting with '>' without counting them
 * (seeq.c:367-374).  For each counted line one entry is appended to
 * line_nhits (if non-NULL); hit records (line is 1-based, seeq.c:377) go to
 * rec[] as (line,start,end,dist) u64 quadruples in left-to-right order per
 * line for ALL, single record for FIRST/BEST.  Returns number of records
 * (may exceed rec_cap), *nlines = counted lines, *nmatchlines = lines with >=1
 * hit. */
long orc_buffer_scan(const char *buf, size_t nbytes, const char *keys, int m,
                     int tau, int options, int fasta,
                     uint64_t *rec, size_t rec_cap,
                     uint32_t *line_nhits, size_t line_cap,
                     uint64_t *nlines, uint64_t *nmatchlines);

/* Synthetic shape-R reads: counter-based, so any
 * range [first, first+n) can be generated independently.  Writes n lines of
 * `len` bases + '\n' to out (n*(len+1) bytes). */
void orc_synth_reads(char *out, uint64_t first, uint64_t n, int len,
                     const char *pattern_plain, int plen, int tau,
                     uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
