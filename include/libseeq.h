/*
 * include/libseeq.h -- per-line approximate-match API of seeq-mi355x.
 *
 * Drop-in for the reference's public header (reference src/libseeq.h:30-100):
 * same option bits, same public struct layouts (callers read sq->match[0],
 * sq->string, sq->keys ... directly: reference seeqmodule.c:778-829,
 * seeq.c:128-171, test/testset.c:772-801), same prototypes, same `seeqerr`
 * convention.  The implementation behind it is new: the matching itself runs
 * in HIP kernels on an MI355X (seeq_amd/csrc/seeq_device.hip), there is no CPU
 * matcher in this library and every entry point fails with -1/NULL
 * (seeqerr = 0, errno set) when no GPU / HIP runtime is usable.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif

#ifndef SEEQ_AMD_LIBSEEQ_H_
#define SEEQ_AMD_LIBSEEQ_H_

#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIBSEEQ_VERSION "libseeq-1.1"       /* reference libseeq.h:30 */
#define COLOR_TERMINAL 1

/* Match options (reference libseeq.h:34-48).  One value per group, OR-ed. */
#define SQ_FIRST      0x00
#define SQ_BEST       0x01
#define SQ_ALL        0x02
#define SQ_COUNT      0x03

#define SQ_FAIL       0x00
#define SQ_CONVERT    0x04
#define SQ_IGNORE     0x08

#define SQ_LINES      0x00
#define SQ_STREAM     0x10

#define MASK_MATCH    0x03
#define MASK_NONDNA   0x0C
#define MASK_INPUT    0x10

#define INITIAL_MATCH_STACK_SIZE 16         /* reference libseeq.h:52 */

/* 0 => consult errno; otherwise an index understood by seeqPrintError()
 * (reference libseeq.c:28-41; seeqOpen/seeqClose store a raw errno here,
 * reference seeq.c:226,234,286). */
extern int seeqerr;

typedef struct seeq_t   seeq_t;
typedef struct match_t  match_t;
typedef struct mstack_t mstack_t;

/* reference libseeq.h:62-66; `end` is exclusive. */
struct match_t {
   size_t   start;
   size_t   end;
   size_t   dist;
};

/* reference libseeq.h:68-80.  Field order and types are ABI.  `dfa` and
 * `rdfa` are opaque; here they point at the device-side pattern handle
 * (seeqdev_pattern_t, include/seeq_amd.h) instead of a lazily built DFA. */
struct seeq_t {
   size_t    hits;
   size_t    stacksize;
   match_t * match;
   size_t    bufsz;
   char    * string;
   int       tau;
   int       wlen;
   char    * keys;
   char    * rkeys;
   void    * dfa;
   void    * rdfa;
};

/* reference libseeq.h:82-86 (the match-stack utilities below). */
struct mstack_t {
   size_t  size;
   size_t  pos;
   match_t match[];
};

/* reference libseeq.h:89-95 */
seeq_t     * seeqNew         (const char * pattern, int mismatches, size_t maxmemory);
void         seeqFree        (seeq_t * sq);
match_t    * seeqMatchIter   (seeq_t * sq);
char       * seeqGetString   (seeq_t * sq);
long         seeqStringMatch (const char * data, seeq_t * sq, int options);
const char * seeqPrintError  (void);
int          seeqAddMatch    (seeq_t * sq, match_t match);
/* reference libseeq.h:96-100: match-stack utilities (the reference's own call
 * sites are commented out, libseeq.c:231-234,340-343; they work here as there). */
mstack_t   * stackNew        (size_t size);
int          stackAddMatch   (mstack_t ** stackp, match_t match);
int          recursive_merge (size_t start, size_t end, int tau, seeq_t * sq, mstack_t ** stackp);

#define RESET       "\033[0m"
#define BOLDRED     "\033[1m\033[31m"
#define BOLDGREEN   "\033[1m\033[32m"

#ifdef __cplusplus
}
#endif
#endif
