/*
 * include/seeq.h -- file-level API of seeq-mi355x.
 *
 * Drop-in for the reference's src/seeq.h:26-73: `seeqFileMatch` keeps the
 * reference's resumable one-result-per-call contract (reference
 * seeq.c:293-392) but is implemented as read-ahead + one batched GPU scan per
 * chunk + replay (seeq_amd/csrc/seeq_file.c).
 */
#ifndef SEEQ_AMD_SEEQ_H_
#define SEEQ_AMD_SEEQ_H_

#define SEEQ_VERSION "seeq-1.2"              /* reference seeq.h:26 */

#include "libseeq.h"
#include <stdlib.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct seeqfile_t seeqfile_t;

/* reference seeq.h:35-53 (filled by the CLI, reference seeq-main.c:422-439). */
struct seeqarg_t {
   int showdist;
   int showpos;
   int showline;
   int printline;
   int matchonly;
   int count;
   int compact;
   int dist;
   int verbose;
   int endline;
   int prefix;
   int split;
   int invert;
   int best;
   int non_dna;
   int all;
   size_t memory;
};

/* reference seeq.h:55-60.  flags bit0 = FASTA input (reference seeq.c:243-252).
 * Private read-ahead state hangs off a side table keyed by the pointer, so
 * the public layout is unchanged. */
struct seeqfile_t {
   int     flags;
   size_t  line;
   char  * info;
   FILE  * fdi;
};

/* reference seeq.h:64-68 */
#define SQ_ANY        0
#define SQ_MATCH      1
#define SQ_NOMATCH    2
#define SQ_COUNTLINES 3
#define SQ_COUNTMATCH 4

/* reference seeq.h:70-73 */
int          seeq            (char * expression, char * input, struct seeqarg_t args);
long         seeqFileMatch   (seeqfile_t * sqfile, seeq_t * sq, int match_opt, int file_opt);
seeqfile_t * seeqOpen        (const char * file);
int          seeqClose       (seeqfile_t * sqfile);

#ifdef __cplusplus
}
#endif
#endif
