/*
 * seeq_internal.h -- private host-side state shared by libseeq_api.c and
 * seeq_file.c.  Not installed.
 */
#ifndef SEEQ_INTERNAL_H_
#define SEEQ_INTERNAL_H_

#include <stddef.h>

#include "libseeq.h"
#include "seeq_amd.h"

#define SEEQ_ENGINE_MAGIC 0x5EE9A3D5u

/* What sq->dfa points at.  The reference stores a dfa_t there and its CLI
 * type-puns it as size_t[] in verbose mode (reference seeq.c:184-189: word 0
 * and *(size_t*)word 4); the leading compat words keep that read harmless
 * should the reference's own seeq.c ever be linked on top of this library. */
typedef struct seeq_engine_t {
   size_t             compat_f[5];
   size_t             compat_r[5];     /* sq->rdfa points here */
   size_t             compat_zero;
   unsigned           magic;
   unsigned long      id;              /* unique per seeqNew: cache key for seeqFileMatch */
   seeqdev_pattern_t *pat;             /* Peq tables in HBM */
   seeqdev_scan_t    *scan;            /* stream + workspace, created on first use */
   seeqdev_hit_t     *rec;             /* host copy of the last records */
   size_t             rec_cap;
} seeq_engine_t;

seeq_engine_t  *seeq_engine_of(const seeq_t *sq);
seeqdev_scan_t *seeq_engine_scan(seeq_engine_t *eng);
int             seeq_store_hits(seeq_t *sq, const seeqdev_hit_t *rec, size_t n);
/* seeq_file.c: an engine is about to be freed -- open files whose read-ahead scans use its pattern let go of it */
void            seeq_file_forget_engine(unsigned long eng_id);

#endif
