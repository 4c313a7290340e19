/*
 * seeq_post.h -- host-side launchers of the post-pass kernels that live in their own translation unit
 * (seeq_verify.hip: compiled beside seeq_device.hip).
 */
#ifndef SEEQ_POST_H_
#define SEEQ_POST_H_

#include <hip/hip_runtime.h>
#include "seeq_types.h"
#include "seeq_scan_common.h"      /* FusedArgs */

#define VERIFY_ANY  0          /* presence is enough, or the first emission is the record (SQ_FIRST / SQ_COUNT, line counts) */
#define VERIFY_BEST 1          /* SQ_BEST records: the first emission with the smallest distance */
#define VERIFY_ALL  2          /* every emission counts (SQ_ALL records, match counts) */

/* k_verify<W, VAR> (seeq_verify.h) + k_nh_top: fw = column words (1, 2); var = VERIFY_ANY / _BEST / _ALL; grid workgroups of 256 (an
   EMIT pass of k_exact1 behind it must use the same grid); a.nz_sum != NULL: Counters.seg_nmatch = entries with >= 1 hit;
   a.fin != 0: k_nh_top ends the segment */
void seeq_launch_verify(int fw, int var, unsigned grid, hipStream_t st, const ScanArgs &a, const uint32_t *eq, const uint32_t *hit_col,
                        uint4 *cache);
/* k_verify_packed<W, VAR> (seeq_verify_packed.h) + k_nh_top: the same over the windows of a packed read batch, read from the batch itself */
struct VerifyPacked;
void seeq_launch_verify_packed(int fw, int var, unsigned grid, hipStream_t st, const ScanArgs &a, const void *bases, const void *nmask, uint32_t stride,
                               uint32_t nstride, uint32_t read_len, uint64_t total_bytes, uint64_t ntotal_bytes, const uint32_t *eq, const uint32_t *hit_col, uint4 *cache);
/* k_emit1: the records of a segment with one record per line at most, from k_verify's cache */
void seeq_launch_emit1(unsigned grid, hipStream_t st, const ScanArgs &a, const uint4 *cache);
/* k_emit_all: SQ_ALL records behind k_verify<VERIFY_ALL> -- first records from the cache, the others from the overflow lists of the
   `vgrid` workgroups k_verify ran with; when a list did not fit: k_exact1's EMIT pass over the same arguments */
void seeq_launch_emit_all(int fw, unsigned grid, unsigned vgrid, hipStream_t st, const ScanArgs &a, const uint32_t *eq, const uint32_t *hit_col, uint4 *cache);

/* seeq_order.h: per-wave hit slices -> ordered per-line arrays on read-length lines (three launches).  bsum: 2 * nb words of
   workspace, nb = blocks of 2 048 tiles (<= SEEQ_ORDER_MAX_BLOCKS); ent: one uint4 per hit-list entry */
#define SEEQ_ORDER_MAX_BLOCKS 1024
#define SEEQ_ORDER_BLOCK      2048
void seeq_launch_tiles_post(hipStream_t st, const FusedArgs &f, uint32_t nslices, uint32_t *bsum, uint32_t nb);
void seeq_launch_order(unsigned grid, hipStream_t st, const FusedArgs &f, uint32_t nslices, const uint32_t *bsum, uint32_t nb, uint4 *ent);
void seeq_launch_bounds2(unsigned grid, hipStream_t st, const ScanArgs &a, uint4 *ent, uint32_t *hit_col);

#endif
