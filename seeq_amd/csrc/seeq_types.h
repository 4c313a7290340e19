/*
 * seeq_types.h -- the argument blocks every kernel of the scan shares (seeq_device.hip, seeq_verify.hip):
 * the per-run counters in HBM and the per-segment arguments of the post-pass.
 */
#ifndef SEEQ_TYPES_H_
#define SEEQ_TYPES_H_

#include <stdint.h>
#include "seeq_amd.h"

/* ========================================================================== */
/* Device-side bookkeeping                                                    */
/* ========================================================================== */
struct Counters {
   /* per segment */
   uint32_t seg_nlines;     /* raw lines starting in the segment (FASTA headers included) */
   uint32_t seg_nhitlines;
   uint32_t seg_nheaders;
   uint32_t seg_nrec;       /* hits (records) of the segment */
   /* running totals over segments */
   uint64_t lines;          /* counted lines (headers excluded) */
   uint64_t matchlines;
   uint64_t hits;
   uint64_t records;
   uint64_t headers;
   /* workspace overflow report */
   uint32_t overflow;       /* 1 lines, 2 hit lines, 4 records (workspace too small); 8, 16: k_stream cannot serve this text; 32: wants its long-line variant; 64: a hit entry points outside its segment (a bug: the scan fails); 128: k_pair, a line with candidates on both sides of a seam; 256: long lines, a leader's fresh start lies inside the walk before it (the run is void, the next one keeps a line in one lane) */
   uint32_t need_lines;     /* max over segments */
   uint32_t need_hitlines;  /* max over segments */
   uint32_t seg_novf;       /* k_exact1: 1 when a wave's overflow list (emissions beyond the first of their lines, COUNT -> EMIT) did not fit */
   uint32_t seg_nmatch;     /* lines of the segment with >= 1 verified hit (superset filters) */
   uint32_t dirty;          /* k_stream: the text holds bytes outside {ACGTN, acgtn, '\n'}: its hit lines need verifying */
   uint64_t need_records;   /* total */
   uint32_t prev_hit_line;  /* k_stream: line number of the last hit line of the previous segment (a line can span segments) */
   uint32_t seg_last_nl;    /* k_stream: segment-relative offset + 1 of the last newline of the segment (0: none) */
   uint32_t seg_dirty_tiles; /* k_stream, long-line mode: tiles of the segment that hold a non-alphabet byte */
   uint32_t emit_nhl;       /* k_nh_top -> k_emit1: the segment's hit-list length and record base, kept past the end of the segment */
   uint32_t pad4[2];
   uint64_t emit_base;
};

struct ScanArgs {
   const uint8_t *text;      /* whole buffer */
   uint64_t       nbytes;
   uint64_t       seg_base;  /* first byte of the segment */
   uint32_t       seg_len;
   uint32_t       first_seg; /* 1 for segment 0 */
   const uint32_t *peq;      /* [2][5][W]: forward, reverse */
   int            m, tau, options, want;
   uint32_t      *line_start;   uint32_t cap_lines;
   uint32_t      *tile_cnt;     uint32_t ntiles;
   uint64_t      *hitmask;
   uint64_t      *hdrmask;
   uint32_t      *wave_off;
   uint32_t      *hdr_off;
   uint32_t      *hit_start;    /* per hit line: segment-relative offset of its first byte */
   uint32_t      *hit_line;     /* per hit line: 1-based counted line number (reference seeq.c:377) */
   uint32_t       cap_hitlines;
   uint32_t      *nh;           /* per hit line: hits, then exclusive offsets */
   seeqdev_hit_t *records;      uint64_t cap_records;
   uint64_t      *rec_off;      /* per record: byte offset (in the whole buffer) of the line it belongs to */
   uint32_t       use_nh;       /* record slots / line verdicts come from the per-line counts nh[]: 1 = ALL, COUNTMATCH;
                                   3 = k_stream (superset only when Counters.dirty or `filter`,
                                   hit list may repeat a line: hit_start = 0xFFFFFFFF marks a repeat) */
   uint32_t       pos_bias;     /* k_stream: seg_base here is the segment's base minus this (multiple of 128) */
   const uint32_t *tile_dirty;  /* k_stream, long-line mode: exclusive prefix of the per-tile "holds a non-alphabet byte" flags */
   const uint64_t *tile_dmask;  /* ... and per tile one bit per 128-byte chunk */
   uint32_t       stream_ntiles, stream_tile_bytes;
   uint32_t       stream_ch;    /* k_stream: bytes per lane chunk (0 = another kernel made the hit list) */
   uint32_t       filter;       /* k_stream walked a partition FILTER automaton: every hit line is only a candidate */
   uint32_t       skip_back;    /* columns before a candidate from which a fresh column gives exact scores: m + tau - 1 */
   const uint32_t *hit_last;    /* packed read batches: per hit line the column of its LAST candidate (else NULL: the repeats in the hit list say) */
   uint32_t       *walk_end;    /* long lines, leaders (seeq_stream.h): per entry where the walk of the group before it ended; NULL: off */
   const uint32_t *hit_idx;     /* several patterns, one walk: this pattern's list holds indices into the shared per-line arrays (else NULL) */
   uint32_t       window_ok;    /* k_pair: every candidate the walk dropped is announced (nh[] bit 1 of the kept one) and repeats of a line
                                   follow it in the hit list -- a line with ONE candidate is scanned over that candidate's window only */
   uint32_t       *nh_sum;      /* k_verify: nh[] holds offsets inside chunks of 256 entries, nh_sum[k >> 8] the records before the chunk (NULL: nh[] holds
                                   the segment-wide offsets, made by the three-launch scan) */
   uint32_t        walk_ext;    /* window walk behind a partition FILTER on long lines: columns a window runs on behind the chunk of its candidate --
                                   m + tau + 2: an occurrence ends no further behind the part occurrence the filter saw (0: the candidates are hit ends) */
   uint32_t        ll_restart;  /* the long-line filter walked its RESTART table (seeq_dfa_restart_variant): windows end m + tau + 2 behind a candidate */
   uint32_t        rec_pitch;   /* packed read batches: the exact pass runs on a private staging text -- a record's line offset is reported as
                                   (line - 1) * rec_pitch, the offset the same read has in the ASCII form of the batch (0: the line's real offset) */
   uint32_t       *nz_sum;      /* k_verify: per chunk the entries with >= 1 hit (NULL: not wanted) */
   uint32_t        fin;         /* != 0: k_nh_top ends the segment (seg_end_body with flags fin - 1); 0: k_seg_end does, behind the EMIT pass */
   uint32_t        ig_thr;      /* k_pair under SQ_IGNORE: m - tau (line markers: seeq_pair.h IG); else 0 */
   uint32_t        ig_need, ig_bval, ig_bmask;      /* ... the frequency bound of seeq_order.h: at least ig_need bytes c with (c & ig_bmask) == ig_bval in a marker's line (0: off) */
   const uint4    *ig_ent;      /* ... the ordered entries (seeq_order.h): word 3, bit 1 = the entry is a line marker (bit 0: ... that stands on a candidate of the walk); else NULL */
   uint32_t        vrange;      /* k_verify: entries per workgroup range, 256 .. 1024 (0: 256): the repeats of a range are packed away before the walk */
   Counters      *cnt;
};

/* per hit-list entry: records of the segment before it (see nh_sum) */
#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t nh_at(const ScanArgs &a, uint32_t k) { return a.nh[k] + (a.nh_sum ? a.nh_sum[k >> 8] : 0u); }
/* what seeqdevScanCopyOffsets reports for a record of line number `line` whose text starts at byte `off` of the scanned buffer */
__device__ __forceinline__ uint64_t rec_off_of(const ScanArgs &a, uint64_t off, uint32_t line) { return a.rec_pitch ? (uint64_t)(line - 1u) * a.rec_pitch : off; }
#endif


#endif
