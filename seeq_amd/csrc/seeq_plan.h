/*
 * seeq_plan.h -- which kernels serve a scan: the decision of run_segments (seeq_device.hip) as a PURE host function.
 *
 * seeq_plan_scan() looks at the pattern (length, distance, what automata it has), the options, the line length of the text
 * and the fall-back flags a scan context has collected, and returns a ScanPlan -- nothing else decides the path; run_segments
 * only executes it (kernel instances, grids, launches).  No HIP in here: tests/host_harness.cpp compiles the same function for
 * the CPU and tests/test_kernel_core_host.py checks the plans of the BASELINE configurations without a GPU; SEEQ_EXPLAIN=1 makes
 * every scan print its plan (seeq_plan_print) on stderr.
 *
 * The automata of a pattern are built on first use: the planner asks for them through `ensure` (run_segments: build + upload
 * under the pattern's lock; the host test: seeq_dfa.h alone) only when the options and the text admit their kernel.
 */
#ifndef SEEQ_PLAN_H_
#define SEEQ_PLAN_H_

#include <stdint.h>
#include <stdio.h>
#include <string.h>

/* Test knobs, read from the environment ONCE per scan context (seeqdevScanNew).  (Round 5 removed the knobs that kept superseded kernels and
   timing-only experiments compiled in -- SEEQ_VERIFY / SEEQ_ORDER / SEEQ_EMIT_ALL = old, SEEQ_NO_SKIPCOUNT, SEEQ_NO_LL_FILTER, SEEQ_PAIR_PF, SEEQ_PAIR_EXP,
   SEEQ_DFA_WGS, SEEQ_EXACT, SEEQ_PACKED_STAGE: tag r05-before-prune builds them.) */
struct ScanKnobs {
   int  kernel;          /* SEEQ_FUSED_KERNEL: 0 auto, 1 "stream" (k_stream, never k_pair), 2 "direct", 3 "pair" (k_pair wherever the pattern has a pair automaton, selective or not) */
   int  tile_bytes;      /* SEEQ_TILE_BYTES: k_direct region size */
   bool no_filter;       /* SEEQ_NO_FILTER=1: complete automata only */
   int  min_wu;          /* SEEQ_STREAM_WU=6|8: at least this many warm-up dwords (tests: the 16-byte warm-up off) */
   bool no_window;       /* SEEQ_NO_WINDOW=1: behind k_pair the exact pass scans a candidate line to its end, as behind the other filters */
   bool no_myers;        /* SEEQ_NO_MYERS=1: long lines without an automaton go to the generic path (one line per lane) as before */
   bool no_leaders;      /* SEEQ_NO_LEADERS=1: long lines are walked by one lane each whatever the number of their candidates (A/B, tests) */
   bool no_packed_quad;  /* SEEQ_PACKED_QUAD=0: the packed walk over the pair table even where the pattern has a quad table (A/B, tests) */
   bool no_sub;          /* SEEQ_STREAM_SUB=0: SQ_CONVERT text with non-DNA bytes is re-run on the per-line kernels (as SQ_IGNORE) */
   bool explain;         /* SEEQ_EXPLAIN=1: every scan prints its plan on stderr */
};

/* what the planner needs to know of a pattern's automata (seeq_dfa.h); state: 0 not tried yet, 1 there, -1 none fits */
struct PlanAutomata {
   int    sdfa_state, sdfa_parts, sdfa_warm;
   double sdfa_pacc;
   int    pair_state, pair_warm;
   double pair_pacc;
};

struct PlanIn {
   int    wlen, tau, options, want;
   double avg_line;        /* bytes per line incl. newline (hint or sample) */
   double line_hint;       /* the caller's hint; 0: avg_line is a sample */
   int    force_path;      /* 0 auto, 1 generic, 2 fused */
   bool   no_stream, force_ll, no_stream_nd, no_window, no_leaders, sample_dirty, multi_active;
   size_t seg_bytes;
   const ScanKnobs *kn;
};

struct ScanPlan {
   int  rc;                /* 0; -2: a multi-pattern scan on text / options that are not k_pair's -- a scan per pattern */
   int  path;              /* seeqdevScanLastPath: 1 generic, 3 k_direct, 5 k_stream, 6 k_pair, 7 k_stream's Myers mode */
   int  fw;                /* words of the one-pass kernels' column (1: <= 30 positions, 2: <= 62) */
   bool fusable, use_fused, use_stream, use_pair, use_myers, use_direct;
   bool can_sub;           /* SQ_CONVERT / SQ_IGNORE served by k_stream's substituting variant */
   bool filter;            /* every hit line of the scan kernel is a candidate */
   bool stream_ll;         /* long-line variant: bookkeeping for the window walk */
   int  stream_sub;        /* 0, 1 SQ_CONVERT, 2 SQ_IGNORE */
   int  stream_wu;         /* warm-up dwords of the scan kernel */
   bool superset, need_nh, nh_is_count;
   bool window_ok;         /* k_pair: the exact pass scans candidate windows */
   bool ll_filter;         /* k_stream's long-line variant over a partition filter */
   bool pair_ll;           /* k_pair's long-line variant (round 5) */
   bool ll_restart;        /* ... walking the filter's RESTART table: every part occurrence is a candidate, windows of m + tau either side (round 5) */
   bool leaders;           /* long lines: candidates far behind the one before them get lanes of their own */
   bool lead_best;
   bool verify;            /* the exact pass is k_verify (+ k_nh_top, k_emit1): filters on text without skipped bytes, read-length lines */
   bool order2;            /* the hit list is made by seeq_order.h's three launches (read-length lines behind k_pair / k_stream) */
   bool ig;                /* k_pair under SQ_IGNORE: line markers (seeq_pair.h IG), k_exact1 behind it */
   uint32_t skip_back, walk_ext, skip_thr;
};

typedef void (*seeq_plan_ensure_fn)(void *ctx, int which /* 0: k_stream's automaton, 1: the pair automaton */, int complete_only, PlanAutomata *au);

/* libseeq.h / seeq_amd.h bits, restated so that the header stands alone */
#define PLAN_MASK_NONDNA 0x0C
#define PLAN_MASK_INPUT  0x10
#define PLAN_SQ_BEST     0x01
#define PLAN_SQ_ALL      0x02
#define PLAN_SQ_CONVERT  0x04
#define PLAN_SQ_IGNORE   0x08
#define PLAN_SQ_STREAM   0x10
#define PLAN_FASTA       0x100
#define PLAN_SINGLELINE  0x200
#define PLAN_WANT_COUNTMATCH 1
#define PLAN_WANT_RECORDS    2
#define PLAN_MAX_WLEN   30
#define PLAN_MAX_WLEN2  62

static inline ScanPlan seeq_plan_scan(const PlanIn &in, PlanAutomata &au, seeq_plan_ensure_fn ensure, void *ctx)
{
   ScanPlan p;
   memset(&p, 0, sizeof p);
   const ScanKnobs &kn = *in.kn;
   const int options = in.options, want = in.want;
   const bool fasta = (options & PLAN_FASTA) != 0;
   const bool single = (options & PLAN_SINGLELINE) != 0;
   const int match_opt = options & 3;
   p.need_nh = want == PLAN_WANT_COUNTMATCH || (want == PLAN_WANT_RECORDS && match_opt == PLAN_SQ_ALL);
   p.nh_is_count = p.need_nh;                            /* nh[] = hits per line; else (superset filters) a 0/1 verdict per line */
   /* Path selection.  Patterns of <= 62 positions (one or two Myers words with two spare flag bits) on line input
      take a ONE-PASS scan kernel + the exact pass; everything else the generic index + k_forward<W> path. */
   p.fw = in.wlen <= PLAN_MAX_WLEN ? 1 : 2;
   p.fusable = !single && in.wlen <= PLAN_MAX_WLEN2;
   const int nd = options & PLAN_MASK_NONDNA;
   const int stream_ch = 128;                            /* bytes per lane of k_stream / k_pair */
   /* k_stream: line-agnostic table-driven scan (seeq_stream.h), the default whenever the pattern has an automaton that
      fits LDS (seeq_dfa.h): the complete Levenshtein automaton (its verdicts are exact) or, for longer patterns /
      larger distances, a partition FILTER automaton (its hit lines are candidates: the exact pass verifies them). */
   {
      /* SQ_FAIL: always.  SQ_CONVERT: exact through the SUB variant (non-DNA bytes replaced by 'N' in registers).
         SQ_IGNORE on read-length lines: the SUB variant with skip bytes (its hit lines become candidates where a skipped
         byte sits in a warm-up window).  Otherwise (SQ_IGNORE on long lines; without SUB) k_stream is exact on clean text
         only: it runs until it meets a non-DNA byte (Counters.dirty -> overflow flag 16: the scan is re-run on the per-line
         kernels, for good), and not on FASTA input (header lines are made of such bytes). */
      const bool long_lines = in.avg_line > 600.0 || in.force_ll;
      p.can_sub = (nd == PLAN_SQ_CONVERT || (nd == PLAN_SQ_IGNORE && !long_lines && !in.no_stream_nd)) && !fasta && !kn.no_sub;
      const bool dfa_opts = (options & PLAN_MASK_INPUT) == 0 && (nd == 0 || p.can_sub || (!in.no_stream_nd && !fasta));
      if (p.fusable && in.force_path != 1 && dfa_opts && !in.no_stream && kn.kernel != 2) {
         if (au.sdfa_state == 0) ensure(ctx, 0, kn.no_filter ? 1 : 0, &au);
         p.use_stream = au.sdfa_state == 1 && in.seg_bytes % (64u * (unsigned)stream_ch) == 0;
         if (p.use_stream && au.sdfa_parts > 1) {
            /* a filter: worth it while few lines are false candidates (each costs a whole-line exact scan).  On long lines the rate is judged per
               BYTE: a candidate costs the window walk ~170 columns, the Myers mode steps every byte -- measured on the published sweep's shape
               (3.2 GB): the filter walk 1.2 ms + 3.3 ms per candidate-per-KB; the Myers mode 3.1 ms with one word, 4.6 ms with two (round 5: its loop
               unrolled at last, lean steps; 6 / 11.5 - 13.5 ms before, when the filter won below 1.4 / 3.2 candidates per KB) -- so the filter wins
               below 0.58 / 1.0 candidates per KB (profiles/r05_chrom_sweep.txt); the walk runs on m + tau + 2 columns behind a candidate's chunk
               (walk_ext: at most a block, so that a leader's fresh start still lies behind the walk before it) */
            const double ll_pacc_max = p.fw == 1 ? 0.00058 : 0.0010;
            if (long_lines ? (au.sdfa_pacc > ll_pacc_max || in.wlen + in.tau + 2 > 64) : au.sdfa_pacc * in.avg_line > 0.25) p.use_stream = false;
         }
      }
   }
   /* k_pair (seeq_pair.h): the same walk, two text bytes per table step, over the pattern's pair automaton -- a prefix or a
      partition filter, so every hit line of it is a candidate.  Read-length lines under SQ_FAIL / SQ_CONVERT (aliased bytes
      keep a superset a superset; a skipped byte, SQ_IGNORE, does not), while it makes few false candidates. */
   {
      const bool long_lines = (in.avg_line > 600.0 && kn.kernel != 3) || in.force_ll;      /* (a candidate inside a line of a whole tile sets force_ll) */
      /* (round 5: long lines too -- LL in seeq_pair.h -- under SQ_FAIL / SQ_CONVERT on plain text, for one pattern, while its automaton flags few enough
         positions for the window walk: the rule of k_stream's long-line filters below, with the restart walk's cost per candidate) */
      const bool pair_ll = long_lines && (nd == 0 || nd == PLAN_SQ_CONVERT) && !fasta && !in.multi_active && kn.kernel != 1 && !kn.no_window;
      /* (round 5: SQ_IGNORE too -- a line that holds a skipped byte is named whole by a marker, every other line is what it is under SQ_FAIL: IG in
         seeq_pair.h; not FASTA input, not several patterns at once) */
      const bool ig_ok = nd == PLAN_SQ_IGNORE && !fasta && !in.multi_active;
      if (p.fusable && in.force_path != 1 && (options & PLAN_MASK_INPUT) == 0 && (nd == 0 || nd == PLAN_SQ_CONVERT || ig_ok) && (!long_lines || pair_ll) && !in.no_stream &&
          /* (round 5: text full of foreign bytes -- FASTQ records -- stays here: a tile that fails the fast alphabet check makes its newline
             masks again from its registers, and the exact pass looks at the bytes before a window, seeq_verify.h.  FASTA records with a
             header per read stay with k_stream: the header test of the FA variant reads a byte per newline) */
          (kn.kernel == 3 || (kn.kernel == 0 && !(in.sample_dirty && in.line_hint <= 0 && fasta))) && in.seg_bytes % (64u * 128u) == 0) {
         if (au.pair_state == 0) ensure(ctx, 1, 0, &au);
         p.use_pair = au.pair_state == 1 && (kn.kernel == 3 || in.multi_active || (long_lines ? au.pair_pacc > 0.0 && au.pair_pacc <= (p.fw == 1 ? 0.00058 : 0.0010) && in.wlen + in.tau + 2 <= 64      /* (0: no estimate -- a prefix that accepts nearly everything, m = 42 with 15 errors) */
                                                                                                   : au.pair_pacc * in.avg_line <= 0.25));
         p.pair_ll = p.use_pair && long_lines;
      }
      if (p.use_pair) { p.use_stream = true; p.can_sub = false; p.ig = nd == PLAN_SQ_IGNORE; }
      if (in.multi_active && !p.use_pair) { p.rc = -2; return p; }      /* this text / these options are not k_pair's: a scan per pattern */
   }
   /* k_stream's Myers mode: no automaton fits (or only a filter that is not selective enough), the lines are too long for
      the per-line kernels -- the same line-agnostic chunks, the bit-vector column instead of the table (seeq_stream.h) */
   if (!p.use_stream && !p.use_pair && p.fusable && in.force_path != 1 && (options & PLAN_MASK_INPUT) == 0 && (nd == 0 || nd == PLAN_SQ_CONVERT) &&
       kn.kernel != 2 && !kn.no_myers && !in.no_stream && (in.avg_line > 260.0 || in.force_ll) && in.seg_bytes % (64u * 128u) == 0) {
      p.use_myers = true; p.use_stream = true; p.can_sub = false;
   }
   p.filter = p.use_pair || (p.use_stream && !p.use_myers && au.sdfa_parts > 1);
   p.use_fused = p.fusable && (in.avg_line <= 260.0 || p.use_stream) && in.force_path != 1;      /* k_direct regions are <= 16 KiB (~62 lines) */
   p.stream_wu = !p.use_stream ? 8 : au.sdfa_warm <= 16 ? 4 : au.sdfa_warm <= 24 ? 6 : 8;        /* warm-up dwords */
   if (p.use_pair) p.stream_wu = au.pair_warm <= 16 ? 4 : (au.pair_warm + 3) / 4;
   if (p.stream_wu < kn.min_wu) p.stream_wu = kn.min_wu >= 8 ? 8 : 6;
   if (p.use_myers) p.stream_wu = in.wlen + in.tau - 1 <= 64 ? 16 : 32;       /* (an 8-word instance exists in principle; the one-word one the compiler makes of it spills 189 registers) */
   if (p.use_fused && p.use_stream) {
      p.stream_ll = in.avg_line > 600.0 || in.force_ll;   /* long lines: bookkeeping for the window walk */
      p.stream_sub = p.can_sub ? (nd == PLAN_SQ_IGNORE ? 2 : 1) : 0;
      if (p.use_myers) { p.stream_ll = true; p.stream_sub = 0; }       /* (the window walk of the exact pass serves every line length) */
      if (p.use_pair) { p.stream_ll = p.pair_ll; p.stream_sub = 0; }
   }
   p.use_direct = p.use_fused && !p.use_stream;
   p.path = p.use_fused ? (p.use_pair ? 6 : p.use_myers ? 7 : p.use_stream ? 5 : 3) : 1;
   p.superset = p.use_stream;                             /* the scan kernel's hit lines are candidates: nh[] decides */
   if (p.superset) p.need_nh = true;
   /* ---- the post-pass ---- */
   p.skip_back = (uint32_t)(in.wlen + in.tau - 1) + (p.use_pair ? 1u : 0u);      /* (k_pair reports the second byte of a pair) */
   p.window_ok = p.use_pair && !p.pair_ll && !in.no_window && !kn.no_window;
   p.ll_filter = p.use_fused && p.use_stream && p.stream_ll && p.filter && !p.use_pair && !p.use_myers;
   /* Which table the long-line filter walks.  Absorbing (round 4): one candidate per chain and line, the exact pass scans the rest of the candidate's
      chunk (~145 columns, lanes of very different lengths).  Restart (round 5): every part occurrence is a candidate (about twice as many), the exact
      pass scans m + tau either side of each (~2 (m + tau) columns, all lanes alike); its warm-up remembers acceptances (2 VALU per warm-up byte and
      chain).  The restart table pays where the exact pass is the larger part of the scan: filters that flag more than ~1 position in 20 KB. */
   p.ll_restart = (p.ll_filter && !kn.no_window && au.sdfa_pacc > 0.00005) || p.pair_ll;      /* (SEEQ_NO_WINDOW=1: the absorbing table, A/B and tests; k_pair's automata are restart automata) */
   p.walk_ext = p.ll_filter ? (uint32_t)(in.wlen + in.tau + 2) : 0u;
   p.skip_thr = (p.use_fused && p.use_stream && !p.use_pair && !p.use_myers && p.stream_sub == 2 && au.sdfa_parts == 1) ? (uint32_t)(in.wlen - in.tau) : 0u;
   p.lead_best = want == PLAN_WANT_RECORDS && match_opt == PLAN_SQ_BEST;      /* (one record per line: the groups' best hits are reduced per line) */
   p.leaders = p.use_stream && p.stream_ll && (p.nh_is_count || p.lead_best) && p.use_fused && !kn.no_leaders && !in.no_leaders && !in.multi_active;
   p.order2 = p.use_fused && p.use_stream && !p.stream_ll;
   p.verify = p.need_nh && p.use_fused && p.filter && !p.stream_ll && !in.multi_active &&
              (options & (PLAN_SQ_IGNORE | PLAN_SQ_STREAM)) == 0;
   return p;
}

static inline void seeq_plan_print(FILE *f, const PlanIn &in, const PlanAutomata &au, const ScanPlan &p)
{
   static const char *const kname[8] = {"?", "generic (k_nl_* + k_forward)", "?", "k_direct", "?", "k_stream", "k_pair", "k_stream, Myers mode"};
   fprintf(f, "seeq plan: m=%d tau=%d options=0x%x want=%d avg_line=%.1f%s -> %s%s%s%s%s | column words %d, warm-up %d B, sub %d | "
              "automata: stream %s (parts %d, warm %d, p_acc %.2g), pair %s (warm %d, p_acc %.2g) | "
              "post-pass: %s, %s%s%s, skip_back %u%s\n",
           in.wlen, in.tau, (unsigned)in.options, in.want, in.avg_line, in.line_hint > 0 ? " (hint)" : "",
           kname[p.path & 7], p.filter ? " [candidates: filter / prefix]" : "", p.pair_ll ? " [long lines: LL variant, windows of the restart walk]" : p.stream_ll ? " [long lines]" : "", p.ll_restart ? " [filter on long lines, restart table]" : p.ll_filter ? " [filter on long lines]" : "",
           p.rc == -2 ? " [multi: not k_pair's -- a scan per pattern]" : "",
           p.fw, 4 * p.stream_wu, p.stream_sub,
           au.sdfa_state == 1 ? "yes" : au.sdfa_state == 0 ? "not asked" : "none", au.sdfa_parts, au.sdfa_warm, au.sdfa_pacc,
           au.pair_state == 1 ? "yes" : au.pair_state == 0 ? "not asked" : "none", au.pair_warm, au.pair_pacc,
           p.order2 ? "k_tiles_post + k_order + k_bounds2" : p.use_fused ? "k_fused_post + k_scanset_* + reorder + bounds" : "k_compact",
           p.verify ? "k_verify + k_nh_top" : p.need_nh ? "k_exact1<COUNT> + scan" : "no count pass",
           p.window_ok ? " on candidate windows" : "", p.leaders ? " [leaders]" : "", p.skip_back, p.skip_thr ? ", skip count" : "");
}

#endif
