/*
 * seeq_main.c -- command line front end of seeq-mi355x.
 *
 * Same flags, defaults, masking rules and messages as the reference CLI
 * (src/seeq-main.c:36-58 usage text, :118-141 options, :403-440 defaults and
 * masking); flag parsing is not a data path, so this is a thin table-driven
 * getopt loop around seeq() (seeq_file.c).
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "seeq.h"

static const char usage_text[] =
   "Usage:"
   "  seeq [options] pattern inputfile\n"
   "\n   MATCHING OPTIONS:\n"
   "    -d --distance [#]    maximum Levenshtein distance [default 0]\n"
   "    -i --invert          return only the non-matching lines\n"
   "    -b --best            scan the whole line to find the best match [default: first match only]\n"
   "    -a --all             returns all the matches (implies -m) [default: first match only]\n"
   "    -x --nondna [0,1,2]  non-DNA characters: 0-skip line, 1-convert to 'N', 2-ignore. [default 0]\n"
   "\n   FORMAT OPTIONS:\n"
   "    -c --count           returns the count of matching lines\n"
   "    -m --match-only      print only the matched sequence\n"
   "    -n --no-printline    do not print the matched sequence\n"
   "    -l --lines           shows the line number of the match\n"
   "    -p --positions       shows the position of the match\n"
   "    -k --print-dist      shows the Levenshtein distance of the match\n"
   "    -f --compact         prints output in compact format (line:pos:dist)\n"
   "    -e --end             print only the end of the line, starting after the match\n"
   "    -r --prefix          print only the prefix, ending before the match\n"
   "\n   OTHER OPTIONS:\n"
   "    -v --version         print version\n"
   "    -y --memory          set DFA memory limit (in MB)\n"
   "    -z --verbose         verbose using stderr\n";

static void say_version(void) { fprintf(stderr, SEEQ_VERSION "\n"); }

static int die(const char *msg)
{
   say_version();
   fprintf(stderr, "%s", msg);
   fprintf(stderr, "use '-h' for help.\n");
   return EXIT_FAILURE;
}

/* One row per flag: short name, long name, takes a value, "set twice" message. */
struct flag_t {
   char        c;
   const char *name;
   int         has_arg;
   const char *twice;
   int         value;      /* -1 = unset */
};

enum { F_D, F_I, F_B, F_A, F_X, F_C, F_M, F_N, F_L, F_P, F_K, F_F, F_E, F_R, F_Y, F_Z, NFLAGS };

int main(int argc, char **argv)
{
   struct flag_t fl[NFLAGS] = {
      [F_D] = {'d', "distance", 1, "error: distance option set more than once.\n", -1},
      [F_I] = {'i', "invert", 0, "error: invert option set more than once.\n", -1},
      [F_B] = {'b', "best", 0, "error: 'best' option set more than once.\n", -1},
      [F_A] = {'a', "all", 0, "error: 'all' option set more than once.\n", -1},
      [F_X] = {'x', "nondna", 1, "error: 'nondna' option set more than once.\n", -1},
      [F_C] = {'c', "count", 0, "error: count option set more than once.\n", -1},
      [F_M] = {'m', "match-only", 0, "error: match-only option set more than once.\n", -1},
      [F_N] = {'n', "no-printline", 0, "error: no-printline option set more than once.\n", -1},
      [F_L] = {'l', "lines", 0, "error: show-line option set more than once.\n", -1},
      [F_P] = {'p', "positions", 0, "error: show-position option set more than once.\n", -1},
      [F_K] = {'k', "print-dist", 0, "error: show-distance option set more than once.\n", -1},
      [F_F] = {'f', "format-compact", 0, "error: format-compact option set more than once.\n", -1},
      [F_E] = {'e', "end", 0, "error: line-end option set more than once.\n", -1},
      [F_R] = {'r', "prefix", 0, "error: 'prefix' option set more than once.\n", -1},
      [F_Y] = {'y', "memory", 1, "error: memory option set more than once.\n", -1},
      [F_Z] = {'z', "verbose", 0, "error: verbose option set more than once.\n", -1},
   };
   if (argc == 1) {
      say_version();
      fprintf(stderr, "%s\n", usage_text);
      return EXIT_SUCCESS;
   }
   struct option longopts[NFLAGS + 3];
   char shortopts[3 * NFLAGS + 8];
   size_t so = 0;
   for (int i = 0; i < NFLAGS; i++) {
      longopts[i] = (struct option){fl[i].name, fl[i].has_arg ? required_argument : no_argument, 0, fl[i].c};
      shortopts[so++] = fl[i].c;
      if (fl[i].has_arg) shortopts[so++] = ':';
   }
   longopts[NFLAGS] = (struct option){"version", no_argument, 0, 'v'};
   longopts[NFLAGS + 1] = (struct option){"help", no_argument, 0, 'h'};
   longopts[NFLAGS + 2] = (struct option){0, 0, 0, 0};
   shortopts[so++] = 'v';
   shortopts[so++] = 'h';
   shortopts[so] = 0;

   int c;
   while ((c = getopt_long(argc, argv, shortopts, longopts, NULL)) != -1) {
      if (c == 'v') { say_version(); return EXIT_SUCCESS; }
      if (c == 'h') { say_version(); fprintf(stderr, "%s\n", usage_text); return EXIT_SUCCESS; }
      for (int i = 0; i < NFLAGS; i++) {
         if (fl[i].c != c) continue;
         if (fl[i].value >= 0) return die(fl[i].twice);
         int v = fl[i].has_arg ? atoi(optarg) : 1;
         if (i == F_N) v = 0;                                    /* -n clears printline */
         if (i == F_D && v < 0) return die("error: distance must be a positive integer.\n");
         if (i == F_Y && v < 0) return die("error: memory limit must be a positive integer.\n");
         if (i == F_X && (v < 0 || v > 2)) return die("error: nondna value must be either 0, 1 or 2.\n");
         fl[i].value = v;
      }
   }
   if (optind == argc) return die("error: not enough arguments.\n");
   char *expr = argv[optind++];
   char *input = NULL;
   if (optind < argc) {
      if (optind != argc - 1) return die("error: too many options.\n");
      input = argv[optind];
   }
   /* Defaults (reference seeq-main.c:389-404): printline is on unless another body format was asked. */
   int v[NFLAGS];
   for (int i = 0; i < NFLAGS; i++) v[i] = fl[i].value;
   for (int i = 0; i < NFLAGS; i++)
      if (i != F_N && v[i] < 0) v[i] = 0;
   const int printline = v[F_N] >= 0 ? v[F_N] : (!v[F_M] && !v[F_E] && !v[F_R]);
   if (!v[F_K] && !v[F_P] && !printline && !v[F_M] && !v[F_L] && !v[F_C] && !v[F_F] && !v[F_R] && !v[F_E])
      return die("Invalid options: No output will be generated.\n");

   /* Masking (reference seeq-main.c:419-438): -c hides every format flag, -i most of them. */
   const int nocount = !v[F_C];
   const int noinvert = !v[F_I] * nocount;
   struct seeqarg_t args;
   memset(&args, 0, sizeof args);          /* the reference leaves .split uninitialised */
   args.showdist  = v[F_K] * noinvert;
   args.showpos   = v[F_P] * noinvert;
   args.showline  = v[F_L] * nocount;
   args.printline = printline * noinvert;
   args.matchonly = v[F_M] * noinvert;
   args.count     = v[F_C];
   args.compact   = v[F_F] * noinvert;
   args.dist      = v[F_D];
   args.verbose   = v[F_Z];
   args.endline   = v[F_E] * noinvert;
   args.prefix    = v[F_R] * noinvert;
   args.invert    = v[F_I] * nocount;
   args.best      = v[F_B] * noinvert;
   args.non_dna   = v[F_X];
   args.all       = v[F_A];
   args.memory    = (size_t)v[F_Y] * 1024 * 1024;
   static char obuf[1 << 20];
   setvbuf(stdout, obuf, _IOFBF, sizeof obuf);             /* nothing has been written to stdout yet */
   return seeq(expr, input, args);
}
