/*
 * seeq_pattern.h -- host-side pattern compiler of seeq-mi355x (plain C).
 *
 *   seeq_compile_pattern : pattern text -> one key byte per position
 *                          (semantics of reference libseeq.c:511-603)
 *   seeq_build_peq       : key bytes -> Peq[5][W] bit masks for the
 *                          Myers kernels (bit i of Peq[c] set iff text class c
 *                          matches pattern position i, i.e. the test
 *                          `value & exp[i-1]` of reference libseeq.c:737,781)
 */
#ifndef SEEQ_PATTERN_H_
#define SEEQ_PATTERN_H_

#include <stdint.h>
#include <string.h>

/* seeqerr codes produced by the pattern compiler (reference libseeq.c:28-41). */
#define SEEQ_ERR_DIST          1
#define SEEQ_ERR_DOUBLE_OPEN   2
#define SEEQ_ERR_DOUBLE_CLOSE  3
#define SEEQ_ERR_ILLEGAL_CHAR  4
#define SEEQ_ERR_MISSING_CLOSE 5
#define SEEQ_ERR_DIST_GE_LEN   9

/* Key bits (reference libseeq.c:521-526): A=0x01 C=0x02 G=0x04 T/U=0x08,
 * N = 0x1F (matches every text class including text 'N', class 4). */
static inline int seeq_key_of_base(char c)
{
   switch (c | 0x20) {           /* ASCII lower-case fold */
   case 'a': return 0x01;
   case 'c': return 0x02;
   case 'g': return 0x04;
   case 't': case 'u': return 0x08;
   case 'n': return 0x1F;
   default:  return 0;
   }
}

/* Returns the number of positions, or -1 with *err = seeqerr code.
 * keys must have room for strlen(expr) bytes.  A bracket group ORs its
 * members into one position; an empty group "[]" contributes no position
 * (reference libseeq.c:573-587). */
static inline int seeq_compile_pattern(const char *expr, char *keys, int *err)
{
   const size_t n = strlen(expr);
   int npos = 0;
   int open = 0;          /* inside [...] */
   int members = 0;       /* bases seen in the current group */
   *err = 0;
   if (n) memset(keys, 0, n);
   for (size_t i = 0; i < n; i++) {
      const char c = expr[i];
      if (c == '[') {
         if (open) { *err = SEEQ_ERR_DOUBLE_OPEN; return -1; }
         open = 1;
         members = 0;
      } else if (c == ']') {
         if (!open) { *err = SEEQ_ERR_DOUBLE_CLOSE; return -1; }
         open = 0;
         if (members) npos++;        /* group complete */
         else keys[npos] = 0;        /* "[]": nothing */
      } else {
         const int k = seeq_key_of_base(c);
         if (!k || !((c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'))) {
            *err = SEEQ_ERR_ILLEGAL_CHAR;
            return -1;
         }
         keys[npos] |= (char)k;
         if (open) members++;
         else npos++;
      }
   }
   if (open) { *err = SEEQ_ERR_MISSING_CLOSE; return -1; }
   return npos;
}

static inline int seeq_words_for(int wlen) { return (wlen + 31) / 32; }

/* peq must hold 5*W words, laid out [class][word].  Rows >= wlen stay 0. */
static inline void seeq_build_peq(const char *keys, int wlen, int W, uint32_t *peq)
{
   memset(peq, 0, (size_t)(5 * W) * sizeof(uint32_t));
   for (int i = 0; i < wlen; i++)
      for (int c = 0; c < 5; c++)
         if ((keys[i] >> c) & 1) peq[c * W + (i >> 5)] |= 1u << (i & 31);
}

#endif
