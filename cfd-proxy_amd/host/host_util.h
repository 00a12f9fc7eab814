/*
 * host_util.h -- error convention + small helpers of the C host side.
 * Convention follows the reference: a failed check prints file:line and exits
 * (ASSERT -> exit(EXIT_FAILURE), reference src/error_handling.h:25-30).
 */
#ifndef CFDP_HOST_UTIL_H
#define CFDP_HOST_UTIL_H

#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define CFDP_ASSERT(x)                                                                  \
  do {                                                                                  \
    if (!(x)) {                                                                         \
      fprintf(stderr, "Error: '%s' [%s:%i]\n", #x, __FILE__, __LINE__);                 \
      exit(EXIT_FAILURE);                                                               \
    }                                                                                   \
  } while (0)

static inline void *cfdp_malloc(size_t bytes) {
  void *p = malloc(bytes ? bytes : 1);
  if (!p) {
    fprintf(stderr, "Error: out of memory (%zu bytes)\n", bytes);
    exit(EXIT_FAILURE);
  }
  return p;
}
static inline void *cfdp_calloc(size_t n, size_t sz) {
  void *p = calloc(n ? n : 1, sz ? sz : 1);
  if (!p) {
    fprintf(stderr, "Error: out of memory (%zu x %zu bytes)\n", n, sz);
    exit(EXIT_FAILURE);
  }
  return p;
}

/* counter-based hash RNG (splitmix64 finaliser): same value for the same key in every
 * domain file, so a face shared by two files gets the same normal in both            */
static inline uint64_t cfdp_mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline double cfdp_u01(uint64_t key) { /* (0,1) */
  return ((double)(cfdp_mix64(key) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

double cfdp_now(void); /* seconds, monotonic */
int cfdp_host_threads(void); /* threads for the library's own OpenMP regions: the cores really granted (dualgrid_gen.c) */

#endif
