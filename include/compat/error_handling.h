/* include/compat/error_handling.h -- the reference's error convention (src/error_handling.h:4-50)
 * for host code compiled unchanged against the drop-in boundary: ERR(e) = libnetcdf message +
 * exit(2); ASSERT(x) / ASSERT_INT(x, y) = message with file:line + exit(EXIT_FAILURE). */
#ifndef CFDP_COMPAT_ERROR_HANDLING_H
#define CFDP_COMPAT_ERROR_HANDLING_H
#include <stdio.h>
#include <stdlib.h>
#include "../cfdproxy_dropin.h"

#define ERRCODE 2
#define ERR(e)                                   \
  do {                                           \
    printf("Error: %s\n", nc_strerror(e));       \
    exit(ERRCODE);                               \
  } while (0)
#define ASSERT(...)                                                                    \
  do {                                                                                 \
    if (!(__VA_ARGS__)) {                                                              \
      fprintf(stderr, "Error: '%s' [%s:%i]\n", #__VA_ARGS__, __FILE__, __LINE__);      \
      exit(EXIT_FAILURE);                                                              \
    }                                                                                  \
  } while (0)
#define ASSERT_INT(x, y)                                                                               \
  do {                                                                                                 \
    if ((x) != (y)) {                                                                                  \
      fprintf(stderr, "Error: '%s' != '%s' %d != %d [%s:%i]\n", #x, #y, (x), (y), __FILE__, __LINE__); \
      exit(EXIT_FAILURE);                                                                              \
    }                                                                                                  \
  } while (0)
#endif
