/* call_election.h -- which of the threads calling an entry point enqueues its work (see call_election.c) */
#ifndef CFDP_CALL_ELECTION_H
#define CFDP_CALL_ELECTION_H

#include <pthread.h>

#include "cfdproxy_dropin.h" /* CFDP_CALLS_*, cfdp_set_call_mode */

#define CFDP_ELECT_RING 64  /* calls whose attendance is remembered */
#define CFDP_ELECT_GRACE 4  /* a call must be fully attended once this many later calls exist (the reference's
                               barrier per iteration allows two: gradients + flux) */

typedef struct cfdp_election {
  pthread_mutex_t mtx;        /* held by the performing thread between _begin and _end: calls are serialised */
  unsigned long id;           /* keys the callers' thread-local ordinals */
  unsigned long team_calls;   /* calls performed in team mode */
  unsigned long serial_calls; /* calls performed by serial callers */
  struct { unsigned long ordinal; int attended, team, kind; pthread_t performer; } ring[CFDP_ELECT_RING];
} cfdp_election;

void cfdp_elect_init(cfdp_election *el);
void cfdp_elect_destroy(cfdp_election *el);
/* 1: the calling thread performs the call and holds the mutex until cfdp_elect_end(); 0: a team mate has */
int cfdp_elect_begin(cfdp_election *el, int kind);
void cfdp_elect_end(cfdp_election *el);
int cfdp_get_call_mode(void);

#endif
