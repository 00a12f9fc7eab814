/*
 * experiments.c -- the one gate in front of every environment switch that can make the library compute something
 * other than the product path: timing experiments whose values are wrong, ablations of the exchange protocol, fault
 * and failure injection, test-only delays.
 *
 * Such a switch is honoured only when CFDP_EXPERIMENTS=1 is set as well, says so once on stderr when it is, and says
 * once that it has been IGNORED when the master key is missing -- a variable left over in a shell can therefore never
 * change what a run computes.  bench.py refuses to print a line while any of them is active
 * (cfdp_experiment_switches / cfdp_experiments_active), so no measurement of an experiment can pass for the product.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cfdproxy_host.h"

static const char *const g_switches[] = {
    "CFDP_EXP_OWNED_NORMALS", /* host/tiling.c: cut-face normals stored once, fetch free -- values WRONG */
    "CFDP_DEBUG_ABLATE",      /* csrc/gpu_abi.hip: ablation bits of the pushing pass (no wait / pushes / counting) */
    "CFDP_IPC_FAULT",         /* csrc/gpu_exchange.hip: boundary tiles skip their wait (fault injection) */
    "CFDP_IPC_JITTER_US",     /* csrc/gpu_exchange.hip: random idle time in front of every step (tests) */
    "CFDP_PLAN_FAIL_STAGE",   /* csrc/plan_kernels.hip: a device plan stage fails on purpose (tests) */
    "CFDP_EXP_SKIP_PRE",      /* csrc/gpu_abi.hip: the fused pass without its pass over the staged rows (EXPERIMENTS.md D.2) -- values WRONG */
    "CFDP_EXP_ASYNC_SETUP_COPIES", /* csrc/gpu_ctx.h: set-up memsets / copies do NOT wait (the defect of DESIGN C.5, for its regression test) */
};
enum { NSWITCH = sizeof g_switches / sizeof g_switches[0] };
static int g_said[NSWITCH];

static int master_key(void) {
  const char *m = getenv("CFDP_EXPERIMENTS");
  return m && atoi(m) == 1;
}

const char *cfdp_experiment_getenv(const char *name) {
  const char *v = getenv(name);
  if (!v || !*v) return NULL;
  int k = -1;
  for (int i = 0; i < NSWITCH; i++)
    if (!strcmp(name, g_switches[i])) k = i;
  if (k < 0) {
    fprintf(stderr, "Error: %s is not a registered experiment switch [%s:%d]\n", name, __FILE__, __LINE__);
    exit(EXIT_FAILURE);
  }
  const int on = master_key();
  if (!__atomic_exchange_n(&g_said[k], 1, __ATOMIC_RELAXED)) {
    if (on)
      fprintf(stderr, "[cfdp] EXPERIMENT SWITCH ACTIVE: %s=%s (CFDP_EXPERIMENTS=1) -- NOT the product path; values or timings of "
                      "this run must not be reported\n", name, v);
    else
      fprintf(stderr, "[cfdp] %s=%s IGNORED: experiment switches are honoured only with CFDP_EXPERIMENTS=1\n", name, v);
  }
  return on ? v : NULL;
}

const char *cfdp_experiment_switches(void) {
  static char list[512];
  if (!list[0]) {
    size_t n = 0;
    for (int i = 0; i < NSWITCH; i++) n += (size_t)snprintf(list + n, sizeof list - n, "%s%s", i ? " " : "", g_switches[i]);
  }
  return list;
}

int cfdp_experiments_active(char *buf, size_t len) {
  int n = 0;
  size_t at = 0;
  if (buf && len) buf[0] = 0;
  if (!master_key()) return 0;
  for (int i = 0; i < NSWITCH; i++) {
    const char *v = getenv(g_switches[i]);
    if (!v || !*v) continue;
    n++;
    if (buf && at < len) at += (size_t)snprintf(buf + at, len - at, "%s%s=%s", at ? " " : "", g_switches[i], v);
  }
  return n;
}
