// ultr_prof.h — optional per-kernel HIP-event timing (used by bench.py for the roofline line).
// Disabled by default: the launch sites pay one predictable branch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum UltrKernelId {
  ULTR_K_FWD = 0, ULTR_K_LOSS = 1, ULTR_K_BWD = 2, ULTR_K_WGRAD = 3, ULTR_K_REDUCE = 4, ULTR_K_UPDATE = 5,
  ULTR_K_NDCG = 6, ULTR_K_COUNT = 8
};

extern uint32_t g_ultr_prof_mask;
void ultr_prof_mark(int kid, int phase, hipStream_t st);  // phase 0 = before launch, 1 = after

struct UltrProfScope {
  int kid;
  hipStream_t st;
  bool on;
  UltrProfScope(int k, hipStream_t s) : kid(k), st(s), on((g_ultr_prof_mask >> k) & 1u) {
    if (on) ultr_prof_mark(kid, 0, st);
  }
  ~UltrProfScope() {
    if (on) ultr_prof_mark(kid, 1, st);
  }
};
