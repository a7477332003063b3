// ultr_prof.h — optional per-kernel timing (used by bench.py for the roofline line).
// Disabled by default: the launch sites pay one predictable branch.  When armed for a kernel id, that kernel is
// launched through hipExtLaunchKernelGGL with a start and a stop event: the two timestamps are taken from the
// kernel's OWN dispatch packet (what rocprofv3 --kernel-trace reports), not from extra marker packets around it.
// (hipEventRecord brackets were measured at ~5 us per pair on the stream - more than the small kernels they timed.)  The events are
// created with hipEventDisableSystemFence (round 5): with the default flags every completion was a system-scope release, ~1.5 us of
// the driver's 20-step figure per step.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

enum UltrKernelId {
  ULTR_K_FWD = 0, ULTR_K_LOSS = 1, ULTR_K_BWD = 2, ULTR_K_WGRAD = 3, ULTR_K_REDUCE = 4, ULTR_K_UPDATE = 5,
  ULTR_K_NDCG = 6,
  ULTR_K_FUSED = 7,  // forward + loss + backward in one launch (small batches)
  ULTR_K_COUNT = 8
};

extern uint32_t g_ultr_prof_mask;
extern bool g_ultr_prof_live;  // false on the steps the sampling stride skips
// (Off since round 5 - see ultr_prof_tick; ULTR_PROF_SHADOW=1 -)  Kernels that get a start/stop event pair WITHOUT being counted ("shadow"): with a sampling stride only the armed kernel of one
// step in `stride` is timed, and a timed launch whose PREDECESSOR on the stream is untimed absorbs the tail of that predecessor
// into its own interval (measured, tools/prof_mask_test.py: fused kernel 22.2 us with every kernel timed, 24.0 alone, 22.5 with
// the update launch in front of it timed as well).  So the launches in front of the armed kernel are timed too, uncounted: the
// update launch of the step before the sampled one, and the kernels of the sampled step that run before the armed one.
extern uint32_t g_ultr_prof_shadow;  // kernel-id mask for the CURRENT step
// called once at the top of ultr_train_step: decides whether THIS step is timed (every stride-th step while armed)
void ultr_prof_tick();
// reserves a sample (start/stop event pair) for kernel `kid`; false when the pool is exhausted
bool ultr_prof_take(int kid, hipEvent_t* a, hipEvent_t* b);

struct UltrProfScope {
  hipEvent_t a, b;
  bool on;
  UltrProfScope(int k, hipStream_t) : a(nullptr), b(nullptr), on(false) {
    if (g_ultr_prof_live && ((g_ultr_prof_mask >> k) & 1u)) on = ultr_prof_take(k, &a, &b);
    else if ((g_ultr_prof_shadow >> k) & 1u) on = ultr_prof_take(-1, &a, &b);
  }
};

// launch KERNEL (parenthesise template-ids) under the scope PS: timed when armed, a plain launch otherwise
#define ULTR_LAUNCH(PS, KERNEL, GRID, BLOCK, LDS, ST, ...)                                                   \
  do {                                                                                                       \
    if ((PS).on) hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, (uint32_t)(LDS), ST, (PS).a, (PS).b, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, ST, __VA_ARGS__);                                      \
  } while (0)
