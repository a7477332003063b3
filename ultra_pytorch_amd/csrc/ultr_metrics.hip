// ultr_metrics.hip — validation-side kernels: padding mask + NDCG@topn.
//
// Replaces BaseAlgorithm.remove_padding_for_metric_eval (reference base_algorithm.py:88-116) and
// ultra.utils.metrics.normalized_discounted_cumulative_gain with weights=None (metrics.py:191-265, 456-495).
// One wavefront per list; the descending sort is rank-by-counting from LDS (stable: ties keep index order,
// which is what torch's CPU sort yields for the all-equal padding scores).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ultr_hip.h"
#include "ultr_device.h"
#include "ultr_plan.h"
#include "ultr_prof.h"

#define NDCG_LPW 4
#define NDCG_MAX_TOPN 16

struct TopN {
  int n;
  int v[NDCG_MAX_TOPN];
};
// ultr_ndcg_report: the batch mean inside the SAME launch - the workgroup that arrives last at `counter` sums the per-list values in
// the order ndcg_mean_kernel uses (same bits), writes ndcg_out and, when host != NULL, the values + the sequence number into
// host-mapped memory (what the training step's report does for the loss: the host spins on one word instead of a stream
// synchronisation + device-to-host copy).  counter == NULL: per-list values only (ultr_ndcg's second launch follows).
struct NdcgTail {
  uint32_t* counter;
  float* out;
  float* host;
  uint32_t seq;
};

__global__ __launch_bounds__(NDCG_LPW * 64) void ndcg_list_kernel(const float* __restrict__ scores,
                                                                 const float* __restrict__ labels,
                                                                 const int32_t* __restrict__ docids, int64_t n_docs,
                                                                 int B, int L, TopN topn, float* __restrict__ per_list,
                                                                 int32_t* __restrict__ order_out,
                                                                 float* __restrict__ masked_out, NdcgTail tail) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sm_s = smem;                 // [LPW][L] masked + validated predictions
  float* sm_y = sm_s + NDCG_LPW * L;  // [LPW][L] validated labels
  float* sm_d = sm_y + NDCG_LPW * L;  // [LPW][L] discounted gains by predicted rank
  float* sm_i = sm_d + NDCG_LPW * L;  // [LPW][L] discounted gains by ideal rank
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * NDCG_LPW + wave;
  unsigned* arrived = reinterpret_cast<unsigned*>(sm_i + NDCG_LPW * L);  // waves of this workgroup that finished their list (ultr_ndcg_report)
  if (tail.counter != nullptr) {
    if (threadIdx.x == 0) *arrived = 0u;
    __syncthreads();  // (before the waves without a list leave)
  }
  if (b >= B) return;
  float* ms = sm_s + wave * L;
  float* my = sm_y + wave * L;
  float* md = sm_d + wave * L;
  float* mi = sm_i + wave * L;
  // pad mask, then metrics.py:251-264: invalid labels (< 0) -> label 0, prediction rowmin - 1e-6
  // (score, doc id and label of a position are requested together: the kernel is a chain of dependent round trips, not bytes)
  float mn = INFINITY;
  for (int l = lane; l < L; l += 64) {
    float s = scores[(int64_t)b * L + l];
    const float y = labels[(int64_t)l * B + b];
    if (docids != nullptr && (int64_t)docids[(int64_t)l * B + b] == n_docs) s = ULTR_PAD_SCORE;
    if (masked_out != nullptr) masked_out[(int64_t)b * L + l] = s;
    ms[l] = s;
    my[l] = y;
    mn = fminf(mn, s);
  }
  mn = -wave_max(-mn);
  for (int l = lane; l < L; l += 64) {
    const float y = my[l];
    const bool ok = y >= 0.f;
    my[l] = ok ? y : 0.f;
    if (!ok) ms[l] = -1e-6f + mn;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  for (int i = lane; i < L; i += 64) {
    const float si = ms[i], yi = my[i];
    int rs = 0, ry = 0;
    for (int j = 0; j < L; ++j) {
      const float sj = ms[j], yj = my[j];
      rs += (sj > si || (sj == si && j < i)) ? 1 : 0;
      ry += (yj > yi || (yj == yi && j < i)) ? 1 : 0;
    }
    const float gain = exp2f(yi) - 1.0f;  // weights = 1: gains = 2^label - 1 (metrics.py:213)
    md[rs] = gain * (1.0f / log2f((float)rs + 2.0f));
    mi[ry] = gain * (1.0f / log2f((float)ry + 2.0f));
    if (order_out != nullptr) order_out[(int64_t)b * L + rs] = i;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  for (int k = 0; k < topn.n; ++k) {
    const int n = topn.v[k] < L ? topn.v[k] : L;  // topn clipped to the list size (metrics.py:249)
    float d = 0.f, id = 0.f;
    for (int r = lane; r < n; r += 64) {
      d += md[r];
      id += mi[r];
    }
    d = wave_sum(d);
    id = wave_sum(id);
    if (lane == 0) {
      const float v = (id == 0.f) ? 0.f : d / id;  // _safe_div
      if (tail.counter != nullptr) coh_st1(make_src(per_list, (int64_t)B * topn.n), (unsigned)(((int64_t)b * topn.n + k) * 4), v);  // written through
      else per_list[(int64_t)b * topn.n + k] = v;
    }
  }
  if (tail.counter == nullptr) return;
  // ---- the last workgroup to arrive forms the batch means: every wave waits for its write-through stores to be acknowledged, the
  // workgroup's waves meet (waves without a list left the kernel at its top: the counter counts LISTS, a workgroup adds its own
  // number of them with one relaxed agent-scope increment - 64 increments on one address instead of 256 at config 2) ----------------
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const int lists_here = (B - (int)blockIdx.x * NDCG_LPW) < NDCG_LPW ? (B - (int)blockIdx.x * NDCG_LPW) : NDCG_LPW;
  // (the waves of a partial last workgroup that returned early never reach this point: count arrivals instead of a barrier)
  unsigned mine = 0;
  if (lane == 0) mine = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
  mine = (unsigned)__builtin_amdgcn_readfirstlane((int)mine);
  if (mine != (unsigned)lists_here - 1u) return;  // not the last wave of this workgroup
  unsigned old = 0;
  if (lane == 0) old = __hip_atomic_fetch_add(tail.counter, (unsigned)lists_here, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
  if (old + (unsigned)lists_here != (unsigned)B) return;
  const Src pl = make_src(per_list, (int64_t)B * topn.n);
  float vals[NDCG_MAX_TOPN];
  for (int k = 0; k < topn.n; ++k) {
    float s = 0.f;
    for (int bb = lane; bb < B; bb += 64) s += coh_ld1(pl, (unsigned)(((int64_t)bb * topn.n + k) * 4));  // served from the coherence point
    s = wave_sum(s);
    vals[k] = s / (float)B;
  }
  if (lane == 0) {
    for (int k = 0; k < topn.n; ++k) tail.out[k] = vals[k];
    __hip_atomic_store(tail.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch on this stream
    if (tail.host != nullptr) {
      for (int k = 0; k < topn.n; ++k) __hip_atomic_store(tail.host + k, vals[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(reinterpret_cast<uint32_t*>(tail.host) + NDCG_MAX_TOPN, tail.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void ndcg_mean_kernel(const float* __restrict__ per_list, int B, int n_topn, float* __restrict__ out) {
  // mean over the batch, lanes stride the lists, fixed-order wave reduction (one wave per cutoff)
  const int k = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int b = lane; b < B; b += 64) s += per_list[(int64_t)b * n_topn + k];
  s = wave_sum(s);
  if (lane == 0) out[k] = s / (float)B;
}

extern "C" int ultr_ndcg(const float* scores, const float* labels, const int32_t* docids, int64_t n_docs, int32_t batch,
                         int32_t list_size, const int32_t* topn, int32_t n_topn, float* ndcg_out, int32_t* order_out,
                         float* masked_out, float* ndcg_ws, void* stream) {
  if (!scores || !labels || !topn || !ndcg_out || !ndcg_ws || batch <= 0 || list_size <= 0 || n_topn <= 0 ||
      n_topn > NDCG_MAX_TOPN)
    return ULTR_E_BADARG;
  TopN t;
  t.n = n_topn;
  for (int k = 0; k < n_topn; ++k) {
    if (topn[k] <= 0) return ULTR_E_BADARG;
    t.v[k] = topn[k];
  }
  const size_t lds = ((size_t)NDCG_LPW * 4 * list_size + 4) * sizeof(float);
  if (lds > 64 * 1024) return ULTR_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  {
    UltrProfScope prof(ULTR_K_NDCG, st);
    ULTR_LAUNCH(prof, ndcg_list_kernel, dim3((batch + NDCG_LPW - 1) / NDCG_LPW), dim3(NDCG_LPW * 64), lds, st, scores, labels, docids,
                n_docs, (int)batch, (int)list_size, t, ndcg_ws, order_out, masked_out, NdcgTail{nullptr, nullptr, nullptr, 0u});
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(ndcg_mean_kernel, dim3(n_topn), dim3(64), 0, st, (const float*)ndcg_ws, (int)batch, (int)n_topn,
                     ndcg_out);
  return (int)hipGetLastError();
}

// ONE launch: per-list NDCG, batch means by the last wave to finish, optional report into host-mapped memory (include/ultr_hip.h)
extern "C" int ultr_ndcg_report(const float* scores, const float* labels, const int32_t* docids, int64_t n_docs, int32_t batch,
                                int32_t list_size, const int32_t* topn, int32_t n_topn, float* ndcg_out, int32_t* order_out,
                                float* masked_out, float* ndcg_ws, uint32_t* counter, float* host_report, uint32_t seq, void* stream) {
  if (!scores || !labels || !topn || !ndcg_out || !ndcg_ws || !counter || batch <= 0 || list_size <= 0 || n_topn <= 0 ||
      n_topn > NDCG_MAX_TOPN)
    return ULTR_E_BADARG;
  TopN t;
  t.n = n_topn;
  for (int k = 0; k < n_topn; ++k) {
    if (topn[k] <= 0) return ULTR_E_BADARG;
    t.v[k] = topn[k];
  }
  const size_t lds = ((size_t)NDCG_LPW * 4 * list_size + 4) * sizeof(float);
  if (lds > 64 * 1024) return ULTR_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  UltrProfScope prof(ULTR_K_NDCG, st);
  ULTR_LAUNCH(prof, ndcg_list_kernel, dim3((batch + NDCG_LPW - 1) / NDCG_LPW), dim3(NDCG_LPW * 64), lds, st, scores, labels, docids,
              n_docs, (int)batch, (int)list_size, t, ndcg_ws, order_out, masked_out, NdcgTail{counter, ndcg_out, host_report, seq});
  return (int)hipGetLastError();
}
