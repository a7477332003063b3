// ultr_feed.hip — device-side click simulation + batch assembly (a "next" row).
//
// The reference builds every training batch in pure Python (ClickSimulationFeed.get_batch,
// click_simulation_feed.py:101-174: ~15 ms at config 2, 100x longer than the GPU step).  With the whole dataset
// resident in HBM (features [n_docs, F], initial lists [n_queries, Lmax] padded with -1, labels), a batch is just
// B x L document ids and clicks: this kernel draws them on the device and the step kernels gather the feature rows
// straight from the resident matrix (global doc ids; PAD id = n_docs).
//
// Per batch slot (one wavefront): pick a query uniformly, sample position-biased clicks
//   click_l ~ Bernoulli(exam_prob[min(l, n_exam-1)] * click_prob[min(label_l, n_rel-1)])      (click_models.py:68-110)
// - the cascade model (click_models.py:187-236) draws the same way and reports only the FIRST click of a list (a ballot over
// the wavefront's positions); the user-browsing model (click_models.py:113-186) examines rank r with a probability that depends
// on how far back the last click was (a triangular table exam[rank][distance - 1]): a wave-uniform walk down the list, one
// v_readlane per position - and, as the reference feed does with check_validation, redraw the whole list while it has no click.
// Randomness: a counter-based generator (Philox-4x32-10 keyed by (seed, step); counter = slot, attempt, position), so a
// batch is a pure function of (seed, step) - reproducible and independent of launch geometry.  It is NOT the
// Python Mersenne-Twister stream: parity with the reference feed is distributional (tests/test_gpu_feed.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ultr_hip.h"
#include "ultr_device.h"

__global__ __launch_bounds__(256) void click_batch_kernel(const int32_t* __restrict__ lists, const float* __restrict__ rel,
                                                          int64_t n_queries, int Lmax, int64_t n_docs,
                                                          const float* __restrict__ exam, int n_exam,
                                                          const float* __restrict__ cprob, int n_rel, int model, uint64_t seed,
                                                          uint64_t step, int B, int L, int max_tries,
                                                          int32_t* __restrict__ docids, float* __restrict__ clicks,
                                                          int32_t* __restrict__ qidx) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  Philox rng{(uint32_t)seed ^ (uint32_t)(step * 0x9E3779B97F4A7C15ull >> 32), (uint32_t)(seed >> 32) ^ (uint32_t)step};
  int64_t q = 0;
  for (int attempt = 0; attempt < max_tries; ++attempt) {
    uint32_t c[4] = {(uint32_t)b, (uint32_t)attempt, 0xFFFFFFFFu, 0x51ED270Bu};
    rng(c);
    q = (int64_t)((double)u01(c[0]) * (double)n_queries);  // uniform query pick (click_simulation_feed.py:126)
    if (q >= n_queries) q = n_queries - 1;
    float any = 0.f;
    bool clicked_before = false;  // cascade: a click in an earlier chunk of 64 positions
    int last_click = -1;          // user-browsing model: rank of the last click so far
    for (int l0 = 0; l0 < L; l0 += 64) {
      const int l = l0 + lane;
      float ck = 0.f;
      int32_t id = (int32_t)n_docs;
      if (l < L) {
        const int32_t d = (l < Lmax) ? lists[q * Lmax + l] : -1;
        // a PAD position counts as a label-0 document and CAN be clicked, exactly as in the reference feed
        // (click_simulation_feed.py:74-81 builds the label list with 0 for pads and samples every position)
        float y = 0.f;
        if (d >= 0) {
          id = d;
          y = rel[q * Lmax + l];
        }
        const int lab = y > 0.f ? (int)y : 0;
        uint32_t r[4] = {(uint32_t)b, (uint32_t)attempt, (uint32_t)(l >> 2), 0x2545F491u};
        rng(r);
        const float cp = cprob[lab < n_rel ? lab : n_rel - 1];
        const float u = u01(r[l & 3]);
        if (model == ULTR_CLICK_UBM) {
          ck = u / cp;  // (click iff u < exam x cp with cp > 0: the walk below compares u / cp with the examination probability)
        } else {
          ck = (u < exam[l < n_exam ? l : n_exam - 1] * cp) ? 1.f : 0.f;
        }
      }
      if (model == ULTR_CLICK_UBM) {
        // exam = dense [n_exam][n_exam] image of the triangular table (row = rank, column = distance - 1); getExamProb,
        // click_models.py:175-186: beyond the table the LAST row serves - the last entry when no click precedes the position,
        // else column distance - 1 saturating at the second-to-last
        const float ratio = ck;
        ck = 0.f;
        const int hi = (L - l0) < 64 ? (L - l0) : 64;
        for (int k = 0; k < hi; ++k) {
          const int rank = l0 + k, dist = rank - last_click;
          float ex;
          if (rank < n_exam) ex = exam[rank * n_exam + dist - 1];
          else if (dist > rank) ex = exam[(n_exam - 1) * n_exam + n_exam - 1];
          else ex = exam[(n_exam - 1) * n_exam + (dist < n_exam - 1 ? dist - 1 : n_exam - 2)];
          const float rk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ratio), k));
          const bool hit = rk < ex;
          if (hit) last_click = rank;
          if (lane == k && hit) ck = 1.f;
        }
      }
      if (model == ULTR_CLICK_CASCADE) {  // only the first click of the list counts (the draws behind it are made and ignored, as in the reference)
        const uint64_t hit = __ballot(ck > 0.f);
        const int first = hit ? (int)__builtin_ctzll(hit) : 64;
        if (clicked_before || lane > first) ck = 0.f;
        clicked_before = clicked_before || hit != 0;
      }
      if (l < L) {
        docids[(int64_t)l * B + b] = id;
        clicks[(int64_t)l * B + b] = ck;
      }
      any += ck;
    }
    if (wave_sum(any) > 0.f) break;  // lists without a click are rejected (click_simulation_feed.py:89-91)
  }
  if (lane == 0 && qidx != nullptr) qidx[b] = (int32_t)q;
}

extern "C" int ultr_click_batch(const int32_t* lists, const float* labels, int64_t n_queries, int32_t lmax, int64_t n_docs,
                                const float* exam_prob, int32_t n_exam, const float* click_prob, int32_t n_rel, int32_t click_model,
                                uint64_t seed, uint64_t step, int32_t batch, int32_t list_size, int32_t max_tries,
                                int32_t* docids, float* clicks, int32_t* query_idx, void* stream) {
  if (!lists || !labels || !exam_prob || !click_prob || !docids || !clicks || n_queries <= 0 || lmax <= 0 || batch <= 0 ||
      list_size <= 0 || n_exam <= 0 || n_rel <= 0 || max_tries <= 0 || (click_model != ULTR_CLICK_PBM && click_model != ULTR_CLICK_CASCADE && click_model != ULTR_CLICK_UBM) || n_docs < 0 || n_docs >= ((int64_t)1 << 31))
    return ULTR_E_BADARG;
  hipLaunchKernelGGL(click_batch_kernel, dim3((batch + 3) / 4), dim3(256), 0, (hipStream_t)stream, lists, labels, n_queries,
                     (int)lmax, n_docs, exam_prob, (int)n_exam, click_prob, (int)n_rel, (int)click_model, seed, step, (int)batch,
                     (int)list_size, (int)max_tries, docids, clicks, query_idx);
  return (int)hipGetLastError();
}
