// ultr_feed.h - the click draw of ultr_feed.hip as a device function: its own launch (click_batch_kernel) or a rider on the update
// launch of the step in front of it (update_tiled_kernel: extra workgroups behind the update's own - ultr_feed_train_step).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ultr_hip.h"
#include "ultr_device.h"

// one workgroup of 256 threads = four batch slots (one per wave); `block` = the workgroup's index among the draw's (batch + 3) / 4
__device__ __forceinline__ void click_draw(const ultr_click_args& ca, int block) {
  const int32_t* __restrict__ lists = ca.lists;
  const float* __restrict__ rel = ca.labels;
  const int64_t n_queries = ca.n_queries, n_docs = ca.n_docs;
  const int Lmax = ca.lmax, n_exam = ca.n_exam, n_rel = ca.n_rel, model = ca.click_model, B = ca.batch, L = ca.list_size, max_tries = ca.max_tries;
  const float* __restrict__ exam = ca.exam_prob;
  const float* __restrict__ cprob = ca.click_prob;
  const uint64_t seed = ca.seed, step = ca.step;
  int32_t* __restrict__ docids = ca.docids;
  float* __restrict__ clicks = ca.clicks;
  int32_t* __restrict__ qidx = ca.query_idx;
  const int lane = threadIdx.x & 63;
  const int b = block * 4 + (threadIdx.x >> 6);
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
          // click iff u < exam x cp: the walk below compares u / cp with the examination probability; cp == 0 (a relevance level that
          // is never clicked) gives +inf or NaN, and both compare false against every probability - no click, as intended
          ck = u / cp;
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
          else ex = exam[(n_exam - 1) * n_exam + (dist < n_exam - 1 ? dist - 1 : (n_exam >= 2 ? n_exam - 2 : 0))];
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


// the draw that rides on the next update launch of this thread (set by ultr_feed_train_step around ultr_train_step; the update launch
// that takes it clears the pointer)
extern thread_local const ultr_click_args* g_ultr_click_rider;
inline bool ultr_click_args_ok(const ultr_click_args* c) {
  return c && c->lists && c->labels && c->exam_prob && c->click_prob && c->docids && c->clicks && c->n_queries > 0 && c->lmax > 0 && c->batch > 0 &&
         c->list_size > 0 && c->n_exam > 0 && c->n_rel > 0 && c->max_tries > 0 &&
         (c->click_model == ULTR_CLICK_PBM || c->click_model == ULTR_CLICK_CASCADE || c->click_model == ULTR_CLICK_UBM) && c->n_docs >= 0 &&
         c->n_docs < ((int64_t)1 << 31) && !(c->click_model == ULTR_CLICK_UBM && c->n_exam < 2);
}
