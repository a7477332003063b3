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
#include "ultr_feed.h"

__global__ __launch_bounds__(256) void click_batch_kernel(ultr_click_args ca) { click_draw(ca, (int)blockIdx.x); }

thread_local const ultr_click_args* g_ultr_click_rider = nullptr;

extern "C" int ultr_click_batch(const int32_t* lists, const float* labels, int64_t n_queries, int32_t lmax, int64_t n_docs,
                                const float* exam_prob, int32_t n_exam, const float* click_prob, int32_t n_rel, int32_t click_model,
                                uint64_t seed, uint64_t step, int32_t batch, int32_t list_size, int32_t max_tries,
                                int32_t* docids, float* clicks, int32_t* query_idx, void* stream) {
  ultr_click_args c;
  c.lists = lists; c.labels = labels; c.n_queries = n_queries; c.n_docs = n_docs; c.exam_prob = exam_prob; c.click_prob = click_prob;
  c.lmax = lmax; c.n_exam = n_exam; c.n_rel = n_rel; c.click_model = click_model; c.seed = seed; c.step = step;
  c.batch = batch; c.list_size = list_size; c.max_tries = max_tries; c.pad_ = 0;
  c.docids = docids; c.clicks = clicks; c.query_idx = query_idx;
  return ultr_click_batch_args(&c, stream);
}

extern "C" int ultr_click_batch_args(const ultr_click_args* c, void* stream) {
  if (!ultr_click_args_ok(c)) return ULTR_E_BADARG;
  hipLaunchKernelGGL(click_batch_kernel, dim3((c->batch + 3) / 4), dim3(256), 0, (hipStream_t)stream, *c);
  return (int)hipGetLastError();
}
// train(input_feed) of a plugin algorithm on a DeviceClickFeed batch as ONE host call (include/ultr_hip.h): the step, and the draw of
// the next batch as extra workgroups of the step's update launch (a launch at its latency floor: the draw costs nothing there) - or as
// a launch of its own behind the step when the update launch did not take it (no weight copy, process-group exchange)
extern "C" int ultr_feed_train_step(const ultr_step_args* a, const ultr_click_args* next, void* stream) {
  if (next != nullptr && !ultr_click_args_ok(next)) return ULTR_E_BADARG;
  g_ultr_click_rider = next;
  const int rc = ultr_train_step(a, stream);
  const bool pending = g_ultr_click_rider != nullptr;
  g_ultr_click_rider = nullptr;
  if (rc != 0 || !pending) return rc;
  return ultr_click_batch_args(next, stream);
}
