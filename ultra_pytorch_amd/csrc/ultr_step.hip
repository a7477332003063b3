// ultr_step.hip — one host call per training step: forward -> loss -> backward -> (update).
// Exists to keep the HOST out of the critical path: at ~100 us of GPU work per step, four separate FFI calls from
// Python (argument marshalling + launches) cost more than a third of that.  No new device code here.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ultr_hip.h"
#include "ultr_plan.h"
#include "ultr_prof.h"

// behind the backward: (data parallel) the one-kernel exchange of grads[P + tail] with its sum-of-squares partials, then the
// update - or stop for the caller (skip_update: it runs a process-group all-reduce + ultr_grad_sumsq + ultr_apply_update)
static int finish_step(const ultr_step_args* a, void* stream) {
  if (a->comm != nullptr && !a->skip_update) {
    const int64_t P = ultr_dnn_param_count(a->desc);
    if (P <= 0) return ULTR_E_BADARG;
    const int64_t n = P + ultr_tail_len(a->list_size);
    const ultr_update_desc* u0 = a->upd;
    // (er.host is null when the weight-gradient launch of this step already exchanged the head of the tail and reported the loss)
    const EarlyReport er = g_ultr_step_xchg.er;
    if (!g_ultr_step_xchg.done) {  // (the slab reduction of this step's backward did not run the exchange itself)
      const int rc = ultr_comm_allreduce_ex(a->comm, a->comm_step, a->grads, n, P, a->grads, a->bwd_ws, (int32_t)((n + 63) / 64), stream, er);
      if (rc) return rc;
    }
    // the update behind the exchange is guarded by the communicator's status word: after a timed-out peer wait (here or on
    // any peer - the rank that times out raises the word everywhere) no replica moves its parameters again
    ultr_update_desc u = *a->upd;
    u.guard = ultr_comm_status_word(a->comm);
    // (level-2 partials: only when the slab-reduction launch exchanged the vector itself - the stand-alone exchange kernel writes level 1)
    return ultr_apply_update_ex(&u, a->desc, a->params, a->wt, a->state, a->grads, a->aux, a->bwd_ws, a->scalars,
                                g_ultr_step_xchg.done ? g_ultr_step_nsq2 : 0, stream);
  } else if (a->skip_update) {
    return 0;
  }
  return ultr_apply_update_ex(a->upd, a->desc, a->params, a->wt, a->state, a->grads, a->aux, a->bwd_ws, a->scalars, g_ultr_step_nsq2, stream);
}

thread_local EarlyReport g_ultr_early = {nullptr, 0u, 0, 1.0f};
thread_local const float* g_ultr_step_wt = nullptr;
thread_local StepXchg g_ultr_step_xchg = {nullptr, 0, {nullptr, 0u, 0, 1.0f}, false};
thread_local int g_ultr_step_nsq2 = 0;

namespace {
// early loss report for the backward call(s) of this step (EarlyReport, ultr_plan.h): only where the local loss sums ARE the
// batch's (no data-parallel exchange) and the reported loss has no L2 term (that one is formed by the update launch)
struct EarlyScope {
  explicit EarlyScope(const ultr_step_args* a) {
    const ultr_update_desc* u = a->upd;
    const bool ok = u->host_scalars != nullptr && a->comm == nullptr && !a->skip_update && u->l2_loss == 0.f;
    g_ultr_early = {ok ? u->host_scalars : nullptr, u->seq, u->algo, u->ranker_loss_weight};
    g_ultr_step_wt = a->wt;
    g_ultr_step_nsq2 = 0;
    const bool dp = a->comm != nullptr && !a->skip_update;
    const bool early_dp = dp && u->host_scalars != nullptr && u->l2_loss == 0.f;
    g_ultr_step_xchg = {dp ? a->comm : nullptr, a->comm_step, {early_dp ? u->host_scalars : nullptr, u->seq, u->algo, u->ranker_loss_weight}, false};
  }
  ~EarlyScope() {
    g_ultr_early.host = nullptr;
    g_ultr_step_wt = nullptr;
    g_ultr_step_xchg.comm = nullptr;
    g_ultr_step_nsq2 = 0;
  }
};
}  // namespace

extern "C" int ultr_train_step(const ultr_step_args* a, void* stream) {
  if (!a || !a->desc || !a->upd) return ULTR_E_BADARG;
  ultr_prof_tick();
  EarlyScope early(a);
  int rc;
  if (a->upd->algo == ULTR_ALGO_SOFTMAX) {
    // small batches (NA / IPW): forward + loss + backward as ONE launch when the shape qualifies
    rc = ultr_fused_step_softmax(a->desc, a->params, a->wt, a->features, a->n_docs, a->docids, a->batch, a->list_size,
                                 a->scores, a->saved, a->labels, a->pw, a->ipw_table, a->n_ipw, a->dscores, a->loss_ws,
                                 a->bwd_ws, a->grads, stream);
    if (rc == 0) return finish_step(a, stream);
    if (rc != ULTR_E_UNSUPPORTED) return rc;
  }
  rc = ultr_dnn_forward(a->desc, a->params, a->wt, a->features, a->n_docs, a->docids, a->batch, a->list_size,
                            a->scores, a->saved, stream);
  if (rc) return rc;
  if (a->upd->algo == ULTR_ALGO_SOFTMAX) {
    // NA / IPW: the loss is fused into the backward kernel's prologue (one launch and one dependent kernel
    // boundary fewer); ultr_softmax_ce stays available as the stand-alone stage
    rc = ultr_dnn_backward_softmax(a->desc, a->params, a->features, a->n_docs, a->docids, a->batch, a->list_size, a->saved,
                                   a->scores, a->labels, a->pw, a->ipw_table, a->n_ipw, a->dscores, a->loss_ws, a->bwd_ws,
                                   a->grads, stream);
    if (rc) return rc;
    return finish_step(a, stream);
  }
  switch (a->upd->algo) {
    case ULTR_ALGO_DLA:
      rc = ultr_dla_loss(a->scores, a->labels, a->aux, a->upd->logits_to_prob, a->batch, a->list_size, a->dscores,
                         a->loss_ws, stream);
      break;
    case ULTR_ALGO_PAIRDEBIAS:
      rc = ultr_pairdebias_loss(a->scores, a->labels, a->aux, a->aux ? a->aux + a->list_size : nullptr, a->batch,
                                a->list_size, a->batch_total > 0 ? a->batch_total : a->batch, a->dscores, a->loss_ws, stream);
      break;
    case ULTR_ALGO_LAMBDARANK:
      rc = ultr_lambdarank_loss(a->scores, a->labels, a->aux, a->aux ? a->aux + a->list_size : nullptr, a->sigma, a->batch,
                                a->list_size, a->dscores, a->loss_ws, stream);
      break;
    case ULTR_ALGO_REGEM:
      rc = ultr_regem_loss(a->scores, a->labels, a->aux, a->uniforms, a->rng_seed, a->rng_step, a->batch, a->list_size,
                           a->dscores, nullptr, a->loss_ws, stream);
      break;
    default:
      return ULTR_E_BADARG;
  }
  if (rc) return rc;
  rc = ultr_dnn_backward(a->desc, a->params, a->features, a->n_docs, a->docids, a->batch, a->list_size, a->saved,
                         a->dscores, a->loss_ws, a->bwd_ws, a->grads, stream);
  if (rc) return rc;
  return finish_step(a, stream);
}

