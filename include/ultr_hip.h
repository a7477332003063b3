/*
 * ultr_hip.h — C ABI of libultr_hip.so, the MI355X (gfx950) hot path of the unbiased
 * learning-to-rank engine.
 *
 * This is the drop-in boundary.  Every entry point replaces one stretch of the
 * reference's (ULTR-Community/ULTRA_pytorch) pure-PyTorch hot path; the reference
 * file:line each one stands in for is cited on the declaration.  The Python plugin
 * classes in ultra_pytorch_amd/ (same constructor / train / validation / build
 * contracts as ultra.learning_algorithm.* and ultra.ranking_model.DNN) bind these
 * symbols with ctypes — see INTEGRATION.md for the stub a reference maintainer adds.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes.  No torch types.
 *   - ALL data pointers are DEVICE pointers owned by the caller, fp32 unless noted.
 *   - Every function enqueues work on `stream` (a hipStream_t passed as void*) and
 *     returns immediately: 0 on success, a hipError_t value (>0) on a HIP failure,
 *     or a negative ULTR_E_* code for a bad argument.  Nothing throws, nothing syncs,
 *     nothing allocates: scratch comes from caller-provided workspaces sized by the
 *     ultr_*_workspace_bytes() queries (host-only arithmetic, callable without a GPU).
 *   - Layouts follow the reference's feed (click_simulation_feed.py:141-156):
 *       features  [n_docs, F]   row-major; the PAD document has id == n_docs and is an
 *                               all-zero row that is NOT stored (base_algorithm.py:148-149)
 *       docids    [L, B] int32  position-major (docid_input{l}[b])
 *       labels    [L, B]        position-major (label{l}[b]) — clicks or relevance
 *       scores    [B, L]        list-major, what BaseAlgorithm.ranking_model returns
 *                               (base_algorithm.py:118-132)
 *     Internally a "row" is one (query, position) document: row n = b*L + l.
 *   - Parameters travel as ONE flat fp32 vector in the reference's state_dict order
 *     (DNN.py:41-55): for j = 0..k:  layer_norm{j}.weight[K_j], layer_norm{j}.bias[K_j],
 *     linear{j}.weight[M_j,K_j] row-major, linear{j}.bias[M_j]; K_0 = F, M_k = 1.
 */
#ifndef ULTR_HIP_H
#define ULTR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ULTR_ABI_VERSION 8
#define ULTR_MAX_HIDDEN 7 /* hidden layers; Linear layers = hidden + 1 <= 8 */

#define ULTR_E_BADARG (-1)
#define ULTR_E_UNSUPPORTED (-2)
#define ULTR_E_WORKSPACE (-3)
#define ULTR_E_COMM_TIMEOUT (-4) /* a peer did not arrive within the bounded wait of ultr_comm_allreduce */
/* bit of the status word of the step report (ultr_update_desc::host_scalars[8]) that is NOT a communication failure: a hidden
 * weight reached |w| >= 128, outside the range of the split-half (fp16 hi / lo, x 2^8) weight copies the wide layers' products
 * read - results from then on are not to be trusted; run with ULTR_FB_H3=0 ULTR_FWD_H3=0 ULTR_BWD_H3=0 (fp32 matrix-core
 * products).  Raised by ultr_dnn_build_wt / ultr_apply_update, reported with the step after the one that wrote the weight. */
#define ULTR_STATUS_H3_RANGE 0x100u
/* ... and the early warning: a hidden weight reached |w| >= 64, half of that range, while every copy is still exact.  An optimizer
 * step moves a weight by at most learning_rate x max_gradient_norm, so a host that reads the report every few steps has time to
 * switch THIS model to the fp32 products (ultr_dnn_desc::flags |= ULTR_MODEL_FP32_PRODUCTS) before anything overflows - engine.StepEngine does, with a
 * warning: like the reference (base_algorithm.py:208-226) the library then trains any weight magnitude. */
#define ULTR_STATUS_H3_NEAR 0x200u
/* bits of a range-flag word (ultr_update_desc::range_flag, the word inside `wt`): a scaled weight overflowed / is near the edge */
#define ULTR_H3_FLAG_OVER 1u
#define ULTR_H3_FLAG_NEAR 2u

/* base_ranking_model.py:63-69 (ACT_FUNC_DIC): elu, relu, tanh, sigmoid.  ('selu' is listed there as a plain function and
 * nn.Sequential.add_module rejects it - TypeError at DNN.py:52-53 - so it is not an option of the reference.) */
enum ultr_activation { ULTR_ACT_ELU = 0, ULTR_ACT_RELU = 1, ULTR_ACT_TANH = 2, ULTR_ACT_SIGMOID = 3 };

/* DNN.__init__ hyper-parameters (DNN.py:25-38).  norm is always 'layer'
 * (LayerNorm before EVERY Linear, DNN.py:43-47). */
typedef struct ultr_dnn_desc {
  int32_t feature_size;            /* F = K_0 */
  int32_t n_hidden;                /* k; 0 reproduces ultra.ranking_model.Linear */
  int32_t hidden[ULTR_MAX_HIDDEN]; /* hidden_layer_sizes */
  int32_t activation;              /* ultr_activation */
  /* ABI 6: per-MODEL switches (0 = defaults).  ULTR_MODEL_FP32_PRODUCTS: every product of this model on the fp32 matrix cores
   * (v_mfma_f32_16x16x4_f32), whatever the process-wide ULTR_*_H3 knobs say - what a caller sets when a hidden weight of THIS
   * model approaches the range of the split-half weight copies (ULTR_STATUS_H3_NEAR); other models of the process keep their plan. */
  int32_t flags;
} ultr_dnn_desc;
#define ULTR_MODEL_FP32_PRODUCTS 1

/* ---- host-only queries ------------------------------------------------------------- */
int ultr_abi_version(void);
/* number of fp32 parameters P (0 on bad desc) */
int64_t ultr_dnn_param_count(const ultr_dnn_desc* d);
/* offsets[4*(k+1)] of ln.weight, ln.bias, linear.weight, linear.bias per layer */
int ultr_dnn_param_offsets(const ultr_dnn_desc* d, int64_t* offsets);
/* bytes of `saved` (activations + LayerNorm statistics kept by a training forward) */
int64_t ultr_dnn_saved_bytes(const ultr_dnn_desc* d, int64_t n_rows);
/* bytes of `bwd_ws` (dz buffers + deterministic partial-sum slabs) */
int64_t ultr_dnn_bwd_workspace_bytes(const ultr_dnn_desc* d, int64_t n_rows);
/* floats in the step vector that follows the P gradients in `grads` (see ultr_dnn_backward) */
int64_t ultr_step_tail_floats(int32_t list_size);
/* bytes of `loss_ws` for B lists of size L */
int64_t ultr_loss_workspace_bytes(int64_t batch, int32_t list_size);
/* step-tail partials the stand-alone ultr_*_loss kernels leave in `loss_ws` (one per workgroup) = the n_loss_parts of
 * ultr_setrank_backward */
int64_t ultr_loss_part_count(int64_t batch);

/* ---- a2 + a3: gather + DNN forward --------------------------------------------------
 * Replaces BaseAlgorithm.get_ranking_scores + ranking_model (base_algorithm.py:118-154)
 * and DNN.build (DNN.py:58-88): np.take of feature rows (zero PAD row for id == n_docs),
 * then [LayerNorm -> Linear -> act] x k -> LayerNorm -> Linear(.,1), scores as [B, L].
 * saved == NULL: inference (validation) forward; otherwise activations and LayerNorm
 * statistics are kept for ultr_dnn_backward. */
int ultr_dnn_forward(const ultr_dnn_desc* d, const float* params, const float* wt, const float* features,
                     int64_t n_docs, const int32_t* docids, int32_t batch, int32_t list_size, float* scores,
                     void* saved, void* stream);

/* k-major copy of the hidden Linear weights (WT_j = W_j^T): lets the forward stream MFMA B-fragments as
 * 256-byte contiguous pieces.  `wt` has ultr_dnn_wt_floats(d) floats; build it once after the parameters
 * were written from outside (load_state_dict / init), ultr_apply_update keeps it current afterwards.
 * Passing wt == NULL to ultr_dnn_forward selects the (slower) generic path that reads W directly. */
int64_t ultr_dnn_wt_floats(const ultr_dnn_desc* d);
int ultr_dnn_build_wt(const ultr_dnn_desc* d, const float* params, float* wt, void* stream);
/* ABI 5.  Range state of the split-half (fp16 hi / lo) weight copies inside `wt` (ULTR_STATUS_H3_NEAR / _RANGE above):
 * 0 = every hidden weight is below 64 in magnitude, 1 = near the edge (|w| >= 64, copies still exact), 2 = a copy overflowed
 * (|w| >= 128).  SYNCHRONISES the stream (one 4-byte device-to-host copy): meant to be called once behind ultr_dnn_build_wt,
 * i.e. after parameters arrived from outside (checkpoint, init) - the reference accepts any weight there (DNN.py:58-88 computes
 * in fp32), so the caller switches to the fp32 products (ULTR_FB_H3=0 ULTR_FWD_H3=0 ULTR_BWD_H3=0 + ultr_config_reload) when
 * this is not 0.  During training the same information arrives with the step report (host_scalars[8]). */
int ultr_dnn_wt_range(const ultr_dnn_desc* d, const float* wt, void* stream);
/* Which forward kernel ultr_dnn_forward launches for n_rows = batch x list_size rows when it is given aligned operands and the
 * weight copies (host-only, for tests and the bench line): 16 / 32 = dnn_fwd_kernel with that many rows per workgroup,
 * 1000 + R = the wide-tile kernel dnn_fwdw_kernel with R = 17 .. 48 rows per workgroup (round 5), 0 = the per-layer path
 * (training forward only), < 0 = bad descriptor. */
int32_t ultr_dnn_forward_tile_rows(const ultr_dnn_desc* d, int64_t n_rows, int32_t training);
/* ... and the row-local backward kernel of ultr_train_step (aligned operands, dscores from a loss kernel): 16 / 32 = dnn_bwd2_kernel /
 * dnn_bwd_kernel, 1000 + R = the wide-tile kernel dnn_bwdw_kernel, 0 = the per-layer path. */
int32_t ultr_dnn_backward_tile_rows(const ultr_dnn_desc* d, int64_t n_rows);

/* ---- a5 (backward half): what loss.backward() does for the DNN ----------------------
 * Replaces autograd through DNN.sequential (called from BaseAlgorithm.opt_step,
 * base_algorithm.py:208-226).  Input dscores[B, L] = d(loss)/d(scores) up to the scalar
 * factor the loss kernels keep separate (see tail).  Output grads[P + tail]: the flat
 * parameter gradient (UNSCALED) followed by the step vector `tail` =
 *   [0] loss_sum  [1] D  [2] loss2_sum  [3] D2  [4..4+2L) per-position sums
 * copied (summed in fixed order) from the loss workspace `loss_ws`.  Per-block partial sums of
 * squares of the P unscaled gradients are left at the head of bwd_ws (single-GPU fast path).
 * The whole buffer grads[0 .. P+tail) is what a data-parallel caller all-reduces; it then calls
 * ultr_grad_sumsq to refresh the partials before ultr_apply_update. */
int ultr_dnn_backward(const ultr_dnn_desc* d, const float* params, const float* features, int64_t n_docs,
                      const int32_t* docids, int32_t batch, int32_t list_size, const void* saved,
                      const float* dscores, const void* loss_ws, void* bwd_ws, float* grads, void* stream);

/* Same, with the NA / IPW loss (ultr_softmax_ce) FUSED into the backward kernel's prologue: takes the scores
 * and the feed's labels instead of dscores.  dscores_out may be NULL.  Identical results to
 * ultr_softmax_ce + ultr_dnn_backward (tests/test_gpu_parity.py::test_fused_softmax_backward). */
int ultr_dnn_backward_softmax(const ultr_dnn_desc* d, const float* params, const float* features, int64_t n_docs,
                              const int32_t* docids, int32_t batch, int32_t list_size, const void* saved,
                              const float* scores, const float* labels, const float* pw, const float* ipw_table,
                              int32_t n_ipw, float* dscores_out, void* loss_ws, void* bwd_ws, float* grads, void* stream);

/* partial sums of squares of grads[0..P) -> head of bwd_ws; only needed after an all-reduce */
int ultr_grad_sumsq(float* grads, int64_t n_params, int32_t list_size, void* bwd_ws, void* stream);

/* ---- a4 / a7 / a8: listwise softmax cross entropy (NA, IPW) -------------------------
 * Replaces BaseAlgorithm.softmax_loss + softmax_cross_entropy_with_logits
 * (base_algorithm.py:18-30, 309-330) and, when ipw_table != NULL, the per-list Python loop
 * over BasicPropensityEstimator.getPropensityForOneList (ipw_rank.py:115-128,
 * propensity_estimator.py:22-42):  pw[b,l] = labels[l,b] > 0 ? ipw_table[min(l,n_ipw-1)] : 0.
 * pw (explicit [B, L] weights) and ipw_table may both be NULL (NA: weights = 1).
 * Writes dscores[b,l] = softmax(s_b)_l * S_b - w_bl   (the gradient times the global
 * normaliser D = sum w) and per-workgroup partial sums of (loss_sum, D) into loss_ws. */
int ultr_softmax_ce(const float* scores, const float* labels, const float* pw, const float* ipw_table,
                    int32_t n_ipw, int32_t batch, int32_t list_size, float* dscores, void* loss_ws, void* stream);

/* ---- a9: DLA dual loss ---------------------------------------------------------------
 * Replaces DLA.train's loss section (dla.py:196-237) + DenoisingNet.forward (dla.py:33-48)
 * + get_normalized_weights (dla.py:287-306).  prop_params = [W(L) | bias] of the
 * DenoisingNet's Linear(L,1).  logits_to_prob: 0 softmax, 1 sigmoid (dla.py:21-22).
 * dscores = rank-loss gradient x D_rank (the ranker_loss_weight is applied by the update);
 * loss_ws gets (rank_loss_sum, D_rank, exam_loss_sum, D_exam) and the per-position sums of
 * d(exam_loss)/d(propensity) x D_exam. */
int ultr_dla_loss(const float* scores, const float* labels, const float* prop_params, int32_t logits_to_prob,
                  int32_t batch, int32_t list_size, float* dscores, void* loss_ws, void* stream);

/* ---- a10: PairDebias pairwise loss ---------------------------------------------------
 * Replaces the 2-level Python pair loop of PairDebias.train (pairwise_debias.py:142-157)
 * + pairwise_cross_entropy_loss (base_algorithm.py:228-248), including the reference's xB
 * broadcast inflation; `batch_total` is the GLOBAL batch size (== batch on one GPU).
 * loss_ws gets loss_sum and the per-position t_plus_loss / t_minus_loss sums. */
int ultr_pairdebias_loss(const float* scores, const float* labels, const float* t_plus, const float* t_minus,
                         int32_t batch, int32_t list_size, int32_t batch_total, float* dscores, void* loss_ws,
                         void* stream);

/* ---- a11: LambdaRank -----------------------------------------------------------------
 * Replaces LambdaRank.train's loss section (lambda_rank.py:116-135) + dcg/compute_delta_ndcg
 * (lambda_rank.py:247-291): per-list descending sort, pairwise delta-NDCG-weighted
 * BCE-with-logits on sigma(s_i - s_j), batch-global natural-log IDCG (kept separate as D). */
int ultr_lambdarank_loss(const float* scores, const float* labels, const float* t_plus, const float* t_minus,
                         float sigma, int32_t batch, int32_t list_size, float* dscores, void* loss_ws,
                         void* stream);

/* ---- next row 8f.3: RegressionEM -----------------------------------------------------
 * Replaces RegressionEM.train's estimation + loss section (regression_EM.py:128-151, get_bernoulli_sample
 * :20-34): gamma = sigmoid(s); posteriors with the propensity [L]; pseudo-labels ceil(p_r1 - u) with u from
 * `uniforms` [B, L] (teacher-forced) or, when NULL, Philox keyed by (seed, step); BCE-with-logits, mean over
 * B*L (the count is left in the tail as D).  loss_ws also gets the per-position sums of the M-step
 * (regression_EM.py:180-183), applied by ultr_apply_update (aux = propensity).  pseudo_labels_out may be NULL. */
int ultr_regem_loss(const float* scores, const float* labels, const float* propensity, const float* uniforms,
                    uint64_t seed, uint64_t step, int32_t batch, int32_t list_size, float* dscores,
                    float* pseudo_labels_out, void* loss_ws, void* stream);

/* ---- next row 8f.1: the SetRank ranking model --------------------------------------------
 * Replaces SetRank.build / Encoder.forward (ranking_model/SetRank.py:143-156, 229-255) and its autograd
 * backward: input LayerNorm (eps 1e-6) -> FFN(F -> dff -> d_model) -> num_layers x [multi-head self-attention
 * WITHOUT projections (heads = slices of x) -> dense -> residual + LayerNorm -> FFN -> residual + LayerNorm]
 * -> FFN(d_model -> dff -> 1).  fp32 (attention optionally with fp16 operands, see attention_dtype); every kernel is
 * hand-written: the Linear layers run on the library's LDS-tiled matrix-core GEMM with fused bias / ReLU epilogues.
 * params: flat vector in SetRank.state_dict() order (Encoder_layer.input_layer_norm, input_embedding.{0,2},
 * output_layer.{0,2}, then per encoder: mha.dense, ffn.{0,2}, layernorm1, layernorm2).
 * forward: scores [B, L]; `saved` (ultr_setrank_saved_bytes) receives every activation the backward needs and is
 * required for validation too (it doubles as scratch).  list_size <= 256.
 * backward: dscores [B, L] from any ultr_*_loss kernel (x D convention), loss_ws/n_loss_parts = that kernel's
 * partials; writes grads [P + step tail]; follow with ultr_grad_sumsq + ultr_apply_update(wt = NULL).
 * list_size <= 128 on the matrix-core attention path (head depth 16 / 32 / 64), <= 120 otherwise. */
enum ultr_attention_dtype { ULTR_ATTN_FP32 = 0, ULTR_ATTN_FP16 = 1 };
typedef struct ultr_setrank_desc {
  int32_t feature_size, d_model, num_heads, num_layers, dff;
  /* operand type of the self-attention products (ABI 2).  ULTR_ATTN_FP32 (default): exact fp32 matrix cores, the 1e-5
   * parity path.  ULTR_ATTN_FP16: fp16 operands with fp32 accumulation on v_mfma_f32_16x16x32_f16 (what BASELINE
   * config 5 names); scores / softmax / gradients algebra stay fp32; parity is ORDERING-level (~1e-3 relative), so it is
   * opt-in.  Needs head depth 32 or 64 and list_size <= 128, otherwise the fp32 kernels run. */
  int32_t attention_dtype;
  int32_t flags;  /* ABI 6: ULTR_MODEL_FP32_PRODUCTS - the Linear products and d x d weight gradients of this model on the fp32 matrix cores */
} ultr_setrank_desc;
int64_t ultr_setrank_param_count(const ultr_setrank_desc* c);
int64_t ultr_setrank_saved_bytes(const ultr_setrank_desc* c, int64_t n_rows);
int64_t ultr_setrank_workspace_bytes(const ultr_setrank_desc* c, int64_t n_rows);
/* ABI 6: float offset into `saved` of the word ultr_setrank_forward raises ULTR_H3_FLAG_* bits in when it builds the split-half
 * planes of the weights (zeroed by every forward; ultr_update_desc::range_flag); -1 on a bad desc */
int64_t ultr_setrank_range_flag_offset(const ultr_setrank_desc* c, int64_t n_rows);
int ultr_setrank_forward(const ultr_setrank_desc* c, const float* params, const float* features, int64_t n_docs,
                         const int32_t* docids, int32_t batch, int32_t list_size, float* scores, void* saved, void* stream);
int ultr_setrank_backward(const ultr_setrank_desc* c, const float* params, int32_t batch, int32_t list_size, const void* saved,
                          const float* dscores, const void* loss_ws, int32_t n_loss_parts, void* workspace, float* grads,
                          void* stream);

/* ---- a5 (clip) + a6 (optimizer) + EM / propensity updates ----------------------------
 * Replaces torch.nn.utils.clip_grad_norm_ + Adagrad.step / SGD.step
 * (base_algorithm.py:223-226; ipw_rank.py:96), DLA.separate_gradient_update
 * (dla.py:141-177: per-model clip, fresh = stateless Adagrad), and the t_plus/t_minus EM
 * updates (pairwise_debias.py:159-163, lambda_rank.py:136-142). */
enum ultr_algo { ULTR_ALGO_SOFTMAX = 0, ULTR_ALGO_DLA = 1, ULTR_ALGO_PAIRDEBIAS = 2, ULTR_ALGO_LAMBDARANK = 3, ULTR_ALGO_REGEM = 4 };
enum ultr_opt { ULTR_OPT_ADAGRAD = 0, ULTR_OPT_SGD = 1 };

typedef struct ultr_update_desc {
  int32_t algo;            /* ultr_algo */
  int32_t optimizer;       /* ultr_opt (grad_strategy 'ada' / 'sgd') */
  int32_t list_size;       /* L */
  int32_t logits_to_prob;  /* DLA only */
  int64_t n_params;        /* P */
  float learning_rate;
  float max_gradient_norm; /* <= 0 disables clipping */
  float adagrad_eps;       /* 1e-10 */
  float ranker_loss_weight;     /* DLA */
  float propensity_learning_rate; /* DLA */
  float em_step_size;      /* PairDebias / LambdaRank */
  float regulation_p;      /* PairDebias / LambdaRank */
  /* l2_loss hyper-parameter (ipw_rank.py:154-157, navie_algorithm.py:109-114, pairwise_debias.py:166-169,
   * regression_EM.py:166-169, dla.py:146-150): loss += l2_loss * sum_p ||p||^2 / 2 over the ranking model's parameters, i.e.
   * g += l2_loss * p.  As in the reference, l2_loss > 0 DISABLES the gradient clip for every algorithm but DLA (the
   * `params` generator handed to clip_grad_norm_ is already exhausted by the L2 loop); DLA adds the term
   * to rank_loss (so it is scaled by ranker_loss_weight) and clips the full gradient. */
  float l2_loss;
  /* ---- ABI 4: per-step reporting / safety (all optional) ----
   * guard: device word; when it is non-zero at launch the update changes NOTHING (parameters, optimizer state, aux stay as
   *   they are) - ultr_train_step points it at the communicator's status word, so a rank whose gradient exchange timed
   *   out (or that was told so by a peer) never applies a partly reduced gradient.
   * host_scalars: 16 floats of HOST-mapped pinned memory (device-accessible pointer): block 0 writes scalars_out[0..8) to
   *   [0..8), then the guard word to [8] and `seq` to [9] and [10] (as uint32, system-scope stores, seq LAST).  The host reads
   *   the step scalars by spinning on [9] == seq instead of a stream synchronisation + device-to-host copy.  [10] == seq means
   *   "the LOSS of step seq is in [0]": ultr_train_step raises it EARLIER where it can - single GPU, l2_loss = 0: from the
   *   weight-gradient launch, as soon as the loss is final, while that step's reduction and update still run (the reference's
   *   loss.item() then costs no pipeline bubble: the next step queues behind the update on the stream). */
  const uint32_t* guard;
  float* host_scalars;
  uint32_t seq;
  uint32_t pad_;
  /* ABI 6 - range_flag: optional device word of ULTR_H3_FLAG_* bits raised by whatever builds split-half (fp16 hi / lo) weight
   *   planes for the caller's model; block 0 of the update reports them in host_scalars[8] as ULTR_STATUS_H3_NEAR / _RANGE.
   *   ultr_train_step / ultr_apply_update with `wt` find the DNN's own word themselves (NULL is fine); a SetRank caller passes
   *   saved + ultr_setrank_range_flag_offset(). */
  const uint32_t* range_flag;
} ultr_update_desc;

/* params/state [P] updated in place; grads = the buffer ultr_dnn_backward filled (possibly
 * all-reduced).  aux = prop_params[L+1] (DLA) or [t_plus(L) | t_minus(L)] (PairDebias /
 * LambdaRank), updated in place; NULL for SOFTMAX.  d + wt (both may be NULL): keep the k-major weight copy
 * in sync with the updated parameters.  bwd_ws = the workspace ultr_dnn_backward (or
 * ultr_grad_sumsq) left the sum-of-squares partials in.
 * scalars_out[16]: [0] loss [1] ranker grad norm (pre-clip) [2] clip coef [3] D
 *                  [4] rank_loss (DLA) [5] exam_loss (DLA) [6] propensity grad norm (DLA) [7] sum g^2
 *                  [8] sum p^2  [9] sum g.p  (written by a pre-pass only when l2_loss > 0)  [10..16) reserved */
int ultr_apply_update(const ultr_update_desc* u, const ultr_dnn_desc* d, float* params, float* wt, float* state,
                      const float* grads, float* aux, const void* bwd_ws, float* scalars_out, void* stream);

/* ---- one call per training step -----------------------------------------------------------
 * ultr_dnn_forward -> ultr_<loss by upd->algo> -> ultr_dnn_backward -> ultr_apply_update, enqueued by ONE host
 * call (what `model.train(input_feed)` does between marshalling the feed and `loss.item()`).  A data-parallel
 * caller sets skip_update, all-reduces `grads`, then calls ultr_grad_sumsq + ultr_apply_update itself.
 * aux: prop_params (DLA) or [t_plus | t_minus] (PairDebias / LambdaRank) or propensity [L] (RegressionEM) or NULL. */
typedef struct ultr_step_args {
  const ultr_dnn_desc* desc;
  const ultr_update_desc* upd;
  float* params;
  float* wt;
  float* state;
  float* aux;
  const float* features;
  const int32_t* docids;
  const float* labels;
  const float* pw;
  const float* ipw_table;
  float* scores;
  float* dscores;
  void* saved;
  void* loss_ws;
  void* bwd_ws;
  float* grads;
  float* scalars;
  int64_t n_docs;
  int32_t n_ipw;
  int32_t batch;
  int32_t list_size;
  int32_t batch_total; /* PairDebias: global batch (0 = batch) */
  int32_t skip_update;
  float sigma;         /* LambdaRank */
  const float* uniforms; /* RegressionEM: [B, L] uniforms of the Bernoulli draw, or NULL = Philox(rng_seed, rng_step) */
  uint64_t rng_seed;
  uint64_t rng_step;
  /* data parallel (ABI 3): with a communicator the call runs the WHOLE sharded step - backward, then ultr_comm_allreduce of
   * grads[P + tail] (step number comm_step: the caller counts), then the update - instead of stopping for the host to issue
   * the exchange and the update as two more calls (skip_update = 1 keeps that protocol for a process-group all-reduce). */
  struct ultr_comm* comm;
  uint64_t comm_step;
} ultr_step_args;
int ultr_train_step(const ultr_step_args* a, void* stream);

/* ---- a12 + a13: validation metrics ---------------------------------------------------
 * Replaces remove_padding_for_metric_eval (base_algorithm.py:88-116) and
 * normalized_discounted_cumulative_gain (metrics.py:191-265, 456-495):  pads
 * (docid == n_docs) -> -100000, labels < 0 -> label 0 / score rowmin-1e-6, stable descending
 * sort, gains 2^l - 1, log2 discounts, topn clipped to L, mean over the batch.
 * ndcg_out[n_topn]; order_out (may be NULL) [B, L] int32 = the descending permutation;
 * masked_out (may be NULL) [B, L] = the masked scores.  ndcg_ws: batch*n_topn floats. */
int ultr_ndcg(const float* scores, const float* labels, const int32_t* docids, int64_t n_docs, int32_t batch,
              int32_t list_size, const int32_t* topn, int32_t n_topn, float* ndcg_out, int32_t* order_out,
              float* masked_out, float* ndcg_ws, void* stream);
/* ABI 7: the same as ONE launch with the result in host-mapped memory.  The wave that finishes last (a device counter the caller
 * provides zeroed once: 4 bytes, reset by every launch) sums the per-list values in ultr_ndcg's order (identical bits), writes
 * ndcg_out and - host_report != NULL: a pinned, device-mapped page of >= 17 floats - host_report[0 .. n_topn) followed by the word
 * host_report[16] = seq.  The host then reads a validation batch's metrics by spinning on that word instead of a stream
 * synchronisation + device-to-host copy (the reference's `.item()` per metric, ipw_rank.py:204-210: 18 us per batch at config 2). */
int ultr_ndcg_report(const float* scores, const float* labels, const int32_t* docids, int64_t n_docs, int32_t batch,
                     int32_t list_size, const int32_t* topn, int32_t n_topn, float* ndcg_out, int32_t* order_out,
                     float* masked_out, float* ndcg_ws, uint32_t* counter, float* host_report, uint32_t seq, void* stream);
/* ABI 8: validation() as ONE host call: ultr_dnn_forward (saved = NULL) followed by ultr_ndcg_report of its scores on the same stream
 * (base_algorithm.py:88-154 validation forward + metrics.py:456-495).  Same arguments, same results. */
int ultr_dnn_forward_ndcg(const ultr_dnn_desc* d, const float* params, const float* wt, const float* features, int64_t n_docs,
                          const int32_t* docids, const float* labels, int32_t batch, int32_t list_size, float* scores,
                          const int32_t* topn, int32_t n_topn, float* ndcg_out, int32_t* order_out, float* masked_out,
                          float* ndcg_ws, uint32_t* counter, float* host_report, uint32_t seq, void* stream);

/* ---- next row: device-side click simulation + batch assembly -------------------
 * Counterpart of ClickSimulationFeed.get_batch (click_simulation_feed.py:70-174) + PositionBiasedModel
 * (click_models.py:68-110) for a dataset RESIDENT in HBM: lists [n_queries, lmax] int32 (doc index, -1 = pad),
 * labels [n_queries, lmax] relevance.  Draws `batch` queries uniformly, samples PBM clicks
 * (exam_prob[min(l, n_exam-1)] * click_prob[min(label, n_rel-1)]), redraws lists without a click (up to
 * max_tries), and writes docids [L, B] (global doc ids, PAD = n_docs) + clicks [L, B] ready for ultr_train_step
 * with features = the resident matrix.  query_idx (may be NULL) [B] = the sampled queries.  Counter-based RNG:
 * the batch is a pure function of (seed, step).  Parity with the Python feed is distributional.
 * ABI 6: click_model = ULTR_CLICK_PBM (click_models.py:68-110) or ULTR_CLICK_CASCADE (:187-236: the same draw per position, every
 * position behind the first click reports no click) or ULTR_CLICK_UBM.  A changing bias severity (dynamic_bias_eta_change, click_simulation_feed.py:
 * 165-172) is the caller's new exam_prob table. */
#define ULTR_CLICK_PBM 0
#define ULTR_CLICK_CASCADE 1
#define ULTR_CLICK_UBM 2 /* user-browsing model (click_models.py:113-186): exam_prob = dense [n_exam][n_exam] image of the triangular
                          * table exam[rank][distance - 1] (distance to the last click; entries beyond the diagonal unused) */
int ultr_click_batch(const int32_t* lists, const float* labels, int64_t n_queries, int32_t lmax, int64_t n_docs,
                     const float* exam_prob, int32_t n_exam, const float* click_prob, int32_t n_rel, int32_t click_model,
                     uint64_t seed, uint64_t step, int32_t batch, int32_t list_size, int32_t max_tries, int32_t* docids,
                     float* clicks, int32_t* query_idx, void* stream);
/* ABI 7: the same call from a cached argument block (the host fills it once per feed and only advances `step`), and ONE host call
 * for what `train(input_feed)` of a plugin algorithm does with a DeviceClickFeed batch: the step on the batch that is already
 * drawn (exactly ultr_train_step(a, stream)) and, behind it on the same stream, the draw of the NEXT batch into the feed's other
 * buffer (`next` may be NULL) - the draw depends on (seed, step) only, so it runs under the step's reduction / update while the host
 * is still reading this step's loss (reference: main.py:153-156 get_batch + train per step; click_simulation_feed.py:101-174).
 * ULTR_CLICK_UBM needs n_exam >= 2 (its table is rank x distance); ULTR_E_BADARG otherwise. */
typedef struct ultr_click_args {
  const int32_t* lists;
  const float* labels;
  int64_t n_queries, n_docs;
  const float* exam_prob;
  const float* click_prob;
  int32_t lmax, n_exam, n_rel, click_model;
  uint64_t seed, step;
  int32_t batch, list_size, max_tries, pad_;
  int32_t* docids;
  float* clicks;
  int32_t* query_idx;
} ultr_click_args;
int ultr_click_batch_args(const ultr_click_args* c, void* stream);
int ultr_feed_train_step(const ultr_step_args* a, const ultr_click_args* next, void* stream);

/* ---- e: data-parallel gradient exchange over xGMI -----------------------------
 * No reference counterpart: the reference is single-process.  One process per GPU; queries shard across ranks,
 * parameters / optimizer / EM state are replicated, and ONE sum per step of the flat vector
 * grads[P + tail] (what ultr_dnn_backward / ultr_train_step(skip_update) leave) makes every rank's update identical.
 * ultr_comm_allreduce is that sum as ONE kernel: every rank publishes its vector into a fine-grained exchange
 * buffer mapped by all peers (hipIpc handles), reads every peer's copy over xGMI, adds them in rank order (bitwise
 * identical results on all ranks, deterministic) and leaves the sum-of-squares partials of the reduced gradient
 * where ultr_apply_update expects them (so no ultr_grad_sumsq pass).  Waits are bounded; a timeout is reported by
 * ultr_comm_status (ULTR_E_COMM_TIMEOUT) and the caller falls back to its collective library.
 * The exchange buffer is the one allocation this library makes (it must be IPC-exportable fine-grained memory);
 * handles are ULTR_COMM_HANDLE_BYTES opaque bytes the caller moves between processes (parallel.py: all_gather).
 *   create(rank, world, n_floats)  n_floats >= P + tail; zeroed flags, device-synchronising
 *   export(handle_out) / import(peer, handle)   once per peer, before the first all-reduce
 *   allreduce(step, src, n, n_params, out, sumsq_ws, sumsq_parts, stream)   step = 0, 1, 2, ... in lockstep on all
 *       ranks; src/out [n] local device vectors (may alias); sumsq_ws = head of bwd_ws, sumsq_parts =
 *       ceil((P + tail) / 64) as for ultr_grad_sumsq.  world == 1 degenerates to copy + partials. */
#define ULTR_COMM_MAX_WORLD 8
#define ULTR_COMM_HANDLE_BYTES 64
typedef struct ultr_comm ultr_comm;
int ultr_comm_create(int32_t rank, int32_t world, int64_t n_floats, ultr_comm** out);
int ultr_comm_export(ultr_comm* c, void* handle_out);
int ultr_comm_import(ultr_comm* c, int32_t peer, const void* handle);
int ultr_comm_allreduce(ultr_comm* c, uint64_t step, const float* src, int64_t n, int64_t n_params, float* out,
                        void* sumsq_ws, int32_t sumsq_parts, void* stream);
/* synchronises `stream`, then 0 or ULTR_E_COMM_TIMEOUT */
int ultr_comm_status(ultr_comm* c, void* stream);
int ultr_comm_destroy(ultr_comm* c);

/* ---- measurement hooks (bench.py only; not part of the reference's interface) ----------
 * Per-kernel timers: an armed kernel is launched with a start and a stop event taken from its OWN dispatch
 * packet (hipExtLaunchKernelGGL), i.e. the duration rocprofv3 --kernel-trace reports, without marker packets
 * on the stream.  kernel ids: 0 forward, 1 loss, 2 backward (dgrad chain), 3 weight gradients, 4 gradient
 * reduction, 5 update.  enable(mask, n) arms up to n samples for the kernels in mask (0 disarms);
 * set_stride(k) times only every k-th launch of an armed kernel; collect() synchronises the recorded events,
 * returns per-kernel total milliseconds and sample counts in arrays of 8, and rearms. */
int ultr_prof_enable(uint32_t kernel_mask, int32_t max_samples);
int ultr_prof_set_stride(int32_t every_nth_launch);
int ultr_prof_collect(double* total_ms, int64_t* counts);

/* The ULTR_* tuning knobs (README.md) are read from the environment once, at first use; this re-reads them (tests and the
 * A/B tools flip knobs inside one process). */
int ultr_config_reload(void);

#ifdef __cplusplus
}
#endif
#endif /* ULTR_HIP_H */
