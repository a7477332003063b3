// ultr_update.hip — clip + optimizer + EM / propensity updates, one launch per step.
//
// Replaces torch.nn.utils.clip_grad_norm_ + torch.optim.Adagrad.step / SGD.step (reference
// base_algorithm.py:208-226), DLA.separate_gradient_update (dla.py:141-177) and the t_plus / t_minus EM
// updates (pairwise_debias.py:159-163, lambda_rank.py:136-142).  Every workgroup recomputes the handful of
// step scalars (global normaliser, gradient norm, clip coefficient) from the same inputs in the same order,
// so no grid barrier and no atomics are needed and all workgroups agree bit-for-bit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/ultr_hip.h"
#include "ultr_device.h"
#include "ultr_plan.h"
#include "ultr_feed.h"
#include "ultr_prof.h"

__device__ __forceinline__ float block_sum256(float v, float* sm) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  return ((sm[0] + sm[1]) + sm[2]) + sm[3];
}

__device__ __forceinline__ float opt_step(float p, float g, float* st, int opt, bool stateless, float lr, float eps) {
  if (opt == ULTR_OPT_SGD) return p - lr * g;
  // torch.optim.Adagrad (lr_decay 0, weight_decay 0, initial accumulator 0): s += g*g; p -= lr * g / (sqrt(s) + eps).
  // stateless: DLA builds fresh optimizers every step (dla.py:153-154) -> the accumulator is always g*g.
  const float s = (stateless ? 0.f : *st) + g * g;
  if (!stateless) *st = s;
  return p - lr * (g / (sqrtf(s) + eps));
}

// Step report into HOST-mapped pinned memory (ultr_update_desc::host_scalars): eight scalars + the guard word, waited for
// (the stores are acknowledged by the host bridge), THEN the sequence number the host spins on - what loss.item() costs
// becomes one posted PCIe write instead of a stream synchronisation and a device-to-host copy.
__device__ __forceinline__ void host_report(const ultr_update_desc& u, const float (&v)[8], uint32_t status) {
  float* hs = u.host_scalars;
#pragma unroll
  for (int k = 0; k < 8; ++k) __hip_atomic_store(hs + k, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(reinterpret_cast<uint32_t*>(hs) + 8, status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __hip_atomic_store(reinterpret_cast<uint32_t*>(hs) + 9, u.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // [10]: "the loss of step seq is in [0]" - already raised by the weight-gradient launch when an early report was possible
  __hip_atomic_store(reinterpret_cast<uint32_t*>(hs) + 10, u.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// This thread's share of the sum of the level-1 sum-of-squares partials, ASSOCIATED AS THE LEVEL-2 PATH WOULD: groups of four in
// order, ((p[4j] + p[4j+1]) + p[4j+2]) + p[4j+3] with missing partials = 0 (exactly what the reduction launches store as level-2
// partial j), thread t taking groups t, t + 256, ...  A data-parallel step may reach the update through the exchange inside the slab
// reduction (level-2 partials) on one rank and through the stand-alone exchange (level-1 only) on another - unequal local batches -
// and both must produce the SAME bits for the gradient norm and the clip coefficient, or the replicas fork.
__device__ __forceinline__ float sumsq_level1_as_level2(const float* __restrict__ part, int nsq, int t) {
  float ss = 0.f;
  const int ngroups = (nsq + 3) >> 2;
  for (int j = t; j < ngroups; j += 256) {
    const int k = 4 * j;
    const float p0 = part[k], p1 = k + 1 < nsq ? part[k + 1] : 0.f, p2 = k + 2 < nsq ? part[k + 2] : 0.f, p3 = k + 3 < nsq ? part[k + 3] : 0.f;
    ss += ((p0 + p1) + p2) + p3;
  }
  return ss;
}

// guard word set (a timed-out gradient exchange on this rank or on a peer): the launch must change nothing.  Block 0 still
// reports, with the status, so that the host's read of the loss raises instead of waiting for a sequence number forever.
__device__ __forceinline__ bool update_guarded(const ultr_update_desc& u) {
  if (u.guard == nullptr) return false;
  const uint32_t gv = __hip_atomic_load(u.guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (gv == 0u) return false;
  if (blockIdx.x == 0 && threadIdx.x == 0 && u.host_scalars != nullptr) {
    const float nanv = __uint_as_float(0x7fc00000u);
    const float v[8] = {nanv, nanv, 0.f, 0.f, nanv, nanv, 0.f, 0.f};
    host_report(u, v, gv);
  }
  return true;
}

// pre-pass of ultr_apply_update when l2_loss > 0: out[0] = sum p^2, out[1] = sum g.p over the P parameters (one workgroup,
// fixed order: thread-strided partials, wave sums, 16 wave partials added in order)
__global__ __launch_bounds__(1024) void l2_sums_kernel(const float* __restrict__ params, const float* __restrict__ grads,
                                                       int64_t P, float* __restrict__ out) {
  __shared__ float sm[2][16];
  float a = 0.f, b = 0.f;
  for (int64_t e = threadIdx.x; e < P; e += 1024) {
    const float p = params[e];
    a += p * p;
    b += grads[e] * p;
  }
  a = wave_sum(a);
  b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) {
    sm[0][threadIdx.x >> 6] = a;
    sm[1][threadIdx.x >> 6] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float sa = 0.f, sb = 0.f;
    for (int w = 0; w < 16; ++w) {
      sa += sm[0][w];
      sb += sm[1][w];
    }
    out[0] = sa;
    out[1] = sb;
  }
}

// Everything after the gradient and the sum of squares are known: step scalars, clip, optimizer, k-major copy /
// image, and (block 0) the per-position state.  EPT elements per thread, indices e_a (-1 = none); the new parameter
// values come back in p_new_a for the caller's k-major / image writes.
template <int EPT>
__device__ __forceinline__ void update_body(const ultr_update_desc& u, const DnnPlan& dp, float* __restrict__ params,
                                            float* __restrict__ state, const float* __restrict__ tail,
                                            float* __restrict__ aux, float ss, const int64_t (&e_a)[EPT],
                                            const float (&g_raw_a)[EPT], const float (&p_old_a)[EPT],
                                            const float (&s_old_a)[EPT], float (&p_new_a)[EPT], float* sm,
                                            float* __restrict__ scalars_out, const float* __restrict__ l2_sums,
                                            const uint32_t* __restrict__ h3flag = nullptr) {
  const int64_t P = u.n_params;
  const int L = u.list_size;
  const float loss_sum = tail[0], D = tail[1], loss2 = tail[2], D2 = tail[3];
  float gs = 1.0f, loss = loss_sum, rank_loss = 0.f, exam_loss = 0.f;
  switch (u.algo) {
    case ULTR_ALGO_SOFTMAX:  // loss = sum_b loss_b / sum w   (base_algorithm.py:330)
      gs = 1.0f / D;
      loss = loss_sum / D;
      break;
    case ULTR_ALGO_DLA:  // loss = exam_loss + ranker_loss_weight * rank_loss   (dla.py:237)
      rank_loss = loss_sum / D;
      exam_loss = loss2 / D2;
      gs = u.ranker_loss_weight / D;
      loss = exam_loss + u.ranker_loss_weight * rank_loss;
      break;
    case ULTR_ALGO_PAIRDEBIAS:
      gs = 1.0f;
      loss = loss_sum;
      break;
    case ULTR_ALGO_LAMBDARANK:  // gains are normalised by the batch-global IDCG (lambda_rank.py:280-282)
      gs = 1.0f / D;
      loss = loss_sum / D;
      break;
    case ULTR_ALGO_REGEM:  // BCEWithLogits, mean over the D = B*L elements (regression_EM.py:149-151)
      gs = 1.0f / D;
      loss = loss_sum / D;
      break;
  }
  float norm = fabsf(gs) * sqrtf(ss);
  // l2_loss > 0 (ipw_rank.py:154-157 and siblings): g += lam * p, loss += l2_loss * sum p^2 / 2.  Every algorithm but DLA
  // hands clip_grad_norm_ a parameter generator the L2 loop has already exhausted, so NOTHING is clipped (Appendix A.8);
  // DLA adds the term to rank_loss (scaled by ranker_loss_weight with it) and clips the full gradient (dla.py:146-162).
  float lam = 0.f;
  bool clip = u.max_gradient_norm > 0.f;
  if (u.l2_loss > 0.f && l2_sums != nullptr) {
    const float sp2 = l2_sums[0], sgp = l2_sums[1];
    if (u.algo == ULTR_ALGO_DLA) {
      lam = u.ranker_loss_weight * u.l2_loss;
      rank_loss += u.l2_loss * (0.5f * sp2);
      loss = exam_loss + u.ranker_loss_weight * rank_loss;
    } else {
      lam = u.l2_loss;
      loss += u.l2_loss * (0.5f * sp2);
      clip = false;
    }
    norm = sqrtf(fmaxf(gs * gs * ss + 2.0f * gs * lam * sgp + lam * lam * sp2, 0.f));
  }
  const float coef = clip ? fminf(1.0f, u.max_gradient_norm / (norm + 1e-6f)) : 1.0f;
  const bool stateless = (u.algo == ULTR_ALGO_DLA);
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int64_t e = e_a[k];  // -1: no element
    const float g_raw = g_raw_a[k], p_old = p_old_a[k], s_old = s_old_a[k];
    p_new_a[k] = 0.f;
    if (e >= 0 && e < P) {
      float g = g_raw * gs;
      if (lam != 0.f) g += lam * p_old;
      g *= coef;
      float s_new = s_old;
      const float pn = opt_step(p_old, g, &s_new, u.optimizer, stateless || state == nullptr, u.learning_rate, u.adagrad_eps);
      params[e] = pn;
      if (state != nullptr && !stateless && u.optimizer != ULTR_OPT_SGD) state[e] = s_new;
      p_new_a[k] = pn;
    }
  }
  if (blockIdx.x != 0) return;
  // ---- block 0: step scalars + the small per-position state -----------------------------------------
  float pnorm = 0.f;
  if (u.algo == ULTR_ALGO_DLA && aux != nullptr) {
    // DenoisingNet backward: propensity[b,l] = ELU(W_l + bias) for every b, so
    //   dW_l = ELU'(W_l + bias) * (1/D_exam) * sum_b d exam/d propensity[b,l],  dbias = sum_l dW_l
    const float bias = aux[L];
    float gsum = 0.f, gsq = 0.f;
    for (int l = threadIdx.x; l < L; l += 256) {
      const float z = aux[l] + bias;
      const float g = (tail[ULTR_TAIL_FIXED + l] / D2) * (z > 0.f ? 1.0f : expf(z));
      gsum += g;
      gsq += g * g;
    }
    gsum = block_sum256(gsum, sm);
    gsq = block_sum256(gsq, sm);
    pnorm = sqrtf(gsq + gsum * gsum);
    const float pc = (u.max_gradient_norm > 0.f) ? fminf(1.0f, u.max_gradient_norm / (pnorm + 1e-6f)) : 1.0f;
    for (int l = threadIdx.x; l < L; l += 256) {
      const float z = aux[l] + bias;
      const float g = (tail[ULTR_TAIL_FIXED + l] / D2) * (z > 0.f ? 1.0f : expf(z)) * pc;
      aux[l] = opt_step(aux[l], g, nullptr, u.optimizer, true, u.propensity_learning_rate, u.adagrad_eps);
    }
    __syncthreads();  // every thread has read the old bias
    if (threadIdx.x == 0) aux[L] = opt_step(bias, gsum * pc, nullptr, u.optimizer, true, u.propensity_learning_rate, u.adagrad_eps);
  } else if ((u.algo == ULTR_ALGO_PAIRDEBIAS || u.algo == ULTR_ALGO_LAMBDARANK) && aux != nullptr) {
    // t <- (1 - a) t + a * (t_loss / t_loss[0]) ^ (1 / (p + 1));   LambdaRank divides with _safe_div
    const bool safe = (u.algo == ULTR_ALGO_LAMBDARANK);
    const float ex = 1.0f / (u.regulation_p + 1.0f);
    const float a = u.em_step_size, oma = (float)(1.0 - (double)u.em_step_size);
    const float tp0 = tail[ULTR_TAIL_FIXED], tm0 = tail[ULTR_TAIL_FIXED + L];
    for (int l = threadIdx.x; l < 2 * L; l += 256) {
      const float den = (l < L) ? tp0 : tm0;
      const float num = tail[ULTR_TAIL_FIXED + l];
      const float ratio = (safe && den == 0.f) ? 0.f : num / den;
      const float pw = (ex == 0.5f) ? sqrtf(ratio) : powf(ratio, ex);
      aux[l] = oma * aux[l] + a * pw;
    }
  } else if (u.algo == ULTR_ALGO_REGEM && aux != nullptr) {
    // M-step: propensity <- (1 - a) propensity + a * mean_b(c + (1 - c) P(e=1, r=0 | c=0))   (regression_EM.py:180-183)
    const float a = u.em_step_size, oma = (float)(1.0 - (double)u.em_step_size);
    const float nb = D / (float)L;  // lists in the (global) batch
    for (int l = threadIdx.x; l < L; l += 256) aux[l] = oma * aux[l] + a * (tail[ULTR_TAIL_FIXED + l] / nb);
  }
  if (threadIdx.x == 0 && scalars_out != nullptr) {
    scalars_out[0] = loss;
    scalars_out[1] = norm;
    scalars_out[2] = coef;
    scalars_out[3] = D;
    scalars_out[4] = rank_loss;
    scalars_out[5] = exam_loss;
    scalars_out[6] = pnorm;
    scalars_out[7] = ss;
  }
  if (threadIdx.x == 0 && u.host_scalars != nullptr) {
    const float v[8] = {loss, norm, coef, D, rank_loss, exam_loss, pnorm, ss};
    // a weight outside the range of the split-half copies (DnnPlan::h3_flag_off: raised by the build kernel or by an earlier
    // update launch - this launch's own tiles report with the next step): the host's read of the loss raises
    uint32_t st = 0u;
    if (h3flag != nullptr) {
      const uint32_t f = __hip_atomic_load(h3flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      st = ((f & ULTR_H3_FLAG_OVER) ? ULTR_STATUS_H3_RANGE : 0u) | ((f & ULTR_H3_FLAG_NEAR) ? ULTR_STATUS_H3_NEAR : 0u);
    }
    host_report(u, v, st);
  }
}

__global__ __launch_bounds__(256) void update_kernel(ultr_update_desc u, DnnPlan dp, float* __restrict__ params,
                                                     float* __restrict__ state, const float* __restrict__ grads,
                                                     float* __restrict__ aux, const float* __restrict__ sumsq_part, int nsq,
                                                     float* __restrict__ scalars_out, const float* __restrict__ l2_sums, int nsq2) {
  if (update_guarded(u)) return;
  // flat variant (no k-major copy to maintain): one element per thread, loads issued BEFORE the norm reduction so
  // that the two memory round trips overlap
  __shared__ float sm[4];
  const int64_t P = u.n_params;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = e < P;
  const int64_t ea[1] = {live ? e : -1};
  const float g_raw[1] = {live ? grads[e] : 0.f};
  const float p_old[1] = {live ? params[e] : 0.f};
  const float s_old[1] = {(live && state != nullptr) ? state[e] : 0.f};
  float ss = 0.f;
  if (nsq2 > 0) {  // level-2 partials of this step's reduction launch (ultr_sumsq2_off)
    const float* s2 = sumsq_part + ultr_sumsq2_off(P);
    for (int k = threadIdx.x; k < nsq2; k += 256) ss += s2[k];
  } else {
    ss = sumsq_level1_as_level2(sumsq_part, nsq, threadIdx.x);
  }
  ss = block_sum256(ss, sm);
  float pn[1];
  update_body<1>(u, dp, params, state, grads + P, aux, ss, ea, g_raw, p_old, s_old, pn, sm, scalars_out, l2_sums, u.range_flag);
}

// Variant that keeps the k-major weight copy and the vector-parameter image current (DnnPlan::wt_*).  Workgroups
// [0, n_tiles): one 16x16 tile of a hidden W_j - thread (r, c) updates W_j[m0 + r][k0 + c] (64-byte row segments) and
// the tile goes out transposed through LDS as WT_j[k0 + r][m0 + c] (64-byte segments again; a per-element scatter wrote
// 4 bytes per 64-byte sector: 3.5 MB of write traffic for 0.4 MB of data and ~1 us of the kernel).  The remaining
// workgroups walk the vector parameters segment by segment and write the image.
// TPW tiles (or 256-element pieces of the vector parameters) per workgroup: every workgroup re-sums ALL sum-of-squares
// partials (one per 64 parameters) for its copy of the clip coefficient - at 0.5 M parameters (config 4) that is 32 KB per
// workgroup and 2050 workgroups; four units per workgroup quarter that traffic (update 14 -> 8 us there).
template <int TPW>
__global__ __launch_bounds__(256) void update_tiled_kernel(ultr_update_desc u, DnnPlan dp, float* __restrict__ params,
                                                           float* __restrict__ state, const float* __restrict__ grads,
                                                           float* __restrict__ aux, float* __restrict__ wt,
                                                           const float* __restrict__ sumsq_part, int nsq,
                                                           float* __restrict__ scalars_out, int n_tile_blocks,
                                                           const float* __restrict__ l2_sums, int nsq2, int rider_first, ultr_click_args rider) {
  __shared__ float sm[4];
  __shared__ float tile[TPW][16][17];
  // workgroups behind the update's own: the click draw of the NEXT batch (ultr_feed_train_step) - independent of the update, and of
  // its guard (a feed keeps drawing when a data-parallel step was refused)
  if ((int)blockIdx.x >= rider_first) {
    click_draw(rider, (int)blockIdx.x - rider_first);
    return;
  }
  if (update_guarded(u)) return;
  const int64_t P = u.n_params;
  const int n_tiles = dp.upd_tile_begin[dp.nl - 1];
  const int tid = threadIdx.x, r = tid >> 4, c = tid & 15;
  int64_t ea[TPW];
  int jj[TPW], m0[TPW], k0[TPW], pvpos[TPW];
  const bool is_tile = (int)blockIdx.x < n_tile_blocks;
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    ea[q] = -1; jj[q] = -1; m0[q] = 0; k0[q] = 0; pvpos[q] = -1;
    if (is_tile) {
      const int tb = (int)blockIdx.x * TPW + q;
      if (tb < n_tiles) {
        int j = 0;
        while (j + 1 < dp.nl - 1 && tb >= dp.upd_tile_begin[j + 1]) ++j;
        const int t = tb - dp.upd_tile_begin[j];
        const int tm = t / dp.upd_ntk[j];
        jj[q] = j;
        m0[q] = tm * 16;
        k0[q] = (t - tm * dp.upd_ntk[j]) * 16;
        if (m0[q] + r < dp.M[j] && k0[q] + c < dp.K[j]) ea[q] = dp.off_w[j] + (int64_t)(m0[q] + r) * dp.K[j] + k0[q] + c;
      }
    } else {
      const int v = (((int)blockIdx.x - n_tile_blocks) * TPW + q) * 256 + tid;
      if (v < dp.vs_begin[dp.n_vs]) {
        int sg = 0;
        while (sg + 1 < dp.n_vs && v >= dp.vs_begin[sg + 1]) ++sg;
        ea[q] = dp.vs_off[sg] + (v - dp.vs_begin[sg]);
        pvpos[q] = dp.vs_pv[sg] + (v - dp.vs_begin[sg]);
      }
    }
  }
  float g_raw[TPW], p_old[TPW], s_old[TPW];
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const bool live = ea[q] >= 0;
    g_raw[q] = live ? grads[ea[q]] : 0.f;
    p_old[q] = live ? params[ea[q]] : 0.f;
    s_old[q] = (live && state != nullptr) ? state[ea[q]] : 0.f;
  }
  float ss = 0.f;
  if (nsq2 > 0) {  // level-2 partials of this step's reduction launch (ultr_sumsq2_off): a quarter of the words every workgroup reads
    const float* s2 = sumsq_part + ultr_sumsq2_off(P);
    for (int k = tid; k < nsq2; k += 256) ss += s2[k];
  } else {
    ss = sumsq_level1_as_level2(sumsq_part, nsq, tid);
  }
  ss = block_sum256(ss, sm);
  float pn[TPW];
  // block 0's extra duties (EM / propensity updates, step scalars) run inside update_body and need all 256 threads
  if (blockIdx.x != 0) {
    update_body<TPW>(u, dp, params, state, grads + P, aux, ss, ea, g_raw, p_old, s_old, pn, sm, nullptr, l2_sums);
  } else {
    update_body<TPW>(u, dp, params, state, grads + P, aux, ss, ea, g_raw, p_old, s_old, pn, sm, scalars_out, l2_sums,
                     // (only a model whose weights some split-half product may READ reports: with the knobs off, or for a model
                     // switched to the fp32 products, a large weight is harmless)
                     (dp.h3_flag_off > 0 && dp.h3_watch) ? reinterpret_cast<const uint32_t*>(wt + dp.h3_flag_off) : u.range_flag);
  }
  if (is_tile) {
    // range of EVERY hidden weight (not only of the layers with fragment copies: the per-layer big-batch path builds split-half planes
    // of any hidden layer): |w| >= 64 raises NEAR, >= 128 (or NaN) OVER - reported with the next step
    if (dp.h3_flag_off > 0) {
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
        const float aw = fabsf(pn[q]) * ULTR_H3_WSCALE;
        if (ea[q] >= 0 && !(aw < ULTR_H3_WNEAR))
          flag_or(reinterpret_cast<uint32_t*>(wt + dp.h3_flag_off), !(aw < ULTR_H3_WMAX) ? (ULTR_H3_FLAG_OVER | ULTR_H3_FLAG_NEAR) : ULTR_H3_FLAG_NEAR);
      }
    }
#pragma unroll
    for (int q = 0; q < TPW; ++q) tile[q][r][c] = pn[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
      const int j = jj[q];
      if (j < 0) continue;
      const int k = k0[q] + r, m = m0[q] + c;  // transposed ownership
      if (k < dp.K[j] && m < dp.M[j]) wt[dp.wt_off[j] + (int64_t)k * dp.M[j] + m] = tile[q][c][r];
      if ((dp.h3f[j] | dp.h3b[j]) && tid >= 128) {
        // the split-half copies (DnnPlan::whf_off / whb_off, ultr_h3_index): the 16 x 16 tile is 64 pieces of 8 halves (16 bytes)
        // of each: 2 planes x 2 quarter-steps x 2 column tiles x 8 lanes; threads 128..191 write the forward copy, 192..255 the
        // dgrad copy
        const int t6 = tid & 63, hl = t6 >> 5, qq = (t6 >> 4) & 1, tt = (t6 >> 3) & 1, jj = t6 & 7;
        const bool fwd = tid < 192;
        if (fwd ? dp.h3f[j] != 0 : dp.h3b[j] != 0) {
          // forward: output column = m (tile row), contraction = k (tile column); dgrad: the other way round
          const int c0 = fwd ? m0[q] : k0[q], z0 = fwd ? k0[q] : m0[q];
          const int nks = ((fwd ? dp.K[j] : dp.M[j]) + 31) >> 5;
          const int col = 2 * jj + tt;  // output column inside the tile
          typedef _Float16 h8v __attribute__((ext_vector_type(8)));
          h8v piece;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float w = (fwd ? tile[q][col][8 * qq + e] : tile[q][8 * qq + e][col]) * ULTR_H3_WSCALE;
            const _Float16 hi = (_Float16)w;
            piece[e] = hl ? (_Float16)(w - (float)hi) : hi;
          }
          const int64_t pos = ((((int64_t)(c0 >> 5) * nks + (z0 >> 5)) * 4 + (2 * tt + hl)) * 64 +
                               ((((z0 & 31) >> 3) + qq) * 16 + ((c0 & 31) >> 1) + jj)) * 8;
          _Float16* dst = reinterpret_cast<_Float16*>(wt + (fwd ? dp.whf_off[j] : dp.whb_off[j]));
          *reinterpret_cast<h8v*>(dst + pos) = piece;
        }
      }
      if (dp.sw_ok && tid < 128) {
        // the fragment-major copies (DnnPlan::wsf_off / wsb_off): the 16 x 16 tile is 64 float4 pieces of each, 8 pieces
        // (128 bytes) contiguous; threads 0..63 write the forward copy, 64..127 the dgrad copy (layers >= 1).  Elements
        // beyond K are zero in the tile (and M is a multiple of 32), as the padding of the copies requires.
        const int t6 = tid & 63, uu = t6 >> 5, qq = (t6 >> 3) & 3, ii = t6 & 7;
        const int a = 4 * qq + 2 * uu, b = 2 * ii;  // a: contraction offset inside the tile, b: output-column offset
        if (tid < 64) {
          const int h = (k0[q] & 31) >> 4, ntr = (dp.K[j] + 31) >> 5;
          const int64_t pos = ((((int64_t)(m0[q] >> 5) * ntr + (k0[q] >> 5)) * 4 + (2 * h + uu)) * 64 +
                               (qq * 16 + ((m0[q] & 31) >> 1) + ii)) * 4;
          st4(wt + dp.wsf_off[j] + pos, make_float4(tile[q][b][a], tile[q][b + 1][a], tile[q][b][a + 1], tile[q][b + 1][a + 1]));
        } else if (j >= 1) {
          const int h = (m0[q] & 31) >> 4, ntr = (dp.M[j] + 31) >> 5;
          const int64_t pos = ((((int64_t)(k0[q] >> 5) * ntr + (m0[q] >> 5)) * 4 + (2 * h + uu)) * 64 +
                               (qq * 16 + ((k0[q] & 31) >> 1) + ii)) * 4;
          st4(wt + dp.wsb_off[j] + pos, make_float4(tile[q][a][b], tile[q][a][b + 1], tile[q][a + 1][b], tile[q][a + 1][b + 1]));
        }
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < TPW; ++q)
      if (pvpos[q] >= 0) wt[dp.wt_pv_off + pvpos[q]] = pn[q];
  }
}

extern "C" int ultr_apply_update(const ultr_update_desc* u, const ultr_dnn_desc* d, float* params, float* wt, float* state,
                                 const float* grads, float* aux, const void* bwd_ws, float* scalars_out, void* stream) {
  return ultr_apply_update_ex(u, d, params, wt, state, grads, aux, bwd_ws, scalars_out, 0, stream);
}
// nsq2 > 0: the step's slab-reduction launch left level-2 sum-of-squares partials (ultr_plan.h) - library-internal (ultr_train_step)
int ultr_apply_update_ex(const ultr_update_desc* u, const ultr_dnn_desc* d, float* params, float* wt, float* state, const float* grads,
                         float* aux, const void* bwd_ws, float* scalars_out, int nsq2, void* stream) {
  if (!u || !params || !grads || !bwd_ws || u->n_params <= 0 || u->list_size <= 0) return ULTR_E_BADARG;
  DnnPlan dp;
  memset(&dp, 0, sizeof(dp));
  if (wt != nullptr) {
    if (!ultr_make_dnn_plan(d, 0, &dp) || dp.P != u->n_params) return ULTR_E_BADARG;
  }
  if (u->algo < 0 || u->algo > ULTR_ALGO_REGEM || (u->optimizer != ULTR_OPT_ADAGRAD && u->optimizer != ULTR_OPT_SGD))
    return ULTR_E_BADARG;
  if (u->algo != ULTR_ALGO_SOFTMAX && !aux) return ULTR_E_BADARG;
  if (u->optimizer == ULTR_OPT_ADAGRAD && u->algo != ULTR_ALGO_DLA && !state) return ULTR_E_BADARG;
  if (u->l2_loss < 0.f || (u->l2_loss > 0.f && !scalars_out) || (u->l2_loss > 0.f && u->algo == ULTR_ALGO_LAMBDARANK))
    return ULTR_E_BADARG;  // LambdaRank has no l2_loss hyper-parameter (lambda_rank.py:42-49)
  const int tail = (int)ultr_tail_len(u->list_size);
  const int nsq = (int)ultr_red_blocks(u->n_params, tail);
  const float* l2_sums = nullptr;
  if (u->l2_loss > 0.f) {
    hipLaunchKernelGGL(l2_sums_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)params, grads, u->n_params,
                       scalars_out + 8);
    l2_sums = scalars_out + 8;
  }
  UltrProfScope prof(ULTR_K_UPDATE, (hipStream_t)stream);
  if (wt != nullptr) {
    const int n_tiles = dp.upd_tile_begin[dp.nl - 1], n_vec = dp.vs_begin[dp.n_vs];
    // a click draw waiting for a launch to ride on (ultr_feed_train_step): its (batch + 3) / 4 workgroups follow the update's
    ultr_click_args rider;
    memset(&rider, 0, sizeof(rider));
    int rider_blocks = 0;
    if (g_ultr_click_rider != nullptr) {
      rider = *g_ultr_click_rider;
      rider_blocks = (rider.batch + 3) / 4;
      g_ultr_click_rider = nullptr;
    }
    if (n_tiles + (n_vec + 255) / 256 < 1536) {  // one unit per workgroup keeps the launch wide (config 2: 401 units, config 3: 940 -
                                                 // four per workgroup left 235 workgroups for 256 CUs: 7.0 -> 9.0 us)
      const int own = n_tiles + (n_vec + 255) / 256;
      ULTR_LAUNCH(prof, update_tiled_kernel<1>, dim3(own + rider_blocks), dim3(256), 0, (hipStream_t)stream, *u, dp, params, state,
                  grads, aux, wt, (const float*)bwd_ws, nsq, scalars_out, n_tiles, l2_sums, nsq2, own, rider);
    } else {
      const int tb = (n_tiles + 3) / 4;
      const int own = tb + (n_vec + 1023) / 1024;
      ULTR_LAUNCH(prof, update_tiled_kernel<4>, dim3(own + rider_blocks), dim3(256), 0, (hipStream_t)stream, *u, dp, params, state,
                  grads, aux, wt, (const float*)bwd_ws, nsq, scalars_out, tb, l2_sums, nsq2, own, rider);
    }
  } else {
    const int nblk = (int)((u->n_params + 255) / 256);
    ULTR_LAUNCH(prof, update_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, *u, dp, params, state, grads, aux,
                (const float*)bwd_ws, nsq, scalars_out, l2_sums, nsq2);
  }
  return (int)hipGetLastError();
}
