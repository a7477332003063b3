// ultr_loss.hip — gfx950 list-level loss kernels (reference ultra/learning_algorithm/*.py)
//
// One 64-lane wavefront owns one ranked list: its scores / labels are staged in LDS, list-level
// reductions (softmax max/sum, pair sums, rank-by-counting sort) are wavefront shuffles — no atomics, no
// cross-workgroup traffic.  A workgroup owns LPW lists (currently one) and emits ONE partial "step tail"
// [loss_sum, D, loss2_sum, D2, per-position sums (2L)] that the gradient reduction adds up in fixed order,
// so every step is bit-reproducible.  The global normalisers (D) are NOT applied here: the kernels emit
// d(loss)/d(scores) x D, the update kernel applies 1/D — this is what makes the data-parallel all-reduce
// exact and saves a grid-wide barrier on one GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ultr_hip.h"
#include "ultr_device.h"
#include "ultr_plan.h"
#include "ultr_prof.h"

#define LPW ULTR_LOSS_LISTS_PER_WG  // lists (= waves) per workgroup

extern "C" int64_t ultr_loss_part_count(int64_t batch) { return batch > 0 ? ultr_loss_parts(batch) : 0; }

extern "C" int64_t ultr_loss_workspace_bytes(int64_t batch, int32_t list_size) {
  if (batch <= 0 || list_size <= 0) return 0;
  // one tail partial per workgroup of the stand-alone loss kernels or per 16-row block (loss fused into the backward)
  // or per workgroup of the fused forward+backward kernel (at most one per list)
  int64_t parts_fused = (batch * (int64_t)list_size + 15) / 16;
  if (batch > parts_fused) parts_fused = batch;
  const int64_t parts = ultr_loss_parts(batch) > parts_fused ? ultr_loss_parts(batch) : parts_fused;
  return (parts * ultr_tail_len(list_size) + 4) * (int64_t)sizeof(float);
}

// sum the per-wave tails of a workgroup in fixed order and store the workgroup's partial
__device__ __forceinline__ void store_wg_tail(const float* sm_tail /*[LPW][tail]*/, int tail, float* part) {
  __syncthreads();
  for (int t = threadIdx.x; t < tail; t += blockDim.x) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < LPW; ++w) s += sm_tail[w * tail + t];
    part[t] = s;
  }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ------------------------------------------------------------------------------------------------
// a4/a7/a8: softmax cross entropy with (IPW) weights          base_algorithm.py:309-330
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LPW * 64) void softmax_ce_kernel(const float* __restrict__ scores,
                                                             const float* __restrict__ labels,
                                                             const float* __restrict__ pw,
                                                             const float* __restrict__ ipw, int n_ipw, int B, int L,
                                                             float* __restrict__ dscores, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tail = (int)ultr_tail_len(L);
  float* sm_tail = smem;                       // [LPW][tail]
  float* sm_s = sm_tail + LPW * tail;          // [LPW][L]
  float* sm_w = sm_s + LPW * L;                // [LPW][L]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * LPW + wave;
  float* ms = sm_s + wave * L;
  float* mw = sm_w + wave * L;
  float* mt = sm_tail + wave * tail;
  for (int t = lane; t < tail; t += 64) mt[t] = 0.f;
  if (b < B) {
    float mx = -INFINITY, S = 0.f;
    for (int l = lane; l < L; l += 64) {
      const float s = scores[(int64_t)b * L + l];
      const float y = labels[(int64_t)l * B + b];
      float p = 1.0f;
      if (pw != nullptr) p = pw[(int64_t)b * L + l];
      else if (ipw != nullptr) p = (y > 0.f) ? ipw[l < n_ipw ? l : n_ipw - 1] : 0.f;  // propensity_estimator.py:22-42
      const float w = (y + 0.0000001f) * p;  // the reference's 1e-7 smoothing
      ms[l] = s;
      mw[l] = w;
      mx = fmaxf(mx, s);
      S += w;
    }
    mx = wave_max(mx);
    S = wave_sum(S);
    float se = 0.f;
    for (int l = lane; l < L; l += 64) se += expf(ms[l] - mx);
    const float lse = mx + logf(wave_sum(se));
    float lb = 0.f;
    for (int l = lane; l < L; l += 64) {
      const float s = ms[l], w = mw[l];
      lb += w * (lse - s);                                   // -w * log_softmax(s)
      dscores[(int64_t)b * L + l] = expf(s - lse) * S - w;   // x D
    }
    lb = wave_sum(lb);
    if (lane == 0) {
      mt[0] = lb;
      mt[1] = S;
    }
  }
  store_wg_tail(sm_tail, tail, part + (int64_t)blockIdx.x * tail);
}

extern "C" int ultr_softmax_ce(const float* scores, const float* labels, const float* pw, const float* ipw_table,
                               int32_t n_ipw, int32_t batch, int32_t list_size, float* dscores, void* loss_ws,
                               void* stream) {
  if (!scores || !labels || !dscores || !loss_ws || batch <= 0 || list_size <= 0 || (ipw_table && n_ipw <= 0))
    return ULTR_E_BADARG;
  const int tail = (int)ultr_tail_len(list_size);
  const size_t lds = (size_t)LPW * (tail + 2 * list_size) * sizeof(float);
  if (lds > 64 * 1024) return ULTR_E_UNSUPPORTED;
  UltrProfScope prof(ULTR_K_LOSS, (hipStream_t)stream);
  ULTR_LAUNCH(prof, softmax_ce_kernel, dim3((unsigned)ultr_loss_parts(batch)), dim3(LPW * 64), lds, (hipStream_t)stream,
                     scores, labels, pw, ipw_table, (int)n_ipw, (int)batch, (int)list_size, dscores, (float*)loss_ws);
  return (int)hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// next row 8f.3: RegressionEM                                         regression_EM.py:108-193
// ------------------------------------------------------------------------------------------------
// E-step posteriors from the CURRENT scores and propensity, Bernoulli pseudo-labels y = ceil(p_r1 - u), loss =
// BCEWithLogits(s, y) averaged over all B*L elements (D = element count in the tail, applied by the update kernel),
// per-position sums of c + (1 - c) P(e=1, r=0 | c=0) for the M-step.  u comes from `uniforms` (teacher-forced
// parity with the reference's recorded torch.rand draw) or, when NULL, from Philox keyed by (seed, step).
__global__ __launch_bounds__(LPW * 64) void regem_kernel(const float* __restrict__ scores, const float* __restrict__ labels,
                                                        const float* __restrict__ propensity,
                                                        const float* __restrict__ uniforms, uint64_t seed, uint64_t step,
                                                        int B, int L, float* __restrict__ dscores,
                                                        float* __restrict__ pseudo, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tail = (int)ultr_tail_len(L);
  float* sm_tail = smem;  // [LPW][tail]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * LPW + wave;
  float* mt = sm_tail + wave * tail;
  for (int t = lane; t < tail; t += 64) mt[t] = 0.f;
  if (b < B) {
    const Philox rng{(uint32_t)seed ^ (uint32_t)(step * 0x9E3779B97F4A7C15ull >> 32), (uint32_t)(seed >> 32) ^ (uint32_t)step};
    float lsum = 0.f, cnt = 0.f;
    for (int l = lane; l < L; l += 64) {
      const float s = scores[(int64_t)b * L + l];
      const float c = labels[(int64_t)l * B + b];
      const float pr = propensity[l];
      const float gamma = sigmoidf_(s);
      const float den = 1.0f - pr * gamma;
      const float p_e1_r0_c0 = pr * (1.0f - gamma) / den;
      const float p_e0_r1_c0 = (1.0f - pr) * gamma / den;
      const float p_r1 = c + (1.0f - c) * p_e0_r1_c0;
      float u;
      if (uniforms != nullptr) {
        u = uniforms[(int64_t)b * L + l];
      } else {
        uint32_t ctr[4] = {(uint32_t)b, (uint32_t)l, 0x5245454Du, 0x1u};
        rng(ctr);
        u = u01(ctr[0]);
      }
      const float y = ceilf(p_r1 - u);  // get_bernoulli_sample (regression_EM.py:20-34)
      lsum += fmaxf(s, 0.f) - s * y + log1pf(expf(-fabsf(s)));  // BCEWithLogits element
      cnt += 1.0f;
      dscores[(int64_t)b * L + l] = gamma - y;  // x D
      if (pseudo != nullptr) pseudo[(int64_t)b * L + l] = y;
      mt[ULTR_TAIL_FIXED + l] = c + (1.0f - c) * p_e1_r0_c0;
    }
    lsum = wave_sum(lsum);
    cnt = wave_sum(cnt);
    if (lane == 0) {
      mt[0] = lsum;
      mt[1] = cnt;
    }
  }
  store_wg_tail(sm_tail, tail, part + (int64_t)blockIdx.x * tail);
}

extern "C" int ultr_regem_loss(const float* scores, const float* labels, const float* propensity, const float* uniforms,
                               uint64_t seed, uint64_t step, int32_t batch, int32_t list_size, float* dscores,
                               float* pseudo_labels_out, void* loss_ws, void* stream) {
  if (!scores || !labels || !propensity || !dscores || !loss_ws || batch <= 0 || list_size <= 0) return ULTR_E_BADARG;
  const int tail = (int)ultr_tail_len(list_size);
  const size_t lds = (size_t)LPW * tail * sizeof(float);
  if (lds > 64 * 1024) return ULTR_E_UNSUPPORTED;
  UltrProfScope prof(ULTR_K_LOSS, (hipStream_t)stream);
  ULTR_LAUNCH(prof, regem_kernel, dim3((unsigned)ultr_loss_parts(batch)), dim3(LPW * 64), lds, (hipStream_t)stream, scores,
              labels, propensity, uniforms, seed, step, (int)batch, (int)list_size, dscores, pseudo_labels_out,
              (float*)loss_ws);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// a9: DLA dual loss                                             dla.py:196-237, 24-48, 287-306
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LPW * 64) void dla_loss_kernel(const float* __restrict__ scores,
                                                           const float* __restrict__ labels,
                                                           const float* __restrict__ prop, int l2p, int B, int L,
                                                           float* __restrict__ dscores, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tail = (int)ultr_tail_len(L);
  float* sm_tail = smem;               // [LPW][tail]
  float* sm_s = sm_tail + LPW * tail;  // [LPW][L] scores
  float* sm_y = sm_s + LPW * L;        // [LPW][L] clicks
  float* sm_p = sm_y + LPW * L;        // [LPW][L] propensity logits ELU(W_l + b)  (batch independent)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * LPW + wave;
  float* ms = sm_s + wave * L;
  float* my = sm_y + wave * L;
  float* mp = sm_p + wave * L;
  float* mt = sm_tail + wave * tail;
  for (int t = lane; t < tail; t += 64) mt[t] = 0.f;
  if (b < B) {
    const float pbias = prop[L];
    float mxs = -INFINITY, mxp = -INFINITY, sums = 0.f, sump = 0.f;
    for (int l = lane; l < L; l += 64) {
      const float s = scores[(int64_t)b * L + l];
      const float z = prop[l] + pbias;
      const float pl = z > 0.f ? z : expm1f(z);  // DenoisingNet: Linear(one-hot) -> ELU
      ms[l] = s;
      my[l] = labels[(int64_t)l * B + b];
      mp[l] = pl;
      mxs = fmaxf(mxs, s);
      mxp = fmaxf(mxp, pl);
      sums += s;
      sump += pl;
    }
    mxs = wave_max(mxs);
    mxp = wave_max(mxp);
    const float means = wave_sum(sums) / (float)L, meanp = wave_sum(sump) / (float)L;
    float ses = 0.f, sep = 0.f;
    for (int l = lane; l < L; l += 64) {
      ses += expf(ms[l] - mxs);
      sep += expf(mp[l] - mxp);
    }
    const float lses = mxs + logf(wave_sum(ses));
    const float lsep = mxp + logf(wave_sum(sep));
    // logits_to_prob at position 0 (softmax, or sigmoid(x - mean), dla.py:21-22)
    const float p0 = (l2p == 1) ? sigmoidf_(mp[0] - meanp) : expf(mp[0] - lsep);
    const float r0 = (l2p == 1) ? sigmoidf_(ms[0] - means) : expf(ms[0] - lses);
    float Sr = 0.f, Se = 0.f;
    for (int l = lane; l < L; l += 64) {
      const float pl = (l2p == 1) ? sigmoidf_(mp[l] - meanp) : expf(mp[l] - lsep);
      const float rl = (l2p == 1) ? sigmoidf_(ms[l] - means) : expf(ms[l] - lses);
      Sr += (my[l] + 0.0000001f) * (p0 / pl);  // propensity weights, get_normalized_weights
      Se += (my[l] + 0.0000001f) * (r0 / rl);  // relevance weights
    }
    Sr = wave_sum(Sr);
    Se = wave_sum(Se);
    float lr = 0.f, le = 0.f;
    for (int l = lane; l < L; l += 64) {
      const float pl = (l2p == 1) ? sigmoidf_(mp[l] - meanp) : expf(mp[l] - lsep);
      const float rl = (l2p == 1) ? sigmoidf_(ms[l] - means) : expf(ms[l] - lses);
      const float wr = (my[l] + 0.0000001f) * (p0 / pl);
      const float we = (my[l] + 0.0000001f) * (r0 / rl);
      lr += wr * (lses - ms[l]);
      le += we * (lsep - mp[l]);
      dscores[(int64_t)b * L + l] = expf(ms[l] - lses) * Sr - wr;  // d rank_loss / d s  x D_rank
      mt[ULTR_TAIL_FIXED + l] = expf(mp[l] - lsep) * Se - we;      // d exam_loss / d propensity_l  x D_exam
    }
    lr = wave_sum(lr);
    le = wave_sum(le);
    if (lane == 0) {
      mt[0] = lr;
      mt[1] = Sr;
      mt[2] = le;
      mt[3] = Se;
    }
  }
  store_wg_tail(sm_tail, tail, part + (int64_t)blockIdx.x * tail);
}

extern "C" int ultr_dla_loss(const float* scores, const float* labels, const float* prop_params, int32_t logits_to_prob,
                             int32_t batch, int32_t list_size, float* dscores, void* loss_ws, void* stream) {
  if (!scores || !labels || !prop_params || !dscores || !loss_ws || batch <= 0 || list_size <= 0) return ULTR_E_BADARG;
  const int tail = (int)ultr_tail_len(list_size);
  const size_t lds = (size_t)LPW * (tail + 3 * list_size) * sizeof(float);
  if (lds > 64 * 1024) return ULTR_E_UNSUPPORTED;
  UltrProfScope prof(ULTR_K_LOSS, (hipStream_t)stream);
  ULTR_LAUNCH(prof, dla_loss_kernel, dim3((unsigned)ultr_loss_parts(batch)), dim3(LPW * 64), lds, (hipStream_t)stream,
                     scores, labels, prop_params, (int)logits_to_prob, (int)batch, (int)list_size, dscores,
                     (float*)loss_ws);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// a10: PairDebias                                               pairwise_debias.py:142-157
// ------------------------------------------------------------------------------------------------
// Lane i owns position i and walks the partner positions j, so every gradient / EM sum it owns is produced locally (the
// "wavefront pair-diff kernel").  Of the ordered pairs (i,j) and (j,i) at most one is valid (c_i > c_j or c_j > c_i): ONE
// branch-free evaluation per unordered pair - e = exp(-|x|) gives both softplus(x) = max(x,0) + log1p(e) and sigmoid(x) -
// with the reciprocals of t_plus / t_minus staged once.  A list is small (L^2 pairs) and there are only `batch` of them, so
// the kernel is one wavefront's instruction stream long: PD_JW wavefronts share a list (each a slice of the j range; their
// four partial sums per position are combined in fixed order through LDS).
#ifndef PD_JW
#define PD_JW 16  // wavefronts per list (round 6: 4 -> 16, one list per CU keeps four waves per SIMD busy: pairdebias 7.6 -> 6.2 us,
#endif           // lambdarank 15.3 -> 12.7 us at config 4; tools/ab.sh "--config 4lambda" product jw8 jw16)
__global__ __launch_bounds__(LPW * PD_JW * 64) void pairdebias_kernel(const float* __restrict__ scores,
                                                                     const float* __restrict__ labels,
                                                                     const float* __restrict__ t_plus,
                                                                     const float* __restrict__ t_minus, int B, int L,
                                                                     float bscale, float* __restrict__ dscores,
                                                                     float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tail = (int)ultr_tail_len(L);
  float* sm_tail = smem;               // [LPW][tail]
  float* sm_s = sm_tail + LPW * tail;  // [LPW][L]
  float* sm_c = sm_s + LPW * L;        // [LPW][L]
  float* sm_rtp = sm_c + LPW * L;      // [L] 1 / t_plus
  float* sm_rtm = sm_rtp + L;          // [L] 1 / t_minus
  float* sm_acc = sm_rtm + L;          // [LPW][PD_JW][L][4]: g, t_plus_loss, t_minus_loss, loss per (slice, position)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lw = wave / PD_JW, jw = wave - lw * PD_JW;   // list of the workgroup, slice of the j range
  const int b = blockIdx.x * LPW + lw;
  float* ms = sm_s + lw * L;
  float* mc = sm_c + lw * L;
  float* mt = sm_tail + lw * tail;
  for (int t = threadIdx.x; t < L; t += blockDim.x) {
    sm_rtp[t] = 1.0f / t_plus[t];
    sm_rtm[t] = 1.0f / t_minus[t];
  }
  if (jw == 0) {
    for (int t = lane; t < tail; t += 64) mt[t] = 0.f;
    if (b < B)
      for (int l = lane; l < L; l += 64) {
        ms[l] = scores[(int64_t)b * L + l];
        mc[l] = labels[(int64_t)l * B + b];
      }
  }
  __syncthreads();
  const int jlen = (L + PD_JW - 1) / PD_JW, j0 = jw * jlen, j1 = (j0 + jlen < L) ? j0 + jlen : L;
  if (b < B) {
    for (int i = lane; i < L; i += 64) {
      const float si = ms[i], ci = mc[i], rtpi = sm_rtp[i], rtmi = sm_rtm[i];
      float g = 0.f, tpl = 0.f, tml = 0.f, li = 0.f;
      for (int j = j0; j < j1; ++j) {
        const float dc = ci - mc[j];                 // > 0: (i, j) is the valid pair; < 0: (j, i); 0 (incl. j == i): neither
        const bool fwd = dc > 0.f;
        const float m = fminf(1.0f, fabsf(dc));      // valid_pair_mask value
        const float x = fwd ? ms[j] - si : si - ms[j];   // s_neg - s_pos of the valid ordered pair
        const float e = __expf(-fabsf(x));
        const float r = 1.0f / (1.0f + e);
        const float sp = fmaxf(x, 0.f) + log1pf(e);      // softplus(x)
        const float sg = (x >= 0.f) ? r : e * r;         // sigmoid(x)
        const float pl = bscale * m * sp;                // PL contribution of this list (x B)
        const float gw = bscale * m * sg;
        const float rj = fwd ? sm_rtm[j] : sm_rtp[j];    // 1 / t_minus[j]  or  1 / t_plus[j]
        // (i, j): t_plus_loss[i] += PL_ij / t-_j; loss += PL_ij / t+_i / t-_j; d/ds_i = -gw / t+_i / t-_j
        // (j, i): t_minus_loss[i] += PL_ji / t+_j;                              d/ds_i = +gw / t+_j / t-_i
        const float plr = pl * rj;
        tpl += fwd ? plr : 0.f;
        tml += fwd ? 0.f : plr;
        li += fwd ? plr * rtpi : 0.f;
        g += fwd ? -(gw * rj * rtpi) : gw * rj * rtmi;
      }
      float* a = sm_acc + ((size_t)(lw * PD_JW + jw) * L + i) * 4;
      a[0] = g; a[1] = tpl; a[2] = tml; a[3] = li;
    }
  }
  __syncthreads();
  if (b < B && jw == 0) {
    float lsum = 0.f;
    for (int i = lane; i < L; i += 64) {
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float t = sm_acc[((size_t)(lw * PD_JW + 0) * L + i) * 4 + k];
#pragma unroll
        for (int w = 1; w < PD_JW; ++w) t += sm_acc[((size_t)(lw * PD_JW + w) * L + i) * 4 + k];
        v[k] = t;
      }
      dscores[(int64_t)b * L + i] = v[0];
      mt[ULTR_TAIL_FIXED + i] = v[1];
      mt[ULTR_TAIL_FIXED + L + i] = v[2];
      lsum += v[3];
    }
    lsum = wave_sum(lsum);
    if (lane == 0) {
      mt[0] = lsum;
      mt[1] = 1.0f;  // unused normaliser
    }
  }
  store_wg_tail(sm_tail, tail, part + (int64_t)blockIdx.x * tail);
}

extern "C" int ultr_pairdebias_loss(const float* scores, const float* labels, const float* t_plus, const float* t_minus,
                                    int32_t batch, int32_t list_size, int32_t batch_total, float* dscores, void* loss_ws,
                                    void* stream) {
  if (!scores || !labels || !t_plus || !t_minus || !dscores || !loss_ws || batch <= 0 || list_size <= 0 || batch_total <= 0)
    return ULTR_E_BADARG;
  const int tail = (int)ultr_tail_len(list_size);
  const size_t lds = ((size_t)LPW * (tail + 2 * list_size) + 2 * list_size + (size_t)LPW * PD_JW * list_size * 4) * sizeof(float);
  if (lds > 160 * 1024) return ULTR_E_UNSUPPORTED;
  if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(pairdebias_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return ULTR_E_UNSUPPORTED;
  UltrProfScope prof(ULTR_K_LOSS, (hipStream_t)stream);
  ULTR_LAUNCH(prof, pairdebias_kernel, dim3((unsigned)ultr_loss_parts(batch)), dim3(LPW * PD_JW * 64), lds, (hipStream_t)stream,
                     scores, labels, t_plus, t_minus, (int)batch, (int)list_size, (float)batch_total, dscores,
                     (float*)loss_ws);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// a11: LambdaRank                                               lambda_rank.py:116-135, 247-291
// ------------------------------------------------------------------------------------------------
// Sort = rank by counting inside the wavefront (stable: ties broken by original index), L^2 compares from
// LDS; the same trick orders the labels for the ideal DCG.  Then the pair walk of PairDebias on the SORTED
// list, with delta-NDCG weights and the reference's BCE-with-logits-on-a-probability quirk.

// PD_JW wavefronts share a list: wave 0 of the list sorts (rank by counting), then every wave walks a slice of the
// partner positions c for all positions r; the four partial sums per position are combined in fixed order through
// LDS.  Per pair ONE exp gives both probabilities without cancellation - with ez = exp(-|z|), z = sigma (s_r - s_c):
// sigmoid(|z|) = 1 / (1 + ez), sigmoid(-|z|) = ez / (1 + ez) - and one more exp per probability gives both its
// BCE-with-logits value and its derivative (sigmoid(x) - target).  Discounts and reciprocals are staged per position.
__global__ __launch_bounds__(LPW * PD_JW * 64) void lambdarank_kernel(const float* __restrict__ scores,
                                                                     const float* __restrict__ labels,
                                                                     const float* __restrict__ t_plus,
                                                                     const float* __restrict__ t_minus, float sigma, int B,
                                                                     int L, float* __restrict__ dscores,
                                                                     float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tail = (int)ultr_tail_len(L);
  float* sm_tail = smem;                // [LPW][tail]
  float* sm_s = sm_tail + LPW * tail;   // [LPW][L] raw scores, later d(loss)/d(sorted score)
  float* sm_y = sm_s + LPW * L;         // [LPW][L] raw labels
  float* sm_ps = sm_y + LPW * L;        // [LPW][L] scores sorted desc
  float* sm_ls = sm_ps + LPW * L;       // [LPW][L] labels in score order
  float* sm_g = sm_ls + LPW * L;        // [LPW][L] gains 2^l - 1 in score order
  float* sm_tp = sm_g + LPW * L;        // [L] t_plus
  float* sm_tm = sm_tp + L;             // [L] t_minus
  float* sm_rtp = sm_tm + L;            // [L] 1 / t_plus
  float* sm_rtm = sm_rtp + L;           // [L] 1 / t_minus
  float* sm_d = sm_rtm + L;             // [L] discount 1 / log2(rank + 2)
  int* sm_pos = reinterpret_cast<int*>(sm_d + L);             // [LPW][L] sorted position of original index
  float* sm_acc = reinterpret_cast<float*>(sm_pos + LPW * L); // [LPW][PD_JW][L][4]: g, t_plus_loss, t_minus_loss, loss
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lw = wave / PD_JW, jw = wave - lw * PD_JW;
  const int b = blockIdx.x * LPW + lw;
  float* ms = sm_s + lw * L;
  float* my = sm_y + lw * L;
  float* ps = sm_ps + lw * L;
  float* ls = sm_ls + lw * L;
  float* gs = sm_g + lw * L;
  int* pos = sm_pos + lw * L;
  float* mt = sm_tail + lw * tail;
  for (int t = threadIdx.x; t < L; t += blockDim.x) {
    const float tp = t_plus[t], tm = t_minus[t];
    sm_tp[t] = tp;
    sm_tm[t] = tm;
    sm_rtp[t] = 1.0f / tp;
    sm_rtm[t] = 1.0f / tm;
    sm_d[t] = 1.0f / log2f((float)t + 2.0f);
  }
  if (jw == 0) {
    for (int t = lane; t < tail; t += 64) mt[t] = 0.f;
    if (b < B)
      for (int l = lane; l < L; l += 64) {
        ms[l] = scores[(int64_t)b * L + l];
        my[l] = labels[(int64_t)l * B + b];
      }
  }
  __syncthreads();
  float idcg = 0.f;
  if (b < B && jw == 0) {
    for (int i = lane; i < L; i += 64) {
      const float si = ms[i], yi = my[i];
      int rs = 0, ry = 0;
      for (int j = 0; j < L; ++j) {
        const float sj = ms[j], yj = my[j];
        rs += (sj > si || (sj == si && j < i)) ? 1 : 0;
        ry += (yj > yi || (yj == yi && j < i)) ? 1 : 0;
      }
      pos[i] = rs;
      ps[rs] = si;
      ls[rs] = yi;
      gs[rs] = exp2f(yi) - 1.0f;
      // dcg(): sum (2^l - 1) / ln(rank + 1), rank 1-based -> ln(ry + 2)   (lambda_rank.py:262-266)
      idcg += (exp2f(yi) - 1.0f) / logf((float)ry + 2.0f);
    }
    idcg = wave_sum(idcg);
  }
  __syncthreads();  // the sorted arrays are visible to the list's other waves
  const int clen = (L + PD_JW - 1) / PD_JW, c0 = jw * clen, c1 = (c0 + clen < L) ? c0 + clen : L;
  if (b < B) {
    for (int r = lane; r < L; r += 64) {
      const float sr = ps[r], lr_ = ls[r], gr = gs[r], tpr = sm_tp[r], tmr = sm_tm[r], rtpr = sm_rtp[r], rtmr = sm_rtm[r];
      const float dr = sm_d[r];
      float g = 0.f, tpl = 0.f, tml = 0.f, li = 0.f;
      for (int c = c0; c < c1; ++c) {
        const float delta = fabsf(gr - gs[c]) * fabsf(dr - sm_d[c]);        // x 1/IDCG applied later
        const float S = fminf(1.0f, fmaxf(lr_ - ls[c], -1.0f));
        const float pb_rc = 0.5f * (1.0f + S), pb_cr = 0.5f * (1.0f - S);
        const float z = sigma * (sr - ps[c]);
        const float ez = __expf(-fabsf(z));
        const float r0 = 1.0f / (1.0f + ez);
        const float xhi = r0, xlo = ez * r0;                                // sigmoid(|z|), sigmoid(-|z|)
        const float x_rc = (z >= 0.f) ? xhi : xlo, x_cr = (z >= 0.f) ? xlo : xhi;
        // BCEWithLogits on a probability x in (0, 1): x - x t + log1p(exp(-x)); derivative sigmoid(x) - t
        const float e1 = __expf(-x_rc), e2 = __expf(-x_cr);
        const float l_rc = delta * (x_rc - x_rc * pb_rc + log1pf(e1));      // PL[r, c] contribution
        const float l_cr = delta * (x_cr - x_cr * pb_cr + log1pf(e2));      // PL[c, r]
        const float rtmc = sm_rtm[c], rtpc = sm_rtp[c];
        const bool ok_rc = (tpr * sm_tm[c]) != 0.f, ok_cr = (sm_tp[c] * tmr) != 0.f;   // _safe_div
        tpl += l_rc * rtmc;   // t_plus_loss[r]  = sum_c PL[r,c] / t_minus[c]
        tml += l_cr * rtpc;   // t_minus_loss[r] = sum_c PL[c,r] / t_plus[c]
        li += ok_rc ? l_rc * rtpr * rtmc : 0.f;
        // d PL[r,c]/d s_r  and  d PL[c,r]/d s_r  (x = sigmoid(+-z): dx/ds_r = +-sigma x (1 - x), x (1 - x) = xhi xlo)
        const float xx = sigma * xhi * xlo;
        const float d_rc = delta * (1.0f / (1.0f + e1) - pb_rc) * xx;
        const float d_cr = delta * (1.0f / (1.0f + e2) - pb_cr) * xx;
        g += ok_rc ? d_rc * rtpr * rtmc : 0.f;
        g -= ok_cr ? d_cr * rtpc * rtmr : 0.f;
      }
      float* a = sm_acc + ((size_t)(lw * PD_JW + jw) * L + r) * 4;
      a[0] = g; a[1] = tpl; a[2] = tml; a[3] = li;
    }
  }
  __syncthreads();
  if (b < B && jw == 0) {
    float lsum = 0.f;
    for (int r = lane; r < L; r += 64) {
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float t = sm_acc[((size_t)(lw * PD_JW + 0) * L + r) * 4 + k];
#pragma unroll
        for (int w = 1; w < PD_JW; ++w) t += sm_acc[((size_t)(lw * PD_JW + w) * L + r) * 4 + k];
        v[k] = t;
      }
      mt[ULTR_TAIL_FIXED + r] = v[1];
      mt[ULTR_TAIL_FIXED + L + r] = v[2];
      ms[r] = v[0];  // raw scores are dead after the sort: reuse as d(loss)/d(sorted score r)
      lsum += v[3];
    }
    lsum = wave_sum(lsum);
    if (lane == 0) {
      mt[0] = lsum;
      mt[1] = idcg;
    }
  }
  // ms[] (now gradients by sorted position) written by lane r, read by the lane owning the original index
  __syncthreads();
  if (b < B && jw == 0)
    for (int i = lane; i < L; i += 64) dscores[(int64_t)b * L + i] = ms[pos[i]];
  store_wg_tail(sm_tail, tail, part + (int64_t)blockIdx.x * tail);
}

extern "C" int ultr_lambdarank_loss(const float* scores, const float* labels, const float* t_plus, const float* t_minus,
                                    float sigma, int32_t batch, int32_t list_size, float* dscores, void* loss_ws,
                                    void* stream) {
  if (!scores || !labels || !t_plus || !t_minus || !dscores || !loss_ws || batch <= 0 || list_size <= 0)
    return ULTR_E_BADARG;
  const int tail = (int)ultr_tail_len(list_size);
  const size_t lds = ((size_t)LPW * (tail + 6 * list_size) + 5 * list_size + (size_t)LPW * PD_JW * list_size * 4) * sizeof(float);
  if (lds > 160 * 1024) return ULTR_E_UNSUPPORTED;
  if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(lambdarank_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return ULTR_E_UNSUPPORTED;
  UltrProfScope prof(ULTR_K_LOSS, (hipStream_t)stream);
  ULTR_LAUNCH(prof, lambdarank_kernel, dim3((unsigned)ultr_loss_parts(batch)), dim3(LPW * PD_JW * 64), lds, (hipStream_t)stream,
                     scores, labels, t_plus, t_minus, sigma, (int)batch, (int)list_size, dscores, (float*)loss_ws);
  return (int)hipGetLastError();
}
