// ultr_setrank.hip — the SetRank ranking model (reference ultra/ranking_model/SetRank.py:23-255),
// forward and backward.  Everything is a kernel of this library (no vendor BLAS):
//   * the token-local Linear layers run on the LDS-tiled fp32 matrix-core GEMM of ultr_gemm.h: forward Y = X.W^T + b
//     (+ReLU) with the reference's [out, in] weights as the n-major operand and the bias / activation in the epilogue,
//     dgrad dX (+)= dY.W with the accumulate / ReLU-mask epilogue; the two M = 1 products of the scorer have row kernels;
//   * feature gather + LayerNorm, residual + LayerNorm (forward, and backward fused with the gamma / beta / bias column
//     sums), weight gradients (sr_wgrad_kernel: 64 x 64 blocks x row chunks -> slabs); the per-(list, head) self-attention
//     WITHOUT Q/K/V projections (the heads are slices of x itself, SetRank.py:57-66) lives in ultr_sr_attn.hip, the fused
//     persistent launches of the encoder blocks in ultr_sr_fwd.hip / ultr_sr_bwd.hip (round 6);
//   * deterministic: no atomics; every partial sum (slabs, row-block partials) is folded in a fixed order, all folds of a
//     backward in ONE launch at its end (sr_fold_all_kernel).
// Token n = b*L + l (list-major), all activations row-major [T, width].  Parameters: ONE flat vector in the
// reference's state_dict order (ultr_setrank_param_offsets).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/ultr_hip.h"
#include "ultr_device.h"
#ifdef ULTR_TRACE
__device__ unsigned long long g_ugemm_trace[64 * 32];
#define UGEMM_TRACE_STAMP(slot)                                                                    \
  do {                                                                                             \
    if (threadIdx.x == 0 && (blockIdx.x & 7) == 0 && (blockIdx.x >> 3) < 64 && g_ugemm_trace[(blockIdx.x >> 3) * 32 + 31] == 0ull) \
      g_ugemm_trace[(blockIdx.x >> 3) * 32 + (slot)] = __builtin_amdgcn_s_memtime();              \
  } while (0)
extern "C" int ultr_gemm_trace_read(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ugemm_trace), sizeof(unsigned long long) * 64 * 32);
}
extern "C" int ultr_gemm_trace_arm(int on) {  // slot 31 != 0: frozen
  unsigned long long z[64 * 32];
  for (int k = 0; k < 64 * 32; ++k) z[k] = (on || (k & 31) != 31) ? 0ull : 1ull;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ugemm_trace), z, sizeof(z));
}
#endif
#ifndef UGEMM_TRACE_STAMP
#define UGEMM_TRACE_STAMP(slot) \
  do {                          \
  } while (0)
#endif
#include "ultr_gemm.h"
#include "ultr_h3.h"
#include "ultr_plan.h"
#include "ultr_sr_attn.h"
#include "ultr_sr_bwd.h"

#define SR_EPS 1e-6f  // nn.LayerNorm(eps=1e-6) everywhere in SetRank.py (:100-101, :134)
#define SR_ROWS 4     // rows per workgroup (= waves) of the row-wise kernels
#define SR_CS_ROWS 128  // rows per partial of the column-sum kernels
#define SR_LB_ROWS 64   // rows per workgroup (and per partial) of the fused LayerNorm backward

namespace {

struct SrLayer {  // offsets (floats) into the flat parameter vector
  int64_t wd, bd, wf1, bf1, wf2, bf2, g1, b1, g2, b2;
};
struct SrPlan {
  int F, d, H, nl, dff, dh;
  int att_f16;  // 1: fp16-operand attention kernels (ultr_setrank_desc::attention_dtype)
  int64_t T;
  int64_t g_in, b_in, w1, b1, w2, b2, wo1, bo1, wo2, bo2;
  SrLayer lay[8];
  int64_t P;
  // saved activations (floats)
  int64_t sv_xg, sv_mean_in, sv_rstd_in, sv_xn0, sv_h0, sv_oh;
  int64_t sv_x[9];  // x_0 .. x_nl  [T, d]
  int64_t sv_A[8], sv_s1[8], sv_m1[8], sv_r1[8], sv_out1[8], sv_f[8], sv_s2[8], sv_m2[8], sv_r2[8];
  int64_t sv_lse[8];  // [T, H] attention row statistic (log-sum-exp of the scaled scores)
  int64_t sv_total;
  // split-half weight planes (round 4; ultr_gemm.h run_h3): behind the saved activations, rebuilt by every forward - for every
  // weight matrix W[M][K] with M >= 16 the planes hi | lo of 2^8 W as [M][ldK] halves (forward: Y = X W^T) and of W^T as [K][ldM]
  // halves (dgrad: dX = dY W), ldK / ldM = K / M rounded up to 32, zero-padded.  Offsets in HALVES from sv_planes (a float offset).
  int64_t sv_planes, planes_halves;
  int64_t sv_flag;   // float offset in `saved` of the range word behind the planes (ULTR_H3_FLAG_*: raised by sr_split_planes_kernel)
  int no_h3;         // ultr_setrank_desc::flags & ULTR_MODEL_FP32_PRODUCTS
  struct SplitMat { int64_t off; int M, K, ldK, ldM; int64_t f_off, t_off, g_off, gt_off; };  // g_off: fragment-major forward copy (sr_block_fwd_kernel), -1: none
                                                                                            // gt_off: fragment-major copy of W^T (out K, contraction M: the dgrad
                                                                                            // products of ultr_sr_bwd.hip), -1: none
  int n_split;
  SplitMat split[32];
  // workspace (floats)
  int maxw;
  int64_t ws_g[3];   // three [T, maxw] gradient buffers
  int64_t ws_part;   // [n_cs][3 * maxw] column-sum partials
  int n_cs, n_lb;
  int64_t ws_wg;     // [wg_split][max M*K] partial weight gradients (split over lists: one launch would run a
                     // [M, K] = dY^T X product with T = 100k contraction rows on a handful of workgroups)
  int wg_split;      // chunks of whole lists, divides the batch
  int64_t wg_floats; // floats of one weight-gradient partial region
  // the backward queues every partial-sum fold and runs them as ONE launch at its end (sr_fold_all_kernel): partials then
  // cannot share scratch, each producer takes a fresh piece of this arena
  int64_t ws_arena, arena_floats;
  int64_t ws_total;
  int bwd_fused;     // the widths ultr_sr_bwd.hip takes: transposed fragment copies exist, the arena holds its per-workgroup partials
};

// row chunks of sr_wgrad_kernel for an [M, K] gradient: about two workgroups per CU, chunks of a multiple of 32 rows
// (4 waves x 8 rows per trip), at most 128 of them
int wgrad_chunks(int64_t T, int M, int K, int* rows_per_chunk) {
  const int tiles = ((M + 63) / 64) * ((K + 63) / 64);
  int S = (512 + tiles - 1) / tiles;
  if (S > 128) S = 128;
  int64_t rps = (T + S - 1) / S;
  rps = (rps + 31) / 32 * 32;
  if (rps < 32) rps = 32;
  *rows_per_chunk = (int)rps;
  return (int)((T + rps - 1) / rps);
}

bool make_plan(const ultr_setrank_desc* c, int64_t T, SrPlan* p) {
  if (!c || c->feature_size <= 0 || c->d_model <= 0 || c->num_heads <= 0 || c->num_layers <= 0 || c->num_layers > 8 ||
      c->dff <= 0 || c->d_model % c->num_heads != 0 || (c->attention_dtype != ULTR_ATTN_FP32 && c->attention_dtype != ULTR_ATTN_FP16))
    return false;
  memset(p, 0, sizeof(*p));
  p->F = c->feature_size; p->d = c->d_model; p->H = c->num_heads; p->nl = c->num_layers; p->dff = c->dff;
  p->dh = p->d / p->H;
  p->att_f16 = (c->attention_dtype == ULTR_ATTN_FP16) ? 1 : 0;
  p->no_h3 = (c->flags & ULTR_MODEL_FP32_PRODUCTS) ? 1 : 0;
  p->T = T;
  const int64_t F = p->F, d = p->d, dff = p->dff;
  int64_t o = 0;
  p->g_in = o; o += F;  p->b_in = o; o += F;
  p->w1 = o; o += dff * F;  p->b1 = o; o += dff;
  p->w2 = o; o += d * dff;  p->b2 = o; o += d;
  p->wo1 = o; o += dff * d; p->bo1 = o; o += dff;
  p->wo2 = o; o += dff;     p->bo2 = o; o += 1;
  for (int l = 0; l < p->nl; ++l) {
    SrLayer& y = p->lay[l];
    y.wd = o; o += d * d;    y.bd = o; o += d;
    y.wf1 = o; o += dff * d; y.bf1 = o; o += dff;
    y.wf2 = o; o += d * dff; y.bf2 = o; o += d;
    y.g1 = o; o += d; y.b1 = o; o += d;
    y.g2 = o; o += d; y.b2 = o; o += d;
  }
  p->P = o;
  int64_t s = 0;
  auto take = [&](int64_t n) { const int64_t at = s; s += (n + 3) & ~(int64_t)3; return at; };
  p->sv_xg = take(T * F); p->sv_mean_in = take(T); p->sv_rstd_in = take(T); p->sv_xn0 = take(T * F);
  p->sv_h0 = take(T * dff);
  for (int l = 0; l <= p->nl; ++l) p->sv_x[l] = take(T * d);
  for (int l = 0; l < p->nl; ++l) {
    p->sv_A[l] = take(T * d); p->sv_s1[l] = take(T * d); p->sv_m1[l] = take(T); p->sv_r1[l] = take(T);
    p->sv_out1[l] = take(T * d); p->sv_f[l] = take(T * dff);
    p->sv_s2[l] = take(T * d); p->sv_m2[l] = take(T); p->sv_r2[l] = take(T);
    p->sv_lse[l] = take(T * p->H);
  }
  p->sv_oh = take(T * dff);
  p->sv_total = s;
  {
    int64_t h = 0;
    auto add = [&](int64_t off, int64_t M, int64_t K) {
      SrPlan::SplitMat& m = p->split[p->n_split++];
      m.off = off; m.M = (int)M; m.K = (int)K;
      m.ldK = (int)((K + 31) / 32 * 32); m.ldM = (int)((M + 31) / 32 * 32);
      m.f_off = h; h += 2 * M * m.ldK;
      m.t_off = h; h += 2 * K * m.ldM;
      m.g_off = -1;
      m.gt_off = -1;
    };
    add(p->w1, dff, F); add(p->w2, d, dff); add(p->wo1, dff, d);
    for (int l = 0; l < p->nl; ++l) { add(p->lay[l].wd, d, d); add(p->lay[l].wf1, dff, d); add(p->lay[l].wf2, d, dff); }
    // fragment-major split-half copies (ultr_h3_index) of the encoder blocks' three matrices for the fused block kernel
    if (d % 32 == 0 && dff % 32 == 0)
      for (int k = 0; k < p->n_split; ++k) {  // (every matrix: the embedding FFN and the output FFN have fused kernels too)
        SrPlan::SplitMat& m = p->split[k];
        if (m.M % 32 != 0) continue;
        h = (h + 7) & ~(int64_t)7;  // 16-byte aligned: the kernel streams it with 16-byte buffer loads
        m.g_off = h;
        h += 2 * (int64_t)m.M * m.ldK;
      }
    // ... and of the transposes of the encoder blocks' matrices for the fused backward launches (ultr_sr_bwd.hip)
    p->bwd_fused = (d == SR_BWD_D && dff == SR_BWD_DFF) ? 1 : 0;
    if (p->bwd_fused)
      for (int k = 0; k < p->n_split; ++k) {  // (every matrix: the output FFN and the embedding FFN have fused backward launches too)
        SrPlan::SplitMat& m = p->split[k];
        h = (h + 7) & ~(int64_t)7;
        m.gt_off = h;
        h += 2 * (int64_t)((m.K + 31) / 32 * 32) * m.ldM;  // whole 32-column chunks (a ragged last chunk keeps its tail unwritten: those
                                                           // output columns do not exist)
      }
    p->sv_planes = (p->sv_total + 4 + 7) & ~(int64_t)7;
    p->planes_halves = h;
    p->sv_flag = p->sv_planes + (h + 1) / 2;  // (ultr_setrank_saved_bytes leaves four floats behind the planes)
  }
  p->maxw = (int)(F > d ? F : d);
  if (dff > p->maxw) p->maxw = (int)dff;
  int64_t w = 0;
  for (int k = 0; k < 3; ++k) { p->ws_g[k] = w; w += (T * p->maxw + 3) & ~(int64_t)3; }
  p->n_cs = (int)((T + SR_CS_ROWS - 1) / SR_CS_ROWS);
  p->n_lb = (int)((T + SR_LB_ROWS - 1) / SR_LB_ROWS);
  p->ws_part = w; w += (int64_t)p->n_lb * 3 * p->maxw;  // n_lb >= n_cs
  // sum-of-squares partials for ultr_apply_update live at offset 0 of a SEPARATE region at the end (ultr_grad_sumsq
  // writes them at the start of the pointer it is given)
  // weight-gradient split: the largest divisor of T that leaves chunks of >= 512 rows, at most 128 chunks
  p->wg_split = 1;
  for (int sdiv = 2; sdiv <= 128 && T / sdiv >= 512; ++sdiv)
    if (T % sdiv == 0) p->wg_split = sdiv;
  int64_t maxmk = dff * F;
  if (d * dff > maxmk) maxmk = d * dff;
  if (d * d > maxmk) maxmk = d * d;
  int64_t wg_floats = (int64_t)p->wg_split * maxmk;  // the plain chunked kernel (unaligned shapes)
  const int shapes[4][2] = {{(int)dff, (int)F}, {(int)d, (int)dff}, {(int)d, (int)d}, {(int)dff, (int)d}};  // [M, K]
  for (int k = 0; k < 4; ++k) {
    int rps = 0;
    const int64_t need = (int64_t)wgrad_chunks(T, shapes[k][0], shapes[k][1], &rps) * ((int64_t)shapes[k][0] * shapes[k][1] + shapes[k][0]);
    if (need > wg_floats) wg_floats = need;
  }
  p->ws_wg = w; w += wg_floats;
  p->wg_floats = wg_floats;
  p->arena_floats = (int64_t)(3 * p->nl + 4) * ((wg_floats + 3) & ~(int64_t)3) + (int64_t)(2 * p->nl + 5) * p->n_lb * 3 * p->maxw;
  if (p->bwd_fused)  // two per block, the output FFN's, the embedding FFN's
    p->arena_floats += (int64_t)(2 * p->nl + 1) * SR_BWD_MAXWG * (d * dff + dff + 3 * d + 4) + (int64_t)SR_BWD_MAXWG * (2 * F + dff * F + dff + d * dff + d + 4);
  w = (w + 3) & ~(int64_t)3;  // 16-byte pieces (the vector form of sr_fold_all_kernel)
  p->ws_arena = w; w += p->arena_floats;
  p->ws_total = w;
  return true;
}

// ---------------------------------------------------------------------------------------------------------
// row-wise kernels: one wavefront per row, SR_ROWS rows per workgroup
// ---------------------------------------------------------------------------------------------------------
// y = LayerNorm(a [+ (b [+ bias])]) * gamma + beta; optionally stores the pre-norm sum and the statistics.
// (b + bias is the preceding Linear's output: its bias add rides along instead of costing a pass of its own.)
// a_gather: a is the feature matrix and rows are gathered through the doc ids (PAD -> zero row).
__global__ __launch_bounds__(SR_ROWS * 64) void sr_ln_fwd_kernel(const float* a, const float* b /* may alias y */,
                                                                const float* __restrict__ bias /* added to b, may be NULL */,
                                                                const int32_t* __restrict__ docids, int64_t n_docs, int B,
                                                                int L, int64_t T, int W, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ sum_out,
                                                                float* y, float* __restrict__ mean_out,
                                                                float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * SR_ROWS + wave;
  if (n >= T) return;
  const float* ra = a + n * W;
  if (docids != nullptr) {
    const int bb = (int)(n / L), ll = (int)(n % L);
    const int id = docids[(int64_t)ll * B + bb];
    ra = (id >= 0 && id < n_docs) ? a + (int64_t)id * W : nullptr;
  }
  if (W <= 64 * 16) {
    // the row lives in registers between the passes (one read of a and b)
    float v[16];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int c = lane + 64 * k;
      v[k] = 0.f;
      if (c < W) {
        v[k] = ra ? ra[c] : 0.f;
        if (b != nullptr) v[k] += (bias != nullptr) ? b[n * W + c] + bias[c] : b[n * W + c];
        if (sum_out != nullptr) sum_out[n * W + c] = v[k];
      }
      s += v[k];
    }
    const float mean = wave_sum(s) / (float)W;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float dlt = (lane + 64 * k < W) ? v[k] - mean : 0.f;
      q += dlt * dlt;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)W + SR_EPS);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int c = lane + 64 * k;
      if (c < W) y[n * W + c] = (v[k] - mean) * rstd * gamma[c] + beta[c];
    }
    if (lane == 0) {
      if (mean_out) mean_out[n] = mean;
      if (rstd_out) rstd_out[n] = rstd;
    }
    return;
  }
  float s = 0.f;
  for (int c = lane; c < W; c += 64) {
    float v = ra ? ra[c] : 0.f;
    if (b != nullptr) v += (bias != nullptr) ? b[n * W + c] + bias[c] : b[n * W + c];
    if (sum_out != nullptr) sum_out[n * W + c] = v;
    s += v;
  }
  const float mean = wave_sum(s) / (float)W;
  float q = 0.f;
  for (int c = lane; c < W; c += 64) {
    float v = ra ? ra[c] : 0.f;
    if (b != nullptr) v += (bias != nullptr) ? b[n * W + c] + bias[c] : b[n * W + c];
    const float dlt = v - mean;
    q += dlt * dlt;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)W + SR_EPS);
  for (int c = lane; c < W; c += 64) {
    float v = ra ? ra[c] : 0.f;
    if (b != nullptr) v += (bias != nullptr) ? b[n * W + c] + bias[c] : b[n * W + c];
    y[n * W + c] = (v - mean) * rstd * gamma[c] + beta[c];
  }
  if (lane == 0) {
    if (mean_out) mean_out[n] = mean;
    if (rstd_out) rstd_out[n] = rstd;
  }
}

// the input LayerNorm on gathered feature rows with 16-byte accesses (W % 4 == 0, W <= 1024): lane owns the float4
// chunks lane, lane + 64, ... of the row; stores the gathered row (for the backward) and the normalised row
__global__ __launch_bounds__(SR_ROWS * 64) void sr_ln_gather_v4_kernel(const float* __restrict__ feats, const int32_t* __restrict__ docids,
                                                                      int64_t n_docs, int B, int L, int64_t T, int W,
                                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                      float* __restrict__ sum_out, float* __restrict__ y,
                                                                      float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * SR_ROWS + wave;
  if (n >= T) return;
  const int bb = (int)(n / L), ll = (int)(n % L);
  const int id = docids[(int64_t)ll * B + bb];
  const float* ra = (id >= 0 && id < n_docs) ? feats + (int64_t)id * W : nullptr;  // PAD -> zero row
  const int nq = W / 4;
  float4 v[4];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int qd = lane + 64 * k;
    v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (qd < nq) {
      if (ra != nullptr) v[k] = ld4(ra + 4 * qd);
      st4(sum_out + n * W + 4 * qd, v[k]);
    }
    s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  }
  const float mean = wave_sum(s) / (float)W;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (lane + 64 * k < nq) {
      const float dx = v[k].x - mean, dy = v[k].y - mean, dz = v[k].z - mean, dw = v[k].w - mean;
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)W + SR_EPS);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int qd = lane + 64 * k, c = 4 * qd;
    if (qd < nq)
      st4(y + n * W + c, make_float4((v[k].x - mean) * rstd * gamma[c] + beta[c], (v[k].y - mean) * rstd * gamma[c + 1] + beta[c + 1],
                                     (v[k].z - mean) * rstd * gamma[c + 2] + beta[c + 2], (v[k].w - mean) * rstd * gamma[c + 3] + beta[c + 3]));
  }
  if (lane == 0) {
    mean_out[n] = mean;
    rstd_out[n] = rstd;
  }
}

// the residual LayerNorms (W = d_model a multiple of 256, no gather): 16-byte accesses, NV float4 per lane
template <int NV>
__global__ __launch_bounds__(SR_ROWS * 64) void sr_ln_fwd_v4_kernel(const float* a, const float* b /* may alias y */,
                                                                   const float* __restrict__ bias, int64_t T,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float* __restrict__ sum_out, float* y,
                                                                   float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  constexpr int W = NV * 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * SR_ROWS + wave;
  if (n >= T) return;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = 4 * lane + 256 * k;
    const float4 av = ld4(a + n * W + c);
    if (b != nullptr) {  // wave-uniform: b == NULL = `a` already is the pre-norm sum (a GEMM epilogue wrote it)
      float4 bv = ld4(b + n * W + c);
      if (bias != nullptr) {
        bv.x += bias[c]; bv.y += bias[c + 1]; bv.z += bias[c + 2]; bv.w += bias[c + 3];  // parameters: any float offset
      }
      v[k] = make_float4(av.x + bv.x, av.y + bv.y, av.z + bv.z, av.w + bv.w);
      st4(sum_out + n * W + c, v[k]);
    } else {
      v[k] = av;
    }
    s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  }
  const float mean = wave_sum(s) / (float)W;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const float dx = v[k].x - mean, dy = v[k].y - mean, dz = v[k].z - mean, dw = v[k].w - mean;
    q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)W + SR_EPS);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = 4 * lane + 256 * k;
    st4(y + n * W + c, make_float4((v[k].x - mean) * rstd * gamma[c] + beta[c], (v[k].y - mean) * rstd * gamma[c + 1] + beta[c + 1],
                                   (v[k].z - mean) * rstd * gamma[c + 2] + beta[c + 2], (v[k].w - mean) * rstd * gamma[c + 3] + beta[c + 3]));
  }
  if (lane == 0) {
    mean_out[n] = mean;
    rstd_out[n] = rstd;
  }
}

// LayerNorm backward per row: ds = rstd * (g - mean(g) - xh * mean(g * xh)), g = dy * gamma, xh = (s - mean) * rstd.
// ds_out may be NULL (only the parameter gradients are wanted); accumulate: ds_out += instead of =.
__global__ __launch_bounds__(SR_ROWS * 64) void sr_ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ s,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, int64_t T, int W,
                                                                float* __restrict__ ds_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * SR_ROWS + wave;
  if (n >= T || ds_out == nullptr) return;
  const float m = mean[n], r = rstd[n];
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < W; c += 64) {
    const float g = dy[n * W + c] * gamma[c];
    const float xh = (s[n * W + c] - m) * r;
    s1 += g;
    s2 += g * xh;
  }
  s1 = wave_sum(s1) / (float)W;
  s2 = wave_sum(s2) / (float)W;
  for (int c = lane; c < W; c += 64) {
    const float g = dy[n * W + c] * gamma[c];
    const float xh = (s[n * W + c] - m) * r;
    ds_out[n * W + c] = r * (g - s1 - xh * s2);
  }
}

// LayerNorm backward AND every column sum that hangs off it, one pass over dy and s:
//   ds = the row gradient above;  part[blk][0:W] = sum_r dy xhat (dgamma),  [W:2W] = sum_r dy (dbeta),
//   [2W:3W] = sum_r ds (the bias gradient of the Linear that produced s).
// A workgroup owns SR_LB_ROWS rows, a wave every 4th of them (row in registers, KMAX columns per lane); the four
// waves' column partials are folded in fixed order through LDS.
template <int KMAX>
__global__ __launch_bounds__(256) void sr_ln_bwd_cs_kernel(const float* __restrict__ dy, const float* __restrict__ s,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, int64_t T, int W,
                                                           float* __restrict__ ds_out, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [4 waves][3][W]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * SR_LB_ROWS;
  const int64_t r1 = (r0 + SR_LB_ROWS < T) ? r0 + SR_LB_ROWS : T;
  float gm[KMAX], ag[KMAX], ab[KMAX], ad[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int c = lane + 64 * k;
    gm[k] = (c < W) ? gamma[c] : 0.f;
    ag[k] = ab[k] = ad[k] = 0.f;
  }
  const float invw = 1.0f / (float)W;
  for (int64_t n = r0 + wave; n < r1; n += 4) {
    const float m = mean[n], r = rstd[n];
    float g[KMAX], xh[KMAX];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const int c = lane + 64 * k;
      const float d = (c < W) ? dy[n * W + c] : 0.f;
      xh[k] = (c < W) ? (s[n * W + c] - m) * r : 0.f;
      g[k] = d * gm[k];
      s1 += g[k];
      s2 += g[k] * xh[k];
      ag[k] += d * xh[k];
      ab[k] += d;
    }
    s1 = wave_sum(s1) * invw;
    s2 = wave_sum(s2) * invw;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const int c = lane + 64 * k;
      const float v = r * (g[k] - s1 - xh[k] * s2);
      if (c < W) {
        ds_out[n * W + c] = v;
        ad[k] += v;
      }
    }
  }
  float* mine = smem + (size_t)wave * 3 * W;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int c = lane + 64 * k;
    if (c < W) {
      mine[c] = ag[k];
      mine[W + c] = ab[k];
      mine[2 * W + c] = ad[k];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 3 * W; e += 256)
    part[(int64_t)blockIdx.x * 3 * W + e] = ((smem[e] + smem[3 * W + e]) + smem[6 * W + e]) + smem[9 * W + e];
}

// the same with 16-byte accesses (W a multiple of 256): lane owns columns 4 lane .. 4 lane + 3 (+ 256 k)
template <int NV>
__global__ __launch_bounds__(256) void sr_ln_bwd_cs_v4_kernel(const float* __restrict__ dy, const float* __restrict__ s,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma, int64_t T,
                                                              float* __restrict__ ds_out, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [4 waves][3][W]
  constexpr int W = NV * 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * SR_LB_ROWS;
  const int64_t r1 = (r0 + SR_LB_ROWS < T) ? r0 + SR_LB_ROWS : T;
  float4 gm[NV], ag[NV], ab[NV], ad[NV];
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = 4 * lane + 256 * k;
    gm[k] = make_float4(gamma[c], gamma[c + 1], gamma[c + 2], gamma[c + 3]);  // parameters: any float offset
    ag[k] = ab[k] = ad[k] = z4;
  }
  const float invw = 1.0f / (float)W;
  for (int64_t n = r0 + wave; n < r1; n += 4) {
    const float m = mean[n], r = rstd[n];
    float4 g[NV], xh[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = 4 * lane + 256 * k;
      const float4 d = ld4(dy + n * W + c), sv = ld4(s + n * W + c);
      xh[k] = make_float4((sv.x - m) * r, (sv.y - m) * r, (sv.z - m) * r, (sv.w - m) * r);
      g[k] = make_float4(d.x * gm[k].x, d.y * gm[k].y, d.z * gm[k].z, d.w * gm[k].w);
      s1 += (g[k].x + g[k].y) + (g[k].z + g[k].w);
      s2 += (g[k].x * xh[k].x + g[k].y * xh[k].y) + (g[k].z * xh[k].z + g[k].w * xh[k].w);
      ag[k].x += d.x * xh[k].x; ag[k].y += d.y * xh[k].y; ag[k].z += d.z * xh[k].z; ag[k].w += d.w * xh[k].w;
      ab[k].x += d.x; ab[k].y += d.y; ab[k].z += d.z; ab[k].w += d.w;
    }
    s1 = wave_sum(s1) * invw;
    s2 = wave_sum(s2) * invw;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = 4 * lane + 256 * k;
      const float4 v = make_float4(r * (g[k].x - s1 - xh[k].x * s2), r * (g[k].y - s1 - xh[k].y * s2),
                                   r * (g[k].z - s1 - xh[k].z * s2), r * (g[k].w - s1 - xh[k].w * s2));
      st4(ds_out + n * W + c, v);
      ad[k].x += v.x; ad[k].y += v.y; ad[k].z += v.z; ad[k].w += v.w;
    }
  }
  float* mine = smem + (size_t)wave * 3 * W;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = 4 * lane + 256 * k;
    st4(mine + c, ag[k]);
    st4(mine + W + c, ab[k]);
    st4(mine + 2 * W + c, ad[k]);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 3 * W; e += 256)
    part[(int64_t)blockIdx.x * 3 * W + e] = ((smem[e] + smem[3 * W + e]) + smem[6 * W + e]) + smem[9 * W + e];
}

// ---------------------------------------------------------------------------------------------------------
// element-wise epilogues
// ---------------------------------------------------------------------------------------------------------
// column sums (bias gradients, LayerNorm gamma/beta gradients): partials per SR_CS_ROWS rows, then a fixed-order fold
// ---------------------------------------------------------------------------------------------------------
// part[blk][c]      = sum_r a[r][c]                       (mode 0)
// part[blk][c]      = sum_r dy[r][c] * xh[r][c]           (mode 1: gamma gradient; xh from s, mean, rstd)
__global__ __launch_bounds__(256) void sr_colsum_kernel(const float* __restrict__ a, const float* __restrict__ s,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        int64_t T, int W, int mode, float* __restrict__ part) {
  const int64_t r0 = (int64_t)blockIdx.x * SR_CS_ROWS;
  const int64_t r1 = (r0 + SR_CS_ROWS < T) ? r0 + SR_CS_ROWS : T;
  for (int c = threadIdx.x; c < W; c += 256) {
    float acc = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
      float v = a[r * W + c];
      if (mode == 1) v *= (s[r * W + c] - mean[r]) * rstd[r];
      acc += v;
    }
    part[(int64_t)blockIdx.x * W + c] = acc;
  }
}
// LayerNorm parameter gradients in ONE pass over dy: part[blk][c] = sum_r dy xhat, part[blk][W + c] = sum_r dy
__global__ __launch_bounds__(256) void sr_colsum_ln_kernel(const float* __restrict__ dy, const float* __restrict__ s,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           int64_t T, int W, float* __restrict__ part) {
  const int64_t r0 = (int64_t)blockIdx.x * SR_CS_ROWS;
  const int64_t r1 = (r0 + SR_CS_ROWS < T) ? r0 + SR_CS_ROWS : T;
  for (int c = threadIdx.x; c < W; c += 256) {
    float ag = 0.f, ab = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
      const float g = dy[r * W + c];
      ag += g * ((s[r * W + c] - mean[r]) * rstd[r]);
      ab += g;
    }
    part[(int64_t)blockIdx.x * 2 * W + c] = ag;
    part[(int64_t)blockIdx.x * 2 * W + W + c] = ab;
  }
}
// the same with 16-byte accesses (W % 4 == 0, W <= 1024): SR_LB_ROWS rows per workgroup, a wave every 4th row, a lane the
// float4 chunks lane, lane + 64, ...; the four waves' partials are folded in fixed order through LDS
__global__ __launch_bounds__(256) void sr_colsum_ln_v4_kernel(const float* __restrict__ dy, const float* __restrict__ s,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              int64_t T, int W, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [4 waves][2][W]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * SR_LB_ROWS;
  const int64_t r1 = (r0 + SR_LB_ROWS < T) ? r0 + SR_LB_ROWS : T;
  const int nq = W / 4;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 ag[4] = {z4, z4, z4, z4}, ab[4] = {z4, z4, z4, z4};
  for (int64_t n = r0 + wave; n < r1; n += 4) {
    const float m = mean[n], r = rstd[n];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int qd = lane + 64 * k;
      if (qd < nq) {
        const float4 g = ld4(dy + n * W + 4 * qd), sv = ld4(s + n * W + 4 * qd);
        ag[k].x += g.x * ((sv.x - m) * r); ag[k].y += g.y * ((sv.y - m) * r);
        ag[k].z += g.z * ((sv.z - m) * r); ag[k].w += g.w * ((sv.w - m) * r);
        ab[k].x += g.x; ab[k].y += g.y; ab[k].z += g.z; ab[k].w += g.w;
      }
    }
  }
  float* mine = smem + (size_t)wave * 2 * W;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int qd = lane + 64 * k;
    if (qd < nq) {
      st4(mine + 4 * qd, ag[k]);
      st4(mine + W + 4 * qd, ab[k]);
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * W; e += 256)
    part[(int64_t)blockIdx.x * 2 * W + e] = ((smem[e] + smem[2 * W + e]) + smem[4 * W + e]) + smem[6 * W + e];
}
// dst[c] = sum over nparts partials (canonical order)
__global__ __launch_bounds__(256) void sr_fold_kernel(const float* __restrict__ part, int64_t stride, int nparts, int len,
                                                      float* __restrict__ dst) {
  __shared__ float sm[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  sm[grp][lane] = (c < len) ? strided_sum(part + c, stride, nparts, grp) : 0.f;
  __syncthreads();
  if (grp == 0 && c < len) dst[c] = ((sm[0][lane] + sm[1][lane]) + sm[2][lane]) + sm[3][lane];
}

// the same for MANY partials (row-block partials of the LayerNorm / column-sum kernels: hundreds to thousands): 16 columns
// per workgroup, 16 groups of threads each summing every 16th partial in order, fixed-order combine - 4x the workgroups
// and 4x shorter serial chains than sr_fold_kernel
__global__ __launch_bounds__(256) void sr_fold16_kernel(const float* __restrict__ part, int64_t stride, int nparts, int len,
                                                        float* __restrict__ dst) {
  __shared__ float sm[16][16];
  const int cl = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float a = 0.f;
  if (c < len) {
#pragma unroll 8
    for (int k = grp; k < nparts; k += 16) a += part[(int64_t)k * stride + c];
  }
  sm[grp][cl] = a;
  __syncthreads();
  if (grp == 0 && c < len) {
    float t = sm[0][cl];
#pragma unroll
    for (int g = 1; g < 16; ++g) t += sm[g][cl];
    dst[c] = t;
  }
}
// ---- deferred folds -------------------------------------------------------------------------------------------------------
// ultr_setrank_backward produces ~20 sets of partial sums (weight-gradient slabs per row chunk, LayerNorm / bias column
// partials per row block); their folds only feed the update at the very end.  Instead of one 7 us launch behind every
// producer (21 of the step's 69 launches, 0.15 ms) the backward queues them and ONE launch folds all - same arithmetic
// order per job as sr_fold_kernel / sr_fold16_kernel, so the gradients keep their bits.
#define SR_MAX_FOLDS 72
struct FoldJob {
  const float* part;
  float* dst;
  int64_t stride;
  int nparts, len, wide, blk_begin;  // wide: 16 columns per workgroup (many partials) instead of 64
};
struct FoldTable {
  int n, nblocks;
  FoldJob job[SR_MAX_FOLDS];
};
__global__ __launch_bounds__(256) void sr_fold_all_kernel(FoldTable t) {
  __shared__ float sm[256];
  int j = 0;
  while (j + 1 < t.n && (int)blockIdx.x >= t.job[j + 1].blk_begin) ++j;
  const FoldJob jb = t.job[j];
  const int b = (int)blockIdx.x - jb.blk_begin;
  if (jb.wide == 2) {
    // many partials of a long vector (the per-workgroup partials of ultr_sr_bwd.hip: 256 x 66 - 125 KB): the 16-group order of the
    // `wide` form with FOUR columns per thread - a workgroup reads whole 256-byte runs of a partial instead of 64-byte pieces whose
    // other halves a workgroup on another XCD fetched again (45 -> 2x fewer bytes from HBM); same sums, same order per column
    const int cl = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int c = b * 64 + 4 * cl;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < jb.len) {
#pragma unroll 8
      for (int k = grp; k < jb.nparts; k += 16) {
        const float4 v = ld4(jb.part + (int64_t)k * jb.stride + c);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
    }
    __shared__ float4 sm4[256];
    sm4[grp * 16 + cl] = a;
    __syncthreads();
    if (grp == 0 && c < jb.len) {
      float4 v = sm4[cl];
#pragma unroll
      for (int g = 1; g < 16; ++g) {
        const float4 w = sm4[g * 16 + cl];
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
      }
      const float o[4] = {v.x, v.y, v.z, v.w};  // (the destination sits at any float offset of the gradient vector)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (c + i < jb.len) jb.dst[c + i] = o[i];
    }
  } else if (jb.wide) {
    const int cl = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int c = b * 16 + cl;
    float a = 0.f;
    if (c < jb.len) {
#pragma unroll 8
      for (int k = grp; k < jb.nparts; k += 16) a += jb.part[(int64_t)k * jb.stride + c];
    }
    sm[grp * 16 + cl] = a;
    __syncthreads();
    if (grp == 0 && c < jb.len) {
      float v = sm[cl];
#pragma unroll
      for (int g = 1; g < 16; ++g) v += sm[g * 16 + cl];
      jb.dst[c] = v;
    }
  } else {
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = b * 64 + lane;
    sm[grp * 64 + lane] = (c < jb.len) ? strided_sum(jb.part + c, jb.stride, jb.nparts, grp) : 0.f;
    __syncthreads();
    if (grp == 0 && c < jb.len) jb.dst[c] = ((sm[lane] + sm[64 + lane]) + sm[128 + lane]) + sm[192 + lane];
  }
}
struct FoldCtx {
  bool active = false;
  int64_t used = 0, cap = 0;
  float* arena = nullptr;
  FoldTable tab;
};
thread_local FoldCtx g_folds;
// scratch for one producer's partials: a fresh arena piece while a backward is queueing, the shared region otherwise
float* part_scratch(float* shared_region, int64_t floats) {
  FoldCtx& f = g_folds;
  const int64_t need = (floats + 3) & ~(int64_t)3;
  if (f.active && f.used + need <= f.cap && f.tab.n + 2 <= SR_MAX_FOLDS) {
    float* at = f.arena + f.used;
    f.used += need;
    return at;
  }
  return shared_region;
}
// a piece of the arena or NULL (producers whose partials do not fit the shared region)
float* arena_piece(int64_t floats) {
  FoldCtx& f = g_folds;
  const int64_t need = (floats + 3) & ~(int64_t)3;
  if (!(f.active && f.used + need <= f.cap && f.tab.n + 4 <= SR_MAX_FOLDS)) return nullptr;
  float* at = f.arena + f.used;
  f.used += need;
  return at;
}
void fold(const float* part, int64_t stride, int nparts, int len, float* dst, hipStream_t st, bool own_buffer = false) {
  FoldCtx& f = g_folds;
  const bool wide = nparts >= 256;
  const bool in_arena = f.active && part >= f.arena && part < f.arena + f.cap;
  if (f.active && (in_arena || own_buffer) && f.tab.n < SR_MAX_FOLDS) {
    FoldJob& j = f.tab.job[f.tab.n++];
    // (a partial is a 16-byte-aligned run of `stride` = 4 k floats and `part` starts 4 m floats into it: the last float4 of a row stays inside)
    const bool vec4 = wide && len >= 1024 && (stride & 3) == 0 && ((uintptr_t)part & 15) == 0;
    j.part = part; j.dst = dst; j.stride = stride; j.nparts = nparts; j.len = len; j.wide = vec4 ? 2 : (wide ? 1 : 0);
    j.blk_begin = f.tab.nblocks;
    f.tab.nblocks += (wide && !vec4) ? (len + 15) / 16 : (len + 63) / 64;
    return;
  }
  if (wide) hipLaunchKernelGGL(sr_fold16_kernel, dim3((len + 15) / 16), dim3(256), 0, st, part, stride, nparts, len, dst);
  else hipLaunchKernelGGL(sr_fold_kernel, dim3((len + 63) / 64), dim3(256), 0, st, part, stride, nparts, len, dst);
}
// RAII: queue folds for the lifetime of the scope; flush() launches them (an early error return just drops the queue)
struct FoldScope {
  FoldScope(float* arena, int64_t cap) {
    FoldCtx& f = g_folds;
    f.active = true; f.used = 0; f.cap = cap; f.arena = arena; f.tab.n = 0; f.tab.nblocks = 0;
  }
  ~FoldScope() { g_folds.active = false; }
  void flush(hipStream_t st) {
    FoldCtx& f = g_folds;
    if (f.tab.n > 0) hipLaunchKernelGGL(sr_fold_all_kernel, dim3(f.tab.nblocks), dim3(256), 0, st, f.tab);
    f.tab.n = 0; f.tab.nblocks = 0; f.used = 0;
  }
};

// ---------------------------------------------------------------------------------------------------------
// weight gradients of the plain Linears:  dW[M, K] = dY[T, M]^T X[T, K],  db[M] = column sums of dY
// ---------------------------------------------------------------------------------------------------------
// T is ~100k rows and the outputs are small (at most d x d), so the contraction is split over row chunks:
// workgroup = (64 x 64 output block, chunk); its four waves take a quarter of the chunk each, stream the two
// operand rows straight from global memory into MFMA fragments (lane (i, q): row 4 step + q, float4 at column 4 i -
// the four floats feed the four 16-wide sub-tiles, i.e. the block's columns are dealt round-robin to the
// sub-tiles, which the store undoes), 16 accumulator tiles per wave, a three-deep register ring of operands,
// then a fixed-order LDS reduction over the waves and one slab [M*K + M] per chunk; sr_fold_kernel sums the slabs.
// (rocBLAS ran these as strided-batched 32x32 macro-tiles at ~43 TFLOP/s; same design as dnn_wgrad_kernel.)
constexpr int SRW_SPT = 2;  // steps (of 4 rows) per trip
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void sr_wgrad_kernel(const float* __restrict__ dY, const float* __restrict__ X, int64_t T, int M,
                                                       int K, int nkb, int nsplit, int rps, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float (*red)[64 * 64] = reinterpret_cast<float (*)[64 * 64]>(smem);
  float (*bred)[64] = reinterpret_cast<float (*)[64]>(smem + 4 * 64 * 64);
  const int split = blockIdx.x % nsplit, tile = blockIdx.x / nsplit;
  const int mb = tile / nkb, kb = tile - mb * nkb;
  const int m0 = mb * 64, k0 = kb * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int64_t n0 = (int64_t)split * rps;
  const int rows_here = (int)((n0 + rps < T) ? rps : T - n0);
  const int rpw = rps / 4;
  const int rbeg = wave * rpw;
  const int rend = (rbeg + rpw < rows_here) ? rbeg + rpw : rows_here;
  // descriptors cover exactly this chunk: a masked lane presents ULTR_OOB and reads 0
  const Src ys = make_src(dY + n0 * M, (int64_t)rows_here * M);
  const Src xs = make_src(X + n0 * K, (int64_t)rows_here * K);
  const unsigned mc = (unsigned)(m0 + 4 * i) * 4u, kc = (unsigned)(k0 + 4 * i) * 4u;
  const unsigned mstride = (unsigned)M * 4u, kstride = (unsigned)K * 4u;
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  struct StepRegs {
    float4 a, x;
  };
  // only dY must be exactly zero for rows outside the wave's slice; X rows there are other rows of the chunk (finite)
  auto load_step = [&](int r, StepRegs& sr) {
    sr.a = buf_ld4(ys, (r < rend) ? (unsigned)r * mstride + mc : ULTR_OOB);
    sr.x = buf_ld4(xs, (unsigned)r * kstride + kc);
  };
  int rr = rbeg;
  auto trip = [&](StepRegs(&cu)[SRW_SPT], StepRegs(&nx)[SRW_SPT]) {
#pragma unroll
    for (int u = 0; u < SRW_SPT; ++u) load_step(rr + 4 * SRW_SPT * 2 + 4 * u + q, nx[u]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < SRW_SPT; ++u) {
      const float4 a_c = cu[u].a, x_c = cu[u].x;
      bsum.x += a_c.x;
      bsum.y += a_c.y;
      bsum.z += a_c.z;
      bsum.w += a_c.w;
      const float av[4] = {a_c.x, a_c.y, a_c.z, a_c.w};
      const float bv[4] = {x_c.x, x_c.y, x_c.z, x_c.w};
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = mfma16(av[ta], bv[tb], acc[ta][tb]);
    }
    rr += 4 * SRW_SPT;
  };
  const int ntrip = (rend - rbeg + 4 * SRW_SPT - 1) / (4 * SRW_SPT);
  StepRegs ra[SRW_SPT], rb[SRW_SPT], rc[SRW_SPT];
#pragma unroll
  for (int u = 0; u < SRW_SPT; ++u) {
    load_step(rbeg + 4 * u + q, ra[u]);
    load_step(rbeg + 4 * SRW_SPT + 4 * u + q, rb[u]);
  }
  int t = 0;
  for (; t + 2 < ntrip; t += 3) {
    trip(ra, rc);
    trip(rb, ra);
    trip(rc, rb);
  }
  if (t < ntrip) trip(ra, rc);
  if (t + 1 < ntrip) trip(rb, ra);
  // lane holds D_{ta,tb}[row = 4q + r][col = i]  ->  block-local (m = 4 (4q + r) + ta, k = 4 i + tb)
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ml = 4 * (4 * q + r) + ta;
      st4(&red[wave][ml * 64 + 4 * i], make_float4(acc[ta][0][r], acc[ta][1][r], acc[ta][2][r], acc[ta][3][r]));
    }
  {
    float4 sb = bsum;  // the 4 row groups q sit in lanes i, i+16, i+32, i+48
    sb.x = quad_sum(sb.x); sb.y = quad_sum(sb.y); sb.z = quad_sum(sb.z); sb.w = quad_sum(sb.w);
    if (q == 0) st4(&bred[wave][4 * i], sb);
  }
  __syncthreads();
  float* slab = part + (int64_t)split * ((int64_t)M * K + M);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int e = tid + 256 * it;  // float4 index inside the 64x64 block
    const int ml = e >> 4, k4 = (e & 15) * 4;
    const float4 v0 = ld4(&red[0][ml * 64 + k4]), v1 = ld4(&red[1][ml * 64 + k4]);
    const float4 v2 = ld4(&red[2][ml * 64 + k4]), v3 = ld4(&red[3][ml * 64 + k4]);
    float4 o;
    o.x = ((v0.x + v1.x) + v2.x) + v3.x;
    o.y = ((v0.y + v1.y) + v2.y) + v3.y;
    o.z = ((v0.z + v1.z) + v2.z) + v3.z;
    o.w = ((v0.w + v1.w) + v2.w) + v3.w;
    const int m = m0 + ml, k = k0 + k4;
    if (m < M && k < K) st4(slab + (int64_t)m * K + k, o);  // K % 4 == 0
  }
  if (kb == 0 && tid < 64 && m0 + tid < M)
    slab[(int64_t)M * K + m0 + tid] = ((bred[0][tid] + bred[1][tid]) + bred[2][tid]) + bred[3][tid];
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
// The token-local Linear layers (round 1: rocBLAS sgemm + separate bias / ReLU / mask passes) run on the library's own
// LDS-tiled matrix-core GEMM (ultr_gemm.h) with the bias, the ReLU, the ReLU mask of the backward and the accumulation
// into an existing gradient fused into the epilogue.  Shapes the vector path cannot take (contraction or row length not
// a multiple of 4, unaligned bases, single-column outputs) go through plain one-thread-per-output kernels: they only
// occur in toy configurations and for the width-1 scorer.
__global__ __launch_bounds__(256) void sr_gemm_xwT_ref_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                              const float* __restrict__ bias, float* __restrict__ Y, int64_t T,
                                                              int K, int M, int relu) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= T * M) return;
  const int64_t t = e / M;
  const int m = (int)(e - t * M);
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc = fmaf(X[t * K + k], W[(int64_t)m * K + k], acc);
  if (bias != nullptr) acc += bias[m];
  Y[e] = relu ? fmaxf(acc, 0.f) : acc;
}
__global__ __launch_bounds__(256) void sr_gemm_dyw_ref_kernel(const float* __restrict__ dY, const float* __restrict__ W,
                                                              float* __restrict__ dX, const float* __restrict__ mask, int64_t T,
                                                              int K, int M, int accumulate) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= T * K) return;
  const int64_t t = e / K;
  const int k = (int)(e - t * K);
  float acc = accumulate ? dX[e] : 0.f;
  for (int m = 0; m < M; ++m) acc = fmaf(dY[t * M + m], W[(int64_t)m * K + k], acc);
  if (mask != nullptr && !(mask[e] > 0.f)) acc = 0.f;
  dX[e] = acc;
}
// part[chunk][m * K + k] = sum over the chunk's rows of dY[t][m] X[t][k]
__global__ __launch_bounds__(256) void sr_gemm_dyTx_ref_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                               float* __restrict__ part, int64_t rows, int K, int M) {
  const int64_t t0 = (int64_t)blockIdx.y * rows;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < M * K; e += gridDim.x * 256) {
    const int m = e / K, k = e - m * K;
    float acc = 0.f;
    for (int64_t t = t0; t < t0 + rows; ++t) acc = fmaf(dY[t * M + m], X[t * K + k], acc);
    part[(int64_t)blockIdx.y * M * K + e] = acc;
  }
}

// activations (a, c) must be 16-byte aligned with rows a multiple of 4 floats (LDS staging and float4 stores); the weight
// matrix only needs its natural 4-byte alignment: in SetRank's flat parameter vector every encoder matrix sits at an odd
// float offset (the width-1 scorer bias precedes them), and buffer_load_dwordx4 takes dword-aligned addresses
// width-1 output (the scorer, SetRank.py:136): y[t] = x[t] . w + b, one wavefront per row, 4 rows per workgroup
__global__ __launch_bounds__(256) void sr_rowdot_kernel(const float* __restrict__ X, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int64_t T, int K) {
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc = fmaf(X[t * K + k], w[k], acc);
  acc = wave_sum(acc);
  if (lane == 0) y[t] = acc + (bias != nullptr ? bias[0] : 0.f);
}
// its weight gradient: part[blk][k] = sum over SR_CS_ROWS rows of dy[t] X[t][k]
__global__ __launch_bounds__(256) void sr_colsum_w_kernel(const float* __restrict__ dy, const float* __restrict__ X, int64_t T,
                                                          int K, float* __restrict__ part) {
  const int64_t r0 = (int64_t)blockIdx.x * SR_CS_ROWS;
  const int64_t r1 = (r0 + SR_CS_ROWS < T) ? r0 + SR_CS_ROWS : T;
  for (int c = threadIdx.x; c < K; c += 256) {
    float acc = 0.f;
    for (int64_t r = r0; r < r1; ++r) acc = fmaf(dy[r], X[r * K + c], acc);
    part[(int64_t)blockIdx.x * K + c] = acc;
  }
}

// The scorer's backward in ONE pass over oh [T, K] (K = dff <= 256): G[t][k] = oh[t][k] > 0 ? dy[t] * w[k] : 0 (the dgrad
// through the width-1 Linear and the ReLU in front of it), part[blk] = { sum_t dy[t] oh[t][k] (d w) | sum_t dy[t] (d b) } per
// SR_CS_ROWS rows.  Replaces three launches (weighted column sums with a quarter of the lanes busy, a column sum of one
// column, a generic [T, 1] x [1, K] product): 64 -> ~15 us at config 5.  A wave takes every fourth row, a lane every 64th
// column; the four waves' partials are combined in fixed order.
__global__ __launch_bounds__(256) void sr_head_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ oh,
                                                          const float* __restrict__ w, int64_t T, int K, float* __restrict__ G,
                                                          float* __restrict__ part) {
  __shared__ float sm[4][260];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * SR_CS_ROWS;
  const int64_t r1 = (r0 + SR_CS_ROWS < T) ? r0 + SR_CS_ROWS : T;
  float wv[4], aw[4] = {0.f, 0.f, 0.f, 0.f}, ab = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) wv[k] = (lane + 64 * k < K) ? w[lane + 64 * k] : 0.f;
  for (int64_t r = r0 + wave; r < r1; r += 4) {
    const float d = dy[r];
    ab += d;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = lane + 64 * k;
      if (c < K) {
        const float x = oh[r * K + c];
        aw[k] = fmaf(d, x, aw[k]);
        G[r * K + c] = (x > 0.f) ? d * wv[k] : 0.f;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (lane + 64 * k < K) sm[wave][lane + 64 * k] = aw[k];
  if (lane == 0) sm[wave][256] = ab;
  __syncthreads();
  float* dst = part + (int64_t)blockIdx.x * (K + 1);
  for (int c = threadIdx.x; c < K; c += 256) dst[c] = ((sm[0][c] + sm[1][c]) + sm[2][c]) + sm[3][c];
  if (threadIdx.x == 0) dst[K] = ((sm[0][256] + sm[1][256]) + sm[2][256]) + sm[3][256];
}

// ---- split-half weight planes ------------------------------------------------------------------------------------------------
struct SrSplitTable {
  int n;
  SrPlan::SplitMat m[32];
};
// blockIdx.y = 4 * matrix + phase (0: the forward planes [M][ldK], 1: the transposed planes [K][ldM], 2: the fragment-major forward copy
// of the fused block kernel - ultr_h3_index - where the plan has one, 3: the fragment-major copy of the transpose for the fused backward
// launches); one element per thread
__global__ __launch_bounds__(256) void sr_split_planes_kernel(SrSplitTable tb, const float* __restrict__ params, _Float16* __restrict__ planes,
                                                              uint32_t* __restrict__ range_flag) {
  const SrPlan::SplitMat m = tb.m[blockIdx.y / 4];
  const int phase = blockIdx.y % 4;
  if (phase == 3) {  // fragment-major copy of W^T: output column k of W, contraction over m
    if (m.gt_off < 0) return;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)m.K * m.ldM) return;
    const int k = (int)(e / m.ldM), mm = (int)(e - (int64_t)k * m.ldM);
    const float w = mm < m.M ? params[m.off + (int64_t)mm * m.K + k] * ULTR_H3_WSCALE : 0.f;
    const _Float16 hi = (_Float16)w, lo = (_Float16)(w - (float)hi);
    _Float16* dst = planes + m.gt_off;
    dst[ultr_h3_index(k, mm, m.ldM >> 5, 0)] = hi;
    dst[ultr_h3_index(k, mm, m.ldM >> 5, 1)] = lo;
    return;
  }
  if (phase == 2) {
    if (m.g_off < 0) return;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)m.M * m.ldK) return;
    const int mm = (int)(e / m.ldK), k = (int)(e - (int64_t)mm * m.ldK);
    const float w = k < m.K ? params[m.off + (int64_t)mm * m.K + k] * ULTR_H3_WSCALE : 0.f;
    const _Float16 hi = (_Float16)w, lo = (_Float16)(w - (float)hi);
    _Float16* dst = planes + m.g_off;
    dst[ultr_h3_index(mm, k, m.ldK >> 5, 0)] = hi;
    dst[ultr_h3_index(mm, k, m.ldK >> 5, 1)] = lo;
    return;
  }
  const int rows = phase ? m.K : m.M, ld = phase ? m.ldM : m.ldK, cols = phase ? m.M : m.K;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)rows * ld) return;
  const int r = (int)(e / ld), c = (int)(e - (int64_t)r * ld);
  float w = 0.f;
  if (c < cols) w = params[m.off + (phase ? ((int64_t)c * m.K + r) : ((int64_t)r * m.K + c))] * UGEMM_H3_WSCALE;
  // |w| >= 64 raises ULTR_H3_FLAG_NEAR, >= 128 (or NaN: the planes overflow) ULTR_H3_FLAG_OVER - the step's update launch reports the word
  // (ultr_update_desc::range_flag) and the engine switches this model to the fp32 products
  if (!(fabsf(w) < 16384.0f)) flag_or(range_flag, !(fabsf(w) < 32768.0f) ? (ULTR_H3_FLAG_OVER | ULTR_H3_FLAG_NEAR) : ULTR_H3_FLAG_NEAR);
  const _Float16 hi = (_Float16)w, lo = (_Float16)(w - (float)hi);
  _Float16* dst = planes + (phase ? m.t_off : m.f_off);
  dst[e] = hi;
  dst[(int64_t)rows * ld + e] = lo;
}
// the planes of the step in flight (set by ultr_setrank_forward / _backward around their GEMM calls)
struct SrH3Ctx {
  const float* params;
  const _Float16* planes;
  const SrPlan* plan;
};
thread_local SrH3Ctx g_sr_h3 = {nullptr, nullptr, nullptr};
// knobs of this file: read at first use, re-read after ultr_config_reload()
int g_sr_knob_h3 = -1, g_sr_knob_attn_h3 = -1, g_sr_knob_attn_mask = 0, g_sr_knob_wg_h3 = -1, g_sr_knob_bwd_fused = 1;
void sr_knobs_load() {
  if (g_sr_knob_h3 >= 0) return;
  const char* e = getenv("ULTR_SR_H3");
  g_sr_knob_h3 = (e && *e) ? atoi(e) : 1;
  e = getenv("ULTR_SR_ATTN_H3");  // 0: fp32 matrix cores; 1 (default): split-half BACKWARD kernel (the forward always runs on the fp32 matrix cores)
  g_sr_knob_attn_h3 = (e && *e) ? atoi(e) : 1;
  e = getenv("ULTR_SR_ATTN_H3_MASK");  // debug: bit (2 layer + dir), dir 0 forward / 1 backward
  g_sr_knob_attn_mask = (e && *e) ? atoi(e) : -1;
  e = getenv("ULTR_SR_WG_H3");
  g_sr_knob_wg_h3 = (e && *e) ? atoi(e) : 1;
  e = getenv("ULTR_SR_BWD_FUSED");  // bits (default 7 = all): 1 the row-local chain of a block's backward as two launches (ultr_sr_bwd.hip)
  g_sr_knob_bwd_fused = (e && *e) ? atoi(e) : 7;  // instead of seven, 2 the output FFN's backward as one instead of three, 4 the embedding FFN's
}
int sr_h3_enabled() {
  sr_knobs_load();
  return g_sr_knob_h3;
}
const SrPlan::SplitMat* sr_find_split(const float* W, int M, int K) {
  if (g_sr_h3.plan == nullptr || g_sr_h3.planes == nullptr || !sr_h3_enabled()) return nullptr;
  const int64_t off = W - g_sr_h3.params;
  for (int k = 0; k < g_sr_h3.plan->n_split; ++k) {
    const SrPlan::SplitMat& m = g_sr_h3.plan->split[k];
    if (m.off == off && m.M == M && m.K == K) return &m;
  }
  return nullptr;
}

bool vec_ok(const void* a, const void* w, const void* c, int K, int ld_out) {
  (void)w;
  return K % 4 == 0 && ld_out % 4 == 0 && ((((uintptr_t)a | (uintptr_t)c) & 15) == 0);
}
// row-major  Y[T, M] = act(X[T, K] . W[M, K]^T + bias)      (bias may be NULL; relu 0 / 1)
int gemm_xwT(const float* X, const float* W, const float* bias, float* Y, int64_t T, int K, int M, int relu, hipStream_t st) {
  if (M >= 16 && vec_ok(X, W, Y, K, M) && T * (K > M ? K : M) * 4 < ((int64_t)1 << 31)) {
    const ugemm::APlain a{X, T, K, K};
    const ugemm::EBiasAct e{Y, bias, M, relu ? 1 : -1};
    if (const SrPlan::SplitMat* sm = sr_find_split(W, M, K)) {  // split-half planes of W: the product on the fp16 matrix cores
      const ugemm::Dims dh{T, M, K, sm->ldK};
      const _Float16* hi = g_sr_h3.planes + sm->f_off;
      return ugemm::run_h3(dh, a, hi, hi + (int64_t)M * sm->ldK, e, st) == hipSuccess ? 0 : ULTR_E_UNSUPPORTED;
    }
    const ugemm::Dims d{T, M, K, K};
    return ugemm::run<true>(d, a, W, e, st) == hipSuccess ? 0 : ULTR_E_UNSUPPORTED;
  }
  if (M == 1 && !relu) {
    hipLaunchKernelGGL(sr_rowdot_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, st, X, W, bias, Y, T, K);
    return 0;
  }
  hipLaunchKernelGGL(sr_gemm_xwT_ref_kernel, dim3((unsigned)((T * M + 255) / 256)), dim3(256), 0, st, X, W, bias, Y, T, K, M, relu);
  return 0;
}
// S[T, M] = (X[T, K] . W[M, K]^T + bias) + res: a Linear onto the residual stream, fused (the tiled GEMM only); false = the
// shape does not take it and the caller runs the Linear and the residual pass separately
bool gemm_xwT_res(const float* X, const float* W, const float* bias, const float* res, float* S, int64_t T, int K, int M, hipStream_t st) {
  if (!(M >= 16 && M % 4 == 0 && vec_ok(X, W, S, K, M) && (((uintptr_t)res) & 15) == 0 && T * (K > M ? K : M) * 4 < ((int64_t)1 << 31)))
    return false;
  const ugemm::APlain a{X, T, K, K};
  const ugemm::EBiasRes e{S, bias, res, M};
  if (const SrPlan::SplitMat* sm = sr_find_split(W, M, K)) {
    const ugemm::Dims dh{T, M, K, sm->ldK};
    const _Float16* hi = g_sr_h3.planes + sm->f_off;
    return ugemm::run_h3(dh, a, hi, hi + (int64_t)M * sm->ldK, e, st) == hipSuccess;
  }
  const ugemm::Dims d{T, M, K, K};
  return ugemm::run<true>(d, a, W, e, st) == hipSuccess;
}
// row-major  dX[T, K] = (accumulate ? dX : 0) + dY[T, M] . W[M, K], then zeroed where mask <= 0 (mask may be NULL)
int gemm_dyw(const float* dY, const float* W, float* dX, const float* mask, int64_t T, int K, int M, int accumulate, hipStream_t st) {
  if (M % 4 == 0 && K >= 16 && vec_ok(dY, W, dX, K, K) && (mask == nullptr || ((uintptr_t)mask & 15) == 0) &&
      T * (K > M ? K : M) * 4 < ((int64_t)1 << 31)) {
    const ugemm::APlain a{dY, T, M, M};
    const ugemm::EStore e{dX, mask, K, accumulate};
    if (const SrPlan::SplitMat* sm = K >= 16 ? sr_find_split(W, M, K) : nullptr) {  // the planes of W^T: [K][ldM], contraction over M
      const ugemm::Dims dh{T, K, M, sm->ldM};
      const _Float16* hi = g_sr_h3.planes + sm->t_off;
      return ugemm::run_h3(dh, a, hi, hi + (int64_t)K * sm->ldM, e, st) == hipSuccess ? 0 : ULTR_E_UNSUPPORTED;
    }
    const ugemm::Dims d{T, K, M, K};
    return ugemm::run<false>(d, a, W, e, st) == hipSuccess ? 0 : ULTR_E_UNSUPPORTED;
  }
  hipLaunchKernelGGL(sr_gemm_dyw_ref_kernel, dim3((unsigned)((T * K + 255) / 256)), dim3(256), 0, st, dY, W, dX, mask, T, K, M, accumulate);
  return 0;
}
// row-major  dW[M, K] = dY[T, M]^T . X[T, K] for the shapes sr_wgrad_kernel does not take: `wg_split` equal row chunks
// into partials, folded in canonical order
int gemm_dyTx(const SrPlan& p, const float* dY, const float* X, float* dW, int64_t T, int K, int M, float* ws, hipStream_t st) {
  if (M == 1 && K <= 3 * p.maxw) {  // the scorer's weight row: weighted column sums, partials per SR_CS_ROWS rows
    float* cpart = part_scratch(ws + p.ws_part, (int64_t)p.n_cs * K);
    hipLaunchKernelGGL(sr_colsum_w_kernel, dim3(p.n_cs), dim3(256), 0, st, dY, X, T, K, cpart);
    fold(cpart, (int64_t)K, p.n_cs, K, dW, st);
    return 0;
  }
  const int S = p.wg_split;
  const int64_t rows = T / S;
  const int len = M * K;
  float* part = part_scratch(ws + p.ws_wg, (int64_t)S * len);
  hipLaunchKernelGGL(sr_gemm_dyTx_ref_kernel, dim3((len + 255) / 256, S), dim3(256), 0, st, dY, X, part, rows, K, M);
  fold(part, (int64_t)len, S, len, dW, st);
  return 0;
}

// dW = dY^T X and (db != NULL) db = column sums of dY.  Aligned shapes go through sr_wgrad_kernel; the rest through
// the plain chunked kernel + the column-sum kernels.
void colsum(const SrPlan& p, const float* a, const float* s, const float* mean, const float* rstd, int W, int mode, float* ws,
            float* dst, hipStream_t st);
int wgrad(const SrPlan& p, const float* dY, const float* X, float* dW, float* db, int64_t T, int K, int M, float* ws, hipStream_t st) {
  const bool ok = M % 4 == 0 && K % 4 == 0 && (((uintptr_t)dY | (uintptr_t)X) & 15) == 0;
  if (!ok) {
    const int rc = gemm_dyTx(p, dY, X, dW, T, K, M, ws, st);
    if (rc != 0) return rc;
    if (db != nullptr) colsum(p, dY, nullptr, nullptr, nullptr, M, 0, ws, db, st);
    return 0;
  }
  {
    // square-ish products (d x d: 66 % matrix-core occupancy on the fp32 instruction) take the DNN's split-half weight-gradient
    // kernel: same slab layout, so the fold below is the same.  The thin ones (dff = 64 wide) run at their HBM floor already.
    int S2 = 0, rps2 = 0;
    if (sr_h3_enabled() && !p.no_h3 && g_sr_knob_wg_h3 != 0 && M >= 128 && K >= 128 && ultr_wgrad_h3_geometry(T, M, K, &S2, &rps2) &&
        (int64_t)S2 * ((int64_t)M * K + M) <= p.wg_floats) {
      float* part2 = part_scratch(ws + p.ws_wg, (int64_t)S2 * ((int64_t)M * K + M));
      const int rc = ultr_wgrad_h3_plain(dY, X, T, M, K, part2, st);
      if (rc == 0) {
        const int64_t stride2 = (int64_t)M * K + M;
        const int len2 = M * K;
        if (db == dW + len2) {
          fold(part2, stride2, S2, len2 + M, dW, st);
        } else {
          fold(part2, stride2, S2, len2, dW, st);
          if (db != nullptr) fold(part2 + len2, stride2, S2, M, db, st);
        }
        return 0;
      }
      if (rc != ULTR_E_UNSUPPORTED) return rc;
    }
  }
  int rps = 0;
  const int S = wgrad_chunks(T, M, K, &rps);
  if ((int64_t)rps * (M > K ? M : K) * 4 >= ((int64_t)1 << 31)) return ULTR_E_UNSUPPORTED;
  const int nmb = (M + 63) / 64, nkb = (K + 63) / 64;
  float* part = part_scratch(ws + p.ws_wg, (int64_t)S * ((int64_t)M * K + M));
  const size_t lds = (size_t)(4 * 64 * 64 + 4 * 64) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(sr_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
      return ULTR_E_UNSUPPORTED;
    attr_set = true;
  }
  hipLaunchKernelGGL(sr_wgrad_kernel, dim3(nmb * nkb * S), dim3(256), lds, st, dY, X, T, M, K, nkb, S, rps, part);
  const int64_t stride = (int64_t)M * K + M;
  const int len = M * K;
  if (db == dW + len) {  // weight and bias are neighbours in the flat parameter vector, as in the slab: one fold
    fold(part, stride, S, len + M, dW, st);
  } else {
    fold(part, stride, S, len, dW, st);
    if (db != nullptr) fold(part + len, stride, S, M, db, st);
  }
  return 0;
}

// residual LayerNorm forward: y = LN(a + (b + bias)); float4 kernel when the rows allow it
void ln_residual_fwd(const float* a, const float* b, const float* bias, int64_t T, int W, const float* gamma, const float* beta,
                     float* sum_out, float* y, float* mean_out, float* rstd_out, int batch, int L, hipStream_t st) {
  const unsigned rblk = (unsigned)((T + SR_ROWS - 1) / SR_ROWS);
  const bool v4 = (W == 256 || W == 512 || W == 768 || W == 1024) &&
                  ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)sum_out | (uintptr_t)y) & 15) == 0);  // b / sum_out may be NULL
  if (v4 && W == 256) hipLaunchKernelGGL(sr_ln_fwd_v4_kernel<1>, dim3(rblk), dim3(SR_ROWS * 64), 0, st, a, b, bias, T, gamma, beta, sum_out, y, mean_out, rstd_out);
  else if (v4 && W == 512) hipLaunchKernelGGL(sr_ln_fwd_v4_kernel<2>, dim3(rblk), dim3(SR_ROWS * 64), 0, st, a, b, bias, T, gamma, beta, sum_out, y, mean_out, rstd_out);
  else if (v4 && W == 768) hipLaunchKernelGGL(sr_ln_fwd_v4_kernel<3>, dim3(rblk), dim3(SR_ROWS * 64), 0, st, a, b, bias, T, gamma, beta, sum_out, y, mean_out, rstd_out);
  else if (v4) hipLaunchKernelGGL(sr_ln_fwd_v4_kernel<4>, dim3(rblk), dim3(SR_ROWS * 64), 0, st, a, b, bias, T, gamma, beta, sum_out, y, mean_out, rstd_out);
  else
    hipLaunchKernelGGL(sr_ln_fwd_kernel, dim3(rblk), dim3(SR_ROWS * 64), 0, st, a, b, bias, (const int32_t*)nullptr, (int64_t)0, batch, L, T, W,
                       gamma, beta, sum_out, y, mean_out, rstd_out);
}

#define SR_CHECK(call)        \
  do {                        \
    const int rc_ = (call);   \
    if (rc_ != 0) return rc_; \
  } while (0)

// dst[0..W) = column sums of a (mode 0) or of a o xhat (mode 1)
void colsum(const SrPlan& p, const float* a, const float* s, const float* mean, const float* rstd, int W, int mode, float* ws,
            float* dst, hipStream_t st) {
  float* part = part_scratch(ws + p.ws_part, (int64_t)p.n_cs * W);
  hipLaunchKernelGGL(sr_colsum_kernel, dim3(p.n_cs), dim3(256), 0, st, a, s, mean, rstd, p.T, W, mode, part);
  fold(part, (int64_t)W, p.n_cs, W, dst, st);
}

// dst[0..W) = d gamma, dst[W..2W) = d beta (adjacent in the flat layout: <ln>.weight then <ln>.bias)
// dx = LayerNorm backward of dy; dst_gb = dgamma | dbeta; dst_bias = column sums of dx.  W <= 1024.
int ln_bwd_cs(const SrPlan& p, const float* dy, const float* s, const float* mean, const float* rstd, const float* gamma, int W,
              float* dx, float* ws, float* dst_gb, float* dst_bias, hipStream_t st) {
  float* part = part_scratch(ws + p.ws_part, (int64_t)p.n_lb * 3 * W);
  const size_t lds = (size_t)4 * 3 * W * sizeof(float);
  const bool v4 = (W == 256 || W == 512) && ((((uintptr_t)dy | (uintptr_t)s | (uintptr_t)dx) & 15) == 0);
  if (v4 && W == 256) hipLaunchKernelGGL(sr_ln_bwd_cs_v4_kernel<1>, dim3(p.n_lb), dim3(256), lds, st, dy, s, mean, rstd, gamma, p.T, dx, part);
  else if (v4) hipLaunchKernelGGL(sr_ln_bwd_cs_v4_kernel<2>, dim3(p.n_lb), dim3(256), lds, st, dy, s, mean, rstd, gamma, p.T, dx, part);
  else if (W <= 256) hipLaunchKernelGGL(sr_ln_bwd_cs_kernel<4>, dim3(p.n_lb), dim3(256), lds, st, dy, s, mean, rstd, gamma, p.T, W, dx, part);
  else hipLaunchKernelGGL(sr_ln_bwd_cs_kernel<16>, dim3(p.n_lb), dim3(256), lds, st, dy, s, mean, rstd, gamma, p.T, W, dx, part);
  fold(part, (int64_t)3 * W, p.n_lb, 2 * W, dst_gb, st);
  fold(part + 2 * W, (int64_t)3 * W, p.n_lb, W, dst_bias, st);
  return 0;
}
void colsum_ln(const SrPlan& p, const float* dy, const float* s, const float* mean, const float* rstd, int W, float* ws, float* dst,
               hipStream_t st) {
  float* part = part_scratch(ws + p.ws_part, (int64_t)p.n_lb * 2 * W);  // n_lb >= n_cs
  if (W % 4 == 0 && W <= 1024 && ((((uintptr_t)dy | (uintptr_t)s) & 15) == 0)) {
    hipLaunchKernelGGL(sr_colsum_ln_v4_kernel, dim3(p.n_lb), dim3(256), (size_t)8 * W * sizeof(float), st, dy, s, mean, rstd, p.T, W, part);
    fold(part, (int64_t)2 * W, p.n_lb, 2 * W, dst, st);
    return;
  }
  hipLaunchKernelGGL(sr_colsum_ln_kernel, dim3(p.n_cs), dim3(256), 0, st, dy, s, mean, rstd, p.T, W, part);
  fold(part, (int64_t)2 * W, p.n_cs, 2 * W, dst, st);
}

// ---------------------------------------------------------------------------------------------------------
// Everything of an encoder block behind the attention, one launch (round 5): the wide-tile geometry of dnn_fwdw_kernel
// ---------------------------------------------------------------------------------------------------------
//   s1 = x + (A Wd^T + bd),  out1 = LN1(s1),  f = relu(out1 Wf1^T + bf1),  s2 = out1 + (f Wf2^T + bf2),  x' = LN2(s2)
// (SetRank.py:92-111) were two Linear + residual GEMMs, a Linear + ReLU GEMM and two LayerNorm launches: 1.1 GB of HBM traffic per
// block at config 5 for five tensors the backward wants saved (s1, out1, f, s2, x': 0.45 GB) and two it has to read (A, x: 0.2 GB).
// Here a workgroup of sixteen waves owns R <= 60 token rows through all of it: the three products run on split-half fragment
// copies of the weights (384 KB per block at config 5, streamed once per workgroup; built next to the GEMM planes by
// sr_split_planes_kernel), the row tiles of the d-wide products are split between two groups of eight waves (wave = one 32-column
// chunk x two row tiles over the WHOLE contraction: no partial sums), the dff-wide product takes one wave per row tile so that the
// row maximum its output planes are scaled by is a 16-lane reduction; activations live in three LDS buffers; every saved tensor
// leaves through a buffer resource clipped to the rows that exist.
struct SrBlockArgs {
  int R, d, dff;
  int64_t T;
  int64_t bd, bf1, bf2, g1, b1, g2, b2;                  // float offsets into the parameter vector
  int64_t gd, gf1, gf2;                                  // HALF offsets of the fragment-major copies of Wd, Wf1, Wf2 from the planes
  int64_t A, x, s1, m1, r1, out1, f, s2, m2, r2, xn;     // float offsets into `saved`
  int p0, p1, p2, pv;                                    // float offsets into dynamic LDS
  // the last block also runs the output FFN (SetRank.py:136, 153): oh = relu(x' Wo1^T + bo1), score = oh . wo2 + bo2
  int head;
  int skip_out1;  // the backward recomputes out1 from s1 (sr_bwd_proj_kernel): the forward does not write it (105 MB per block at config 5)
  int64_t go1, bo1, wo2, bo2, oh;                        // fragment copy of Wo1 (halves), parameter offsets, saved oh
};

__device__ __forceinline__ float2 buf_ld2(const Src& s, unsigned byte_off) {
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(s.rs, byte_off, 0, 0);
  return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}
// maximum over the 16 lanes of a DPP row (the lanes that hold one accumulator row group)
__device__ __forceinline__ void row16_sum4(float (&v)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] += dpp_or<0xb1>(0.f, v[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] += dpp_or<0x4e>(0.f, v[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] += dpp_or<0x124>(0.f, v[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] += dpp_or<0x128>(0.f, v[k]);
}
__device__ __forceinline__ void row16_max4(float (&v)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) ULTR_DPP_MAX(v[k], "quad_perm:[1,0,3,2] row_mask:0xf");
#pragma unroll
  for (int k = 0; k < 4; ++k) ULTR_DPP_MAX(v[k], "quad_perm:[2,3,0,1] row_mask:0xf");
#pragma unroll
  for (int k = 0; k < 4; ++k) ULTR_DPP_MAX(v[k], "row_ror:4 row_mask:0xf");
#pragma unroll
  for (int k = 0; k < 4; ++k) ULTR_DPP_MAX(v[k], "row_ror:8 row_mask:0xf");
}

// NW = 16: up to 64 rows, one workgroup per CU; NW = 8: up to 32 rows and half the LDS - two workgroups per CU, each other's memory
// waits and phases overlapping
template <int NW>
__global__ __launch_bounds__(NW * 64) void sr_block_fwd_kernel(SrBlockArgs a, const float* __restrict__ params,
                                                            const _Float16* __restrict__ planes, float* __restrict__ sv,
                                                            float* __restrict__ scores) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = NW * 64, RT = 4;  // a wave owns rows wave + NW q, q < 4, in the row-wise phases
  const int R = a.R, d = a.d, dff = a.dff;
  const int ld = d + 8, ldf = dff + 8;  // row strides: floats of an fp32 row = halves of a plane row
  float* P0 = smem + a.p0;   // planes of the current d-wide A operand (attention output, then out1)
  float* P1 = smem + a.p1;   // fp32 rows: s1 -> out1 -> s2
  float* P2 = smem + a.p2;   // planes of f
  float* PV = smem + a.pv;   // bd | g1 | b1 | bf2 | g2 | b2 | bf1 | bo1 | wo2 | bo2 (+ 3 pad)
  float* OS = PV + 6 * d + 3 * dff + 4;  // [64] row scales of the d-wide operand, [64] of f
  const int tid = threadIdx.x, lane_id = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t n0 = (int64_t)blockIdx.x * R;
  const int vr = (int)((a.T - n0) < R ? (a.T - n0) : R);
  const float* pbd = PV, *pg1 = PV + d, *pb1 = PV + 2 * d, *pbf2 = PV + 3 * d, *pg2 = PV + 4 * d, *pb2 = PV + 5 * d, *pbf1 = PV + 6 * d;
  const float invd = 1.0f / (float)d;

  UGEMM_TRACE_STAMP(20);
  // the residual rows of x in the accumulator layout of the first product (wave = chunk x pair of row tiles): requested FIRST - they
  // are the only HBM operand of that product's epilogue and land while the prologue and the product run (measured against a drained
  // request: no difference, 2.57 - 2.59 ms either way - the phases wait for each other, not for HBM)
  float2 xres[2][4];
  {
    const int ch = wave & 7, g = wave >> 3, i = lane_id & 15, q = lane_id >> 4;
    const Src xs = make_src(sv + a.x + n0 * d, (int64_t)vr * d);
    const int col = 32 * ch + 2 * i;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        xres[t][r] = buf_ld2(xs, ch < (d >> 5) ? (unsigned)((16 * (2 * g + t) + 4 * q + r) * d + col) * 4u : ULTR_OOB);
  }
  // ---- prologue: parameter vectors to LDS; the attention rows as two fp16 planes scaled per row -------------------------------
  {
    const int lane = lane_id;
    for (int e = tid; e < 6 * d + dff; e += NT) {
      const int v = e / d;  // 0 .. 5: the d-long vectors; 6 ..: bf1
      const int64_t src = v == 0 ? a.bd : v == 1 ? a.g1 : v == 2 ? a.b1 : v == 3 ? a.bf2 : v == 4 ? a.g2 : v == 5 ? a.b2 : a.bf1;
      PV[e] = params[src + (v < 6 ? e - v * d : e - 6 * d)];
    }
    if (a.head)
      for (int e = tid; e < 2 * dff + 1; e += NT)
        PV[6 * d + dff + e] = params[e < dff ? a.bo1 + e : e < 2 * dff ? a.wo2 + (e - dff) : a.bo2];
    const Src as = make_src(sv + a.A + n0 * d, (int64_t)vr * d);
    const int c = 4 * lane;
    float4 av[RT];
    float am[RT];
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      const int r = wave + NW * q;
      av[q] = buf_ld4(as, c < d ? (unsigned)(r * d + c) * 4u : ULTR_OOB);
      am[q] = fmaxf(fmaxf(fabsf(av[q].x), fabsf(av[q].y)), fmaxf(fabsf(av[q].z), fabsf(av[q].w)));
    }
    wave_max_n<RT>(am);
    _Float16* AH = reinterpret_cast<_Float16*>(P0);
    _Float16* AL = AH + (R + 1) * ld;
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      const int r = wave + NW * q, rc = r < R ? r : R;
      float rs, inv;
      fb_h3_scale(am[q], rs, inv);
      if (c < d) {
        fbh4 hi, lo;
        fb_h3_split4(av[q], rs, hi, lo);
        *reinterpret_cast<fbh4*>(AH + rc * ld + c) = hi;
        *reinterpret_cast<fbh4*>(AL + rc * ld + c) = lo;
      }
      if (lane == 0) OS[r] = inv * (1.0f / ULTR_H3_WSCALE);
    }
  }
  lds_barrier();

  // a d-wide product: wave = (32-column chunk, pair of row tiles); Y = (planes . W) x row scale + bias + residual -> P1 + saved
  auto product_d = [&](const float* Ap, int lda, int nks, int64_t gw, int Kw, const float* bias, const float* os, bool res_lds,
                       int64_t out_off) {
    int lane = lane_id;
    asm volatile("" : "+v"(lane));
    const int nch = d >> 5;
    const int ch = wave & 7, g = wave >> 3;
    const bool has = ch < nch;
    const int i = lane & 15, q = lane >> 4;
    const _Float16* AH = reinterpret_cast<const _Float16*>(Ap);
    const int lo_off = (R + 1) * lda;
    const _Float16* pa[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = 16 * (2 * g + t) + i;
      pa[t] = AH + (row < R ? row : R) * lda + 8 * q;
    }
    const Src Wh = make_src(reinterpret_cast<const float*>(planes + gw), (int64_t)Kw * d / 2 * 2);
    const int col = 32 * ch + 2 * i;
    PipeH3W<2, 2> ph;
    ph.begin(Wh, ch, nks, 0, nks, has, lane);
    f32x4 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) acc[t][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ph.run(pa, lo_off, Wh, has ? nks : 0, acc);
    if (has) {
      const Dst dout = make_dst(sv + out_off + n0 * d, (int64_t)vr * d);
      const float2 bv = *reinterpret_cast<const float2*>(bias + col);
      const unsigned gv = (unsigned)(4 * q * d + 2 * i) * 4u;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float4 o4 = ld4(os + 16 * (2 * g + t) + 4 * q);
        const float o[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * (2 * g + t) + 4 * q + r;
          const int rc = row < R ? row : R;
          float2* dst = reinterpret_cast<float2*>(P1 + rc * ld + col);
          const float2 rv = res_lds ? *dst : xres[t][r];
          const float2 y = make_float2(rv.x + (acc[t][0][r] * o[r] + bv.x), rv.y + (acc[t][1][r] * o[r] + bv.y));
          *dst = y;
          buf_st2(dout, gv, (unsigned)((16 * (2 * g + t) + r) * d + 32 * ch) * 4u, y);
        }
      }
    }
  };
  // LayerNorm of the rows in P1 (a wave owns rows wave + 16 q): statistics and the output to `saved`; planes: the output also stays
  // in P1 (the residual of the next sum) and goes to P0 as the two fp16 planes of the next product's operand
  auto layer_norm = [&](const float* gam, const float* bet, int64_t mean_off, int64_t rstd_off, int64_t out_off, bool to_planes, bool store_out = true) {
    int lane = lane_id;
    asm volatile("" : "+v"(lane));
    const int c = 4 * lane;
    const bool cok = c < d;
    float4 v[RT];
    float s[RT], qv[RT];
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      const int r = wave + NW * q;
      v[q] = cok ? ld4(P1 + (r < R ? r : R) * ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      s[q] = (v[q].x + v[q].y) + (v[q].z + v[q].w);
    }
    wave_sum_n<RT>(s);
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      s[q] *= invd;
      if (cok) {
        v[q].x -= s[q]; v[q].y -= s[q]; v[q].z -= s[q]; v[q].w -= s[q];
      }
      qv[q] = (v[q].x * v[q].x + v[q].y * v[q].y) + (v[q].z * v[q].z + v[q].w * v[q].w);
    }
    wave_sum_n<RT>(qv);
    const Dst dmean = make_dst(sv + mean_off + n0, vr), drstd = make_dst(sv + rstd_off + n0, vr);
    const Dst dout = make_dst(sv + out_off + n0 * d, (int64_t)vr * d);
    const unsigned l0 = lane == 0 ? 0u : ULTR_OOB;
    const float4 g4 = cok ? ld4(gam + c) : make_float4(0.f, 0.f, 0.f, 0.f), b4 = cok ? ld4(bet + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    float am[RT];
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      const int r = wave + NW * q;
      const float rstd = 1.0f / sqrtf(qv[q] * invd + SR_EPS);
      v[q] = make_float4(v[q].x * rstd * g4.x + b4.x, v[q].y * rstd * g4.y + b4.y, v[q].z * rstd * g4.z + b4.z, v[q].w * rstd * g4.w + b4.w);
      buf_st4(dout, (cok && store_out) ? (unsigned)c * 4u : ULTR_OOB, (unsigned)(r * d) * 4u, v[q]);
      buf_st1(dmean, l0, (unsigned)r * 4u, s[q]);
      buf_st1(drstd, l0, (unsigned)r * 4u, rstd);
      am[q] = fmaxf(fmaxf(fabsf(v[q].x), fabsf(v[q].y)), fmaxf(fabsf(v[q].z), fabsf(v[q].w)));
    }
    if (to_planes) {
      wave_max_n<RT>(am);
      _Float16* AH = reinterpret_cast<_Float16*>(P0);
      _Float16* AL = AH + (R + 1) * ld;
#pragma unroll
      for (int q = 0; q < RT; ++q) {
        const int r = wave + NW * q, rc = r < R ? r : R;
        float rs, inv;
        fb_h3_scale(am[q], rs, inv);
        if (cok) {
          st4(P1 + rc * ld + c, v[q]);
          fbh4 hi, lo;
          fb_h3_split4(v[q], rs, hi, lo);
          *reinterpret_cast<fbh4*>(AH + rc * ld + c) = hi;
          *reinterpret_cast<fbh4*>(AL + rc * ld + c) = lo;
        }
        if (lane == 0) OS[r] = inv * (1.0f / ULTR_H3_WSCALE);
      }
    }
  };

  // a dff-wide product with ReLU (dff = 32 / 64 / 128: 1 / 2 / 4 chunks of 32 columns): wave = (row tile, chunk, slice of the contraction),
  // the partial tiles summed slice by slice into P2 as fp32 rows; then every wave finishes its own rows (wave + NW q): bias, ReLU, the
  // row to `saved`, and either the two fp16 planes scaled by the row maximum (the block's f) or the dot product with a vector (the scorer).
  // (The first version gave a whole row tile to ONE wave - the row maximum a 16-lane reduction - and two of eight waves did all of
  // it: ~50 us of the launch's 190.)
  auto product_f = [&](int64_t gw, const float* bias, int64_t out_off, const float* wdot, float bdot) {
    float* F32 = P2;
    {
      int lane = lane_id;
      asm volatile("" : "+v"(lane));
      const int i = lane & 15, q = lane >> 4, nks = d >> 5, nchf = dff >> 5, ksplit = 4 / nchf;
      const int rt = wave >> 2;
      int ch = wave & 3, ks = 0;
      while (ch >= nchf) { ch -= nchf; ++ks; }
      const int len = (nks + ksplit - 1) / ksplit, k0 = ks * len;
      const int cnt = k0 >= nks ? 0 : (k0 + len < nks ? len : nks - k0);
      const _Float16* AH = reinterpret_cast<const _Float16*>(P0);
      const int lo_off = (R + 1) * ld;
      const int rowi = 16 * rt + i;
      const _Float16* pa[1] = {AH + (rowi < R ? rowi : R) * ld + 8 * q + 32 * k0};
      const Src Wh = make_src(reinterpret_cast<const float*>(planes + gw), (int64_t)d * dff);
      f32x4 acc[1][2] = {{(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}}};
      PipeH3W<1, 2> ph;
      ph.begin(Wh, ch, nks, k0, cnt, cnt > 0, lane);
      ph.run(pa, lo_off, Wh, cnt, acc);
      const int col = 32 * ch + 2 * i;
      for (int sl = 0; sl < ksplit; ++sl) {
        if (ks == sl) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * rt + 4 * q + r, rc = row < R ? row : R;
            float2* dst = reinterpret_cast<float2*>(F32 + rc * ldf + col);
            float2 y = make_float2(acc[0][0][r], acc[0][1][r]);
            if (sl > 0) {
              const float2 o = *dst;
              y.x += o.x;
              y.y += o.y;
            }
            *dst = y;
          }
        }
        lds_barrier();
      }
    }
    int lane = lane_id;
    asm volatile("" : "+v"(lane));
    const int c = 4 * lane;
    const bool cok = c < dff;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 b4 = cok ? ld4(bias + c) : z4, w4 = (cok && wdot != nullptr) ? ld4(wdot + c) : z4;
    const Dst df = make_dst(sv + out_off + n0 * dff, (int64_t)vr * dff);
    float4 v[RT];
    float am[RT];
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      const int r = wave + NW * q, rc = r < R ? r : R;
      const float os = OS[r];
      v[q] = cok ? ld4(F32 + rc * ldf + c) : z4;
      v[q] = make_float4(fmaxf(v[q].x * os + b4.x, 0.f), fmaxf(v[q].y * os + b4.y, 0.f), fmaxf(v[q].z * os + b4.z, 0.f), fmaxf(v[q].w * os + b4.w, 0.f));
      buf_st4(df, cok ? (unsigned)c * 4u : ULTR_OOB, (unsigned)(r * dff) * 4u, v[q]);
      am[q] = wdot != nullptr ? (v[q].x * w4.x + v[q].y * w4.y) + (v[q].z * w4.z + v[q].w * w4.w)
                              : fmaxf(fmaxf(v[q].x, v[q].y), fmaxf(v[q].z, v[q].w));
    }
    if (wdot != nullptr) {
      wave_sum_n<RT>(am);
      const Dst dsc = make_dst(scores + n0, vr);
#pragma unroll
      for (int q = 0; q < RT; ++q) buf_st1(dsc, lane == 0 ? 0u : ULTR_OOB, (unsigned)(wave + NW * q) * 4u, am[q] + bdot);
      return;
    }
    wave_max_n<RT>(am);
    lds_barrier();  // every wave holds its rows: the planes may overwrite them
    _Float16* FH = reinterpret_cast<_Float16*>(P2);
    _Float16* FL = FH + (R + 1) * ldf;
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      const int r = wave + NW * q, rc = r < R ? r : R;
      float rs, inv;
      fb_h3_scale(am[q], rs, inv);
      if (cok) {
        fbh4 hi, lo;
        fb_h3_split4(v[q], rs, hi, lo);
        *reinterpret_cast<fbh4*>(FH + rc * ldf + c) = hi;
        *reinterpret_cast<fbh4*>(FL + rc * ldf + c) = lo;
      }
      if (lane == 0) OS[64 + r] = inv * (1.0f / ULTR_H3_WSCALE);
    }
  };

  UGEMM_TRACE_STAMP(21);
  product_d(P0, ld, d >> 5, a.gd, d, pbd, OS, false, a.s1);   // s1 = x + (A Wd^T + bd)
  UGEMM_TRACE_STAMP(22);
  lds_barrier();
  UGEMM_TRACE_STAMP(23);
  layer_norm(pg1, pb1, a.m1, a.r1, a.out1, true, a.skip_out1 == 0);  // out1 = LN1(s1)
  UGEMM_TRACE_STAMP(24);
  lds_barrier();
  UGEMM_TRACE_STAMP(25);
  product_f(a.gf1, pbf1, a.f, nullptr, 0.f);                      // f = relu(out1 Wf1^T + bf1)
  UGEMM_TRACE_STAMP(26);
  lds_barrier();
  UGEMM_TRACE_STAMP(27);
  product_d(P2, ldf, dff >> 5, a.gf2, dff, pbf2, OS + 64, true, a.s2);  // s2 = out1 + (f Wf2^T + bf2)
  UGEMM_TRACE_STAMP(28);
  lds_barrier();
  UGEMM_TRACE_STAMP(29);
  layer_norm(pg2, pb2, a.m2, a.r2, a.xn, a.head != 0);                    // x' = LN2(s2)
  UGEMM_TRACE_STAMP(30);
  if (a.head) {
    lds_barrier();
    const float* pbo1 = PV + 6 * d + dff;
    product_f(a.go1, pbo1, a.oh, pbo1 + dff, pbo1[2 * dff]);               // oh = relu(x' Wo1^T + bo1), score = oh . wo2 + bo2
  }
}


// Will the backward of this step run the encoder blocks' row-local chain as the fused launches of ultr_sr_bwd.hip?  ONE predicate
// for the forward (which then does not write out1: sr_bwd_proj_kernel recomputes it from s1) and for the backward (which then
// refuses to fall back to the launches that read out1).  A function of the plan and the knobs only.
bool sr_bwd_blocks_fused(const SrPlan& p) {
  sr_knobs_load();
  int R = 0, nt = 0, nw = 0;
  return p.bwd_fused && (g_sr_knob_bwd_fused & 1) && sr_h3_enabled() && !p.no_h3 && p.T * (int64_t)p.d * 4 < ((int64_t)1 << 31) &&
         sr_bwd_geometry(p.T, 256, &R, &nt, &nw);
}
// sr_block_fwd_kernel: legal for widths that are multiples of 32 (d <= 256, dff <= 128) with the split-half products on; rows per
// workgroup = whole rounds of one workgroup per CU, as many as the LDS holds (<= 60 at d = 256)
int g_sr_knob_block = -1;
bool block_fwd(const SrPlan& p, int l, const float* params, float* sv, float* scores, hipStream_t st, int* rc) {
  if (g_sr_knob_block < 0) {
    const char* e = getenv("ULTR_SR_BLOCK");
    g_sr_knob_block = (e && *e) ? atoi(e) : 3;  // 0: off; 1: one 16-wave workgroup per CU; 2: two 8-wave workgroups per CU; 3 (default): the persistent kernel of round 6 where its widths apply, 2 elsewhere
  }
  const int d = p.d, dff = p.dff;
  if (!g_sr_knob_block || !sr_h3_enabled() || p.no_h3 || g_sr_h3.planes == nullptr || d % 32 != 0 || (dff != 32 && dff != 64 && dff != 128) || d > 256 || d < 32)
    return false;
  const SrPlan::SplitMat* md = sr_find_split(params + p.lay[l].wd, d, d);
  const SrPlan::SplitMat* m1 = sr_find_split(params + p.lay[l].wf1, dff, d);
  const SrPlan::SplitMat* m2 = sr_find_split(params + p.lay[l].wf2, d, dff);
  if (!md || !m1 || !m2 || md->g_off < 0 || m1->g_off < 0 || m2->g_off < 0 || (((uintptr_t)sv | (uintptr_t)g_sr_h3.planes) & 15) != 0) return false;
  if (g_sr_knob_block >= 3 && d == SR_BWD_D && dff == SR_BWD_DFF && p.T * (int64_t)d * 4 < ((int64_t)1 << 31)) {
    // round 6: ONE persistent 8-wave workgroup per CU over 60-row tiles (sr_fwd_block_kernel, ultr_sr_fwd.hip)
    int dev = 0, cus = 256, R = 0, nt = 0, nw = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (sr_bwd_geometry(p.T, cus, &R, &nt, &nw)) {
      const SrLayer& y = p.lay[l];
      SrFwdBlockArgs fa;
      memset(&fa, 0, sizeof(fa));
      fa.R = R; fa.d = d; fa.dff = dff; fa.ntiles = nt; fa.T = p.T;
      fa.bd = y.bd; fa.bf1 = y.bf1; fa.bf2 = y.bf2; fa.g1 = y.g1; fa.b1 = y.b1; fa.g2 = y.g2; fa.b2 = y.b2;
      fa.gd = md->g_off; fa.gf1 = m1->g_off; fa.gf2 = m2->g_off;
      fa.A = p.sv_A[l]; fa.x = p.sv_x[l]; fa.s1 = p.sv_s1[l]; fa.m1 = p.sv_m1[l]; fa.r1 = p.sv_r1[l]; fa.out1 = p.sv_out1[l]; fa.f = p.sv_f[l];
      fa.s2 = p.sv_s2[l]; fa.m2 = p.sv_m2[l]; fa.r2 = p.sv_r2[l]; fa.xn = p.sv_x[l + 1];
      fa.skip_out1 = sr_bwd_blocks_fused(p) ? 1 : 0;
      if (l == p.nl - 1 && p.bo2 == p.wo2 + dff) {
        const SrPlan::SplitMat* mo = sr_find_split(params + p.wo1, dff, d);
        if (mo != nullptr && mo->g_off >= 0 && scores != nullptr) {
          fa.head = 1;
          fa.go1 = mo->g_off; fa.bo1 = p.bo1; fa.wo2 = p.wo2; fa.bo2 = p.bo2; fa.oh = p.sv_oh;
        }
      }
      *rc = sr_fwd_block_launch(fa, nw, params, g_sr_h3.planes, sv, scores, st);
      if (fa.head && *rc == 0) *rc = -1;
      return true;
    }
  }
  const int64_t per_row = (int64_t)(2 * (d + 8) + (dff + 8)) * 4, fixed = (int64_t)(6 * d + 3 * dff + 4 + 128) * 4;
  const bool two = g_sr_knob_block != 1;  // two 8-wave workgroups per CU (default) / 1: one 16-wave workgroup
  int64_t rmax = ((two ? 80 : 160) * 1024 - fixed) / per_row - 1;
  if (rmax > (two ? 32 : 64)) rmax = two ? 32 : 64;
  if (rmax < 16) return false;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  const int64_t slots = (int64_t)cus * (two ? 2 : 1);
  const int64_t rounds = (p.T + slots * rmax - 1) / (slots * rmax);
  int64_t R = (p.T + slots * rounds - 1) / (slots * rounds);
  if (R < 16) R = 16;
  SrBlockArgs a;
  memset(&a, 0, sizeof(a));
  a.R = (int)R; a.d = d; a.dff = dff; a.T = p.T;
  const SrLayer& y = p.lay[l];
  a.bd = y.bd; a.bf1 = y.bf1; a.bf2 = y.bf2; a.g1 = y.g1; a.b1 = y.b1; a.g2 = y.g2; a.b2 = y.b2;
  a.gd = md->g_off; a.gf1 = m1->g_off; a.gf2 = m2->g_off;
  a.A = p.sv_A[l]; a.x = p.sv_x[l]; a.s1 = p.sv_s1[l]; a.m1 = p.sv_m1[l]; a.r1 = p.sv_r1[l]; a.out1 = p.sv_out1[l]; a.f = p.sv_f[l];
  a.s2 = p.sv_s2[l]; a.m2 = p.sv_m2[l]; a.r2 = p.sv_r2[l]; a.xn = p.sv_x[l + 1];
  a.skip_out1 = sr_bwd_blocks_fused(p) ? 1 : 0;
  if (l == p.nl - 1 && p.bo2 == p.wo2 + dff) {  // the output FFN rides along with the last block
    const SrPlan::SplitMat* mo = sr_find_split(params + p.wo1, dff, d);
    if (mo != nullptr && mo->g_off >= 0 && scores != nullptr) {
      a.head = 1;
      a.go1 = mo->g_off; a.bo1 = p.bo1; a.wo2 = p.wo2; a.bo2 = p.bo2; a.oh = p.sv_oh;
    }
  }
  a.p0 = 0;
  a.p1 = (int)((R + 1) * (d + 8));
  a.p2 = 2 * a.p1;
  a.pv = a.p2 + (int)((R + 1) * (dff + 8));
  const size_t lds = (size_t)(a.pv + 6 * d + 3 * dff + 4 + 128) * sizeof(float);
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(two ? reinterpret_cast<const void*>(sr_block_fwd_kernel<8>) : reinterpret_cast<const void*>(sr_block_fwd_kernel<16>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    *rc = ULTR_E_UNSUPPORTED;
    return true;
  }
  if (two) hipLaunchKernelGGL(sr_block_fwd_kernel<8>, dim3((unsigned)((p.T + R - 1) / R)), dim3(512), lds, st, a, params, g_sr_h3.planes, sv, scores);
  else hipLaunchKernelGGL(sr_block_fwd_kernel<16>, dim3((unsigned)((p.T + R - 1) / R)), dim3(1024), lds, st, a, params, g_sr_h3.planes, sv, scores);
  *rc = (int)hipGetLastError();
  if (a.head && *rc == 0) *rc = -1;  // (-1: launched, and the output FFN is done as well)
  return true;
}
// ---------------------------------------------------------------------------------------------------------
// The input side in one launch (round 5): gather + LayerNorm + embedding FFN (SetRank.py:134-135, 146)
// ---------------------------------------------------------------------------------------------------------
//   xg = features[ids],  xn0 = LN_in(xg),  h0 = relu(xn0 W1^T + b1),  x_0 = h0 W2^T + b2      (was: a gather + LayerNorm launch and two GEMMs)
// Same geometry as sr_block_fwd_kernel: a workgroup owns R token rows, the two products run on the fragment-major split-half copies.
struct SrEmbedArgs {
  int R, F, d, dff;
  int64_t T, n_docs;
  int B, L;
  int64_t g_in, b_in, b1, b2;              // float offsets into the parameter vector
  int64_t gw1, gw2;                        // HALF offsets of the fragment copies of W1 [dff][F] and W2 [d][dff]
  int64_t xg, mean_in, rstd_in, xn0, h0, x0;  // float offsets into `saved`
  int p0, p2, pv;                          // float offsets into dynamic LDS
};

template <int NW>
__global__ __launch_bounds__(NW * 64) void sr_embed_fwd_kernel(SrEmbedArgs a, const float* __restrict__ params, const float* __restrict__ feats,
                                                              const int32_t* __restrict__ docids, const _Float16* __restrict__ planes,
                                                              float* __restrict__ sv) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = NW * 64, RT = 4;
  const int R = a.R, F = a.F, d = a.d, dff = a.dff;
  const int K16 = round_up(F, 32), ld0 = K16 + 8, ldf = dff + 8;
  float* P0 = smem + a.p0;   // planes of xn0
  float* P2 = smem + a.p2;   // fp32 partial tiles of h0, then its planes
  float* PV = smem + a.pv;   // g_in | b_in | b1 | b2
  float* OS = PV + 2 * F + dff + d;
  const int tid = threadIdx.x, lane_id = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t n0 = (int64_t)blockIdx.x * R;
  const int vr = (int)((a.T - n0) < R ? (a.T - n0) : R);
  const float* pg = PV, *pb = PV + F, *pb1 = PV + 2 * F, *pb2 = PV + 2 * F + dff;

  // ---- gather + LayerNorm: a wave owns rows wave + NW q; lane q < 4 resolves the id of row q ------------------------------------
  {
    const int lane = lane_id;
    for (int e = tid; e < 2 * F + dff + d; e += NT)
      PV[e] = params[e < F ? a.g_in + e : e < 2 * F ? a.b_in + (e - F) : e < 2 * F + dff ? a.b1 + (e - 2 * F) : a.b2 + (e - 2 * F - dff)];
    const int rme = wave + NW * (lane < RT ? lane : 0);
    const bool idok = lane < RT && rme < vr;
    const uint32_t nme = idok ? (uint32_t)(n0 + rme) : 0u;
    const int bb = (int)(nme / (uint32_t)a.L), ll = (int)(nme % (uint32_t)a.L);
    const int id_raw = docids[(int64_t)ll * a.B + bb];
    const int myid = (idok && id_raw >= 0 && id_raw < a.n_docs) ? id_raw : -1;  // PAD -> zero row
    const Src fs = make_src(feats, a.n_docs * F);
    const int c = 4 * lane;
    const bool cok = c < F;
    float4 v[RT];
    float s[RT], qv[RT];
    const Dst dxg = make_dst(sv + a.xg + n0 * F, (int64_t)vr * F);
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      const int id = __builtin_amdgcn_readlane(myid, q);
      v[q] = buf_ld4(fs, (id >= 0 && cok) ? (unsigned)(((int64_t)id * F + c) * 4) : ULTR_OOB);
    }
    lds_barrier();  // the parameter vectors are in LDS
    const float invF = 1.0f / (float)F;
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      buf_st4(dxg, cok ? (unsigned)c * 4u : ULTR_OOB, (unsigned)((wave + NW * q) * F) * 4u, v[q]);
      s[q] = (v[q].x + v[q].y) + (v[q].z + v[q].w);
    }
    wave_sum_n<RT>(s);
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      s[q] *= invF;
      if (cok) {
        v[q].x -= s[q]; v[q].y -= s[q]; v[q].z -= s[q]; v[q].w -= s[q];
      }
      qv[q] = (v[q].x * v[q].x + v[q].y * v[q].y) + (v[q].z * v[q].z + v[q].w * v[q].w);
    }
    wave_sum_n<RT>(qv);
    const Dst dmean = make_dst(sv + a.mean_in + n0, vr), drstd = make_dst(sv + a.rstd_in + n0, vr);
    const Dst dxn = make_dst(sv + a.xn0 + n0 * F, (int64_t)vr * F);
    const unsigned l0 = lane == 0 ? 0u : ULTR_OOB;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 g4 = cok ? ld4(pg + c) : z4, b4 = cok ? ld4(pb + c) : z4;
    float am[RT];
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      const int r = wave + NW * q;
      const float rstd = 1.0f / sqrtf(qv[q] * invF + SR_EPS);
      v[q] = make_float4(v[q].x * rstd * g4.x + b4.x, v[q].y * rstd * g4.y + b4.y, v[q].z * rstd * g4.z + b4.z, v[q].w * rstd * g4.w + b4.w);
      buf_st4(dxn, cok ? (unsigned)c * 4u : ULTR_OOB, (unsigned)(r * F) * 4u, v[q]);
      buf_st1(dmean, l0, (unsigned)r * 4u, s[q]);
      buf_st1(drstd, l0, (unsigned)r * 4u, rstd);
      am[q] = fmaxf(fmaxf(fabsf(v[q].x), fabsf(v[q].y)), fmaxf(fabsf(v[q].z), fabsf(v[q].w)));
    }
    wave_max_n<RT>(am);
    _Float16* AH = reinterpret_cast<_Float16*>(P0);
    _Float16* AL = AH + (R + 1) * ld0;
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      const int r = wave + NW * q, rc = r < R ? r : R;
      float rs, inv;
      fb_h3_scale(am[q], rs, inv);
      if (c < K16) {  // (columns F .. K16 - 1: zeros - v is 0 * rstd * 0 + 0 there)
        fbh4 hi, lo;
        fb_h3_split4(v[q], rs, hi, lo);
        *reinterpret_cast<fbh4*>(AH + rc * ld0 + c) = hi;
        *reinterpret_cast<fbh4*>(AL + rc * ld0 + c) = lo;
      }
      if (lane == 0) OS[r] = inv * (1.0f / ULTR_H3_WSCALE);
    }
  }
  lds_barrier();
  // ---- h0 = relu(xn0 W1^T + b1): wave = (row tile, chunk, slice of the contraction) -> fp32 partial tiles in P2; then the owner waves
  {
    float* F32 = P2;
    {
      int lane = lane_id;
      asm volatile("" : "+v"(lane));
      const int i = lane & 15, q = lane >> 4, nks = K16 >> 5, nchf = dff >> 5, ksplit = 4 / nchf;
      const int rt = wave >> 2;
      int ch = wave & 3, ks = 0;
      while (ch >= nchf) { ch -= nchf; ++ks; }
      const int len = (nks + ksplit - 1) / ksplit, k0 = ks * len;
      const int cnt = k0 >= nks ? 0 : (k0 + len < nks ? len : nks - k0);
      const _Float16* AH = reinterpret_cast<const _Float16*>(P0);
      const int lo_off = (R + 1) * ld0;
      const int rowi = 16 * rt + i;
      const _Float16* pa[1] = {AH + (rowi < R ? rowi : R) * ld0 + 8 * q + 32 * k0};
      const Src Wh = make_src(reinterpret_cast<const float*>(planes + a.gw1), (int64_t)K16 * dff);
      f32x4 acc[1][2] = {{(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}}};
      PipeH3W<1, 2> ph;
      ph.begin(Wh, ch, nks, k0, cnt, cnt > 0, lane);
      ph.run(pa, lo_off, Wh, cnt, acc);
      const int col = 32 * ch + 2 * i;
      for (int sl = 0; sl < ksplit; ++sl) {
        if (ks == sl) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * rt + 4 * q + r, rc = row < R ? row : R;
            float2* dst = reinterpret_cast<float2*>(F32 + rc * ldf + col);
            float2 y = make_float2(acc[0][0][r], acc[0][1][r]);
            if (sl > 0) {
              const float2 o = *dst;
              y.x += o.x;
              y.y += o.y;
            }
            *dst = y;
          }
        }
        lds_barrier();
      }
    }
    int lane = lane_id;
    asm volatile("" : "+v"(lane));
    const int c = 4 * lane;
    const bool cok = c < dff;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 b4 = cok ? ld4(pb1 + c) : z4;
    const Dst dh = make_dst(sv + a.h0 + n0 * dff, (int64_t)vr * dff);
    float4 v[RT];
    float am[RT];
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      const int r = wave + NW * q, rc = r < R ? r : R;
      const float os = OS[r];
      v[q] = cok ? ld4(F32 + rc * ldf + c) : z4;
      v[q] = make_float4(fmaxf(v[q].x * os + b4.x, 0.f), fmaxf(v[q].y * os + b4.y, 0.f), fmaxf(v[q].z * os + b4.z, 0.f), fmaxf(v[q].w * os + b4.w, 0.f));
      buf_st4(dh, cok ? (unsigned)c * 4u : ULTR_OOB, (unsigned)(r * dff) * 4u, v[q]);
      am[q] = fmaxf(fmaxf(v[q].x, v[q].y), fmaxf(v[q].z, v[q].w));
    }
    wave_max_n<RT>(am);
    lds_barrier();  // every wave holds its rows: the planes may overwrite them
    _Float16* FH = reinterpret_cast<_Float16*>(P2);
    _Float16* FL = FH + (R + 1) * ldf;
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      const int r = wave + NW * q, rc = r < R ? r : R;
      float rs, inv;
      fb_h3_scale(am[q], rs, inv);
      if (cok) {
        fbh4 hi, lo;
        fb_h3_split4(v[q], rs, hi, lo);
        *reinterpret_cast<fbh4*>(FH + rc * ldf + c) = hi;
        *reinterpret_cast<fbh4*>(FL + rc * ldf + c) = lo;
      }
      if (lane == 0) OS[64 + r] = inv * (1.0f / ULTR_H3_WSCALE);
    }
  }
  lds_barrier();
  // ---- x_0 = h0 W2^T + b2: wave = (32-column chunk, pair of row tiles), straight to `saved` --------------------------------------
  {
    int lane = lane_id;
    asm volatile("" : "+v"(lane));
    const int nch = d >> 5, nks = dff >> 5;
    const int ch = wave & 7, g = wave >> 3;
    const bool has = ch < nch;
    const int i = lane & 15, q = lane >> 4;
    const _Float16* AH = reinterpret_cast<const _Float16*>(P2);
    const int lo_off = (R + 1) * ldf;
    const _Float16* pa[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = 16 * (2 * g + t) + i;
      pa[t] = AH + (row < R ? row : R) * ldf + 8 * q;
    }
    const Src Wh = make_src(reinterpret_cast<const float*>(planes + a.gw2), (int64_t)dff * d);
    PipeH3W<2, 2> ph;
    ph.begin(Wh, ch, nks, 0, nks, has, lane);
    f32x4 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) acc[t][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ph.run(pa, lo_off, Wh, has ? nks : 0, acc);
    if (has) {
      const int col = 32 * ch + 2 * i;
      const Dst dout = make_dst(sv + a.x0 + n0 * d, (int64_t)vr * d);
      const float2 bv = *reinterpret_cast<const float2*>(pb2 + col);
      const unsigned gv = (unsigned)(4 * q * d + 2 * i) * 4u;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float4 o4 = ld4(OS + 64 + 16 * (2 * g + t) + 4 * q);
        const float o[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
          buf_st2(dout, gv, (unsigned)((16 * (2 * g + t) + r) * d + 32 * ch) * 4u,
                  make_float2(acc[t][0][r] * o[r] + bv.x, acc[t][1][r] * o[r] + bv.y));
      }
    }
  }
}

// legal: F a multiple of 4 and at most 256, widths as for sr_block_fwd_kernel; two 8-wave workgroups per CU
bool embed_fwd(const SrPlan& p, const float* params, const float* feats, const int32_t* docids, int64_t n_docs, int batch, int L, float* sv,
               hipStream_t st, int* rc) {
  const int F = p.F, d = p.d, dff = p.dff;
  if (g_sr_knob_block < 0) {
    const char* e = getenv("ULTR_SR_BLOCK");
    g_sr_knob_block = (e && *e) ? atoi(e) : 3;
  }
  if (!g_sr_knob_block || !sr_h3_enabled() || p.no_h3 || g_sr_h3.planes == nullptr || F % 4 != 0 || F > 256 || d % 32 != 0 || d > 256 || d < 32 ||
      (dff != 32 && dff != 64 && dff != 128) || n_docs * (int64_t)F * 4 >= ((int64_t)1 << 31))
    return false;
  const SrPlan::SplitMat* m1 = sr_find_split(params + p.w1, dff, F);
  const SrPlan::SplitMat* m2 = sr_find_split(params + p.w2, d, dff);
  if (!m1 || !m2 || m1->g_off < 0 || m2->g_off < 0 || (((uintptr_t)sv | (uintptr_t)g_sr_h3.planes | (uintptr_t)feats) & 15) != 0) return false;
  const int K16 = (F + 31) / 32 * 32;
  const int64_t per_row = (int64_t)((K16 + 8) + (dff + 8)) * 4, fixed = (int64_t)(2 * F + dff + d + 128) * 4;
  int64_t rmax = (80 * 1024 - fixed) / per_row - 1;
  if (rmax > 32) rmax = 32;
  if (rmax < 16) return false;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  const int64_t slots = (int64_t)cus * 2;
  const int64_t rounds = (p.T + slots * rmax - 1) / (slots * rmax);
  int64_t R = (p.T + slots * rounds - 1) / (slots * rounds);
  if (R < 16) R = 16;
  SrEmbedArgs a;
  memset(&a, 0, sizeof(a));
  a.R = (int)R; a.F = F; a.d = d; a.dff = dff; a.T = p.T; a.n_docs = n_docs; a.B = batch; a.L = L;
  a.g_in = p.g_in; a.b_in = p.b_in; a.b1 = p.b1; a.b2 = p.b2;
  a.gw1 = m1->g_off; a.gw2 = m2->g_off;
  a.xg = p.sv_xg; a.mean_in = p.sv_mean_in; a.rstd_in = p.sv_rstd_in; a.xn0 = p.sv_xn0; a.h0 = p.sv_h0; a.x0 = p.sv_x[0];
  a.p0 = 0;
  a.p2 = (int)((R + 1) * (K16 + 8));
  a.pv = a.p2 + (int)((R + 1) * (dff + 8));
  const size_t lds = (size_t)(a.pv + 2 * F + dff + d + 128) * sizeof(float);
  if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(sr_embed_fwd_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int)lds) != hipSuccess) {
    *rc = ULTR_E_UNSUPPORTED;
    return true;
  }
  hipLaunchKernelGGL(sr_embed_fwd_kernel<8>, dim3((unsigned)((p.T + R - 1) / R)), dim3(512), lds, st, a, params, feats, docids, g_sr_h3.planes, sv);
  *rc = (int)hipGetLastError();
  return true;
}

}  // namespace

void ultr_setrank_knobs_reload() {
  g_sr_knob_h3 = -1;
  g_sr_knob_block = -1;
}

extern "C" int64_t ultr_setrank_param_count(const ultr_setrank_desc* c) {
  SrPlan p;
  return make_plan(c, 0, &p) ? p.P : 0;
}
extern "C" int64_t ultr_setrank_saved_bytes(const ultr_setrank_desc* c, int64_t n_rows) {
  SrPlan p;
  return (n_rows >= 0 && make_plan(c, n_rows, &p)) ? (p.sv_planes + (p.planes_halves + 1) / 2 + 4) * (int64_t)sizeof(float) : 0;
}
extern "C" int64_t ultr_setrank_range_flag_offset(const ultr_setrank_desc* c, int64_t n_rows) {
  SrPlan p;
  return (n_rows >= 0 && make_plan(c, n_rows, &p)) ? p.sv_flag : -1;
}
extern "C" int64_t ultr_setrank_workspace_bytes(const ultr_setrank_desc* c, int64_t n_rows) {
  SrPlan p;
  return (n_rows >= 0 && make_plan(c, n_rows, &p)) ? (p.ws_total + 4) * (int64_t)sizeof(float) : 0;
}

extern "C" int ultr_setrank_forward(const ultr_setrank_desc* c, const float* params, const float* features, int64_t n_docs,
                                    const int32_t* docids, int32_t batch, int32_t list_size, float* scores, void* saved,
                                    void* stream) {
  if (!params || !docids || !scores || !saved || batch <= 0 || list_size <= 0 || n_docs < 0 || (n_docs > 0 && !features))
    return ULTR_E_BADARG;
  const int64_t T = (int64_t)batch * list_size;
  SrPlan p;
  if (!make_plan(c, T, &p)) return ULTR_E_BADARG;
  const int L = list_size;
  const SrAttnShape ash = {p.d, p.dh, p.H, p.att_f16};
  SR_CHECK(sr_attn_supported(ash, L, 0));
  if (T > 0x7fffffff / 4) return ULTR_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  float* sv = (float*)saved;
  const unsigned rblk = (unsigned)((T + SR_ROWS - 1) / SR_ROWS);
  const int F = p.F, d = p.d, dff = p.dff;
  // split-half planes of every weight matrix (the weights change with every update: rebuilt per forward, ~2 MB, one small launch)
  struct H3Scope {
    ~H3Scope() { g_sr_h3 = {nullptr, nullptr, nullptr}; }
  } h3scope;
  SR_CHECK((int)hipMemsetAsync(sv + p.sv_flag, 0, sizeof(uint32_t), st));  // this step's range word
  if (sr_h3_enabled() && !p.no_h3) {
    _Float16* planes = reinterpret_cast<_Float16*>(sv + p.sv_planes);
    SrSplitTable tb;
    tb.n = p.n_split;
    int64_t maxe = 0;
    for (int k = 0; k < p.n_split; ++k) {
      tb.m[k] = p.split[k];
      const int64_t a = (int64_t)p.split[k].M * p.split[k].ldK, b = (int64_t)p.split[k].K * p.split[k].ldM;
      maxe = a > maxe ? a : maxe;
      maxe = b > maxe ? b : maxe;
    }
    hipLaunchKernelGGL(sr_split_planes_kernel, dim3((unsigned)((maxe + 255) / 256), (unsigned)(4 * p.n_split)), dim3(256), 0, st, tb, params, planes,
                       reinterpret_cast<uint32_t*>(sv + p.sv_flag));
    g_sr_h3 = {params, planes, &p};
  }
  // input LayerNorm on the gathered rows, then the embedding FFN (SetRank.py:134-135, 146)
  int erc = 0;
  const bool embedded = n_docs > 0 && embed_fwd(p, params, features, docids, n_docs, (int)batch, L, sv, st, &erc);  // one launch (sr_embed_fwd_kernel)
  if (embedded && erc) return erc;
  if (embedded) {
  } else if (F % 4 == 0 && F <= 1024 && ((uintptr_t)features & 15) == 0)
    hipLaunchKernelGGL(sr_ln_gather_v4_kernel, dim3(rblk), dim3(SR_ROWS * 64), 0, st, features, docids, n_docs, (int)batch, L, T, F,
                       params + p.g_in, params + p.b_in, sv + p.sv_xg, sv + p.sv_xn0, sv + p.sv_mean_in, sv + p.sv_rstd_in);
  else
    hipLaunchKernelGGL(sr_ln_fwd_kernel, dim3(rblk), dim3(SR_ROWS * 64), 0, st, features, (const float*)nullptr, (const float*)nullptr, docids, n_docs,
                       (int)batch, L, T, F, params + p.g_in, params + p.b_in, sv + p.sv_xg, sv + p.sv_xn0, sv + p.sv_mean_in,
                       sv + p.sv_rstd_in);
  if (!embedded) {
    SR_CHECK(gemm_xwT(sv + p.sv_xn0, params + p.w1, params + p.b1, sv + p.sv_h0, T, F, dff, 1, st));
    SR_CHECK(gemm_xwT(sv + p.sv_h0, params + p.w2, params + p.b2, sv + p.sv_x[0], T, dff, d, 0, st));
  }
  bool head_done = false;
  for (int l = 0; l < p.nl; ++l) {
    const SrLayer& y = p.lay[l];
    const float* x = sv + p.sv_x[l];
    SR_CHECK(sr_attn_forward(ash, x, batch, L, sv + p.sv_A[l], sv + p.sv_lse[l], st));
    // A Wd^T lands in out1's buffer, then out1 = LN1(x + (A Wd^T + bd)) in place (s1 keeps the pre-norm sum)
    // the Linear's epilogue writes the pre-norm sum s1 = x + (A Wd^T + bd) straight into `saved` (one pass over [T, d] less on
    // each side of the LayerNorm); shapes the tiled GEMM does not take: Linear, then the residual pass
    {
      int brc = 0;
      if (block_fwd(p, l, params, sv, scores, st, &brc)) {  // everything behind the attention in one launch (sr_block_fwd_kernel)
        if (brc > 0 || brc < -1) return brc;
        head_done = brc == -1;
        continue;
      }
    }
    const bool ln_v4 = (d == 256 || d == 512 || d == 768 || d == 1024);
    if (ln_v4 && gemm_xwT_res(sv + p.sv_A[l], params + y.wd, params + y.bd, x, sv + p.sv_s1[l], T, d, d, st)) {
      ln_residual_fwd(sv + p.sv_s1[l], nullptr, nullptr, T, d, params + y.g1, params + y.b1, nullptr, sv + p.sv_out1[l],
                      sv + p.sv_m1[l], sv + p.sv_r1[l], (int)batch, L, st);
    } else {
      SR_CHECK(gemm_xwT(sv + p.sv_A[l], params + y.wd, nullptr, sv + p.sv_out1[l], T, d, d, 0, st));
      ln_residual_fwd(x, sv + p.sv_out1[l], params + y.bd, T, d, params + y.g1, params + y.b1, sv + p.sv_s1[l], sv + p.sv_out1[l],
                      sv + p.sv_m1[l], sv + p.sv_r1[l], (int)batch, L, st);
    }
    SR_CHECK(gemm_xwT(sv + p.sv_out1[l], params + y.wf1, params + y.bf1, sv + p.sv_f[l], T, d, dff, 1, st));
    if (ln_v4 && gemm_xwT_res(sv + p.sv_f[l], params + y.wf2, params + y.bf2, sv + p.sv_out1[l], sv + p.sv_s2[l], T, dff, d, st)) {
      ln_residual_fwd(sv + p.sv_s2[l], nullptr, nullptr, T, d, params + y.g2, params + y.b2, nullptr, sv + p.sv_x[l + 1],
                      sv + p.sv_m2[l], sv + p.sv_r2[l], (int)batch, L, st);
    } else {
      SR_CHECK(gemm_xwT(sv + p.sv_f[l], params + y.wf2, nullptr, sv + p.sv_x[l + 1], T, dff, d, 0, st));
      ln_residual_fwd(sv + p.sv_out1[l], sv + p.sv_x[l + 1], params + y.bf2, T, d, params + y.g2, params + y.b2, sv + p.sv_s2[l],
                      sv + p.sv_x[l + 1], sv + p.sv_m2[l], sv + p.sv_r2[l], (int)batch, L, st);
    }
  }
  // output FFN (SetRank.py:136, 153)
  if (!head_done) {
    SR_CHECK(gemm_xwT(sv + p.sv_x[p.nl], params + p.wo1, params + p.bo1, sv + p.sv_oh, T, d, dff, 1, st));
    SR_CHECK(gemm_xwT(sv + p.sv_oh, params + p.wo2, params + p.bo2, scores, T, dff, 1, 0, st));
  }
  return (int)hipGetLastError();
}

extern "C" int ultr_setrank_backward(const ultr_setrank_desc* c, const float* params, int32_t batch, int32_t list_size,
                                     const void* saved, const float* dscores, const void* loss_ws, int32_t n_loss_parts, void* ws_,
                                     float* grads, void* stream) {
  if (!params || !saved || !dscores || !ws_ || !grads || batch <= 0 || list_size <= 0) return ULTR_E_BADARG;
  const int64_t T = (int64_t)batch * list_size;
  SrPlan p;
  if (!make_plan(c, T, &p)) return ULTR_E_BADARG;
  const int L = list_size;
  const SrAttnShape ash = {p.d, p.dh, p.H, p.att_f16};
  SR_CHECK(sr_attn_supported(ash, L, 1));
  sr_knobs_load();
  // ULTR_SR_ATTN_H3 and its mask (bit 2 layer + 1: this layer's backward) allow the split-half attention backward kernel
  auto attn_split_half = [&](int layer) { return g_sr_knob_attn_h3 >= 1 && ((g_sr_knob_attn_mask >> (2 * layer + 1)) & 1) != 0; };
  hipStream_t st = (hipStream_t)stream;
  const float* sv = (const float*)saved;
  float* ws = (float*)ws_;
  float* G0 = ws + p.ws_g[0];
  float* G1 = ws + p.ws_g[1];
  float* G2 = ws + p.ws_g[2];
  const unsigned rblk = (unsigned)((T + SR_ROWS - 1) / SR_ROWS);
  const int F = p.F, d = p.d, dff = p.dff;
  struct H3Scope {
    ~H3Scope() { g_sr_h3 = {nullptr, nullptr, nullptr}; }
  } h3scope;
  if (sr_h3_enabled() && !p.no_h3) g_sr_h3 = {params, reinterpret_cast<const _Float16*>(sv + p.sv_planes), &p};  // built by this step's forward
  // ---- output FFN:  s = oh wo2^T + bo2,  oh = relu(x_nl Wo1^T + bo1) ----------------------------------------------
  FoldScope folds(ws + p.ws_arena, p.arena_floats);  // every fold below is queued; ONE launch at the end
  // the fused launches of ultr_sr_bwd.hip: config 5's widths, split-half products on, the transposed fragment copies built by this
  // step's forward
  int fz_R = 0, fz_tiles = 0, fz_nwg = 0;
  bool fused = p.bwd_fused && g_sr_knob_bwd_fused != 0 && g_sr_h3.planes != nullptr && T * (int64_t)d * 4 < ((int64_t)1 << 31);
  if (fused) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    fused = sr_bwd_geometry(T, cus, &fz_R, &fz_tiles, &fz_nwg);
  }
  bool head_done = false;
  if (fused && (g_sr_knob_bwd_fused & 2) && p.bo1 == p.wo1 + (int64_t)dff * d && p.wo2 == p.bo1 + dff && p.bo2 == p.wo2 + dff) {
    const SrPlan::SplitMat* mo = sr_find_split(params + p.wo1, dff, d);
    const int64_t sh = ((int64_t)dff * d + 2 * dff + 2 + 3) & ~(int64_t)3;  // (whole float4s: sr_fold_all_kernel's vector form)
    float* ph = nullptr;
    if (mo && mo->gt_off >= 0 && (ph = arena_piece(fz_nwg * sh)) != nullptr) {
      SrBwdHeadArgs ha;
      memset(&ha, 0, sizeof(ha));
      ha.R = fz_R; ha.d = d; ha.dff = dff; ha.ntiles = fz_tiles; ha.T = T;
      ha.dx = p.ws_g[0]; ha.x = p.sv_x[p.nl]; ha.oh = p.sv_oh; ha.wo2 = p.wo2; ha.gto1 = mo->gt_off;
      ha.part = ph - ws; ha.part_stride = sh;
      SR_CHECK(sr_bwd_head_launch(ha, fz_nwg, params, g_sr_h3.planes, sv, dscores, ws, st));  // G0 = d x_nl
      fold(ph, sh, fz_nwg, dff * d + 2 * dff + 1, grads + p.wo1, st);                         // d Wo1 | d bo1 | d wo2 | d bo2
      head_done = true;
    }
  }
  if (head_done) {
  } else
  if (dff <= 256 && p.bo2 == p.wo2 + dff) {  // one pass: G1 = d oh [T, dff] (ReLU mask fused), d wo2 | d bo2 partials
    float* hpart = part_scratch(ws + p.ws_part, (int64_t)p.n_cs * (dff + 1));
    hipLaunchKernelGGL(sr_head_bwd_kernel, dim3(p.n_cs), dim3(256), 0, st, dscores, (const float*)(sv + p.sv_oh), params + p.wo2, T, (int)dff,
                       G1, hpart);
    fold(hpart, (int64_t)dff + 1, p.n_cs, (int)dff + 1, grads + p.wo2, st);
  } else {
    SR_CHECK(gemm_dyTx(p, dscores, sv + p.sv_oh, grads + p.wo2, T, dff, 1, ws, st));
    colsum(p, dscores, nullptr, nullptr, nullptr, 1, 0, ws, grads + p.bo2, st);
    SR_CHECK(gemm_dyw(dscores, params + p.wo2, G1, sv + p.sv_oh, T, dff, 1, 0, st));  // G1 = d oh  [T, dff], ReLU mask fused
  }
  if (!head_done) {
    SR_CHECK(wgrad(p, G1, sv + p.sv_x[p.nl], grads + p.wo1, grads + p.bo1, T, d, dff, ws, st));
    SR_CHECK(gemm_dyw(G1, params + p.wo1, G0, nullptr, T, d, dff, 0, st));     // G0 = d x_nl  [T, d]
  }
  for (int l = p.nl - 1; l >= 0; --l) {
    const SrLayer& y = p.lay[l];
    if (fused && (g_sr_knob_bwd_fused & 1)) {
      const SrPlan::SplitMat* md = sr_find_split(params + y.wd, d, d);
      const SrPlan::SplitMat* m1 = sr_find_split(params + y.wf1, dff, d);
      const SrPlan::SplitMat* m2 = sr_find_split(params + y.wf2, d, dff);
      const int64_t sa = (int64_t)d * dff + 3 * d, sb = (int64_t)dff * d + dff + 3 * d;
      float* pa = nullptr;
      float* pb = nullptr;
      if (md && m1 && m2 && md->gt_off >= 0 && m1->gt_off >= 0 && m2->gt_off >= 0 && y.bf2 == y.wf2 + (int64_t)d * dff && y.b2 == y.g2 + d &&
          y.bf1 == y.wf1 + (int64_t)dff * d && y.b1 == y.g1 + d && (pa = arena_piece(fz_nwg * sa)) != nullptr &&
          (pb = arena_piece(fz_nwg * sb)) != nullptr) {
        SrBwdFfnArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.R = fz_R; fa.d = d; fa.dff = dff; fa.ntiles = fz_tiles; fa.T = T;
        fa.dy = p.ws_g[0]; fa.dF = p.ws_g[1]; fa.dx = p.ws_g[2];
        fa.s = p.sv_s2[l]; fa.mean = p.sv_m2[l]; fa.rstd = p.sv_r2[l]; fa.f = p.sv_f[l];
        fa.gamma = y.g2; fa.gt2 = m2->gt_off; fa.gt1 = m1->gt_off;
        fa.part = pa - ws; fa.part_stride = sa;
        SR_CHECK(sr_bwd_ffn_launch(fa, fz_nwg, params, g_sr_h3.planes, sv, ws, st));  // G1 = d f, G2 = d out1
        fold(pa, sa, fz_nwg, d * dff + d, grads + y.wf2, st);            // d Wf2 | d bf2
        fold(pa + (int64_t)d * dff + d, sa, fz_nwg, 2 * d, grads + y.g2, st);  // d g2 | d b2
        SrBwdProjArgs pr;
        memset(&pr, 0, sizeof(pr));
        pr.R = fz_R; pr.d = d; pr.dff = dff; pr.ntiles = fz_tiles; pr.T = T;
        pr.dy = p.ws_g[2]; pr.dF = p.ws_g[1]; pr.ds = p.ws_g[0]; pr.dx = p.ws_g[2];
        pr.s = p.sv_s1[l]; pr.mean = p.sv_m1[l]; pr.rstd = p.sv_r1[l];
        pr.gamma = y.g1; pr.beta = y.b1; pr.gtd = md->gt_off;
        pr.part = pb - ws; pr.part_stride = sb;
        SR_CHECK(sr_bwd_proj_launch(pr, fz_nwg, params, g_sr_h3.planes, sv, ws, st));  // G0 = d s1 (= d x_l through the residual), G2 = d A
        fold(pb, sb, fz_nwg, dff * d + dff, grads + y.wf1, st);                           // d Wf1 | d bf1
        fold(pb + (int64_t)dff * d + dff, sb, fz_nwg, d, grads + y.bd, st);               // d bd
        fold(pb + (int64_t)dff * d + dff + d, sb, fz_nwg, 2 * d, grads + y.g1, st);       // d g1 | d b1
        SR_CHECK(wgrad(p, G0, sv + p.sv_A[l], grads + y.wd, nullptr, T, d, d, ws, st));
        // G0 += attention path -> d x_l
        SR_CHECK(sr_attn_backward(ash, attn_split_half(l), sv + p.sv_x[l], G2, sv + p.sv_A[l], sv + p.sv_lse[l], batch, L, G0, st));
        continue;
      }
    }
    // (the forward did not write out1 when it expected the fused launches: sr_block_fwd_kernel's skip_out1 - same predicate)
    if (sr_bwd_blocks_fused(p) && g_sr_knob_block != 0) return ULTR_E_UNSUPPORTED;
    // x_{l+1} = LN2(s2),  s2 = out1 + ffn
    if (d <= 1024) {  // g2 | b2, G2 = d s2 = d out1 (residual) = d ffn, bf2: one pass
      SR_CHECK(ln_bwd_cs(p, G0, sv + p.sv_s2[l], sv + p.sv_m2[l], sv + p.sv_r2[l], params + y.g2, d, G2, ws, grads + y.g2,
                         grads + y.bf2, st));
    } else {
      colsum_ln(p, G0, sv + p.sv_s2[l], sv + p.sv_m2[l], sv + p.sv_r2[l], d, ws, grads + y.g2, st);
      hipLaunchKernelGGL(sr_ln_bwd_kernel, dim3(rblk), dim3(SR_ROWS * 64), 0, st, (const float*)G0, sv + p.sv_s2[l], sv + p.sv_m2[l],
                         sv + p.sv_r2[l], params + y.g2, T, d, G2);
      colsum(p, G2, nullptr, nullptr, nullptr, d, 0, ws, grads + y.bf2, st);
    }
    SR_CHECK(wgrad(p, G2, sv + p.sv_f[l], grads + y.wf2, nullptr, T, dff, d, ws, st));
    SR_CHECK(gemm_dyw(G2, params + y.wf2, G1, sv + p.sv_f[l], T, dff, d, 0, st));  // G1 = d f  [T, dff], ReLU mask fused
    SR_CHECK(wgrad(p, G1, sv + p.sv_out1[l], grads + y.wf1, grads + y.bf1, T, d, dff, ws, st));
    SR_CHECK(gemm_dyw(G1, params + y.wf1, G2, nullptr, T, d, dff, 1, st));   // G2 = d out1 (both paths): accumulated
    // out1 = LN1(s1),  s1 = x_l + o
    if (d <= 1024) {  // g1 | b1, G0 = d s1 = d x_l (residual) = d o, bd
      SR_CHECK(ln_bwd_cs(p, G2, sv + p.sv_s1[l], sv + p.sv_m1[l], sv + p.sv_r1[l], params + y.g1, d, G0, ws, grads + y.g1,
                         grads + y.bd, st));
    } else {
      colsum_ln(p, G2, sv + p.sv_s1[l], sv + p.sv_m1[l], sv + p.sv_r1[l], d, ws, grads + y.g1, st);
      hipLaunchKernelGGL(sr_ln_bwd_kernel, dim3(rblk), dim3(SR_ROWS * 64), 0, st, (const float*)G2, sv + p.sv_s1[l], sv + p.sv_m1[l],
                         sv + p.sv_r1[l], params + y.g1, T, d, G0);
      colsum(p, G0, nullptr, nullptr, nullptr, d, 0, ws, grads + y.bd, st);
    }
    SR_CHECK(wgrad(p, G0, sv + p.sv_A[l], grads + y.wd, nullptr, T, d, d, ws, st));
    SR_CHECK(gemm_dyw(G0, params + y.wd, G1, nullptr, T, d, d, 0, st));      // G1 = d A  [T, d]
    // G0 += attention path -> d x_l
    SR_CHECK(sr_attn_backward(ash, attn_split_half(l), sv + p.sv_x[l], G1, sv + p.sv_A[l], sv + p.sv_lse[l], batch, L, G0, st));
  }
  // ---- embedding FFN and the input LayerNorm's parameters -------------------------------------------------------------
  bool embed_done = false;
  if (fused && (g_sr_knob_bwd_fused & 4) && F % 4 == 0 && F <= SR_BWD_D && p.b_in == p.g_in + F && p.w1 == p.b_in + F && p.b1 == p.w1 + (int64_t)dff * F &&
      p.w2 == p.b1 + dff && p.b2 == p.w2 + (int64_t)d * dff) {
    const SrPlan::SplitMat* m1 = sr_find_split(params + p.w1, dff, F);
    const SrPlan::SplitMat* m2 = sr_find_split(params + p.w2, d, dff);
    const int64_t se = (int64_t)2 * F + (int64_t)dff * F + dff + (int64_t)d * dff + d;
    float* pe = nullptr;
    if (m1 && m2 && m1->gt_off >= 0 && m2->gt_off >= 0 && (pe = arena_piece(fz_nwg * se)) != nullptr) {
      SrBwdEmbedArgs ea;
      memset(&ea, 0, sizeof(ea));
      ea.R = fz_R; ea.d = d; ea.dff = dff; ea.F = F; ea.ntiles = fz_tiles; ea.T = T;
      ea.dy = p.ws_g[0]; ea.h0 = p.sv_h0; ea.xg = p.sv_xg; ea.mean = p.sv_mean_in; ea.rstd = p.sv_rstd_in;
      ea.g_in = p.g_in; ea.b_in = p.b_in; ea.gt2 = m2->gt_off; ea.gt1 = m1->gt_off;
      ea.part = pe - ws; ea.part_stride = se;
      SR_CHECK(sr_bwd_embed_launch(ea, fz_nwg, params, g_sr_h3.planes, sv, ws, st));
      fold(pe, se, fz_nwg, (int)se, grads + p.g_in, st);  // d g_in | d b_in | d W1 | d b1 | d W2 | d b2
      embed_done = true;
    }
  }
  if (!embed_done) {
  SR_CHECK(wgrad(p, G0, sv + p.sv_h0, grads + p.w2, grads + p.b2, T, dff, d, ws, st));
  SR_CHECK(gemm_dyw(G0, params + p.w2, G1, sv + p.sv_h0, T, dff, d, 0, st));  // ReLU mask fused
  SR_CHECK(wgrad(p, G1, sv + p.sv_xn0, grads + p.w1, grads + p.b1, T, F, dff, ws, st));
  SR_CHECK(gemm_dyw(G1, params + p.w1, G2, nullptr, T, F, dff, 0, st));      // G2 = d xn0  [T, F]
  colsum_ln(p, G2, sv + p.sv_xg, sv + p.sv_mean_in, sv + p.sv_rstd_in, F, ws, grads + p.g_in, st);  // g_in | b_in
  }
  // ---- step tail: fold the loss partials behind the gradient ----------------------------------------------------------
  if (loss_ws != nullptr && n_loss_parts > 0) {
    const int tail = (int)ultr_tail_len(list_size);
    fold((const float*)loss_ws, (int64_t)tail, (int)n_loss_parts, tail, grads + p.P, st, /*own_buffer=*/true);
  }
  folds.flush(st);
  return (int)hipGetLastError();
}
