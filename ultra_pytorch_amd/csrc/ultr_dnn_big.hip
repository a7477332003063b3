// ultr_dnn_big.hip — the DNN ranking model's training forward / backward for LARGE batches (tens of thousands of rows).
//
// The kernels of ultr_dnn.hip keep a 16-row tile on chip through every layer and re-stream every weight matrix per tile:
// right where latency decides (BASELINE config 2: 2 560 rows).  For big batches every layer can be ONE pass over the rows
// instead.  Which half is taken is a measured rule of the launcher (ultr_dnn.hip, big_fwd_wanted / big_bwd_wanted):
// the BACKWARD from 16 384 rows, and from 4 096 rows when a layer is wider than 512 (config 4: 12 800 rows x [512, 256, 128]
// behind 700 features, 128 -> 85 us); the FORWARD only for rows too wide for the row tile's LDS (it measures 10-40 % slower
// than the row-tile forward where both run).
//   forward   per layer  row statistics (mean, rstd; HBM-bound, a wave per row)  ->  tiled GEMM (ultr_gemm.h) whose
//             A-operand producer applies the LayerNorm on the fly (and gathers the feature rows by document id for
//             layer 0) and whose epilogue adds the bias, applies the activation and writes x_{j+1} for the backward;
//             the last statistics pass also folds the width-1 scorer;
//   backward  a row kernel for the top layer (du = ds * w on the fly, LayerNorm backward, activation', dz_{top-1} + the
//             column partials of gamma / beta / scorer), then per layer  tiled GEMM du_j = dz_j . W_j  ->  row kernel
//             (LayerNorm backward through x_j -> dz_{j-1}, column partials); layer 0's dgrad does not exist (the
//             weight-gradient launch contracts with xhat_0, BwdPlan::l0g).  The weight gradients, the slab reduction and
//             the update are the launches of ultr_dnn.hip / ultr_update.hip, unchanged.
// Activations make one HBM round trip per layer (>= 100 flops per byte at these shapes), nothing re-streams a weight
// matrix per 16 rows, and every launch has thousands of waves.  Same arithmetic (two-pass statistics, rsqrt + Newton,
// expm1-accurate ELU) and the same outputs as the fused kernels: `saved` (x_j, mean, rstd), scores, dz_j, vector slabs.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <string.h>

#include "../../include/ultr_hip.h"
#include "ultr_device.h"
#include "ultr_gemm.h"
#include "ultr_plan.h"

namespace {

constexpr int BIG_ROWS = ULTR_BIG_ROWS;  // rows per workgroup of the row kernels = rows per vector slab
constexpr int BIG_XC_MAX = 4;            // float4 chunks per lane: rows up to 1024 wide

// ---- row statistics (+ scorer for the top layer) ----------------------------------------------------------------------
// x rows: consecutive (ids == nullptr) or gathered through the feed's position-major ids (PAD / rows past N: zeros)
template <int XC>
__global__ __launch_bounds__(256) void big_rowstats_kernel(const float* __restrict__ x, int64_t x_rows, const int32_t* __restrict__ ids,
                                                           int64_t n_docs, int B, int L, int64_t N, int K, float* __restrict__ mean_out,
                                                           float* __restrict__ rstd_out, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ wlast,
                                                           const float* __restrict__ blast, float* __restrict__ scores,
                                                           float* __restrict__ xhat_out, float* __restrict__ marker) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // layer 0's pass tells the weight-gradient launch whether saved.x_0 holds xhat_0 (DnnPlan::sv_total: the marker word)
  if (marker != nullptr && blockIdx.x == 0 && threadIdx.x == 0) marker[0] = xhat_out != nullptr ? 1.0f : 0.0f;
  const Src xs = make_src(x, x_rows * K);
  const float invK = 1.0f / (float)K;
  const bool top = scores != nullptr;
  float4 gw[XC], bw[XC];
  float bsum = 0.f;
  if (top) {
#pragma unroll
    for (int u = 0; u < XC; ++u) {
      const int c = 4 * lane + 256 * u;
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 g = c < K ? ld4(gamma + c) : z4, b = c < K ? ld4(beta + c) : z4, w = c < K ? ld4(wlast + c) : z4;
      gw[u] = make_float4(g.x * w.x, g.y * w.y, g.z * w.z, g.w * w.w);
      bw[u] = make_float4(b.x * w.x, b.y * w.y, b.z * w.z, b.w * w.w);
      bsum += (bw[u].x + bw[u].y) + (bw[u].z + bw[u].w);
    }
    bsum = wave_sum(bsum);
  }
  for (int64_t r = (int64_t)blockIdx.x * BIG_ROWS + wave; r < (int64_t)(blockIdx.x + 1) * BIG_ROWS && r < N; r += 4) {
    int64_t src = r;
    bool live = true;
    if (ids != nullptr) {
      const uint32_t rr = (uint32_t)r;
      const int32_t id = ids[(int64_t)(rr % (uint32_t)L) * B + rr / (uint32_t)L];
      live = id >= 0 && id < n_docs;
      src = id;
    }
    float4 v[XC];
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < XC; ++u) {
      const int c = 4 * lane + 256 * u;
      v[u] = buf_ld4(xs, (live && c < K) ? (unsigned)((src * K + c) * 4) : ULTR_OOB);
      s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
    }
    const float mean = wave_sum(s) * invK;
    float q = 0.f, t = 0.f;
#pragma unroll
    for (int u = 0; u < XC; ++u) {
      const int c = 4 * lane + 256 * u;
      if (c < K) {
        v[u].x -= mean; v[u].y -= mean; v[u].z -= mean; v[u].w -= mean;
      }
      q += (v[u].x * v[u].x + v[u].y * v[u].y) + (v[u].z * v[u].z + v[u].w * v[u].w);
      if (top) t += (v[u].x * gw[u].x + v[u].y * gw[u].y) + (v[u].z * gw[u].z + v[u].w * gw[u].w);
    }
    const float rstd = rsqrt_nr(wave_sum(q) * invK + ULTR_LN_EPS);
    if (xhat_out != nullptr) {  // the normalised row, contiguous: the layer's GEMM (and nothing else) reads it back
#pragma unroll
      for (int u = 0; u < XC; ++u) {
        const int c = 4 * lane + 256 * u;
        if (c < K) st4_out(xhat_out + r * K + c, make_float4(v[u].x * rstd, v[u].y * rstd, v[u].z * rstd, v[u].w * rstd));
      }
    }
    if (top) t = wave_sum(t);
    if (lane == 0) {
      mean_out[r] = mean;
      rstd_out[r] = rstd;
      if (top) scores[r] = rstd * t + bsum + blast[0];
    }
  }
}

// ---- LayerNorm backward through x_j (+ activation' of the layer below) and the column partials -----------------------
//   du      [N, K] gradient w.r.t. the LayerNorm OUTPUT u_j   (TOP: du = ds[r] * wlast, formed on the fly)
//   dz_out  [N, K] = LN'(du) * act'(x)          x is the previous layer's post-activation output (act' from the output)
//   vslab[blk]:  dgamma_j | dbeta_j (| TOP: d wlast, d blast) partial sums over the block's rows
template <int XC, bool TOP>
__global__ __launch_bounds__(256) void big_lnbwd_kernel(const float* __restrict__ du, const float* __restrict__ ds,
                                                        const float* __restrict__ x, const float* __restrict__ mean_in,
                                                        const float* __restrict__ rstd_in, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ wlast, int64_t N, int K,
                                                        int act, float* __restrict__ dz_out, float* __restrict__ vslab, int vlen,
                                                        int off_g, int off_b, int off_wk, int off_bk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [4][3 K4] cross-wave fold
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invK = 1.0f / (float)K;
  const int K4 = round_up(K, 4);
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 g4[XC], b4[XC], w4[XC], pg[XC], pb[XC], pw[XC];
#pragma unroll
  for (int u = 0; u < XC; ++u) {
    const int c = 4 * lane + 256 * u;
    g4[u] = c < K ? ld4(gamma + c) : z4;
    b4[u] = (TOP && c < K) ? ld4(beta + c) : z4;
    w4[u] = (TOP && c < K) ? ld4(wlast + c) : z4;
    pg[u] = pb[u] = pw[u] = z4;
  }
  float pds = 0.f;
  const int64_t rbeg = (int64_t)blockIdx.x * BIG_ROWS;
  for (int64_t r = rbeg + wave; r < rbeg + BIG_ROWS && r < N; r += 4) {
    const float mean = mean_in[r], rstd = rstd_in[r];
    const float dsr = TOP ? ds[r] : 0.f;
    float4 xv[XC], gx[XC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < XC; ++u) {
      const int c = 4 * lane + 256 * u;
      const bool in = c < K;
      xv[u] = in ? ld4(x + r * K + c) : z4;
      float4 d4;
      if (TOP) d4 = make_float4(dsr * w4[u].x, dsr * w4[u].y, dsr * w4[u].z, dsr * w4[u].w);
      else d4 = in ? ld4(du + r * K + c) : z4;
      const float4 xh = make_float4((xv[u].x - mean) * rstd, (xv[u].y - mean) * rstd, (xv[u].z - mean) * rstd, (xv[u].w - mean) * rstd);
      gx[u] = make_float4(d4.x * g4[u].x, d4.y * g4[u].y, d4.z * g4[u].z, d4.w * g4[u].w);
      s1 += (gx[u].x + gx[u].y) + (gx[u].z + gx[u].w);
      s2 += (gx[u].x * xh.x + gx[u].y * xh.y) + (gx[u].z * xh.z + gx[u].w * xh.w);
      if (in) {
        pg[u].x += d4.x * xh.x; pg[u].y += d4.y * xh.y; pg[u].z += d4.z * xh.z; pg[u].w += d4.w * xh.w;
        pb[u].x += d4.x; pb[u].y += d4.y; pb[u].z += d4.z; pb[u].w += d4.w;
        if (TOP) {
          pw[u].x += dsr * (g4[u].x * xh.x + b4[u].x); pw[u].y += dsr * (g4[u].y * xh.y + b4[u].y);
          pw[u].z += dsr * (g4[u].z * xh.z + b4[u].z); pw[u].w += dsr * (g4[u].w * xh.w + b4[u].w);
        }
      }
    }
    if (TOP) pds += dsr;
    float red[2] = {s1, s2};
    wave_sum_n<2>(red);
    s1 = red[0] * invK;
    s2 = red[1] * invK;
#pragma unroll
    for (int u = 0; u < XC; ++u) {
      const int c = 4 * lane + 256 * u;
      if (c < K) {
        float4 dz;
        dz.x = rstd * (gx[u].x - s1 - (xv[u].x - mean) * rstd * s2) * act_grad_from_out(xv[u].x, act);
        dz.y = rstd * (gx[u].y - s1 - (xv[u].y - mean) * rstd * s2) * act_grad_from_out(xv[u].y, act);
        dz.z = rstd * (gx[u].z - s1 - (xv[u].z - mean) * rstd * s2) * act_grad_from_out(xv[u].z, act);
        dz.w = rstd * (gx[u].w - s1 - (xv[u].w - mean) * rstd * s2) * act_grad_from_out(xv[u].w, act);
        st4_out(dz_out + r * K + c, dz);
      }
    }
  }
  // fold the four waves' column partials in fixed order, one slab per block
  float* mine = smem + wave * 3 * K4;
#pragma unroll
  for (int u = 0; u < XC; ++u) {
    const int c = 4 * lane + 256 * u;
    if (c < K) {
      st4(mine + c, pg[u]);
      st4(mine + K4 + c, pb[u]);
      if (TOP) st4(mine + 2 * K4 + c, pw[u]);
    }
  }
  float pdsw = 0.f;
  if (TOP) pdsw = pds;  // every lane of a wave carries the same running sum
  __shared__ float sm_ds[4];
  if (TOP && lane == 0) sm_ds[wave] = pdsw;
  __syncthreads();
  float* slab = vslab + (int64_t)blockIdx.x * vlen;
  for (int c = threadIdx.x; c < K; c += 256) {
    slab[off_g + c] = ((smem[c] + smem[3 * K4 + c]) + smem[6 * K4 + c]) + smem[9 * K4 + c];
    slab[off_b + c] = ((smem[K4 + c] + smem[4 * K4 + c]) + smem[7 * K4 + c]) + smem[10 * K4 + c];
    if (TOP) slab[off_wk + c] = ((smem[2 * K4 + c] + smem[5 * K4 + c]) + smem[8 * K4 + c]) + smem[11 * K4 + c];
  }
  if (TOP && threadIdx.x == 0) slab[off_bk] = ((sm_ds[0] + sm_ds[1]) + sm_ds[2]) + sm_ds[3];
}

template <typename Kern>
hipError_t big_set_lds(Kern k, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// launch under optional dispatch-packet timestamps (ultr_prof.h: the group first launch .. last launch is one sample)
#define BIG_LAUNCH(KERNEL, GRID, LDS, ST, EA, EB, ...)                                                                  \
  do {                                                                                                                  \
    if ((EA) != nullptr || (EB) != nullptr) hipExtLaunchKernelGGL(KERNEL, GRID, dim3(256), (uint32_t)(LDS), ST, EA, EB, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL, GRID, dim3(256), LDS, ST, __VA_ARGS__);                                             \
  } while (0)

struct StatsArgs {
  const float* x; int64_t x_rows; const int32_t* ids; int64_t n_docs; int B, L; int64_t N; int K;
  float *mean, *rstd; const float *gamma, *beta, *wlast, *blast; float* scores; float* xhat_out; float* marker;
};
template <int XC>
void stats_launch(const StatsArgs& a, unsigned blocks, hipStream_t st, hipEvent_t ea, hipEvent_t eb) {
  BIG_LAUNCH(big_rowstats_kernel<XC>, dim3(blocks), 0, st, ea, eb, a.x, a.x_rows, a.ids, a.n_docs, a.B, a.L, a.N, a.K, a.mean, a.rstd,
             a.gamma, a.beta, a.wlast, a.blast, a.scores, a.xhat_out, a.marker);
}
void stats_dispatch(const StatsArgs& a, unsigned blocks, hipStream_t st, hipEvent_t ea, hipEvent_t eb) {
  const int xc = (a.K + 255) / 256;
  if (xc <= 1) stats_launch<1>(a, blocks, st, ea, eb);
  else if (xc == 2) stats_launch<2>(a, blocks, st, ea, eb);
  else if (xc == 3) stats_launch<3>(a, blocks, st, ea, eb);
  else stats_launch<4>(a, blocks, st, ea, eb);
}

struct LnbArgs {
  const float *du, *ds, *x, *mean, *rstd, *gamma, *beta, *wlast; int64_t N; int K, act; float *dz_out, *vslab;
  int vlen, off_g, off_b, off_wk, off_bk;
};
template <int XC, bool TOP>
hipError_t lnb_launch(const LnbArgs& a, unsigned blocks, hipStream_t st, hipEvent_t ea, hipEvent_t eb) {
  const size_t lds = (size_t)4 * 3 * round_up(a.K, 4) * sizeof(float);
  const hipError_t e = big_set_lds(big_lnbwd_kernel<XC, TOP>, lds);
  if (e != hipSuccess) return e;
  BIG_LAUNCH((big_lnbwd_kernel<XC, TOP>), dim3(blocks), lds, st, ea, eb, a.du, a.ds, a.x, a.mean, a.rstd, a.gamma, a.beta, a.wlast, a.N, a.K,
             a.act, a.dz_out, a.vslab, a.vlen, a.off_g, a.off_b, a.off_wk, a.off_bk);
  return hipSuccess;
}
template <bool TOP>
hipError_t lnb_dispatch(const LnbArgs& a, unsigned blocks, hipStream_t st, hipEvent_t ea, hipEvent_t eb) {
  const int xc = (a.K + 255) / 256;
  if (xc <= 1) return lnb_launch<1, TOP>(a, blocks, st, ea, eb);
  if (xc == 2) return lnb_launch<2, TOP>(a, blocks, st, ea, eb);
  if (xc == 3) return lnb_launch<3, TOP>(a, blocks, st, ea, eb);
  return lnb_launch<4, TOP>(a, blocks, st, ea, eb);
}

// hi | lo fp16 planes of 2^8 W_j^T ([K_j][ldM_j], zero-padded) for the split-half dgrad GEMMs: blockIdx.y = layer j
struct WtPlaneTable {
  int64_t off_w[ULTR_MAXL], plane[ULTR_MAXL];
  int M[ULTR_MAXL], K[ULTR_MAXL], ldM[ULTR_MAXL];
};
__global__ __launch_bounds__(256) void big_split_wT_kernel(WtPlaneTable tb, const float* __restrict__ params, _Float16* __restrict__ planes) {
  const int j = blockIdx.y + 1;
  const int K = tb.K[j], M = tb.M[j], ld = tb.ldM[j];
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)K * ld) return;
  const int k = (int)(e / ld), m = (int)(e - (int64_t)k * ld);
  const float w = m < M ? params[tb.off_w[j] + (int64_t)m * K + k] * UGEMM_H3_WSCALE : 0.f;
  const _Float16 hi = (_Float16)w, lo = (_Float16)(w - (float)hi);  // (|w| >= 128 overflows to inf: NaN gradients, loud)
  _Float16* dst = planes + tb.plane[j];
  dst[e] = hi;
  dst[(int64_t)K * ld + e] = lo;
}

// ... and of 2^8 W_j ([M_j][ldK_j]) for the split-half forward GEMMs: blockIdx.y = layer j
__global__ __launch_bounds__(256) void big_split_w_kernel(WtPlaneTable tb, const float* __restrict__ params, _Float16* __restrict__ planes) {
  const int j = blockIdx.y;
  const int K = tb.K[j], M = tb.M[j], ld = tb.ldM[j];  // (ldM holds ldK here)
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)M * ld) return;
  const int m = (int)(e / ld), k = (int)(e - (int64_t)m * ld);
  const float w = k < K ? params[tb.off_w[j] + (int64_t)m * K + k] * UGEMM_H3_WSCALE : 0.f;
  const _Float16 hi = (_Float16)w, lo = (_Float16)(w - (float)hi);
  _Float16* dst = planes + tb.plane[j];
  dst[e] = hi;
  dst[(int64_t)M * ld + e] = lo;
}

}  // namespace

// every layer must take the vector paths (rows a multiple of 4 floats wide, at most 1024; 32-bit buffer offsets)
bool ultr_dnn_big_ok(const DnnPlan& p, int64_t N, int64_t n_docs) {
  if (p.nl < 2) return false;
  const int64_t lim = (int64_t)1 << 31;
  for (int j = 0; j < p.nl; ++j) {
    if (p.K[j] % 4 != 0 || p.K[j] > 256 * BIG_XC_MAX || p.off_w[j] % 4 != 0 || p.off_lnw[j] % 4 != 0 || p.off_lnb[j] % 4 != 0) return false;
    if (j < p.nl - 1 && (p.M[j] % 4 != 0 || p.wt_off[j] % 4 != 0)) return false;
    if (N * p.K[j] * 4 >= lim) return false;
  }
  return n_docs * p.K[0] * 4 < lim && p.P * 4 < lim;
}

int ultr_dnn_big_forward(const DnnPlan& p, const float* params, const float* wt, const float* features, int64_t n_docs,
                         const int32_t* docids, int B, int L, float* scores, float* saved, hipStream_t st, hipEvent_t ev_start,
                         hipEvent_t ev_stop, bool split_half) {
  const int64_t N = (int64_t)B * L;
  const unsigned rblocks = (unsigned)((N + BIG_ROWS - 1) / BIG_ROWS);
  const int top = p.nl - 1;
  // split-half GEMMs (round 4, ultr_gemm.h run_h3): the planes of every W_j are rebuilt here, behind the saved activations
  _Float16* planes = reinterpret_cast<_Float16*>(saved + ultr_fwp_off(p));
  if (split_half) {
    WtPlaneTable tb;
    memset(&tb, 0, sizeof(tb));
    int64_t maxe = 0;
    for (int j = 0; j < top; ++j) {
      tb.off_w[j] = p.off_w[j]; tb.plane[j] = ultr_fwp_layer(p, j); tb.M[j] = p.M[j]; tb.K[j] = p.K[j]; tb.ldM[j] = (p.K[j] + 31) / 32 * 32;
      const int64_t e = (int64_t)p.M[j] * tb.ldM[j];
      maxe = e > maxe ? e : maxe;
    }
    hipLaunchKernelGGL(big_split_w_kernel, dim3((unsigned)((maxe + 255) / 256), (unsigned)top), dim3(256), 0, st, tb, params, planes);
  }
  for (int j = 0; j <= top; ++j) {
    const int K = p.K[j];
    const float* x = (j == 0) ? features : saved + p.sv_x[j];
    const int64_t x_rows = (j == 0) ? n_docs : N;
    const int32_t* ids = (j == 0) ? docids : nullptr;
    float* mean = saved + p.sv_mean[j];
    float* rstd = saved + p.sv_rstd[j];
    const bool last = (j == top);
    // wide gathered input (config 4: 700 features): the statistics pass also writes the normalised rows xhat_0 contiguously
    // (saved.x_0 is free in this mode) and the GEMM applies gamma / beta to plain rows - the LayerNorm + gather producer
    // costs the first GEMM a third (145 vs 108 us there)
    const bool pre0 = (j == 0) && !last && K > 512;
    float* xhat0 = pre0 ? saved + p.sv_x[0] : nullptr;
    const StatsArgs sa{x, x_rows, ids, n_docs, B, L, N, K, mean, rstd, params + p.off_lnw[j], params + p.off_lnb[j],
                       last ? params + p.off_w[j] : nullptr, last ? params + p.off_b[j] : nullptr, last ? scores : nullptr, xhat0,
                       j == 0 ? saved + p.sv_total : nullptr};
    stats_dispatch(sa, rblocks, st, j == 0 ? ev_start : nullptr, last ? ev_stop : nullptr);
    if (last) break;
    const int M = p.M[j];
    if (pre0) {
      const ugemm::AAffine a{xhat0, params, N, p.P, p.off_lnw[j], p.off_lnb[j], K};
      const ugemm::Dims d0{N, M, K, M};
      const ugemm::EBiasAct e0{saved + p.sv_x[j + 1], params + p.off_b[j], M, p.act};
      const int ldK = (K + 31) / 32 * 32;
      const ugemm::Dims dh0{N, M, K, ldK};
      const _Float16* hi0 = planes + ultr_fwp_layer(p, j);
      const hipError_t rc0 = split_half ? ugemm::run_h3(dh0, a, hi0, hi0 + (int64_t)M * ldK, e0, st) : ugemm::run<false>(d0, a, wt + p.wt_off[j], e0, st);
      if (rc0 != hipSuccess) return (int)rc0;
      continue;
    }
    ugemm::ALayerNorm a;
    a.x = x;
    a.gb = params;
    a.x_rows = x_rows;
    a.gb_floats = p.P;
    a.mean = mean;
    a.rstd = rstd;
    a.ids = ids;
    a.R = N;
    a.n_docs = n_docs;
    a.g_off = p.off_lnw[j];
    a.b_off = p.off_lnb[j];
    a.K = K;
    a.B = B;
    a.L = L;
    const ugemm::Dims d{N, M, K, M};
    const ugemm::EBiasAct e{saved + p.sv_x[j + 1], params + p.off_b[j], M, p.act};
    const int ldK = (K + 31) / 32 * 32;
    const ugemm::Dims dh{N, M, K, ldK};
    const _Float16* hi = planes + ultr_fwp_layer(p, j);
    const hipError_t rc = split_half ? ugemm::run_h3(dh, a, hi, hi + (int64_t)M * ldK, e, st)
                                     : ugemm::run<false>(d, a, wt + p.wt_off[j], e, st);  // B = the k-major copy WT_j [K][M]
    if (rc != hipSuccess) return (int)rc;
  }
  return (int)hipGetLastError();
}

// the row-local half of the backward: dz_j for every hidden Linear + the vector slabs (one per BIG_ROWS rows); bp.nrb must be
// ceil(N / BIG_ROWS) and bp.l0g == 1 (the caller's weight-gradient launch makes up for the missing layer-0 dgrad)
int ultr_dnn_big_backward(const DnnPlan& p, const BwdPlan& bp, const float* params, const float* saved, const float* dscores, float* ws,
                          hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop, bool split_half) {
  const int64_t N = bp.N;
  const unsigned rblocks = (unsigned)((N + BIG_ROWS - 1) / BIG_ROWS);
  const int top = p.nl - 1;
  float* vslab = ws + bp.vslab_off;
  float* du = ws + bp.du_off;
  // split-half dgrad GEMMs (round 4, ultr_gemm.h run_h3): the planes of every W_j^T are rebuilt here (the weights changed with the
  // last update; ~1 MB, one small launch in front of the row kernel of the top layer, which does not read them)
  _Float16* planes = reinterpret_cast<_Float16*>(ws + bp.dgp_off);
  split_half = split_half && top >= 2;
  if (split_half) {
    WtPlaneTable tb;
    memset(&tb, 0, sizeof(tb));
    int64_t maxe = 0;
    for (int j = 1; j < top; ++j) {
      tb.off_w[j] = p.off_w[j]; tb.plane[j] = ultr_dgp_layer(p, j); tb.M[j] = p.M[j]; tb.K[j] = p.K[j]; tb.ldM[j] = (p.M[j] + 31) / 32 * 32;
      const int64_t e = (int64_t)p.K[j] * tb.ldM[j];
      maxe = e > maxe ? e : maxe;
    }
    hipLaunchKernelGGL(big_split_wT_kernel, dim3((unsigned)((maxe + 255) / 256), (unsigned)(top - 1)), dim3(256), 0, st, tb, params, planes);
  }
  for (int j = top; j >= 1; --j) {
    const int K = p.K[j];
    LnbArgs la{du, dscores, saved + p.sv_x[j], saved + p.sv_mean[j], saved + p.sv_rstd[j], params + p.off_lnw[j], params + p.off_lnb[j],
               params + p.off_w[top], N, K, p.act, ws + bp.dz_off[j - 1], vslab, bp.vlen, bp.voff_g[j], bp.voff_b[j], bp.voff_wk, bp.voff_bk};
    hipError_t e;
    if (j == top) {
      e = lnb_dispatch<true>(la, rblocks, st, ev_start, j == 1 ? ev_stop : nullptr);
    } else {
      // du_j = dz_j . W_j   (W_j [M_j][K_j] row-major: k-major for the contraction over M_j)
      const int M = p.M[j];
      const ugemm::Dims d{N, K, M, K};
      const ugemm::APlain a{ws + bp.dz_off[j], N, M, M};
      const ugemm::EStore es{du, nullptr, K, 0};
      if (split_half) {
        const int ldM = (M + 31) / 32 * 32;
        const ugemm::Dims dh{N, K, M, ldM};
        const _Float16* hi = planes + ultr_dgp_layer(p, j);
        e = ugemm::run_h3(dh, a, hi, hi + (int64_t)K * ldM, es, st);
      } else {
        e = ugemm::run<false>(d, a, params + p.off_w[j], es, st);
      }
      if (e != hipSuccess) return (int)e;
      e = lnb_dispatch<false>(la, rblocks, st, nullptr, j == 1 ? ev_stop : nullptr);
    }
    if (e != hipSuccess) return (int)e;
  }
  return (int)hipGetLastError();
}
