// ultr_dnn.hip — gfx950 kernels for the DNN ranking model (reference ultra/ranking_model/DNN.py)
//
//   dnn_fwd_kernel    gather + [LayerNorm -> Linear -> act] x k + LayerNorm -> Linear(.,1), one launch.
//                     A workgroup owns R = 16/32 document rows end to end: activations never leave LDS
//                     between layers; weights stream from L2 straight into MFMA B-fragments.
//   dnn_bwd_kernel    the row-local half of backward (dgrad chain + LayerNorm backward + act'), one launch,
//                     same ownership; emits dz_j to HBM for the weight gradients and per-row-block partial
//                     sums for every vector parameter (LayerNorm gamma/beta, the M=1 scorer).
//   dnn_wgrad_kernel  all weight/bias gradients of the hidden Linears in one launch: dW = dz^T u with the
//                     contraction over the N rows; 64x64 output blocks x row splits, LayerNorm re-applied to
//                     the B operand on the fly, deterministic partial slabs (no atomics).
//   grad_reduce_kernel  fixed-order slab reduction -> flat gradient (+ step tail + sum-of-squares partials).
//
// All matrix math is v_mfma_f32_16x16x4_f32 (exact fp32).  Wave = 64 lanes everywhere.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/ultr_hip.h"
#include "ultr_device.h"
#include "ultr_plan.h"

// ------------------------------------------------------------------------------------------------
// LDS leading dimensions
// ------------------------------------------------------------------------------------------------
// forward buffers: float4 A-fragment reads -> ld % 4 == 0 and (ld/4) odd
__host__ __device__ static inline int fwd_ld(int maxdim) { return round_up(maxdim, 16) + 4; }
// backward dz buffer: scalar A-fragment reads [i][m0+q] -> ld == 2 (mod 32) is conflict-free
__host__ __device__ static inline int bwd_ldz(int maxdim) { return round_up(maxdim, 32) + 2; }
// backward du buffer: float4 epilogue stores -> ld % 4 == 0
__host__ __device__ static inline int bwd_ldu(int maxdim) { return round_up(maxdim, 16) + 4; }

// ------------------------------------------------------------------------------------------------
// GEMM building blocks (one wave, A in LDS, B streamed from global/L2)
// ------------------------------------------------------------------------------------------------
// "NT" form (forward):  Y[r, o] = sum_k Xs[r, k] * W[o, k]      W row-major [M, K]
// One call = one chunk of 16*CT output columns starting at o0, for RT row tiles of 16.
// B fragments: lane (i = l&15, q = l>>4) loads W[o0 + 16t + i][k0 + 4q .. +3] (float4 along k), which is the
// B operand of four consecutive k-steps (any fixed permutation of k inside the contraction is legal).
template <int RT, int CT>
__device__ __forceinline__ void gemm_nt_chunk(const float* __restrict__ Xs, int ldx, int K, int K16,
                                              const float* __restrict__ W, bool vec, const float* __restrict__ bias,
                                              int M, int o0, int act, float* __restrict__ Ys, int ldy,
                                              float* __restrict__ gout, int rows_valid, int lane) {
  const int i = lane & 15, q = lane >> 4;
  f32x4 acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* wrow[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int o = o0 + 16 * t + i;
    wrow[t] = (o < M) ? (W + (int64_t)o * K) : nullptr;
  }
  float4 bc[CT], bn[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    bc[t] = ld4_masked(wrow[t], 4 * q, K, vec);
    bn[t] = bc[t];
  }
  for (int k0 = 0; k0 < K16; k0 += 16) {
    if (k0 + 16 < K16) {
#pragma unroll
      for (int t = 0; t < CT; ++t) bn[t] = ld4_masked(wrow[t], k0 + 16 + 4 * q, K, vec);
    }
    float4 a[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) a[rt] = ld4(Xs + (rt * 16 + i) * ldx + k0 + 4 * q);
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt][t] = mfma16(a[rt].x, bc[t].x, acc[rt][t]);
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt][t] = mfma16(a[rt].y, bc[t].y, acc[rt][t]);
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt][t] = mfma16(a[rt].z, bc[t].z, acc[rt][t]);
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt][t] = mfma16(a[rt].w, bc[t].w, acc[rt][t]);
#pragma unroll
    for (int t = 0; t < CT; ++t) bc[t] = bn[t];
  }
  // epilogue: + bias, activation; to LDS (next layer's input) and, when training, to HBM
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int col = o0 + 16 * t + i;
    if (col < M) {
      const float bv = bias[col];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rt * 16 + 4 * q + r;
          const float v = act_fwd(acc[rt][t][r] + bv, act);
          Ys[row * ldy + col] = v;
          if (gout != nullptr && row < rows_valid) gout[(int64_t)row * M + col] = v;
        }
    }
  }
}

// "NN" form (dgrad):  DU[r, c] = sum_m DZs[r, m] * W[m, c]       W row-major [M, K]
// One call = one chunk of 16*CT output columns starting at c0.  Lane (i, q) loads CT consecutive floats
// W[m0 + 4s + q][c0 + CT*i .. ] for s = 0..3: the B operands of CT interleaved column tiles
// (tile t holds columns c0 + CT*j + t), four m-steps per iteration.
template <int RT, int CT>
__device__ __forceinline__ void gemm_nn_chunk(const float* __restrict__ DZs, int ldz, int M, const float* __restrict__ W,
                                              int K, bool vec, int c0, float* __restrict__ DUs, int ldu, int lane) {
  const int i = lane & 15, q = lane >> 4;
  const int col = c0 + CT * i;
  f32x4 acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int M16 = round_up(M, 16);
  float bc[4][CT], bn[4][CT];
  auto load_w = [&](float(&dst)[4][CT], int m0) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int m = m0 + 4 * s + q;
      const float* row = (m < M) ? (W + (int64_t)m * K) : nullptr;
      if constexpr (CT == 4) {
        const float4 v = ld4_masked(row, col, K, vec);
        dst[s][0] = v.x;
        dst[s][1] = v.y;
        dst[s][2] = v.z;
        dst[s][3] = v.w;
      } else {
#pragma unroll
        for (int t = 0; t < CT; ++t) dst[s][t] = (row != nullptr && col + t < K) ? row[col + t] : 0.f;
      }
    }
  };
  load_w(bc, 0);
  for (int m0 = 0; m0 < M16; m0 += 16) {
    if (m0 + 16 < M16) load_w(bn, m0 + 16);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int m = m0 + 4 * s + q;
      float a[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) a[rt] = (m < M) ? DZs[(rt * 16 + i) * ldz + m] : 0.f;
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt][t] = mfma16(a[rt], bc[s][t], acc[rt][t]);
    }
    if (m0 + 16 < M16) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < CT; ++t) bc[s][t] = bn[s][t];
    }
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rt * 16 + 4 * q + r;
      if constexpr (CT == 4) {
        if (col + 3 < K) {
          st4(DUs + row * ldu + col, make_float4(acc[rt][0][r], acc[rt][1][r], acc[rt][2][r], acc[rt][3][r]));
          continue;
        }
      }
#pragma unroll
      for (int t = 0; t < CT; ++t)
        if (col + t < K) DUs[row * ldu + col + t] = acc[rt][t][r];
    }
}

__device__ __forceinline__ int pick_ct(int width, int nw) {
  // widest column chunk (16*CT) that still gives every wave a chunk
  if (width >= 64 * nw) return 4;
  if (width >= 32 * nw) return 2;
  return 1;
}

// ------------------------------------------------------------------------------------------------
// Forward
// ------------------------------------------------------------------------------------------------
template <int R, int NW>
__global__ __launch_bounds__(NW * 64) void dnn_fwd_kernel(DnnPlan p, const float* __restrict__ params,
                                                          const float* __restrict__ features, int64_t n_docs,
                                                          const int32_t* __restrict__ docids, int B, int L,
                                                          float* __restrict__ scores, float* __restrict__ saved,
                                                          int vecmask) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int RT = R / 16;
  const int64_t N = (int64_t)B * L;
  const int ld = fwd_ld(p.maxdim);
  float* X = smem;
  float* Y = smem + R * ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t n0 = (int64_t)blockIdx.x * R;
  const int rows_valid = (int)((N - n0) < R ? (N - n0) : R);

  // ---- a2: gather feature rows (zero row for the PAD id == n_docs and for rows past N) ----------
  {
    const int F = p.K[0];
    const int F16 = round_up(F, 16);
    const bool vecf = (vecmask >> 31) & 1;
    for (int r = wave; r < R; r += NW) {
      const int64_t n = n0 + r;
      const float* src = nullptr;
      if (n < N) {
        const int b = (int)(n / L), l = (int)(n % L);
        const int64_t id = docids[(int64_t)l * B + b];
        if (id >= 0 && id < n_docs) src = features + id * F;
      }
      for (int c = lane * 4; c < F16; c += 256) st4(X + r * ld + c, ld4_masked(src, c, F, vecf));
    }
  }
  __syncthreads();

  for (int j = 0; j < p.nl; ++j) {
    const int K = p.K[j], M = p.M[j];
    const int K16 = round_up(K, 16);
    const float* lnw = params + p.off_lnw[j];
    const float* lnb = params + p.off_lnb[j];
    // ---- LayerNorm (biased variance, eps 1e-5, affine), in place; two-pass statistics -----------
    for (int r = wave; r < R; r += NW) {
      float* row = X + r * ld;
      float s = 0.f;
      for (int c = lane; c < K; c += 64) s += row[c];
      const float mean = wave_sum(s) / (float)K;
      float v = 0.f;
      for (int c = lane; c < K; c += 64) {
        const float d = row[c] - mean;
        v += d * d;
      }
      const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)K + ULTR_LN_EPS);
      for (int c = lane; c < K16; c += 64) row[c] = (c < K) ? ((row[c] - mean) * rstd * lnw[c] + lnb[c]) : 0.f;
      if (saved != nullptr && lane == 0 && n0 + r < N) {
        saved[p.sv_mean[j] + n0 + r] = mean;
        saved[p.sv_rstd[j] + n0 + r] = rstd;
      }
    }
    __syncthreads();
    const float* W = params + p.off_w[j];
    const float* bias = params + p.off_b[j];
    if (j < p.nl - 1) {
      // ---- Linear + activation on the matrix cores ------------------------------------------------
      const bool vec = (vecmask >> j) & 1;
      float* gout = (saved != nullptr) ? (saved + p.sv_x[j + 1] + n0 * M) : nullptr;
      const int ct = pick_ct(M, NW);
      if (ct == 4) {
        for (int ch = wave; ch * 64 < M; ch += NW)
          gemm_nt_chunk<RT, 4>(X, ld, K, K16, W, vec, bias, M, ch * 64, p.act, Y, ld, gout, rows_valid, lane);
      } else if (ct == 2) {
        for (int ch = wave; ch * 32 < M; ch += NW)
          gemm_nt_chunk<RT, 2>(X, ld, K, K16, W, vec, bias, M, ch * 32, p.act, Y, ld, gout, rows_valid, lane);
      } else {
        for (int ch = wave; ch * 16 < M; ch += NW)
          gemm_nt_chunk<RT, 1>(X, ld, K, K16, W, vec, bias, M, ch * 16, p.act, Y, ld, gout, rows_valid, lane);
      }
      __syncthreads();
      float* t = X;
      X = Y;
      Y = t;
    } else {
      // ---- final Linear(K, 1): a dot product per row, wave-shuffle reduction ---------------------
      for (int r = wave; r < R; r += NW) {
        const float* row = X + r * ld;
        float s = 0.f;
        for (int c = lane; c < K; c += 64) s += row[c] * W[c];
        s = wave_sum(s);
        if (lane == 0 && n0 + r < N) scores[n0 + r] = s + bias[0];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Backward, row-local half
// ------------------------------------------------------------------------------------------------
template <int R, int NW>
__global__ __launch_bounds__(NW * 64) void dnn_bwd_kernel(DnnPlan p, BwdPlan bp, const float* __restrict__ params,
                                                          const float* __restrict__ features, int64_t n_docs,
                                                          const int32_t* __restrict__ docids, int B, int L,
                                                          const float* __restrict__ saved,
                                                          const float* __restrict__ dscores, float* __restrict__ ws,
                                                          int vecmask) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int RT = R / 16;
  constexpr int NT = NW * 64;
  const int64_t N = (int64_t)B * L;
  const int ldz = bwd_ldz(p.maxdim), ldu = bwd_ldu(p.maxdim);
  float* DU = smem;                    // [R][ldu]   (first: 16-byte aligned float4 stores)
  float* DZ = DU + R * ldu;            // [R][ldz]
  float* sm_ds = DZ + R * ldz;         // [R]
  float* sm_mean2 = sm_ds + R;         // [2][R]  double-buffered by layer parity (no extra barrier)
  float* sm_rstd2 = sm_mean2 + 2 * R;  // [2][R]
  int64_t* sm_id = reinterpret_cast<int64_t*>(sm_rstd2 + 2 * R);  // [R] feature row id or -1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t n0 = (int64_t)blockIdx.x * R;
  float* vslab = ws + bp.vslab_off + (int64_t)blockIdx.x * bp.vlen;

  if (tid < R) {
    const int64_t n = n0 + tid;
    float ds = 0.f;
    int64_t id = -1;
    if (n < N) {
      ds = dscores[n];
      const int b = (int)(n / L), l = (int)(n % L);
      const int64_t d = docids[(int64_t)l * B + b];
      if (d >= 0 && d < n_docs) id = d;
    }
    sm_ds[tid] = ds;
    sm_id[tid] = id;
  }

  for (int j = p.nl - 1; j >= 0; --j) {
    const int K = p.K[j], M = p.M[j];
    const bool last = (j == p.nl - 1);
    const float* lnw = params + p.off_lnw[j];
    const float* lnb = params + p.off_lnb[j];
    const float* W = params + p.off_w[j];
    float* sm_mean = sm_mean2 + (j & 1) * R;
    float* sm_rstd = sm_rstd2 + (j & 1) * R;
    if (tid < R) {
      const int64_t n = n0 + tid;
      sm_mean[tid] = (n < N) ? saved[p.sv_mean[j] + n] : 0.f;
      sm_rstd[tid] = (n < N) ? saved[p.sv_rstd[j] + n] : 0.f;
    }
    __syncthreads();  // sm_* visible; DZ of the previous iteration complete
    // ---- du_j = dz_j . W_j ------------------------------------------------------------------------
    if (last) {
      for (int r = wave; r < R; r += NW) {
        const float ds = sm_ds[r];
        for (int c = lane; c < K; c += 64) DU[r * ldu + c] = ds * W[c];
      }
    } else {
      const bool vec = (vecmask >> j) & 1;
      const int ct = pick_ct(K, NW);
      if (ct == 4) {
        for (int ch = wave; ch * 64 < K; ch += NW) gemm_nn_chunk<RT, 4>(DZ, ldz, M, W, K, vec, ch * 64, DU, ldu, lane);
      } else if (ct == 2) {
        for (int ch = wave; ch * 32 < K; ch += NW) gemm_nn_chunk<RT, 2>(DZ, ldz, M, W, K, false, ch * 32, DU, ldu, lane);
      } else {
        for (int ch = wave; ch * 16 < K; ch += NW) gemm_nn_chunk<RT, 1>(DZ, ldz, M, W, K, false, ch * 16, DU, ldu, lane);
      }
    }
    __syncthreads();
    // ---- column pass: per-row-block partial sums of the vector-parameter gradients ---------------
    //   dgamma_j[c] = sum_r du[r,c] xhat[r,c]   dbeta_j[c] = sum_r du[r,c]
    //   final layer: dW[c] = sum_r ds[r] u[r,c], db = sum_r ds[r]
    for (int c = tid; c < K; c += NT) {
      float pg = 0.f, pb = 0.f, pw = 0.f;
      const float g = lnw[c], be = lnb[c];
      for (int r = 0; r < R; ++r) {
        const int64_t n = n0 + r;
        if (n >= N) break;
        float x;
        if (j == 0) {
          const int64_t id = sm_id[r];
          x = (id >= 0) ? features[id * K + c] : 0.f;
        } else {
          x = saved[p.sv_x[j] + n * K + c];
        }
        const float xh = (x - sm_mean[r]) * sm_rstd[r];
        const float du = DU[r * ldu + c];
        pg += du * xh;
        pb += du;
        if (last) pw += sm_ds[r] * (g * xh + be);
      }
      vslab[bp.voff_g[j] + c] = pg;
      vslab[bp.voff_b[j] + c] = pb;
      if (last) vslab[bp.voff_wk + c] = pw;
    }
    if (last && tid == 0) {
      float s = 0.f;
      for (int r = 0; r < R; ++r) s += sm_ds[r];
      vslab[bp.voff_bk] = s;
    }
    // ---- row pass: LayerNorm backward, then through the previous activation -> dz_{j-1} ----------
    if (j > 0) {
      float* dzg = ws + bp.dz_off[j - 1];
      for (int r = wave; r < R; r += NW) {
        const int64_t n = n0 + r;
        const bool valid = n < N;
        const float mean = sm_mean[r], rstd = sm_rstd[r];
        const float* xrow = valid ? (saved + p.sv_x[j] + n * K) : nullptr;
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < K; c += 64) {
          const float x = valid ? xrow[c] : 0.f;
          const float xh = (x - mean) * rstd;
          const float gx = DU[r * ldu + c] * lnw[c];
          s1 += gx;
          s2 += gx * xh;
        }
        s1 = wave_sum(s1) / (float)K;
        s2 = wave_sum(s2) / (float)K;
        for (int c = lane; c < K; c += 64) {
          const float x = valid ? xrow[c] : 0.f;
          const float xh = (x - mean) * rstd;
          const float gx = DU[r * ldu + c] * lnw[c];
          const float dx = rstd * (gx - s1 - xh * s2);
          const float dzv = dx * act_grad_from_out(x, p.act);
          DZ[r * ldz + c] = dzv;
          if (valid) dzg[n * K + c] = dzv;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradients of the hidden Linears: dW_j[m,k] = sum_n dz_j[n,m] u_j[n,k],  db_j[m] = sum_n dz_j[n,m]
// ------------------------------------------------------------------------------------------------
// Workgroup = 4 waves on ONE 64x64 output block; each wave contracts a different quarter of the block's row
// split, then the four 64x64 partials are summed through LDS in fixed order and written to the split's slab.
// Per step a lane issues two 16-byte loads (dz row piece along m, x row piece along k) feeding 16 MFMAs:
// A[i][kk] = dz[n+kk][m0+4i+ta], B[kk][j] = u[n+kk][k0+4j+tb]  ->  D_{ta,tb}[i][j] = dW[m0+4i+ta][k0+4j+tb].
__global__ __launch_bounds__(256) void dnn_wgrad_kernel(DnnPlan p, BwdPlan bp, const float* __restrict__ params,
                                                        const float* __restrict__ features, int64_t n_docs,
                                                        const int32_t* __restrict__ docids, int B, int L,
                                                        const float* __restrict__ saved, float* __restrict__ ws,
                                                        int vecf) {
  __shared__ __attribute__((aligned(16))) float red[4][64 * 64];
  __shared__ float bred[4][64];
  const int64_t N = bp.N;
  int j = 0;
  while (j + 1 < p.nl - 1 && (int)blockIdx.x >= bp.wl[j + 1].blk_begin) ++j;
  const WgradLayer wl = bp.wl[j];
  const int local = blockIdx.x - wl.blk_begin;
  const int split = local % wl.nsplit;
  const int tile = local / wl.nsplit;
  const int mb = tile / wl.nkb, kb = tile % wl.nkb;
  const int M = wl.M, K = wl.K;
  const int m0 = mb * 64, k0 = kb * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const bool vec = wl.vec != 0;
  const bool vecx = (j == 0) ? (vecf != 0) : vec;
  const int rpw = wl.rows_per_split / 4;
  const int64_t nbeg = (int64_t)split * wl.rows_per_split + (int64_t)wave * rpw;
  int64_t nend = nbeg + rpw;
  if (nend > N) nend = N;

  const float* dz = ws + wl.dz_off;
  const float* xs = (j == 0) ? nullptr : (saved + p.sv_x[j]);
  const float* meanp = saved + p.sv_mean[j];
  const float* rstdp = saved + p.sv_rstd[j];
  const float4 gam = ld4_masked(params + p.off_lnw[j], k0 + 4 * i, K, false);
  const float4 bet = ld4_masked(params + p.off_lnb[j], k0 + 4 * i, K, false);
  const int kc = k0 + 4 * i;
  const bool k_ok0 = kc < K, k_ok1 = kc + 1 < K, k_ok2 = kc + 2 < K, k_ok3 = kc + 3 < K;

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

  auto load_step = [&](int64_t n, float4& a4, float4& b4) {
    a4 = make_float4(0.f, 0.f, 0.f, 0.f);
    b4 = a4;
    if (n < nend) {
      a4 = ld4_masked(dz + n * M, m0 + 4 * i, M, vec);
      const float* xrow;
      if (j == 0) {
        const int b = (int)(n / L), l = (int)(n % L);
        const int64_t id = docids[(int64_t)l * B + b];
        xrow = (id >= 0 && id < n_docs) ? (features + id * K) : nullptr;
      } else {
        xrow = xs + n * K;
      }
      const float4 x4 = ld4_masked(xrow, kc, K, vecx);
      const float mean = meanp[n], rstd = rstdp[n];
      b4.x = k_ok0 ? ((x4.x - mean) * rstd * gam.x + bet.x) : 0.f;
      b4.y = k_ok1 ? ((x4.y - mean) * rstd * gam.y + bet.y) : 0.f;
      b4.z = k_ok2 ? ((x4.z - mean) * rstd * gam.z + bet.z) : 0.f;
      b4.w = k_ok3 ? ((x4.w - mean) * rstd * gam.w + bet.w) : 0.f;
    }
  };

  float4 a_c, b_c, a_n, b_n;
  load_step(nbeg + q, a_c, b_c);
  for (int64_t n = nbeg; n < nend; n += 4) {
    load_step(n + 4 + q, a_n, b_n);
    bsum.x += a_c.x;
    bsum.y += a_c.y;
    bsum.z += a_c.z;
    bsum.w += a_c.w;
    const float av[4] = {a_c.x, a_c.y, a_c.z, a_c.w};
    const float bv[4] = {b_c.x, b_c.y, b_c.z, b_c.w};
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = mfma16(av[ta], bv[tb], acc[ta][tb]);
    a_c = a_n;
    b_c = b_n;
  }
  // ---- cross-wave reduction through LDS (fixed order) -------------------------------------------
  // lane holds D_{ta,tb}[row = 4q + r][col = i]  ->  block-local (m = 4*(4q+r) + ta, k = 4*i + tb)
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ml = 4 * (4 * q + r) + ta;
      st4(&red[wave][ml * 64 + 4 * i], make_float4(acc[ta][0][r], acc[ta][1][r], acc[ta][2][r], acc[ta][3][r]));
    }
  // bias partial: sum over the 4 row groups q (lanes i, i+16, i+32, i+48)
  {
    float4 s = bsum;
    s.x += __shfl_xor(s.x, 16, 64); s.y += __shfl_xor(s.y, 16, 64); s.z += __shfl_xor(s.z, 16, 64); s.w += __shfl_xor(s.w, 16, 64);
    s.x += __shfl_xor(s.x, 32, 64); s.y += __shfl_xor(s.y, 32, 64); s.z += __shfl_xor(s.z, 32, 64); s.w += __shfl_xor(s.w, 32, 64);
    if (q == 0) {
      bred[wave][4 * i + 0] = s.x;
      bred[wave][4 * i + 1] = s.y;
      bred[wave][4 * i + 2] = s.z;
      bred[wave][4 * i + 3] = s.w;
    }
  }
  __syncthreads();
  float* slab = ws + wl.slab_off + (int64_t)split * ((int64_t)M * K + M);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int e = tid + 256 * it;  // float4 index inside the 64x64 block
    const int ml = e >> 4, k4 = (e & 15) * 4;
    const float4 v0 = ld4(&red[0][ml * 64 + k4]), v1 = ld4(&red[1][ml * 64 + k4]);
    const float4 v2 = ld4(&red[2][ml * 64 + k4]), v3 = ld4(&red[3][ml * 64 + k4]);
    float4 s;
    s.x = ((v0.x + v1.x) + v2.x) + v3.x;
    s.y = ((v0.y + v1.y) + v2.y) + v3.y;
    s.z = ((v0.z + v1.z) + v2.z) + v3.z;
    s.w = ((v0.w + v1.w) + v2.w) + v3.w;
    const int m = m0 + ml, k = k0 + k4;
    if (m < M && k < K) {
      float* dst = slab + (int64_t)m * K + k;
      if (vec && k + 3 < K) {
        st4(dst, s);
      } else {
        dst[0] = s.x;
        if (k + 1 < K) dst[1] = s.y;
        if (k + 2 < K) dst[2] = s.z;
        if (k + 3 < K) dst[3] = s.w;
      }
    }
  }
  if (kb == 0 && tid < 64 && m0 + tid < M)
    slab[(int64_t)M * K + m0 + tid] = ((bred[0][tid] + bred[1][tid]) + bred[2][tid]) + bred[3][tid];
}

// ------------------------------------------------------------------------------------------------
// Slab reduction -> flat gradient, step tail, sum-of-squares partials
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* sm) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  const float t = ((sm[0] + sm[1]) + sm[2]) + sm[3];
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(256) void grad_reduce_kernel(RedPlan rp, int64_t P, int tail, const float* __restrict__ ws,
                                                          const float* __restrict__ loss_part, int n_loss_part,
                                                          float* __restrict__ grads, float* __restrict__ sumsq_part) {
  __shared__ float sm[4];
  float sq = 0.f;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int64_t e = (int64_t)blockIdx.x * 1024 + it * 256 + threadIdx.x;
    if (e < P) {
      int s = 0;
      while (s + 1 < rp.nseg && e >= rp.seg[s + 1].off) ++s;
      const RedSeg sg = rp.seg[s];
      const float* src = ws + sg.base + (e - sg.off);
      float g = 0.f;
      for (int k = 0; k < sg.nparts; ++k) g += src[(int64_t)k * sg.stride];
      grads[e] = g;
      sq += g * g;
    } else if (e < P + tail) {
      const int t = (int)(e - P);
      float g = 0.f;
      if (loss_part != nullptr)
        for (int k = 0; k < n_loss_part; ++k) g += loss_part[(int64_t)k * tail + t];
      grads[e] = g;
    }
  }
  const float tot = block_sum_256(sq, sm);
  if (threadIdx.x == 0) sumsq_part[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void grad_sumsq_kernel(int64_t P, const float* __restrict__ grads,
                                                         float* __restrict__ sumsq_part) {
  __shared__ float sm[4];
  float sq = 0.f;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int64_t e = (int64_t)blockIdx.x * 1024 + it * 256 + threadIdx.x;
    if (e < P) {
      const float g = grads[e];
      sq += g * g;
    }
  }
  const float tot = block_sum_256(sq, sm);
  if (threadIdx.x == 0) sumsq_part[blockIdx.x] = tot;
}

// ================================================================================================
// Host side: plans and launches
// ================================================================================================
static int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}

extern "C" int ultr_abi_version(void) { return ULTR_ABI_VERSION; }

static bool desc_ok(const ultr_dnn_desc* d) {
  if (!d || d->feature_size <= 0 || d->n_hidden < 0 || d->n_hidden > ULTR_MAX_HIDDEN) return false;
  for (int j = 0; j < d->n_hidden; ++j)
    if (d->hidden[j] <= 0) return false;
  return d->activation == ULTR_ACT_ELU || d->activation == ULTR_ACT_RELU;
}

bool ultr_make_dnn_plan(const ultr_dnn_desc* d, int64_t N, DnnPlan* p) {
  if (!desc_ok(d)) return false;
  memset(p, 0, sizeof(*p));
  p->nl = d->n_hidden + 1;
  p->act = d->activation;
  int k = d->feature_size;
  int64_t off = 0;
  p->maxdim = k;
  for (int j = 0; j < p->nl; ++j) {
    const int m = (j < d->n_hidden) ? d->hidden[j] : 1;
    p->K[j] = k;
    p->M[j] = m;
    p->off_lnw[j] = off; off += k;
    p->off_lnb[j] = off; off += k;
    p->off_w[j] = off;   off += (int64_t)m * k;
    p->off_b[j] = off;   off += m;
    if (m > p->maxdim) p->maxdim = m;
    k = m;
  }
  p->P = off;
  int64_t sv = 0;
  for (int j = 1; j < p->nl; ++j) {
    p->sv_x[j] = sv;
    sv += N * p->K[j];
    sv = (sv + 3) & ~(int64_t)3;
  }
  for (int j = 0; j < p->nl; ++j) {
    p->sv_mean[j] = sv; sv += N;
    p->sv_rstd[j] = sv; sv += N;
  }
  p->sv_total = sv;
  return true;
}

static int fwd_rows_per_wg(const DnnPlan& p, int64_t N) {
  int r = env_int("ULTR_FWD_R", 0);
  if (r == 16 || r == 32) return r;
  const size_t lds32 = (size_t)2 * 32 * fwd_ld(p.maxdim) * sizeof(float);
  return ((N + 15) / 16 > 512 && lds32 <= 160 * 1024) ? 32 : 16;
}
static size_t bwd_lds_bytes(const DnnPlan& p, int R) {
  return ((size_t)R * (bwd_ldu(p.maxdim) + bwd_ldz(p.maxdim)) + 5 * (size_t)R) * sizeof(float) + (size_t)R * sizeof(int64_t);
}
static int bwd_rows_per_wg(const DnnPlan& p, int64_t N) {
  int r = env_int("ULTR_BWD_R", 0);
  if (r == 16 || r == 32) return r;
  return ((N + 15) / 16 > 512 && bwd_lds_bytes(p, 32) <= 160 * 1024) ? 32 : 16;
}

bool ultr_make_bwd_plan(const DnnPlan& p, int64_t N, BwdPlan* bp) {
  memset(bp, 0, sizeof(*bp));
  bp->N = N;
  bp->rblk = bwd_rows_per_wg(p, N);
  bp->nrb = (int)((N + bp->rblk - 1) / bp->rblk);
  int v = 0;
  for (int j = 0; j < p.nl; ++j) {
    bp->voff_g[j] = v; v += p.K[j];
    bp->voff_b[j] = v; v += p.K[j];
  }
  bp->voff_wk = v; v += p.K[p.nl - 1];
  bp->voff_bk = v; v += 1;
  bp->vlen = v;
  int64_t off = 0;
  // sum-of-squares partials first (fixed, small)
  const int64_t tail_max = 4096;  // generous: tail is 4 + 2L floats
  bp->n_red_blocks = (int)((p.P + tail_max + 1023) / 1024);
  bp->sumsq_off = off; off += bp->n_red_blocks; off = (off + 3) & ~(int64_t)3;
  bp->vslab_off = off; off += (int64_t)bp->nrb * bp->vlen; off = (off + 3) & ~(int64_t)3;
  for (int j = 0; j < p.nl - 1; ++j) {
    bp->dz_off[j] = off; off += N * p.M[j]; off = (off + 3) & ~(int64_t)3;
  }
  // wgrad geometry
  int tiles = 0;
  for (int j = 0; j < p.nl - 1; ++j) tiles += ((p.M[j] + 63) / 64) * ((p.K[j] + 63) / 64);
  const int target = env_int("ULTR_WGRAD_WGS", 384);
  int blk = 0;
  for (int j = 0; j < p.nl - 1; ++j) {
    WgradLayer& w = bp->wl[j];
    w.M = p.M[j]; w.K = p.K[j];
    w.nmb = (w.M + 63) / 64; w.nkb = (w.K + 63) / 64;
    int nsplit = tiles > 0 ? (target + tiles - 1) / tiles : 1;
    if (nsplit < 1) nsplit = 1;
    int64_t rps = (N + nsplit - 1) / nsplit;
    rps = (rps + 15) / 16 * 16;
    if (rps < 64) rps = 64;
    w.rows_per_split = (int)rps;
    w.nsplit = (int)((N + rps - 1) / rps);
    w.blk_begin = blk;
    blk += w.nmb * w.nkb * w.nsplit;
    w.vec = (w.M % 4 == 0 && w.K % 4 == 0) ? 1 : 0;
    w.dz_off = bp->dz_off[j];
    w.slab_off = off; off += (int64_t)w.nsplit * ((int64_t)w.M * w.K + w.M); off = (off + 3) & ~(int64_t)3;
  }
  bp->wgrad_blocks = blk;
  bp->total = off;
  return true;
}

static void make_red_plan(const DnnPlan& p, const BwdPlan& bp, RedPlan* rp) {
  int s = 0;
  for (int j = 0; j < p.nl; ++j) {
    const bool last = (j == p.nl - 1);
    rp->seg[s++] = RedSeg{p.off_lnw[j], bp.vslab_off + bp.voff_g[j], bp.vlen, p.K[j], bp.nrb};
    rp->seg[s++] = RedSeg{p.off_lnb[j], bp.vslab_off + bp.voff_b[j], bp.vlen, p.K[j], bp.nrb};
    if (last) {
      rp->seg[s++] = RedSeg{p.off_w[j], bp.vslab_off + bp.voff_wk, bp.vlen, p.K[j], bp.nrb};
      rp->seg[s++] = RedSeg{p.off_b[j], bp.vslab_off + bp.voff_bk, bp.vlen, 1, bp.nrb};
    } else {
      const WgradLayer& w = bp.wl[j];
      const int64_t stride = (int64_t)w.M * w.K + w.M;
      rp->seg[s++] = RedSeg{p.off_w[j], w.slab_off, stride, w.M * w.K, w.nsplit};
      rp->seg[s++] = RedSeg{p.off_b[j], w.slab_off + (int64_t)w.M * w.K, stride, w.M, w.nsplit};
    }
  }
  rp->nseg = s;
}

static int vecmask_for(const DnnPlan& p, const float* params, const float* features) {
  int mask = 0;
  const bool pa = ((uintptr_t)params & 15) == 0;
  for (int j = 0; j < p.nl; ++j)
    if (pa && p.K[j] % 4 == 0 && p.off_w[j] % 4 == 0) mask |= (1 << j);
  if (features != nullptr && ((uintptr_t)features & 15) == 0 && p.K[0] % 4 == 0) mask |= (1u << 31);
  return mask;
}

extern "C" int64_t ultr_dnn_param_count(const ultr_dnn_desc* d) {
  DnnPlan p;
  return ultr_make_dnn_plan(d, 0, &p) ? p.P : 0;
}
extern "C" int ultr_dnn_param_offsets(const ultr_dnn_desc* d, int64_t* offsets) {
  DnnPlan p;
  if (!offsets || !ultr_make_dnn_plan(d, 0, &p)) return ULTR_E_BADARG;
  for (int j = 0; j < p.nl; ++j) {
    offsets[4 * j + 0] = p.off_lnw[j];
    offsets[4 * j + 1] = p.off_lnb[j];
    offsets[4 * j + 2] = p.off_w[j];
    offsets[4 * j + 3] = p.off_b[j];
  }
  return 0;
}
extern "C" int64_t ultr_dnn_saved_bytes(const ultr_dnn_desc* d, int64_t n_rows) {
  DnnPlan p;
  if (n_rows < 0 || !ultr_make_dnn_plan(d, n_rows, &p)) return 0;
  return (p.sv_total + 4) * (int64_t)sizeof(float);
}
extern "C" int64_t ultr_dnn_bwd_workspace_bytes(const ultr_dnn_desc* d, int64_t n_rows) {
  DnnPlan p;
  BwdPlan bp;
  if (n_rows < 0 || !ultr_make_dnn_plan(d, n_rows, &p)) return 0;
  // worst case over the env-tunable geometry: size for both row-block choices
  ultr_make_bwd_plan(p, n_rows, &bp);
  int64_t t = bp.total;
  const int64_t extra = ((n_rows + 15) / 16) * (int64_t)bp.vlen;  // if rblk were 16
  return (t + extra + 1024) * (int64_t)sizeof(float);
}
extern "C" int64_t ultr_step_tail_floats(int32_t list_size) { return ultr_tail_len(list_size); }

template <typename KernelT>
static hipError_t set_lds(KernelT k, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

extern "C" int ultr_dnn_forward(const ultr_dnn_desc* d, const float* params, const float* features, int64_t n_docs,
                                const int32_t* docids, int32_t batch, int32_t list_size, float* scores, void* saved,
                                void* stream) {
  if (!params || !docids || !scores || batch <= 0 || list_size <= 0 || n_docs < 0 || (n_docs > 0 && !features))
    return ULTR_E_BADARG;
  const int64_t N = (int64_t)batch * list_size;
  DnnPlan p;
  if (!ultr_make_dnn_plan(d, N, &p)) return ULTR_E_BADARG;
  const int R = fwd_rows_per_wg(p, N);
  const size_t lds = (size_t)2 * R * fwd_ld(p.maxdim) * sizeof(float);
  if (lds > 160 * 1024) return ULTR_E_UNSUPPORTED;
  const int nw = env_int("ULTR_FWD_NW", 8);
  const int vm = vecmask_for(p, params, features);
  const dim3 grid((unsigned)((N + R - 1) / R));
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipSuccess;
#define LAUNCH_FWD(RR, NWW)                                                                                     \
  do {                                                                                                          \
    e = set_lds(dnn_fwd_kernel<RR, NWW>, lds);                                                                  \
    if (e != hipSuccess) return (int)e;                                                                         \
    hipLaunchKernelGGL((dnn_fwd_kernel<RR, NWW>), grid, dim3(NWW * 64), lds, st, p, params, features, n_docs,   \
                       docids, (int)batch, (int)list_size, scores, (float*)saved, vm);                         \
  } while (0)
  if (R == 16 && nw == 4) LAUNCH_FWD(16, 4);
  else if (R == 16) LAUNCH_FWD(16, 8);
  else if (nw == 4) LAUNCH_FWD(32, 4);
  else LAUNCH_FWD(32, 8);
#undef LAUNCH_FWD
  return (int)hipGetLastError();
}

extern "C" int ultr_dnn_backward(const ultr_dnn_desc* d, const float* params, const float* features, int64_t n_docs,
                                 const int32_t* docids, int32_t batch, int32_t list_size, const void* saved,
                                 const float* dscores, const void* loss_ws, void* bwd_ws, float* grads, void* stream) {
  if (!params || !docids || !saved || !dscores || !bwd_ws || !grads || batch <= 0 || list_size <= 0 || n_docs < 0 ||
      (n_docs > 0 && !features))
    return ULTR_E_BADARG;
  const int64_t N = (int64_t)batch * list_size;
  DnnPlan p;
  BwdPlan bp;
  if (!ultr_make_dnn_plan(d, N, &p) || !ultr_make_bwd_plan(p, N, &bp)) return ULTR_E_BADARG;
  const int tail = (int)ultr_tail_len(list_size);
  if (tail > 4096) return ULTR_E_UNSUPPORTED;
  const size_t lds = bwd_lds_bytes(p, bp.rblk);
  if (lds > 160 * 1024) return ULTR_E_UNSUPPORTED;
  const int nw = env_int("ULTR_BWD_NW", 8);
  const int vm = vecmask_for(p, params, features);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipSuccess;
  float* ws = (float*)bwd_ws;
#define LAUNCH_BWD(RR, NWW)                                                                                        \
  do {                                                                                                             \
    e = set_lds(dnn_bwd_kernel<RR, NWW>, lds);                                                                     \
    if (e != hipSuccess) return (int)e;                                                                            \
    hipLaunchKernelGGL((dnn_bwd_kernel<RR, NWW>), dim3(bp.nrb), dim3(NWW * 64), lds, st, p, bp, params, features,  \
                       n_docs, docids, (int)batch, (int)list_size, (const float*)saved, dscores, ws, vm);         \
  } while (0)
  if (bp.rblk == 16 && nw == 4) LAUNCH_BWD(16, 4);
  else if (bp.rblk == 16) LAUNCH_BWD(16, 8);
  else if (nw == 4) LAUNCH_BWD(32, 4);
  else LAUNCH_BWD(32, 8);
#undef LAUNCH_BWD
  e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  if (bp.wgrad_blocks > 0) {
    hipLaunchKernelGGL(dnn_wgrad_kernel, dim3(bp.wgrad_blocks), dim3(256), 0, st, p, bp, params, features, n_docs, docids,
                       (int)batch, (int)list_size, (const float*)saved, ws, (vm >> 31) & 1);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  RedPlan rp;
  make_red_plan(p, bp, &rp);
  const int nblk = (int)((p.P + tail + 1023) / 1024);
  const float* lp = (const float*)loss_ws;
  hipLaunchKernelGGL(grad_reduce_kernel, dim3(nblk), dim3(256), 0, st, rp, p.P, tail, (const float*)ws, lp,
                     (int)ultr_loss_parts(batch), grads, ws + bp.sumsq_off);
  return (int)hipGetLastError();
}

extern "C" int ultr_grad_sumsq(float* grads, int64_t n_params, int32_t list_size, void* bwd_ws, void* stream) {
  if (!grads || !bwd_ws || n_params <= 0) return ULTR_E_BADARG;
  const int tail = (int)ultr_tail_len(list_size);
  const int nblk = (int)((n_params + tail + 1023) / 1024);
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, n_params, (const float*)grads,
                     (float*)bwd_ws);  // sumsq partials live at offset 0 of bwd_ws
  return (int)hipGetLastError();
}
