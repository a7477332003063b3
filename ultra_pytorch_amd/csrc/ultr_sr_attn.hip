// ultr_sr_attn.hip - SetRank's self-attention (reference ultra/ranking_model/SetRank.py:33-37, 54-66, 159-195: MultiHeadAttention WITHOUT
// Q / K / V projections - the heads are slices of x itself), forward and backward, one (list, head) per workgroup with the score tile in
// registers.  Kernel families: fp32 matrix cores (`sr_attn_fwd_mfma_kernel`, `sr_attn_bwd_mfma_kernel`: exact fp32, the default forward),
// the split-half BACKWARD (`sr_attn_bwd_h3_kernel`: hi / lo f16 operands, scores unchanged bit for bit; default at head depth 32 / 64),
// plain-f16 operands on request (`*_f16_kernel`: ultr_setrank_desc.attention_dtype = ULTR_ATTN_FP16, ordering-level parity) and scalar
// kernels for head depths the matrix cores do not take.  Split out of ultr_setrank.hip in round 6 (launch interface: ultr_sr_attn.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/ultr_hip.h"
#include "ultr_device.h"
#include "ultr_sr_attn.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// self-attention over one list, one head per workgroup (no projections: q = k = v = x[:, head slice])
// ---------------------------------------------------------------------------------------------------------
#define SR_KPL 4  // keys per lane: lists up to 256 documents

// A[i, :] = softmax_j(x_i . x_j / sqrt(dh)) @ x        (SetRank.py:159-195, mask = None)
__global__ __launch_bounds__(256) void sr_attn_fwd_kernel(const float* __restrict__ x, int L, int d, int dh, float* __restrict__ A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ldx = dh + 1;
  float* xs = smem;                 // [L][dh + 1]
  float* ps = xs + L * ldx;         // [4 waves][L] probabilities of the wave's current query row
  const int b = blockIdx.x, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xb = x + (int64_t)b * L * d + h * dh;
  for (int e = tid; e < L * dh; e += 256) {
    const int r = e / dh, c = e - r * dh;
    xs[r * ldx + c] = xb[(int64_t)r * d + c];
  }
  __syncthreads();
  const float scale = 1.0f / sqrtf((float)dh);
  float* pw = ps + wave * L;
  for (int i = wave; i < L; i += 4) {
    float sc[SR_KPL];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < SR_KPL; ++k) {
      const int j = lane + 64 * k;
      float acc = 0.f;
      if (j < L)
        for (int c = 0; c < dh; ++c) acc += xs[i * ldx + c] * xs[j * ldx + c];
      sc[k] = (j < L) ? acc * scale : -INFINITY;
      mx = fmaxf(mx, sc[k]);
    }
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < SR_KPL; ++k) {
      const int j = lane + 64 * k;
      sc[k] = (j < L) ? expf(sc[k] - mx) : 0.f;
      se += sc[k];
    }
    se = wave_sum(se);
#pragma unroll
    for (int k = 0; k < SR_KPL; ++k) {
      const int j = lane + 64 * k;
      if (j < L) pw[j] = sc[k] / se;
    }
    __builtin_amdgcn_wave_barrier();
    // lane c < dh: A[i][c] = sum_j p_j x_j[c]
    for (int c = lane; c < dh; c += 64) {
      float acc = 0.f;
      for (int j = 0; j < L; ++j) acc += pw[j] * xs[j * ldx + c];
      A[((int64_t)b * L + i) * d + h * dh + c] = acc;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// dx[:, head slice] += dq + dk + dv   with  P recomputed from x,  dP = dA x^T,  dS = P o (dP - rowsum(P o dP)),
//   dq_i = scale * sum_j dS_ij x_j,  dk_j = scale * sum_i dS_ij x_i,  dv_j = sum_i P_ij dA_i      (L <= 120)
__global__ __launch_bounds__(256) void sr_attn_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dA, int L, int d,
                                                          int dh, float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ldx = dh + 1;
  float* xs = smem;                 // [L][dh + 1]
  float* das = xs + L * ldx;        // [L][dh + 1]
  float* Pm = das + L * ldx;        // [L][L]
  float* Sm = Pm + L * L;           // [L][L]  dS
  const int b = blockIdx.x, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t base = (int64_t)b * L * d + h * dh;
  for (int e = tid; e < L * dh; e += 256) {
    const int r = e / dh, c = e - r * dh;
    xs[r * ldx + c] = x[base + (int64_t)r * d + c];
    das[r * ldx + c] = dA[base + (int64_t)r * d + c];
  }
  __syncthreads();
  const float scale = 1.0f / sqrtf((float)dh);
  // pass 1: rows of P and dS (one wave per query row), dq
  for (int i = wave; i < L; i += 4) {
    float sc[2], dp[2];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int j = lane + 64 * k;
      float acc = 0.f, accp = 0.f;
      if (j < L)
        for (int c = 0; c < dh; ++c) {
          acc += xs[i * ldx + c] * xs[j * ldx + c];
          accp += das[i * ldx + c] * xs[j * ldx + c];
        }
      sc[k] = (j < L) ? acc * scale : -INFINITY;
      dp[k] = accp;
      mx = fmaxf(mx, sc[k]);
    }
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int j = lane + 64 * k;
      sc[k] = (j < L) ? expf(sc[k] - mx) : 0.f;
      se += sc[k];
    }
    se = wave_sum(se);
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      sc[k] /= se;  // p_ij
      t += sc[k] * dp[k];
    }
    t = wave_sum(t);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int j = lane + 64 * k;
      if (j < L) {
        Pm[i * L + j] = sc[k];
        Sm[i * L + j] = sc[k] * (dp[k] - t);
      }
    }
  }
  __syncthreads();
  // pass 2: one thread per (token r, channel c): dq (row r of dS), dk and dv (column r of dS / P)
  for (int e = tid; e < L * dh; e += 256) {
    const int r = e / dh, c = e - r * dh;
    float dq = 0.f, dk = 0.f, dv = 0.f;
    for (int j = 0; j < L; ++j) {
      dq += Sm[r * L + j] * xs[j * ldx + c];
      dk += Sm[j * L + r] * xs[j * ldx + c];
      dv += Pm[j * L + r] * das[j * ldx + c];
    }
    dx[base + (int64_t)r * d + c] += scale * (dq + dk) + dv;
  }
}

// ---------------------------------------------------------------------------------------------------------
// the same attention on the matrix cores (fp32 v_mfma_f32_16x16x4_f32): list_size <= 128, head depth 16 / 32 / 64
// ---------------------------------------------------------------------------------------------------------
// One workgroup per (list, head), one WAVE per block of 16 tokens; the only LDS is the head slice x_h [Lp, DH]
// (Lp = L rounded up to 16, zero rows past L).  The [L, L] score matrix never exists in memory:
//  * SetRank has no Q/K/V projections, so S = x x^T is symmetric.  The wave computes the TRANSPOSED tiles
//    D[key][query] = x[key tile] . x[query block]^T: in the MFMA result layout lane (i, q) then holds, for ITS query i,
//    the scores of keys {16 t + 4 q + r}: the whole softmax row sits in 4 lanes (in-register max / sum + two
//    cross-lane steps), and the probabilities are ALREADY the A operand of P.V (any fixed permutation of the
//    contraction index is legal as long as the B operand follows it: B reads V[key(q, t, r)][c] from LDS).
//  * A lane group q = lane >> 4 owns a contiguous quarter of the head depth in the S contraction (float4 LDS reads).
constexpr int SR_MAXT = 8;  // 16-token blocks per list (list_size <= 128)

// stage the head slice(s) into LDS, zero rows past L
template <int DH, bool WITH_DA>
__device__ __forceinline__ void sr_stage_head(const float* __restrict__ x, const float* __restrict__ dA, int64_t base, int L, int Lp,
                                              int d, float* xs, float* das, int tid, int nthr) {
  constexpr int LDX = DH + 4;
  for (int e = tid; e < Lp * (DH / 4); e += nthr) {
    const int r = e / (DH / 4), c4 = (e - r * (DH / 4)) * 4;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    st4(xs + r * LDX + c4, r < L ? ld4(x + base + (int64_t)r * d + c4) : z4);
    if (WITH_DA) st4(das + r * LDX + c4, r < L ? ld4(dA + base + (int64_t)r * d + c4) : z4);
  }
}

// tile  D[row = 16 ta + (4 q + r)][col = block of bf] = sum_k  a[16 ta + i][k] * bf[i][k]   (a from LDS, bf = fragments)
template <int DH>
__device__ __forceinline__ f32x4 sr_tile_nt(const float* arow, const float4 (&bf)[DH / 16]) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int f = 0; f < DH / 16; ++f) {
    const float4 a4 = ld4(arow + 4 * f);
    acc = mfma16(a4.x, bf[f].x, acc);
    acc = mfma16(a4.y, bf[f].y, acc);
    acc = mfma16(a4.z, bf[f].z, acc);
    acc = mfma16(a4.w, bf[f].w, acc);
  }
  return acc;
}

template <int DH>
__global__ __launch_bounds__(SR_MAXT * 64) void sr_attn_fwd_mfma_kernel(const float* __restrict__ x, int L, int d,
                                                                       float* __restrict__ A, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int KQ = DH / 4, LDX = DH + 4, NC = DH / 16, NF = DH / 16;
  const int Lp = round_up(L, 16), NTL = Lp / 16;
  float* xs = smem;  // [Lp][LDX]
  const int b = blockIdx.x, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
  const int64_t base = (int64_t)b * L * d + h * DH;
  sr_stage_head<DH, false>(x, nullptr, base, L, Lp, d, xs, nullptr, tid, NTL * 64);
  __syncthreads();
  const float scale = 1.0f / sqrtf((float)DH);
  float4 bq[NF];  // this wave's query block
#pragma unroll
  for (int f = 0; f < NF; ++f) bq[f] = ld4(xs + (wave * 16 + i) * LDX + q * KQ + 4 * f);
  f32x4 pr[SR_MAXT];  // pr[t][r]: key 16 t + 4 q + r, query = this lane's i
  float mx = -INFINITY;  // of the RAW scores (scale > 0)
#pragma unroll
  for (int t = 0; t < SR_MAXT; ++t) {
    if (t < NTL) {
      pr[t] = sr_tile_nt<DH>(xs + (t * 16 + i) * LDX + q * KQ, bq);
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, pr[t][r]);  // padding keys score 0 <= the diagonal |x_i|^2: never the maximum
    }
  }
  mx = quad_max(mx);
  // exp(scale * (s - mx)) = 2^(s * c1 - mx * c1): one fma + v_exp_f32 per probability (VALU work does not hide behind MFMAs)
  const float c1 = scale * 1.44269504088896341f, mx2 = mx * c1;
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < SR_MAXT; ++t) {
    if (t < NTL) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pr[t][r] = __builtin_amdgcn_exp2f(fmaf(pr[t][r], c1, -mx2));
      if (t == NTL - 1) {  // only the last block can hold padding keys
#pragma unroll
        for (int r = 0; r < 4; ++r) pr[t][r] = (t * 16 + 4 * q + r < L) ? pr[t][r] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) sum += pr[t][r];
    }
  }
  sum = quad_sum(sum);
  const float inv = 1.0f / sum;
  // row statistic for the backward: P[i][j] = exp(S[i][j] - lse[i])
  if (lse != nullptr && q == 0 && wave * 16 + i < L) lse[((int64_t)b * L + wave * 16 + i) * gridDim.y + h] = mx * scale + logf(sum);
  f32x4 o[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) o[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < SR_MAXT; ++t) {
    if (t < NTL) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = pr[t][r] * inv;
        const float* vr = xs + (t * 16 + 4 * q + r) * LDX + i;
#pragma unroll
        for (int c = 0; c < NC; ++c) o[c] = mfma16(pv, vr[c * 16], o[c]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wave * 16 + 4 * q + r;
      if (row < L) A[base + (int64_t)row * d + c * 16 + i] = o[c][r];
    }
  }
}

// backward.  With P and dP = dA x^T,  dS = scale * P o (dP - t),  t = rowsum(P o dP):   dx += dS x + dS^T x + P^T dA.
// Flash-attention bookkeeping removes every row reduction from the kernel: the forward left lse[i] (P = exp(S - lse)),
// and t[i] = rowsum(P o dP) = dA[i] . A[i] (A = P x is the saved forward output) is a 32-float dot product formed
// while the tiles are staged.  A wave owns a block of 16 tokens in BOTH roles and walks the 7 partner blocks once:
//  * ONE score tile serves both roles - S = x x^T is symmetric, so the register that holds S[key 16 t + 4 q + r][query i]
//    for the wave's query i also is S[query 16 t + 4 q + r][key i] for its key i;
//  * as QUERIES it needs dP^T[key][query] = x[key] . dA[query]  ->  P, dS for its query in 4 lanes = the A operand of
//    dq = dS x (contraction over the partner block's keys);
//  * as KEYS it needs dP[query][key] = dA[query] . x[key] and the partner queries' lse / t  ->  P, dS columns = the A
//    operand of dk = dS^T x and dv = P^T dA (contraction over the partner block's queries).
// 48 MFMAs per partner block (3 x 8 tiles + 3 x 8 products), no softmax pass, one barrier (after staging), nothing
// [L, L]-sized anywhere; the three products land on the same output tile (separate accumulator chains, summed at the end).
template <int DH>
__global__ __launch_bounds__(SR_MAXT * 64) __attribute__((amdgpu_waves_per_eu(4, 8))) void sr_attn_bwd_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ dA, const float* __restrict__ Aout, const float* __restrict__ lse, int L,
    int d, float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int KQ = DH / 4, LDX = DH + 4, NC = DH / 16, NF = DH / 16, RT = DH / 4;  // RT threads stage one row
  const int Lp = round_up(L, 16), NTL = Lp / 16;
  float* xs = smem;               // [Lp][LDX]
  float* das = xs + Lp * LDX;     // [Lp][LDX]
  float* st_l = das + Lp * LDX;   // [Lp] lse * log2(e) (padding rows: +inf -> P = 0)
  float* st_t = st_l + Lp;        // [Lp] t = dA . A
  const int b = blockIdx.x, h = blockIdx.y, H = gridDim.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
  const int64_t base = (int64_t)b * L * d + h * DH;
  for (int e = tid; e < Lp * RT; e += NTL * 64) {
    const int r = e / RT, c4 = (e - r * RT) * 4;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool ok = r < L;
    const float4 xv = ok ? ld4(x + base + (int64_t)r * d + c4) : z4;
    const float4 gv = ok ? ld4(dA + base + (int64_t)r * d + c4) : z4;
    const float4 av = ok ? ld4(Aout + base + (int64_t)r * d + c4) : z4;
    st4(xs + r * LDX + c4, xv);
    st4(das + r * LDX + c4, gv);
    // the RT threads of a row are consecutive lanes of one wave (RT = 4 / 8 / 16 divides 64): butterfly over them
    float tt = gv.x * av.x + gv.y * av.y + gv.z * av.z + gv.w * av.w;
#pragma unroll
    for (int m = 1; m < RT; m <<= 1) tt += __shfl_xor(tt, m);
    if ((e - r * RT) == 0) {
      st_t[r] = tt;
      st_l[r] = ok ? lse[((int64_t)b * L + r) * H + h] * 1.44269504088896341f : INFINITY;
    }
  }
  __syncthreads();
  const float scale = 1.0f / sqrtf((float)DH);
  float4 bx[NF], bg[NF];  // this wave's block: x rows and dA rows
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    bx[f] = ld4(xs + (wave * 16 + i) * LDX + q * KQ + 4 * f);
    bg[f] = ld4(das + (wave * 16 + i) * LDX + q * KQ + 4 * f);
  }
  // VALU instructions do not hide behind the MFMAs here (SQ_VALU_MFMA_COEXEC_CYCLES = 0, and MFMA busy + 4 x VALU count
  // adds up to the kernel time), so the probabilities cost two instructions each: P = 2^(S * scale*log2(e) - lse*log2(e))
  // as one fma + v_exp_f32 (st_l holds lse * log2(e); an unreduced argument costs ~|arg| * 4e-8 of relative accuracy).
  // A padding QUERY has lse = +inf -> P = 0 by itself; padding KEYS exist only in the last block and are masked there.
  const float c1 = scale * 1.44269504088896341f;
  const float l_own = st_l[wave * 16 + i], t_own = st_t[wave * 16 + i];
  f32x4 acc[NC], acv[NC];  // dq + dk, dv: separate chains, summed at the end
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = acv[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto partner_block = [&](int t, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    const float* xrow = xs + (t * 16 + i) * LDX + q * KQ;
    const float* grow = das + (t * 16 + i) * LDX + q * KQ;
    // three independent accumulation chains, interleaved:
    //   sc  = S[16 t + 4 q + r][own i]  (symmetric: serves both roles)
    //   dpt = dP^T[key 16 t + 4 q + r][query own i] = x[key] . dA[query]
    //   dpn = dP[query 16 t + 4 q + r][key own i]   = dA[query] . x[key]
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dpt = sc, dpn = sc;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const float4 xa = ld4(xrow + 4 * f), ga = ld4(grow + 4 * f);
      sc = mfma16(xa.x, bx[f].x, sc); dpt = mfma16(xa.x, bg[f].x, dpt); dpn = mfma16(ga.x, bx[f].x, dpn);
      sc = mfma16(xa.y, bx[f].y, sc); dpt = mfma16(xa.y, bg[f].y, dpt); dpn = mfma16(ga.y, bx[f].y, dpn);
      sc = mfma16(xa.z, bx[f].z, sc); dpt = mfma16(xa.z, bg[f].z, dpt); dpn = mfma16(ga.z, bx[f].z, dpn);
      sc = mfma16(xa.w, bx[f].w, sc); dpt = mfma16(xa.w, bg[f].w, dpt); dpn = mfma16(ga.w, bx[f].w, dpn);
    }
    const float4 l4 = ld4(st_l + t * 16 + 4 * q), t4 = ld4(st_t + t * 16 + 4 * q);
    const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, tq[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int pr = t * 16 + 4 * q + r;   // the partner token of this register
      // own token as query, partner as key
      float pa = __builtin_amdgcn_exp2f(fmaf(sc[r], c1, -l_own));
      if (LAST) pa = (pr < L) ? pa : 0.f;
      const float dsa = (scale * pa) * (dpt[r] - t_own);
      // own token as key, partner as query
      const float pb = __builtin_amdgcn_exp2f(fmaf(sc[r], c1, -lq[r]));
      const float dsab = dsa + (scale * pb) * (dpn[r] - tq[r]);
      const float* xr = xs + pr * LDX + i;
      const float* gr = das + pr * LDX + i;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const float xv = xr[c * 16];
        acc[c] = mfma16(dsab, xv, acc[c]);           // dq + dk: (dS[own][partner] + dS[partner][own]) x[partner]  (q = k = x)
        acv[c] = mfma16(pb, gr[c * 16], acv[c]);     // dv: P[partner][own] dA[partner]
      }
    }
  };
  for (int t = 0; t < NTL - 1; ++t) partner_block(t, std::false_type{});
  partner_block(NTL - 1, std::true_type{});
#pragma unroll
  for (int c = 0; c < NC; ++c) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wave * 16 + 4 * q + r;
      if (row < L) dx[base + (int64_t)row * d + c * 16 + i] += acc[c][r] + acv[c][r];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// the same attention with fp16 operands on v_mfma_f32_16x16x32_f16 (opt-in: ultr_setrank_desc::attention_dtype = 1)
// ---------------------------------------------------------------------------------------------------------
// BASELINE config 5 names "fp16 MFMA attention over the list".  The fp32 kernels above are bound by matrix-core ISSUE
// (fp32 MFMA runs at 1/16 of the fp16 rate: 251 of the backward's 388 us are MFMA issue slots).  Here every product takes
// fp16 operands with fp32 accumulation - one instruction contracts 32 indices instead of 4 - and everything else
// (scores, softmax, lse, t, the dS algebra, outputs) stays fp32 in registers, exactly as above.  What changes:
//  * the head slice is staged TWICE in fp16: row-major [token][DH] (8 consecutive halves of a token = the A / B operand
//    of the NT tiles S^T, dP^T, dP over the head depth) and transposed [DH][token] (4 consecutive tokens of a column = half
//    of the B operand of the products that contract over tokens: P V, dS x, dS^T x, P^T dA);
//  * those products contract over a PAIR of 16-token blocks per instruction: the lane already holds, for its own token,
//    the 4 + 4 probabilities / dS values of partner tokens {16 t + 4 q + r} and {16 (t + 1) + 4 q + r} - converted to
//    halves they ARE the 8-element A operand (contraction index k = 8 q + j <-> token 16 (t + j / 4) + 4 q + j % 4; the B
//    operand follows the same permutation: two 8-byte LDS reads of the transposed slice).
// Numerics: operands rounded to fp16 (2^-11 relative), so scores / gradients agree with the fp32 path to ~1e-3 relative -
// an ORDERING-level parity (tests/test_gpu_setrank.py: identical top-10 on >= 99 % of lists, NDCG@10 within 1e-3), not
// the 1e-5 bar; that is why it is opt-in.  Head depth 32 or 64 (one or two contraction chunks), list_size <= 128.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma_h(h8 a, h8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ h8 pack_h8(const f32x4& lo, const f32x4& hi) {
  h8 o;
  o[0] = (_Float16)lo[0]; o[1] = (_Float16)lo[1]; o[2] = (_Float16)lo[2]; o[3] = (_Float16)lo[3];
  o[4] = (_Float16)hi[0]; o[5] = (_Float16)hi[1]; o[6] = (_Float16)hi[2]; o[7] = (_Float16)hi[3];
  return o;
}
__device__ __forceinline__ h8 join_h4(h4 lo, h4 hi) {
  h8 o;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
  o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
  return o;
}
constexpr int SRH_LT = 16 * SR_MAXT + 16 + 8;  // row length (halves) of the transposed slices: 9 blocks (the pair partner of an
                                               // odd last block reads zeros) + 8 halves of padding (272 B: 16-byte multiple)

// stage x (and dA) of one (list, head) as fp16: row-major xh [Lp][DH + 8] and transposed xt [DH][SRH_LT]; zero past L
template <int DH, bool WITH_G>
__device__ __forceinline__ void srh_stage(const float* __restrict__ x, const float* __restrict__ g, int64_t base, int L, int Lp, int d,
                                          _Float16* xh, _Float16* xt, _Float16* gh, _Float16* gt, int tid, int nthr) {
  constexpr int LDH = DH + 8;
  for (int e = tid; e < DH * (SRH_LT / 8); e += nthr) {  // zero the transposed slices (padding columns are read)
    const h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    *reinterpret_cast<h8*>(xt + e * 8) = z;
    if (WITH_G) *reinterpret_cast<h8*>(gt + e * 8) = z;
  }
  __syncthreads();
  for (int e = tid; e < Lp * (DH / 4); e += nthr) {
    const int r = e / (DH / 4), c4 = (e - r * (DH / 4)) * 4;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 xv = r < L ? ld4(x + base + (int64_t)r * d + c4) : z4;
    h4 hv = {(_Float16)xv.x, (_Float16)xv.y, (_Float16)xv.z, (_Float16)xv.w};
    *reinterpret_cast<h4*>(xh + r * LDH + c4) = hv;
    xt[(c4 + 0) * SRH_LT + r] = hv[0]; xt[(c4 + 1) * SRH_LT + r] = hv[1];
    xt[(c4 + 2) * SRH_LT + r] = hv[2]; xt[(c4 + 3) * SRH_LT + r] = hv[3];
    if (WITH_G) {
      const float4 gv = r < L ? ld4(g + base + (int64_t)r * d + c4) : z4;
      h4 hg = {(_Float16)gv.x, (_Float16)gv.y, (_Float16)gv.z, (_Float16)gv.w};
      *reinterpret_cast<h4*>(gh + r * LDH + c4) = hg;
      gt[(c4 + 0) * SRH_LT + r] = hg[0]; gt[(c4 + 1) * SRH_LT + r] = hg[1];
      gt[(c4 + 2) * SRH_LT + r] = hg[2]; gt[(c4 + 3) * SRH_LT + r] = hg[3];
    }
  }
}
// NT tile over the head depth: D[row 4 q + r][col i] = sum_k a[16 ta + i'][k] * bfrag[i][k]  (a rows from the row-major slice)
template <int DH>
__device__ __forceinline__ f32x4 srh_tile(const _Float16* arow, const h8 (&bf)[DH / 32]) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int f = 0; f < DH / 32; ++f) acc = mfma_h(*reinterpret_cast<const h8*>(arow + 32 * f), bf[f], acc);
  return acc;
}
// B operand of a token-contraction over the block pair (t, t + 1): column `col` of the transposed slice
__device__ __forceinline__ h8 srh_pair_b(const _Float16* tslice, int col, int t, int q) {
  const _Float16* p = tslice + col * SRH_LT + 16 * t + 4 * q;
  return join_h4(*reinterpret_cast<const h4*>(p), *reinterpret_cast<const h4*>(p + 16));
}

template <int DH>
__global__ __launch_bounds__(SR_MAXT * 64) void sr_attn_fwd_f16_kernel(const float* __restrict__ x, int L, int d,
                                                                      float* __restrict__ A, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LDH = DH + 8, NC = DH / 16, NF = DH / 32;
  const int Lp = round_up(L, 16), NTL = Lp / 16;
  _Float16* xh = reinterpret_cast<_Float16*>(smem);  // [Lp][LDH]
  _Float16* xt = xh + 16 * SR_MAXT * LDH;            // [DH][SRH_LT]
  const int b = blockIdx.x, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
  const int64_t base = (int64_t)b * L * d + h * DH;
  srh_stage<DH, false>(x, nullptr, base, L, Lp, d, xh, xt, nullptr, nullptr, tid, NTL * 64);
  __syncthreads();
  const float scale = 1.0f / sqrtf((float)DH);
  h8 bq[NF];  // this wave's query block: lane (i, q) holds x[query i][32 f + 8 q .. + 7]
#pragma unroll
  for (int f = 0; f < NF; ++f) bq[f] = *reinterpret_cast<const h8*>(xh + (wave * 16 + i) * LDH + 32 * f + 8 * q);
  f32x4 pr[SR_MAXT + 1];  // pr[t][r]: key 16 t + 4 q + r, query = this lane's i   (+1: the zero partner of an odd last block)
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t <= SR_MAXT; ++t) pr[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < SR_MAXT; ++t) {
    if (t < NTL) {
      pr[t] = srh_tile<DH>(xh + (t * 16 + i) * LDH + 8 * q, bq);
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, pr[t][r]);
    }
  }
  mx = quad_max(mx);
  const float c1 = scale * 1.44269504088896341f, mx2 = mx * c1;
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < SR_MAXT; ++t) {
    if (t < NTL) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pr[t][r] = __builtin_amdgcn_exp2f(fmaf(pr[t][r], c1, -mx2));
      if (t == NTL - 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) pr[t][r] = (t * 16 + 4 * q + r < L) ? pr[t][r] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) sum += pr[t][r];
    }
  }
  sum = quad_sum(sum);
  const float inv = 1.0f / sum;
  if (lse != nullptr && q == 0 && wave * 16 + i < L) lse[((int64_t)b * L + wave * 16 + i) * gridDim.y + h] = mx * scale + logf(sum);
  f32x4 o[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) o[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < SR_MAXT; t += 2) {
    if (t < NTL) {
      f32x4 lo = pr[t], hi = pr[t + 1];
#pragma unroll
      for (int r = 0; r < 4; ++r) { lo[r] *= inv; hi[r] *= inv; }
      const h8 pa = pack_h8(lo, hi);
#pragma unroll
      for (int c = 0; c < NC; ++c) o[c] = mfma_h(pa, srh_pair_b(xt, 16 * c + i, t, q), o[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wave * 16 + 4 * q + r;
      if (row < L) A[base + (int64_t)row * d + c * 16 + i] = o[c][r];
    }
  }
}

template <int DH>
__global__ __launch_bounds__(SR_MAXT * 64) void sr_attn_bwd_f16_kernel(const float* __restrict__ x, const float* __restrict__ dA,
                                                                      const float* __restrict__ Aout, const float* __restrict__ lse,
                                                                      int L, int d, float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LDH = DH + 8, NC = DH / 16, NF = DH / 32, RT = DH / 4;
  const int Lp = round_up(L, 16), NTL = Lp / 16;
  _Float16* xh = reinterpret_cast<_Float16*>(smem);   // [16 SR_MAXT][LDH]
  _Float16* gh = xh + 16 * SR_MAXT * LDH;
  _Float16* xt = gh + 16 * SR_MAXT * LDH;             // [DH][SRH_LT]
  _Float16* gt = xt + DH * SRH_LT;
  float* st_l = reinterpret_cast<float*>(gt + DH * SRH_LT);  // [16 (SR_MAXT + 1)] lse * log2(e) (+inf past L: P = 0)
  float* st_t = st_l + 16 * (SR_MAXT + 1);                   // [16 (SR_MAXT + 1)] t = dA . A
  const int b = blockIdx.x, h = blockIdx.y, H = gridDim.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
  const int64_t base = (int64_t)b * L * d + h * DH;
  srh_stage<DH, true>(x, dA, base, L, Lp, d, xh, xt, gh, gt, tid, NTL * 64);
  for (int e = tid; e < 16 * (SR_MAXT + 1) * RT; e += NTL * 64) {  // t = dA . A in fp32 from the fp32 inputs; lse
    const int r = e / RT, c4 = (e - r * RT) * 4;
    const bool ok = r < L;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 gv = ok ? ld4(dA + base + (int64_t)r * d + c4) : z4;
    const float4 av = ok ? ld4(Aout + base + (int64_t)r * d + c4) : z4;
    float tt = gv.x * av.x + gv.y * av.y + gv.z * av.z + gv.w * av.w;
#pragma unroll
    for (int m = 1; m < RT; m <<= 1) tt += __shfl_xor(tt, m);
    if ((e - r * RT) == 0) {
      st_t[r] = tt;
      st_l[r] = ok ? lse[((int64_t)b * L + r) * H + h] * 1.44269504088896341f : INFINITY;
    }
  }
  __syncthreads();
  const float scale = 1.0f / sqrtf((float)DH);
  h8 bx[NF], bg[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    bx[f] = *reinterpret_cast<const h8*>(xh + (wave * 16 + i) * LDH + 32 * f + 8 * q);
    bg[f] = *reinterpret_cast<const h8*>(gh + (wave * 16 + i) * LDH + 32 * f + 8 * q);
  }
  const float c1 = scale * 1.44269504088896341f;
  const float l_own = st_l[wave * 16 + i], t_own = st_t[wave * 16 + i];
  f32x4 acc[NC], acv[NC];  // dq + dk, dv
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = acv[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  // one partner block: the three NT tiles and the element-wise dS algebra; results in (dsa, dsb, pb) for tokens 16 t + 4 q + r
  auto block_vals = [&](int t, bool live, f32x4& dsa, f32x4& dsb, f32x4& pb) {
    dsa = dsb = pb = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!live) return;
    const _Float16* xrow = xh + (t * 16 + i) * LDH + 8 * q;
    const _Float16* grow = gh + (t * 16 + i) * LDH + 8 * q;
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dpt = sc, dpn = sc;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const h8 xa = *reinterpret_cast<const h8*>(xrow + 32 * f), ga = *reinterpret_cast<const h8*>(grow + 32 * f);
      sc = mfma_h(xa, bx[f], sc);    // S[16 t + 4 q + r][own i]  (symmetric: both roles)
      dpt = mfma_h(xa, bg[f], dpt);  // dP^T[key 16 t + 4 q + r][query own i] = x[key] . dA[query]
      dpn = mfma_h(ga, bx[f], dpn);  // dP[query 16 t + 4 q + r][key own i]   = dA[query] . x[key]
    }
    const float4 l4 = ld4(st_l + t * 16 + 4 * q), t4 = ld4(st_t + t * 16 + 4 * q);
    const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, tq[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int prt = t * 16 + 4 * q + r;
      float pa = __builtin_amdgcn_exp2f(fmaf(sc[r], c1, -l_own));  // own token as query, partner as key
      pa = (prt < L) ? pa : 0.f;                                    // padding keys (only the last block has any)
      dsa[r] = (scale * pa) * (dpt[r] - t_own);
      const float pbv = __builtin_amdgcn_exp2f(fmaf(sc[r], c1, -lq[r]));  // own token as key, partner as query (lse = +inf past L)
      pb[r] = pbv;
      dsb[r] = (scale * pbv) * (dpn[r] - tq[r]);
    }
  };
  for (int t = 0; t < NTL; t += 2) {
    f32x4 a0, b0, p0, a1, b1, p1;
    block_vals(t, true, a0, b0, p0);
    block_vals(t + 1, t + 1 < NTL, a1, b1, p1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // q = k = x: dq and dk share the x[partner] operand - summed in fp32 before the conversion
      a0[r] += b0[r];
      a1[r] += b1[r];
    }
    const h8 ha = pack_h8(a0, a1), hp = pack_h8(p0, p1);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const h8 xb = srh_pair_b(xt, 16 * c + i, t, q), gb = srh_pair_b(gt, 16 * c + i, t, q);
      acc[c] = mfma_h(ha, xb, acc[c]);  // dq + dk: (dS[own][partner] + dS[partner][own]) x[partner]
      acv[c] = mfma_h(hp, gb, acv[c]);  // dv: P[partner][own] dA[partner]
    }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wave * 16 + 4 * q + r;
      if (row < L) dx[base + (int64_t)row * d + c * 16 + i] += acc[c][r] + acv[c][r];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Split-half attention BACKWARD (round 4; default, ULTR_SR_ATTN_H3=0 for the fp32 matrix cores): the fp16-operand kernel above with every operand as hi + lo
// fp16 planes and every product as three f16 MFMAs with fp32 accumulation.  Scales are powers
// of two per (list, head) slice: x to just below 2^10, dA to just below 2^4 (dS = P (dP - t) / sqrt(dh) stays far below fp16's 65504
// for |x| up to several hundred), probabilities by 2^12; all are removed exactly in the epilogues.
// WHY THERE IS NO SPLIT-HALF FORWARD KERNEL (one was written and measured in round 4, removed in round 5; git 9876d59 has it).  The operands are exact to 2^-33 this way, but the instruction is not an fp32 dot product: inside
// v_mfma_f32_16x16x32_f16 the 32 products are aligned to the LARGEST of them and truncated ~25 bits below it before they are added
// (tools/mfma_f16_accum_test.hip: 2^20 + 31 x 0.111 comes out 0.44 low, 3.5 fp32 ulps, and the result depends on where the large
// product sits).  A dot product with one dominant term therefore carries a biased error of up to ~2^-20 of that term.  The DNN's
// products never notice (LayerNorm rows, 1e-6 on scores at every BASELINE config); attention does: a token with |x|^2 ~ 150 in a
// head has a logit of ~26 and its softmax row is a difference of such logits - config 5 at full size: ONE score in each of ~14 of
// 1024 lists moves by 5e-6 .. 3e-5 against the fp32 matrix-core kernels (everything else agrees to 5e-7), the permutation test
// of tests/test_gpu_setrank.py (2e-5) fails.  The BACKWARD kernel has no such amplifier behind it: its P is recomputed from the same
// truncated S, but the error enters dS, dq, dk, dv linearly and is summed over 102 400 tokens into the weight gradients - 1.8e-7 of
// the largest gradient entry at config 5 (bar: 1e-5 relative + 2e-6 of the largest), scores bit for bit those of the fp32 forward.
// Config 5: 3 020 -> 2 866 us with the backward kernel (383 -> 309 us per launch); a forward kernel on top gave 2 845 and failed the bar above.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float srs_pow2_scale(float amax, int target_exp) {  // amax * scale < 2^target_exp
  int se = 253 + target_exp - (int)((__float_as_uint(amax) >> 23) & 0xffu);
  se = se < 1 ? 1 : (se > 253 ? 253 : se);
  return __uint_as_float((unsigned)se << 23);
}
template <int DH>
__device__ __forceinline__ void srs_load(const float* __restrict__ src, int64_t base, int L, int d, int tid, int nthr, float4 (&v)[DH / 16],
                                         float& amax) {
#pragma unroll
  for (int k = 0; k < DH / 16; ++k) {
    const int e = tid + nthr * k;
    const int r = e / (DH / 4), c4 = (e - r * (DH / 4)) * 4;
    v[k] = r < L ? ld4(src + base + (int64_t)r * d + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[k].x), fabsf(v[k].y)), fmaxf(fabsf(v[k].z), fabsf(v[k].w))));
  }
}
// row-major planes rh / rl [Lp][DH + 8] and transposed planes th / tl [DH][lt] of the scaled values
// (r3 != nullptr: a THIRD row-major plane, the fp16 of what hi + lo leave - 33 bits for the operands of S = x x^T, whose error the
// softmax exponentiates: with two pieces the scores of config 5 moved by up to 4e-5 under a permutation of the documents, 2e-7 with three)
template <int DH>
__device__ __forceinline__ void srs_store(const float4 (&v)[DH / 16], float scale, int tid, int nthr, _Float16* rh, _Float16* rl,
                                          _Float16* th, _Float16* tl, int lt, _Float16* r3 = nullptr, _Float16* t3 = nullptr) {
  constexpr int LDH = DH + 8;
#pragma unroll
  for (int k = 0; k < DH / 16; ++k) {
    const int e = tid + nthr * k;
    const int r = e / (DH / 4), c4 = (e - r * (DH / 4)) * 4;
    const float a[4] = {v[k].x * scale, v[k].y * scale, v[k].z * scale, v[k].w * scale};
    h4 hi, lo, l3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      hi[j] = (_Float16)a[j];
      float r1 = a[j] - (float)hi[j];
      asm volatile("" : "+v"(r1));  // keeps hipcc from fusing the subtraction and the conversion into v_fma_mix*_f16
      lo[j] = (_Float16)r1;
      float r2 = r1 - (float)lo[j];
      asm volatile("" : "+v"(r2));
      l3[j] = (_Float16)r2;
    }
    *reinterpret_cast<h4*>(rh + r * LDH + c4) = hi;
    *reinterpret_cast<h4*>(rl + r * LDH + c4) = lo;
    if (r3 != nullptr) *reinterpret_cast<h4*>(r3 + r * LDH + c4) = l3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      th[(c4 + j) * lt + r] = hi[j];
      tl[(c4 + j) * lt + r] = lo[j];
      if (t3 != nullptr) t3[(c4 + j) * lt + r] = l3[j];
    }
  }
}
__device__ __forceinline__ h8 srs_pair_b(const _Float16* tslice, int lt, int col, int t, int q) {
  const _Float16* p = tslice + col * lt + 16 * t + 4 * q;
  return join_h4(*reinterpret_cast<const h4*>(p), *reinterpret_cast<const h4*>(p + 16));
}
// the 4 + 4 values of a block pair times `scale` as hi / lo halves
__device__ __forceinline__ void srs_pack(const f32x4& a, const f32x4& b, float scale, h8& hi, h8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = (j < 4 ? a[j & 3] : b[j & 3]) * scale;
    const _Float16 h = (_Float16)v;
    hi[j] = h;
    float r1 = v - (float)h;
    asm volatile("" : "+v"(r1));
    lo[j] = (_Float16)r1;
  }
}
// NT tile over the head depth with split operands: sum_k a[k] b[k], a = row of the row-major planes, b = a held fragment
template <int DH>
__device__ __forceinline__ f32x4 srs_tile(const _Float16* ah_row, const _Float16* al_row, const h8 (&bh)[DH / 32], const h8 (&bl)[DH / 32]) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int f = 0; f < DH / 32; ++f) {
    const h8 ah = *reinterpret_cast<const h8*>(ah_row + 32 * f), al = *reinterpret_cast<const h8*>(al_row + 32 * f);
    acc = mfma_h(ah, bl[f], acc);
    acc = mfma_h(al, bh[f], acc);
    acc = mfma_h(ah, bh[f], acc);
  }
  return acc;
}



template <int DH>
__global__ __launch_bounds__(SR_MAXT * 64) void sr_attn_bwd_h3_kernel(const float* __restrict__ x, const float* __restrict__ dA,
                                                                     const float* __restrict__ Aout, const float* __restrict__ lse,
                                                                     int L, int d, float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LDH = DH + 8, NC = DH / 16, NF = DH / 32, RT = DH / 4;
  const int Lp = round_up(L, 16), NTL = Lp / 16, lt = 16 * NTL + 24;
  _Float16* xh = reinterpret_cast<_Float16*>(smem);  // x: [Lp][LDH] hi, lo
  _Float16* xl = xh + Lp * LDH;
  _Float16* gh = xl + Lp * LDH;                      // dA
  _Float16* gl = gh + Lp * LDH;
  _Float16* xth = gl + Lp * LDH;                     // transposed [DH][lt]: x hi, lo, dA hi, lo
  _Float16* xtl = xth + DH * lt;
  _Float16* gth = xtl + DH * lt;
  _Float16* gtl = gth + DH * lt;
  float* st_l = reinterpret_cast<float*>(gtl + DH * lt);  // [16 (SR_MAXT + 1)] lse * log2(e) (+inf past L: P = 0)
  float* st_t = st_l + 16 * (SR_MAXT + 1);                // [16 (SR_MAXT + 1)] t = dA . A
  float* red = st_t + 16 * (SR_MAXT + 1);                 // [2][SR_MAXT]
  const int b = blockIdx.x, h = blockIdx.y, H = gridDim.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
  const int nthr = NTL * 64;
  const int64_t base = (int64_t)b * L * d + h * DH;
  for (int e = tid; e < 4 * DH * (lt / 8); e += nthr) {  // zero the four transposed planes
    const h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    *reinterpret_cast<h8*>(xth + e * 8) = z;
  }
  float4 xv[DH / 16], gv[DH / 16];
  float amx = 0.f, amg = 0.f;
  srs_load<DH>(x, base, L, d, tid, nthr, xv, amx);
  srs_load<DH>(dA, base, L, d, tid, nthr, gv, amg);
  amx = wave_max(amx);
  amg = wave_max(amg);
  if (lane == 0) {
    red[wave] = amx;
    red[SR_MAXT + wave] = amg;
  }
  // t = dA . A in fp32 from the fp32 inputs (the dA values this thread holds for the planes; the 8 (16) threads of a row are consecutive
  // lanes), and the forward's lse; rows past L: P = 0 through lse = +inf
#pragma unroll
  for (int k = 0; k < DH / 16; ++k) {
    const int e = tid + nthr * k;
    const int r = e / RT, c4 = (e - r * RT) * 4;
    const bool ok = r < L;
    const float4 av = ok ? ld4(Aout + base + (int64_t)r * d + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    float tt = gv[k].x * av.x + gv[k].y * av.y + gv[k].z * av.z + gv[k].w * av.w;
    tt += dpp_or<0xb1>(0.f, tt);
    tt += dpp_or<0x4e>(0.f, tt);
    tt += dpp_or<0x141>(0.f, tt);  // row_half_mirror: the other four lanes of the group of eight
    if (RT == 16) tt += dpp_or<0x140>(0.f, tt);  // row_mirror: the other eight of sixteen
    if ((e - r * RT) == 0) {
      st_t[r] = tt;
      st_l[r] = ok ? lse[((int64_t)b * L + r) * H + h] * 1.44269504088896341f : INFINITY;
    }
  }
  __syncthreads();
  amx = amg = 0.f;
  for (int w = 0; w < NTL; ++w) {
    amx = fmaxf(amx, red[w]);
    amg = fmaxf(amg, red[SR_MAXT + w]);
  }
  const float sx = srs_pow2_scale(amx, 10), isx = 1.0f / sx;
  const float sg = srs_pow2_scale(amg, 4), isg = 1.0f / sg;
  srs_store<DH>(xv, sx, tid, nthr, xh, xl, xth, xtl, lt);
  srs_store<DH>(gv, sg, tid, nthr, gh, gl, gth, gtl, lt);
  __syncthreads();
  const float scale = 1.0f / sqrtf((float)DH);
  h8 bxh[NF], bxl[NF], bgh[NF], bgl[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int o = (wave * 16 + i) * LDH + 32 * f + 8 * q;
    bxh[f] = *reinterpret_cast<const h8*>(xh + o);
    bxl[f] = *reinterpret_cast<const h8*>(xl + o);
    bgh[f] = *reinterpret_cast<const h8*>(gh + o);
    bgl[f] = *reinterpret_cast<const h8*>(gl + o);
  }
  const float c1 = (scale * isx) * isx * 1.44269504088896341f;  // logits * log2(e) from sx^2 S
  const float l_own = st_l[wave * 16 + i], t_own = st_t[wave * 16 + i] * sg;
  f32x4 acc[NC], acv[NC];  // sx sg (dq + dk), 4096 sg dv
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = acv[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  // one partner block: the three NT tiles and the element-wise dS algebra; results (sg dS[own][partner], sg dS[partner][own],
  // P[partner][own]) for tokens 16 t + 4 q + r
  auto block_vals = [&](int t, bool live, f32x4& dsa, f32x4& dsb, f32x4& pb) {
    dsa = dsb = pb = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!live) return;
    const int o = (t * 16 + i) * LDH + 8 * q;
    const f32x4 sc = srs_tile<DH>(xh + o, xl + o, bxh, bxl);   // sx^2 S[16 t + 4 q + r][own i]  (symmetric: both roles; two pieces: a third changes nothing measurable)
    const f32x4 dpt = srs_tile<DH>(xh + o, xl + o, bgh, bgl);  // sx sg dP^T[key 16 t + 4 q + r][query own i] = x[key] . dA[query]
    const f32x4 dpn = srs_tile<DH>(gh + o, gl + o, bxh, bxl);  // sx sg dP[query 16 t + 4 q + r][key own i]   = dA[query] . x[key]
    const float4 l4 = ld4(st_l + t * 16 + 4 * q), t4 = ld4(st_t + t * 16 + 4 * q);
    const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, tq[4] = {t4.x * sg, t4.y * sg, t4.z * sg, t4.w * sg};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int prt = t * 16 + 4 * q + r;
      float pa = __builtin_amdgcn_exp2f(fmaf(sc[r], c1, -l_own));  // own token as query, partner as key
      pa = (prt < L) ? pa : 0.f;                                    // padding keys (only the last block has any)
      dsa[r] = (scale * pa) * (dpt[r] * isx - t_own);
      const float pbv = __builtin_amdgcn_exp2f(fmaf(sc[r], c1, -lq[r]));  // own token as key, partner as query (lse = +inf past L)
      pb[r] = pbv;
      dsb[r] = (scale * pbv) * (dpn[r] * isx - tq[r]);
    }
  };
  for (int t = 0; t < NTL; t += 2) {
    f32x4 a0, b0, p0, a1, b1, p1;
    block_vals(t, true, a0, b0, p0);
    block_vals(t + 1, t + 1 < NTL, a1, b1, p1);
    // q = k = x (SetRank.py:54-59): dq and dk contract with the SAME x[partner] rows, so dS[own][partner] + dS[partner][own] is
    // summed in fp32 BEFORE the split - one pack and three MFMAs per column tile less than two separate products
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      a0[r] += b0[r];
      a1[r] += b1[r];
    }
    h8 hah, hal, hph, hpl;
    srs_pack(a0, a1, 1.0f, hah, hal);
    srs_pack(p0, p1, 4096.0f, hph, hpl);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const h8 xbh = srs_pair_b(xth, lt, 16 * c + i, t, q), xbl = srs_pair_b(xtl, lt, 16 * c + i, t, q);
      const h8 gbh = srs_pair_b(gth, lt, 16 * c + i, t, q), gbl = srs_pair_b(gtl, lt, 16 * c + i, t, q);
      acc[c] = mfma_h(hah, xbl, acc[c]);  // dq + dk: (dS[own][partner] + dS[partner][own]) x[partner]
      acc[c] = mfma_h(hal, xbh, acc[c]);
      acc[c] = mfma_h(hah, xbh, acc[c]);
      acv[c] = mfma_h(hph, gbl, acv[c]);  // dv: P[partner][own] dA[partner]
      acv[c] = mfma_h(hpl, gbh, acv[c]);
      acv[c] = mfma_h(hph, gbh, acv[c]);
    }
  }
  const float s1 = isx * isg, s2 = isg * (1.0f / 4096.0f);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wave * 16 + 4 * q + r;
      if (row < L) dx[base + (int64_t)row * d + c * 16 + i] += acc[c][r] * s1 + acv[c][r] * s2;
    }
  }
}


#define SR_CHECK(call)        \
  do {                        \
    const int rc_ = (call);   \
    if (rc_ != 0) return rc_; \
  } while (0)

// matrix-core attention: list_size <= 128 (one wave per 16-token block), head depth 16 / 32 / 64, 16-byte aligned head
// slices; ULTR_SR_SCALAR_ATTN=1 forces the scalar kernels
bool attn_mfma_ok(const SrAttnShape& p, int L) {
  const char* v = getenv("ULTR_SR_SCALAR_ATTN");
  if (v != nullptr && v[0] == '1') return false;
  return L <= 16 * SR_MAXT && (p.dh == 16 || p.dh == 32 || p.dh == 64) && p.d % 4 == 0;
}
template <typename K>
int set_dyn_lds(K kernel, size_t bytes) {
  if (bytes > 160 * 1024) return ULTR_E_UNSUPPORTED;
  if (bytes > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
    return ULTR_E_UNSUPPORTED;
  return 0;
}
int attn_fwd_mfma(const SrAttnShape& p, const float* x, int batch, int L, float* A, float* lse, hipStream_t st) {
  const int Lp = round_up(L, 16);
  const size_t lds = (size_t)Lp * (p.dh + 4) * sizeof(float);
  const dim3 grid(batch, p.H), block(Lp * 4);  // one wave per 16 tokens
  if (p.dh == 16) hipLaunchKernelGGL(sr_attn_fwd_mfma_kernel<16>, grid, block, lds, st, x, L, p.d, A, lse);
  else if (p.dh == 32) hipLaunchKernelGGL(sr_attn_fwd_mfma_kernel<32>, grid, block, lds, st, x, L, p.d, A, lse);
  else hipLaunchKernelGGL(sr_attn_fwd_mfma_kernel<64>, grid, block, lds, st, x, L, p.d, A, lse);
  return 0;
}
// fp16-operand kernels: requested by the descriptor AND head depth 32 / 64, list_size <= 128 (otherwise the fp32 path runs)
bool attn_f16_ok(const SrAttnShape& p, int L) { return p.att_f16 && attn_mfma_ok(p, L) && (p.dh == 32 || p.dh == 64); }
int attn_fwd_f16(const SrAttnShape& p, const float* x, int batch, int L, float* A, float* lse, hipStream_t st) {
  const int Lp = round_up(L, 16);
  const size_t lds = ((size_t)16 * SR_MAXT * (p.dh + 8) + (size_t)p.dh * SRH_LT) * sizeof(_Float16);
  const dim3 grid(batch, p.H), block(Lp * 4);
  if (p.dh == 32) {
    SR_CHECK(set_dyn_lds(sr_attn_fwd_f16_kernel<32>, lds));
    hipLaunchKernelGGL(sr_attn_fwd_f16_kernel<32>, grid, block, lds, st, x, L, p.d, A, lse);
  } else {
    SR_CHECK(set_dyn_lds(sr_attn_fwd_f16_kernel<64>, lds));
    hipLaunchKernelGGL(sr_attn_fwd_f16_kernel<64>, grid, block, lds, st, x, L, p.d, A, lse);
  }
  return 0;
}
int attn_bwd_f16(const SrAttnShape& p, const float* x, const float* dA, const float* Aout, const float* lse, int batch, int L, float* dx,
                 hipStream_t st) {
  const int Lp = round_up(L, 16);
  const size_t lds = ((size_t)2 * 16 * SR_MAXT * (p.dh + 8) + (size_t)2 * p.dh * SRH_LT) * sizeof(_Float16) +
                     (size_t)2 * 16 * (SR_MAXT + 1) * sizeof(float);
  const dim3 grid(batch, p.H), block(Lp * 4);
  if (p.dh == 32) {
    SR_CHECK(set_dyn_lds(sr_attn_bwd_f16_kernel<32>, lds));
    hipLaunchKernelGGL(sr_attn_bwd_f16_kernel<32>, grid, block, lds, st, x, dA, Aout, lse, L, p.d, dx);
  } else {
    SR_CHECK(set_dyn_lds(sr_attn_bwd_f16_kernel<64>, lds));
    hipLaunchKernelGGL(sr_attn_bwd_f16_kernel<64>, grid, block, lds, st, x, dA, Aout, lse, L, p.d, dx);
  }
  return 0;
}
// split-half BACKWARD kernel for fp32 attention at head depth 32 / 64, list_size <= 128 (ULTR_SR_ATTN_H3=1, default: the forward and
// therefore every score stays on the fp32 matrix cores bit for bit; gradients move by 1.8e-7 of the largest entry at config 5); =0: fp32
// matrix cores everywhere.  LDS of a launch = four hi / lo plane pairs of the (list, head) slice + the per-wave dS scratch:
size_t attn_bwd_h3_lds(const SrAttnShape& p, int L) {
  const int Lp = round_up(L, 16), lt = Lp + 24;
  return ((size_t)4 * Lp * (p.dh + 8) + (size_t)4 * p.dh * lt) * sizeof(_Float16) + ((size_t)2 * 16 * (SR_MAXT + 1) + 2 * SR_MAXT) * sizeof(float);
}
bool attn_h3_ok(const SrAttnShape& p, int L) {
  // the launch must leave room for two workgroups per CU, give or take (config 5, head depth 32 at list size 100: 72 KB - two of 7 waves
  // each; head depth 64 up to list size 64: 83 KB - one, still ahead of the fp32 kernel; beyond that, 100 KB and more, the fp32
  // matrix-core kernel wins).  ONE expression for the gate and the launch.
  if (attn_bwd_h3_lds(p, L) > 84 * 1024) return false;
  return !p.att_f16 && attn_mfma_ok(p, L) && (p.dh == 32 || p.dh == 64);
}
int attn_bwd_h3(const SrAttnShape& p, const float* x, const float* dA, const float* Aout, const float* lse, int batch, int L, float* dx,
                hipStream_t st) {
  const int Lp = round_up(L, 16);
  const size_t lds = attn_bwd_h3_lds(p, L);
  const dim3 grid(batch, p.H), block(Lp * 4);
  if (p.dh == 32) {
    SR_CHECK(set_dyn_lds(sr_attn_bwd_h3_kernel<32>, lds));
    hipLaunchKernelGGL(sr_attn_bwd_h3_kernel<32>, grid, block, lds, st, x, dA, Aout, lse, L, p.d, dx);
  } else {
    SR_CHECK(set_dyn_lds(sr_attn_bwd_h3_kernel<64>, lds));
    hipLaunchKernelGGL(sr_attn_bwd_h3_kernel<64>, grid, block, lds, st, x, dA, Aout, lse, L, p.d, dx);
  }
  return 0;
}
int attn_bwd_mfma(const SrAttnShape& p, const float* x, const float* dA, const float* Aout, const float* lse, int batch, int L, float* dx,
                  hipStream_t st) {
  const int Lp = round_up(L, 16);
  const size_t lds = ((size_t)2 * Lp * (p.dh + 4) + 2 * (size_t)Lp) * sizeof(float);
  const dim3 grid(batch, p.H), block(Lp * 4);
  if (p.dh == 16) {
    SR_CHECK(set_dyn_lds(sr_attn_bwd_mfma_kernel<16>, lds));
    hipLaunchKernelGGL(sr_attn_bwd_mfma_kernel<16>, grid, block, lds, st, x, dA, Aout, lse, L, p.d, dx);
  } else if (p.dh == 32) {
    SR_CHECK(set_dyn_lds(sr_attn_bwd_mfma_kernel<32>, lds));
    hipLaunchKernelGGL(sr_attn_bwd_mfma_kernel<32>, grid, block, lds, st, x, dA, Aout, lse, L, p.d, dx);
  } else {
    SR_CHECK(set_dyn_lds(sr_attn_bwd_mfma_kernel<64>, lds));
    hipLaunchKernelGGL(sr_attn_bwd_mfma_kernel<64>, grid, block, lds, st, x, dA, Aout, lse, L, p.d, dx);
  }
  return 0;
}

}  // namespace

bool sr_attn_mfma_ok(const SrAttnShape& s, int L) { return attn_mfma_ok(s, L); }

int sr_attn_supported(const SrAttnShape& s, int L, int backward) {
  if (L > 64 * SR_KPL) return ULTR_E_UNSUPPORTED;
  if (backward && !attn_mfma_ok(s, L) && L > 120) return ULTR_E_UNSUPPORTED;  // the scalar attention backward keeps two [L, L] matrices in LDS
  return 0;
}

int sr_attn_forward(const SrAttnShape& s, const float* x, int batch, int L, float* A, float* lse, hipStream_t st) {
  if (attn_f16_ok(s, L)) return attn_fwd_f16(s, x, batch, L, A, lse, st);
  if (attn_mfma_ok(s, L)) return attn_fwd_mfma(s, x, batch, L, A, lse, st);
  const size_t lds = ((size_t)L * (s.dh + 1) + 4 * (size_t)L) * sizeof(float);
  SR_CHECK(set_dyn_lds(sr_attn_fwd_kernel, lds));
  hipLaunchKernelGGL(sr_attn_fwd_kernel, dim3(batch, s.H), dim3(256), lds, st, x, L, s.d, s.dh, A);
  return 0;
}

int sr_attn_backward(const SrAttnShape& s, bool split_half, const float* x, const float* dA, const float* Aout, const float* lse, int batch, int L,
                     float* dx, hipStream_t st) {
  if (attn_f16_ok(s, L)) return attn_bwd_f16(s, x, dA, Aout, lse, batch, L, dx, st);
  if (split_half && attn_h3_ok(s, L)) return attn_bwd_h3(s, x, dA, Aout, lse, batch, L, dx, st);
  if (attn_mfma_ok(s, L)) return attn_bwd_mfma(s, x, dA, Aout, lse, batch, L, dx, st);
  const size_t lds = ((size_t)2 * L * (s.dh + 1) + 2 * (size_t)L * L) * sizeof(float);
  SR_CHECK(set_dyn_lds(sr_attn_bwd_kernel, lds));
  hipLaunchKernelGGL(sr_attn_bwd_kernel, dim3(batch, s.H), dim3(256), lds, st, x, dA, L, s.d, s.dh, dx);
  return 0;
}
