// ultr_sr_fwd.hip - SetRank forward, everything of an encoder block behind the attention as ONE persistent launch (round 6).
//
// Reference: ultra/ranking_model/SetRank.py:92-111 (+ the output FFN :136, :153 on the last block):
//   s1 = x + (A Wd^T + bd),  out1 = LN1(s1),  f = relu(out1 Wf1^T + bf1),  s2 = out1 + (f Wf2^T + bf2),  x' = LN2(s2)
//   [last block: oh = relu(x' Wo1^T + bo1),  score = oh . wo2 + bo2]
// sr_block_fwd_kernel (round 5, ultr_setrank.hip) does the same with two 8-wave workgroups per compute unit and R <= 30 rows each: every
// workgroup streams the block's 384 KB of weight fragments for 30 rows (1.36 GB of L2 -> CU traffic per launch), and its 3 500 workgroups
// start and drain their HBM traffic in lock step (3.3 TB/s over the launch).  Here - the geometry of the backward's fused launches,
// ultr_sr_bwd.hip - ONE workgroup per compute unit walks over tiles of 60 rows: four 16-row MFMA tiles behind one weight stream per wave
// (PipeH3W<4, 2>: half the weight traffic per row), the next tile's A and x rows requested a tile ahead, parameters read once per launch.
// Saved tensors, statistics and scores are the ones sr_block_fwd_kernel writes (same layout; out1 only when the backward will read it).
// Shapes: d_model 256, dff 64; ultr_setrank_forward keeps sr_block_fwd_kernel for everything else (ULTR_SR_BLOCK knob).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ultr_hip.h"
#include "ultr_device.h"
#include "ultr_h3.h"
#include "ultr_plan.h"
#include "ultr_sr_bwd.h"
#include "ultr_sr_tiles.h"

#define SRF_EPS 1e-6f  // nn.LayerNorm(eps=1e-6) everywhere in SetRank.py (:100-101, :134)
#ifdef ULTR_TRACE
__device__ unsigned long long g_srf_trace[64 * 32];  // wave 0 of the first 64 workgroups, their second tile (tools/trace_sr_bwd.py)
#define SRF_STAMP(slot)                                                                                                          \
  do {                                                                                                                           \
    if (threadIdx.x == 0 && blockIdx.x < 64 && tile == (int)(blockIdx.x + gridDim.x)) g_srf_trace[blockIdx.x * 32 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
extern "C" int ultr_srf_trace_read(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_srf_trace), sizeof(unsigned long long) * 64 * 32);
}
#else
#define SRF_STAMP(slot) \
  do {                  \
  } while (0)
#endif

namespace {

// LayerNorm of the wave's eight rows (row wave + 8 k) from the fp32 rows in P1, all eight at once (the six dependent DPP steps of eight
// reductions interleave: with two waves per SIMD nothing else hides them - four steps of two rows took 9 - 13k cycles per tile, phase trace):
// statistics and (store_out) the output to `saved`; the output also returns to P1 (the residual of the next sum) and - planes - goes to P0
// as the per-row-scaled fp16 plane pair of the next product's operand
// DEFER: nothing leaves for global memory here - the statistics come back in st_m / st_r and the caller stores them later (the phase
// behind a request of the next tile's rows must not issue vector-memory instructions: they queue behind the loads that wait for HBM)
template <bool DEFER>
__device__ __forceinline__ void ln_fwd_rows(int wave, int lane, int R, int ld, float* P1, float* P0, float* OS, const float4 g4,
                                            const float4 b4, const Dst& dout, const Dst& dmean, const Dst& drstd, bool store_out, bool planes,
                                            float (&st_m)[8], float (&st_r)[8]) {
  constexpr int d = SR_BWD_D, NR = 8;
  constexpr int pr = 0;
  const int c = 4 * lane;
  float4 v[NR];
  float s[NR], qv[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int r = wave + NW * (NR * pr + k);
    v[k] = ld4(P1 + (r < R ? r : R) * ld + c);
    s[k] = (v[k].x + v[k].y) + (v[k].z + v[k].w);
  }
  wave_sum_n<NR>(s);
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    s[k] *= (1.0f / (float)d);
    v[k].x -= s[k]; v[k].y -= s[k]; v[k].z -= s[k]; v[k].w -= s[k];
    qv[k] = (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
  }
  wave_sum_n<NR>(qv);
  const unsigned l0 = lane == 0 ? 0u : ULTR_OOB;
  float am[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int r = wave + NW * (NR * pr + k);
    const float rstd = rsqrt_nr(qv[k] * (1.0f / (float)d) + SRF_EPS);  // v_rsq_f32 + one Newton step (the IEEE sequence is ~45 instructions per row)
    v[k] = make_float4(v[k].x * rstd * g4.x + b4.x, v[k].y * rstd * g4.y + b4.y, v[k].z * rstd * g4.z + b4.z, v[k].w * rstd * g4.w + b4.w);
    if constexpr (DEFER) {
      st_m[k] = s[k];
      st_r[k] = rstd;
    } else {
      buf_st4(dout, store_out ? (unsigned)c * 4u : ULTR_OOB, (unsigned)(r * d) * 4u, v[k]);
      buf_st1(dmean, l0, (unsigned)r * 4u, s[k]);
      buf_st1(drstd, l0, (unsigned)r * 4u, rstd);
    }
    am[k] = max4(v[k]);
  }
  if (!planes) return;
  wave_max_n<NR>(am);
  _Float16* AH = reinterpret_cast<_Float16*>(P0);
  _Float16* AL = AH + (R + 1) * ld;
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int r = wave + NW * (NR * pr + k), rc = r < R ? r : R;
    float sc, inv;
    fb_h3_scale(am[k], sc, inv);
    st4(P1 + rc * ld + c, v[k]);
    fbh4 hi, lo;
    fb_h3_split4(v[k], sc, hi, lo);
    *reinterpret_cast<fbh4*>(AH + rc * ld + c) = hi;
    *reinterpret_cast<fbh4*>(AL + rc * ld + c) = lo;
    if (lane == 0) OS[r] = inv * (1.0f / ULTR_H3_WSCALE);
  }
}

// a dff-wide product with ReLU: wave = (16-row tile wave >> 1, 32-column chunk wave & 1) over the whole contraction; raw x row scale to P2
// (fp32 rows), then - row ownership 4 (wave + 8 q2) + (lane >> 4), lane & 15 = a float4 of the row - bias, ReLU, the row to `saved` and either the
// per-row-scaled plane pair over P2 (the block's f) or the dot product with a vector (the scorer)
__device__ __forceinline__ void product_f_relu(int wave, int lane, int R, const float* P0, float* P2, float* OS, const _Float16* planes, int64_t gw,
                                               const float* bias, const Dst& dfo, const float* wdot, float bdot, const Dst& dsc) {
  constexpr int d = SR_BWD_D, dff = SR_BWD_DFF, ld = d + 8, ldf = dff + 8;
  {
    const int i = lane & 15, q = lane >> 4, rt = wave >> 1, chf = wave & 1;
    const _Float16* AH = reinterpret_cast<const _Float16*>(P0);
    const int rowi = 16 * rt + i;
    const _Float16* pa[1] = {AH + (rowi < R ? rowi : R) * ld + 8 * q};
    const Src Wh = make_src(reinterpret_cast<const float*>(planes + gw), (int64_t)d * dff);
    f32x4 accf[1][2] = {{(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}}};
    PipeH3W<1, 2> ph;
    ph.begin(Wh, chf, d >> 5, 0, d >> 5, true, lane);
    ph.run(pa, (R + 1) * ld, Wh, d >> 5, accf);
    const float4 o4 = ld4(OS + 16 * rt + 4 * q);
    const float o[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * rt + 4 * q + r, rc = row < R ? row : R;
      *reinterpret_cast<float2*>(P2 + rc * ldf + 32 * chf + 2 * i) = make_float2(accf[0][0][r] * o[r], accf[0][1][r] * o[r]);
    }
  }
  lds_barrier();
  const int cf = 4 * (lane & 15);
  const float4 b4 = ld4(bias + cf);
  float4 v[2];
  float am[2];
#pragma unroll
  for (int q2 = 0; q2 < 2; ++q2) {
    const int row = 4 * (wave + NW * q2) + (lane >> 4), rc = row < R ? row : R;
    v[q2] = ld4(P2 + rc * ldf + cf);
    v[q2] = make_float4(fmaxf(v[q2].x + b4.x, 0.f), fmaxf(v[q2].y + b4.y, 0.f), fmaxf(v[q2].z + b4.z, 0.f), fmaxf(v[q2].w + b4.w, 0.f));
    buf_st4(dfo, (unsigned)(row * dff + cf) * 4u, 0u, v[q2]);
  }
  if (wdot != nullptr) {
    const float4 w4 = ld4(wdot + cf);
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
      const int row = 4 * (wave + NW * q2) + (lane >> 4);
      float t = (v[q2].x * w4.x + v[q2].y * w4.y) + (v[q2].z * w4.z + v[q2].w * w4.w);
      t += dpp_or<0xb1>(0.f, t);
      t += dpp_or<0x4e>(0.f, t);
      t += dpp_or<0x124>(0.f, t);
      t += dpp_or<0x128>(0.f, t);  // every lane of the 16-lane row holds the row's dot product
      buf_st1(dsc, (lane & 15) == 0 ? (unsigned)row * 4u : ULTR_OOB, 0u, t + bdot);
    }
    return;
  }
#pragma unroll
  for (int q2 = 0; q2 < 2; ++q2) am[q2] = row16_max(fmaxf(fmaxf(v[q2].x, v[q2].y), fmaxf(v[q2].z, v[q2].w)));
  lds_barrier();  // every fp32 row has been read: the planes may overwrite them
  _Float16* FH = reinterpret_cast<_Float16*>(P2);
  _Float16* FL = FH + (R + 1) * ldf;
#pragma unroll
  for (int q2 = 0; q2 < 2; ++q2) {
    const int row = 4 * (wave + NW * q2) + (lane >> 4), rc = row < R ? row : R;
    float sc, inv;
    fb_h3_scale(am[q2], sc, inv);
    fbh4 hi, lo;
    fb_h3_split4(v[q2], sc, hi, lo);
    *reinterpret_cast<fbh4*>(FH + rc * ldf + cf) = hi;
    *reinterpret_cast<fbh4*>(FL + rc * ldf + cf) = lo;
    if ((lane & 15) == 0) OS[64 + row] = inv * (1.0f / ULTR_H3_WSCALE);
  }
}

__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void sr_fwd_block_kernel(SrFwdBlockArgs a, const float* __restrict__ params,
                                                                                               const _Float16* __restrict__ planes,
                                                                                               float* __restrict__ sv, float* __restrict__ scores) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int d = SR_BWD_D, dff = SR_BWD_DFF, ld = d + 8, ldf = dff + 8;
  const int R = a.R;
  float* P0 = smem + a.p0;  // planes of the d-wide operand: A, then out1, then (last block) x'
  float* P1 = smem + a.p1;  // fp32 rows: x -> s1 -> out1 -> s2
  float* P2 = smem + a.p2;  // dff-wide: the product before bias / ReLU (fp32 rows) -> planes of f
  float* OS = smem + a.os;  // [64] row scales of the d-wide operand, [64] of f
  float* PV = OS + 128;     // bd | g1 | b1 | bf2 | g2 | b2 (6 d) | bf1 | bo1 | wo2 (3 dff) | bo2
  const int tid = threadIdx.x, lane_id = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int e = tid; e < 6 * d + 3 * dff + 1; e += NT) {
    int64_t src;
    if (e < 6 * d) {
      const int v = e / d;
      src = (v == 0 ? a.bd : v == 1 ? a.g1 : v == 2 ? a.b1 : v == 3 ? a.bf2 : v == 4 ? a.g2 : a.b2) + (e - v * d);
    } else {
      const int e2 = e - 6 * d, v = e2 / dff;
      src = v == 0 ? a.bf1 + e2 : !a.head ? a.bf1 : v == 1 ? a.bo1 + (e2 - dff) : v == 2 ? a.wo2 + (e2 - 2 * dff) : a.bo2;
    }
    PV[e] = params[src];
  }
  const float* pbd = PV, *pg1 = PV + d, *pb1 = PV + 2 * d, *pbf2 = PV + 3 * d, *pg2 = PV + 4 * d, *pb2 = PV + 5 * d;
  const float* pbf1 = PV + 6 * d, *pbo1 = pbf1 + dff, *pwo2 = pbo1 + dff;
  float4 ar[8], xr[8];
  auto request = [&](int tile) {  // past the last tile: empty extents, every load returns zero without touching memory
    const int64_t n0 = (int64_t)tile * R;
    const bool live = tile < a.ntiles;
    const int vr = live ? (int)((a.T - n0) < R ? (a.T - n0) : R) : 0;
    const Src as = make_src(sv + a.A + (live ? n0 : 0) * d, (int64_t)vr * d), xs = make_src(sv + a.x + (live ? n0 : 0) * d, (int64_t)vr * d);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      ar[k] = buf_ld4s(as, (unsigned)lane_id * 16u, (unsigned)((wave + NW * k) * d) * 4u);
      xr[k] = buf_ld4s(xs, (unsigned)lane_id * 16u, (unsigned)((wave + NW * k) * d) * 4u);
    }
  };
  request(blockIdx.x);
  lds_barrier();  // PV
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int64_t n0 = (int64_t)tile * R;
    const int vr = (int)((a.T - n0) < R ? (a.T - n0) : R);
    int lane = lane_id;
    asm volatile("" : "+v"(lane));
    // ---- the attention rows as a per-row-scaled plane pair, the residual rows as fp32 --------------------------------------------------
    SRF_STAMP(0);
    {
      float am[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) am[k] = max4(ar[k]);
      wave_max_n<8>(am);
      _Float16* AH = reinterpret_cast<_Float16*>(P0);
      _Float16* AL = AH + (R + 1) * ld;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = wave + NW * k, rc = r < R ? r : R;
        float sc, inv;
        fb_h3_scale(am[k], sc, inv);
        fbh4 hi, lo;
        fb_h3_split4(ar[k], sc, hi, lo);
        *reinterpret_cast<fbh4*>(AH + rc * ld + 4 * lane) = hi;
        *reinterpret_cast<fbh4*>(AL + rc * ld + 4 * lane) = lo;
        if (lane == 0) OS[r] = inv * (1.0f / ULTR_H3_WSCALE);
        st4(P1 + rc * ld + 4 * lane, xr[k]);
      }
    }
    SRF_STAMP(1);
    lds_barrier();
    // ---- s1 = x + (A Wd^T + bd): to `saved` and back into P1 -------------------------------------------------------------------------------
    {
      const Dst ds1 = make_dst(sv + a.s1 + n0 * d, (int64_t)vr * d);
      product_d4<true, true, true>(wave, lane, R, P0, ld, d >> 5, planes, a.gd, d, OS, P1, ld, ds1, pbd);
    }
    SRF_STAMP(2);
    lds_barrier();
    // ---- out1 = LN1(s1) -----------------------------------------------------------------------------------------------------------------------
    // The next tile's rows are requested HERE: with out1 not written (the backward recomputes it) and its statistics held back,
    // LayerNorm_1 issues NO vector-memory instruction - any phase that does (stores included) stalls ~7k cycles behind the 128 loads the
    // CU then has waiting for HBM (phase trace: whichever LayerNorm followed the request took 13 - 14k cycles instead of 6 - 7k) - and the
    // f product's first weight wait is 7k cycles away (the vmcnt counter is in order).
    __builtin_amdgcn_sched_barrier(0);
    request(tile + (int)gridDim.x);
    __builtin_amdgcn_sched_barrier(0);
    float m1v[8], r1v[8];
    {
      const Dst dout = make_dst(sv + a.out1 + n0 * d, (int64_t)vr * d), dm = make_dst(sv + a.m1 + n0, vr), dr = make_dst(sv + a.r1 + n0, vr);
      const float4 g4 = ld4(pg1 + 4 * lane), b4 = ld4(pb1 + 4 * lane);
      if (a.skip_out1) ln_fwd_rows<true>(wave, lane, R, ld, P1, P0, OS, g4, b4, dout, dm, dr, false, true, m1v, r1v);
      else ln_fwd_rows<false>(wave, lane, R, ld, P1, P0, OS, g4, b4, dout, dm, dr, true, true, m1v, r1v);
    }
    SRF_STAMP(3);
    lds_barrier();
    // ---- f = relu(out1 Wf1^T + bf1) ---------------------------------------------------------------------------------------------------------
    {
      const Dst dfo = make_dst(sv + a.f + n0 * dff, (int64_t)vr * dff), none = make_dst(sv, 0);
      product_f_relu(wave, lane, R, P0, P2, OS, planes, a.gf1, pbf1, dfo, nullptr, 0.f, none);
      if (a.skip_out1) {  // LayerNorm_1's statistics, held back while the next tile's rows were in flight
        const Dst dm = make_dst(sv + a.m1 + n0, vr), dr = make_dst(sv + a.r1 + n0, vr);
        const unsigned l0 = lane == 0 ? 0u : ULTR_OOB;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          buf_st1(dm, l0, (unsigned)(wave + NW * k) * 4u, m1v[k]);
          buf_st1(dr, l0, (unsigned)(wave + NW * k) * 4u, r1v[k]);
        }
      }
    }
    SRF_STAMP(4);
    lds_barrier();
    // ---- s2 = out1 + (f Wf2^T + bf2) ---------------------------------------------------------------------------------------------------------
    {
      const Dst ds2 = make_dst(sv + a.s2 + n0 * d, (int64_t)vr * d);
      product_d4<true, true, true>(wave, lane, R, P2, ldf, dff >> 5, planes, a.gf2, dff, OS + 64, P1, ld, ds2, pbf2);
    }
    SRF_STAMP(5);
    SRF_STAMP(6);
    lds_barrier();
    // ---- x' = LN2(s2) ------------------------------------------------------------------------------------------------------------------------
    {
      const Dst dout = make_dst(sv + a.xn + n0 * d, (int64_t)vr * d), dm = make_dst(sv + a.m2 + n0, vr), dr = make_dst(sv + a.r2 + n0, vr);
      const float4 g4 = ld4(pg2 + 4 * lane), b4 = ld4(pb2 + 4 * lane);
      float mu[8], ru[8];
      ln_fwd_rows<false>(wave, lane, R, ld, P1, P0, OS, g4, b4, dout, dm, dr, true, a.head != 0, mu, ru);
    }
    SRF_STAMP(7);
    if (a.head) {
      lds_barrier();
      // ---- oh = relu(x' Wo1^T + bo1), score = oh . wo2 + bo2 --------------------------------------------------------------------------------
      const Dst doh = make_dst(sv + a.oh + n0 * dff, (int64_t)vr * dff), dsc = make_dst(scores + n0, vr);
      product_f_relu(wave, lane, R, P0, P2, OS, planes, a.go1, pbo1, doh, pwo2, pwo2[dff], dsc);
    }
    SRF_STAMP(8);
    lds_barrier();
    SRF_STAMP(9);
  }
}

}  // namespace

int sr_fwd_block_launch(SrFwdBlockArgs a, int nwg, const float* params, const _Float16* planes, float* sv, float* scores, hipStream_t st) {
  if (a.d != SR_BWD_D || a.dff != SR_BWD_DFF || a.R < 4 || a.R > 64 || nwg <= 0 || nwg > SR_BWD_MAXWG) return ULTR_E_UNSUPPORTED;
  a.p0 = 0;
  a.p1 = (a.R + 1) * (SR_BWD_D + 8);
  a.p2 = 2 * a.p1;
  a.os = a.p2 + (a.R + 1) * (SR_BWD_DFF + 8);
  const size_t lds = ((size_t)a.os + 128 + 6 * SR_BWD_D + 3 * SR_BWD_DFF + 4) * sizeof(float);
  const int rc = set_lds(sr_fwd_block_kernel, lds);
  if (rc != 0) return rc;
  hipLaunchKernelGGL(sr_fwd_block_kernel, dim3(nwg), dim3(NT), lds, st, a, params, planes, sv, scores);
  return (int)hipGetLastError();
}
