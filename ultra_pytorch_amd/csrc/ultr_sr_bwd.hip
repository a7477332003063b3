// ultr_sr_bwd.hip - SetRank backward, the row-local chain of an encoder block as TWO launches (round 6).
//
// Reference: ultra/ranking_model/SetRank.py:92-111 (MultiHeadSelfAttention's dense + residual + LayerNorm, the FFN + residual +
// LayerNorm), differentiated.  Until round 5 the chain was seven launches per block - LayerNorm backward (+ column sums), two thin
// weight-gradient launches, three dgrad GEMMs, LayerNorm backward again - each reading and writing [T, 256] tensors: 1.5 GB of HBM
// traffic per block at BASELINE config 5 (T = 102 400 token rows).  Here a PERSISTENT workgroup per compute unit (eight waves, up
// to 256 registers each, ~150 KB of LDS) walks over tiles of R <= 64 token rows and keeps everything of a tile on chip:
//
//   sr_bwd_ffn_kernel   d s2 = LN2'(d x')            row-wise, one wave per row, statistics from `saved`
//                       d Wf2 += d s2^T f            fp32 matrix cores (v_mfma_f32_16x16x4_f32) straight from the fp32 rows in LDS;
//                                                    the [256 x 64] accumulator lives in registers across ALL tiles of the workgroup
//                       d f = (d s2 Wf2) o [f > 0]   split-half product (three f16 MFMAs, ultr_h3.h) on a fragment copy of Wf2^T
//                       d out1 = d s2 + d f Wf1      split-half product on a fragment copy of Wf1^T, residual from LDS
//   sr_bwd_proj_kernel  d s1 = LN1'(d out1)          (+ out1 recomputed from s1: the backward never reads the saved out1)
//                       d Wf1 += d f^T out1          fp32 matrix cores, accumulator in registers across tiles
//                       d A = d s1 Wd                split-half product on a fragment copy of Wd^T
//
// HBM traffic per block: reads d x', s2, f | d out1, s1, d f; writes d f, d out1 | d s1, d A  = 0.81 GB.  Column sums (LayerNorm
// gamma / beta, the three biases) ride along in registers; one partial per workgroup leaves at the end (256 partials of 17 k floats
// instead of a 64 KB slab per 800 rows), folded by sr_fold_all_kernel in fixed order: deterministic, no atomics.
// Shapes: d_model 256, dff 64 (BASELINE config 5); ultr_setrank_backward keeps the separate launches for everything else.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ultr_hip.h"
#include "ultr_device.h"
#include "ultr_h3.h"
#include "ultr_plan.h"
#include "ultr_sr_bwd.h"
#include "ultr_sr_tiles.h"

#ifdef ULTR_TRACE
// phase stamps (s_memtime) of wave 0 of the first 64 workgroups on their SECOND tile; slot 16 k + j: kernel k (0 ffn, 1 proj), stamp j
__device__ unsigned long long g_srb_trace[64 * 32];
#define SRB_STAMP(kern, slot)                                                                                         \
  do {                                                                                                                \
    if (threadIdx.x == 0 && blockIdx.x < 64 && tile == (int)(blockIdx.x + gridDim.x)) g_srb_trace[blockIdx.x * 32 + 16 * (kern) + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
extern "C" int ultr_srb_trace_read(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_srb_trace), sizeof(unsigned long long) * 64 * 32);
}
#else
#define SRB_STAMP(kern, slot) \
  do {                        \
  } while (0)
#endif

namespace {

// the HBM operands of a tile's row phase, requested one tile ahead (the workgroup is alone on its CU: nobody else hides the latency):
// rows wave + 8 q of d y and s, the dff-wide rows 4 (wave + 8 q2) + (lane >> 4), and the rows' statistics one per lane (lane k < 8:
// row wave + 8 k; read back with v_readlane - two registers instead of sixteen.  Scalar loads were tried: they share the LDS counter,
// so the first LDS read of the weight-gradient loop waited ~12k cycles for sixteen cold scalar-cache misses)
struct RowRegs {
  float4 dy[8], s[8];
  float mv, rv;
};
__device__ __forceinline__ void load_rows(RowRegs& g, int wave, int lane, const float* dy, const float* s, const float* __restrict__ mean,
                                          const float* __restrict__ rstd, int64_t n0, int vr) {
  constexpr int d = SR_BWD_D;
  const Src dys = make_src(dy + n0 * d, (int64_t)vr * d), ss = make_src(s + n0 * d, (int64_t)vr * d);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int r = wave + NW * k;
    g.dy[k] = buf_ld4s(dys, (unsigned)lane * 16u, (unsigned)(r * d) * 4u);
    g.s[k] = buf_ld4s(ss, (unsigned)lane * 16u, (unsigned)(r * d) * 4u);
  }
  const Src ms = make_src(mean + n0, vr), rs = make_src(rstd + n0, vr);
  g.mv = buf_ld1(ms, lane < 8 ? (unsigned)(wave + NW * lane) * 4u : ULTR_OOB);
  g.rv = buf_ld1(rs, lane < 8 ? (unsigned)(wave + NW * lane) * 4u : ULTR_OOB);
}
// the dff-wide rows 4 (wave + 8 q2) + (lane >> 4) of a tile: requested at the START of its row phase, used at its end (16 KB per tile:
// the LayerNorm arithmetic covers the round trip; eight registers less to carry through the products)
__device__ __forceinline__ void load_frows(float4 (&fv)[2], int wave, int lane, const float* f, int64_t n0, int vr) {
  constexpr int dff = SR_BWD_DFF;
  static_assert(dff == 64, "a wave reads four dff-wide rows as 64 consecutive float4");
  const Src fs = make_src(f + n0 * dff, (int64_t)vr * dff);
#pragma unroll
  for (int q2 = 0; q2 < 2; ++q2) fv[q2] = buf_ld4s(fs, (unsigned)lane * 16u, (unsigned)(4 * (wave + NW * q2) * dff) * 4u);
}

// rows 2 pr, 2 pr + 1 of the wave's eight (pr < 4)
template <int MODE>
__device__ __forceinline__ void ln_bwd_rows(int pr, int wave, int lane, int R, int ld, const RowRegs& rg, const float4 g4, const float4 b4,
                                            const Dst& dso, float* P0, float* P1, float* OS, float4& cg, float4& cb, float4& cd) {
  constexpr int d = SR_BWD_D, NR = 2;
  const int c = 4 * lane;
  float4 dy[NR], xh[NR];
  float m[NR], rr[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    dy[k] = rg.dy[NR * pr + k];
    xh[k] = rg.s[NR * pr + k];
    m[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rg.mv), NR * pr + k));
    rr[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rg.rv), NR * pr + k));
  }
  float ss[2 * NR];
  float4 g[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    xh[k] = make_float4((xh[k].x - m[k]) * rr[k], (xh[k].y - m[k]) * rr[k], (xh[k].z - m[k]) * rr[k], (xh[k].w - m[k]) * rr[k]);
    g[k] = make_float4(dy[k].x * g4.x, dy[k].y * g4.y, dy[k].z * g4.z, dy[k].w * g4.w);
    ss[k] = (g[k].x + g[k].y) + (g[k].z + g[k].w);
    ss[NR + k] = (g[k].x * xh[k].x + g[k].y * xh[k].y) + (g[k].z * xh[k].z + g[k].w * xh[k].w);
    cg.x += dy[k].x * xh[k].x; cg.y += dy[k].y * xh[k].y; cg.z += dy[k].z * xh[k].z; cg.w += dy[k].w * xh[k].w;
    cb.x += dy[k].x; cb.y += dy[k].y; cb.z += dy[k].z; cb.w += dy[k].w;
  }
  wave_sum_n<2 * NR>(ss);
  float am[NR];
  float4 v[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const float a1 = ss[k] * (1.0f / (float)d), a2 = ss[NR + k] * (1.0f / (float)d);
    v[k] = make_float4(rr[k] * (g[k].x - a1 - xh[k].x * a2), rr[k] * (g[k].y - a1 - xh[k].y * a2),
                       rr[k] * (g[k].z - a1 - xh[k].z * a2), rr[k] * (g[k].w - a1 - xh[k].w * a2));
    cd.x += v[k].x; cd.y += v[k].y; cd.z += v[k].z; cd.w += v[k].w;
    am[k] = max4(v[k]);
  }
  wave_max_n<NR>(am);
  _Float16* AH = reinterpret_cast<_Float16*>(P0);
  _Float16* AL = AH + (R + 1) * ld;
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int r = wave + NW * (NR * pr + k), rc = r < R ? r : R;
    float sc, inv;
    fb_h3_scale(am[k], sc, inv);
    fbh4 hi, lo;
    fb_h3_split4(v[k], sc, hi, lo);
    *reinterpret_cast<fbh4*>(AH + rc * ld + c) = hi;
    *reinterpret_cast<fbh4*>(AL + rc * ld + c) = lo;
    if (lane == 0) OS[r] = inv * (1.0f / ULTR_H3_WSCALE);
    if constexpr (MODE == 0) {
      st4(P1 + rc * ld + c, v[k]);
    } else {
      buf_st4(dso, (unsigned)c * 4u, (unsigned)(r * d) * 4u, v[k]);
      st4(P1 + rc * ld + c, make_float4(xh[k].x * g4.x + b4.x, xh[k].y * g4.y + b4.y, xh[k].z * g4.z + b4.z, xh[k].w * g4.w + b4.w));
    }
  }
}

// d W [dff x d] += A^T B on the fp32 matrix cores, A = fp32 rows [R][dff] at PA (row stride dff + 8), B = fp32 rows [R][d] at PB (row
// stride d + 8): wave = 32 output columns, lane (i, q) reads row 4 step + q - a float4 of A at column 4 i feeds the four 16-row
// sub-tiles (rows 4 i + ta of d W), a float2 of B at column 32 wave + 2 i the two column sub-tiles.  The next step's operands leave LDS
// while this step's eight MFMAs run.  accW[ta][tb][r] = d W[4 (4 q + r) + ta][32 wave + 2 i + tb].
__device__ __forceinline__ void wgrad_thin_wide(f32x4 (&accW)[4][2], const float* PA, const float* PB, int R, int wave, int lane) {
  constexpr int ld = SR_BWD_D + 8, ldf = SR_BWD_DFF + 8;
  const int i = lane & 15, q = lane >> 4;
  const float* pa = PA + q * ldf + 4 * i;
  const float* pb = PB + q * ld + 32 * wave + 2 * i;
  const int nk = R >> 2;
  float4 av = ld4(pa);
  float2 bv = ld2(pb);
  for (int kk = 0; kk < nk; ++kk) {
    const int kn = kk + 1 < nk ? kk + 1 : kk;
    const float4 an = ld4(pa + 4 * kn * ldf);
    const float2 bn = ld2(pb + 4 * kn * ld);
    const float aa[4] = {av.x, av.y, av.z, av.w};
    const float bb[2] = {bv.x, bv.y};
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) accW[ta][tb] = mfma16(aa[ta], bb[tb], accW[ta][tb]);
    av = an;
    bv = bn;
  }
}
__device__ __forceinline__ void wgrad_thin_wide_store(const f32x4 (&accW)[4][2], float* pw, int wave, int lane) {
  constexpr int d = SR_BWD_D;
  const int i = lane & 15, q = lane >> 4;
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = 4 * (4 * q + r) + ta;
      *reinterpret_cast<float2*>(pw + m * d + 32 * wave + 2 * i) = make_float2(accW[ta][0][r], accW[ta][1][r]);
    }
}

// the eight waves' column sums through LDS, fixed order: smem[wave][nv][d] -> out[v * d + c]
__device__ __forceinline__ void colsum_store(float* smem, int wave, int lane, const float4& c0, const float4& c1, const float4& c2) {
  constexpr int d = SR_BWD_D;
  float* mine = smem + wave * 3 * d;
  st4(mine + 4 * lane, c0);
  st4(mine + d + 4 * lane, c1);
  st4(mine + 2 * d + 4 * lane, c2);
}
__device__ __forceinline__ float colsum_fold(const float* smem, int e) {
  constexpr int d = SR_BWD_D;
  float t = smem[e];
#pragma unroll
  for (int w = 1; w < NW; ++w) t += smem[w * 3 * d + e];
  return t;
}

__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void sr_bwd_ffn_kernel(SrBwdFfnArgs a, const float* __restrict__ params,
                                                                                             const _Float16* __restrict__ planes,
                                                                                             const float* __restrict__ sv, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int d = SR_BWD_D, dff = SR_BWD_DFF, ld = d + 8, ldf = dff + 8;
  const int R = a.R;
  float* P0 = smem + a.p0;  // planes of d s2
  float* P1 = smem + a.p1;  // fp32 rows of d s2
  float* P2 = smem + a.p2;  // f (fp32 rows) -> d f before the mask (fp32 rows) -> planes of d f
  float* OS = smem + a.os;  // [64] row scales of d s2, [64] of d f
  const int tid = threadIdx.x, lane_id = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* PG = OS + 128;     // gamma (re-read per tile: four registers less across the products)
  for (int e = tid; e < d; e += NT) PG[e] = params[a.gamma + e];
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 cg = z4, cb = z4, cd = z4;
  f32x4 accW[4][2];  // d Wf2: rows 64 (wave >> 1) + 4 (4 q + r) + ta, columns 32 (wave & 1) + 2 i + tb
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) accW[ta][tb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int mblk = wave >> 1, nhalf = wave & 1;

  RowRegs rg;
  auto request = [&](int tile) {  // past the last tile: empty extents, every load returns zero without touching memory
    const int64_t n0 = (int64_t)tile * R;
    const int vr = tile < a.ntiles ? (int)((a.T - n0) < R ? (a.T - n0) : R) : 0;
    load_rows(rg, wave, lane_id, ws + a.dy, sv + a.s, sv + a.mean, sv + a.rstd, tile < a.ntiles ? n0 : 0, vr);
  };
  request(blockIdx.x);
  lds_barrier();  // PG
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int64_t n0 = (int64_t)tile * R;
    const int vr = (int)((a.T - n0) < R ? (a.T - n0) : R);
    int lane = lane_id;
    asm volatile("" : "+v"(lane));
    // ---- row phase: LayerNorm_2 backward, f to LDS -------------------------------------------------------------------------
    SRB_STAMP(0, 0);
    unsigned fmask = 0;
    {
      const Dst none = make_dst(ws, 0);
      float4 fv[2];
      load_frows(fv, wave, lane, sv + a.f, n0, vr);
      const float4 g4 = ld4(PG + 4 * lane);
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) ln_bwd_rows<0>(pr, wave, lane, R, ld, rg, g4, z4, none, P0, P1, OS, cg, cb, cd);
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const int row = 4 * (wave + NW * q2) + (lane >> 4), rc = row < R ? row : R;
        st4(P2 + rc * ldf + 4 * (lane & 15), fv[q2]);
        fmask |= ((fv[q2].x > 0.f ? 1u : 0u) | (fv[q2].y > 0.f ? 2u : 0u) | (fv[q2].z > 0.f ? 4u : 0u) | (fv[q2].w > 0.f ? 8u : 0u)) << (4 * q2);
      }
    }
    SRB_STAMP(0, 1);
    lds_barrier();
    SRB_STAMP(0, 2);
    // ---- d f = d s2 Wf2 (before the mask) on the split-half copies; then the NEXT tile's rows are requested and
    //      d Wf2 += d s2^T f runs on the fp32 matrix cores from LDS while they travel ---------------------------------------------------
    f32x4 accf[1][2] = {{(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}}};
    const int rt = wave >> 1, chf = wave & 1;
    {
      const int i = lane & 15, q = lane >> 4;
      const _Float16* AH = reinterpret_cast<const _Float16*>(P0);
      const int rowi = 16 * rt + i;
      const _Float16* pa[1] = {AH + (rowi < R ? rowi : R) * ld + 8 * q};
      const Src Wh = make_src(reinterpret_cast<const float*>(planes + a.gt2), (int64_t)d * dff);
      PipeH3W<1, 2> ph;
      ph.begin(Wh, chf, d >> 5, 0, d >> 5, true, lane);
      ph.run(pa, (R + 1) * ld, Wh, d >> 5, accf);
    }
    SRB_STAMP(0, 3);
    __builtin_amdgcn_sched_barrier(0);
    request(tile + (int)gridDim.x);
    __builtin_amdgcn_sched_barrier(0);
    SRB_STAMP(0, 4);
    {
      const int i = lane & 15, q = lane >> 4;
      const float* pa = P1 + q * ld + 64 * mblk + 4 * i;
      const float* pb = P2 + q * ldf + 32 * nhalf + 2 * i;
      const int nk = R >> 2;
      float4 av = ld4(pa);
      float2 bv = ld2(pb);
      for (int kk = 0; kk < nk; ++kk) {  // the next step's operands leave LDS while this step's eight MFMAs run
        const int kn = kk + 1 < nk ? kk + 1 : kk;
        const float4 an = ld4(pa + 4 * kn * ld);
        const float2 bn = ld2(pb + 4 * kn * ldf);
        const float aa[4] = {av.x, av.y, av.z, av.w};
        const float bb[2] = {bv.x, bv.y};
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
          for (int tb = 0; tb < 2; ++tb) accW[ta][tb] = mfma16(aa[ta], bb[tb], accW[ta][tb]);
        av = an;
        bv = bn;
      }
    }
    SRB_STAMP(0, 5);
    lds_barrier();  // every wave has read f
    SRB_STAMP(0, 6);
    {
      const int i = lane & 15, q = lane >> 4;
      const float4 o4 = ld4(OS + 16 * rt + 4 * q);
      const float o[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * rt + 4 * q + r, rc = row < R ? row : R;
        *reinterpret_cast<float2*>(P2 + rc * ldf + 32 * chf + 2 * i) = make_float2(accf[0][0][r] * o[r], accf[0][1][r] * o[r]);
      }
    }
    lds_barrier();
    // ---- the ReLU mask; d f to global memory and, scaled per row, to its plane pair ----------------------------------------------
    {
      const Dst dfo = make_dst(ws + a.dF + n0 * dff, (int64_t)vr * dff);
      float4 v[2];
      float am[2];
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const int row = 4 * (wave + NW * q2) + (lane >> 4), rc = row < R ? row : R;
        v[q2] = ld4(P2 + rc * ldf + 4 * (lane & 15));
        const unsigned mk = fmask >> (4 * q2);
        v[q2].x = (mk & 1u) ? v[q2].x : 0.f;
        v[q2].y = (mk & 2u) ? v[q2].y : 0.f;
        v[q2].z = (mk & 4u) ? v[q2].z : 0.f;
        v[q2].w = (mk & 8u) ? v[q2].w : 0.f;
        buf_st4(dfo, (unsigned)(row * dff + 4 * (lane & 15)) * 4u, 0u, v[q2]);
        am[q2] = row16_max(max4(v[q2]));
      }
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
        *reinterpret_cast<fbh4*>(FH + rc * ldf + 4 * (lane & 15)) = hi;
        *reinterpret_cast<fbh4*>(FL + rc * ldf + 4 * (lane & 15)) = lo;
        if ((lane & 15) == 0) OS[64 + row] = inv * (1.0f / ULTR_H3_WSCALE);
      }
    }
    lds_barrier();
    SRB_STAMP(0, 7);
    // ---- d out1 = d s2 + d f Wf1 ---------------------------------------------------------------------------------------------------
    {
      const Dst dxo = make_dst(ws + a.dx + n0 * d, (int64_t)vr * d);
      product_d4<true>(wave, lane, R, P2, ldf, dff >> 5, planes, a.gt1, dff, OS + 64, P1, ld, dxo);
    }
    SRB_STAMP(0, 8);
    lds_barrier();
    SRB_STAMP(0, 9);
  }
  // ---- the workgroup's partial: d Wf2 | d bf2 | d g2 | d b2 -------------------------------------------------------------------------
  float* pw = ws + a.part + (int64_t)blockIdx.x * a.part_stride;
  {
    const int i = lane_id & 15, q = lane_id >> 4;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 64 * mblk + 4 * (4 * q + r) + ta;
        *reinterpret_cast<float2*>(pw + m * dff + 32 * nhalf + 2 * i) = make_float2(accW[ta][0][r], accW[ta][1][r]);
      }
  }
  colsum_store(smem, wave, lane_id, cd, cg, cb);
  lds_barrier();
  for (int e = tid; e < 3 * d; e += NT) pw[d * dff + e] = colsum_fold(smem, e);
}

__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void sr_bwd_proj_kernel(SrBwdProjArgs a, const float* __restrict__ params,
                                                                                              const _Float16* __restrict__ planes,
                                                                                              const float* __restrict__ sv, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int d = SR_BWD_D, dff = SR_BWD_DFF, ld = d + 8, ldf = dff + 8;
  const int R = a.R;
  float* P0 = smem + a.p0;  // planes of d s1
  float* P1 = smem + a.p1;  // fp32 rows of out1 (recomputed)
  float* PF = smem + a.pf;  // fp32 rows of d f
  float* OS = smem + a.os;  // [64] row scales of d s1
  const int tid = threadIdx.x, lane_id = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* PG = OS + 128;     // gamma | beta (re-read per tile: eight registers less across the products)
  for (int e = tid; e < 2 * d; e += NT) PG[e] = params[(e < d ? a.gamma : a.beta - d) + e];
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 cg = z4, cb = z4, cd = z4, cf = z4;
  f32x4 accW[4][2];  // d Wf1: rows 4 (4 q + r) + ta, columns 32 wave + 2 i + tb
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) accW[ta][tb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  RowRegs rg;
  auto request = [&](int tile) {  // past the last tile: empty extents, every load returns zero without touching memory
    const int64_t n0 = (int64_t)tile * R;
    const int vr = tile < a.ntiles ? (int)((a.T - n0) < R ? (a.T - n0) : R) : 0;
    load_rows(rg, wave, lane_id, ws + a.dy, sv + a.s, sv + a.mean, sv + a.rstd, tile < a.ntiles ? n0 : 0, vr);
  };
  request(blockIdx.x);
  lds_barrier();  // PG
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int64_t n0 = (int64_t)tile * R;
    const int vr = (int)((a.T - n0) < R ? (a.T - n0) : R);
    int lane = lane_id;
    asm volatile("" : "+v"(lane));
    SRB_STAMP(1, 0);
    {
      const Dst dso = make_dst(ws + a.ds + n0 * d, (int64_t)vr * d);
      float4 fv[2];
      load_frows(fv, wave, lane, ws + a.dF, n0, vr);
      const float4 g4 = ld4(PG + 4 * lane), b4 = ld4(PG + d + 4 * lane);
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) ln_bwd_rows<1>(pr, wave, lane, R, ld, rg, g4, b4, dso, P0, P1, OS, cg, cb, cd);
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const int row = 4 * (wave + NW * q2) + (lane >> 4), rc = row < R ? row : R;
        st4(PF + rc * ldf + 4 * (lane & 15), fv[q2]);
        cf.x += fv[q2].x; cf.y += fv[q2].y; cf.z += fv[q2].z; cf.w += fv[q2].w;
      }
    }
    SRB_STAMP(1, 1);
    lds_barrier();
    SRB_STAMP(1, 2);
    // ---- d A = d s1 Wd (in place over d out1 when a.dx == a.dy: this tile's rows were read above) -----------------------------------
    {
      const Dst dxo = make_dst(ws + a.dx + n0 * d, (int64_t)vr * d);
      product_d4<false>(wave, lane, R, P0, ld, d >> 5, planes, a.gtd, d, OS, nullptr, ld, dxo);
    }
    // the next tile's rows travel while the weight gradient runs from LDS (when d A overwrites d out1 in place, the next tile's rows
    // are other rows: every tile belongs to exactly one workgroup)
    SRB_STAMP(1, 3);
    __builtin_amdgcn_sched_barrier(0);
    request(tile + (int)gridDim.x);
    __builtin_amdgcn_sched_barrier(0);
    SRB_STAMP(1, 4);
    // ---- d Wf1 += d f^T out1 ----------------------------------------------------------------------------------------------------------
    wgrad_thin_wide(accW, PF, P1, R, wave, lane);
    SRB_STAMP(1, 5);
    lds_barrier();
    SRB_STAMP(1, 6);
  }
  // ---- the workgroup's partial: d Wf1 | d bf1 | d bd | d g1 | d b1 ---------------------------------------------------------------------
  float* pw = ws + a.part + (int64_t)blockIdx.x * a.part_stride;
  wgrad_thin_wide_store(accW, pw, wave, lane_id);
  colsum_store(smem, wave, lane_id, cd, cg, cb);
  float* sf = smem + NW * 3 * d;  // [8 waves x 4 row groups][dff]
  st4(sf + (wave * 4 + (lane_id >> 4)) * dff + 4 * (lane_id & 15), cf);
  lds_barrier();
  if (tid < dff) {
    float t = sf[tid];
#pragma unroll
    for (int k = 1; k < 4 * NW; ++k) t += sf[k * dff + tid];
    pw[dff * d + tid] = t;
  }
  for (int e = tid; e < 3 * d; e += NT) pw[dff * d + dff + e] = colsum_fold(smem, e);
}

// sr_bwd_head_kernel: the output FFN's backward (SetRank.py:136, 153 backwards): score = oh . wo2 + bo2, oh = relu(x Wo1^T + bo1)
//   d oh = d score wo2 o [oh > 0],  d wo2 = sum d score oh,  d bo2 = sum d score,  d Wo1 += d oh^T x,  d bo1 = sum d oh,  d x = d oh Wo1
// (was: sr_head_bwd_kernel + a thin weight-gradient launch + a dgrad GEMM - d oh written and read twice, x read once: 315 MB; here
// 236 MB, and d oh never leaves the chip).  Partial per workgroup: [d Wo1 (dff x d) | d bo1 (dff) | d wo2 (dff) | d bo2 | pad].
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void sr_bwd_head_kernel(SrBwdHeadArgs a, const float* __restrict__ params,
                                                                                              const _Float16* __restrict__ planes,
                                                                                              const float* __restrict__ sv, const float* __restrict__ dscores,
                                                                                              float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int d = SR_BWD_D, dff = SR_BWD_DFF, ld = d + 8, ldf = dff + 8;
  const int R = a.R;
  float* P0 = smem + a.p0;  // planes of d oh
  float* P1 = smem + a.p1;  // fp32 rows of x
  float* PF = smem + a.pf;  // fp32 rows of d oh
  float* OS = smem + a.os;  // [64] row scales of d oh
  const int tid = threadIdx.x, lane_id = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* pw2 = params + a.wo2 + 4 * (lane_id & 15);
  const float4 w4 = make_float4(pw2[0], pw2[1], pw2[2], pw2[3]);
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 cw = z4, cb1 = z4;
  float cds = 0.f;
  f32x4 accW[4][2];
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) accW[ta][tb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 xr[8];
  auto request = [&](int tile) {
    const int64_t n0 = (int64_t)tile * R;
    const int vr = tile < a.ntiles ? (int)((a.T - n0) < R ? (a.T - n0) : R) : 0;
    const Src xs = make_src(sv + a.x + (tile < a.ntiles ? n0 : 0) * d, (int64_t)vr * d);
#pragma unroll
    for (int k = 0; k < 8; ++k) xr[k] = buf_ld4s(xs, (unsigned)lane_id * 16u, (unsigned)((wave + NW * k) * d) * 4u);
  };
  request(blockIdx.x);
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int64_t n0 = (int64_t)tile * R;
    const int vr = (int)((a.T - n0) < R ? (a.T - n0) : R);
    int lane = lane_id;
    asm volatile("" : "+v"(lane));
    {
      const Src dys = make_src(dscores + n0, vr), ohs = make_src(sv + a.oh + n0 * dff, (int64_t)vr * dff);
      float4 ov[2];
      float dsv[2];
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        ov[q2] = buf_ld4s(ohs, (unsigned)lane * 16u, (unsigned)(4 * (wave + NW * q2) * dff) * 4u);
        dsv[q2] = buf_ld1(dys, (unsigned)(4 * (wave + NW * q2) + (lane >> 4)) * 4u);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = wave + NW * k;
        st4(P1 + (r < R ? r : R) * ld + 4 * lane, xr[k]);
      }
      _Float16* FH = reinterpret_cast<_Float16*>(P0);
      _Float16* FL = FH + (R + 1) * ldf;
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const int row = 4 * (wave + NW * q2) + (lane >> 4), rc = row < R ? row : R;
        const float ds = dsv[q2];
        const float4 o = ov[q2];
        const float4 g = make_float4(o.x > 0.f ? ds * w4.x : 0.f, o.y > 0.f ? ds * w4.y : 0.f, o.z > 0.f ? ds * w4.z : 0.f, o.w > 0.f ? ds * w4.w : 0.f);
        cw.x = fmaf(ds, o.x, cw.x); cw.y = fmaf(ds, o.y, cw.y); cw.z = fmaf(ds, o.z, cw.z); cw.w = fmaf(ds, o.w, cw.w);
        cb1.x += g.x; cb1.y += g.y; cb1.z += g.z; cb1.w += g.w;
        cds += (lane & 15) == 0 ? ds : 0.f;
        st4(PF + rc * ldf + 4 * (lane & 15), g);
        const float am = row16_max(max4(g));
        float sc, inv;
        fb_h3_scale(am, sc, inv);
        fbh4 hi, lo;
        fb_h3_split4(g, sc, hi, lo);
        *reinterpret_cast<fbh4*>(FH + rc * ldf + 4 * (lane & 15)) = hi;
        *reinterpret_cast<fbh4*>(FL + rc * ldf + 4 * (lane & 15)) = lo;
        if ((lane & 15) == 0) OS[row] = inv * (1.0f / ULTR_H3_WSCALE);
      }
    }
    lds_barrier();
    {
      const Dst dxo = make_dst(ws + a.dx + n0 * d, (int64_t)vr * d);
      product_d4<false>(wave, lane, R, P0, ldf, dff >> 5, planes, a.gto1, dff, OS, nullptr, ld, dxo);
    }
    __builtin_amdgcn_sched_barrier(0);
    request(tile + (int)gridDim.x);
    __builtin_amdgcn_sched_barrier(0);
    wgrad_thin_wide(accW, PF, P1, R, wave, lane);
    lds_barrier();
  }
  float* pw = ws + a.part + (int64_t)blockIdx.x * a.part_stride;
  wgrad_thin_wide_store(accW, pw, wave, lane_id);
  float* sf = smem;  // [8 waves x 4 row groups][2 dff + 1 (+ 3 pad)]
  constexpr int sl = 2 * dff + 4;
  float* mine = sf + (wave * 4 + (lane_id >> 4)) * sl;
  st4(mine + 4 * (lane_id & 15), cb1);
  st4(mine + dff + 4 * (lane_id & 15), cw);
  if ((lane_id & 15) == 0) mine[2 * dff] = cds;
  lds_barrier();
  if (tid < 2 * dff + 1) {
    float t = sf[tid];
#pragma unroll
    for (int k = 1; k < 4 * NW; ++k) t += sf[k * sl + tid];
    pw[dff * d + tid] = t;
  }
}

// sr_bwd_embed_kernel: the embedding FFN's and the input LayerNorm's backward (SetRank.py:134-135, 146 backwards):
//   x_0 = h0 W2^T + b2,  h0 = relu(xn0 W1^T + b1),  xn0 = LN_in(xg) = xh g_in + b_in
//   d W2 += d x_0^T h0,  d b2 = sum d x_0,  d h0 = (d x_0 W2) o [h0 > 0],  d b1 = sum d h0,
//   d W1 = d h0^T xn0 = (d h0^T xh) diag(g_in) + d b1 (x) b_in   (xh is what LDS holds: the scale and the rank-one term are applied to the
//                                                                   workgroup's partial - both are linear in the rows),
//   d xn0 = d h0 W1,  d g_in = sum d xn0 o xh,  d b_in = sum d xn0          (no gradient flows into the features)
// (was: two thin weight-gradient launches, two dgrad GEMMs and a column-sum launch: d h0 and d xn0 written and re-read, d x_0 read
// three times - 0.65 GB; here d x_0, h0 and xg are read once, 0.22 GB, and nothing but the partials is written).
// Partial per workgroup, in the parameter vector's order: [d g_in (F) | d b_in (F) | d W1 (dff x F) | d b1 (dff) | d W2 (d x dff) | d b2 (d)].
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void sr_bwd_embed_kernel(SrBwdEmbedArgs a, const float* __restrict__ params,
                                                                                               const _Float16* __restrict__ planes,
                                                                                               const float* __restrict__ sv, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int d = SR_BWD_D, dff = SR_BWD_DFF, ld = d + 8, ldf = dff + 8;
  const int R = a.R, F = a.F;
  float* P0 = smem + a.p0;  // planes of d x_0 -> planes of d h0
  float* P1 = smem + a.p1;  // fp32 rows of d x_0 -> fp32 rows of xh
  float* P2 = smem + a.p2;  // fp32 rows of h0 -> d h0 before the mask -> d h0
  float* OS = smem + a.os;  // [64] row scales of d x_0, [64] of d h0
  const int tid = threadIdx.x, lane_id = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 cb2 = z4, cb1 = z4;
  float cgi[2] = {0.f, 0.f}, cbi[2] = {0.f, 0.f};
  f32x4 accW2[4][2], accW1[4][2];
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) accW2[ta][tb] = accW1[ta][tb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int mblk = wave >> 1, nhalf = wave & 1;
  const int nchF = (F + 31) >> 5;
  float4 xr[8];
  auto request = [&](int tile) {
    const int64_t n0 = (int64_t)tile * R;
    const int vr = tile < a.ntiles ? (int)((a.T - n0) < R ? (a.T - n0) : R) : 0;
    const Src xs = make_src(ws + a.dy + (tile < a.ntiles ? n0 : 0) * d, (int64_t)vr * d);
#pragma unroll
    for (int k = 0; k < 8; ++k) xr[k] = buf_ld4s(xs, (unsigned)lane_id * 16u, (unsigned)((wave + NW * k) * d) * 4u);
  };
  request(blockIdx.x);
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int64_t n0 = (int64_t)tile * R;
    const int vr = (int)((a.T - n0) < R ? (a.T - n0) : R);
    int lane = lane_id;
    asm volatile("" : "+v"(lane));
    // ---- row phase: d x_0 to LDS (fp32 rows + plane pair), h0 to LDS ------------------------------------------------------------------
    unsigned hmask = 0;
    {
      float4 hv[2];
      load_frows(hv, wave, lane, sv + a.h0, n0, vr);
      float am[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        am[k] = max4(xr[k]);
        cb2.x += xr[k].x; cb2.y += xr[k].y; cb2.z += xr[k].z; cb2.w += xr[k].w;
      }
      wave_max_n<8>(am);
      _Float16* AH = reinterpret_cast<_Float16*>(P0);
      _Float16* AL = AH + (R + 1) * ld;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = wave + NW * k, rc = r < R ? r : R;
        float sc, inv;
        fb_h3_scale(am[k], sc, inv);
        fbh4 hi, lo;
        fb_h3_split4(xr[k], sc, hi, lo);
        *reinterpret_cast<fbh4*>(AH + rc * ld + 4 * lane) = hi;
        *reinterpret_cast<fbh4*>(AL + rc * ld + 4 * lane) = lo;
        if (lane == 0) OS[r] = inv * (1.0f / ULTR_H3_WSCALE);
        st4(P1 + rc * ld + 4 * lane, xr[k]);
      }
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const int row = 4 * (wave + NW * q2) + (lane >> 4), rc = row < R ? row : R;
        st4(P2 + rc * ldf + 4 * (lane & 15), hv[q2]);
        hmask |= ((hv[q2].x > 0.f ? 1u : 0u) | (hv[q2].y > 0.f ? 2u : 0u) | (hv[q2].z > 0.f ? 4u : 0u) | (hv[q2].w > 0.f ? 8u : 0u)) << (4 * q2);
      }
    }
    lds_barrier();
    // ---- d h0 = d x_0 W2 (before the mask); the next tile's rows requested; d W2 += d x_0^T h0 ----------------------------------------
    f32x4 accf[1][2] = {{(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}}};
    const int rt = wave >> 1, chf = wave & 1;
    {
      const int i = lane & 15, q = lane >> 4;
      const _Float16* AH = reinterpret_cast<const _Float16*>(P0);
      const int rowi = 16 * rt + i;
      const _Float16* pa[1] = {AH + (rowi < R ? rowi : R) * ld + 8 * q};
      const Src Wh = make_src(reinterpret_cast<const float*>(planes + a.gt2), (int64_t)d * dff);
      PipeH3W<1, 2> ph;
      ph.begin(Wh, chf, d >> 5, 0, d >> 5, true, lane);
      ph.run(pa, (R + 1) * ld, Wh, d >> 5, accf);
    }
    __builtin_amdgcn_sched_barrier(0);
    // the feature rows of THIS tile: they travel while the weight gradient runs
    float4 xg[8];
    {
      const Src gs = make_src(sv + a.xg + n0 * F, (int64_t)vr * F);
#pragma unroll
      for (int k = 0; k < 8; ++k) xg[k] = buf_ld4s(gs, 4 * lane < F ? (unsigned)lane * 16u : ULTR_OOB, (unsigned)((wave + NW * k) * F) * 4u);
    }
    const Src ms = make_src(sv + a.mean + n0, vr), rs = make_src(sv + a.rstd + n0, vr);
    const float mv = buf_ld1(ms, lane < 8 ? (unsigned)(wave + NW * lane) * 4u : ULTR_OOB);
    const float rv = buf_ld1(rs, lane < 8 ? (unsigned)(wave + NW * lane) * 4u : ULTR_OOB);
    {
      const int i = lane & 15, q = lane >> 4;
      const float* pa = P1 + q * ld + 64 * mblk + 4 * i;
      const float* pb = P2 + q * ldf + 32 * nhalf + 2 * i;
      const int nk = R >> 2;
      float4 av = ld4(pa);
      float2 bv = ld2(pb);
      for (int kk = 0; kk < nk; ++kk) {
        const int kn = kk + 1 < nk ? kk + 1 : kk;
        const float4 an = ld4(pa + 4 * kn * ld);
        const float2 bn = ld2(pb + 4 * kn * ldf);
        const float aa[4] = {av.x, av.y, av.z, av.w};
        const float bb[2] = {bv.x, bv.y};
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
          for (int tb = 0; tb < 2; ++tb) accW2[ta][tb] = mfma16(aa[ta], bb[tb], accW2[ta][tb]);
        av = an;
        bv = bn;
      }
    }
    lds_barrier();  // d x_0 (fp32) and h0 have been read
    {
      const int i = lane & 15, q = lane >> 4;
      const float4 o4 = ld4(OS + 16 * rt + 4 * q);
      const float o[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * rt + 4 * q + r, rc = row < R ? row : R;
        *reinterpret_cast<float2*>(P2 + rc * ldf + 32 * chf + 2 * i) = make_float2(accf[0][0][r] * o[r], accf[0][1][r] * o[r]);
      }
      // xh = (xg - mean) rstd into P1 (columns past F keep d x_0's values: finite, never stored)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = wave + NW * k, rc = r < R ? r : R;
        const float m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mv), k));
        const float rr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rv), k));
        if (4 * lane < F) st4(P1 + rc * ld + 4 * lane, make_float4((xg[k].x - m) * rr, (xg[k].y - m) * rr, (xg[k].z - m) * rr, (xg[k].w - m) * rr));
      }
    }
    lds_barrier();
    // ---- the ReLU mask: d h0 back to its fp32 rows, its plane pair over d x_0's ----------------------------------------------------------
    {
      _Float16* FH = reinterpret_cast<_Float16*>(P0);
      _Float16* FL = FH + (R + 1) * ldf;
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const int row = 4 * (wave + NW * q2) + (lane >> 4), rc = row < R ? row : R;
        float4 v = ld4(P2 + rc * ldf + 4 * (lane & 15));
        const unsigned mk = hmask >> (4 * q2);
        v.x = (mk & 1u) ? v.x : 0.f;
        v.y = (mk & 2u) ? v.y : 0.f;
        v.z = (mk & 4u) ? v.z : 0.f;
        v.w = (mk & 8u) ? v.w : 0.f;
        st4(P2 + rc * ldf + 4 * (lane & 15), v);
        cb1.x += v.x; cb1.y += v.y; cb1.z += v.z; cb1.w += v.w;
        const float am = row16_max(max4(v));
        float sc, inv;
        fb_h3_scale(am, sc, inv);
        fbh4 hi, lo;
        fb_h3_split4(v, sc, hi, lo);
        *reinterpret_cast<fbh4*>(FH + rc * ldf + 4 * (lane & 15)) = hi;
        *reinterpret_cast<fbh4*>(FL + rc * ldf + 4 * (lane & 15)) = lo;
        if ((lane & 15) == 0) OS[64 + row] = inv * (1.0f / ULTR_H3_WSCALE);
      }
    }
    lds_barrier();
    // ---- d xn0 = d h0 W1, folded on the spot into d g_in / d b_in; the NEXT tile's rows are requested; d W1' += d h0^T xh ---------------
    {
      const int ch = wave;
      const bool has = ch < nchF;
      const int i = lane & 15, q = lane >> 4;
      const _Float16* AH = reinterpret_cast<const _Float16*>(P0);
      const int lo_off = (R + 1) * ldf;
      const _Float16* pa[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row = 16 * t + i;
        pa[t] = AH + (row < R ? row : R) * ldf + 8 * q;
      }
      const Src Wh = make_src(reinterpret_cast<const float*>(planes + a.gt1), (int64_t)dff * 32 * nchF);
      PipeH3W<4, 2> ph;
      ph.begin(Wh, ch, dff >> 5, 0, dff >> 5, has, lane);
      f32x4 acc[4][2];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) acc[t][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      ph.run(pa, lo_off, Wh, has ? (dff >> 5) : 0, acc);
      const int col = 32 * ch + 2 * i;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float4 o4 = ld4(OS + 64 + 16 * t + 4 * q);
        const float o[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * t + 4 * q + r;
          const float2 xh = ld2(P1 + (row < R ? row : R) * ld + col);
          const float y0 = row < vr ? acc[t][0][r] * o[r] : 0.f, y1 = row < vr ? acc[t][1][r] * o[r] : 0.f;
          cgi[0] = fmaf(y0, xh.x, cgi[0]);
          cgi[1] = fmaf(y1, xh.y, cgi[1]);
          cbi[0] += y0;
          cbi[1] += y1;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    request(tile + (int)gridDim.x);
    __builtin_amdgcn_sched_barrier(0);
    wgrad_thin_wide(accW1, P2, P1, R, wave, lane);
    lds_barrier();
  }
  // ---- the workgroup's partial -----------------------------------------------------------------------------------------------------------
  float* pw = ws + a.part + (int64_t)blockIdx.x * a.part_stride;
  const int oW1 = 2 * F, oB1 = oW1 + dff * F, oW2 = oB1 + dff, oB2 = oW2 + d * dff;
  {
    const int i = lane_id & 15, q = lane_id >> 4;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 64 * mblk + 4 * (4 * q + r) + ta;
        *reinterpret_cast<float2*>(pw + oW2 + m * dff + 32 * nhalf + 2 * i) = make_float2(accW2[ta][0][r], accW2[ta][1][r]);
      }
  }
  // column sums through LDS: [8 waves][d] d b2 | [32][dff] d b1 | [8 waves][4 q][2 x 32] d g_in, d b_in
  float* sB2 = smem;
  float* sB1 = sB2 + NW * d;
  float* sGI = sB1 + 4 * NW * dff;
  float* sFin = sGI + NW * 4 * 64;  // [dff] d b1 of the workgroup, [2][256] d g_in | d b_in
  st4(sB2 + wave * d + 4 * lane_id, cb2);
  st4(sB1 + (wave * 4 + (lane_id >> 4)) * dff + 4 * (lane_id & 15), cb1);
  {
    const int i = lane_id & 15, q = lane_id >> 4;
    float* g = sGI + (wave * 4 + q) * 64;
    *reinterpret_cast<float2*>(g + 2 * i) = make_float2(cgi[0], cgi[1]);
    *reinterpret_cast<float2*>(g + 32 + 2 * i) = make_float2(cbi[0], cbi[1]);
  }
  lds_barrier();
  for (int e = tid; e < d; e += NT) {
    float t = sB2[e];
#pragma unroll
    for (int w = 1; w < NW; ++w) t += sB2[w * d + e];
    pw[oB2 + e] = t;
  }
  if (tid < dff) {
    float t = sB1[tid];
#pragma unroll
    for (int k = 1; k < 4 * NW; ++k) t += sB1[k * dff + tid];
    pw[oB1 + tid] = t;
    sFin[tid] = t;
  }
  {
    // thread = (which: 0 g_in / 1 b_in, column c = 32 wave' + j): sum over the four row groups q in order
    const int which = tid >> 8, c = tid & 255, wv = c >> 5, j = c & 31;
    const float* g = sGI + wv * 4 * 64 + 32 * which + j;
    const float t = ((g[0] + g[64]) + g[128]) + g[192];
    if (c < F) pw[which * F + c] = t;
  }
  lds_barrier();
  {
    // d W1 = g_in[c] S[m][c] + b_in[c] d b1[m]
    const int i = lane_id & 15, q = lane_id >> 4;
    const int c = 32 * wave + 2 * i;
    if (c < F) {
      const float g0 = params[a.g_in + c], g1 = params[a.g_in + c + 1], b0 = params[a.b_in + c], b1 = params[a.b_in + c + 1];
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 4 * (4 * q + r) + ta;
          const float db = sFin[m];
          *reinterpret_cast<float2*>(pw + oW1 + m * F + c) = make_float2(fmaf(g0, accW1[ta][0][r], b0 * db), fmaf(g1, accW1[ta][1][r], b1 * db));
        }
    }
  }
}

size_t tile_lds_floats(int R) {
  const size_t tile = (size_t)(R + 1) * (2 * (SR_BWD_D + 8) + (SR_BWD_DFF + 8)) + 128 + 2 * SR_BWD_D;
  const size_t tail = (size_t)NW * 3 * SR_BWD_D + (size_t)4 * NW * SR_BWD_DFF;
  return tile > tail ? tile : tail;
}
}  // namespace

// whole rounds of one workgroup per compute unit: as few rounds as 64-row tiles allow, rows per tile a multiple of 4 (the
// weight-gradient products step over 4 rows)
bool sr_bwd_geometry(int64_t T, int cus, int* R, int* ntiles, int* nwg) {
  if (T <= 0 || T > 0x7fffffff / (4 * SR_BWD_D)) return false;
  int n = cus > 0 ? cus : 256;
  if (n > SR_BWD_MAXWG) n = SR_BWD_MAXWG;
  const int64_t rounds = (T + (int64_t)n * 64 - 1) / ((int64_t)n * 64);
  int64_t r = (T + n * rounds - 1) / (n * rounds);
  r = (r + 3) / 4 * 4;
  if (r < 4) r = 4;
  if (r > 64) r = 64;
  *R = (int)r;
  *ntiles = (int)((T + r - 1) / r);
  *nwg = *ntiles < n ? *ntiles : n;
  return true;
}

int sr_bwd_ffn_launch(SrBwdFfnArgs a, int nwg, const float* params, const _Float16* planes, const float* sv, float* ws, hipStream_t st) {
  if (a.d != SR_BWD_D || a.dff != SR_BWD_DFF || a.R < 4 || a.R > 64 || (a.R & 3) || nwg <= 0 || nwg > SR_BWD_MAXWG) return ULTR_E_UNSUPPORTED;
  a.p0 = 0;
  a.p1 = (a.R + 1) * (SR_BWD_D + 8);
  a.p2 = 2 * a.p1;
  a.os = a.p2 + (a.R + 1) * (SR_BWD_DFF + 8);
  const size_t lds = tile_lds_floats(a.R) * sizeof(float);
  const int rc = set_lds(sr_bwd_ffn_kernel, lds);
  if (rc != 0) return rc;
  hipLaunchKernelGGL(sr_bwd_ffn_kernel, dim3(nwg), dim3(NT), lds, st, a, params, planes, sv, ws);
  return (int)hipGetLastError();
}
int sr_bwd_proj_launch(SrBwdProjArgs a, int nwg, const float* params, const _Float16* planes, const float* sv, float* ws, hipStream_t st) {
  if (a.d != SR_BWD_D || a.dff != SR_BWD_DFF || a.R < 4 || a.R > 64 || (a.R & 3) || nwg <= 0 || nwg > SR_BWD_MAXWG) return ULTR_E_UNSUPPORTED;
  a.p0 = 0;
  a.p1 = (a.R + 1) * (SR_BWD_D + 8);
  a.pf = 2 * a.p1;
  a.os = a.pf + (a.R + 1) * (SR_BWD_DFF + 8);
  const size_t lds = tile_lds_floats(a.R) * sizeof(float);
  const int rc = set_lds(sr_bwd_proj_kernel, lds);
  if (rc != 0) return rc;
  hipLaunchKernelGGL(sr_bwd_proj_kernel, dim3(nwg), dim3(NT), lds, st, a, params, planes, sv, ws);
  return (int)hipGetLastError();
}

int sr_bwd_head_launch(SrBwdHeadArgs a, int nwg, const float* params, const _Float16* planes, const float* sv, const float* dscores, float* ws,
                       hipStream_t st) {
  if (a.d != SR_BWD_D || a.dff != SR_BWD_DFF || a.R < 4 || a.R > 64 || (a.R & 3) || nwg <= 0 || nwg > SR_BWD_MAXWG) return ULTR_E_UNSUPPORTED;
  a.p0 = 0;
  a.pf = (a.R + 1) * (SR_BWD_DFF + 8);
  a.p1 = 2 * a.pf;
  a.os = a.p1 + (a.R + 1) * (SR_BWD_D + 8);
  size_t fl = (size_t)a.os + 64;
  const size_t tail = (size_t)4 * NW * (2 * SR_BWD_DFF + 4);
  if (fl < tail) fl = tail;
  const size_t lds = fl * sizeof(float);
  const int rc = set_lds(sr_bwd_head_kernel, lds);
  if (rc != 0) return rc;
  hipLaunchKernelGGL(sr_bwd_head_kernel, dim3(nwg), dim3(NT), lds, st, a, params, planes, sv, dscores, ws);
  return (int)hipGetLastError();
}

int sr_bwd_embed_launch(SrBwdEmbedArgs a, int nwg, const float* params, const _Float16* planes, const float* sv, float* ws, hipStream_t st) {
  if (a.d != SR_BWD_D || a.dff != SR_BWD_DFF || a.R < 4 || a.R > 64 || (a.R & 3) || nwg <= 0 || nwg > SR_BWD_MAXWG || a.F < 4 || a.F > SR_BWD_D ||
      (a.F & 3))
    return ULTR_E_UNSUPPORTED;
  a.p0 = 0;
  a.p1 = (a.R + 1) * (SR_BWD_D + 8);
  a.p2 = 2 * a.p1;
  a.os = a.p2 + (a.R + 1) * (SR_BWD_DFF + 8);
  size_t fl = (size_t)a.os + 128;
  const size_t tail = (size_t)NW * SR_BWD_D + (size_t)4 * NW * SR_BWD_DFF + (size_t)NW * 4 * 64 + SR_BWD_DFF + 8;
  if (fl < tail) fl = tail;
  const size_t lds = fl * sizeof(float);
  const int rc = set_lds(sr_bwd_embed_kernel, lds);
  if (rc != 0) return rc;
  hipLaunchKernelGGL(sr_bwd_embed_kernel, dim3(nwg), dim3(NT), lds, st, a, params, planes, sv, ws);
  return (int)hipGetLastError();
}
