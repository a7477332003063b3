// ultr_sr_tiles.h - what SetRank's persistent fused launches share (ultr_sr_bwd.hip: the backward's row-local chains; ultr_sr_fwd.hip: the
// encoder block's forward): eight waves per workgroup, one workgroup per compute unit walking over tiles of R <= 64 token rows, the
// d-wide split-half product with four 16-row tiles behind one weight stream per wave.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ultr_hip.h"
#include "ultr_device.h"
#include "ultr_h3.h"
#include "ultr_plan.h"
#include "ultr_sr_bwd.h"

namespace {

constexpr int NW = 8, NT = NW * 64;

__device__ __forceinline__ float row16_max(float v) {  // maximum over the 16 lanes of a DPP row, in every lane of the row
  ULTR_DPP_MAX(v, "quad_perm:[1,0,3,2] row_mask:0xf");
  ULTR_DPP_MAX(v, "quad_perm:[2,3,0,1] row_mask:0xf");
  ULTR_DPP_MAX(v, "row_ror:4 row_mask:0xf");
  ULTR_DPP_MAX(v, "row_ror:8 row_mask:0xf");
  return v;
}
__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ float max4(float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }

// LayerNorm backward of four rows per wave (rows wave + 8 (4 half + k), lane = columns 4 lane .. + 3; d = 256):
//   xh = (s - mean) rstd,  g = dy gamma,  v = rstd (g - mean(g) - xh mean(g xh))         (same expressions as sr_ln_bwd_cs_v4_kernel)
// v goes to the fp16 plane pair in P0 scaled per row (OS[row] = the inverse scale x 2^-8: the weights are stored x 2^8) and, by MODE,
//   MODE 0: to P1 as fp32 rows (the residual and the weight-gradient operand of the FFN kernel)
//   MODE 1: to global memory (d s1), while P1 receives out1 = xh gamma + beta (the weight-gradient operand of the projection kernel)
// Column sums: cg += dy xh, cb += dy, cd += v.
// a d-wide split-half product over the tile's four 16-row tiles: wave = one 32-column chunk of the output, the A operand = the fp16
// plane pair at Ap (row stride lda halves), the weights = fragment copy at planes + gw (nks steps of 32 along the contraction);
// Y = acc x os[row] (+ bias[col]) (+ the fp32 row in Pres; TO_LDS: the sum goes back into that row - the residual stream of the forward);
// rows past the tile's valid rows are dropped by the destination's extent
template <bool RES, bool BIAS = false, bool TO_LDS = false>
__device__ __forceinline__ void product_d4(int wave, int lane, int R, const float* Ap, int lda, int nks, const _Float16* planes,
                                           int64_t gw, int Kw, const float* os, float* Pres, int ld, const Dst& dout,
                                           const float* bias = nullptr) {
  constexpr int d = SR_BWD_D;
  asm volatile("" : "+v"(lane));
  const int ch = wave;
  const int i = lane & 15, q = lane >> 4;
  const _Float16* AH = reinterpret_cast<const _Float16*>(Ap);
  const int lo_off = (R + 1) * lda;
  const _Float16* pa[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int row = 16 * t + i;
    pa[t] = AH + (row < R ? row : R) * lda + 8 * q;
  }
  const Src Wh = make_src(reinterpret_cast<const float*>(planes + gw), (int64_t)Kw * d);
  PipeH3W<4, 2> ph;
  ph.begin(Wh, ch, nks, 0, nks, true, lane);
  f32x4 acc[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) acc[t][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  ph.run(pa, lo_off, Wh, nks, acc);
  const int col = 32 * ch + 2 * i;
  const unsigned gv = (unsigned)(4 * q * d + 2 * i) * 4u;
  float2 bv = make_float2(0.f, 0.f);
  if constexpr (BIAS) bv = ld2(bias + col);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float4 o4 = ld4(os + 16 * t + 4 * q);
    const float o[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * t + 4 * q + r;
      float2 y = make_float2(acc[t][0][r] * o[r], acc[t][1][r] * o[r]);
      if constexpr (BIAS) {
        y.x += bv.x;
        y.y += bv.y;
      }
      if constexpr (RES) {
        float2* pr = reinterpret_cast<float2*>(Pres + (row < R ? row : R) * ld + col);
        const float2 rv = *pr;
        y.x += rv.x;
        y.y += rv.y;
        if constexpr (TO_LDS) *pr = y;
      }
      buf_st2(dout, gv, (unsigned)((16 * t + r) * d + 32 * ch) * 4u, y);
    }
  }
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
  if (bytes > 160 * 1024) return ULTR_E_UNSUPPORTED;
  if (bytes > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
    return ULTR_E_UNSUPPORTED;
  return 0;
}


}  // namespace
