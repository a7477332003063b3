// ultr_h3.h - split-half (fp16 hi / lo) product building blocks shared by the DNN kernels (ultr_dnn.hip) and SetRank's fused block
// kernel (ultr_setrank.hip): operand split, the wide-tile weight pipeline PipeH3W, stores through buffer resources.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "ultr_device.h"

typedef _Float16 fbh8 __attribute__((ext_vector_type(8)));
typedef _Float16 fbh4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 fb_mfma_h(fbh8 a, fbh8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ fbh8 fb_as_h8(float4 v) {
  union { float4 f; fbh8 h; } u;
  u.f = v;
  return u.h;
}
// power-of-two scale that brings a row's largest magnitude just below 2^14 (fp16 overflows at 65504), and its inverse
__device__ __forceinline__ void fb_h3_scale(float amax, float& rs, float& inv) {
  int se = 267 - (int)((__float_as_uint(amax) >> 23) & 0xffu);  // amax in [2^(E-127), 2^(E-126)): amax * 2^(se-127) < 2^14
  se = se < 1 ? 1 : (se > 253 ? 253 : se);
  rs = __uint_as_float((unsigned)se << 23);
  inv = __uint_as_float((unsigned)(254 - se) << 23);
}
__device__ __forceinline__ void fb_h3_split4(float4 v, float rs, fbh4& hi, fbh4& lo) {
  const float a[4] = {v.x * rs, v.y * rs, v.z * rs, v.w * rs};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    hi[k] = (_Float16)a[k];
    lo[k] = (_Float16)(a[k] - (float)hi[k]);
  }
}
#ifndef FWDW_DEPTH
#define FWDW_DEPTH 2  // weight steps (4 KiB per wave) in flight per wave
#endif
template <int RT, int D>
struct PipeH3W {
  float4 b[D][4];
  unsigned of;
  int left;
  template <int S>
  __device__ __forceinline__ void fetch(const Src& W) {
    const bool ok = left > 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) b[S][u] = buf_ld4(W, ok ? (of + (unsigned)u * 1024u) : ULTR_OOB);
    --left;
    of += 4096u;
  }
  // steps [ks0, ks0 + cnt) of chunk `chunk` (nks steps per chunk in the copy)
  __device__ __forceinline__ void begin(const Src& W, int chunk, int nks, int ks0, int cnt, bool valid, int lane) {
    of = (((unsigned)chunk * (unsigned)nks + (unsigned)ks0) * 256u + (unsigned)lane) * 16u;
    left = valid ? cnt : 0;
    fetch<0>(W);
    if constexpr (D > 2) fetch<1>(W);
    static_assert(D >= 2 && D <= 3, "pipeline depth");
  }
  // pa[rt]: this lane's A row of row tile rt in the hi plane (+ lo_off halves: the lo plane), at the current step.
  // ONE accumulator per output tile: ah.wh, ah.wl and al.wh are added into it in three sweeps over the RT x 2 tiles (a dependent
  // MFMA is 2 .. 4 instructions behind its predecessor).  The 8-wave kernels keep the cross terms in their own accumulators; here
  // 16 waves share the register file (128 per wave) and a second accumulator set is what spilled (each reload a memory round trip
  // in the middle of the stream).  The sum is the same three products in fp32; scores move by < 1e-6 (tests/test_gpu_wide_fwd.py).
  template <int S>
  __device__ __forceinline__ void consume(const _Float16* const (&pa)[RT], int lo_off, int kofs, f32x4 (&acc)[RT][2]) {
#pragma unroll
    for (int r0 = 0; r0 < RT; r0 += 2) {
      constexpr int NP = 2;
      fbh8 ah[NP], al[NP];
#pragma unroll
      for (int d = 0; d < NP; ++d)
        if (r0 + d < RT) {
          ah[d] = *reinterpret_cast<const fbh8*>(pa[r0 + d] + kofs);
          al[d] = *reinterpret_cast<const fbh8*>(pa[r0 + d] + lo_off + kofs);
        }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int d = 0; d < NP; ++d)
          if (r0 + d < RT) acc[r0 + d][t] = fb_mfma_h(al[d], fb_as_h8(b[S][2 * t]), acc[r0 + d][t]);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int d = 0; d < NP; ++d)
          if (r0 + d < RT) acc[r0 + d][t] = fb_mfma_h(ah[d], fb_as_h8(b[S][2 * t + 1]), acc[r0 + d][t]);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int d = 0; d < NP; ++d)
          if (r0 + d < RT) acc[r0 + d][t] = fb_mfma_h(ah[d], fb_as_h8(b[S][2 * t]), acc[r0 + d][t]);
    }
  }
  __device__ __forceinline__ void run(const _Float16* const (&pa)[RT], int lo_off, const Src& W, int cnt, f32x4 (&acc)[RT][2]) {
    int kofs = 0;
    auto step = [&](auto uc) {
      constexpr int U = decltype(uc)::value;
      fetch<(U + D - 1) % D>(W);
      __builtin_amdgcn_sched_barrier(0);
      consume<U>(pa, lo_off, kofs, acc);
      kofs += 32;
    };
    int t = 0;
    for (; t + D <= cnt; t += D) {
      step(std::integral_constant<int, 0>());
      step(std::integral_constant<int, 1>());
      if constexpr (D > 2) step(std::integral_constant<int, 2>());
    }
    if (t < cnt) { consume<0>(pa, lo_off, kofs, acc); kofs += 32; }
    if constexpr (D > 2) if (t + 1 < cnt) { consume<1>(pa, lo_off, kofs, acc); kofs += 32; }
  }
};

// stores through a buffer resource: the address is (wave-uniform descriptor) + a 32-bit lane offset + a scalar offset, and rows past
// the end of the described extent are dropped by the hardware range check - no 64-bit address per unrolled store, no branches
// around the stores of a ragged tile (per-store 64-bit addresses + their spill reloads were 36k of this kernel's first 147k cycles)
struct Dst {
  __amdgpu_buffer_rsrc_t rs;
};
__device__ __forceinline__ Dst make_dst(float* base, int64_t nfloats) {
  Dst d;
  d.rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(nfloats * 4), 0x00020000);
  return d;
}
__device__ __forceinline__ void buf_st4(const Dst& d, unsigned voff, unsigned soff, float4 v) {
  const u32x4 x = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(x, d.rs, voff, soff, 0);
}
__device__ __forceinline__ void buf_st2(const Dst& d, unsigned voff, unsigned soff, float2 v) {
  const u32x2 x = {__float_as_uint(v.x), __float_as_uint(v.y)};
  __builtin_amdgcn_raw_buffer_store_b64(x, d.rs, voff, soff, 0);
}
__device__ __forceinline__ void buf_st1(const Dst& d, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), d.rs, voff, soff, 0);
}

