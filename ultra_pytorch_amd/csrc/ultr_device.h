// ultr_device.h — device-side helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ULTR_LN_EPS 1e-5f       // nn.LayerNorm default eps (reference DNN.py:46)
#define ULTR_PAD_SCORE -100000.0f  // BaseAlgorithm.PADDING_SCORE (base_algorithm.py:36)

// v_mfma_f32_16x16x4_f32: D[16x16] += A[16x4] * B[4x16], exact fp32 (an fmaf chain over k).
// lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15];
// lane l receives D[row = 4*(l >> 4) + r][col = l & 15] in acc[r].
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// activation and its derivative expressed through the activation OUTPUT a = act(z)
//   elu  (alpha 1): a = z > 0 ? z : expm1(z)   (torch.s CPU ELU kernel uses expm1 - verified numerically);
//                   act'(z) = z > 0 ? 1 : exp(z) = a + 1
//   relu          : act'(z) = a > 0
__device__ __forceinline__ float act_fwd(float z, int act) {
  if (act == 0) return z > 0.0f ? z : expm1f(z);
  return fmaxf(z, 0.0f);
}
__device__ __forceinline__ float act_grad_from_out(float a, int act) {
  if (act == 0) return a > 0.0f ? 1.0f : (a + 1.0f);
  return a > 0.0f ? 1.0f : 0.0f;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// masked 4-wide row load: elements [c, c+4) of a row of length `len`; vec => 16-byte aligned fast path
__device__ __forceinline__ float4 ld4_masked(const float* row, int c, int len, bool vec) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row == nullptr || c >= len) return v;
  if (vec && c + 3 < len) return ld4(row + c);
  v.x = row[c];
  if (c + 1 < len) v.y = row[c + 1];
  if (c + 2 < len) v.z = row[c + 2];
  if (c + 3 < len) v.w = row[c + 3];
  return v;
}

__host__ __device__ __forceinline__ int round_up(int x, int m) { return (x + m - 1) / m * m; }
