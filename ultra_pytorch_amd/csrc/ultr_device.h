// ultr_device.h — device-side helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define ULTR_LN_EPS 1e-5f       // nn.LayerNorm default eps (reference DNN.py:46)
#define ULTR_PAD_SCORE -100000.0f  // BaseAlgorithm.PADDING_SCORE (base_algorithm.py:36)

// v_mfma_f32_16x16x4_f32: D[16x16] += A[16x4] * B[4x16], exact fp32 (an fmaf chain over k).
// lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15];
// lane l receives D[row = 4*(l >> 4) + r][col = l & 15] in acc[r].
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Wavefront (64-lane) reductions on the DPP path: VALU cross-lane moves, no LDS round trips (hipcc lowers
// __shfl_xor to ds_bpermute_b32, ~100+ cycles per dependent step).  quad_perm [1,0,3,2], quad_perm [2,3,0,1],
// row_ror:4, row_ror:8 leave every lane of a 16-lane row with the row total; row_bcast:15 / row_bcast:31 chain
// the four rows into lane 63, which is read back as a wave-uniform scalar.  Fixed order -> deterministic.
template <int CTRL>
__device__ __forceinline__ float dpp_or(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_or<0xb1>(0.f, v);
  v += dpp_or<0x4e>(0.f, v);
  v += dpp_or<0x124>(0.f, v);
  v += dpp_or<0x128>(0.f, v);
  v += dpp_or<0x142>(0.f, v);
  v += dpp_or<0x143>(0.f, v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// N independent sums at once: the six dependent DPP steps of the N chains interleave (a single chain leaves the
// VALU idle for most of each step's latency)
template <int N>
__device__ __forceinline__ void wave_sum_n(float (&v)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += dpp_or<0xb1>(0.f, v[k]);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += dpp_or<0x4e>(0.f, v[k]);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += dpp_or<0x124>(0.f, v[k]);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += dpp_or<0x128>(0.f, v[k]);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += dpp_or<0x142>(0.f, v[k]);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += dpp_or<0x143>(0.f, v[k]);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[k]), 63));
}
// wave maximum: ONE instruction per step (v_max_f32_dpp).  Written as asm because the compiler cannot fold a maximum into the DPP move
// (fmaxf(v, dpp(v)) became v_mov + s_nop + v_mov_dpp + a canonicalising v_max + v_max: five issue slots per step, 60 instead of
// 12 for the two interleaved chains of a LayerNorm - on kernels that are instruction-issue bound with two waves per SIMD).  The
// s_nop supplies the two wait states a DPP read of a just-written VGPR needs; lanes without a source (row_bcast) keep their value.
#define ULTR_DPP_MAX(x, ctrl) asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 " ctrl " bank_mask:0xf" : "+v"(x))
template <int N>
__device__ __forceinline__ void wave_max_n(float (&v)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) ULTR_DPP_MAX(v[k], "quad_perm:[1,0,3,2] row_mask:0xf");
#pragma unroll
  for (int k = 0; k < N; ++k) ULTR_DPP_MAX(v[k], "quad_perm:[2,3,0,1] row_mask:0xf");
#pragma unroll
  for (int k = 0; k < N; ++k) ULTR_DPP_MAX(v[k], "row_ror:4 row_mask:0xf");
#pragma unroll
  for (int k = 0; k < N; ++k) ULTR_DPP_MAX(v[k], "row_ror:8 row_mask:0xf");
#pragma unroll
  for (int k = 0; k < N; ++k) ULTR_DPP_MAX(v[k], "row_bcast:15 row_mask:0xa");
#pragma unroll
  for (int k = 0; k < N; ++k) ULTR_DPP_MAX(v[k], "row_bcast:31 row_mask:0xc");
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[k]), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  float a[1] = {v};
  wave_max_n<1>(a);
  return a[0];
}

// activation and its derivative expressed through the activation OUTPUT a = act(z)
//   elu  (alpha 1): a = z > 0 ? z : expm1(z)   (torch.s CPU ELU kernel uses expm1 - verified numerically);
//                   act'(z) = z > 0 ? 1 : exp(z) = a + 1
//   relu          : act'(z) = a > 0
// expm1 for z <= 0 in ~14 VALU instructions, both sides evaluated (no divergence): near zero the degree-8 Taylor
// polynomial (|z| <= 0.35: truncation 0.35^8/9! relative = 4e-10), elsewhere exp(z) - 1 where the subtraction is
// benign (|result| >= 0.29) and v_exp_f32's ~1 ulp is far inside the parity bar.  OCML's expm1f is ~3x the
// instructions; at 4 cycles of SIMD time per wave64 VALU instruction the ELU epilogue was ~1k cycles per layer.
__device__ __forceinline__ float expm1_neg(float z) {
  float p = 2.4801587e-5f;          // 1/8!
  p = fmaf(p, z, 1.9841270e-4f);    // 1/7!
  p = fmaf(p, z, 1.3888889e-3f);    // 1/6!
  p = fmaf(p, z, 8.3333333e-3f);    // 1/5!
  p = fmaf(p, z, 4.1666667e-2f);    // 1/4!
  p = fmaf(p, z, 1.6666667e-1f);    // 1/3!
  p = fmaf(p, z, 0.5f);
  p = fmaf(p, z, 1.0f);
  p *= z;
  const float e = __expf(z) - 1.0f;
  return z > -0.35f ? p : e;
}
//   tanh          : a = tanh z; act' = 1 - a^2.  |z| < 0.25: odd Taylor polynomial to z^7 (truncation 62/2835 z^9 < 8e-8
//                   relative); elsewhere 1 - 2 / (e^{2z} + 1) with v_exp_f32 / v_rcp_f32 (~1 ulp each, no cancellation there)
//   sigmoid       : a = 1 / (1 + e^{-z}); act' = a (1 - a)
// `act` is a kernel argument (wave-uniform): the chain below is scalar branches, elu first.
__device__ __forceinline__ float act_fwd(float z, int act) {
  if (act <= 1) {  // elu / relu share one select: the negative branch is expm1(z) or 0 (wave-uniform factor)
    const float neg = expm1_neg(z) * (act == 0 ? 1.0f : 0.0f);
    return z > 0.0f ? z : neg;
  }
  if (act == 2) {
    const float z2 = z * z;
    float p = -5.3968254e-2f;            // -17/315
    p = fmaf(p, z2, 1.3333333e-1f);      // 2/15
    p = fmaf(p, z2, -3.3333333e-1f);     // -1/3
    p = fmaf(p, z2, 1.0f) * z;
    const float zc = fminf(fmaxf(z, -15.0f), 15.0f);
    const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * zc) + 1.0f);
    return fabsf(z) < 0.25f ? p : t;
  }
  const float zc = fminf(fmaxf(z, -30.0f), 30.0f);
  return __builtin_amdgcn_rcpf(1.0f + __expf(-zc));
}
// One branch-free form for all four: act'(z) from the OUTPUT a is  a > thr ? 1 : c0 + a (c1 + c2 a)  with wave-uniform
// constants (elu: thr 0, 1 + a; relu: thr 0, 0; tanh: never, 1 - a^2; sigmoid: never, a - a^2).  `act` is a kernel argument:
// the constants are scalar selects hoisted out of every loop, the per-element cost is two fmas, a compare and a select
// (an if-chain over four formulas in the backward row passes cost config 3 8 us per step).
__device__ __forceinline__ float act_grad_from_out(float a, int act) {
  const float thr = act < 2 ? 0.0f : __builtin_huge_valf();
  const float c0 = (act == 0 || act == 2) ? 1.0f : 0.0f;
  const float c1 = (act == 0 || act == 3) ? 1.0f : 0.0f;
  const float c2 = act >= 2 ? -1.0f : 0.0f;
  return a > thr ? 1.0f : fmaf(a, fmaf(a, c2, c1), c0);
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// Streaming stores for data that the NEXT launch consumes (never this one): `nt` (global_store ... nt).  A kernel boundary
// writes back every dirty L2 line before the dependent launch starts (each XCD has its own L2); bulk outputs that leave
// with streaming stores are already on their way while the kernel still computes, so the boundary has less to flush
// (measured at config 2: the fused kernel's 10 MB of weight-gradient operands, step 55.4 -> 53.6 us).
__device__ __forceinline__ void st4_stream(float* p, float4 v) {
  const f32x4 x = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(p));
}
__device__ __forceinline__ void st1_stream(float* p, float v) { __builtin_nontemporal_store(v, p); }
// bulk kernel OUTPUTS that only later launches read (saved activations, dz, GEMM results): plain stores (streaming stores lose at the
// larger configs: config 4 437 -> 454 us, config 5 3.32 -> 3.34 ms - the consumer wants them in L2 / the memory-side cache)
__device__ __forceinline__ void st2_out(float* p, float2 v) {
  *reinterpret_cast<float2*>(p) = v;
}
__device__ __forceinline__ void st4_out(float* p, float4 v) {
  st4(p, v);
}

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

// Branch-free masked row loads for the hot loops.
// hipcc only emits COUNTED s_waitcnt vmcnt(N) (i.e. keeps a software prefetch ring in flight) when the loads are
// not wrapped in control flow; an `if` around a load - or a select on its result, which CodeGenPrepare turns back
// into a branch - makes every use drain vmcnt(0) (measured: prefetch depth 1 -> 8 changed nothing).  So the VEC
// path goes through a buffer resource (SRSRC): the hardware bounds check returns 0 for an out-of-range offset,
// and masked lanes simply present ULTR_OOB.  Descriptors are built from kernel arguments only (wave-uniform).
// VEC == false is the generic (unaligned / ragged) path and keeps the masked scalar loads.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define ULTR_OOB 0x80000000u  // > any buffer we describe (sizes are checked < 2 GiB on the host)

struct Src {
  const float* base;
  __amdgpu_buffer_rsrc_t rs;
};
__device__ __forceinline__ Src make_src(const float* base, int64_t nfloats) {
  Src s;
  s.base = base;
  s.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(nfloats * 4), 0x00020000);
  return s;
}
template <bool VEC>
__device__ __forceinline__ float4 ld4_sel(const Src& s, int64_t off, bool ok, int c, int len) {
  if constexpr (VEC) {
    const unsigned bo = (ok && c < len) ? (unsigned)((off + c) * 4) : ULTR_OOB;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(s.rs, bo, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  } else {
    return ld4_masked(ok ? (s.base + off) : nullptr, c, len, false);
  }
}
// unmasked forms for operands whose out-of-range elements only ever meet a zero on the other side of the MFMA
// (or produce outputs that are discarded): past-the-end offsets are zeroed by the hardware bounds check
__device__ __forceinline__ float4 buf_ld4(const Src& s, unsigned byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(s.rs, byte_off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// lane offset + a wave-uniform (scalar register) offset: unrolled loads of several rows share ONE offset register per lane; the
// range check covers the sum (gfx950: the ragged tiles of the fused SetRank kernels rely on it, tests/test_gpu_setrank.py)
__device__ __forceinline__ float4 buf_ld4s(const Src& s, unsigned lane_off, unsigned uniform_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(s.rs, lane_off, uniform_off, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float buf_ld1(const Src& s, unsigned byte_off) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(s.rs, byte_off, 0, 0));
}
// Agent-scope coherent accesses (cache policy sc1): a store is written through to the device coherence point and a
// load is served from it, so two workgroups on different XCDs (each XCD has its own L2) can hand data over inside ONE
// launch without the L2-wide writeback + invalidate of a device-scope fence (measured: __threadfence() in every
// workgroup of the wgrad kernel took it from 16 us to 89 us).
#define ULTR_SC1 0x10
__device__ __forceinline__ void coh_st4(const Src& s, unsigned byte_off, float4 v) {
  const u32x4 d = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(d, s.rs, byte_off, 0, ULTR_SC1);
}
__device__ __forceinline__ void coh_st1(const Src& s, unsigned byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), s.rs, byte_off, 0, ULTR_SC1);
}
__device__ __forceinline__ float4 coh_ld4(const Src& s, unsigned byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(s.rs, byte_off, 0, ULTR_SC1);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float coh_ld1(const Src& s, unsigned byte_off) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(s.rs, byte_off, 0, ULTR_SC1));
}

template <bool VEC>
__device__ __forceinline__ float ld1_sel(const Src& s, int64_t idx, bool ok) {
  if constexpr (VEC) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(s.rs, ok ? (unsigned)(idx * 4) : ULTR_OOB, 0, 0));
  } else {
    return ok ? s.base[idx] : 0.f;
  }
}

// 1/sqrt(x): v_rsq_f32 (1 ulp) + one Newton step.  The IEEE sequence 1.0f / sqrtf(x) costs ~45 VALU instructions per
// value (two v_div_* fix-up chains); a wave64 VALU instruction occupies its SIMD for 4 cycles, so in the latency-bound
// LayerNorm phases instruction count is time.
__device__ __forceinline__ float rsqrt_nr(float x) {
  const float y = __builtin_amdgcn_rsqf(x);
  return y * (1.5f - 0.5f * x * y * y);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence + s_barrier and hipcc
// implements the fence as s_waitcnt vmcnt(0) lgkmcnt(0): every global store of an epilogue and every prefetched
// global load would be drained at each phase boundary.  The kernels here never communicate through global memory
// inside a launch, so waiting for the LDS queue is sufficient.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Canonical order of every cross-workgroup partial sum (slabs, loss partials): part k belongs to group k & 3; a group
// adds its parts in chunks of 8, ((v0+v1)+(v2+v3))+((v4+v5)+(v6+v7)) with missing parts = 0, chunks accumulate in
// order; total = ((S0+S1)+S2)+S3.  strided_sum = ONE group (4 cooperating threads + an LDS combine), full_sum = all four
// groups by one thread with 32 loads in flight: both produce the same bits.
// NC chunks (8 NC loads) are requested per trip and added in chunk order: the same bits for every NC.  The spare workgroups of the
// weight-gradient launch use NC = 4 - a quarter of the dependent round trips (their folds of 256 - 1024 partials ended after the
// launch's matrix blocks: config 3 wgrad 43.7 -> 41.7 us, config 4 84 -> 81, 10 240 rows x [256,256] 36 -> 31); the reduction kernels
// keep NC = 1 (the 32 live registers of NC = 4 cost them occupancy: config 4's reduction 10.1 -> 13.7 us).
template <int NC = 1>
__device__ __forceinline__ float strided_sum(const float* __restrict__ src, int64_t stride, int nparts, int grp) {
  float part = 0.f;
  for (int k = grp; k < nparts; k += 32 * NC) {
    float v[NC][8];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int kk = k + 32 * c + 4 * i;
        v[c][i] = (kk < nparts) ? src[(int64_t)kk * stride] : 0.f;
      }
#pragma unroll
    for (int c = 0; c < NC; ++c)
      if (k + 32 * c < nparts) part += ((v[c][0] + v[c][1]) + (v[c][2] + v[c][3])) + ((v[c][4] + v[c][5]) + (v[c][6] + v[c][7]));
  }
  return part;
}
__device__ __forceinline__ float full_sum(const float* __restrict__ src, int64_t stride, int nparts) {
  float S[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < nparts; k0 += 32) {
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = (k0 + i < nparts) ? src[(int64_t)(k0 + i) * stride] : 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      S[g] += ((v[g] + v[g + 4]) + (v[g + 8] + v[g + 12])) + ((v[g + 16] + v[g + 20]) + (v[g + 24] + v[g + 28]));
  }
  return ((S[0] + S[1]) + S[2]) + S[3];
}

// Counter-based generator (Philox-4x32-10): a draw is a pure function of (key, counter) - no state, no ordering
struct Philox {
  uint32_t k0, k1;
  __device__ __forceinline__ void round(uint32_t (&c)[4], uint32_t ka, uint32_t kb) const {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ ka, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ kb;
    c[1] = (uint32_t)p1;
    c[3] = (uint32_t)p0;
    c[0] = n0;
    c[2] = n2;
  }
  __device__ __forceinline__ void operator()(uint32_t (&c)[4]) const {
    uint32_t ka = k0, kb = k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      round(c, ka, kb);
      ka += 0x9E3779B9u;
      kb += 0xBB67AE85u;
    }
  }
};
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }  // [0, 1)

__device__ __forceinline__ float quad_max(float v) {  // over the 4 lanes {i, i+16, i+32, i+48}
  v = fmaxf(v, __shfl_xor(v, 16));
  return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float quad_sum(float v) {
  v += __shfl_xor(v, 16);
  return v + __shfl_xor(v, 32);
}
__host__ __device__ __forceinline__ int round_up(int x, int m) { return (x + m - 1) / m * m; }

// raise bits of a status / flag word other workgroups and later launches read (rare path: relaxed, agent scope)
__device__ __forceinline__ void flag_or(uint32_t* w, uint32_t bits) { __hip_atomic_fetch_or(w, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
