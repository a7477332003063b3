// ultr_dnn_kernels.h - what the translation units of the DNN ranking model share (ultr_dnn.hip: plans, knobs, entry points;
// ultr_dnn_fwd.hip, ultr_dnn_bwd.hip, ultr_dnn_fb.hip, ultr_dnn_wgrad.hip: the kernels and their launchers): LDS strides, the
// matrix-core GEMM building blocks (gemm_nt_chunk, gemm_nn, GemmPipe, PipeSw, PipeH3), small plan structs, the knobs, the launchers'
// declarations.  (Round 6: one 5 300-line unit became five.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <utility>

#include "../../include/ultr_hip.h"
#include "ultr_comm.h"
#include "ultr_device.h"
#include "ultr_h3.h"
#include "ultr_plan.h"
#include "ultr_prof.h"

// ------------------------------------------------------------------------------------------------
// LDS leading dimensions
// ------------------------------------------------------------------------------------------------
// Row stride of the LDS tiles that feed MFMA A-fragments: a multiple of 32 columns (zero padding read by the pipelined GEMM)
// plus ULTR_LD_PAD floats.  The 16 lanes of a float4 A read sit in 16 different rows: conflict-free when (stride / 4) is odd,
// i.e. stride = 4 (mod 8).  4 is the smallest such pad; it keeps a 512-wide forward tile pair + parameter image under 80 KB
// (two workgroups per CU at BASELINE config 3).
#ifndef ULTR_LD_PAD
#define ULTR_LD_PAD 4
#endif
// forward buffers: float4 epilogue stores / float4 A reads of the generic path -> ld % 4 == 0; rows padded so that
// the pipelined GEMM may read (masked) up to 31 columns past K
__host__ __device__ static inline int fwd_ld(int maxdim) { return round_up(maxdim, 32) + ULTR_LD_PAD; }
// dnn_fwd_kernel with split-half layers (DnnPlan::fwd_h3): the two fp16 planes of the A tile overlay the fp32 tile they were made
// from, with a row stride of round_up(maxdim, 32) + 8 halves (16-byte reads, 4 banks per row apart) - they fit once the fp32 row
// stride is that + 4 floats, which is odd in units of 4 floats just like the default (conflict-free float4 reads)
__host__ __device__ static inline int fwd_ldh(int maxdim) { return round_up(maxdim, 32) + 8; }
__host__ __device__ static inline int fwd_ld_of(int maxdim, int h3) { return h3 ? fwd_ldh(maxdim) + 4 : fwd_ld(maxdim); }
// backward dz buffer: float4 A-fragment reads, rows zero-padded to a multiple of 32 (see gemm_nn)
__host__ __device__ static inline int bwd_ldz(int maxdim) { return round_up(maxdim, 32) + ULTR_LD_PAD; }
// dnn_bwd2_kernel with split-half dgrad products (DnnPlan::bwd_h3): the dz tile holds two fp16 planes instead (see fwd_ld_of)
__host__ __device__ static inline int bwd_ldz_of(int maxdim, int h3) { return h3 ? round_up(maxdim, 32) + 12 : bwd_ldz(maxdim); }
// backward du buffer: float4 epilogue stores -> ld % 4 == 0
__host__ __device__ static inline int bwd_ldu(int maxdim) { return round_up(maxdim, 16) + 4; }

// ------------------------------------------------------------------------------------------------
// GEMM building blocks (one wave, A in LDS, B streamed from global/L2)
// ------------------------------------------------------------------------------------------------
// "NT" form (forward):  Y[r, o] = sum_k Xs[r, k] * W[o, k]      W row-major [M, K]
// One call = one chunk of 16*CT output columns starting at o0, for RT row tiles of 16.
// B fragments: lane (i = l&15, q = l>>4) loads W[o0 + 16t + i][k0 + 4q .. +3] (float4 along k), which is the
// B operand of four consecutive k-steps (any fixed permutation of k inside the contraction is legal).
template <int RT, int CT, bool VEC>
__device__ __forceinline__ void gemm_nt_chunk(const float* __restrict__ Xs, int ldx, int K, int K16,
                                              const Src& W, const float* __restrict__ bias,
                                              int M, int o0, int act, float* __restrict__ Ys, int ldy,
                                              float* __restrict__ gout, int rows_valid, int lane) {
  // PF-deep register ring of B fragments: the step is latency-bound (weights come from L2, ~700 cycles), so
  // every wave keeps PF*CT 16-byte loads in flight instead of one iteration's worth.
  constexpr int PF = (CT == 4) ? 4 : 8;
  const int i = lane & 15, q = lane >> 4;
  f32x4 acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int64_t woff[CT];
  bool wok[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int o = o0 + 16 * t + i;
    wok[t] = o < M;
    woff[t] = (int64_t)o * K;
  }
  const int nit = K16 >> 4;
  float4 bq[PF][CT];
#pragma unroll
  for (int u = 0; u < PF; ++u)
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      // VEC: no masks at all - rows o >= M fall past the described buffer (hardware returns 0), k >= K only
      // meets the zero padding of the A tile in LDS
      if constexpr (VEC) bq[u][t] = buf_ld4(W, (unsigned)(woff[t] + 16 * u + 4 * q) * 4u);
      else bq[u][t] = ld4_sel<VEC>(W, woff[t], wok[t] && u < nit, 16 * u + 4 * q, K);
    }
  // A fragments are software-pipelined one step ahead as well (the ds_read_b128 -> MFMA dependency would
  // otherwise expose the LDS latency in every step)
  float4 a[RT], an[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) a[rt] = ld4(Xs + (rt * 16 + i) * ldx + 4 * q);
  for (int it0 = 0; it0 < nit; it0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int it = it0 + u;
      if (it < nit) {
        const int k0 = it * 16;
        const int kn = (it + 1 < nit) ? (k0 + 16) : k0;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) an[rt] = ld4(Xs + (rt * 16 + i) * ldx + kn + 4 * q);
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc[rt][t] = mfma16(a[rt].x, bq[u][t].x, acc[rt][t]);
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc[rt][t] = mfma16(a[rt].y, bq[u][t].y, acc[rt][t]);
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc[rt][t] = mfma16(a[rt].z, bq[u][t].z, acc[rt][t]);
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc[rt][t] = mfma16(a[rt].w, bq[u][t].w, acc[rt][t]);
        if (it + PF < nit) {
#pragma unroll
          for (int t = 0; t < CT; ++t) {
            if constexpr (VEC) bq[u][t] = buf_ld4(W, (unsigned)(woff[t] + k0 + 16 * PF + 4 * q) * 4u);
            else bq[u][t] = ld4_sel<VEC>(W, woff[t], wok[t], k0 + 16 * PF + 4 * q, K);
          }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) a[rt] = an[rt];
      }
    }
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

// "NN" form (dgrad):  DU[r, c] = sum_{m in [mb, me)} DZs[r, m] * W[m, c]       W row-major [M, K]
// One call = one chunk of 64 output columns starting at c0 over a slice [mb, me) of the contraction.
// Lane (i, q) loads the float4 W[m0 + 4s + q][c0 + 4i .. +3] for s = 0..3: the B operands of four interleaved
// column tiles (tile t holds columns c0 + 4j + t) for four m-steps, i.e. one 16-byte load feeds 4 MFMAs per row
// tile - the same ratio as the forward form, without keeping a transposed copy of the weights.
template <int CT> struct BVec;
template <> struct BVec<4> { typedef f32x4 type; };
template <> struct BVec<2> { typedef f32x2 type; };
template <int CT>
__device__ __forceinline__ typename BVec<CT>::type buf_ldv(const Src& s, unsigned byte_off) {
  if constexpr (CT == 4) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(s.rs, byte_off, 0, 0);
    return (f32x4){__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
  } else {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(s.rs, byte_off, 0, 0);
    return (f32x2){__uint_as_float(v.x), __uint_as_float(v.y)};
  }
}

template <int RT, int CT, bool VEC>
__device__ __forceinline__ void gemm_nn(const float* __restrict__ As, int lda, const Src& W, int K,
                                        int mb, int me, int c0, f32x4 (&acc)[RT][CT], int lane) {
  typedef typename BVec<CT>::type bvec;
  const int i = lane & 15, q = lane >> 4;
  const int col = c0 + CT * i;
  if constexpr (VEC) {
    // Straight-line software pipeline, 32 rows of W (two 16-row groups) per trip, no control flow and no masks:
    //  * the slice [mb, me) starts on a multiple of 32; the A tile in LDS is ZERO beyond the real contraction
    //    length up to the next multiple of 32, so a ragged tail contributes nothing;
    //  * lane (i, q) owns contraction indices m0 + 4q .. 4q+3 of a group: A is ONE ds_read_b128, B four 4*CT-byte
    //    rows W[m0 + 4q + s][c0 + CT*i ..] (64*CT B contiguous per 16 lanes);
    //  * the next trip's B rows are issued at the TOP of the trip into their own registers (reloading in place
    //    would have to wait for the MFMAs that read them - hipcc then sinks every load to the end of the body and
    //    drains vmcnt(0) at the top); past the slice they are fetched with the out-of-bounds offset (no traffic).
    const int npair = (me - mb + 31) >> 5;
    const unsigned rs = (unsigned)K * 4u;  // bytes per row of W
    unsigned o0 = ((unsigned)(mb + 4 * q) * (unsigned)K + (unsigned)col) * 4u;
    const float* ap = As + i * lda + mb + 4 * q;
    int m0 = mb;
    // one trip: prefetch the NEXT 32 rows into (nx0, nx1), consume (cu0, cu1).  The caller alternates the two
    // register sets, so there are no register copies and no in-place reloads.
    auto trip = [&](bvec(&cu0)[4], bvec(&cu1)[4], bvec(&nx0)[4], bvec(&nx1)[4]) {
      const unsigned on = o0 + 32u * rs;
      const bool more0 = m0 + 32 < me, more1 = m0 + 48 < me;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        nx0[s] = buf_ldv<CT>(W, more0 ? (on + (unsigned)s * rs) : ULTR_OOB);
        nx1[s] = buf_ldv<CT>(W, more1 ? (on + (unsigned)(16 + s) * rs) : ULTR_OOB);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch at the top of the trip
      float4 a0[RT], a1[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        a0[rt] = ld4(ap + rt * 16 * lda);
        a1[rt] = ld4(ap + rt * 16 * lda + 16);
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const float av[4] = {a0[rt].x, a0[rt].y, a0[rt].z, a0[rt].w};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < CT; ++t) acc[rt][t] = mfma16(av[s], cu0[s][t], acc[rt][t]);
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const float av[4] = {a1[rt].x, a1[rt].y, a1[rt].z, a1[rt].w};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < CT; ++t) acc[rt][t] = mfma16(av[s], cu1[s][t], acc[rt][t]);
      }
      o0 = on;
      ap += 32;
      m0 += 32;
    };
    bvec p0[4], p1[4], r0[4], r1[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      p0[s] = buf_ldv<CT>(W, o0 + (unsigned)s * rs);
      p1[s] = buf_ldv<CT>(W, (mb + 16 < me) ? (o0 + (unsigned)(16 + s) * rs) : ULTR_OOB);
    }
    int pr = 0;
    for (; pr + 1 < npair; pr += 2) {
      trip(p0, p1, r0, r1);
      trip(r0, r1, p0, p1);
    }
    if (pr < npair) trip(p0, p1, r0, r1);
  } else {
    static_assert(VEC || CT == 4, "generic path is 4-wide");
    // generic path (unaligned / ragged shapes): masked scalar loads, no pipelining
    for (int m0 = mb; m0 < me; m0 += 4) {
      const int m = m0 + q;
      const float4 b = ld4_sel<false>(W, (int64_t)m * K, m < me, col, K);
      const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const float a = (m < me) ? As[(rt * 16 + i) * lda + m] : 0.f;
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[rt][t] = mfma16(a, bv[t], acc[rt][t]);
      }
    }
  }
}

// The same contraction as gemm_nn<.., true>, split into an ISSUE half and a CONSUME half so that a wave can put the
// first D-1 trips of its W panel in flight BEFORE the phase that produces the A tile (LayerNorm, the elementwise
// backward passes) and keep D-1 trips in flight while it computes: with 16-row tiles a trip is only 16*CT MFMAs
// (0.5-1k cycles per wave), less than one L2/HBM round trip, so a depth-1 pipeline exposes the latency every trip.
// One slot = one trip = 32 rows of W = 8 loads of 4*CT bytes per lane.  Slots are indexed by compile-time constants
// only (fully unrolled), there is no control flow around any load (out-of-range trips fetch the out-of-bounds
// offset: zeros, no traffic), so hipcc keeps counted s_waitcnt vmcnt(N) throughout.
template <int RT, int CT, int D, int SCHED = 1>
struct GemmPipe {
  typedef typename BVec<CT>::type bvec;
  bvec b[D][8];
  unsigned of, of0;  // this lane's byte offset of the next trip to fetch / of the slice's first trip
  unsigned rs;       // bytes per row of W
  int mf, mb, me;    // contraction index of the next trip to fetch / slice bounds
  int left;          // trips still to fetch

  template <int S>
  __device__ __forceinline__ void fetch(const Src& W) {
    const bool ok0 = left > 0, ok1 = left > 0 && mf + 16 < me;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if constexpr (SCHED == 2) {  // experiment: no global loads
        b[S][s] = (bvec)(1.0f);
        b[S][4 + s] = (bvec)(1.0f);
      } else {
        b[S][s] = buf_ldv<CT>(W, ok0 ? (of + (unsigned)s * rs) : ULTR_OOB);
        b[S][4 + s] = buf_ldv<CT>(W, ok1 ? (of + (unsigned)(16 + s) * rs) : ULTR_OOB);
      }
    }
    --left;
    mf += 32;
    of += 32u * rs;
    if (mf >= me) {  // wrap: trips are visited in rotated order (see begin)
      mf = mb;
      of = of0;
    }
  }
  // slice [mb_, me_) of the contraction (mb_ a multiple of 32), output columns c0 .. c0 + 16*CT; !valid => no
  // traffic.  rot rotates the ORDER in which the slice's 32-row trips are visited (trip (rot + t) mod n): workgroups
  // that stream the same W in lockstep would otherwise all hit the same few L2 channels at the same moment.
  __device__ __forceinline__ void begin(const Src& W, int ldw, int mb_, int me_, int c0, bool valid, int rot, int lane) {
    const int i = lane & 15, q = lane >> 4;
    const int n = (me_ - mb_ + 31) >> 5;
    rs = (unsigned)ldw * 4u;
    of0 = ((unsigned)(mb_ + 4 * q) * (unsigned)ldw + (unsigned)(c0 + CT * i)) * 4u;
    mb = mb_;
    me = me_;
    left = valid ? n : 0;
    const int r0 = n > 0 ? rot % n : 0;
    mf = mb_ + 32 * r0;
    of = of0 + (unsigned)(32 * r0) * rs;
    if constexpr (D > 1) fetch<0>(W);
    if constexpr (D > 2) fetch<1>(W);
    if constexpr (D > 3) fetch<2>(W);
    if constexpr (D > 4) fetch<3>(W);
    if constexpr (D > 5) fetch<4>(W);
    if constexpr (D > 6) fetch<5>(W);
    if constexpr (D > 7) fetch<6>(W);
    static_assert(D >= 2 && D <= 8, "pipeline depth");
  }
  // Instruction mix of one trip: 8 W loads (for a later trip), 2*RT LDS reads, 8*RT*CT MFMAs.
  //   SCHED 0: all loads first (a burst: every wave of the CU queues on the one texture-address unit while the
  //            matrix cores idle, then all waves compute while the memory pipe idles);
  //   SCHED 1: one load after every RT*CT MFMAs, so address generation runs in the shadow of the MFMAs.
  __device__ __forceinline__ void sched_top() {
    if constexpr (SCHED == 0) __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void sched_mix() {
    if constexpr (SCHED == 1) {
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * RT, 0);  // DS reads (the A fragments)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, RT * CT, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);        // VMEM read
      }
    }
  }
  template <int S>
  __device__ __forceinline__ void consume(const float* __restrict__ ap, int lda, f32x4 (&acc)[RT][CT]) {
    float4 a0[RT], a1[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      a0[rt] = ld4(ap + rt * 16 * lda);
      a1[rt] = ld4(ap + rt * 16 * lda + 16);
    }
    if constexpr (SCHED == 3) {  // experiment: no MFMA
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[0][t][s & 3] += b[S][s][t] * a0[0].x;
      return;
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const float av[4] = {a0[rt].x, a0[rt].y, a0[rt].z, a0[rt].w};
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[rt][t] = mfma16(av[s], b[S][s][t], acc[rt][t]);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const float av[4] = {a1[rt].x, a1[rt].y, a1[rt].z, a1[rt].w};
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[rt][t] = mfma16(av[s], b[S][4 + s][t], acc[rt][t]);
    }
  }
  // consume the slice begun with begin() (same mb_, me_, rot): As = A tile in LDS, zero beyond the real contraction
  // length up to a multiple of 32
  __device__ __forceinline__ void run(const float* __restrict__ As, int lda, const Src& W, int mb_, int me_, int rot,
                                      f32x4 (&acc)[RT][CT], int lane) {
    const int i = lane & 15, q = lane >> 4;
    const int n = (me_ - mb_ + 31) >> 5;
    const float* a_lo = As + i * lda + mb_ + 4 * q;
    const float* a_hi = a_lo + 32 * n;
    const float* ap = a_lo + 32 * (n > 0 ? rot % n : 0);
    auto adv = [&]() {
      ap += 32;
      if (ap == a_hi) ap = a_lo;
    };
    auto step = [&](auto uc) {
      constexpr int U = decltype(uc)::value;
      fetch<(U + D - 1) % D>(W);
      sched_top();
      consume<U>(ap, lda, acc);
      sched_mix();
      adv();
    };
    int t = 0;
    for (; t + D <= n; t += D) {
      step(std::integral_constant<int, 0>());
      step(std::integral_constant<int, 1>());
      if constexpr (D > 2) step(std::integral_constant<int, 2>());
      if constexpr (D > 3) step(std::integral_constant<int, 3>());
      if constexpr (D > 4) step(std::integral_constant<int, 4>());
      if constexpr (D > 5) step(std::integral_constant<int, 5>());
      if constexpr (D > 6) step(std::integral_constant<int, 6>());
      if constexpr (D > 7) step(std::integral_constant<int, 7>());
    }
    // tail (< D trips, already in flight): consume only
    if (t < n) { consume<0>(ap, lda, acc); adv(); }
    if constexpr (D > 2) if (t + 1 < n) { consume<1>(ap, lda, acc); adv(); }
    if constexpr (D > 3) if (t + 2 < n) { consume<2>(ap, lda, acc); adv(); }
    if constexpr (D > 4) if (t + 3 < n) { consume<3>(ap, lda, acc); adv(); }
    if constexpr (D > 5) if (t + 4 < n) { consume<4>(ap, lda, acc); adv(); }
    if constexpr (D > 6) if (t + 5 < n) { consume<5>(ap, lda, acc); adv(); }
    if constexpr (D > 7) if (t + 6 < n) { consume<6>(ap, lda, acc); adv(); }
  }
};

// The same contraction over a FRAGMENT-MAJOR copy of the weights (DnnPlan::wsf_off / wsb_off, ultr_sw_index): a wave owns
// a chunk of 32 output columns (two 16-column MFMA tiles); one trip = 32 steps of the contraction = FOUR buffer_load_dwordx4,
// each 1 KiB contiguous per wave and carrying two steps x two column tiles per lane.  Same products in the same order as
// GemmPipe<1, 2, D> over the k-major copy (bitwise identical results, tools/swz_ubench.hip); the vector L1 returns 16-byte
// lanes at twice the rate of 8-byte ones and half as many load instructions are issued.
template <int D>
struct PipeSw {
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
  // trips [t0, t0 + n) of chunk `chunk` (ntrips per chunk in the matrix); !valid => no traffic
  __device__ __forceinline__ void begin(const Src& W, int chunk, int ntrips, int t0, int n, bool valid, int lane) {
    of = (((unsigned)chunk * (unsigned)ntrips + (unsigned)t0) * 256u + (unsigned)lane) * 16u;
    left = valid ? n : 0;
    if constexpr (D > 1) fetch<0>(W);
    if constexpr (D > 2) fetch<1>(W);
    if constexpr (D > 3) fetch<2>(W);
    static_assert(D >= 2 && D <= 4, "pipeline depth");
  }
  template <int S>
  __device__ __forceinline__ void consume(const float* __restrict__ ap, f32x4 (&acc)[2]) {
    const float4 a0 = ld4(ap), a1 = ld4(ap + 16);
    const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc[0] = mfma16(av[2 * u], b[S][u].x, acc[0]);
      acc[1] = mfma16(av[2 * u], b[S][u].y, acc[1]);
      acc[0] = mfma16(av[2 * u + 1], b[S][u].z, acc[0]);
      acc[1] = mfma16(av[2 * u + 1], b[S][u].w, acc[1]);
    }
  }
  // As = A tile in LDS (zero beyond the real contraction length up to a multiple of 32); consumes the n trips begun above
  __device__ __forceinline__ void run(const float* __restrict__ As, int lda, const Src& W, int t0, int n, f32x4 (&acc)[2], int lane) {
    const int i = lane & 15, q = lane >> 4;
    const float* ap = As + i * lda + 32 * t0 + 4 * q;
    auto step = [&](auto uc) {
      constexpr int U = decltype(uc)::value;
      fetch<(U + D - 1) % D>(W);
      __builtin_amdgcn_sched_barrier(0);
      consume<U>(ap, acc);
      ap += 32;
    };
    int t = 0;
    for (; t + D <= n; t += D) {
      step(std::integral_constant<int, 0>());
      step(std::integral_constant<int, 1>());
      if constexpr (D > 2) step(std::integral_constant<int, 2>());
      if constexpr (D > 3) step(std::integral_constant<int, 3>());
    }
    if (t < n) { consume<0>(ap, acc); ap += 32; }
    if constexpr (D > 2) if (t + 1 < n) { consume<1>(ap, acc); ap += 32; }
    if constexpr (D > 3) if (t + 2 < n) { consume<2>(ap, acc); ap += 32; }
  }
};
#ifndef FB_SWD
#define FB_SWD 2  // trips in flight per wave of the fragment-major pipeline (tools/swz_ubench.hip: 2, 3, 4 within 4 %)
#endif
#ifndef FB_SW
#define FB_SW 1   // dnn_fb_kernel streams the fragment-major copies when the plan has them (0: the k-major / row-major paths)
#endif

// Products on the fp16 matrix cores with SPLIT operands (DnnPlan::whf_off / whb_off, ultr_h3_index): the A tile lives in LDS as two
// fp16 planes (hi, lo of the row-scaled activations), the weights arrive as hi / lo fragments, and a . w = ah.wh + (ah.wl + al.wh)
// with fp32 accumulation on v_mfma_f32_16x16x32_f16 - 22 bits of operand mantissa, 6 MFMAs of 16 cycles per 32-deep step and
// two column tiles where the fp32 path issues 16 MFMAs of 32 cycles.  One step = FOUR buffer_load_dwordx4 per lane (tile 0 hi,
// tile 0 lo, tile 1 hi, tile 1 lo), each 1 KiB contiguous per wave: the bytes of the fp32 copy.  The cross terms go to their
// own accumulators (they are 2^-11 of the main term) and are added at the end.
template <int D>
struct PipeH3 {
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
  __device__ __forceinline__ void begin(const Src& W, int chunk, int nks, bool valid, int lane) {
    of = ((unsigned)chunk * (unsigned)nks * 256u + (unsigned)lane) * 16u;
    left = valid ? nks : 0;
    if constexpr (D > 1) fetch<0>(W);
    if constexpr (D > 2) fetch<1>(W);
    static_assert(D >= 2 && D <= 3, "pipeline depth");
  }
  // Three accumulator sets per column tile (ah.wh | ah.wl | al.wh): three independent MFMA chains, every accumulator written once per
  // step.  (Round 3 presented this layout as the cure for an intermittent wrong result.  It was not: the cause was a packed fp32
  // multiply in the EPILOGUE - v_pk_mul_f32 .. op_sel:[0,1] reads its operand as zero in lanes 48..63 while the SIMD's other wave
  // is inside an MFMA loop - reproduced in isolation by tools/pkmul_coexec_test.hip; the library is built without packed fp32
  // instructions since, build.py.  Two chained sets are deterministic too: tools/h3_repro.sh variant C; profiles/r04_h3_rootcause.md.)
  template <int S>
  __device__ __forceinline__ void consume(const _Float16* __restrict__ ah_p, const _Float16* __restrict__ al_p, f32x4 (&acc)[2],
                                          f32x4 (&accx)[2], f32x4 (&accy)[2]) {
    const fbh8 ah = *reinterpret_cast<const fbh8*>(ah_p), al = *reinterpret_cast<const fbh8*>(al_p);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const fbh8 wh = fb_as_h8(b[S][2 * t]), wl = fb_as_h8(b[S][2 * t + 1]);
      acc[t] = fb_mfma_h(ah, wh, acc[t]);
      accx[t] = fb_mfma_h(ah, wl, accx[t]);
      accy[t] = fb_mfma_h(al, wh, accy[t]);
    }
  }
  // Ah / Al: the two planes of the A tile, row stride ldh halves, zero beyond the real contraction length up to nks * 32
  __device__ __forceinline__ void run(const _Float16* __restrict__ Ah, const _Float16* __restrict__ Al, int ldh, const Src& W, int nks,
                                      f32x4 (&acc)[2], f32x4 (&accx)[2], int lane) {
    const int i = lane & 15, q = lane >> 4;
    const _Float16* ph = Ah + i * ldh + 8 * q;
    const _Float16* pl = Al + i * ldh + 8 * q;
    f32x4 accy[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    auto step = [&](auto uc) {
      constexpr int U = decltype(uc)::value;
      fetch<(U + D - 1) % D>(W);
      __builtin_amdgcn_sched_barrier(0);
      consume<U>(ph, pl, acc, accx, accy);
      ph += 32;
      pl += 32;
    };
    int t = 0;
    for (; t + D <= nks; t += D) {
      step(std::integral_constant<int, 0>());
      step(std::integral_constant<int, 1>());
      if constexpr (D > 2) step(std::integral_constant<int, 2>());
    }
    if (t < nks) { consume<0>(ph, pl, acc, accx, accy); ph += 32; pl += 32; }
    if constexpr (D > 2) if (t + 1 < nks) { consume<1>(ph, pl, acc, accx, accy); ph += 32; pl += 32; }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) accx[tt] += accy[tt];
  }
};
// rows 4 q + r of the accumulators times the per-row output scales (1 / (row scale x weight scale)), cross terms folded in
__device__ __forceinline__ void fb_h3_finish(f32x4 (&acc)[1][2], const f32x4 (&accx)[2], const float* __restrict__ os, int lane) {
  const float4 o4 = ld4(os + 4 * (lane >> 4));
  const float o[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[0][t][r] = (acc[0][t][r] + accx[t][r]) * o[r];
}

// forward epilogue of the last contraction slice: (+ partial sums of earlier slices) + bias, activation; to LDS
// (next layer's input) and, when training, to HBM — 4*CT-byte stores, the lane owns CT consecutive output columns
template <int RT, int CT>
__device__ __forceinline__ void finish_fwd_nn(const f32x4 (&acc)[RT][CT], float* __restrict__ Ys, int ldy, int M, int c0,
                                              int lane, const float* __restrict__ bias, int act,
                                              float* __restrict__ gout, int rows_valid) {
  // VEC path only: M % 4 == 0, so a lane's CT columns are all inside or all outside
  const int i = lane & 15, q = lane >> 4;
  const int col = c0 + CT * i;
  if (col >= M) return;
  float bv[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) bv[t] = bias[col + t];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rt * 16 + 4 * q + r;
      float* dst = Ys + row * ldy + col;
      float v[CT];
#pragma unroll
      for (int t = 0; t < CT; ++t) v[t] = act_fwd(acc[rt][t][r] + bv[t], act);
      if constexpr (CT == 4) st4(dst, make_float4(v[0], v[1], v[2], v[3]));
      else *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
      if (gout != nullptr && row < rows_valid) {
        float* g = gout + (int64_t)row * M + col;
        if constexpr (CT == 4) st4_out(g, make_float4(v[0], v[1], v[2], v[3]));
        else st2_out(g, make_float2(v[0], v[1]));
      }
    }
}

// epilogue of gemm_nn: lane holds D_t[row = 4q + r][j = i] = DU[row][c0 + CT*i + t]
template <int RT, int CT>
__device__ __forceinline__ void store_nn(const f32x4 (&acc)[RT][CT], float* __restrict__ DUs, int ldu, int K, int c0,
                                         int lane, bool add) {
  const int i = lane & 15, q = lane >> 4;
  const int col = c0 + CT * i;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* dst = DUs + (rt * 16 + 4 * q + r) * ldu + col;
      float vv[CT];
#pragma unroll
      for (int t = 0; t < CT; ++t) vv[t] = acc[rt][t][r];
      if (col + CT - 1 < K) {
        if constexpr (CT == 4) {
          float4 v = make_float4(vv[0], vv[1], vv[2], vv[3]);
          if (add) {
            const float4 o = ld4(dst);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          }
          st4(dst, v);
        } else {
          float2 v = make_float2(vv[0], vv[1]);
          if (add) {
            const float2 o = *reinterpret_cast<const float2*>(dst);
            v.x += o.x; v.y += o.y;
          }
          *reinterpret_cast<float2*>(dst) = v;
        }
      } else {
#pragma unroll
        for (int t = 0; t < CT; ++t)
          if (col + t < K) dst[t] = add ? (dst[t] + vv[t]) : vv[t];
      }
    }
}

__device__ __forceinline__ int pick_ct(int width, int nw) {
  // widest column chunk (16*CT) that still gives every wave a chunk
  if (width >= 64 * nw) return 4;
  if (width >= 32 * nw) return 2;
  return 1;
}

// Optional phase tracing (build with -DULTR_TRACE): wave 0 of every 32nd workgroup stamps s_memtime at phase
// boundaries into g_ultr_trace; tools/trace_phases.py prints the deltas.  Compiled out by default.
#ifdef ULTR_TRACE
// three banks: 0 = the 8-wave kernels (their slot numbers overlap each other: trace one kernel at a time), 1 = dnn_fwdw_kernel,
// 2 = dnn_bwdw_kernel (a training step runs all of them).  One array PER TRANSLATION UNIT (the kernels of a bank live in one unit:
// ultr_dnn_fwd.hip banks 0 / 1, ultr_dnn_bwd.hip 0 / 2, ultr_dnn_fb.hip and ultr_dnn_wgrad.hip 0); ultr_trace_read (ultr_dnn.hip) adds the units' arrays
static __device__ unsigned long long g_ultr_trace[3 * 64 * 32];
#define TRACE_STAMP_B(bank, slot)                                                                   \
  do {                                                                                              \
    if (threadIdx.x == 0 && (blockIdx.x & 31) == 0 && (slot) < 32 && (blockIdx.x >> 5) < 64)        \
      g_ultr_trace[(bank) * 2048 + (blockIdx.x >> 5) * 32 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
// the same with the constant 100 MHz counter every XCD shares (s_memrealtime): when did the workgroup start / end inside the launch
#define TRACE_REAL_B(bank, slot)                                                                    \
  do {                                                                                              \
    if (threadIdx.x == 0 && (blockIdx.x & 31) == 0 && (slot) < 32 && (blockIdx.x >> 5) < 64)        \
      g_ultr_trace[(bank) * 2048 + (blockIdx.x >> 5) * 32 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#define ULTR_TRACE_READER(name)                                                                                              \
  int name(unsigned long long* host_out) {                                                                                  \
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ultr_trace), sizeof(unsigned long long) * 3 * 64 * 32);          \
  }
#else
#define TRACE_STAMP_B(bank, slot) \
  do {                            \
  } while (0)
#define TRACE_REAL_B(bank, slot) \
  do {                           \
  } while (0)
#define ULTR_TRACE_READER(name)
#endif
#define TRACE_STAMP(slot) TRACE_STAMP_B(0, slot)

// prefetch depth (trips of 32 W rows) of the forward GEMM pipeline
#ifndef FWD_SW
#define FWD_SW 1  // dnn_fwd_kernel: layers with >= 8 chunks of 32 output columns stream the fragment-major copy (PipeSw)
#endif
#ifndef BWD_SW
#define BWD_SW 1  // dnn_bwd2_kernel: the dgrad products of layers >= 1 with >= 8 chunks stream the fragment-major copy of W_j
#endif
#ifndef FWD_D
#define FWD_D 2
#endif

// ---- plan structs handed to the kernels -----------------------------------------------------------------------------------------
// dnn_fwdw_kernel (ultr_dnn_fwd.hip)
struct WidePlan {
  int R;                   // rows per workgroup
  int buf[2];              // float offsets of the two activation buffers in dynamic LDS (layer j reads buf[j & 1])
  int pv;                  // float offset of the vector-parameter image, followed by the 64 per-row output scales
  int ksplit[ULTR_MAXL];   // slices of layer j's contraction (waves = chunks x slices)
  int kslen[ULTR_MAXL];    // 32-deep steps per slice
};
// inputs of the fused NA / IPW loss (scores == nullptr: dscores come from a separate loss kernel)
struct FusedSoftmax {
  const float* scores;   // [B, L]
  const float* labels;   // [L, B]
  const float* pw;       // [B, L] or nullptr
  const float* ipw;      // [n_ipw] or nullptr
  int n_ipw;
  float* dscores_out;    // [B, L] or nullptr
  float* loss_part;      // [nrb][tail]
};

__device__ __forceinline__ int64_t sm_id_raw(const int32_t* __restrict__ docids, int64_t n, int B, int L, int64_t n_docs) {
  const int b = (int)(n / L), l = (int)(n % L);
  const int64_t d = docids[(int64_t)l * B + b];
  return (d >= 0 && d < n_docs) ? d : -1;
}

// floats of column partials per wave: dgamma | dbeta (| scorer dW for the top layer), each round_up(K_j, 4) long
__host__ __device__ static inline int bwd2_cp_stride(const DnnPlan& p) {
  int cpw = 0;
  for (int j = 0; j < p.nl; ++j) {
    const int v = (j == p.nl - 1 ? 3 : 2) * round_up(p.K[j], 4);
    cpw = v > cpw ? v : cpw;
  }
  return cpw;
}
__host__ __device__ static inline size_t bwd2_lds_floats(const DnnPlan& p, int R, int NW) {
  const size_t ldu = bwd_ldu(p.maxdim), ldz = bwd_ldz_of(p.maxdim, R == 16 ? p.bwd_h3 : 0);
  return (size_t)R * (2 * ldu + ldz) + 5 * ldu + (size_t)NW * bwd2_cp_stride(p) + 5 * (size_t)R + 2 * (size_t)NW + 8;
}

// dnn_bwdw_kernel (ultr_dnn_bwd.hip)
struct WideBwd {
  int R;
  int dz, du, ds;         // float offsets in dynamic LDS: dz planes [(R + 1)][M_j + 8] x 2 halves; du tile [(R + 1)][K_j + 8] (and the
                          // column partials [16][2 or 3][K_j]); ds[64] followed by the per-row plane scales [64]
  int ksplit[ULTR_MAXL];  // product j: slices of its contraction
  int kslen[ULTR_MAXL];   // 32-deep steps per slice
};

// dnn_fb_kernel (ultr_dnn_fb.hip)
__host__ __device__ static inline size_t fb_lds_floats(const DnnPlan& p) {
  const size_t ld = fwd_ld(p.maxdim), ldu = bwd_ldu(p.maxdim);
  return (size_t)16 * ld * (p.nl + 1) + 16 * ldu + (size_t)8 * bwd2_cp_stride(p) + (size_t)p.pv_total + 2 * 16 * (size_t)p.nl +
         2 * 16 + 2 * 8 + 8 +
         (p.h3_ok ? (size_t)16 * (round_up(p.maxdim, 32) + 8) + 8 : 0);  // two fp16 planes [16][ldh] (4 bytes per element)
}

// dnn_wgrad_h3_kernel (ultr_dnn_wgrad.hip): LDS geometry
#define WH_LDH 48  // halves per column of a plane: 32 contraction rows + 16 pad = 96 bytes; with the 16-byte slot index XORed with
                   // (column >> 2) & 3 the operand reads (ds_read_b128, lane (i, q) -> column i, slot q) are conflict-free and the
                   // staging writes (lane (c16, rg) -> column 4 c16 + c, slot rg) 2-way (13 -> 16 cycles): brute-forced over the
                   // lane groups of MI355X_MICROARCH.md's LDS table; the first layout (80 bytes, no XOR: reads 2-way, writes 4-way)
                   // spent 1 800 of 4 600 cycles per step between the two barriers around the plane writes
#define WH_ROWS_CAP 2048
#define WH_GROUP_HALVES (4 * 2 * 64 * WH_LDH)
#define WH_PLANES_BYTES (2 * WH_GROUP_HALVES * 2)
#define WH_TAB_ROWS (WH_ROWS_CAP + 160)
#define WH_MAIN_BYTES (WH_PLANES_BYTES + WH_TAB_ROWS * 12)  // planes of both groups | (mean, rstd) per row | doc id per row
#define WH_LDS_BYTES (WH_MAIN_BYTES + 64 + 4 * 64 * 4)

// ---- knobs (ultr_dnn.hip: read once, re-read by ultr_config_reload) ----------------------------------------------------------------
struct Knobs {
  int fwd_r, bwd_r, wgrad_wgs, fwd_nw, bwd_nw, no_vec, no_fused_fb, fb_max_wg_per_cu, fwd_q4, big_fwd, big_bwd, fb_h3, fwd_h3, bwd_h3, wg_h3, wg_h3_min_rows, wg_h3_wgs, fwd_wide, bwd_wide, fwd_wide_rmax;
  bool loaded;
};
const Knobs& ultr_knobs();
template <typename KernelT>
static hipError_t set_lds(KernelT k, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// ---- launchers (each next to its kernels; a kernel can only be launched from the unit that holds it: -fno-gpu-rdc) -----------------
struct UltrProfScope;
// ultr_dnn_fwd.hip
int ultr_launch_dnn_fwd(UltrProfScope& prof, const DnnPlan& p, int R, int nw, bool av, bool q4, size_t lds, hipStream_t st, const float* params,
                        const float* features, int64_t n_docs, const int32_t* docids, int batch, int list_size, float* scores, float* saved,
                        const float* wt, int vm);
int ultr_launch_dnn_fwdw(UltrProfScope& prof, const DnnPlan& p, const WidePlan& wp, size_t wlds, hipStream_t st, const float* features,
                         int64_t n_docs, const int32_t* docids, int batch, int list_size, float* scores, float* saved, const float* wt);
// ultr_dnn_bwd.hip
int ultr_launch_dnn_bwd(UltrProfScope& prof, const DnnPlan& p, const BwdPlan& bp, int nw, bool av, size_t lds, hipStream_t st, const float* params,
                        const float* features, int64_t n_docs, const int32_t* docids, int batch, int list_size, const float* saved,
                        const float* dscores, float* ws, int vm, const FusedSoftmax& fl);
int ultr_launch_dnn_bwd2(UltrProfScope& prof, const DnnPlan& p, const BwdPlan& bp, size_t lds2, hipStream_t st, const float* params,
                         const float* features, int64_t n_docs, const int32_t* docids, int batch, int list_size, const float* saved,
                         const float* dscores, float* ws, const FusedSoftmax& fl, const float* wt);
int ultr_launch_dnn_bwdw(UltrProfScope& prof, const DnnPlan& p, const BwdPlan& bp, const WideBwd& wb, size_t wblds, hipStream_t st,
                         const float* saved, const float* dscores, float* ws, const float* wt);
// ultr_dnn_fb.hip
int ultr_launch_dnn_fb(UltrProfScope& prof, const DnnPlan& p, const BwdPlan& bp, size_t lds, int64_t nblk, hipStream_t st, const float* params,
                       const float* wt, const float* features, int64_t n_docs, const int32_t* docids, int batch, int L, int lpb, float* scores,
                       float* saved, float* ws, const FusedSoftmax& fl, const FbPlan& fp);
// ultr_dnn_wgrad.hip
int ultr_launch_dnn_wgrad(UltrProfScope& prof, const DnnPlan& p, const BwdPlan& bp, bool av, bool h3, size_t wlds, dim3 wgrid, hipStream_t st,
                          const float* params, const float* features, int64_t n_docs, const int32_t* docids, int batch, int list_size,
                          const float* saved, float* ws, int l0_vec, float* grads, const float* lp, int nlp, int tail, const EarlyReport& er,
                          const CommDev& cd);
int ultr_launch_grad_reduce(UltrProfScope& prof, const RedPlan& rp, const DnnPlan& p, const BwdPlan& bp, int tail, int nblk, int maxparts,
                            hipStream_t st, float* ws, float* grads, int* nsq2_out);
int ultr_launch_grad_reduce_xchg(UltrProfScope& prof, const RedPlan& rp, const DnnPlan& p, const BwdPlan& bp, int tail, int nblk, const CommDev& cd,
                                 const EarlyReport& er, hipStream_t st, float* ws, float* grads, int* nblocks_out);
#ifdef ULTR_TRACE
int ultr_trace_read_fwd(unsigned long long* host_out);
int ultr_trace_read_bwd(unsigned long long* host_out);
int ultr_trace_read_fb(unsigned long long* host_out);
int ultr_trace_read_wgrad(unsigned long long* host_out);
#endif
