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
// Matrix math: v_mfma_f32_16x16x4_f32 (exact fp32) - and, for layers with >= 256 outputs / inputs where the plan carries split-half
// weight copies (DnnPlan::h3f / h3b; knobs ULTR_FB_H3 / ULTR_FWD_H3 / ULTR_BWD_H3, default on), three v_mfma_f32_16x16x32_f16 on
// hi / lo fp16 halves of both operands with fp32 accumulation (PipeH3: fp32-grade results).  Wave = 64 lanes.
// No packed fp32 VALU instructions anywhere (build.py NO_PACKED_FP32: a gfx950 hazard next to the f16 MFMAs).
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
// 2 = dnn_bwdw_kernel (a training step runs all of them)
__device__ unsigned long long g_ultr_trace[3 * 64 * 32];
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
extern "C" int ultr_trace_read(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ultr_trace), sizeof(unsigned long long) * 3 * 64 * 32);
}
#else
#define TRACE_STAMP_B(bank, slot) \
  do {                            \
  } while (0)
#define TRACE_REAL_B(bank, slot) \
  do {                           \
  } while (0)
#endif
#define TRACE_STAMP(slot) TRACE_STAMP_B(0, slot)

// ------------------------------------------------------------------------------------------------
// Forward
// ------------------------------------------------------------------------------------------------
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
// Q4: the variant with 64-column chunks / 16-byte weight loads in the GEMM phases.  It needs ~170 registers (four
// accumulator tiles + a two-trip ring of 16-byte loads), so the launcher picks it only where the LDS footprint leaves ONE
// workgroup per CU anyway (2 waves per SIMD: 256 registers each) - e.g. BASELINE config 4 (700-wide input: 228 -> 220 us);
// where two workgroups share a CU (config 3) the 128-register build below is the faster one (84 vs 103 us).
template <int R, int NW, bool VEC, bool Q4 = false>
__global__ __launch_bounds__(NW * 64) void dnn_fwd_kernel(DnnPlan p, const float* __restrict__ params,
                                                          const float* __restrict__ features, int64_t n_docs,
                                                          const int32_t* __restrict__ docids, int B, int L,
                                                          float* __restrict__ scores, float* __restrict__ saved,
                                                          const float* __restrict__ wt, int vecmask) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int RT = R / 16;
  const int64_t N = (int64_t)B * L;
  const int ld = fwd_ld_of(p.maxdim, p.fwd_h3);
  float* X = smem;
  float* Y = smem + R * ld;
  float* PV = smem + 2 * R * ld;  // every vector parameter of the model, staged once (see below)
  __shared__ __attribute__((aligned(16))) float sm_os[R];  // split-half layers: per-row output scale of the product
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave: SGPR
  const int64_t n0 = (int64_t)blockIdx.x * R;
  const int rows_valid = (int)((N - n0) < R ? (N - n0) : R);
  // marker word behind the saved activations: does saved.x_0 hold xhat_0 for the weight-gradient launch?  (the LayerNorm fast
  // path below writes it for inputs up to 256 wide; the launch then contracts layer 0 with it instead of gathering by id and
  // normalising again)
  bool write_xhat0 = saved != nullptr && p.nl >= 2 && p.K[0] <= 256;
  if constexpr (VEC && (R == 16 || R == 32) && NW == 8)
    write_xhat0 = write_xhat0 || (saved != nullptr && p.nl >= 2 && p.fwd_h3 != 0 && p.h3f[0] == 1 &&
                                  round_up(p.K[0], 32) <= 768);
  if (saved != nullptr && blockIdx.x == 0 && tid == 0) saved[p.sv_total] = write_xhat0 ? 1.f : 0.f;
  TRACE_STAMP(0);

  // LayerNorm gamma/beta, biases and the scorer's weight row go to LDS up front, overlapped with the feature
  // gather: each later phase would otherwise start with an exposed ~1-2k-cycle global load of a few hundred floats.
  // layout per layer j: gamma[K_j] | beta[K_j] | bias[M_j]; then the last layer's weight row [K_last]
  bool staged = false;
  if constexpr (VEC) {
    constexpr int NT = NW * 64, RPW = R / NW, PVR = 3, FCH = 4;
    if (wt != nullptr && p.pv_total <= PVR * NT * 4 && p.K[0] <= FCH * 256) {
      // Fast prologue, ONE exposed round trip + the dependent gather instead of three serial ones: the ids go
      // first, then the packed vector-parameter image (contiguous 16-byte loads, kept current by the update
      // kernel), then - as soon as the ids are back - every feature row of the wave; only then anything is
      // written to LDS.  No control flow around the loads (out-of-range chunks present the OOB offset).
      const int F = p.K[0];
      const int64_t nme = n0 + wave + NW * (lane < RPW ? lane : 0);
      const bool idok = lane < RPW && nme < N;
      const int bb = (int)((uint32_t)(idok ? nme : 0) / (uint32_t)L), ll = (int)((uint32_t)(idok ? nme : 0) % (uint32_t)L);
      const int myid_raw = docids[(int64_t)ll * B + bb];
      const Src pvs = make_src(wt + p.wt_pv_off, p.pv_total);
      float4 pvr[PVR];
#pragma unroll
      for (int u = 0; u < PVR; ++u) pvr[u] = buf_ld4(pvs, (unsigned)(tid + u * NT) * 16u);
      const int myid = (idok && myid_raw >= 0 && myid_raw < n_docs) ? myid_raw : -1;
      const Src fs = make_src(features, n_docs * F);
      float4 fr[RPW][FCH];
#pragma unroll
      for (int k = 0; k < RPW; ++k) {
        const int id = __builtin_amdgcn_readlane(myid, k);
#pragma unroll
        for (int u = 0; u < FCH; ++u) {
          const int c = lane * 4 + 256 * u;
          fr[k][u] = buf_ld4(fs, (id >= 0 && c < F) ? (unsigned)(((int64_t)id * F + c) * 4) : ULTR_OOB);
        }
      }
#pragma unroll
      for (int u = 0; u < PVR; ++u) {
        const int o = (tid + u * NT) * 4;
        if (o < p.pv_total) st4(PV + o, pvr[u]);
      }
      const int F16 = round_up(F, 16);
#pragma unroll
      for (int k = 0; k < RPW; ++k)
#pragma unroll
        for (int u = 0; u < FCH; ++u) {
          const int c = lane * 4 + 256 * u;
          if (c < F16) st4(X + (wave + NW * k) * ld + c, fr[k][u]);
        }
      staged = true;
    }
  }
  if (!staged) {
    int off = 0;
    for (int j = 0; j < p.nl; ++j) {
      const int K = p.K[j], M = p.M[j];
      for (int c = tid; c < K; c += NW * 64) {
        PV[off + c] = params[p.off_lnw[j] + c];
        PV[off + K + c] = params[p.off_lnb[j] + c];
      }
      for (int c = tid; c < M; c += NW * 64) PV[off + 2 * K + c] = params[p.off_b[j] + c];
      off += 2 * K + M;
    }
    const int Kl = p.K[p.nl - 1];
    for (int c = tid; c < Kl; c += NW * 64) PV[off + c] = params[p.off_w[p.nl - 1] + c];
    // ---- a2: gather feature rows (zero row for the PAD id == n_docs and for rows past N) ----------
    const int F = p.K[0];
    const int F16 = round_up(F, 16);
    const bool vecf = VEC || ((vecmask >> 31) & 1);
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
  lds_barrier();
  TRACE_STAMP(1);

  int pv_off = 0;
  for (int j = 0; j < p.nl; ++j) {
    const DnnPlan::FwdLayer lay = p.fl[j];  // one 64-byte scalar load for everything about this layer
    const int K = lay.K, M = lay.M;
    const int K16 = round_up(K, 32);  // zero-padded width of the A tile (multiple of 32, see gemm_nn)
    const float* lnw = PV + pv_off;
    const float* lnb = PV + pv_off + K;
    const float* bias = PV + pv_off + 2 * K;
    pv_off += 2 * K + M;
    // ---- plan of this layer's GEMM -----------------------------------------------------------------------
    // 32-column chunks.  Enough chunks for every wave: a wave takes chunks wave, wave + NW, .. over the whole
    // contraction.  Fewer: chunks x ksplit slices of the contraction, partial tiles summed in fixed order.
    int ksplit = 1, kb = 0, ke = K, c0 = wave * 32, kslice = 0;
    bool has = false;
    const int nch = (M + 31) >> 5;
    Src Wt = make_src(wt, 0);
    GemmPipe<RT, 2, FWD_D, 0> pipe;
    if constexpr (VEC) {
      if (j < p.nl - 1) {
        Wt = make_src(wt + lay.wt_off, (int64_t)K * M);
        int klen = K;
        if constexpr (NW == 8) {
          ksplit = lay.ksplit;
          klen = lay.klen;
        } else {
          while (ksplit * 2 * nch <= NW) ksplit *= 2;
          klen = round_up((K + ksplit - 1) / ksplit, 32);
        }
        if (ksplit > 1) {
          int wq = 0, wr = wave;  // wave / nch, wave % nch on scalars
          while (wr >= nch) { wr -= nch; ++wq; }
          c0 = wr * 32;
          kslice = wq;
          kb = wq * klen;
          ke = (kb + klen < K) ? (kb + klen) : K;
          has = wave < nch * ksplit && kb < ke;
        } else {
          has = c0 < M;
        }
      }
    }
    // ---- LayerNorm (biased variance, eps 1e-5, affine), in place; two-pass statistics -----------
    bool scored = false;
    bool h3 = false;
    if constexpr (VEC && RT == 1 && NW == 8) h3 = p.fwd_h3 != 0 && j < p.nl - 1 && p.h3f[j] == 1 && K16 <= 768;  // (32-row tiles behind one split-half stream were built and lost: profiles/r04_cfg2_attempts.md)
    if (h3) {
      if constexpr (VEC && RT == 1 && NW == 8) {
       auto ln_h3 = [&](auto xc_tag) {
        // split-half layer (PipeH3): a lane owns columns 4 lane + 256 u; the wave's two rows stay in registers through
        // both passes, and once every wave holds its rows (the barrier) the normalised rows go back over the tile as two
        // fp16 planes, scaled per row by a power of two
        constexpr int RPW = R / NW, XC = decltype(xc_tag)::value;  // rows up to 256 XC wide
        const float invK = 1.0f / (float)K;
        const int ldh = fwd_ldh(p.maxdim);
        _Float16* AH = reinterpret_cast<_Float16*>(X);
        _Float16* AL = AH + R * ldh;
        float4 xq[RPW][XC];
        float s[RPW], v[RPW], am[RPW];
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
          const float* row = X + (wave + NW * q) * ld;
          s[q] = 0.f;
#pragma unroll
          for (int u = 0; u < XC; ++u) {
            const int c = 4 * lane + 256 * u;
            xq[q][u] = (c < K) ? ld4(row + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            s[q] += (xq[q][u].x + xq[q][u].y) + (xq[q][u].z + xq[q][u].w);
          }
        }
        wave_sum_n<RPW>(s);
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
          s[q] *= invK;
          v[q] = 0.f;
#pragma unroll
          for (int u = 0; u < XC; ++u) {
            const int c = 4 * lane + 256 * u;
            float4& x = xq[q][u];
            if (c < K) {
              x.x -= s[q]; x.y -= s[q]; x.z -= s[q]; x.w -= s[q];
            }
            v[q] += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
          }
        }
        wave_sum_n<RPW>(v);
        lds_barrier();  // every wave has read its rows: the planes may overwrite them
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
          const int r = wave + NW * q;
          const float rstd = rsqrt_nr(v[q] * invK + ULTR_LN_EPS);
          am[q] = 0.f;
#pragma unroll
          for (int u = 0; u < XC; ++u) {
            const int c = 4 * lane + 256 * u;
            float4& x = xq[q][u];
            if (c < K) {
              const float4 g = ld4(lnw + c), be = ld4(lnb + c);
              const float4 xh = make_float4(x.x * rstd, x.y * rstd, x.z * rstd, x.w * rstd);
              if (j == 0 && write_xhat0 && n0 + r < N) st4(saved + p.sv_x[0] + (n0 + r) * K + c, xh);
              x = make_float4(xh.x * g.x + be.x, xh.y * g.y + be.y, xh.z * g.z + be.z, xh.w * g.w + be.w);
              am[q] = fmaxf(am[q], fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w))));
            }
          }
          if (lane == 0 && n0 + r < N && saved != nullptr) {
            saved[lay.sv_mean + n0 + r] = s[q];
            saved[lay.sv_rstd + n0 + r] = rstd;
          }
        }
        wave_max_n<RPW>(am);
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
          const int r = wave + NW * q;
          float rs, inv;
          fb_h3_scale(am[q], rs, inv);
#pragma unroll
          for (int u = 0; u < XC; ++u) {
            const int c = 4 * lane + 256 * u;
            if (c < K16) {
              fbh4 hi, lo;
              fb_h3_split4(xq[q][u], rs, hi, lo);
              *reinterpret_cast<fbh4*>(AH + r * ldh + c) = hi;
              *reinterpret_cast<fbh4*>(AL + r * ldh + c) = lo;
            }
          }
          if (lane == 0) sm_os[r] = inv * (1.0f / ULTR_H3_WSCALE);
        }
       };
       if (K16 <= 512) ln_h3(std::integral_constant<int, 2>());
       else ln_h3(std::integral_constant<int, 3>());
      }
    } else if (K <= 256) {
      // fast path: a lane owns columns lane + 64k (k < 4); gamma/beta are fetched once per layer, the wave's
      // rows live in registers between the passes and their reductions are interleaved.  The scorer (last
      // layer, M = 1) is folded in:  score = rstd * sum_c (x_c - mean) gamma_c w_c + sum_c beta_c w_c + b
      constexpr int RPW = (R + NW - 1) / NW;
      const bool last = (j == p.nl - 1);
      const float invK = 1.0f / (float)K;
      const float* wl = PV + pv_off;  // the scorer's weight row (valid when last)
      float g[4], be[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = lane + 64 * k;
        g[k] = (c < K) ? lnw[c] : 0.f;
        be[k] = (c < K) ? lnb[c] : 0.f;
        if (last) {
          const float w = (c < K) ? wl[c] : 0.f;
          g[k] *= w;
          be[k] *= w;
        }
      }
      if (j == 1) TRACE_STAMP(28);
      float x[RPW][4], s[RPW];
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        const int r = wave + NW * q;
        const float* row = X + (r < R ? r : 0) * ld;
        s[q] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int c = lane + 64 * k;
          x[q][k] = (c < K) ? row[c] : 0.f;
          s[q] += x[q][k];
        }
      }
      wave_sum_n<RPW>(s);
      if (j == 1) TRACE_STAMP(29);
      float v[RPW], t[RPW + 1];
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        s[q] *= invK;  // mean
        v[q] = 0.f;
        t[q] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int c = lane + 64 * k;
          x[q][k] = (c < K) ? (x[q][k] - s[q]) : 0.f;
          v[q] += x[q][k] * x[q][k];
          t[q] += x[q][k] * g[k];
        }
      }
      wave_sum_n<RPW>(v);
      if (j == 1) TRACE_STAMP(30);
      if (last) {
        t[RPW] = (be[0] + be[1]) + (be[2] + be[3]);
        wave_sum_n<RPW + 1>(t);
      }
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        const int r = wave + NW * q;
        if (r < R) {
          const float rstd = rsqrt_nr(v[q] * invK + ULTR_LN_EPS);
          if (!last) {
            float* row = X + r * ld;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int c = lane + 64 * k;
              if (c < K16) row[c] = x[q][k] * rstd * g[k] + be[k];  // c in [K, K16): 0 * rstd * 0 + 0 = 0 (zero padding)
            }
            if (j == 0 && write_xhat0 && n0 + r < N) {
              float* xh = saved + p.sv_x[0] + (n0 + r) * K;
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const int c = lane + 64 * k;
                if (c < K) xh[c] = x[q][k] * rstd;
              }
            }
          }
          if (lane == 0 && n0 + r < N) {
            if (saved != nullptr) {
              saved[lay.sv_mean + n0 + r] = s[q];
              saved[lay.sv_rstd + n0 + r] = rstd;
            }
            if (last) scores[n0 + r] = rstd * t[q] + t[RPW] + bias[0];
          }
        }
      }
      scored = last;
      if (j == 1) TRACE_STAMP(31);
    } else {
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
          saved[lay.sv_mean + n0 + r] = mean;
          saved[lay.sv_rstd + n0 + r] = rstd;
        }
      }
    }
    lds_barrier();
    TRACE_STAMP(2 + 3 * j);
    const float* W = params + lay.off_w;
    if (j < p.nl - 1) {
      // ---- Linear + activation on the matrix cores ------------------------------------------------
      float* gout = (saved != nullptr) ? (saved + lay.sv_x_next + n0 * M) : nullptr;
      if constexpr (VEC) {
        // Y = act(X . W^T + b) on the k-major weight copy.  (Issuing the first trips before the LayerNorm was
        // measured SLOWER: hipcc then drains vmcnt(0) inside the LayerNorm / epilogue code, see DESIGN.md.)
        // 64-column chunks with 16-byte weight loads (a lane holds 4 consecutive outputs of a weight row: 256 contiguous
        // bytes per 16 lanes, half the load instructions of the 32-column form) whenever the waves can be kept busy that way:
        // >= NW chunks (a wave walks chunks over the whole contraction) or chunks x equal 32-aligned contraction slices = NW
        int q4 = 0;  // 0: no; else contraction slices
        if constexpr (NW == 8 && Q4) {
          if ((M & 63) == 0) {
            const int nch4 = M >> 6;
            if (nch4 >= NW) q4 = 1;
            else if (NW % nch4 == 0 && K % (32 * (NW / nch4)) == 0 && K / (NW / nch4) >= 64) q4 = NW / nch4;
          }
        }
        bool sw_done = false;
        if constexpr (RT == 1 && NW == 8) {
          if (h3) {
            const int nks = K16 >> 5, ldh = fwd_ldh(p.maxdim);
            const _Float16* AH = reinterpret_cast<const _Float16*>(X);
            const _Float16* AL = AH + R * ldh;
            const Src Wh = make_src(wt + p.whf_off[j], (int64_t)K16 * M);
            PipeH3<FB_SWD> ph;
            const int cs = wave * 32;
            ph.begin(Wh, wave, nks, cs < M, lane);
            for (int cc = cs; cc < M; cc += NW * 32) {
              f32x4 acc[RT][2], accx[2];
#pragma unroll
              for (int t = 0; t < 2; ++t) acc[0][t] = accx[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
              ph.run(AH, AL, ldh, Wh, nks, acc[0], accx, lane);
              if (cc + NW * 32 < M) ph.begin(Wh, (cc + NW * 32) >> 5, nks, true, lane);
              fb_h3_finish(acc, accx, sm_os, lane);
              finish_fwd_nn<RT, 2>(acc, Y, ld, M, cc, lane, bias, p.act, gout, rows_valid);
            }
            sw_done = true;
          } else if (FWD_SW && p.sw_ok && M >= 32 * NW) {
            // fragment-major copy (DnnPlan::wsf_off): 32-column chunks, every wave over the whole contraction
            const int ntr = K16 >> 5;
            const Src Ws = make_src(wt + p.wsf_off[j], (int64_t)K16 * M);
            PipeSw<FB_SWD> ps;
            const int cs = wave * 32;
            ps.begin(Ws, wave, ntr, 0, ntr, cs < M, lane);
            for (int cc = cs; cc < M; cc += NW * 32) {
              f32x4 acc[RT][2];
#pragma unroll
              for (int t = 0; t < 2; ++t) acc[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
              ps.run(X, ld, Ws, 0, ntr, acc[0], lane);
              if (cc + NW * 32 < M) ps.begin(Ws, (cc + NW * 32) >> 5, ntr, 0, ntr, true, lane);
              finish_fwd_nn<RT, 2>(acc, Y, ld, M, cc, lane, bias, p.act, gout, rows_valid);
            }
            sw_done = true;
          }
        }
        if (sw_done) {
        } else if (Q4 && q4 == 1) {
          GemmPipe<RT, 4, FWD_D, 0> pipe4;
          const int c4 = wave * 64;
          pipe4.begin(Wt, M, 0, K, c4, c4 < M, 0, lane);
          for (int cc = c4; cc < M; cc += NW * 64) {
            f32x4 acc[RT][4];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
              for (int t = 0; t < 4; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            pipe4.run(X, ld, Wt, 0, K, 0, acc, lane);
            if (cc + NW * 64 < M) pipe4.begin(Wt, M, 0, K, cc + NW * 64, true, 0, lane);
            finish_fwd_nn<RT, 4>(acc, Y, ld, M, cc, lane, bias, p.act, gout, rows_valid);
          }
        } else if (Q4 && q4 > 1) {
          const int nch4 = NW / q4;
          int wq = 0, wr = wave;
          while (wr >= nch4) { wr -= nch4; ++wq; }
          const int c4 = wr * 64, kl = K / q4, kb4 = wq * kl;
          GemmPipe<RT, 4, FWD_D, 0> pipe4;
          pipe4.begin(Wt, M, kb4, kb4 + kl, c4, true, 0, lane);
          f32x4 acc[RT][4];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
          pipe4.run(X, ld, Wt, kb4, kb4 + kl, 0, acc, lane);
          for (int r = 0; r < q4; ++r) {
            if (wq == r) store_nn<RT, 4>(acc, Y, ld, M, c4, lane, r > 0);
            lds_barrier();
          }
          const int M4 = M >> 2;
          for (int e = tid; e < R * M4; e += NW * 64) {
            const int row = e / M4, c4e = (e - row * M4) * 4;
            float4 v = ld4(Y + row * ld + c4e);
            const float4 b4 = ld4(bias + c4e);
            v.x = act_fwd(v.x + b4.x, p.act);
            v.y = act_fwd(v.y + b4.y, p.act);
            v.z = act_fwd(v.z + b4.z, p.act);
            v.w = act_fwd(v.w + b4.w, p.act);
            st4(Y + row * ld + c4e, v);
            if (gout != nullptr && row < rows_valid) st4_out(gout + (int64_t)row * M + c4e, v);
          }
        } else if (ksplit == 1) {
          pipe.begin(Wt, M, kb, ke, c0, has, 0, lane);
          for (int cc = c0; cc < M; cc += NW * 32) {
            f32x4 acc[RT][2];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
              for (int t = 0; t < 2; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            pipe.run(X, ld, Wt, 0, K, 0, acc, lane);
            if (cc + NW * 32 < M) pipe.begin(Wt, M, 0, K, cc + NW * 32, true, 0, lane);
            finish_fwd_nn<RT, 2>(acc, Y, ld, M, cc, lane, bias, p.act, gout, rows_valid);
          }
        } else {
          pipe.begin(Wt, M, kb, ke, c0, has, 0, lane);
          f32x4 acc[RT][2];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (has) pipe.run(X, ld, Wt, kb, ke, 0, acc, lane);
          // raw partial tiles are summed into Y slice by slice (fixed order), then ALL threads apply bias +
          // activation (the expm1f-heavy epilogue would otherwise run on the last slice's waves only)
          const int ks = kslice;
          for (int r = 0; r < ksplit; ++r) {
            if (wave < nch * ksplit && ks == r) store_nn<RT, 2>(acc, Y, ld, M, c0, lane, r > 0);
            lds_barrier();
          }
          const int M4 = M >> 2;  // VEC path: M % 4 == 0
          for (int e = tid; e < R * M4; e += NW * 64) {
            const int row = e / M4, c4 = (e - row * M4) * 4;
            float4 v = ld4(Y + row * ld + c4);
            const float4 b4 = ld4(bias + c4);
            v.x = act_fwd(v.x + b4.x, p.act);
            v.y = act_fwd(v.y + b4.y, p.act);
            v.z = act_fwd(v.z + b4.z, p.act);
            v.w = act_fwd(v.w + b4.w, p.act);
            st4(Y + row * ld + c4, v);
            if (gout != nullptr && row < rows_valid) st4_out(gout + (int64_t)row * M + c4, v);
          }
        }
      } else {
        const Src Wsrc = make_src(W, (int64_t)M * K);
        const int ct = pick_ct(M, NW);
        if (ct == 4) {
          for (int ch = wave; ch * 64 < M; ch += NW)
            gemm_nt_chunk<RT, 4, false>(X, ld, K, K16, Wsrc, bias, M, ch * 64, p.act, Y, ld, gout, rows_valid, lane);
        } else if (ct == 2) {
          for (int ch = wave; ch * 32 < M; ch += NW)
            gemm_nt_chunk<RT, 2, false>(X, ld, K, K16, Wsrc, bias, M, ch * 32, p.act, Y, ld, gout, rows_valid, lane);
        } else {
          for (int ch = wave; ch * 16 < M; ch += NW)
            gemm_nt_chunk<RT, 1, false>(X, ld, K, K16, Wsrc, bias, M, ch * 16, p.act, Y, ld, gout, rows_valid, lane);
        }
      }
      TRACE_STAMP(3 + 3 * j);
      lds_barrier();
      TRACE_STAMP(4 + 3 * j);
      float* t = X;
      X = Y;
      Y = t;
    } else {
      // ---- final Linear(K, 1): a dot product per row, wave-shuffle reduction ---------------------
      if (!scored)
      for (int r = wave; r < R; r += NW) {
        const float* row = X + r * ld;
        float s = 0.f;
        const float* wl = PV + pv_off;  // the scorer's weight row
        for (int c = lane; c < K; c += 64) s += row[c] * wl[c];
        s = wave_sum(s);
        if (lane == 0 && n0 + r < N) scores[n0 + r] = s + bias[0];
      }
      TRACE_STAMP(3 + 3 * j);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Forward, wide row tiles (round 5)
// ------------------------------------------------------------------------------------------------
// dnn_fwd_kernel streams every weight once per 16 rows and is bound by exactly that stream (31 - 32 B/clk per CU through the
// L2 -> L1 path, the matrix cores a third busy) - and its 16-row tiles quantise badly: config 3 = 640 tiles on 512 slots, config 4 =
// 800 on 256.  This kernel gives a workgroup R = 17 .. 64 rows, chosen by the host so that the grid is a whole number of rounds
// (config 3: 40 rows x 256 workgroups), as RT = ceil(R / 16) MFMA row tiles behind ONE weight stream: every B fragment feeds RT
// row tiles (6 RT MFMAs of 16 cycles per 4 KiB of weights).  Sixteen waves; every hidden layer on the split-half copies
// (DnnPlan::h3f, value 2 = fewer than eight chunks: chunks x slices of the contraction, partial tiles summed in fixed order);
// the activations ping-pong between two LDS buffers sized per layer PARITY (not 2 x the widest layer), a LayerNorm turns the
// fp32 rows of its input buffer into the two fp16 planes in place; the gathered feature rows go from HBM through registers
// straight into LayerNorm_0 (no fp32 staging tile).
struct WidePlan {
  int R;                   // rows per workgroup
  int buf[2];              // float offsets of the two activation buffers in dynamic LDS (layer j reads buf[j & 1])
  int pv;                  // float offset of the vector-parameter image, followed by the 64 per-row output scales
  int ksplit[ULTR_MAXL];   // slices of layer j's contraction (waves = chunks x slices)
  int kslen[ULTR_MAXL];    // 32-deep steps per slice
};

template <int RT>
__global__ __launch_bounds__(1024) void dnn_fwdw_kernel(DnnPlan p, WidePlan wp, const float* __restrict__ features, int64_t n_docs,
                                                        const int32_t* __restrict__ docids, int B, int L,
                                                        float* __restrict__ scores, float* __restrict__ saved,
                                                        const float* __restrict__ wt) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NW = 16, NT = NW * 64, PVR = 2;
  const int R = wp.R;  // the buffers hold R + 1 rows: row R takes whatever the rows R .. 16 RT - 1 of the last MFMA tile produce
  const int64_t N = (int64_t)B * L;
  float* PV = smem + wp.pv;
  float* OS = PV + p.pv_total;  // per-row output scale of the current product (64 floats)
  const int tid = threadIdx.x, lane_id = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = lane_id;
  const int64_t n0 = (int64_t)blockIdx.x * R;
  const int vr = (int)((N - n0) < R ? (N - n0) : R);  // rows of this workgroup that exist
  const bool train = saved != nullptr;
  const int64_t tr = train ? 1 : 0;  // evaluation: every descriptor of `saved` has zero extent
  float* sbase = train ? saved : scores;
  if (train && blockIdx.x == 0 && tid == 0) saved[p.sv_total] = 1.f;  // saved.x_0 holds xhat_0 (see dnn_fwd_kernel)
  TRACE_STAMP_B(1, 0);
  TRACE_REAL_B(1, 30);

  // ---- prologue: ids -> packed vector-parameter image -> feature rows, all in flight before anything is written to LDS ------
  // lane q < RT of a wave holds the id of its row  wave + 16 q; the rows go to buffer 0 as fp32 (LayerNorm_0 reads them like
  // every later LayerNorm reads its input)
  {
    const int rme = wave + NW * (lane < RT ? lane : 0);
    const bool idok = lane < RT && rme < vr;
    const uint32_t nme = idok ? (uint32_t)(n0 + rme) : 0u;
    const int bb = (int)(nme / (uint32_t)L), ll = (int)(nme % (uint32_t)L);
    const int myid_raw = docids[(int64_t)ll * B + bb];
    const Src pvs = make_src(wt + p.wt_pv_off, p.pv_total);
    float4 pvr[PVR];
#pragma unroll
    for (int u = 0; u < PVR; ++u) pvr[u] = buf_ld4(pvs, (unsigned)(tid + u * NT) * 16u);
    const int myid = (idok && myid_raw >= 0 && myid_raw < n_docs) ? myid_raw : -1;
    const int F = p.K[0], F16 = round_up(F, 32), ld0 = F16 + 8;
    const Src fs = make_src(features, n_docs * F);
    float4 fr[RT][3];
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      const int id = __builtin_amdgcn_readlane(myid, q);
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int c = 4 * lane + 256 * u;
        fr[q][u] = buf_ld4(fs, (id >= 0 && c < F) ? (unsigned)(((int64_t)id * F + c) * 4) : ULTR_OOB);
      }
    }
#pragma unroll
    for (int u = 0; u < PVR; ++u) {
      const int o = (tid + u * NT) * 4;
      if (o < p.pv_total) st4(PV + o, pvr[u]);
    }
    float* X0 = smem + wp.buf[0];
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      const int r = wave + NW * q, rc = r < R ? r : R;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int c = 4 * lane + 256 * u;
        if (c < F16) st4(X0 + rc * ld0 + c, fr[q][u]);
      }
    }
  }
  lds_barrier();
  TRACE_STAMP_B(1, 1);

  int pv_off = 0;
  for (int j = 0; j < p.nl; ++j) {
    // (the lane id goes through an opaque move per layer: hipcc otherwise hoists the lane-derived indices and predicates of every
    // phase out of this loop and keeps - or spills - them across all of it)
    int lane_j = lane_id;
    asm volatile("" : "+v"(lane_j));
    const int lane = lane_j;
    const DnnPlan::FwdLayer lay = p.fl[j];
    const int K = lay.K, M = lay.M;
    const int K16 = round_up(K, 32);
    const int ldh = K16 + 8;  // halves per plane row = floats per fp32 row of the same buffer
    const float* lnw = PV + pv_off;
    const float* lnb = PV + pv_off + K;
    const float* bias = PV + pv_off + 2 * K;
    pv_off += 2 * K + M;
    float* Bin = smem + wp.buf[j & 1];
    float* Bout = smem + wp.buf[(j + 1) & 1];
    const bool last = j == p.nl - 1;
    const float invK = 1.0f / (float)K;
    const Dst d_mean = make_dst(sbase + tr * (lay.sv_mean + n0), tr * vr), d_rstd = make_dst(sbase + tr * (lay.sv_rstd + n0), tr * vr);

    // ---- LayerNorm_j: the wave's rows in registers (a lane owns columns 4 lane + 256 u); hidden layers: the normalised rows
    // go back over the buffer as two fp16 planes scaled per row by a power of two; last layer: the scorer is folded in
    auto ln = [&](auto xc_tag) {
      constexpr int XC = decltype(xc_tag)::value;
      float4 xq[RT][XC];
      float s[RT], v[RT];
#pragma unroll
      for (int q = 0; q < RT; ++q) {
        const int r = wave + NW * q;
        const float* row = Bin + (r < R ? r : R) * ldh;
#pragma unroll
        for (int u = 0; u < XC; ++u) {
          const int c = 4 * lane + 256 * u;
          xq[q][u] = (c < K) ? ld4(row + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int q = 0; q < RT; ++q) {
        s[q] = 0.f;
#pragma unroll
        for (int u = 0; u < XC; ++u) s[q] += (xq[q][u].x + xq[q][u].y) + (xq[q][u].z + xq[q][u].w);
      }
      wave_sum_n<RT>(s);
#pragma unroll
      for (int q = 0; q < RT; ++q) {
        s[q] *= invK;
        v[q] = 0.f;
#pragma unroll
        for (int u = 0; u < XC; ++u) {
          const int c = 4 * lane + 256 * u;
          float4& x = xq[q][u];
          if (c < K) {
            x.x -= s[q]; x.y -= s[q]; x.z -= s[q]; x.w -= s[q];
          }
          v[q] += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
        }
      }
      wave_sum_n<RT>(v);
      lds_barrier();  // every wave holds its rows: the planes may overwrite them
      const unsigned l0 = lane == 0 ? 0u : ULTR_OOB;
      if (!last) {
        _Float16* AH = reinterpret_cast<_Float16*>(Bin);
        _Float16* AL = AH + (R + 1) * ldh;
        const Dst d_x0 = make_dst(sbase + tr * (p.sv_x[0] + n0 * K), (j == 0 ? tr : 0) * (int64_t)vr * K);
        float am[RT];
#pragma unroll
        for (int q = 0; q < RT; ++q) {
          const int r = wave + NW * q;
          const float rstd = rsqrt_nr(v[q] * invK + ULTR_LN_EPS);
          am[q] = 0.f;
#pragma unroll
          for (int u = 0; u < XC; ++u) {
            const int c = 4 * lane + 256 * u;
            float4& x = xq[q][u];
            if (c < K) {
              const float4 g = ld4(lnw + c), be = ld4(lnb + c);
              const float4 xh = make_float4(x.x * rstd, x.y * rstd, x.z * rstd, x.w * rstd);
              buf_st4(d_x0, (unsigned)c * 4u, (unsigned)(r * K) * 4u, xh);  // layer 0, training: xhat_0 for the weight gradients
              x = make_float4(xh.x * g.x + be.x, xh.y * g.y + be.y, xh.z * g.z + be.z, xh.w * g.w + be.w);
              am[q] = fmaxf(am[q], fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w))));
            }
          }
          buf_st1(d_mean, l0, (unsigned)r * 4u, s[q]);
          buf_st1(d_rstd, l0, (unsigned)r * 4u, rstd);
        }
        wave_max_n<RT>(am);
#pragma unroll
        for (int q = 0; q < RT; ++q) {
          const int r = wave + NW * q, rc = r < R ? r : R;
          float rs, inv;
          fb_h3_scale(am[q], rs, inv);
#pragma unroll
          for (int u = 0; u < XC; ++u) {
            const int c = 4 * lane + 256 * u;
            if (c < K16) {
              fbh4 hi, lo;
              fb_h3_split4(xq[q][u], rs, hi, lo);
              *reinterpret_cast<fbh4*>(AH + rc * ldh + c) = hi;
              *reinterpret_cast<fbh4*>(AL + rc * ldh + c) = lo;
            }
          }
          if (lane == 0) OS[r] = inv * (1.0f / ULTR_H3_WSCALE);
        }
      } else {
        // score = rstd * sum_c (x_c - mean) gamma_c w_c + sum_c beta_c w_c + b
        const float* wl = PV + pv_off;  // the scorer's weight row
        const Dst d_sc = make_dst(scores + n0, vr);
        float t[RT + 1];
        t[RT] = 0.f;
#pragma unroll
        for (int q = 0; q < RT; ++q) t[q] = 0.f;
#pragma unroll
        for (int u = 0; u < XC; ++u) {
          const int c = 4 * lane + 256 * u;
          if (c < K) {
            const float4 g = ld4(lnw + c), be = ld4(lnb + c), w = ld4(wl + c);
            t[RT] += (be.x * w.x + be.y * w.y) + (be.z * w.z + be.w * w.w);
#pragma unroll
            for (int q = 0; q < RT; ++q) {
              const float4 x = xq[q][u];
              t[q] += (x.x * (g.x * w.x) + x.y * (g.y * w.y)) + (x.z * (g.z * w.z) + x.w * (g.w * w.w));
            }
          }
        }
        wave_sum_n<RT + 1>(t);
#pragma unroll
        for (int q = 0; q < RT; ++q) {
          const int r = wave + NW * q;
          const float rstd = rsqrt_nr(v[q] * invK + ULTR_LN_EPS);
          buf_st1(d_mean, l0, (unsigned)r * 4u, s[q]);
          buf_st1(d_rstd, l0, (unsigned)r * 4u, rstd);
          buf_st1(d_sc, l0, (unsigned)r * 4u, rstd * t[q] + t[RT] + bias[0]);
        }
      }
    };
    if (K16 <= 256) ln(std::integral_constant<int, 1>());
    else if (K16 <= 512) ln(std::integral_constant<int, 2>());
    else ln(std::integral_constant<int, 3>());
    TRACE_STAMP_B(1, 2 + 3 * j);
    if (last) {
      TRACE_REAL_B(1, 31);
      break;
    }
    lds_barrier();
    TRACE_STAMP_B(1, 3 + 3 * j);
    // this wave's share of the product: 32-column chunk(s) x a slice of the contraction.  (Requesting its first weight step in
    // front of the LayerNorm was measured: no change - config 3 forward 43.4 / 43.7 us against 43.2 / 46.0 on the same box.)
    const int nks = K16 >> 5, nch = M >> 5;
    const int ksplit = wp.ksplit[j];
    int ks = 0, ch0 = wave;
    if (ksplit > 1)
      while (ch0 >= nch) { ch0 -= nch; ++ks; }
    const bool has = ksplit > 1 ? ks < ksplit : wave < nch;
    const int k0 = ksplit > 1 ? ks * wp.kslen[j] : 0;
    const int cnt = !has ? 0 : ksplit == 1 ? nks : ((k0 + wp.kslen[j] < nks) ? wp.kslen[j] : (nks - k0));
    const Src Wh = make_src(wt + p.whf_off[j], (int64_t)K16 * M);
    PipeH3W<RT, FWDW_DEPTH> ph;
    ph.begin(Wh, ch0, nks, k0, cnt, has, lane);

    // ---- Linear_j + activation: Y = act((Ah + Al) . (Wh + Wl) x scales + b), 32-column chunks ----------------------------------
    {
      const int ldy = round_up(M, 32) + 8;
      const _Float16* AH = reinterpret_cast<const _Float16*>(Bin);
      const int lo_off = (R + 1) * ldh;
      const Dst d_y = make_dst(sbase + tr * (lay.sv_x_next + n0 * M), tr * (int64_t)vr * M);  // saved x_{j+1} rows of this workgroup
      const int i = lane & 15, q = lane >> 4;
      const _Float16* pa[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int row = 16 * rt + i;
        pa[rt] = AH + (row < R ? row : R) * ldh + 8 * q;
      }
      // this lane's rows 16 rt + 4 q + r of the output tile: LDS row (the rows beyond R collapse onto row R), byte offset in `saved`
      int yrow[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) yrow[rt] = 16 * rt + 4 * q;
      const unsigned gv = (unsigned)(4 * q * M + 2 * i) * 4u;
      if (ksplit == 1) {
        for (int ch = wave; ch < nch; ch += NW) {
          f32x4 acc[RT][2];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
          ph.run(pa, lo_off, Wh, nks, acc);
          if (ch + NW < nch) ph.begin(Wh, ch + NW, nks, 0, nks, true, lane);
          const int col = 32 * ch + 2 * i;
          const float2 bv = *reinterpret_cast<const float2*>(bias + col);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const float4 o4 = ld4(OS + 16 * rt + 4 * q);
            const float o[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = yrow[rt] + r;
              const int rc = (rt < RT - 1 || row < R) ? row : R;
              const float2 y = make_float2(act_fwd(acc[rt][0][r] * o[r] + bv.x, p.act), act_fwd(acc[rt][1][r] * o[r] + bv.y, p.act));
              *reinterpret_cast<float2*>(Bout + rc * ldy + col) = y;
              buf_st2(d_y, gv, (unsigned)((16 * rt + r) * M + 32 * ch) * 4u, y);
            }
          }
        }
      } else {
        // chunks x slices of the contraction: wave = slice * nch + chunk; the raw partial tiles are summed into the output buffer
        // slice by slice (fixed order), then every thread applies scale, bias and activation
        const int ch = ch0;
        f32x4 acc[RT][2];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) pa[rt] += 32 * k0;
        ph.run(pa, lo_off, Wh, cnt, acc);
        const int col = 32 * ch + 2 * i;
        for (int sl = 0; sl < ksplit; ++sl) {
          if (has && ks == sl) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int row = yrow[rt] + r;
                const int rc = (rt < RT - 1 || row < R) ? row : R;
                float2* dst = reinterpret_cast<float2*>(Bout + rc * ldy + col);
                float2 y = make_float2(acc[rt][0][r], acc[rt][1][r]);
                if (sl > 0) {
                  const float2 o = *dst;
                  y.x += o.x;
                  y.y += o.y;
                }
                *dst = y;
              }
          }
          lds_barrier();
        }
#pragma unroll
        for (int qq = 0; qq < RT; ++qq) {
          const int row = wave + NW * qq, rc = row < R ? row : R;
          const float os = OS[row];
          for (int c = 4 * lane; c < M; c += 256) {
            float4 y = ld4(Bout + rc * ldy + c);
            const float4 b4 = ld4(bias + c);
            y.x = act_fwd(y.x * os + b4.x, p.act);
            y.y = act_fwd(y.y * os + b4.y, p.act);
            y.z = act_fwd(y.z * os + b4.z, p.act);
            y.w = act_fwd(y.w * os + b4.w, p.act);
            st4(Bout + rc * ldy + c, y);
            buf_st4(d_y, (unsigned)c * 4u, (unsigned)(row * M) * 4u, y);
          }
        }
      }
    }
    TRACE_STAMP_B(1, 4 + 3 * j);
    lds_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// Backward, row-local half
// ------------------------------------------------------------------------------------------------
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

template <int R, int NW, bool VEC>
__global__ __launch_bounds__(NW * 64) void dnn_bwd_kernel(DnnPlan p, BwdPlan bp, const float* __restrict__ params,
                                                          const float* __restrict__ features, int64_t n_docs,
                                                          const int32_t* __restrict__ docids, int B, int L,
                                                          const float* __restrict__ saved,
                                                          const float* __restrict__ dscores, float* __restrict__ ws,
                                                          int vecmask, FusedSoftmax fl) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int RT = R / 16;
  constexpr int NT = NW * 64;
  const int64_t N = (int64_t)B * L;
  const int ldz = bwd_ldz(p.maxdim), ldu = bwd_ldu(p.maxdim);
  float* DU = smem;                    // [R][ldu]   (first: 16-byte aligned float4 stores)
  float* XS = DU + R * ldu;            // [R][ldu]   input of LayerNorm_j for this row block (staged once per layer)
  float* DZ = XS + R * ldu;            // [R][ldz]
  float* sm_g = DZ + R * ldz;          // [ldu] LayerNorm_j gamma
  float* sm_b = sm_g + ldu;            // [ldu] LayerNorm_j beta
  float* sm_ds = sm_b + ldu;           // [R]
  float* sm_mean2 = sm_ds + R;         // [2][R]  double-buffered by layer parity (no extra barrier)
  float* sm_rstd2 = sm_mean2 + 2 * R;  // [2][R]
  int64_t* sm_id = reinterpret_cast<int64_t*>(sm_rstd2 + 2 * R);  // [R] feature row id or -1
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave: SGPR
  const int64_t n0 = (int64_t)blockIdx.x * R;
  float* vslab = ws + bp.vslab_off + (int64_t)blockIdx.x * bp.vlen;

  if (tid < R) {
    const int64_t n = n0 + tid;
    float ds = 0.f;
    int64_t id = -1;
    if (n < N) {
      if (fl.scores == nullptr) ds = dscores[n];
      const int b = (int)(n / L), l = (int)(n % L);
      const int64_t d = docids[(int64_t)l * B + b];
      if (d >= 0 && d < n_docs) id = d;
    }
    sm_ds[tid] = ds;
    sm_id[tid] = id;
  }
  if (fl.scores != nullptr) {
    // ---- fused listwise softmax cross entropy (NA / IPW): this row block touches at most R/L + 2 lists; one
    // wavefront recomputes each of them (L scores from L2) instead of a separate launch + dependent kernel boundary.
    // A list's loss / normaliser partial is emitted by the block that owns the list's FIRST row, exactly once.
    lds_barrier();  // sm_ds zero-initialised above
    float* sm_lt = DU;  // [NW][2] scratch (DU is not live yet)
    if (lane < 2) sm_lt[wave * 2 + lane] = 0.f;
    const int64_t nlast = (n0 + R < N ? n0 + R : N) - 1;
    const int b_lo = (int)(n0 / L), b_hi = (int)(nlast / L);
    for (int b = b_lo + wave; b <= b_hi; b += NW) {
      float mx = -INFINITY, S = 0.f;
      for (int l = lane; l < L; l += 64) {
        const float sc = fl.scores[(int64_t)b * L + l];
        const float y = fl.labels[(int64_t)l * B + b];
        float pwt = 1.0f;
        if (fl.pw != nullptr) pwt = fl.pw[(int64_t)b * L + l];
        else if (fl.ipw != nullptr) pwt = (y > 0.f) ? fl.ipw[l < fl.n_ipw ? l : fl.n_ipw - 1] : 0.f;
        mx = fmaxf(mx, sc);
        S += (y + 0.0000001f) * pwt;
      }
      mx = wave_max(mx);
      S = wave_sum(S);
      float se = 0.f;
      for (int l = lane; l < L; l += 64) se += expf(fl.scores[(int64_t)b * L + l] - mx);
      const float lse = mx + logf(wave_sum(se));
      float lb = 0.f;
      for (int l = lane; l < L; l += 64) {
        const float sc = fl.scores[(int64_t)b * L + l];
        const float y = fl.labels[(int64_t)l * B + b];
        float pwt = 1.0f;
        if (fl.pw != nullptr) pwt = fl.pw[(int64_t)b * L + l];
        else if (fl.ipw != nullptr) pwt = (y > 0.f) ? fl.ipw[l < fl.n_ipw ? l : fl.n_ipw - 1] : 0.f;
        const float w = (y + 0.0000001f) * pwt;
        const float ds = expf(sc - lse) * S - w;
        lb += w * (lse - sc);
        const int64_t n = (int64_t)b * L + l;
        if (n >= n0 && n <= nlast) {
          sm_ds[n - n0] = ds;
          if (fl.dscores_out != nullptr) fl.dscores_out[n] = ds;
        }
      }
      lb = wave_sum(lb);
      if (lane == 0 && (int64_t)b * L >= n0) {
        sm_lt[wave * 2 + 0] += lb;
        sm_lt[wave * 2 + 1] += S;
      }
    }
    lds_barrier();
    const int tail = (int)ultr_tail_len(L);
    for (int t = tid; t < tail; t += NT) {
      float v = 0.f;
      if (t < 2)
        for (int w = 0; w < NW; ++w) v += sm_lt[w * 2 + t];
      fl.loss_part[(int64_t)blockIdx.x * tail + t] = v;
    }
  }

  const int jlow = bp.l0g ? 1 : 0;  // layer-0 shortcut: du_0 is never formed (BwdPlan::l0g)
  for (int j = p.nl - 1; j >= jlow; --j) {
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
    // stage x_j [R, K] (saved activations, or the gathered feature rows for j == 0) into LDS with every thread's
    // loads in flight at once; the column / row passes below then never touch global memory for x
    // (a per-row serial global read cost ~11k cycles per layer).  XS is free here: its last readers finished
    // before the barrier that ended the previous layer's row pass... which is the one below for j < nl-1.
    if (j < p.nl - 1) lds_barrier();
    for (int c = tid; c < K; c += NT) {
      sm_g[c] = lnw[c];
      sm_b[c] = lnb[c];
    }
    {
      const bool v4 = VEC || (((vecmask >> 31) & 1) && j == 0 && (K & 3) == 0) || (j > 0 && (K & 3) == 0);
      if (v4) {
        const int K4 = K >> 2;
        for (int e = tid; e < R * K4; e += NT) {
          const int r = e / K4, c4 = (e - r * K4) * 4;
          const int64_t n = n0 + r;
          const float* src = nullptr;
          if (n < N) {
            if (j == 0) {
              const int64_t id = sm_id_raw(docids, n, B, L, n_docs);
              if (id >= 0) src = features + id * K;
            } else {
              src = saved + p.sv_x[j] + n * K;
            }
          }
          st4(XS + r * ldu + c4, src ? ld4(src + c4) : make_float4(0.f, 0.f, 0.f, 0.f));
        }
      } else {
        for (int e = tid; e < R * K; e += NT) {
          const int r = e / K, c = e - r * K;
          const int64_t n = n0 + r;
          float x = 0.f;
          if (n < N) {
            if (j == 0) {
              const int64_t id = sm_id_raw(docids, n, B, L, n_docs);
              if (id >= 0) x = features[id * K + c];
            } else {
              x = saved[p.sv_x[j] + n * K + c];
            }
          }
          XS[r * ldu + c] = x;
        }
      }
    }
    lds_barrier();  // sm_*, XS visible; DZ of the previous iteration complete
    TRACE_STAMP(16 + 4 * (p.nl - 1 - j));
    // ---- du_j = dz_j . W_j ------------------------------------------------------------------------
    if (last) {
      for (int r = wave; r < R; r += NW) {
        const float ds = sm_ds[r];
        for (int c = lane; c < K; c += 64) DU[r * ldu + c] = ds * W[c];
      }
    } else {
      // 64-column chunks x slices of the contraction so that all NW waves work; slices are summed into DU
      // in fixed order (slice 0 stores, slice r adds after a barrier) -> deterministic
      const Src Wsrc = make_src(W, (int64_t)M * K);
      const int nch = (K + 63) >> 6;
      int msplit = 1;
      while (msplit * 2 * nch <= NW) msplit *= 2;
      bool done = false;
      if constexpr (VEC) {
        if (msplit > 1 && ((K + 31) >> 5) >= NW) {
          // 32-column chunks give every wave a whole contraction: no partial-tile rounds
          for (int ch = wave; ch * 32 < K; ch += NW) {
            f32x4 acc[RT][2];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
              for (int t = 0; t < 2; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            gemm_nn<RT, 2, true>(DZ, ldz, Wsrc, K, 0, M, ch * 32, acc, lane);
            store_nn<RT, 2>(acc, DU, ldu, K, ch * 32, lane, false);
          }
          done = true;
        }
      }
      if (done) {
      } else if (msplit == 1) {
        for (int ch = wave; ch < nch; ch += NW) {
          f32x4 acc[RT][4];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
          gemm_nn<RT, 4, VEC>(DZ, ldz, Wsrc, K, 0, M, ch * 64, acc, lane);
          store_nn<RT, 4>(acc, DU, ldu, K, ch * 64, lane, false);
        }
      } else {
        const int mlen = round_up((M + msplit - 1) / msplit, 32);
        const bool has = wave < nch * msplit;
        const int ch = wave % nch, ms = wave / nch;
        f32x4 acc[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (has) {
          const int mb = ms * mlen;
          const int me = (mb + mlen < M) ? (mb + mlen) : M;
          if (mb < me) gemm_nn<RT, 4, VEC>(DZ, ldz, Wsrc, K, mb, me, ch * 64, acc, lane);
        }
        for (int r = 0; r < msplit; ++r) {
          if (has && ms == r) store_nn<RT, 4>(acc, DU, ldu, K, ch * 64, lane, r > 0);
          if (r + 1 < msplit) lds_barrier();
        }
      }
    }
    TRACE_STAMP(17 + 4 * (p.nl - 1 - j));
    lds_barrier();
    TRACE_STAMP(18 + 4 * (p.nl - 1 - j));
    // ---- column pass: per-row-block partial sums of the vector-parameter gradients ---------------
    //   dgamma_j[c] = sum_r du[r,c] xhat[r,c]   dbeta_j[c] = sum_r du[r,c]
    //   final layer: dW[c] = sum_r ds[r] u[r,c], db = sum_r ds[r]
    for (int c = tid; c < K; c += NT) {
      float pg = 0.f, pb = 0.f, pw = 0.f;
      const float g = sm_g[c], be = sm_b[c];
#pragma unroll 4
      for (int r = 0; r < R; ++r) {
        if (n0 + r >= N) break;
        const float xh = (XS[r * ldu + c] - sm_mean[r]) * sm_rstd[r];
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
    TRACE_STAMP(19 + 4 * (p.nl - 1 - j));
    // ---- row pass: LayerNorm backward, then through the previous activation -> dz_{j-1} ----------
    if (j > 0) {
      float* dzg = ws + bp.dz_off[j - 1];
      for (int r = wave; r < R; r += NW) {
        const int64_t n = n0 + r;
        const bool valid = n < N;
        const float mean = sm_mean[r], rstd = sm_rstd[r];
        const float* xrow = XS + r * ldu;
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < K; c += 64) {
          const float x = xrow[c];
          const float xh = (x - mean) * rstd;
          const float gx = DU[r * ldu + c] * sm_g[c];
          s1 += gx;
          s2 += gx * xh;
        }
        s1 = wave_sum(s1) / (float)K;
        s2 = wave_sum(s2) / (float)K;
        for (int c = lane; c < K; c += 64) {
          const float x = xrow[c];
          const float xh = (x - mean) * rstd;
          const float gx = DU[r * ldu + c] * sm_g[c];
          const float dx = rstd * (gx - s1 - xh * s2);
          const float dzv = dx * act_grad_from_out(x, p.act);
          DZ[r * ldz + c] = dzv;
          if (valid) dzg[n * K + c] = dzv;
        }
        for (int c = K + lane; c < round_up(K, 32); c += 64) DZ[r * ldz + c] = 0.f;  // zero pad (gemm_nn reads it)
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Backward, row-local half - fast variant (aligned shapes, every K_j <= 256*XC, LDS budget permitting)
// ------------------------------------------------------------------------------------------------
// Same math and outputs as dnn_bwd_kernel; what changes is the schedule:
//  * a wave OWNS rows wave, wave+NW, .. of the block for everything row-local (staging, LayerNorm backward), so the
//    next layer's x tile / statistics / gamma, beta are PREFETCHED into registers right after the GEMM barrier and
//    committed to LDS after the row pass - their latency hides behind the pass instead of heading the next phase;
//  * the kernel's first loads (doc ids, fused-loss inputs, top layer's tile) are issued back to back before anything
//    waits (the old prologue paid four dependent round trips);
//  * the column sums (dgamma, dbeta, scorer dW) are accumulated inside the row pass - per-wave partials in LDS,
//    folded in fixed wave order after the next barrier - instead of a separate pass that re-read XS and DU and
//    recomputed xhat;
//  * 16-byte LDS accesses, the wave's rows interleaved (one set of wave reductions for all of them);
//  * the scorer layer needs no DU tile: du = ds * w is formed on the fly.
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

template <int R, int NW, int XC>
__global__ __launch_bounds__(NW * 64) void dnn_bwd2_kernel(DnnPlan p, BwdPlan bp, const float* __restrict__ params,
                                                           const float* __restrict__ features, int64_t n_docs,
                                                           const int32_t* __restrict__ docids, int B, int L,
                                                           const float* __restrict__ saved,
                                                           const float* __restrict__ dscores, float* __restrict__ ws,
                                                           FusedSoftmax fl, const float* __restrict__ wt) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int RT = R / 16, NT = NW * 64, RPW = R / NW;
  static_assert(R % NW == 0, "a wave owns whole rows");
  const int64_t N = (int64_t)B * L;
  const bool h3on = (RT == 1) && p.bwd_h3 != 0 && wt != nullptr;  // dgrad products on the split-half copies where a layer has one (DnnPlan::h3b)
  const int ldz = bwd_ldz_of(p.maxdim, h3on ? 1 : 0), ldu = bwd_ldu(p.maxdim);
  const int ldh = round_up(p.maxdim, 32) + 8;    // row stride (halves) of the two fp16 planes that then live in DZ
  __shared__ __attribute__((aligned(16))) float sm_os[16];  // their per-row output scales
  float* DU = smem;                    // [R][ldu]
  float* XS = DU + R * ldu;            // [R][ldu]  input of LayerNorm_j (rows written and read by their owner wave only)
  float* DZ = XS + R * ldu;            // [R][ldz]
  float* sm_g2 = DZ + R * ldz;         // [2][ldu]  gamma_j, double-buffered by layer parity
  float* sm_b2 = sm_g2 + 2 * ldu;      // [2][ldu]  beta_j
  float* sm_wl = sm_b2 + 2 * ldu;      // [ldu]     the scorer's weight row
  const int cpw = bwd2_cp_stride(p);
  float* CP = sm_wl + ldu;             // [NW][cpw] per-wave column partials (dgamma | dbeta | scorer dW)
  float* sm_ds = CP + NW * cpw;        // [R]
  float* sm_mean2 = sm_ds + R;         // [2][R]
  float* sm_rstd2 = sm_mean2 + 2 * R;  // [2][R]
  float* sm_lt = sm_rstd2 + 2 * R;     // [NW][2] loss / normaliser partials of the fused loss
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave: SGPR
  const int64_t n0 = (int64_t)blockIdx.x * R;
  float* vslab = ws + bp.vslab_off + (int64_t)blockIdx.x * bp.vlen;
  const Src savedsrc = make_src(saved, p.sv_total);
  const Src featsrc = make_src(features, n_docs * p.K[0]);
  const Src parsrc = make_src(params, p.P);
  const bool fused = fl.scores != nullptr;
  const int top = p.nl - 1;
  TRACE_STAMP(15);

  // ---- every first-round load of the kernel, back to back ------------------------------------------------------
  const int64_t nme = n0 + wave + NW * (lane < RPW ? lane : 0);  // lane k < RPW speaks for the wave's k-th row
  const bool rowok = lane < RPW && nme < N;
  const uint32_t nme32 = rowok ? (uint32_t)nme : 0u;
  const int id_raw = docids[(int64_t)(nme32 % (uint32_t)L) * B + (nme32 / (uint32_t)L)];
  float ds_in = 0.f;
  if (tid < R && !fused && n0 + tid < N) ds_in = dscores[n0 + tid];
  // fused loss, first list of this wave (lists b_lo + wave + NW*i); one element per lane when L <= 64
  const int64_t nlast = (n0 + R < N ? n0 + R : N) - 1;
  const int b_lo = (int)(n0 / L), b_hi = (int)(nlast / L);
  const bool l64 = L <= 64;
  const int b0 = b_lo + wave;
  const bool lact0 = fused && l64 && b0 <= b_hi && lane < L;
  float sc0 = 0.f, y0 = 0.f, pw0 = 1.0f;
  if (lact0) {
    sc0 = fl.scores[(int64_t)b0 * L + lane];
    y0 = fl.labels[(int64_t)lane * B + b0];
    if (fl.pw != nullptr) pw0 = fl.pw[(int64_t)b0 * L + lane];
    else if (fl.ipw != nullptr) pw0 = fl.ipw[lane < fl.n_ipw ? lane : fl.n_ipw - 1];
  }
  const int myid = (rowok && id_raw >= 0 && id_raw < n_docs) ? id_raw : -1;

  struct Stage {
    float4 x[RPW][XC];
    float4 g, b;
    float mean, rstd;
  };
  auto stage_issue = [&](int j, Stage& s) {
    const int K = p.K[j];
    const Src& xsrc = (j == 0) ? featsrc : savedsrc;
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const int64_t n = n0 + wave + NW * k;
      const int id = __builtin_amdgcn_readlane(myid, k);
      const bool ok = (j == 0) ? (id >= 0) : (n < N);
      const int64_t base = (j == 0) ? (int64_t)id * K : (p.sv_x[j] + n * K);
#pragma unroll
      for (int u = 0; u < XC; ++u) {
        const int c = 4 * lane + 256 * u;
        s.x[k][u] = buf_ld4(xsrc, (ok && c < K) ? (unsigned)((base + c) * 4) : ULTR_OOB);
      }
    }
    s.g = buf_ld4(parsrc, (4 * tid < K) ? (unsigned)((p.off_lnw[j] + 4 * tid) * 4) : ULTR_OOB);
    s.b = buf_ld4(parsrc, (4 * tid < K) ? (unsigned)((p.off_lnb[j] + 4 * tid) * 4) : ULTR_OOB);
    s.mean = buf_ld1(savedsrc, rowok ? (unsigned)((p.sv_mean[j] + nme) * 4) : ULTR_OOB);
    s.rstd = buf_ld1(savedsrc, rowok ? (unsigned)((p.sv_rstd[j] + nme) * 4) : ULTR_OOB);
  };
  auto stage_commit = [&](int j, const Stage& s) {
    const int K = p.K[j], par = j & 1;
#pragma unroll
    for (int k = 0; k < RPW; ++k)
#pragma unroll
      for (int u = 0; u < XC; ++u) {
        const int c = 4 * lane + 256 * u;
        if (c < K) st4(XS + (wave + NW * k) * ldu + c, s.x[k][u]);
      }
    if (4 * tid < K) {
      st4(sm_g2 + par * ldu + 4 * tid, s.g);
      st4(sm_b2 + par * ldu + 4 * tid, s.b);
    }
    if (lane < RPW) {
      sm_mean2[par * R + wave + NW * lane] = s.mean;
      sm_rstd2[par * R + wave + NW * lane] = s.rstd;
    }
  };

  Stage st;
  stage_issue(top, st);
  const float4 wl4 = buf_ld4(parsrc, (4 * tid < p.K[top]) ? (unsigned)((p.off_w[top] + 4 * tid) * 4) : ULTR_OOB);

  if (tid < R) sm_ds[tid] = ds_in;
  if (lane < 2) sm_lt[wave * 2 + lane] = 0.f;
  if (fused) {
    // ---- fused listwise softmax cross entropy (NA / IPW): this row block touches at most R/L + 2 lists; a
    // wavefront recomputes each of them.  A list's loss / normaliser partial is emitted by the block that owns the
    // list's FIRST row, exactly once.
    lds_barrier();  // sm_ds / sm_lt initialised
    if (l64) {
      for (int b = b0; b <= b_hi; b += NW) {
        const bool act = lane < L;
        float sc = sc0, y = y0, pwt = pw0;
        if (b != b0 && act) {
          sc = fl.scores[(int64_t)b * L + lane];
          y = fl.labels[(int64_t)lane * B + b];
          pwt = 1.0f;
          if (fl.pw != nullptr) pwt = fl.pw[(int64_t)b * L + lane];
          else if (fl.ipw != nullptr) pwt = fl.ipw[lane < fl.n_ipw ? lane : fl.n_ipw - 1];
        }
        if (fl.pw == nullptr && fl.ipw != nullptr && !(y > 0.f)) pwt = 0.f;
        const float w = act ? (y + 0.0000001f) * pwt : 0.f;
        const float mx = wave_max(act ? sc : -INFINITY);
        const float S = wave_sum(w);
        const float lse = mx + logf(wave_sum(act ? expf(sc - mx) : 0.f));
        const float dsv = expf(sc - lse) * S - w;
        const float lb = wave_sum(act ? w * (lse - sc) : 0.f);
        const int64_t n = (int64_t)b * L + lane;
        if (act && n >= n0 && n <= nlast) {
          sm_ds[n - n0] = dsv;
          if (fl.dscores_out != nullptr) fl.dscores_out[n] = dsv;
        }
        if (lane == 0 && (int64_t)b * L >= n0) {
          sm_lt[wave * 2 + 0] += lb;
          sm_lt[wave * 2 + 1] += S;
        }
      }
    } else {
      for (int b = b_lo + wave; b <= b_hi; b += NW) {
        float mx = -INFINITY, S = 0.f;
        for (int l = lane; l < L; l += 64) {
          const float sc = fl.scores[(int64_t)b * L + l];
          const float y = fl.labels[(int64_t)l * B + b];
          float pwt = 1.0f;
          if (fl.pw != nullptr) pwt = fl.pw[(int64_t)b * L + l];
          else if (fl.ipw != nullptr) pwt = (y > 0.f) ? fl.ipw[l < fl.n_ipw ? l : fl.n_ipw - 1] : 0.f;
          mx = fmaxf(mx, sc);
          S += (y + 0.0000001f) * pwt;
        }
        mx = wave_max(mx);
        S = wave_sum(S);
        float se = 0.f;
        for (int l = lane; l < L; l += 64) se += expf(fl.scores[(int64_t)b * L + l] - mx);
        const float lse = mx + logf(wave_sum(se));
        float lb = 0.f;
        for (int l = lane; l < L; l += 64) {
          const float sc = fl.scores[(int64_t)b * L + l];
          const float y = fl.labels[(int64_t)l * B + b];
          float pwt = 1.0f;
          if (fl.pw != nullptr) pwt = fl.pw[(int64_t)b * L + l];
          else if (fl.ipw != nullptr) pwt = (y > 0.f) ? fl.ipw[l < fl.n_ipw ? l : fl.n_ipw - 1] : 0.f;
          const float w = (y + 0.0000001f) * pwt;
          const float dsv = expf(sc - lse) * S - w;
          lb += w * (lse - sc);
          const int64_t n = (int64_t)b * L + l;
          if (n >= n0 && n <= nlast) {
            sm_ds[n - n0] = dsv;
            if (fl.dscores_out != nullptr) fl.dscores_out[n] = dsv;
          }
        }
        lb = wave_sum(lb);
        if (lane == 0 && (int64_t)b * L >= n0) {
          sm_lt[wave * 2 + 0] += lb;
          sm_lt[wave * 2 + 1] += S;
        }
      }
    }
  }
  TRACE_STAMP(14);
  stage_commit(top, st);
  if (4 * tid < p.K[top]) st4(sm_wl + 4 * tid, wl4);
  lds_barrier();
  if (fused) {
    const int tail = (int)ultr_tail_len(L);
    for (int t = tid; t < tail; t += NT) {
      float v = 0.f;
      if (t < 2)
        for (int w = 0; w < NW; ++w) v += sm_lt[w * 2 + t];
      fl.loss_part[(int64_t)blockIdx.x * tail + t] = v;
    }
  }
  TRACE_STAMP(16);

  // column sums of layer jj: fold the per-wave partials in wave order
  auto finalize = [&](int jj) {
    const int K = p.K[jj], K4 = round_up(K, 4);
    const bool lastl = (jj == top);
    for (int c = tid; c < K; c += NT) {
      float pg = 0.f, pb = 0.f, pw = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        pg += CP[w * cpw + c];
        pb += CP[w * cpw + K4 + c];
        if (lastl) pw += CP[w * cpw + 2 * K4 + c];
      }
      vslab[bp.voff_g[jj] + c] = pg;
      vslab[bp.voff_b[jj] + c] = pb;
      if (lastl) vslab[bp.voff_wk + c] = pw;
    }
    if (lastl && tid == 0) {
      float sds = 0.f;
      for (int r = 0; r < R; ++r) sds += sm_ds[r];
      vslab[bp.voff_bk] = sds;
    }
  };

  const int jlow = bp.l0g ? 1 : 0;  // layer-0 shortcut: du_0 is never formed (BwdPlan::l0g)
  for (int j = top; j >= jlow; --j) {
    const int K = p.K[j], M = p.M[j];
    const bool last = (j == top);
    const int par = j & 1;
    if (!last) {
      finalize(j + 1);
      // ---- du_j = dz_j . W_j  (32-column chunks when every wave gets one; else 64-column chunks x slices of the
      // contraction, summed into DU in fixed order)
      const Src Wsrc = make_src(params + p.off_w[j], (int64_t)M * K);
      static_assert(NW == 8, "the precomputed split is for 8 waves");
      const int nch = p.bwd_nch[j], msplit = p.bwd_msplit[j], mode = p.bwd_mode[j];
      bool sw_done = false;
      if constexpr (RT == 1) {
        if (h3on && wt != nullptr && j >= 1 && p.h3b[j] == 1) {
          // split-half copy of W_j (DnnPlan::whb_off) against the two planes of dz_j the row pass left in DZ
          const int nks = M >> 5;
          const _Float16* AH = reinterpret_cast<const _Float16*>(DZ);
          const _Float16* AL = AH + R * ldh;
          const Src Wh = make_src(wt + p.whb_off[j], (int64_t)M * K);
          PipeH3<FB_SWD> ph;
          ph.begin(Wh, wave, nks, wave * 32 < K, lane);
          for (int ch = wave; ch * 32 < K; ch += NW) {
            f32x4 acc[RT][2], accx[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[0][t] = accx[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            ph.run(AH, AL, ldh, Wh, nks, acc[0], accx, lane);
            if ((ch + NW) * 32 < K) ph.begin(Wh, ch + NW, nks, true, lane);
            // raw sums: the row pass below applies the per-row scale when it reads DU (its rows are the wave's own)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[0][t] += accx[t];
            store_nn<RT, 2>(acc, DU, ldu, K, ch * 32, lane, false);
          }
          sw_done = true;
        } else if (BWD_SW && wt != nullptr && p.sw_ok && j >= 1 && K >= 32 * NW) {
          // fragment-major copy of W_j (DnnPlan::wsb_off; M is a multiple of 32 there): 32-column chunks of K, whole contraction
          const int ntr = M >> 5;
          const Src Wb = make_src(wt + p.wsb_off[j], (int64_t)M * round_up(K, 32));
          PipeSw<FB_SWD> ps;
          ps.begin(Wb, wave, ntr, 0, ntr, wave * 32 < K, lane);
          for (int ch = wave; ch * 32 < K; ch += NW) {
            f32x4 acc[RT][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            ps.run(DZ, ldz, Wb, 0, ntr, acc[0], lane);
            if ((ch + NW) * 32 < K) ps.begin(Wb, ch + NW, ntr, 0, ntr, true, lane);
            store_nn<RT, 2>(acc, DU, ldu, K, ch * 32, lane, false);
          }
          sw_done = true;
        }
      }
      if (sw_done) {
      } else if (mode == 1) {
        for (int ch = wave; ch * 32 < K; ch += NW) {
          f32x4 acc[RT][2];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
          gemm_nn<RT, 2, true>(DZ, ldz, Wsrc, K, 0, M, ch * 32, acc, lane);
          store_nn<RT, 2>(acc, DU, ldu, K, ch * 32, lane, false);
        }
      } else if (mode == 2) {
        for (int ch = wave; ch < nch; ch += NW) {
          f32x4 acc[RT][4];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
          gemm_nn<RT, 4, true>(DZ, ldz, Wsrc, K, 0, M, ch * 64, acc, lane);
          store_nn<RT, 4>(acc, DU, ldu, K, ch * 64, lane, false);
        }
      } else {
        const int mlen = p.bwd_mlen[j];
        const bool has = wave < nch * msplit;
        int ms = 0, ch = wave;  // wave / nch, wave % nch on scalars
        while (ch >= nch) { ch -= nch; ++ms; }
        f32x4 acc[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (has) {
          const int mb = ms * mlen;
          const int me = (mb + mlen < M) ? (mb + mlen) : M;
          if (mb < me) gemm_nn<RT, 4, true>(DZ, ldz, Wsrc, K, mb, me, ch * 64, acc, lane);
        }
        for (int r = 0; r < msplit; ++r) {
          if (has && ms == r) store_nn<RT, 4>(acc, DU, ldu, K, ch * 64, lane, r > 0);
          if (r + 1 < msplit) lds_barrier();
        }
      }
      TRACE_STAMP(17 + 4 * (top - j));
      lds_barrier();
    }
    TRACE_STAMP(18 + 4 * (top - j));
    if (j > jlow) stage_issue(j - 1, st);
    // ---- row pass: LayerNorm backward + activation' -> dz_{j-1}; column partials on the side ------------------
    {
      const float* gs = sm_g2 + par * ldu;
      const float* bs = sm_b2 + par * ldu;
      const float invK = 1.0f / (float)K;
      float mean[RPW], rstd[RPW], dsr[RPW], dus[RPW];
      // du_j came out of the split-half product unscaled: its rows still carry the row scale of the dz planes
      const bool du_scaled = h3on && !last && j >= 1 && p.h3b[j] == 1;
#pragma unroll
      for (int k = 0; k < RPW; ++k) {
        const int r = wave + NW * k;
        mean[k] = sm_mean2[par * R + r];
        rstd[k] = sm_rstd2[par * R + r];
        dsr[k] = sm_ds[r];
        dus[k] = du_scaled ? sm_os[r & 15] : 1.0f;
      }
      float4 xk[RPW][XC], gxk[RPW][XC];
      float red[2 * RPW];
#pragma unroll
      for (int k = 0; k < 2 * RPW; ++k) red[k] = 0.f;
#pragma unroll
      for (int u = 0; u < XC; ++u) {
        const int c = 4 * lane + 256 * u;
        const bool act = c < K;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 g4 = act ? ld4(gs + c) : z4;
        const float4 be4 = (act && last) ? ld4(bs + c) : z4;
        const float4 w4 = (act && last) ? ld4(sm_wl + c) : z4;
        float4 pg = z4, pb = z4, pw = z4;
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
          const int r = wave + NW * k;
          const float4 x4 = act ? ld4(XS + r * ldu + c) : z4;
          float4 du4;
          if (last) du4 = make_float4(dsr[k] * w4.x, dsr[k] * w4.y, dsr[k] * w4.z, dsr[k] * w4.w);
          else {
            du4 = act ? ld4(DU + r * ldu + c) : z4;
            du4.x *= dus[k]; du4.y *= dus[k]; du4.z *= dus[k]; du4.w *= dus[k];
          }
          const float4 xh = make_float4((x4.x - mean[k]) * rstd[k], (x4.y - mean[k]) * rstd[k],
                                        (x4.z - mean[k]) * rstd[k], (x4.w - mean[k]) * rstd[k]);
          const float4 gx = make_float4(du4.x * g4.x, du4.y * g4.y, du4.z * g4.z, du4.w * g4.w);
          red[k] += (gx.x + gx.y) + (gx.z + gx.w);
          red[RPW + k] += (gx.x * xh.x + gx.y * xh.y) + (gx.z * xh.z + gx.w * xh.w);
          if (act) {  // padded lanes would add (0 - mean) * rstd garbage
            pg.x += du4.x * xh.x; pg.y += du4.y * xh.y; pg.z += du4.z * xh.z; pg.w += du4.w * xh.w;
            pb.x += du4.x; pb.y += du4.y; pb.z += du4.z; pb.w += du4.w;
            if (last) {
              pw.x += dsr[k] * (g4.x * xh.x + be4.x); pw.y += dsr[k] * (g4.y * xh.y + be4.y);
              pw.z += dsr[k] * (g4.z * xh.z + be4.z); pw.w += dsr[k] * (g4.w * xh.w + be4.w);
            }
          }
          xk[k][u] = x4;
          gxk[k][u] = gx;
        }
        if (act) {
          const int K4 = round_up(K, 4);
          st4(CP + wave * cpw + c, pg);
          st4(CP + wave * cpw + K4 + c, pb);
          if (last) st4(CP + wave * cpw + 2 * K4 + c, pw);
        }
      }
      if (j > 0) {
        wave_sum_n<2 * RPW>(red);
        float* dzg = ws + bp.dz_off[j - 1];
        // dz_{j-1} feeds the dgrad product of layer j-1: as two fp16 planes when that layer has a split-half copy
        const bool hz = h3on && j >= 2 && p.h3b[j - 1] == 1;
        float amz[RPW];
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
          const int r = wave + NW * k;
          const int64_t n = n0 + r;
          const float s1 = red[k] * invK, s2 = red[RPW + k] * invK;
          amz[k] = 0.f;
#pragma unroll
          for (int u = 0; u < XC; ++u) {
            const int c = 4 * lane + 256 * u;
            float4 dz = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < K) {
              const float4 x4 = xk[k][u], gx = gxk[k][u];
              dz.x = rstd[k] * (gx.x - s1 - (x4.x - mean[k]) * rstd[k] * s2) * act_grad_from_out(x4.x, p.act);
              dz.y = rstd[k] * (gx.y - s1 - (x4.y - mean[k]) * rstd[k] * s2) * act_grad_from_out(x4.y, p.act);
              dz.z = rstd[k] * (gx.z - s1 - (x4.z - mean[k]) * rstd[k] * s2) * act_grad_from_out(x4.z, p.act);
              dz.w = rstd[k] * (gx.w - s1 - (x4.w - mean[k]) * rstd[k] * s2) * act_grad_from_out(x4.w, p.act);
              if (!hz) st4(DZ + r * ldz + c, dz);
              if (n < N) st4_out(dzg + n * K + c, dz);
            }
            if constexpr (RT == 1) {
              gxk[k][u] = dz;  // (gx is dead from here on)
              amz[k] = fmaxf(amz[k], fmaxf(fmaxf(fabsf(dz.x), fabsf(dz.y)), fmaxf(fabsf(dz.z), fabsf(dz.w))));
            }
          }
          if (!hz)
            for (int c = K + lane; c < round_up(K, 32); c += 64) DZ[r * ldz + c] = 0.f;  // zero pad (gemm_nn reads it)
        }
        if constexpr (RT == 1) {
          if (hz) {
            _Float16* AH = reinterpret_cast<_Float16*>(DZ);
            _Float16* AL = AH + R * ldh;
            wave_max_n<RPW>(amz);
#pragma unroll
            for (int k = 0; k < RPW; ++k) {
              const int r = wave + NW * k;
              float rs, inv;
              fb_h3_scale(amz[k], rs, inv);
#pragma unroll
              for (int u = 0; u < XC; ++u) {
                const int c = 4 * lane + 256 * u;
                if (c < K) {  // K is a multiple of 32 here (DnnPlan::h3b)
                  fbh4 hi, lo;
                  fb_h3_split4(gxk[k][u], rs, hi, lo);
                  *reinterpret_cast<fbh4*>(AH + r * ldh + c) = hi;
                  *reinterpret_cast<fbh4*>(AL + r * ldh + c) = lo;
                }
              }
              if (lane == 0) sm_os[r] = inv * (1.0f / ULTR_H3_WSCALE);
            }
          }
        }
      }
    }
    TRACE_STAMP(19 + 4 * (top - j));
    if (j > jlow) stage_commit(j - 1, st);
    lds_barrier();
  }
  finalize(jlow);
}

// ------------------------------------------------------------------------------------------------
// Backward, row-local half - wide row tiles (round 5)
// ------------------------------------------------------------------------------------------------
// The counterpart of dnn_fwdw_kernel for the training step (ultr_train_step hands over the weight copies): R = 17 .. 48 rows per
// workgroup so that the grid is whole rounds of one workgroup per CU (config 3: 40 rows x 256 workgroups where dnn_bwd2_kernel ran
// 640 16-row tiles as three rounds), sixteen waves, every dgrad product du_j = dz_j . W_j on the split-half copies (DnnPlan::whb_off)
// with RT = ceil(R / 16) row tiles behind one weight stream.  Same outputs as dnn_bwd2_kernel: dz_j in HBM for the weight-gradient
// launch, one vector slab (d gamma_j, d beta_j, the scorer's dW / db) per workgroup.  Needs the layer-0 shortcut (BwdPlan::l0g: du_0
// is never formed), dscores from a loss kernel, and LayerNorms of layers >= 1 at most 512 wide.
//   row pass j (top .. 1): a wave owns rows wave + 16 q; x_j, the statistics and gamma_j come straight from `saved` / the
//     parameter image into registers (no LDS tile), du_j from the product's LDS tile (or ds x w for the scorer); dz_{j-1} goes to
//     HBM and - as two fp16 planes scaled per row - to LDS for the next product; the per-wave column partials of d gamma / d beta
//     overlay the du tile once every wave has read its rows, and are folded in wave order.
//   product j (top-1 .. 1): 32-column chunks of K_j x slices of the contraction M_j when there are fewer than sixteen chunks.
struct WideBwd {
  int R;
  int dz, du, ds;         // float offsets in dynamic LDS: dz planes [(R + 1)][M_j + 8] x 2 halves; du tile [(R + 1)][K_j + 8] (and the
                          // column partials [16][2 or 3][K_j]); ds[64] followed by the per-row plane scales [64]
  int ksplit[ULTR_MAXL];  // product j: slices of its contraction
  int kslen[ULTR_MAXL];   // 32-deep steps per slice
};

template <int RT>
__global__ __launch_bounds__(1024) void dnn_bwdw_kernel(DnnPlan p, BwdPlan bp, WideBwd wb, const float* __restrict__ saved,
                                                        const float* __restrict__ dscores, float* __restrict__ ws,
                                                        const float* __restrict__ wt) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NW = 16, NT = NW * 64;
  const int R = wb.R;
  const int64_t N = bp.N;
  float* DZ = smem + wb.dz;
  float* DU = smem + wb.du;
  float* CP = DU;
  float* DS = smem + wb.ds;
  float* OS = DS + 64;
  const int tid = threadIdx.x, lane_id = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = lane_id;
  const int64_t n0 = (int64_t)blockIdx.x * R;
  const int vr = (int)((N - n0) < R ? (N - n0) : R);
  float* vslab = ws + bp.vslab_off + (int64_t)blockIdx.x * bp.vlen;
  const Src svs = make_src(saved, p.sv_total);
  const Src pvs = make_src(wt + p.wt_pv_off, p.pv_total);
  const int top = p.nl - 1;
  TRACE_STAMP_B(2, 0);
  TRACE_REAL_B(2, 30);
  // lane q < RT of a wave speaks for its row  wave + 16 q
  const int rme = wave + NW * (lane < RT ? lane : 0);
  const bool rowok_l = lane < RT && rme < vr;
  const Src dss = make_src(dscores + n0, vr);
  const float ds_l = buf_ld1(dss, rowok_l ? (unsigned)rme * 4u : ULTR_OOB);
  if (lane < RT) DS[rme] = ds_l;

  for (int j = top; j >= 1; --j) {
    // (the lane id goes through an opaque move per layer: hipcc otherwise hoists every lane-derived index and predicate of all
    // phases out of this loop and spills them - 22 registers at three row tiles, each reload a memory round trip)
    int ln = lane_id;
    asm volatile("" : "+v"(ln));
    const int lane = ln;
    const int K = p.K[j];
    const bool last = j == top;
    const int ldu = K + 8;
    const int cpw = (last ? 3 : 2) * K;
    const float invK = 1.0f / (float)K;
    const float mean_l = buf_ld1(svs, rowok_l ? (unsigned)((p.sv_mean[j] + n0 + rme) * 4) : ULTR_OOB);
    const float rstd_l = buf_ld1(svs, rowok_l ? (unsigned)((p.sv_rstd[j] + n0 + rme) * 4) : ULTR_OOB);
    const Dst d_dz = make_dst(ws + bp.dz_off[j - 1] + n0 * K, (int64_t)vr * K);
    const bool planes = j >= 2;  // a product follows: dz_{j-1} also as the two fp16 planes of its A operand
    const int ldh = K + 8;       // (K = M_{j-1}: a multiple of 32)

    auto rowpass = [&](auto xc_tag, auto last_tag) {
      constexpr int XC = decltype(xc_tag)::value;
      constexpr bool LAST = decltype(last_tag)::value;  // the scorer's layer: du = ds x w, its dW on the side
      float4 xk[RT][XC], g4[XC];  // (du is read from its LDS tile twice rather than kept: 24 registers at three row tiles x 512 columns)
      // ---- loads: x_j rows, gamma_j (beta, scorer row for the top layer).  (Requested one product ahead and kept in registers
      // they cost more than the exposed round trip: 80 spilled registers at three row tiles, 63 us instead of 45 at config 3.)
#pragma unroll
      for (int q = 0; q < RT; ++q) {
        const int r = wave + NW * q;
#pragma unroll
        for (int u = 0; u < XC; ++u) {
          const int c = 4 * lane + 256 * u;
          xk[q][u] = buf_ld4(svs, (r < vr && c < K) ? (unsigned)((p.sv_x[j] + (n0 + r) * K + c) * 4) : ULTR_OOB);
        }
      }
#pragma unroll
      for (int u = 0; u < XC; ++u) {
        const int c = 4 * lane + 256 * u;
        g4[u] = buf_ld4(pvs, c < K ? (unsigned)(p.pv_off[j] + c) * 4u : ULTR_OOB);
      }
      float mean[RT], rstd[RT], dsr[RT], dus[RT];
#pragma unroll
      for (int q = 0; q < RT; ++q) {
        const int r = wave + NW * q;
        mean[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mean_l), q));
        rstd[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rstd_l), q));
        dsr[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ds_l), q));
        dus[q] = (!LAST && r < vr) ? OS[r] : 0.f;  // du_j came out of the product unscaled
      }
      float red[2 * RT];
#pragma unroll
      for (int k = 0; k < 2 * RT; ++k) red[k] = 0.f;
      if (j == 1) TRACE_STAMP_B(2, 12);
      float4 pg[XC], pb[XC], pw[LAST ? XC : 1], wk[LAST ? XC : 1], bek[LAST ? XC : 1];
      if constexpr (LAST) {  // beta and the scorer's row: requested with the x rows, not behind them
#pragma unroll
        for (int u = 0; u < XC; ++u) {
          const int c = 4 * lane + 256 * u;
          bek[u] = buf_ld4(pvs, c < K ? (unsigned)(p.pv_off[j] + K + c) * 4u : ULTR_OOB);
          wk[u] = buf_ld4(pvs, c < K ? (unsigned)(p.pv_wlast + c) * 4u : ULTR_OOB);
        }
      }
#pragma unroll
      for (int u = 0; u < XC; ++u) {
        const int c = 4 * lane + 256 * u;
        const bool act = c < K;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 be4 = z4, w4 = z4;
        if constexpr (LAST) {
          be4 = bek[u];
          w4 = wk[u];
          pw[u] = z4;
        }
        pg[u] = pb[u] = z4;
#pragma unroll
        for (int q = 0; q < RT; ++q) {
          const int r = wave + NW * q;
          const float4 x4 = xk[q][u];
          float4 du4;
          if constexpr (LAST) du4 = make_float4(dsr[q] * w4.x, dsr[q] * w4.y, dsr[q] * w4.z, dsr[q] * w4.w);
          else {
            du4 = (act && r < vr) ? ld4(DU + r * ldu + c) : z4;  // (rows that do not exist contribute nothing)
            du4.x *= dus[q]; du4.y *= dus[q]; du4.z *= dus[q]; du4.w *= dus[q];
          }
          const float4 xh = make_float4((x4.x - mean[q]) * rstd[q], (x4.y - mean[q]) * rstd[q], (x4.z - mean[q]) * rstd[q],
                                        (x4.w - mean[q]) * rstd[q]);
          const float4 gx = make_float4(du4.x * g4[u].x, du4.y * g4[u].y, du4.z * g4[u].z, du4.w * g4[u].w);
          red[q] += (gx.x + gx.y) + (gx.z + gx.w);
          red[RT + q] += (gx.x * xh.x + gx.y * xh.y) + (gx.z * xh.z + gx.w * xh.w);
          // (padded lanes: x = gamma = du = 0 -> xh = -mean rstd, but every product with it carries a zero factor)
          pg[u].x += du4.x * xh.x; pg[u].y += du4.y * xh.y; pg[u].z += du4.z * xh.z; pg[u].w += du4.w * xh.w;
          pb[u].x += du4.x; pb[u].y += du4.y; pb[u].z += du4.z; pb[u].w += du4.w;
          if constexpr (LAST) {
            pw[u].x += dsr[q] * (g4[u].x * xh.x + be4.x); pw[u].y += dsr[q] * (g4[u].y * xh.y + be4.y);
            pw[u].z += dsr[q] * (g4[u].z * xh.z + be4.z); pw[u].w += dsr[q] * (g4[u].w * xh.w + be4.w);
          }
        }
      }
      if (j == 1) TRACE_STAMP_B(2, 13);
      wave_sum_n<2 * RT>(red);
      if (j == 1) TRACE_STAMP_B(2, 14);
      float amz[RT];
#pragma unroll
      for (int q = 0; q < RT; ++q) {
        const int r = wave + NW * q;
        const float s1 = red[q] * invK, s2 = red[RT + q] * invK;
        amz[q] = 0.f;
#pragma unroll
        for (int u = 0; u < XC; ++u) {
          const int c = 4 * lane + 256 * u;
          float4 dz = make_float4(0.f, 0.f, 0.f, 0.f);
          if (c < K) {
            const float4 x4 = xk[q][u];
            float4 du4;
            if constexpr (LAST) du4 = make_float4(dsr[q] * wk[u].x, dsr[q] * wk[u].y, dsr[q] * wk[u].z, dsr[q] * wk[u].w);
            else {
              du4 = r < vr ? ld4(DU + r * ldu + c) : make_float4(0.f, 0.f, 0.f, 0.f);
              du4.x *= dus[q]; du4.y *= dus[q]; du4.z *= dus[q]; du4.w *= dus[q];
            }
            const float4 gx = make_float4(du4.x * g4[u].x, du4.y * g4[u].y, du4.z * g4[u].z, du4.w * g4[u].w);
            dz.x = rstd[q] * (gx.x - s1 - (x4.x - mean[q]) * rstd[q] * s2) * act_grad_from_out(x4.x, p.act);
            dz.y = rstd[q] * (gx.y - s1 - (x4.y - mean[q]) * rstd[q] * s2) * act_grad_from_out(x4.y, p.act);
            dz.z = rstd[q] * (gx.z - s1 - (x4.z - mean[q]) * rstd[q] * s2) * act_grad_from_out(x4.z, p.act);
            dz.w = rstd[q] * (gx.w - s1 - (x4.w - mean[q]) * rstd[q] * s2) * act_grad_from_out(x4.w, p.act);
            buf_st4(d_dz, (unsigned)c * 4u, (unsigned)(r * K) * 4u, dz);
          }
          xk[q][u] = dz;  // (x is dead from here on)
          amz[q] = fmaxf(amz[q], fmaxf(fmaxf(fabsf(dz.x), fabsf(dz.y)), fmaxf(fabsf(dz.z), fabsf(dz.w))));
        }
      }
      if (planes) {
        _Float16* AH = reinterpret_cast<_Float16*>(DZ);
        _Float16* AL = AH + (R + 1) * ldh;
        wave_max_n<RT>(amz);
#pragma unroll
        for (int q = 0; q < RT; ++q) {
          const int r = wave + NW * q, rc = r < R ? r : R;
          float rs, inv;
          fb_h3_scale(amz[q], rs, inv);
#pragma unroll
          for (int u = 0; u < XC; ++u) {
            const int c = 4 * lane + 256 * u;
            if (c < K) {
              fbh4 hi, lo;
              fb_h3_split4(xk[q][u], rs, hi, lo);
              *reinterpret_cast<fbh4*>(AH + rc * ldh + c) = hi;
              *reinterpret_cast<fbh4*>(AL + rc * ldh + c) = lo;
            }
          }
          if (lane == 0) OS[r] = inv * (1.0f / ULTR_H3_WSCALE);
        }
      }
      if (j == 1) TRACE_STAMP_B(2, 15);
      lds_barrier();  // every wave has read its rows of the du tile: the column partials may overlay it
      if (j == 1) TRACE_STAMP_B(2, 16);
#pragma unroll
      for (int u = 0; u < XC; ++u) {
        const int c = 4 * lane + 256 * u;
        if (c < K) {
          st4(CP + wave * cpw + c, pg[u]);
          st4(CP + wave * cpw + K + c, pb[u]);
          if constexpr (LAST) st4(CP + wave * cpw + 2 * K + c, pw[u]);
        }
      }
    };
    if (last) {
      if (K <= 256) rowpass(std::integral_constant<int, 1>(), std::true_type());
      else rowpass(std::integral_constant<int, 2>(), std::true_type());
    } else {
      if (K <= 256) rowpass(std::integral_constant<int, 1>(), std::false_type());
      else rowpass(std::integral_constant<int, 2>(), std::false_type());
    }
    TRACE_STAMP_B(2, 1 + 3 * (top - j));
    lds_barrier();
    // ---- column sums of layer j: the per-wave partials in wave order
    for (int e = tid; e < cpw; e += NT) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += CP[w * cpw + e];
      const int which = e >= 2 * K ? 2 : (e >= K ? 1 : 0), c = e - which * K;
      vslab[(which == 0 ? bp.voff_g[j] : which == 1 ? bp.voff_b[j] : bp.voff_wk) + c] = s;
    }
    if (last && tid == 0) {
      float sds = 0.f;
      for (int r = 0; r < R; ++r) sds += DS[r];
      vslab[bp.voff_bk] = sds;
    }
    TRACE_STAMP_B(2, 2 + 3 * (top - j));
    if (j == 1) break;
    lds_barrier();  // the partials are folded: the product may write the du tile
    // ---- du_{j-1} = dz_{j-1} . W_{j-1}: the planes against the split-half copy of W_{j-1} (contraction over its M = K_j outputs)
    {
      const int jj = j - 1;
      const int Ko = p.K[jj], nks = K >> 5, nch = Ko >> 5, ldo = Ko + 8;
      const _Float16* AH = reinterpret_cast<const _Float16*>(DZ);
      const int lo_off = (R + 1) * ldh;
      const Src Wh = make_src(wt + p.whb_off[jj], (int64_t)K * Ko);
      const int i = lane & 15, q = lane >> 4;
      const _Float16* pa[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int row = 16 * rt + i;
        pa[rt] = AH + (row < R ? row : R) * ldh + 8 * q;
      }
      const int ksplit = wb.ksplit[jj];
      PipeH3W<RT, FWDW_DEPTH> ph;
      if (ksplit == 1) {
        ph.begin(Wh, wave, nks, 0, nks, wave < nch, lane);
        for (int ch = wave; ch < nch; ch += NW) {
          f32x4 acc[RT][2];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
          ph.run(pa, lo_off, Wh, nks, acc);
          if (ch + NW < nch) ph.begin(Wh, ch + NW, nks, 0, nks, true, lane);
          const int col = 32 * ch + 2 * i;
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = 16 * rt + 4 * q + r;
              const int rc = (rt < RT - 1 || row < R) ? row : R;
              *reinterpret_cast<float2*>(DU + rc * ldo + col) = make_float2(acc[rt][0][r], acc[rt][1][r]);
            }
        }
      } else {
        int ks = 0, ch = wave;
        while (ch >= nch) { ch -= nch; ++ks; }
        const bool has = ks < ksplit;
        const int k0 = ks * wb.kslen[jj];
        const int cnt = has ? ((k0 + wb.kslen[jj] < nks) ? wb.kslen[jj] : (nks - k0)) : 0;
        f32x4 acc[RT][2];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) pa[rt] += 32 * k0;
        ph.begin(Wh, ch, nks, k0, cnt, has, lane);
        ph.run(pa, lo_off, Wh, cnt, acc);
        const int col = 32 * ch + 2 * i;
        for (int sl = 0; sl < ksplit; ++sl) {
          if (has && ks == sl) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int row = 16 * rt + 4 * q + r;
                const int rc = (rt < RT - 1 || row < R) ? row : R;
                float2* dst = reinterpret_cast<float2*>(DU + rc * ldo + col);
                float2 y = make_float2(acc[rt][0][r], acc[rt][1][r]);
                if (sl > 0) {
                  const float2 o = *dst;
                  y.x += o.x;
                  y.y += o.y;
                }
                *dst = y;
              }
          }
          if (sl + 1 < ksplit) lds_barrier();
        }
      }
    }
    TRACE_STAMP_B(2, 3 + 3 * (top - j));
    lds_barrier();
  }
  TRACE_REAL_B(2, 31);
}

// ------------------------------------------------------------------------------------------------
// Forward + NA/IPW loss + row-local backward in ONE launch (small batches: the latency regime)
// ------------------------------------------------------------------------------------------------
// A workgroup owns LPB = 16 / L WHOLE lists (RB = LPB * L rows of its 16-row MFMA tile), so the listwise loss is
// local to it and the three stages chain inside one kernel: every activation tile x_j stays in LDS from the forward
// to the backward (the copies in `saved` are still written - the weight-gradient kernel reads them), the backward
// needs no prologue of its own (ids, scores, labels, tiles, statistics, gamma/beta are all on chip already), and one
// launch + one dependent kernel boundary disappear.  Same arithmetic as dnn_fwd_kernel + dnn_bwd2_kernel (shared
// building blocks), same outputs: scores, saved, dz_j, vector slabs, loss partials (one per workgroup).
// Chosen only when the grid is at most ONE workgroup per CU (measured, tools/fused_threshold.py: B = 256 lists of 10 -> 62 vs
// 68 us per step; B = 288 -> 94 vs 70 us, a second round of long workgroups); larger batches use the separate kernels,
// whose 16-row tiles are completely live.
__host__ __device__ static inline size_t fb_lds_floats(const DnnPlan& p) {
  const size_t ld = fwd_ld(p.maxdim), ldu = bwd_ldu(p.maxdim);
  return (size_t)16 * ld * (p.nl + 1) + 16 * ldu + (size_t)8 * bwd2_cp_stride(p) + (size_t)p.pv_total + 2 * 16 * (size_t)p.nl +
         2 * 16 + 2 * 8 + 8 +
         (p.h3_ok ? (size_t)16 * (round_up(p.maxdim, 32) + 8) + 8 : 0);  // two fp16 planes [16][ldh] (4 bytes per element)
}

template <int XC, bool H3>
__global__ __launch_bounds__(512) void dnn_fb_kernel(DnnPlan p, BwdPlan bp, const float* __restrict__ params,
                                                     const float* __restrict__ wt, const float* __restrict__ features,
                                                     int64_t n_docs, const int32_t* __restrict__ docids, int B, int L,
                                                     int LPB, float* __restrict__ scores, float* __restrict__ saved,
                                                     float* __restrict__ ws, FusedSoftmax fl, FbPlan fp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int sm_plan[ULTR_MAXL * FbPlan::NFIELD];
  constexpr int R = 16, NW = 8, RT = 1, NT = NW * 64, RPW = R / NW;
  const int64_t N = (int64_t)B * L;
  const int ld = fwd_ld(p.maxdim), ldu = bwd_ldu(p.maxdim), ldz = ld;
  const int cpw = bwd2_cp_stride(p);
  float* XSall = smem;                          // [nl][16][ld]   x_j = input of LayerNorm_j, j = 0..nl-1
  float* UZ = XSall + (size_t)p.nl * R * ld;    // [16][ld]       forward: LayerNorm output (A tile); backward: dz (A tile)
  float* DU = UZ + R * ld;                      // [16][ldu]
  float* CP = DU + R * ldu;                     // [NW][cpw]
  float* PV = CP + NW * cpw;                    // [pv_total]     every vector parameter (the packed image)
  float* sm_mean = PV + p.pv_total;             // [nl][16]
  float* sm_rstd = sm_mean + p.nl * R;          // [nl][16]
  float* sm_s = sm_rstd + p.nl * R;             // [16] scores
  float* sm_ds = sm_s + R;                      // [16]
  float* sm_lt = sm_ds + R;                     // [NW][2]
  // H3: the A tile of every product as two fp16 planes (hi / lo of the row-scaled values) + the per-row output scales
  const int ldh = round_up(p.maxdim, 32) + 8;   // halves per plane row (528-byte rows at 256: conflict-free 16-byte reads)
  _Float16* AH = reinterpret_cast<_Float16*>(sm_lt + 2 * NW);  // [16][ldh]
  _Float16* AL = AH + R * ldh;                                 // [16][ldh]
  __shared__ float sm_os[R];
  const int tid = threadIdx.x, lane_id = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lane = lane_id;  // (re-derived through an opaque move at the top of every layer iteration, see below)
  const int RB = LPB * L;                       // live rows of this block
  const int64_t n0 = (int64_t)blockIdx.x * RB;
  const int b_first = blockIdx.x * LPB;
  const int rows_valid = (int)((N - n0) < RB ? (N - n0) : RB);
  float* vslab = ws + bp.vslab_off + (int64_t)blockIdx.x * bp.vlen;
  const int top = p.nl - 1;
  TRACE_STAMP(0);
  // per-layer records -> LDS (FbPlan, ultr_plan.h): a runtime-indexed read of a by-value kernel argument with a per-thread
  // index is a vector load from the argument segment; visible to every wave behind the prologue's barrier
  if (tid < ULTR_MAXL * FbPlan::NFIELD) sm_plan[tid] = reinterpret_cast<const int*>(&fp)[tid];
  auto rec_of = [&](int jj) { return sm_plan[jj * FbPlan::NFIELD + (lane & (FbPlan::NFIELD - 1))]; };  // lane = field
#define FBF(rv, k) __builtin_amdgcn_readlane((rv), (k))
#define FBF64(rv, k) ((int64_t)(((uint64_t)(uint32_t)FBF(rv, (k) + 1) << 32) | (uint64_t)(uint32_t)FBF(rv, (k))))
  // The plans travel as kernel arguments (~2.7 KB = 43 cache lines in HBM) and are read with scalar loads at the top of every
  // layer of both loops (runtime-indexed records): each first touch of a line was a ~2k-cycle miss on the critical path of
  // every workgroup.  One vector load per workgroup (lane = line) pulls the whole segment into the XCD's L2 from the first
  // cycle; the scalar-cache misses later cost an L2 hit.  The value is never used (kept live to the end of the kernel).
  float ka_pf = 0.f;
  if (wave == 0) {
    const float* ka = (const float*)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int KA_LINES = (int)((sizeof(DnnPlan) + sizeof(BwdPlan) + sizeof(FusedSoftmax) + sizeof(FbPlan) + 96 + 63) / 64);
    static_assert(KA_LINES <= 128, "at most two kernel-argument cache lines per lane");
    ka_pf = ka[(lane < KA_LINES ? lane : 0) * 16];
    if constexpr (KA_LINES > 64) ka_pf += ka[(lane + 64 < KA_LINES ? lane + 64 : 0) * 16];
  }

  // ---- prologue: ids, loss inputs of this wave's list, parameter image, feature rows - all issued back to back -----
  {
    constexpr int PVR = 3, FCH = XC;
    const int F = p.K[0];
    const int rme = wave + NW * (lane < RPW ? lane : 0);
    const bool idok = lane < RPW && rme < rows_valid;
    const uint32_t nme = idok ? (uint32_t)(n0 + rme) : 0u;
    const int myid_raw = docids[(int64_t)(nme % (uint32_t)L) * B + (nme / (uint32_t)L)];
    const Src pvs = make_src(wt + p.wt_pv_off, p.pv_total);
    float4 pvr[PVR];
#pragma unroll
    for (int u = 0; u < PVR; ++u) pvr[u] = buf_ld4(pvs, (unsigned)(tid + u * NT) * 16u);
    const int myid = (idok && myid_raw >= 0 && myid_raw < n_docs) ? myid_raw : -1;
    const Src fs = make_src(features, n_docs * F);
    float4 fr[RPW][FCH];
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const int id = __builtin_amdgcn_readlane(myid, k);
#pragma unroll
      for (int u = 0; u < FCH; ++u) {
        const int c = lane * 4 + 256 * u;
        fr[k][u] = buf_ld4(fs, (id >= 0 && c < F) ? (unsigned)(((int64_t)id * F + c) * 4) : ULTR_OOB);
      }
    }
#pragma unroll
    for (int u = 0; u < PVR; ++u) {
      const int o = (tid + u * NT) * 4;
      if (o < p.pv_total) st4(PV + o, pvr[u]);
    }
    TRACE_STAMP(28);
#pragma unroll
    for (int k = 0; k < RPW; ++k)
#pragma unroll
      for (int u = 0; u < FCH; ++u) {
        const int c = lane * 4 + 256 * u;
        if (c < F) st4(XSall + (wave + NW * k) * ld + c, fr[k][u]);
      }
    TRACE_STAMP(29);
    if (tid < R) sm_ds[tid] = 0.f;
    if (lane < 2) sm_lt[wave * 2 + lane] = 0.f;
  }
  // loss inputs of the wave's first list (lane = position), in flight during the whole forward
  const int li0 = wave;  // list index inside the block handled by this wave (then + NW)
  const bool lact0 = li0 < LPB && b_first + li0 < B && lane < L;
  float y0 = 0.f, pw0 = 1.0f;
  if (lact0) {
    const int b = b_first + li0;
    y0 = fl.labels[(int64_t)lane * B + b];
    if (fl.pw != nullptr) pw0 = fl.pw[(int64_t)b * L + lane];
    else if (fl.ipw != nullptr) pw0 = fl.ipw[lane < fl.n_ipw ? lane : fl.n_ipw - 1];
  }
  lds_barrier();
  TRACE_STAMP(1);

  // =================================== forward ===================================
  for (int j = 0; j < p.nl; ++j) {
    // the lane id through an opaque move per layer (round 5, found in the wide-tile kernels): hipcc otherwise hoists every lane-derived
    // index and predicate of all phases out of the layer loops and carries them - through SGPR / VGPR shuffling - across the kernel;
    // config 2: 47.3 -> 46.9 us per step, fused kernel 22.85 -> 22.46 us in the timed region (three A/B pairs on one box)
    lane = lane_id;
    asm volatile("" : "+v"(lane));
    const int rv = rec_of(j);
    const int K = FBF(rv, FbPlan::K), M = FBF(rv, FbPlan::M);
    const int K32 = round_up(K, 32);
    const bool last = (j == top);
    const float* XS = XSall + (size_t)j * R * ld;
    const float* gs = PV + FBF(rv, FbPlan::PV_OFF);
    const float* bs = gs + K;
    const float* bias = bs + K;
    const float* wlp = PV + p.pv_wlast;
    const float invK = 1.0f / (float)K;
    // ---- LayerNorm_j: XS_j -> UZ (zero-padded to a multiple of 32 columns); the scorer folded into the last one ----
    {
      float4 x[RPW][XC], g4[XC], b4[XC];
      float s[RPW];
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < XC; ++u) {
        const int c = 4 * lane + 256 * u;
        g4[u] = (c < K) ? ld4(gs + c) : z4;
        b4[u] = (c < K) ? ld4(bs + c) : z4;
        if (last) {
          const float4 w4 = (c < K) ? ld4(wlp + c) : z4;
          g4[u].x *= w4.x; g4[u].y *= w4.y; g4[u].z *= w4.z; g4[u].w *= w4.w;
          b4[u].x *= w4.x; b4[u].y *= w4.y; b4[u].z *= w4.z; b4[u].w *= w4.w;
        }
      }
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        const int r = wave + NW * q;
        s[q] = 0.f;
#pragma unroll
        for (int u = 0; u < XC; ++u) {
          const int c = 4 * lane + 256 * u;
          x[q][u] = (c < K) ? ld4(XS + r * ld + c) : z4;
          s[q] += (x[q][u].x + x[q][u].y) + (x[q][u].z + x[q][u].w);
        }
      }
      wave_sum_n<RPW>(s);
      if (j == 0) TRACE_STAMP(30);
      float v[RPW], t[RPW + 1];
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        s[q] *= invK;
        v[q] = 0.f;
        t[q] = 0.f;
#pragma unroll
        for (int u = 0; u < XC; ++u) {
          const int c = 4 * lane + 256 * u;
          float4& xx = x[q][u];
          if (c < K) {
            xx.x -= s[q]; xx.y -= s[q]; xx.z -= s[q]; xx.w -= s[q];
          }
          v[q] += (xx.x * xx.x + xx.y * xx.y) + (xx.z * xx.z + xx.w * xx.w);
          t[q] += (xx.x * g4[u].x + xx.y * g4[u].y) + (xx.z * g4[u].z + xx.w * g4[u].w);
        }
      }
      wave_sum_n<RPW>(v);
      if (j == 0) TRACE_STAMP(31);
      if (last) {
        t[RPW] = 0.f;
#pragma unroll
        for (int u = 0; u < XC; ++u) t[RPW] += (b4[u].x + b4[u].y) + (b4[u].z + b4[u].w);
        wave_sum_n<RPW + 1>(t);
      }
      float4 uq[RPW][XC];  // H3: the rows' LayerNorm outputs wait here for their scale
      float am[RPW];
#pragma unroll
      for (int q = 0; q < RPW; ++q) am[q] = 0.f;
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        const int r = wave + NW * q;
        const float rstd = rsqrt_nr(v[q] * invK + ULTR_LN_EPS);
        if (!last) {
          // the weight gradients' operand goes to HBM from here: u_j, or xhat_0 for the layer-0 shortcut
          const int64_t svx = FBF64(rv, FbPlan::SV_X);
          float* wop = saved + svx + (n0 + r) * K;
          const Src svs = make_src(saved, p.sv_total);
          const unsigned wop_b = (unsigned)((svx + (n0 + r) * K) * 4);
          const bool xhat_only = (j == 0) && bp.l0g != 0;
#pragma unroll
          for (int u = 0; u < XC; ++u) {
            const int c = 4 * lane + 256 * u;
            if (c < K32) {
              const float4 xx = x[q][u];
              const float4 xh = make_float4(xx.x * rstd, xx.y * rstd, xx.z * rstd, xx.w * rstd);
              const float4 uu = make_float4(xh.x * g4[u].x + b4[u].x, xh.y * g4[u].y + b4[u].y, xh.z * g4[u].z + b4[u].z,
                                            xh.w * g4[u].w + b4[u].w);
              if constexpr (H3) {
                uq[q][u] = uu;
                am[q] = fmaxf(am[q], fmaxf(fmaxf(fabsf(uu.x), fabsf(uu.y)), fmaxf(fabsf(uu.z), fabsf(uu.w))));
              } else {
                st4(UZ + r * ld + c, uu);
              }
              if (c < K && r < rows_valid) coh_st4(svs, wop_b + (unsigned)c * 4u, xhat_only ? xh : uu);
            }
          }
        }
        if (lane == 0) {
          const bool valid = r < rows_valid;
          sm_mean[j * R + r] = valid ? s[q] : 0.f;
          sm_rstd[j * R + r] = valid ? rstd : 0.f;
          if (valid) {
            saved[FBF64(rv, FbPlan::SV_MEAN) + n0 + r] = s[q];
            saved[FBF64(rv, FbPlan::SV_RSTD) + n0 + r] = rstd;
          }
          if (last) {
            const float sc = rstd * t[q] + t[RPW] + bias[0];
            sm_s[r] = sc;
            if (valid) scores[n0 + r] = sc;
          }
        }
      }
      if constexpr (H3) {
        if (!last) {
          wave_max_n<RPW>(am);
#pragma unroll
          for (int q = 0; q < RPW; ++q) {
            const int r = wave + NW * q;
            float rs, inv;
            fb_h3_scale(am[q], rs, inv);
#pragma unroll
            for (int u = 0; u < XC; ++u) {
              const int c = 4 * lane + 256 * u;
              if (c < K32) {
                fbh4 hi, lo;
                fb_h3_split4(uq[q][u], rs, hi, lo);
                *reinterpret_cast<fbh4*>(AH + r * ldh + c) = hi;
                *reinterpret_cast<fbh4*>(AL + r * ldh + c) = lo;
              }
            }
            if (lane == 0) sm_os[r] = inv * (1.0f / ULTR_H3_WSCALE);
          }
        }
      }
      if (j == 0) TRACE_STAMP(7);
    }
    lds_barrier();
    TRACE_STAMP(2 + 2 * j);
    if (!last) {
      // ---- Linear_j + activation: UZ . WT_j -> XS_{j+1} (LDS) and saved x_{j+1} (HBM, for the weight gradients) ----
      float* Y = XSall + (size_t)(j + 1) * R * ld;
      float* gout = nullptr;  // x_{j+1} stays on chip; `saved` gets the wgrad operand in the next LayerNorm
      const Src Wt = make_src(wt + FBF64(rv, FbPlan::WT_OFF), (int64_t)K * M);
      const int nch = FBF(rv, FbPlan::NCH), ksplit = FBF(rv, FbPlan::KSPLIT), klen = FBF(rv, FbPlan::KLEN);
      GemmPipe<RT, 2, FWD_D, 0> pipe;
      if constexpr (H3) {
        const int nks = K32 >> 5;
        const Src Wh = make_src(wt + FBF64(rv, FbPlan::WHF_OFF), (int64_t)K32 * M);
        PipeH3<FB_SWD> ph;
        const int c0 = wave * 32;
        ph.begin(Wh, wave, nks, c0 < M, lane);
        for (int cc = c0; cc < M; cc += NW * 32) {
          f32x4 acc[RT][2], accx[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[0][t] = accx[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
          ph.run(AH, AL, ldh, Wh, nks, acc[0], accx, lane);
          if (cc + NW * 32 < M) ph.begin(Wh, (cc + NW * 32) >> 5, nks, true, lane);
          fb_h3_finish(acc, accx, sm_os, lane);
          finish_fwd_nn<RT, 2>(acc, Y, ld, M, cc, lane, bias, p.act, gout, rows_valid);
        }
      } else if (FB_SW && p.sw_ok && ksplit == 1) {
        const int ntr = K32 >> 5;
        const Src Ws = make_src(wt + FBF64(rv, FbPlan::WSF_OFF), (int64_t)K32 * M);
        PipeSw<FB_SWD> ps;
        const int c0 = wave * 32;
        ps.begin(Ws, wave, ntr, 0, ntr, c0 < M, lane);
        for (int cc = c0; cc < M; cc += NW * 32) {
          f32x4 acc[RT][2];
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
          ps.run(UZ, ld, Ws, 0, ntr, acc[0], lane);
          if (cc + NW * 32 < M) ps.begin(Ws, (cc + NW * 32) >> 5, ntr, 0, ntr, true, lane);
          finish_fwd_nn<RT, 2>(acc, Y, ld, M, cc, lane, bias, p.act, gout, rows_valid);
        }
      } else if (ksplit == 1) {
        const int c0 = wave * 32;
        pipe.begin(Wt, M, 0, K, c0, c0 < M, 0, lane);
        for (int cc = c0; cc < M; cc += NW * 32) {
          f32x4 acc[RT][2];
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
          pipe.run(UZ, ld, Wt, 0, K, 0, acc, lane);
          if (cc + NW * 32 < M) pipe.begin(Wt, M, 0, K, cc + NW * 32, true, 0, lane);
          finish_fwd_nn<RT, 2>(acc, Y, ld, M, cc, lane, bias, p.act, gout, rows_valid);
        }
      } else {
        int wq = 0, wr = wave;
        while (wr >= nch) { wr -= nch; ++wq; }
        const int c0 = wr * 32, kb = wq * klen;
        const int ke = (kb + klen < K) ? (kb + klen) : K;
        const bool has = wave < nch * ksplit && kb < ke;
        pipe.begin(Wt, M, kb, ke, c0, has, 0, lane);
        f32x4 acc[RT][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (has) pipe.run(UZ, ld, Wt, kb, ke, 0, acc, lane);
        for (int r = 0; r < ksplit; ++r) {
          if (wave < nch * ksplit && wq == r) store_nn<RT, 2>(acc, Y, ld, M, c0, lane, r > 0);
          lds_barrier();
        }
        const int M4 = M >> 2;
        for (int e = tid; e < R * M4; e += NT) {
          const int row = e / M4, c4 = (e - row * M4) * 4;
          float4 vv = ld4(Y + row * ld + c4);
          const float4 bb = ld4(bias + c4);
          vv.x = act_fwd(vv.x + bb.x, p.act);
          vv.y = act_fwd(vv.y + bb.y, p.act);
          vv.z = act_fwd(vv.z + bb.z, p.act);
          vv.w = act_fwd(vv.w + bb.w, p.act);
          st4(Y + row * ld + c4, vv);
        }
      }
      lds_barrier();
      TRACE_STAMP(3 + 2 * j);
    }
  }

  // =================================== listwise softmax cross entropy ===================================
  for (int li = li0; li < LPB; li += NW) {
    const int b = b_first + li;
    if (b >= B) break;
    const bool act = lane < L;
    float y = y0, pwt = pw0;
    if (li != li0 && act) {
      y = fl.labels[(int64_t)lane * B + b];
      pwt = 1.0f;
      if (fl.pw != nullptr) pwt = fl.pw[(int64_t)b * L + lane];
      else if (fl.ipw != nullptr) pwt = fl.ipw[lane < fl.n_ipw ? lane : fl.n_ipw - 1];
    }
    if (fl.pw == nullptr && fl.ipw != nullptr && !(y > 0.f)) pwt = 0.f;
    const int r = li * L + lane;
    const float sc = act ? sm_s[r] : 0.f;
    const float w = act ? (y + 0.0000001f) * pwt : 0.f;
    const float mx = wave_max(act ? sc : -INFINITY);
    const float S = wave_sum(w);
    const float lse = mx + logf(wave_sum(act ? expf(sc - mx) : 0.f));
    const float dsv = expf(sc - lse) * S - w;
    const float lb = wave_sum(act ? w * (lse - sc) : 0.f);
    if (act) {
      sm_ds[r] = dsv;
      if (fl.dscores_out != nullptr) fl.dscores_out[n0 + r] = dsv;
    }
    if (lane == 0) {
      sm_lt[wave * 2 + 0] += lb;
      sm_lt[wave * 2 + 1] += S;
    }
  }
  lds_barrier();
  {
    const int tail = (int)ultr_tail_len(L);
    for (int t = tid; t < tail; t += NT) {
      float v = 0.f;
      if (t < 2)
        for (int w = 0; w < NW; ++w) v += sm_lt[w * 2 + t];
      fl.loss_part[(int64_t)blockIdx.x * tail + t] = v;
    }
  }
  TRACE_STAMP(16);

  // =================================== backward (as dnn_bwd2_kernel, tiles already on chip) ===================================
  auto finalize = [&](int jj) {
    const int rvf = rec_of(jj);
    const int K = FBF(rvf, FbPlan::K), K4 = round_up(K, 4);
    const int vg = FBF(rvf, FbPlan::VOFF_G), vb = FBF(rvf, FbPlan::VOFF_B);
    const bool lastl = (jj == top);
    for (int c = tid; c < K; c += NT) {
      float pg = 0.f, pb = 0.f, pw = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        pg += CP[w * cpw + c];
        pb += CP[w * cpw + K4 + c];
        if (lastl) pw += CP[w * cpw + 2 * K4 + c];
      }
      vslab[vg + c] = pg;
      vslab[vb + c] = pb;
      if (lastl) vslab[bp.voff_wk + c] = pw;
    }
    if (lastl && tid == 0) {
      float sds = 0.f;
      for (int r = 0; r < R; ++r) sds += sm_ds[r];
      vslab[bp.voff_bk] = sds;
    }
  };
  float* DZ = UZ;
  const int jlow = bp.l0g ? 1 : 0;  // layer-0 shortcut: du_0 is never formed (BwdPlan::l0g)
  for (int j = top; j >= jlow; --j) {
    lane = lane_id;
    asm volatile("" : "+v"(lane));
    const int rv = rec_of(j);
    const int K = FBF(rv, FbPlan::K), M = FBF(rv, FbPlan::M);
    const bool last = (j == top);
    if (!last) {
      finalize(j + 1);
      const Src Wsrc = make_src(params + FBF64(rv, FbPlan::OFF_W), (int64_t)M * K);
      const int nch = FBF(rv, FbPlan::BWD_NCH), msplit = FBF(rv, FbPlan::BWD_MSPLIT), mode = FBF(rv, FbPlan::BWD_MODE);
      if (H3 && j >= 1) {
        // du_j = dz_j . W_j on the fp16 matrix cores: the row pass left dz_j as hi / lo planes with per-row scales
        const int nks = (M + 31) >> 5;
        const Src Wh = make_src(wt + FBF64(rv, FbPlan::WHB_OFF), (int64_t)round_up(M, 32) * round_up(K, 32));
        PipeH3<FB_SWD> ph;
        ph.begin(Wh, wave, nks, wave * 32 < K, lane);
        for (int ch = wave; ch * 32 < K; ch += NW) {
          f32x4 acc[RT][2], accx[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[0][t] = accx[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
          ph.run(AH, AL, ldh, Wh, nks, acc[0], accx, lane);
          if ((ch + NW) * 32 < K) ph.begin(Wh, ch + NW, nks, true, lane);
          // raw sums: the row pass below applies the per-row scale when it reads DU (as dnn_bwd2_kernel)
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[0][t] += accx[t];
          store_nn<RT, 2>(acc, DU, ldu, K, ch * 32, lane, false);
        }
      } else if (FB_SW && p.sw_ok && j >= 1) {
        // du_j = dz_j . W_j over the fragment-major copy of W_j: 32-column chunks of K over the whole contraction M
        const int ntr = (M + 31) >> 5;
        const Src Wb = make_src(wt + FBF64(rv, FbPlan::WSB_OFF), (int64_t)round_up(M, 32) * round_up(K, 32));
        PipeSw<FB_SWD> ps;
        ps.begin(Wb, wave, ntr, 0, ntr, wave * 32 < K, lane);
        for (int ch = wave; ch * 32 < K; ch += NW) {
          f32x4 acc[RT][2];
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
          ps.run(DZ, ldz, Wb, 0, ntr, acc[0], lane);
          if ((ch + NW) * 32 < K) ps.begin(Wb, ch + NW, ntr, 0, ntr, true, lane);
          store_nn<RT, 2>(acc, DU, ldu, K, ch * 32, lane, false);
        }
      } else if (mode == 1) {
        for (int ch = wave; ch * 32 < K; ch += NW) {
          f32x4 acc[RT][2];
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
          gemm_nn<RT, 2, true>(DZ, ldz, Wsrc, K, 0, M, ch * 32, acc, lane);
          store_nn<RT, 2>(acc, DU, ldu, K, ch * 32, lane, false);
        }
      } else if (mode == 2) {
        for (int ch = wave; ch < nch; ch += NW) {
          f32x4 acc[RT][4];
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
          gemm_nn<RT, 4, true>(DZ, ldz, Wsrc, K, 0, M, ch * 64, acc, lane);
          store_nn<RT, 4>(acc, DU, ldu, K, ch * 64, lane, false);
        }
      } else {
        const int mlen = FBF(rv, FbPlan::BWD_MLEN);
        const bool has = wave < nch * msplit;
        int ms = 0, ch = wave;
        while (ch >= nch) { ch -= nch; ++ms; }
        f32x4 acc[RT][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (has) {
          const int mb = ms * mlen;
          const int me = (mb + mlen < M) ? (mb + mlen) : M;
          if (mb < me) gemm_nn<RT, 4, true>(DZ, ldz, Wsrc, K, mb, me, ch * 64, acc, lane);
        }
        for (int r = 0; r < msplit; ++r) {
          if (has && ms == r) store_nn<RT, 4>(acc, DU, ldu, K, ch * 64, lane, r > 0);
          if (r + 1 < msplit) lds_barrier();
        }
      }
      TRACE_STAMP(17 + 4 * (top - j));
      lds_barrier();
    }
    TRACE_STAMP(18 + 4 * (top - j));
    {
      const float* XS = XSall + (size_t)j * R * ld;
      const float* gs = PV + FBF(rv, FbPlan::PV_OFF);
      const float* bs = gs + K;
      const float* wlp = PV + p.pv_wlast;
      const float invK = 1.0f / (float)K;
      float mean[RPW], rstd[RPW], dsr[RPW], dus[RPW];
#pragma unroll
      for (int k = 0; k < RPW; ++k) {
        const int r = wave + NW * k;
        mean[k] = sm_mean[j * R + r];
        rstd[k] = sm_rstd[j * R + r];
        dsr[k] = sm_ds[r];
        dus[k] = (H3 && !last && j >= 1) ? sm_os[r] : 1.0f;  // du_j of the split-half product is stored unscaled
      }
      float4 xk[RPW][XC], gxk[RPW][XC];
      float red[2 * RPW];
#pragma unroll
      for (int k = 0; k < 2 * RPW; ++k) red[k] = 0.f;
#pragma unroll
      for (int u = 0; u < XC; ++u) {
        const int c = 4 * lane + 256 * u;
        const bool act = c < K;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 g4 = act ? ld4(gs + c) : z4;
        const float4 be4 = (act && last) ? ld4(bs + c) : z4;
        const float4 w4 = (act && last) ? ld4(wlp + c) : z4;
        float4 pg = z4, pb = z4, pw = z4;
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
          const int r = wave + NW * k;
          const float4 x4 = act ? ld4(XS + r * ld + c) : z4;
          float4 du4;
          if (last) du4 = make_float4(dsr[k] * w4.x, dsr[k] * w4.y, dsr[k] * w4.z, dsr[k] * w4.w);
          else {
            du4 = act ? ld4(DU + r * ldu + c) : z4;
            if constexpr (H3) { du4.x *= dus[k]; du4.y *= dus[k]; du4.z *= dus[k]; du4.w *= dus[k]; }
          }
          const float4 xh = make_float4((x4.x - mean[k]) * rstd[k], (x4.y - mean[k]) * rstd[k],
                                        (x4.z - mean[k]) * rstd[k], (x4.w - mean[k]) * rstd[k]);
          const float4 gx = make_float4(du4.x * g4.x, du4.y * g4.y, du4.z * g4.z, du4.w * g4.w);
          red[k] += (gx.x + gx.y) + (gx.z + gx.w);
          red[RPW + k] += (gx.x * xh.x + gx.y * xh.y) + (gx.z * xh.z + gx.w * xh.w);
          if (act) {
            pg.x += du4.x * xh.x; pg.y += du4.y * xh.y; pg.z += du4.z * xh.z; pg.w += du4.w * xh.w;
            pb.x += du4.x; pb.y += du4.y; pb.z += du4.z; pb.w += du4.w;
            if (last) {
              pw.x += dsr[k] * (g4.x * xh.x + be4.x); pw.y += dsr[k] * (g4.y * xh.y + be4.y);
              pw.z += dsr[k] * (g4.z * xh.z + be4.z); pw.w += dsr[k] * (g4.w * xh.w + be4.w);
            }
          }
          xk[k][u] = x4;
          gxk[k][u] = gx;
        }
        if (act) {
          const int K4 = round_up(K, 4);
          st4(CP + wave * cpw + c, pg);
          st4(CP + wave * cpw + K4 + c, pb);
          if (last) st4(CP + wave * cpw + 2 * K4 + c, pw);
        }
      }
      if (j > 0) {
        wave_sum_n<2 * RPW>(red);
        const int64_t dzo = FBF64(rec_of(j - 1), FbPlan::DZ_OFF);
        float* dzg = ws + dzo;
        float4 dzq[RPW][XC];  // H3: dz rows wait here for their scale
        float amz[RPW];
#pragma unroll
        for (int k = 0; k < RPW; ++k) amz[k] = 0.f;
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
          const int r = wave + NW * k;
          const float s1 = red[k] * invK, s2 = red[RPW + k] * invK;
#pragma unroll
          for (int u = 0; u < XC; ++u) {
            const int c = 4 * lane + 256 * u;
            if constexpr (H3) dzq[k][u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < K) {
              const float4 x4 = xk[k][u], gx = gxk[k][u];
              float4 dz;
              dz.x = rstd[k] * (gx.x - s1 - (x4.x - mean[k]) * rstd[k] * s2) * act_grad_from_out(x4.x, p.act);
              dz.y = rstd[k] * (gx.y - s1 - (x4.y - mean[k]) * rstd[k] * s2) * act_grad_from_out(x4.y, p.act);
              dz.z = rstd[k] * (gx.z - s1 - (x4.z - mean[k]) * rstd[k] * s2) * act_grad_from_out(x4.z, p.act);
              dz.w = rstd[k] * (gx.w - s1 - (x4.w - mean[k]) * rstd[k] * s2) * act_grad_from_out(x4.w, p.act);
              if constexpr (H3) {
                dzq[k][u] = dz;
                amz[k] = fmaxf(amz[k], fmaxf(fmaxf(fabsf(dz.x), fabsf(dz.y)), fmaxf(fabsf(dz.z), fabsf(dz.w))));
              } else {
                st4(DZ + r * ldz + c, dz);
              }
              if (r < rows_valid) coh_st4(make_src(ws, bp.total), (unsigned)((dzo + (n0 + r) * K + c) * 4), dz);
            }
          }
          if constexpr (!H3)
            for (int c = K + lane; c < round_up(K, 32); c += 64) DZ[r * ldz + c] = 0.f;
        }
        if constexpr (H3) {
          // dz_{j-1} as hi / lo planes for the dgrad product of the next iteration (K = M_{j-1} is a multiple of 32 here)
          wave_max_n<RPW>(amz);
#pragma unroll
          for (int k = 0; k < RPW; ++k) {
            const int r = wave + NW * k;
            float rs, inv;
            fb_h3_scale(amz[k], rs, inv);
#pragma unroll
            for (int u = 0; u < XC; ++u) {
              const int c = 4 * lane + 256 * u;
              if (c < K) {
                fbh4 hi, lo;
                fb_h3_split4(dzq[k][u], rs, hi, lo);
                *reinterpret_cast<fbh4*>(AH + r * ldh + c) = hi;
                *reinterpret_cast<fbh4*>(AL + r * ldh + c) = lo;
              }
            }
            if (lane == 0) sm_os[r] = inv * (1.0f / ULTR_H3_WSCALE);
          }
        }
      }
    }
    TRACE_STAMP(19 + 4 * (top - j));
    lds_barrier();
  }
  finalize(jlow);
  TRACE_STAMP(13);
  asm volatile("" ::"v"(ka_pf));
#undef FBF
#undef FBF64
}

// ------------------------------------------------------------------------------------------------
// Weight gradients of the hidden Linears: dW_j[m,k] = sum_n dz_j[n,m] u_j[n,k],  db_j[m] = sum_n dz_j[n,m]
// ------------------------------------------------------------------------------------------------
// Workgroup = 4 waves on ONE 64x64 output block; each wave contracts a different quarter of the block's row
// split, then the four 64x64 partials are summed through LDS in fixed order and written to the split's slab.
// Per step a lane issues two 16-byte loads (dz row piece along m, x row piece along k) feeding 16 MFMAs:
// A[i][kk] = dz[n+kk][m0+4i+ta], B[kk][j] = u[n+kk][k0+4j+tb]  ->  D_{ta,tb}[i][j] = dW[m0+4i+ta][k0+4j+tb].
#ifndef WG_D
#define WG_D 3  // register sets of the wgrad operand ring (operands requested WG_D - 1 trips ahead)
#endif
template <int N, class F, int... I>
__device__ __forceinline__ void wg_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void wg_static_for(F&& f) {
  wg_static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}
// The spare workgroups of the weight-gradient launch (blockIdx >= bp.wgrad_blocks): vector-slab fold, loss-partial fold + early
// loss report.  Shared by dnn_wgrad_kernel and dnn_wgrad_h3_kernel; 256 threads, `smem` >= 256 floats.
__device__ __forceinline__ void wg_spare_roles(const DnnPlan& p, const BwdPlan& bp, float* __restrict__ smem, float* __restrict__ ws,
                                               float* __restrict__ grads, const float* __restrict__ loss_part, int n_loss_part,
                                               int tail, const EarlyReport& er, const CommDev& cd) {
  if ((int)blockIdx.x >= bp.wgrad_blocks && (int)blockIdx.x < bp.wgrad_blocks + bp.vred_blocks) {
    // spare workgroups: fold the nrb per-row-block vector slabs (LayerNorm gamma/beta, scorer) into ONE slab while
    // the matrix blocks run, so that the reduction kernel's critical path is not a 160-deep serial sum
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int e = ((int)blockIdx.x - bp.wgrad_blocks) * 64 + lane;
    const float part = (e < bp.vlen) ? strided_sum<4>(ws + bp.vslab_off + e, bp.vlen, bp.nrb, grp) : 0.f;
    smem[grp * 64 + lane] = part;
    lds_barrier();
    if (grp == 0 && e < bp.vlen)
      ws[bp.vred_off + e] = ((smem[lane] + smem[64 + lane]) + smem[128 + lane]) + smem[192 + lane];
    return;
  }
  {
    // last spare workgroup(s): fold the loss partials into the step tail grads[P ..] (so the kernels after this one read
    // it with plain loads; the reduction launch then only folds gradient slabs).  More than 1024 partials (one per list
    // for the stand-alone loss stages): bp.lf_chunks workgroups fold bp.lf_len partials each into a scratch row and the
    // reduction launch folds those - a single workgroup would be a serial chain of n / 32 dependent trips
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = (int)blockIdx.x - (bp.wgrad_blocks + bp.vred_blocks);
    const int beg = bp.lf_chunks > 0 ? c * bp.lf_len : 0;
    const int cnt = bp.lf_chunks > 0 ? (n_loss_part - beg < bp.lf_len ? n_loss_part - beg : bp.lf_len) : n_loss_part;
    float* out = bp.lf_chunks > 0 ? ws + bp.lfold_off + (int64_t)c * tail : grads + p.P;
    float head = 0.f;  // group 0, lanes 0..3: loss_sum, D, loss2_sum, D2 of the whole batch
    for (int t0 = 0; t0 < tail; t0 += 64) {
      const int t = t0 + lane;
      smem[grp * 64 + lane] = (t < tail && loss_part != nullptr) ? strided_sum<4>(loss_part + (int64_t)beg * tail + t, tail, cnt, grp) : 0.f;
      lds_barrier();
      if (grp == 0 && t < tail && loss_part != nullptr) {
        const float v = ((smem[lane] + smem[64 + lane]) + smem[128 + lane]) + smem[192 + lane];
        out[t] = v;
        if (t0 == 0) head = v;
      }
      lds_barrier();
    }
    if (er.host != nullptr && grp == 0 && loss_part != nullptr) {
      // early loss report (EarlyReport, ultr_plan.h): the same expressions as update_body, so the update kernel's later
      // report of the same step carries the same bits.  Data parallel (cd.world >= 1): the head of the tail is exchanged with
      // the peers right here (comm_early_head, ultr_comm.h) - the loss needs the GLOBAL sums
      float gh[4];
      if (cd.world >= 1) {
        if (!comm_early_head(cd, head, gh)) return;
      } else {
        gh[0] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(head), 0));
        gh[1] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(head), 1));
        gh[2] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(head), 2));
        gh[3] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(head), 3));
      }
      const float loss_sum = gh[0], D = gh[1], loss2 = gh[2], D2 = gh[3];
      float loss = loss_sum / D;
      if (er.algo == ULTR_ALGO_DLA) loss = loss2 / D2 + er.rlw * (loss_sum / D);
      else if (er.algo == ULTR_ALGO_PAIRDEBIAS) loss = loss_sum;
      if (lane == 0) {
        __hip_atomic_store(er.host, loss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(reinterpret_cast<uint32_t*>(er.host) + 10, er.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    return;
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void dnn_wgrad_kernel(DnnPlan p, BwdPlan bp, const float* __restrict__ params,
                                                        const float* __restrict__ features, int64_t n_docs,
                                                        const int32_t* __restrict__ docids, int B, int L,
                                                        const float* __restrict__ saved, float* __restrict__ ws,
                                                        int vecf, float* __restrict__ grads,
                                                        const float* __restrict__ loss_part, int n_loss_part, int tail,
                                                        EarlyReport er, CommDev cd) {
  // ONE dynamic LDS array: [4][64*64] cross-wave reduction | [4][64] bias partials | [rows_per_split] doc ids
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float (*red)[64 * 64] = reinterpret_cast<float (*)[64 * 64]>(smem);
  float (*bred)[64] = reinterpret_cast<float (*)[64]>(smem + 4 * 64 * 64);
  int* sm_ids = reinterpret_cast<int*>(smem + 4 * 64 * 64 + 4 * 64);
  const int64_t N = bp.N;
  if ((int)blockIdx.x >= bp.wgrad_blocks) {
    wg_spare_roles(p, bp, smem, ws, grads, loss_part, n_loss_part, tail, er, cd);
    return;
  }
  TRACE_STAMP(8);
  int j = 0;
  while (j + 1 < p.nl - 1 && (int)blockIdx.x >= bp.wl[j + 1].blk_begin) ++j;
  const WgradLayer wl = bp.wl[j];
  const int local = blockIdx.x - wl.blk_begin;
  const int split = local % wl.nsplit;
  const int tile = local / wl.nsplit;
  const int mb = tile / wl.nkb, kb = tile % wl.nkb;
  const int M = wl.M, K = wl.K;
  const int m0 = mb * 64, k0 = kb * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave: SGPR
  const int i = lane & 15, q = lane >> 4;
  const bool vec = wl.vec != 0;
  const int rpw = wl.rows_per_split / 4;
  const int64_t nbeg = (int64_t)split * wl.rows_per_split + (int64_t)wave * rpw;
  int64_t nend = nbeg + rpw;
  if (nend > N) nend = N;

  const Src dz = make_src(ws + wl.dz_off, N * M);
  // `saved` holds the ready-made operand (no ids, no gather, no transform): every layer after the fused kernel (wg_prenorm);
  // layer 0 alone after a per-layer forward that wrote xhat_0 (it says so in the marker word behind the saved activations -
  // every forward writes that word, so the two calls cannot disagree)
  const bool prenorm = bp.wg_prenorm != 0 || (VEC && j == 0 && bp.l0g != 0 && saved[p.sv_total] != 0.f);
  const Src xs = (j == 0 && !prenorm) ? make_src(features, n_docs * K) : make_src(saved + p.sv_x[j], N * K);
  const Src meansrc = make_src(saved + p.sv_mean[j], N);
  const Src rstdsrc = make_src(saved + p.sv_rstd[j], N);
  const int64_t nsplit0 = (int64_t)split * wl.rows_per_split;
  if (j == 0 && !prenorm) {
    // layer 0 reads feature rows through the doc ids: resolve them once into LDS so that the main loop has no
    // dependent global load (a docid -> row chain forces vmcnt(0) and drains the prefetch ring)
    for (int r = tid; r < wl.rows_per_split; r += 256) {
      const int64_t n = nsplit0 + r;
      int id = -1;
      if (n < N) {
        const int b = (int)(n / L), l = (int)(n % L);
        const int64_t d = docids[(int64_t)l * B + b];
        if (d >= 0 && d < n_docs) id = (int)d;
      }
      sm_ids[r] = id;
    }
    lds_barrier();
  }
  TRACE_STAMP(9);
  const bool l0g = (j == 0) && bp.l0g != 0;  // layer-0 shortcut: contract with xhat, apply gamma/beta in the epilogue
  const float4 gam = l0g ? make_float4(1.f, 1.f, 1.f, 1.f) : ld4_masked(params + p.off_lnw[j], k0 + 4 * i, K, false);
  const float4 bet = l0g ? make_float4(0.f, 0.f, 0.f, 0.f) : ld4_masked(params + p.off_lnb[j], k0 + 4 * i, K, false);
  // layer-0 shortcut: the epilogue's operands (this thread's four W_0 pieces, gamma_0, beta_0) are requested NOW and ride
  // through the main loop in registers - fetched in the epilogue they added ~4k cycles of exposed latency to its tail
  float4 l0w[4], l0g4 = make_float4(0.f, 0.f, 0.f, 0.f), l0b4 = l0g4;
#pragma unroll
  for (int it = 0; it < 4; ++it) l0w[it] = l0g4;
  if (l0g) {
    const int kq = k0 + (tid & 15) * 4;
    l0g4 = ld4_masked(params + p.off_lnw[0], kq, K, wl.vec != 0);
    l0b4 = ld4_masked(params + p.off_lnb[0], kq, K, wl.vec != 0);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int m = m0 + ((tid + 256 * it) >> 4);
      if (m < M) l0w[it] = ld4_masked(params + p.off_w[0] + (int64_t)m * K, kq, K, wl.vec != 0);
    }
  }
  const int kc = k0 + 4 * i;
  const bool k_ok0 = kc < K, k_ok1 = kc + 1 < K, k_ok2 = kc + 2 < K, k_ok3 = kc + 3 < K;

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

  // Raw operands are kept in the prefetch ring and the LayerNorm transform is applied when a step is CONSUMED:
  // transforming at load time would make every load's first use immediate and drain the ring (measured: ~2.7k
  // cycles per 16-MFMA step, one exposed memory latency each).
  auto mainloop = [&](auto layer0_tag, auto prenorm_tag) {
  constexpr bool LAYER0 = decltype(layer0_tag)::value;
  constexpr bool PRENORM = decltype(prenorm_tag)::value;  // operand ready-made in `saved`: two loads per step, no transform
  auto load_step = [&](int64_t n, float4& a4, float4& x4, float& mean, float& rstd) {
    const bool ok = n < nend;
    if constexpr (VEC) {
      // only dz must be exactly zero for rows outside this wave's slice; x / statistics of such rows are finite
      // (other rows of the batch) or hardware-zeroed (past N), and their products meet a4 == 0.  PAD documents
      // (id < 0) must read as the all-zero feature row -> out-of-bounds offset.
      a4 = buf_ld4(dz, ok ? (unsigned)(n * M + m0 + 4 * i) * 4u : ULTR_OOB);
      if constexpr (LAYER0) {
        const int id = ok ? sm_ids[(int)(n - nsplit0)] : -1;
        x4 = buf_ld4(xs, id >= 0 ? (unsigned)((int64_t)id * K + kc) * 4u : ULTR_OOB);
      } else {
        x4 = buf_ld4(xs, (unsigned)(n * K + kc) * 4u);
      }
      if constexpr (PRENORM) {
        mean = 0.f;
        rstd = 1.f;
      } else {
        mean = buf_ld1(meansrc, (unsigned)n * 4u);
        rstd = buf_ld1(rstdsrc, (unsigned)n * 4u);
      }
    } else {
      a4 = ld4_sel<VEC>(dz, n * M, ok, m0 + 4 * i, M);
      if constexpr (LAYER0) {
        const int id = ok ? sm_ids[(int)(n - nsplit0)] : -1;
        x4 = ld4_sel<VEC>(xs, (int64_t)id * K, id >= 0, kc, K);
      } else {
        x4 = ld4_sel<VEC>(xs, n * K, ok, kc, K);
      }
      mean = ld1_sel<VEC>(meansrc, n, ok);
      rstd = ld1_sel<VEC>(rstdsrc, n, ok);
    }
  };

  // Straight-line software pipeline (same shape as gemm_nn): a trip consumes TWO steps (8 rows, 32 MFMAs) from one
  // register set while the next trip's operands are already in flight into the other; no control flow in the
  // steady state.  Steps past the wave's slice load dz through the out-of-bounds offset (zeros): wasted MFMAs, no
  // wrong sums - the host rounds rows_per_split to a multiple of 32 so that there are none in the common case.
  struct StepRegs {
    float4 a, x;
    float mean, rstd;
  };
  constexpr int SPT = 2;  // steps per trip: 8 rows, 32 MFMAs (4 was measured no faster)
  int64_t nn = nbeg;
  // consume `cu` (trip t) while the operands of trip t + WG_D - 1 go in flight into `nx`
  auto trip = [&](StepRegs(&cu)[SPT], StepRegs(&nx)[SPT]) {
#pragma unroll
    for (int u = 0; u < SPT; ++u)
      load_step(nn + 4 * SPT * (WG_D - 1) + 4 * u + q, nx[u].a, nx[u].x, nx[u].mean, nx[u].rstd);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < SPT; ++u) {
      const float4 a_c = cu[u].a, x_c = cu[u].x;
      const float mean = cu[u].mean, rstd = cu[u].rstd;
      bsum.x += a_c.x;
      bsum.y += a_c.y;
      bsum.z += a_c.z;
      bsum.w += a_c.w;
      const float av[4] = {a_c.x, a_c.y, a_c.z, a_c.w};
      float bv[4];
      if constexpr (PRENORM) {
        bv[0] = x_c.x; bv[1] = x_c.y; bv[2] = x_c.z; bv[3] = x_c.w;
      } else {
        bv[0] = (VEC || k_ok0) ? ((x_c.x - mean) * rstd * gam.x + bet.x) : 0.f;
        bv[1] = (VEC || k_ok1) ? ((x_c.y - mean) * rstd * gam.y + bet.y) : 0.f;
        bv[2] = (VEC || k_ok2) ? ((x_c.z - mean) * rstd * gam.z + bet.z) : 0.f;
        bv[3] = (VEC || k_ok3) ? ((x_c.w - mean) * rstd * gam.w + bet.w) : 0.f;
      }
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = mfma16(av[ta], bv[tb], acc[ta][tb]);
    }
    nn += 4 * SPT;
  };
  const int ntrip = (int)((nend - nbeg + 4 * SPT - 1) / (4 * SPT));
  // WG_D register sets in a ring: the operands of trip t + WG_D - 1 are requested while trip t is consumed - the dz / x rows
  // were written by the previous launches, mostly on other XCDs, and come from beyond the local L2
  StepRegs r[WG_D][SPT];
#pragma unroll
  for (int d = 0; d < WG_D - 1; ++d)
#pragma unroll
    for (int u = 0; u < SPT; ++u) load_step(nbeg + 4 * SPT * d + 4 * u + q, r[d][u].a, r[d][u].x, r[d][u].mean, r[d][u].rstd);
  int t = 0;
  for (; t + WG_D <= ntrip; t += WG_D)
    wg_static_for<WG_D>([&](auto I) { trip(r[decltype(I)::value], r[(decltype(I)::value + WG_D - 1) % WG_D]); });
  wg_static_for<WG_D - 1>([&](auto I) {
    if (t + decltype(I)::value < ntrip) trip(r[decltype(I)::value], r[(decltype(I)::value + WG_D - 1) % WG_D]);
  });
  };  // mainloop
  // the layer-0 variant (doc ids -> feature rows through LDS) and the plain variant are separate straight-line
  // loops: a branch on j inside the loop would put the loads in control flow and drain vmcnt(0) every step
  if constexpr (VEC) {
    if (prenorm) mainloop(std::false_type{}, std::true_type{});
    else if (j == 0) mainloop(std::true_type{}, std::false_type{});
    else mainloop(std::false_type{}, std::false_type{});
  } else {
    if (j == 0) mainloop(std::true_type{}, std::false_type{});
    else mainloop(std::false_type{}, std::false_type{});
  }
  TRACE_STAMP(10);
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
  lds_barrier();
  TRACE_STAMP(11);
  float* slab = ws + wl.slab_off + (int64_t)split * ((int64_t)M * K + M);
  float4 l0pg = make_float4(0.f, 0.f, 0.f, 0.f), l0pb = make_float4(0.f, 0.f, 0.f, 0.f);
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
    if (l0g && m < M && k < K) {
      // G -> dW_0 = gamma o G + S_m * beta;  partial column sums of W_0 o G and W_0 * S_m for d gamma_0 / d beta_0
      const float Sm = ((bred[0][ml] + bred[1][ml]) + bred[2][ml]) + bred[3][ml];
      const float4 g4 = l0g4, b4 = l0b4, w4 = l0w[it];
      l0pg.x += w4.x * s.x; l0pg.y += w4.y * s.y; l0pg.z += w4.z * s.z; l0pg.w += w4.w * s.w;
      l0pb.x += w4.x * Sm; l0pb.y += w4.y * Sm; l0pb.z += w4.z * Sm; l0pb.w += w4.w * Sm;
      s.x = g4.x * s.x + b4.x * Sm; s.y = g4.y * s.y + b4.y * Sm; s.z = g4.z * s.z + b4.z * Sm; s.w = g4.w * s.w + b4.w * Sm;
    }
    if (m < M && k < K) {
      float* dst = slab + (int64_t)m * K + k;
      if (vec && k + 3 < K) {
        st4_stream(dst, s);
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
  if (l0g) {
    // fold the 16 row groups (tid >> 4) of this block in fixed order: 64 columns x {d gamma_0, d beta_0} partials
    lds_barrier();  // everyone is done reading `red`
    float* pgs = &red[0][0];         // [16][64]
    float* pbs = pgs + 16 * 64;      // [16][64]
    const int grp16 = tid >> 4, c4 = (tid & 15) * 4;
    st4(pgs + grp16 * 64 + c4, l0pg);
    st4(pbs + grp16 * 64 + c4, l0pb);
    lds_barrier();
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63;
      const float* src = which ? pbs : pgs;
      float a = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) a += src[g * 64 + c];
      if (k0 + c < K) ws[bp.l0part_off + ((int64_t)(mb * wl.nsplit + split) * 2 + which) * K + k0 + c] = a;
    }
  }
  TRACE_STAMP(12);
}

// ------------------------------------------------------------------------------------------------
// Weight gradients on the fp16 matrix cores with split (hi / lo) operands   (BwdPlan::wg_h3)
// ------------------------------------------------------------------------------------------------
// dnn_wgrad_kernel contracts on v_mfma_f32_16x16x4_f32 straight out of registers: 8 x 32 matrix-core cycles per 16 x 16 x 32 step
// and every dz / u element fetched from L2 once per 64 columns of the other operand (config 4: 125 us, 68 % of the fp32 matrix
// peak).  Here dW_j = dz_j^T u_j runs on v_mfma_f32_16x16x32_f16 with both operands split, a.b = ah.bh + ah.bl + al.bh (22 bits of
// mantissa each, fp32 accumulation): 3 x 16 cycles for the same step, on 128 x 128 blocks staged through LDS (a quarter of the
// L2 -> CU traffic).  What has to be different from the forward / dgrad products (PipeH3): the contraction index is the ROW here,
// so a per-row scale does not factor out of the sum.  The scale is per (half-block, wave group) instead - one power of two for a
// 32-row x 64-column block of an operand, chosen from the block's largest magnitude (< 2^14 after scaling) and only ever lowered
// while the group walks its rows: when a later block raises the maximum the accumulators are multiplied by the (exact) ratio and
// the walk goes on.  An element keeps 1e-5 relative accuracy down to 2^-22 of the largest element the group has seen in its 64
// columns; below that its error is 2^-25 on the scale of that maximum, i.e. invisible in a sum that contains the large terms
// (DESIGN.md section 4).
// Workgroup = TWO groups of 4 waves on a 128 (m) x 128 (k) block of ONE dW_j and one row split; group g takes the 32-row steps
// t = g, g + 2, ... with its own planes and its own accumulators (summed through LDS at the end: an in-workgroup row split that
// costs no slab).  A step of a group is two phases, each closed by ONE workgroup barrier:
//   stage:    wave w of the group takes the 32 x 64 fp32 half-block it loaded two steps earlier (w = 0, 1: dz columns
//             m0 + 64 w ..; w = 2, 3: u columns k0 + 64 (w - 2) ..; a lane holds 8 rows x 4 columns, so the transposition into the
//             MFMA operand order - 8 consecutive rows of one column = 16 bytes - happens in registers), applies LayerNorm where
//             `saved` holds x_j, finds the block maximum, splits, writes the two fp16 planes ([column][32 rows]) and requests the
//             half-block two steps ahead;
//   multiply: wave (wm, wk) = (w >> 1, w & 1) multiplies its 64 x 64 sub-block: 16 ds_read_b128 + 48 MFMAs.
// The groups run in ANTI-PHASE (group 1 starts one phase late): while one group's waves convert and write LDS the other group's
// waves keep the matrix cores busy, by construction - two independent 4-wave workgroups per CU (the first version) drifted in
// and out of phase and left the matrix cores 70 % idle.  The epilogue is dnn_wgrad_kernel's (slabs per row split, bias sums,
// layer-0 gamma / beta fold), so the reduction launch and everything behind it are unchanged.
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
struct WhStep {
  u32x4 v[8];
};
__global__ __launch_bounds__(512) void dnn_wgrad_h3_kernel(DnnPlan p, BwdPlan bp, const float* __restrict__ params,
                                                           const float* __restrict__ features, int64_t n_docs,
                                                           const int32_t* __restrict__ docids, int B, int L,
                                                           const float* __restrict__ saved, float* __restrict__ ws,
                                                           float* __restrict__ grads, const float* __restrict__ loss_part,
                                                           int n_loss_part, int tail, EarlyReport er, CommDev cd) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x >= bp.wgrad_blocks) {
    if (threadIdx.x >= 256) return;  // (the spare roles are written for four waves)
    wg_spare_roles(p, bp, smem, ws, grads, loss_part, n_loss_part, tail, er, cd);
    return;
  }
  const int lin = ((int)blockIdx.x & 7) * bp.wg_chunk + ((int)blockIdx.x >> 3);  // BwdPlan::wg_chunk
  if (lin >= bp.wg_live) return;
  const int split = lin / bp.wg_tiles2;
  const int tix = lin - split * bp.wg_tiles2;
  int j = 0;
  while (j + 1 < p.nl - 1 && tix >= bp.wl[j + 1].blk_begin) ++j;
  const WgradLayer wl = bp.wl[j];
  const int tile = tix - wl.blk_begin;
  const int mb2 = tile / wl.nkb2, kb2 = tile - mb2 * wl.nkb2;
  const int M = wl.M, K = wl.K;
  const int m0 = mb2 * 128, k0 = kb2 * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wv = wave & 3;  // wave group, wave of the group
  const int64_t N = bp.N;
  const int rps = wl.rows_per_split;
  const int64_t nbeg = (int64_t)split * rps;
  const int rows = (int)((N - nbeg) < (int64_t)rps ? (N - nbeg) : (int64_t)rps);
  const int nsteps = (rows + 31) >> 5;
  const bool prenorm = bp.wg_prenorm != 0 || (j == 0 && bp.l0g != 0 && saved[p.sv_total] != 0.f);
  const bool l0g = (j == 0) && bp.l0g != 0;
  const bool gather = (j == 0) && !prenorm;
  const bool xform = !prenorm;
  // ---- LDS: planes [2 groups][4 half-blocks][hi, lo][64 columns][WH_LDH] halves | per-row tables; the epilogue's four 64 x 64
  // fp32 blocks overlay the planes; behind everything: exponents, bias sums
  _Float16* planes = reinterpret_cast<_Float16*>(smem) + (size_t)g * WH_GROUP_HALVES;
  float2* sm_stat = reinterpret_cast<float2*>(smem + WH_PLANES_BYTES / 4);                   // [WH_TAB_ROWS] (mean, rstd)
  int* sm_ids = reinterpret_cast<int*>(smem + WH_PLANES_BYTES / 4 + 2 * WH_TAB_ROWS);        // [WH_TAB_ROWS]
  int* sm_se = reinterpret_cast<int*>(smem + WH_MAIN_BYTES / 4);                             // [8] final scale exponents
  int* sm_bump = sm_se + 8;                                                                  // [8] exponent decrease of the step in LDS
  float* sm_bsum = smem + WH_MAIN_BYTES / 4 + 16;                                            // [2 groups][2][64]
  if (xform) {
    const float* mp = saved + p.sv_mean[j];
    const float* rp = saved + p.sv_rstd[j];
    for (int r = tid; r < 32 * (nsteps + 2); r += 512) sm_stat[r] = (r < rows) ? make_float2(mp[nbeg + r], rp[nbeg + r]) : make_float2(0.f, 0.f);
  }
  if (gather) {
    for (int r = tid; r < 32 * (nsteps + 5); r += 512) {
      int id = -1;
      if (r < rows) {
        const int64_t n = nbeg + r;
        const int b = (int)(n / L), l = (int)(n % L);
        const int64_t d = docids[(int64_t)l * B + b];
        if (d >= 0 && d < n_docs) id = (int)d;
      }
      sm_ids[r] = id;
    }
  }
  // ---- staging role of this wave: one 32-row x 64-column half-block per step of its group -------------------------------------
  const bool isA = wv < 2;
  const int c16 = lane & 15, rg = lane >> 4;
  const int ncols = isA ? M : K;
  const int col = (isA ? m0 + 64 * wv : k0 + 64 * (wv - 2)) + 4 * c16;
  const bool colok = col < ncols;
  // the buffer ends with this split's last row: rows of the tail step beyond it read as zeros, no per-row predicate
  const Src src = isA ? make_src(ws + wl.dz_off, (nbeg + rows) * M)
                      : (gather ? make_src(features, n_docs * K) : make_src(saved + p.sv_x[j], (nbeg + rows) * K));
  const unsigned stride = (unsigned)ncols * 4u;
  unsigned vo = colok ? (unsigned)(((nbeg + 32 * g + 8 * rg) * ncols + col) * 4) : ULTR_OOB;  // advanced by 64 rows per load_step
  int tl = g;  // step the next load_step fetches
  int tc = g;  // step the next convert takes
  float4 gam = make_float4(0.f, 0.f, 0.f, 0.f), bet = gam;
  if (!isA && xform && colok) {
    if (l0g) gam = make_float4(1.f, 1.f, 1.f, 1.f);
    else {
      gam = ld4(params + p.off_lnw[j] + col);
      bet = ld4(params + p.off_lnb[j] + col);
    }
  }
  const int swz_w = c16 & 3;  // (column >> 2) & 3 of the lane's four columns
  _Float16* myplane = planes + (size_t)wv * 2 * 64 * WH_LDH + (4 * c16) * WH_LDH + 8 * (rg ^ swz_w);
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  int se_run = 253;  // biased exponent of the running scale 2^(se - 127)
  fbh8 ch[4], cl[4];
  int bump = 0;
  // ---- compute role: the 64 x 64 sub-block (wm, wk) ---------------------------------------------------------------------------
  const int wm = wv >> 1, wk = wv & 1;
  const int i = lane & 15, q = lane >> 4;
  const int swz_r = (i >> 2) & 3;
  const _Float16* pa = planes + (size_t)wm * 2 * 64 * WH_LDH + i * WH_LDH + 8 * (q ^ swz_r);
  const _Float16* pb = planes + (size_t)(2 + wk) * 2 * 64 * WH_LDH + i * WH_LDH + 8 * (q ^ swz_r);
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  TRACE_STAMP(0);
  lds_barrier();  // tables
  TRACE_STAMP(1);
  const int S = (nsteps - g + 1) >> 1, S0 = (nsteps + 1) >> 1;  // steps of this group / of group 0

  auto mainloop = [&](auto isa_tag, auto xf_tag, auto ga_tag) __attribute__((always_inline)) {
    constexpr bool ISA = decltype(isa_tag)::value, XFORM = decltype(xf_tag)::value, GATHER = decltype(ga_tag)::value;
    auto load_step = [&](WhStep& s) __attribute__((always_inline)) {
      if constexpr (GATHER) {
        const int4 ia = *reinterpret_cast<const int4*>(sm_ids + 32 * tl + 8 * rg);
        const int4 ib = *reinterpret_cast<const int4*>(sm_ids + 32 * tl + 8 * rg + 4);
        const int id[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          unsigned off = (id[r] >= 0 && colok) ? (unsigned)(((int64_t)id[r] * K + col) * 4) : ULTR_OOB;
          s.v[r] = __builtin_amdgcn_raw_buffer_load_b128(src.rs, off, 0, 0);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          s.v[r] = __builtin_amdgcn_raw_buffer_load_b128(src.rs, vo, (unsigned)r * stride, 0);
        }
        vo += 64u * stride;
      }
      tl += 2;
    };
    // scale + split of a half-block into ch / cl; `bump` = how far the running scale went down
    auto convert = [&](const WhStep& s) __attribute__((always_inline)) {
      float v[8][4];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        v[r][0] = __uint_as_float(s.v[r].x); v[r][1] = __uint_as_float(s.v[r].y);
        v[r][2] = __uint_as_float(s.v[r].z); v[r][3] = __uint_as_float(s.v[r].w);
      }
      if constexpr (ISA) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) bsum[c] += v[r][c];
      } else if constexpr (XFORM) {
        const float gg[4] = {gam.x, gam.y, gam.z, gam.w}, be[4] = {bet.x, bet.y, bet.z, bet.w};
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
          const float4 st = *reinterpret_cast<const float4*>(sm_stat + 32 * tc + 8 * rg + r);  // (mean, rstd) of two rows
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            v[r][c] = (v[r][c] - st.x) * (st.y * gg[c]) + be[c];
            v[r + 1][c] = (v[r + 1][c] - st.z) * (st.w * gg[c]) + be[c];
          }
        }
      }
      tc += 2;
      float am = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) am = fmaxf(am, fabsf(v[r][c]));
      am = wave_max(am);
      int se = 267 - (int)((__float_as_uint(am) >> 23) & 0xffu);  // am * 2^(se - 127) < 2^14  (fb_h3_scale)
      se = __builtin_amdgcn_readfirstlane(se);
      se = se < 1 ? 1 : se;
      const int lower = se < se_run ? se : se_run;
      bump = se_run - lower;
      se_run = lower;
      const float rs = __uint_as_float((unsigned)se_run << 23);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float a = v[r][c] * rs;
          const _Float16 hi = (_Float16)a;
          ch[c][r] = hi;
          cl[c][r] = (_Float16)(a - (float)hi);
        }
    };
    auto stage = [&](WhStep& slot) __attribute__((always_inline)) {
      convert(slot);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        *reinterpret_cast<fbh8*>(myplane + c * WH_LDH) = ch[c];
        *reinterpret_cast<fbh8*>(myplane + 64 * WH_LDH + c * WH_LDH) = cl[c];
      }
      if (lane == 0) sm_bump[wave] = bump;
    };
    // ... and the request for the half-block two steps of the group ahead goes out of the MULTIPLY phase (the slot was converted in
    // the phase before; past the end: beyond the buffer - zeros, no traffic): issuing 8 x 1 KiB per wave takes as long as the
    // conversion, and in the stage phase it made that phase twice as long as the products it is meant to hide behind
    auto multiply = [&](WhStep& slot, const WhStep& other) __attribute__((always_inline)) {
      const int d = __builtin_amdgcn_readfirstlane(sm_bump[4 * g + wm] + sm_bump[4 * g + 2 + wk]);
      if (d != 0) {  // an operand's scale went down by 2^d: bring the sums along (exact)
        const float f = d > 126 ? 0.f : __uint_as_float((unsigned)(127 - d) << 23);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] *= f;
      }
      fbh8 bh[4], bl[4];
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) {
        bh[tb] = *reinterpret_cast<const fbh8*>(pb + tb * 16 * WH_LDH);
        bl[tb] = *reinterpret_cast<const fbh8*>(pb + 64 * WH_LDH + tb * 16 * WH_LDH);
      }
#pragma unroll
      for (int ta = 0; ta < 4; ++ta) {
        const fbh8 ah = *reinterpret_cast<const fbh8*>(pa + ta * 16 * WH_LDH);
        const fbh8 al = *reinterpret_cast<const fbh8*>(pa + 64 * WH_LDH + ta * 16 * WH_LDH);
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = fb_mfma_h(ah, bh[tb], acc[ta][tb]);
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = fb_mfma_h(ah, bl[tb], acc[ta][tb]);
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = fb_mfma_h(al, bh[tb], acc[ta][tb]);
      }
      load_step(slot);
    };
    WhStep r0, r1;
    load_step(r0);
    load_step(r1);
    if (g == 1) lds_barrier();  // group 1 runs one phase behind group 0
    for (int s = 0; s < S; s += 2) {
      if (s < 6) TRACE_STAMP(2 + 4 * s);
      stage(r0);
      lds_barrier();
      if (s < 6) TRACE_STAMP(3 + 4 * s);
      multiply(r0, r1);
      lds_barrier();
      if (s < 6) TRACE_STAMP(4 + 4 * s);
      if (s + 1 >= S) break;
      stage(r1);
      lds_barrier();
      if (s < 6) TRACE_STAMP(5 + 4 * s);
      multiply(r1, r0);
      lds_barrier();
    }
  };
  if (isA) mainloop(std::true_type{}, std::false_type{}, std::false_type{});
  else if (!xform) mainloop(std::false_type{}, std::false_type{}, std::false_type{});
  else if (!gather) mainloop(std::false_type{}, std::true_type{}, std::false_type{});
  else mainloop(std::false_type{}, std::true_type{}, std::true_type{});
  // every wave passes 2 S0 + 1 barriers in the walk: group 0 is one short, group 1 two per step it has fewer than group 0
  for (int n = (g == 0) ? 1 : 2 * (S0 - S); n > 0; --n) lds_barrier();
  // ---- epilogue ---------------------------------------------------------------------------------------------------------------
  TRACE_STAMP(30);
  if (lane == 0) sm_se[wave] = se_run;
  if (isA) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      bsum[c] += __shfl_xor(bsum[c], 16, 64);
      bsum[c] += __shfl_xor(bsum[c], 32, 64);
    }
    if (rg == 0) st4(sm_bsum + 128 * g + 64 * wv + 4 * c16, make_float4(bsum[0], bsum[1], bsum[2], bsum[3]));
  }
  // layer-0 fold: this thread's pieces of W_0 for the four sub-blocks, requested now (two per sub-block with 512 threads)
  const int kq0 = k0 + (tid & 15) * 4;
  float4 w4[4][2];
  if (l0g) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int m = m0 + 64 * (s >> 1) + (tid >> 4) + 32 * it, kq = kq0 + 64 * (s & 1);
        w4[s][it] = (m < M && kq < K) ? ld4(params + p.off_w[0] + (int64_t)m * K + kq) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
  }
  lds_barrier();  // last products read, exponents and bias sums visible: the planes may be overwritten
  const float ia = __uint_as_float((unsigned)(254 - sm_se[4 * g + wm]) << 23), ib = __uint_as_float((unsigned)(254 - sm_se[4 * g + 2 + wk]) << 23);
  float* redw = smem + wv * 4096;
  if (g == 1) {
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
      for (int tb = 0; tb < 4; ++tb)
#pragma unroll
        for (int r = 0; r < 4; ++r) redw[(16 * ta + 4 * q + r) * 64 + 16 * tb + i] = (acc[ta][tb][r] * ia) * ib;
  }
  lds_barrier();
  if (g == 0) {
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
      for (int tb = 0; tb < 4; ++tb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* e = redw + (16 * ta + 4 * q + r) * 64 + 16 * tb + i;
          *e = (acc[ta][tb][r] * ia) * ib + *e;
        }
  }
  lds_barrier();
  float* slab = ws + wl.slab_off + (int64_t)split * ((int64_t)M * K + M);
  // layer-0 fold: the per-thread column partials of all four sub-blocks stay in registers through the slab writes and are folded in
  // ONE pass behind them (two barriers; per sub-block it was three barriers and a 32-term sum by a quarter of the threads each time:
  // the layer-0 workgroups - 8 of config 3's 18 tiles, 24 of config 4's 34 - ended 8k cycles after the others)
  float4 l0pg[4], l0pb[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int sm_ = s >> 1, sk = s & 1;
    const int mB = m0 + 64 * sm_, kB = k0 + 64 * sk;
    l0pg[s] = l0pb[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mB >= M || kB >= K) continue;  // (uniform)
    const float* red = smem + s * 4096;
    const int kq = kB + (tid & 15) * 4;
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f), b4 = g4;
    if (l0g && kq < K) {
      g4 = ld4(params + p.off_lnw[0] + kq);
      b4 = ld4(params + p.off_lnb[0] + kq);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int ml = (tid >> 4) + 32 * it;
      const int m = mB + ml;
      float4 v = ld4(red + ml * 64 + (tid & 15) * 4);
      if (m < M && kq < K) {
        if (l0g) {
          // G -> dW_0 = gamma o G + S_m * beta;  partial column sums of W_0 o G and W_0 * S_m for d gamma_0 / d beta_0
          const float Sm = sm_bsum[64 * sm_ + ml] + sm_bsum[128 + 64 * sm_ + ml];
          const float4 w = w4[s][it];
          l0pg[s].x += w.x * v.x; l0pg[s].y += w.y * v.y; l0pg[s].z += w.z * v.z; l0pg[s].w += w.w * v.w;
          l0pb[s].x += w.x * Sm; l0pb[s].y += w.y * Sm; l0pb[s].z += w.z * Sm; l0pb[s].w += w.w * Sm;
          v.x = g4.x * v.x + b4.x * Sm; v.y = g4.y * v.y + b4.y * Sm; v.z = g4.z * v.z + b4.z * Sm; v.w = g4.w * v.w + b4.w * Sm;
        }
        st4_stream(slab + (int64_t)m * K + kq, v);
      }
    }
    if (kb2 == 0 && sk == 0 && tid < 64 && mB + tid < M) slab[(int64_t)M * K + mB + tid] = sm_bsum[64 * sm_ + tid] + sm_bsum[128 + 64 * sm_ + tid];
  }
  if (l0g) {
    // fold scratch: the four sub-blocks' overlay (everyone is past reading it behind this barrier): [sub-block][pg | pb][32 row groups][64]
    lds_barrier();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      st4(smem + s * 4096 + (tid >> 4) * 64 + (tid & 15) * 4, l0pg[s]);
      st4(smem + s * 4096 + 2048 + (tid >> 4) * 64 + (tid & 15) * 4, l0pb[s]);
    }
    lds_barrier();
    {
      const int s = tid >> 7, which = (tid >> 6) & 1, c = tid & 63;  // 512 threads = 4 sub-blocks x {d gamma, d beta} x 64 columns
      const int sm_ = s >> 1, sk = s & 1;
      const int mB = m0 + 64 * sm_, kB = k0 + 64 * sk;
      const float* srcp = smem + s * 4096 + which * 2048;
      float a = 0.f;
#pragma unroll
      for (int gr = 0; gr < 32; ++gr) a += srcp[gr * 64 + c];
      if (mB < M && kB + c < K) ws[bp.l0part_off + ((int64_t)((2 * mb2 + sm_) * wl.nsplit + split) * 2 + which) * K + kB + c] = a;
    }
  }
  TRACE_STAMP(31);
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

// ONE = a thread folds all partials of its element (full_sum: the same bits as the four cooperating groups of strided_sum),
// 256 elements per workgroup: a quarter of the workgroups for the same work when there are at most 32 slabs per segment
// (config 2: 1600 -> 400 workgroups, no change in time; config 4: 11.9 -> 8.8 us).  Sum-of-squares partials keep their geometry (one per 64 elements).
template <bool ONE>
__global__ __launch_bounds__(256) void grad_reduce_kernel(RedPlan rp, int64_t P, int tail, const float* __restrict__ ws,
                                                          const float* __restrict__ loss_part, int n_loss_part,
                                                          float* __restrict__ grads, float* __restrict__ sumsq_part, int nsq,
                                                          float* __restrict__ sumsq2) {
  __shared__ float sm[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  if (n_loss_part > 0 && blockIdx.x == gridDim.x - 1) {
    // second level of the loss-partial fold (see dnn_wgrad_kernel): loss_part = [n_loss_part][tail] chunk sums
    for (int t0 = 0; t0 < tail; t0 += 64) {
      const int t = t0 + lane;
      sm[grp][lane] = t < tail ? strided_sum(loss_part + t, tail, n_loss_part, grp) : 0.f;
      __syncthreads();
      if (grp == 0 && t < tail) grads[P + t] = ((sm[0][lane] + sm[1][lane]) + sm[2][lane]) + sm[3][lane];
      __syncthreads();
    }
    return;
  }
  if constexpr (ONE) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float g = 0.f;
    if (e < P) {
      int s = 0;
      while (s + 1 < rp.nseg && e >= rp.seg[s + 1].off) ++s;
      const RedSeg sg = rp.seg[s];
      g = full_sum(ws + sg.base + (e - sg.off), sg.stride, sg.nparts);
      grads[e] = g;
    }
    // the product must be ROUNDED before the first cross-lane add: left alone (and with __fmul_rn as well) hipcc turns
    // `g * g + shuffled(g * g)` into an fma in this variant and not in the other - one-ulp different partials, a different clip
    // coefficient, forked trajectories.  The empty asm makes the product opaque.
    float gg = g * g;
    asm volatile("" : "+v"(gg));
    const float sq = wave_sum(gg);
    const int k = (int)blockIdx.x * 4 + grp;
    if (lane == 0 && k < nsq) sumsq_part[k] = sq;
    if (sumsq2 != nullptr) {  // level 2: the block's four partials in order (partials beyond nsq are sums of zeros)
      if (lane == 0) sm[0][grp] = sq;
      __syncthreads();
      if (threadIdx.x == 0) sumsq2[blockIdx.x] = ((sm[0][0] + sm[0][1]) + sm[0][2]) + sm[0][3];
    }
    return;
  }
  const int64_t e = (int64_t)blockIdx.x * 64 + lane;
  float part = 0.f;
  if (e < P) {
    int s = 0;
    while (s + 1 < rp.nseg && e >= rp.seg[s + 1].off) ++s;
    const RedSeg sg = rp.seg[s];
    part = strided_sum(ws + sg.base + (e - sg.off), sg.stride, sg.nparts, grp);
  }  // the step tail grads[P ..] was written by the wgrad launch's last spare workgroup
  sm[grp][lane] = part;
  __syncthreads();
  if (grp == 0) {
    const float g = ((sm[0][lane] + sm[1][lane]) + sm[2][lane]) + sm[3][lane];
    if (e < P) grads[e] = g;
    float gg = e < P ? g * g : 0.f;
    asm volatile("" : "+v"(gg));
    const float sq = wave_sum(gg);
    if (lane == 0) sumsq_part[blockIdx.x] = sq;
  }
}

// Data parallel (ultr_step_args::comm): the slab reduction EXCHANGES its own output - a workgroup folds its 256 elements, publishes
// them into the rank's exchange slot, raises / awaits the slice's flags and adds the ranks' slots in rank order (ultr_comm.h: the
// protocol, slots, flags and epochs of the stand-alone exchange kernel, so ranks may mix the two).  The exchange stops being a
// launch: round 3's data-parallel step paid +8.4 us at world size 1 for comm_allreduce_kernel behind the reduction; here W = 1
// is the plain reduction (same bits) and W > 1 adds one publish / flag / peer-read round trip inside a launch that ran anyway.
// Elements P .. P + tail are the step tail the weight-gradient launch already folded (read from grads, exchanged like the rest).
template <int W>
__global__ __launch_bounds__(256) void grad_reduce_xchg_kernel(RedPlan rp, int64_t P, int tail, const float* __restrict__ ws,
                                                               float* __restrict__ grads, float* __restrict__ sumsq_part, int nsq,
                                                               CommDev c, EarlyReport er, float* __restrict__ sumsq2) {
  __shared__ int sm_fail;
  __shared__ float sm_head[4];
  __shared__ float sm_sq[4];
  const int tid = threadIdx.x, lane = tid & 63, grp = tid >> 6;
  const int64_t n = P + tail;
  const int64_t e = (int64_t)blockIdx.x * 256 + tid;
  if (tid == 0) sm_fail = 0;
  float g = 0.f;
  if (e < P) {
    int s = 0;
    while (s + 1 < rp.nseg && e >= rp.seg[s + 1].off) ++s;
    const RedSeg sg = rp.seg[s];
    g = full_sum(ws + sg.base + (e - sg.off), sg.stride, sg.nparts);
  } else if (e < n) {
    g = grads[e];
  }
  float s = g;
  bool landed = true;
  if constexpr (W > 1) {
    sys_st1(sys_rsrc(c.x_local, c.cap), e < c.cap ? (unsigned)(e * 4) : ULTR_OOB, g);
    landed = comm_flags_and_wait<W>(c, blockIdx.x, &sm_fail);
    float v[W];
#pragma unroll
    for (int p = 0; p < W; ++p) v[p] = sys_ld1(sys_rsrc(c.x[p], c.cap), e < c.cap ? (unsigned)(e * 4) : ULTR_OOB);
    s = 0.f;
#pragma unroll
    for (int p = 0; p < W; ++p) s += v[p];
    if (!landed) s = g;  // timed out: the local value stays; the status word (raised on every rank) freezes the updates
  }
  if (e < n) grads[e] = s;
  {
    const int64_t b0 = (int64_t)blockIdx.x * 256;
    if (er.host != nullptr && P >= b0 && P + 4 <= b0 + 256 && P + 4 <= n) {  // block-uniform: the head of the step tail is in this block
      const int64_t idx = e - P;
      if (idx >= 0 && idx < 4) sm_head[idx] = s;
      __syncthreads();
      bool failed = false;
      if constexpr (W > 1)
        failed = !landed || __hip_atomic_load(c.status[c.rank], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
      if (tid == 0 && !failed) comm_early_report(er, sm_head[0], sm_head[1], sm_head[2], sm_head[3]);
    }
  }
  float gg = e < P ? s * s : 0.f;
  asm volatile("" : "+v"(gg));  // (see grad_reduce_kernel: the product is rounded before the first cross-lane add)
  const float sq = wave_sum(gg);
  const int k = (int)blockIdx.x * 4 + grp;
  if (lane == 0 && k < nsq) sumsq_part[k] = sq;
  if (sumsq2 != nullptr) {  // level 2, as grad_reduce_kernel: the same bits on one GPU and on every rank
    if (lane == 0) sm_sq[grp] = sq;
    __syncthreads();
    if (tid == 0) sumsq2[blockIdx.x] = ((sm_sq[0] + sm_sq[1]) + sm_sq[2]) + sm_sq[3];
  }
}

__global__ __launch_bounds__(64) void grad_sumsq_kernel(int64_t P, const float* __restrict__ grads,
                                                        float* __restrict__ sumsq_part) {
  const int64_t e = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const float g = (e < P) ? grads[e] : 0.f;
  float gg = g * g;
  asm volatile("" : "+v"(gg));  // rounded before the first cross-lane add, as in grad_reduce_kernel / comm_allreduce_kernel: same bits
  const float sq = wave_sum(gg);
  if (threadIdx.x == 0) sumsq_part[blockIdx.x] = sq;
}

// ================================================================================================
// Host side: plans and launches
// ================================================================================================
// Tuning / experiment knobs (README.md): read from the environment ONCE (first use), not on every step - a getenv() walk
// per knob per launch is host time on the critical path of a ~50 us step.  ultr_config_reload() re-reads them (tests and
// the A/B tools flip knobs inside one process).
struct Knobs {
  int fwd_r, bwd_r, wgrad_wgs, fwd_nw, bwd_nw, no_vec, no_fused_fb, fb_max_wg_per_cu, fwd_q4, big_fwd, big_bwd, fb_h3, fwd_h3, bwd_h3, wg_h3, wg_h3_min_rows, wg_h3_wgs, fwd_wide, bwd_wide, fwd_wide_rmax;
  bool loaded;
};
static Knobs g_knobs = {};
static int env_read(const char* name, int dflt) {
  const char* s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}
static void knobs_load() {
  Knobs k;
  k.fwd_r = env_read("ULTR_FWD_R", 0);
  k.bwd_r = env_read("ULTR_BWD_R", 0);
  k.wgrad_wgs = env_read("ULTR_WGRAD_WGS", -1);  // -1: the size-dependent default
  k.fwd_nw = env_read("ULTR_FWD_NW", 8);
  k.bwd_nw = env_read("ULTR_BWD_NW", 8);
  k.no_vec = env_read("ULTR_NO_VEC", 0);
  k.no_fused_fb = env_read("ULTR_NO_FUSED_FB", 0);
  k.fb_max_wg_per_cu = env_read("ULTR_FB_MAX_WG_PER_CU", 1);
  k.fwd_q4 = env_read("ULTR_FWD_Q4", 1);
  // the per-layer big-batch path (ultr_dnn_big.hip): 0 never, 1 by the measured rule (big_*_wanted), 2 whenever legal
  k.big_fwd = env_read("ULTR_BIG_FWD", 1);
  k.big_bwd = env_read("ULTR_BIG_BWD", 1);
  // fused small-batch kernel: products as three fp16 MFMAs on hi / lo operand splits (1, default) or fp32 MFMAs (0)
  k.fb_h3 = env_read("ULTR_FB_H3", 1);
  k.fwd_h3 = env_read("ULTR_FWD_H3", 1);
  k.bwd_h3 = env_read("ULTR_BWD_H3", 1);
  k.wg_h3 = env_read("ULTR_WG_H3", 1);                    // weight gradients on the fp16 matrix cores (split-half operands); 2: any batch size
  k.wg_h3_min_rows = env_read("ULTR_WG_H3_MIN_ROWS", 4096);
  k.wg_h3_wgs = env_read("ULTR_WG_H3_WGS", 0);
  k.bwd_wide = env_read("ULTR_BWD_WIDE", 1);  // dnn_bwdw_kernel, the same for the row-local backward of ultr_train_step
  k.fwd_wide_rmax = env_read("ULTR_FWD_WIDE_RMAX", 64);  // most rows per workgroup of dnn_fwdw_kernel (17 .. 64)
  k.fwd_wide = env_read("ULTR_FWD_WIDE", 1);  // dnn_fwdw_kernel (17 .. 64 rows per workgroup): 0 never, 1 by the measured rule (fwd_wide_plan), 2 whenever legal
  k.loaded = true;
  g_knobs = k;
}
static inline const Knobs& knobs() {
  if (!g_knobs.loaded) knobs_load();
  return g_knobs;
}
void ultr_setrank_knobs_reload();  // ultr_setrank.hip
extern "C" int ultr_config_reload(void) {
  knobs_load();
  ultr_setrank_knobs_reload();
  return 0;
}

extern "C" int ultr_abi_version(void) { return ULTR_ABI_VERSION; }

static bool desc_ok(const ultr_dnn_desc* d) {
  if (!d || d->feature_size <= 0 || d->n_hidden < 0 || d->n_hidden > ULTR_MAX_HIDDEN) return false;
  for (int j = 0; j < d->n_hidden; ++j)
    if (d->hidden[j] <= 0) return false;
  return d->activation >= ULTR_ACT_ELU && d->activation <= ULTR_ACT_SIGMOID;
}

bool ultr_make_dnn_plan(const ultr_dnn_desc* d, int64_t N, DnnPlan* p) {
  if (!desc_ok(d)) return false;
  memset(p, 0, sizeof(*p));
  p->nl = d->n_hidden + 1;
  p->act = d->activation;
  p->no_h3 = (d->flags & ULTR_MODEL_FP32_PRODUCTS) ? 1 : 0;
  p->h3_watch = (!p->no_h3 && (knobs().fb_h3 != 0 || knobs().fwd_h3 != 0 || knobs().bwd_h3 != 0)) ? 1 : 0;
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
  const int NWP = 8;  // waves of the fast kernels
  for (int j = 0; j < p->nl - 1; ++j) {
    const int K = p->K[j], M = p->M[j];
    {
      const int nch = (M + 31) >> 5;
      int ks = 1;
      while (ks * 2 * nch <= NWP) ks *= 2;
      p->fwd_nch[j] = nch;
      p->fwd_ksplit[j] = ks;
      p->fwd_klen[j] = (ks > 1) ? round_up((K + ks - 1) / ks, 32) : K;
    }
    {
      const int nch = (K + 63) >> 6;
      int ms = 1;
      while (ms * 2 * nch <= NWP) ms *= 2;
      p->bwd_nch[j] = nch;
      p->bwd_msplit[j] = ms;
      p->bwd_mlen[j] = round_up((M + ms - 1) / ms, 32);
      // 32-column chunks over the whole contraction as soon as they occupy more than half of the waves: no partial-
      // tile rounds, and for a ragged width (136 = 4 x 32 + 8) far fewer wasted columns than 64-column chunks
      p->bwd_mode[j] = (ms > 1 && 2 * ((K + 31) >> 5) > NWP) ? 1 : (ms == 1 ? 2 : 3);
    }
  }
  int64_t wt = 0;
  for (int j = 0; j < p->nl - 1; ++j) {
    p->wt_off[j] = wt;
    wt += (int64_t)p->K[j] * p->M[j];
    wt = (wt + 3) & ~(int64_t)3;
  }
  p->wt_pv_off = wt;
  int pv = 0;
  for (int j = 0; j < p->nl; ++j) {
    p->pv_off[j] = pv;
    pv += 2 * p->K[j] + p->M[j];
  }
  p->pv_wlast = pv;
  pv += p->K[p->nl - 1];
  p->pv_total = (pv + 3) & ~3;
  p->wt_total = wt + p->pv_total;
  {  // fragment-major copies (DnnPlan::wsf_off / wsb_off)
    bool ok = p->nl >= 2;
    for (int j = 0; j < p->nl - 1; ++j) ok = ok && (p->M[j] % 32 == 0);
    p->sw_ok = ok ? 1 : 0;
    int64_t o = (p->wt_total + 255) & ~(int64_t)255;  // 1 KiB aligned
    p->ws_begin = o;
    if (ok) {
      for (int j = 0; j < p->nl - 1; ++j) {
        const int64_t n = (int64_t)round_up(p->K[j], 32) * round_up(p->M[j], 32);
        p->wsf_off[j] = o;
        o += n;
        if (j >= 1) {
          p->wsb_off[j] = o;
          o += n;
        }
      }
      // split-half copies (DnnPlan::whf_off / whb_off), per layer: same element counts, two halves per float
      bool h3 = true;
      for (int j = 0; j < p->nl - 1; ++j) {
        const int64_t n = (int64_t)round_up(p->K[j], 32) * round_up(p->M[j], 32);
        // 1: eight or more 32-column chunks (the 8-wave kernels take the layer on the split-half stream); 2: fewer - only the wide-tile
        // forward (dnn_fwdw_kernel: 16 waves, chunks x slices of the contraction) reads that copy
        p->h3f[j] = (p->M[j] >= 256) ? 1 : 2;
        p->h3b[j] = (j >= 1 && p->K[j] % 32 == 0) ? (p->K[j] >= 256 ? 1 : 2) : 0;  // (2: only dnn_bwdw_kernel reads that copy)
        h3 = h3 && p->h3f[j] == 1 && (j == 0 || p->h3b[j] == 1);
        if (p->h3f[j]) {
          p->whf_off[j] = o;
          o += n;
        }
        if (p->h3b[j]) {
          p->whb_off[j] = o;
          o += n;
        }
      }
      p->h3_ok = h3 ? 1 : 0;
      p->fb_h3 = (h3 && knobs().fb_h3 && !p->no_h3) ? 1 : 0;
      p->bwd_h3 = 0;
      if (knobs().bwd_h3 && !p->no_h3)
        for (int j = 1; j < p->nl - 1; ++j)
          if (p->h3b[j] == 1) p->bwd_h3 = 1;
      p->fwd_h3 = 0;
      if (knobs().fwd_h3 && !p->no_h3)
        for (int j = 0; j < p->nl - 1; ++j)
          if (p->h3f[j] == 1 && round_up(p->K[j], 32) <= 768) p->fwd_h3 = 1;
    }
    // the range word of the hidden weights (every model with a hidden layer: the per-layer big-batch path builds split-half planes of
    // ANY hidden layer, ultr_dnn_big.hip); zeroed by ultr_dnn_build_wt
    if (p->nl >= 2) {
      p->h3_flag_off = o;
      o += 4;
      p->wt_total = o;
    }
  }
  int64_t sv = 0;
  for (int j = 1; j < p->nl; ++j) {
    p->sv_x[j] = sv;
    sv += N * p->K[j];
    sv = (sv + 3) & ~(int64_t)3;
  }
  p->sv_x[0] = sv;  // normalised layer-0 input for the weight gradients (fused kernel, BwdPlan::wg_prenorm)
  sv += N * p->K[0];
  sv = (sv + 3) & ~(int64_t)3;
  for (int j = 0; j < p->nl; ++j) {
    p->sv_mean[j] = sv; sv += N;
    p->sv_rstd[j] = sv; sv += N;
  }
  p->sv_total = sv;
  {  // update-kernel work map
    int t = 0;
    for (int j = 0; j < p->nl - 1; ++j) {
      p->upd_tile_begin[j] = t;
      p->upd_ntk[j] = (p->K[j] + 15) / 16;
      t += ((p->M[j] + 15) / 16) * p->upd_ntk[j];
    }
    p->upd_tile_begin[p->nl - 1] = t;
    int n = 0, v = 0;
    auto seg = [&](int64_t o, int len, int pvpos) {
      p->vs_off[n] = o; p->vs_len[n] = len; p->vs_pv[n] = pvpos; p->vs_begin[n] = v;
      v += len; ++n;
    };
    for (int j = 0; j < p->nl; ++j) {
      seg(p->off_lnw[j], 2 * p->K[j], p->pv_off[j]);  // gamma | beta are adjacent in both layouts
      if (j < p->nl - 1) {
        seg(p->off_b[j], p->M[j], p->pv_off[j] + 2 * p->K[j]);
      } else {
        seg(p->off_w[j], p->K[j], p->pv_wlast);
        seg(p->off_b[j], 1, p->pv_off[j] + 2 * p->K[j]);
      }
    }
    p->n_vs = n;
    p->vs_begin[n] = v;
  }
  for (int j = 0; j < p->nl; ++j) {
    DnnPlan::FwdLayer& l = p->fl[j];
    l.K = p->K[j]; l.M = p->M[j];
    l.ksplit = p->fwd_ksplit[j]; l.klen = p->fwd_klen[j]; l.nch = (p->M[j] + 31) >> 5; l.pad = 0;
    l.wt_off = p->wt_off[j]; l.sv_mean = p->sv_mean[j]; l.sv_rstd = p->sv_rstd[j];
    l.sv_x_next = (j + 1 < p->nl) ? p->sv_x[j + 1] : 0;
    l.off_w = p->off_w[j];
  }
  return true;
}

static size_t fwd_pv_floats(const DnnPlan& p) { return (size_t)p.pv_total; }
static size_t fwd_lds_bytes(const DnnPlan& p, int R) {
  return ((size_t)2 * R * fwd_ld_of(p.maxdim, p.fwd_h3) + fwd_pv_floats(p)) * sizeof(float);
}
static int fwd_rows_per_wg(const DnnPlan& p, int64_t N) {
  int r = knobs().fwd_r;
  if (r == 16 || r == 32) return r;
  // 16-row tiles everywhere: 32-row tiles halve the W stream per row, but their LDS footprint leaves one workgroup per CU and
  // the grid quantises badly (measured at cfg3 / B=1024: 114 -> 100 us and 57 -> 43 us with 16 rows); ULTR_FWD_R=32 forces them
  (void)N;
  return 16;
}
static size_t bwd_lds_bytes(const DnnPlan& p, int R) {
  return ((size_t)R * (2 * bwd_ldu(p.maxdim) + bwd_ldz(p.maxdim)) + 2 * (size_t)bwd_ldu(p.maxdim) + 5 * (size_t)R) * sizeof(float) + (size_t)R * sizeof(int64_t);
}
static int bwd_rows_per_wg(const DnnPlan& p, int64_t N) {
  int r = knobs().bwd_r;
  if (r == 16 || r == 32) return r;
  return ((N + 15) / 16 > 512 && bwd_lds_bytes(p, 32) <= 160 * 1024) ? 32 : 16;
}

bool ultr_make_bwd_plan(const DnnPlan& p, int64_t N, BwdPlan* bp, int wg_mode) {
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
  bp->n_red_blocks = (int)ultr_red_blocks(p.P, (int)tail_max);
  bp->sumsq_off = off; off += bp->n_red_blocks; off = (off + 3) & ~(int64_t)3;
  off += ultr_sumsq2_len(p.P);  // level-2 partials at ultr_sumsq2_off(P) (= here: sumsq_off is 0)
  off = (off + 3) & ~(int64_t)3;
  // sized for the finest row blocking any kernel uses (the fused forward+backward kernel owns >= 9 live rows per block)
  const int64_t nrb_alloc = (N + 8) / 9 + 1 > bp->nrb ? (N + 8) / 9 + 1 : bp->nrb;
  bp->vslab_off = off; off += nrb_alloc * bp->vlen; off = (off + 3) & ~(int64_t)3;
  bp->vred_off = off; off += bp->vlen; off = (off + 3) & ~(int64_t)3;
  bp->vred_blocks = (bp->vlen + 63) / 64;
  for (int j = 0; j < p.nl - 1; ++j) {
    bp->dz_off[j] = off; off += N * p.M[j]; off = (off + 3) & ~(int64_t)3;
  }
  {
    int kmax = 0;
    for (int j = 1; j < p.nl - 1; ++j) kmax = p.K[j] > kmax ? p.K[j] : kmax;
    bp->du_off = off; off += N * kmax; off = (off + 3) & ~(int64_t)3;
  }
  // wgrad geometry
  int tiles = 0, tiles2 = 0;
  bool h3w = wg_mode != 0 && knobs().wg_h3 != 0 && p.nl >= 2 && (knobs().wg_h3 >= 2 || N >= knobs().wg_h3_min_rows);
  for (int j = 0; j < p.nl - 1; ++j) {
    tiles += ((p.M[j] + 63) / 64) * ((p.K[j] + 63) / 64);
    tiles2 += ((p.M[j] + 127) / 128) * ((p.K[j] + 127) / 128);
    if (p.M[j] % 4 != 0 || p.K[j] % 4 != 0 || p.off_w[j] % 4 != 0) h3w = false;
  }
  bp->wg_h3 = h3w ? 1 : 0;
  // workgroups over all hidden Linears: about one per CU for a small batch (every workgroup is a chain of latencies and a
  // second one on the CU only slows both), about two per CU otherwise - measured (tools/sweep_wgrad.sh, bench_configs.py):
  // N = 2560 rows: 224 -> 12.3 us, 392 -> 12.7, 448 -> 13.0;  N = 10240: 224 -> 36, 392 -> 33, 448 -> 30 us
  // Split-half launch (wg_h3): 128 x 128 blocks, a quarter of the tiles - every row split is one more slab of P floats to write and
  // to fold, so the target stays near one workgroup per CU
  const int target = h3w ? (knobs().wg_h3_wgs > 0 ? knobs().wg_h3_wgs : 256) : knobs().wgrad_wgs > 0 ? knobs().wgrad_wgs : (N < 4096 ? 224 : 448);
  const int64_t rps_cap = h3w ? 2048 : 4096;  // the per-row tables of a split live in LDS
  int blk = 0;
  for (int j = 0; j < p.nl - 1; ++j) {
    WgradLayer& w = bp->wl[j];
    w.M = p.M[j]; w.K = p.K[j];
    w.nmb = (w.M + 63) / 64; w.nkb = (w.K + 63) / 64;
    w.nmb2 = (w.M + 127) / 128; w.nkb2 = (w.K + 127) / 128;
    const int tl = h3w ? tiles2 : tiles;
    int nsplit = tl > 0 ? (h3w ? target / tl : (target + tl - 1) / tl) : 1;
    if (nsplit < 1) nsplit = 1;
    const int by_cap = (int)((N + rps_cap - 1) / rps_cap);  // the doc-id table of a split lives in LDS: at most 4096 rows
    if (nsplit < by_cap) nsplit = by_cap;
    // XCD-friendly split count.  Blocks are numbered split-fastest and consecutive block ids go round-robin to the 8 XCDs, so
    // with nsplit a divisor or a multiple of 8 each XCD works on ONE row chunk of a layer at a time and the blocks that
    // re-read the same dz / x rows (every 64 x 64 tile of that chunk) meet in one L2.  Misaligned counts fetch every operand
    // once per tile from HBM: config 4 (rocprofv3) 311 MB per launch at nsplit = 4, 115 us - nsplit 3 / 5: 154 / 158 us;
    // config 3 at nsplit = 7: 342 MB for 74 MB of operands, HBM-bound at 5.6 TB/s.
    if (knobs().wgrad_wgs <= 0 && !h3w) nsplit = nsplit <= 1 ? 1 : nsplit <= 2 ? 2 : nsplit <= 5 ? 4 : nsplit <= 11 ? 8 : (nsplit + 4) / 8 * 8;
    if (nsplit < by_cap) nsplit = h3w ? by_cap : (by_cap + 7) / 8 * 8;
    int64_t rps = (N + nsplit - 1) / nsplit;
    rps = (rps + 31) / 32 * 32;  // 8 rows per wave-trip
    if (rps < 64) rps = 64;
    if (rps > rps_cap) rps = rps_cap;
    w.rows_per_split = (int)rps;
    w.nsplit = (int)((N + rps - 1) / rps);
    w.blk_begin = blk;  // (wg_h3: the first TILE of the layer - see BwdPlan::wg_chunk)
    blk += h3w ? w.nmb2 * w.nkb2 : w.nmb * w.nkb * w.nsplit;
    w.vec = (w.M % 4 == 0 && w.K % 4 == 0) ? 1 : 0;
    w.dz_off = bp->dz_off[j];
    w.slab_off = off; off += (int64_t)w.nsplit * ((int64_t)w.M * w.K + w.M); off = (off + 3) & ~(int64_t)3;
  }
  bp->wgrad_blocks = blk;
  bp->wg_tiles2 = bp->wg_live = bp->wg_chunk = 0;
  if (h3w) {
    bp->wg_tiles2 = tiles2;
    bp->wg_live = tiles2 * bp->wl[0].nsplit;  // (every layer got the same split count: one formula, one N)
    bp->wg_chunk = (bp->wg_live + 7) / 8;
    bp->wgrad_blocks = 8 * bp->wg_chunk;
  }
  bp->lfold_off = off; off += 64 * tail_max;  // second level of the loss-partial fold (more than 1024 partials)
  bp->l0g = 0;
  bp->l0part_off = off;
  if (p.nl >= 2) off += ((int64_t)bp->wl[0].nmb * bp->wl[0].nsplit * 2 * p.K[0] + 3) & ~(int64_t)3;
  {
    const int64_t h = ultr_dgp_layer(p, p.nl - 1);
    off = (off + 7) & ~(int64_t)7;
    bp->dgp_off = off; off += (h + 1) / 2; off = (off + 3) & ~(int64_t)3;
  }
  bp->total = off;
  return true;
}


void ultr_make_red_plan(const DnnPlan& p, const BwdPlan& bp, RedPlan* rp) {
  int s = 0;
  for (int j = 0; j < p.nl; ++j) {
    const bool last = (j == p.nl - 1);
    if (j == 0 && bp.l0g) {
      const int np0 = bp.wl[0].nmb * bp.wl[0].nsplit;
      rp->seg[s++] = RedSeg{p.off_lnw[0], bp.l0part_off, 2 * (int64_t)p.K[0], p.K[0], np0};
      rp->seg[s++] = RedSeg{p.off_lnb[0], bp.l0part_off + p.K[0], 2 * (int64_t)p.K[0], p.K[0], np0};
    } else {
      rp->seg[s++] = RedSeg{p.off_lnw[j], bp.vred_off + bp.voff_g[j], 0, p.K[j], 1};
      rp->seg[s++] = RedSeg{p.off_lnb[j], bp.vred_off + bp.voff_b[j], 0, p.K[j], 1};
    }
    if (last) {
      rp->seg[s++] = RedSeg{p.off_w[j], bp.vred_off + bp.voff_wk, 0, p.K[j], 1};
      rp->seg[s++] = RedSeg{p.off_b[j], bp.vred_off + bp.voff_bk, 0, 1, 1};
    } else {
      const WgradLayer& w = bp.wl[j];
      const int64_t stride = (int64_t)w.M * w.K + w.M;
      rp->seg[s++] = RedSeg{p.off_w[j], w.slab_off, stride, w.M * w.K, w.nsplit};
      rp->seg[s++] = RedSeg{p.off_b[j], w.slab_off + (int64_t)w.M * w.K, stride, w.M, w.nsplit};
    }
  }
  rp->nseg = s;
}

// true when every hot-loop load may take the aligned branch-free float4 path
static bool all_vec(const DnnPlan& p, int vm, int64_t N, int64_t n_docs) {
  // buffer-resource offsets are 32-bit: every described tensor must stay below 2 GiB
  const int64_t lim = (int64_t)1 << 31;
  if (n_docs * p.K[0] * 4 >= lim || N * p.maxdim * 4 >= lim) return false;
  for (int j = 0; j < p.nl; ++j)
    if (!((vm >> j) & 1)) return false;
  for (int j = 0; j < p.nl - 1; ++j)
    if (p.M[j] % 4 != 0) return false;
  return (vm >> 31) & 1;
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
  return (ultr_fwp_off(p) + (ultr_fwp_halves(p) + 1) / 2 + 4) * (int64_t)sizeof(float);
}
extern "C" int64_t ultr_dnn_bwd_workspace_bytes(const ultr_dnn_desc* d, int64_t n_rows) {
  DnnPlan p;
  BwdPlan bp;
  if (n_rows < 0 || !ultr_make_dnn_plan(d, n_rows, &p)) return 0;
  // worst case over the env-tunable geometry: size for both row-block choices
  ultr_make_bwd_plan(p, n_rows, &bp);
  int64_t t = bp.total;
  if (bp.wg_h3) {  // the launcher may fall back to the register kernel's geometry (unaligned pointers)
    BwdPlan b0;
    ultr_make_bwd_plan(p, n_rows, &b0, 0);
    t = b0.total > t ? b0.total : t;
  }
  const int64_t extra = ((n_rows + 15) / 16) * (int64_t)bp.vlen;  // if rblk were 16
  return (t + extra + 1024) * (int64_t)sizeof(float);
}
extern "C" int64_t ultr_step_tail_floats(int32_t list_size) { return ultr_tail_len(list_size); }

extern "C" int64_t ultr_dnn_wt_floats(const ultr_dnn_desc* d) {
  DnnPlan p;
  if (!ultr_make_dnn_plan(d, 0, &p)) return 0;
  return p.wt_total > 0 ? p.wt_total : 4;
}

// WT_j[k, m] = W_j[m, k] for every hidden Linear (coalesced reads, strided writes; ~100k elements) + the PV image
__global__ __launch_bounds__(256) void wt_build_kernel(DnnPlan p, const float* __restrict__ params, float* __restrict__ wt) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= p.P) {
    const int64_t idx = p.pv_wlast + p.K[p.nl - 1] + (e - p.P);  // zero the image's padding (< 4 floats)
    if (idx < p.pv_total) wt[p.wt_pv_off + idx] = 0.f;
    return;
  }
  const int pv = ultr_pv_index(p, e);
  if (pv >= 0) {
    wt[p.wt_pv_off + pv] = params[e];
    return;
  }
  for (int j = 0; j < p.nl - 1; ++j) {
    const int64_t r = e - p.off_w[j];
    if (r >= 0 && r < (int64_t)p.M[j] * p.K[j]) {
      const int m = (int)(r / p.K[j]), k = (int)(r % p.K[j]);
      const float v = params[e];
      wt[p.wt_off[j] + (int64_t)k * p.M[j] + m] = v;
      if (p.sw_ok) {
        wt[p.wsf_off[j] + ultr_sw_index(m, k, (p.K[j] + 31) >> 5)] = v;
        if (j >= 1) wt[p.wsb_off[j] + ultr_sw_index(k, m, (p.M[j] + 31) >> 5)] = v;
      }
      const float sv = v * ULTR_H3_WSCALE;
      if (!(fabsf(sv) < ULTR_H3_WNEAR))  // every hidden weight is watched: the per-layer path builds planes of any layer (NaN counts as out of range)
        flag_or(reinterpret_cast<uint32_t*>(wt + p.h3_flag_off), !(fabsf(sv) < ULTR_H3_WMAX) ? (ULTR_H3_FLAG_OVER | ULTR_H3_FLAG_NEAR) : ULTR_H3_FLAG_NEAR);
      if (p.h3f[j] || p.h3b[j]) {
        const _Float16 hi = (_Float16)sv, lo = (_Float16)(sv - (float)hi);
        if (p.h3f[j]) {
          _Float16* hf = reinterpret_cast<_Float16*>(wt + p.whf_off[j]);
          hf[ultr_h3_index(m, k, (p.K[j] + 31) >> 5, 0)] = hi;
          hf[ultr_h3_index(m, k, (p.K[j] + 31) >> 5, 1)] = lo;
        }
        if (p.h3b[j]) {
          _Float16* hb = reinterpret_cast<_Float16*>(wt + p.whb_off[j]);
          hb[ultr_h3_index(k, m, (p.M[j] + 31) >> 5, 0)] = hi;
          hb[ultr_h3_index(k, m, (p.M[j] + 31) >> 5, 1)] = lo;
        }
      }
      return;
    }
  }
}

extern "C" int ultr_dnn_build_wt(const ultr_dnn_desc* d, const float* params, float* wt, void* stream) {
  DnnPlan p;
  if (!params || !wt || !ultr_make_dnn_plan(d, 0, &p)) return ULTR_E_BADARG;
  const int64_t n = p.P + 4;
  if (p.wt_total > p.ws_begin) {  // the fragment-major copies are padded to whole 32 x 32 blocks: zeros there (and the range word)
    const hipError_t e = hipMemsetAsync(wt + p.ws_begin, 0, (size_t)(p.wt_total - p.ws_begin) * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(wt_build_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, params, wt);
  return (int)hipGetLastError();
}

extern "C" int ultr_dnn_wt_range(const ultr_dnn_desc* d, const float* wt, void* stream) {
  DnnPlan p;
  if (!wt || !ultr_make_dnn_plan(d, 0, &p)) return ULTR_E_BADARG;
  if (p.h3_flag_off <= 0) return 0;  // no hidden layer: nothing is ever split
  uint32_t f = 0;
  hipError_t e = hipMemcpyAsync(&f, wt + p.h3_flag_off, sizeof(f), hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) return -(int)e - 100;
  return (f & ULTR_H3_FLAG_OVER) ? 2 : ((f & ULTR_H3_FLAG_NEAR) ? 1 : 0);
}

// Which shapes take the per-layer path (ultr_dnn_big.hip) instead of the row-tile kernels below; mode = the knob (0 never,
// 1 the measured rule, 2 whenever legal).  Measured ("big-batch path"; us, row-tile -> per-layer):
//   forward   slower in general: config 3 85 -> 120, 81 920 rows x [256,256] 258 -> 297 (the tiled GEMM core runs at 63-85
//             TFLOP/s, the row-tile forward at 61-70 with no activation round trips) - taken when the row-tile kernel's LDS
//             footprint does not fit, and in the case big_fwd_wanted describes (config 4: 220 -> 210);
//   backward  wins when dnn_bwd2_kernel does not apply (a layer wider than 512: config 4 128 -> 85) and for big batches
//             (81 920 x [256,256] 202 -> 182, 163 840 x [512,256,128] 980 -> 830); a tie at config 3 (10 240 rows: 74 / 74).
static int dnn_device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
  }
  return cus;
}
// The one case where the per-layer forward wins: a wide gathered input (K_0 > 512: the statistics pass then writes xhat_0
// contiguously and the first GEMM reads plain rows, ultr_dnn_big.hip) AND both staircases line up - the row tiles would
// run >= 3 full rounds of resident workgroups plus a last round less than a quarter full (their time is a staircase in the
// tile count: config 4 = 800 tiles of a one-per-CU workgroup: 172 us at 750 tiles, 221 us at 800), while the first GEMM's
// 64 x 128 chunks still fit about one round of ITS resident workgroups (3 per CU).  Measured, row tiles -> per-layer (us),
// F700 [512,256,128]: 257 tiles 111 -> 133 (fixed cost of seven launches), 750 tiles 176 -> 193, 800 tiles 221 -> 209,
// 938 tiles 224 -> 215, 1063 tiles 274 -> 281 (the GEMM's own second round).
static bool big_fwd_wanted(const DnnPlan& p, int64_t N, size_t row_tile_lds) {
  if (knobs().big_fwd != 1) return knobs().big_fwd >= 2;
  if (p.K[0] <= 512 || p.nl < 2) return false;
  // round 3: with its first layer on the split-half copies the row-tile forward wins that case too (config 4, 800 tiles:
  // 218 us per-layer, 225 row tiles in fp32, 151 row tiles with the split-half products)
  if (p.fwd_h3 && p.h3f[0] == 1 && round_up(p.K[0], 32) <= 768) return false;
  const int cus = dnn_device_cus();
  const int per_cu = row_tile_lds > 80 * 1024 ? 1 : (row_tile_lds > 53 * 1024 ? 2 : 3);
  const int64_t slots = (int64_t)per_cu * cus, tiles = (N + 15) / 16, full = tiles / slots, rem = tiles % slots;
  const int64_t chunks0 = ((N + 63) / 64) * ((p.M[0] + 127) / 128), gslots = 3 * (int64_t)cus;
  return full >= 3 && rem != 0 && rem * 4 <= slots && chunks0 <= gslots + gslots / 8;
}
static bool big_bwd_wanted(const DnnPlan& p, int64_t N) {
  const int mode = knobs().big_bwd;
  if (mode != 1) return mode >= 2;
  const bool v2 = knobs().bwd_nw == 8 && p.maxdim <= 512 &&
                  bwd2_lds_floats(p, bwd_rows_per_wg(p, N), 8) * sizeof(float) <= 160 * 1024;
  return N >= (v2 ? 16384 : 4096);
}

// dnn_fwdw_kernel: which shapes take it, and with how many rows per workgroup.  Legal: every hidden layer has its split-half
// forward copy (widths multiples of 32) and no LayerNorm is wider than 768 (three float4 per lane).  Rows: as many as the LDS holds
// (two buffers sized per layer parity + the parameter image), at most 64; then the smallest R that still covers the batch in the
// same number of rounds of one workgroup per CU.  Taken when that is more than the 16 rows of dnn_fwd_kernel.
static bool fwd_wide_plan(const DnnPlan& p, int64_t N, WidePlan* wp, size_t* lds_bytes) {
  if (knobs().fwd_wide == 0 || knobs().fwd_h3 == 0 || p.no_h3 || p.nl < 2 || !p.sw_ok || p.pv_total > 2 * 1024 * 4) return false;
  if (knobs().fwd_r != 0 || knobs().fwd_nw != 8) return false;  // an explicit tile geometry of the row-tile kernel was asked for
  int w[2] = {0, 0};
  for (int j = 0; j < p.nl; ++j) {
    const int k16 = round_up(p.K[j], 32);
    if (k16 > 768 || (j < p.nl - 1 && p.h3f[j] == 0)) return false;
    if (k16 + 8 > w[j & 1]) w[j & 1] = k16 + 8;
  }
  const int64_t fixed = ((int64_t)p.pv_total + 64) * 4, per_row = (int64_t)(w[0] + w[1]) * 4;
  int64_t rmax = (160 * 1024 - fixed) / per_row - 1;  // the buffers hold R + 1 rows (dnn_fwdw_kernel: the rows of the last MFMA tile beyond R land in row R)
  const int cap = knobs().fwd_wide_rmax < 17 ? 17 : (knobs().fwd_wide_rmax > 64 ? 64 : knobs().fwd_wide_rmax);  // four MFMA row tiles at most
  if (rmax > cap) rmax = cap;
  if (rmax < 17) return false;
  const int64_t cus = dnn_device_cus();
  const int64_t rounds = (N + cus * rmax - 1) / (cus * rmax);
  const int64_t R = (N + cus * rounds - 1) / (cus * rounds);
  if (R <= 16) return false;
  // Several rounds of small workgroups behind a small weight set are what the 16-row kernel (two or three workgroups per CU, their
  // phases overlapping) does well: validation of config 2's model at 100 candidates (25 600 rows, 0.4 MB of weights), forward us:
  // 16-row tiles 63.7; wide 59.2 at 2 rounds x 50 rows, 67.3 at 3 x 34, 71.2 at 4 x 25.  Against that: config 3 (1 round x 40 rows,
  // 0.94 MB) 60 -> 41, config 4 (2 x 25 rows, 2.1 MB) 151 -> 95.  Wide tiles when one round covers the batch, when the tiles are
  // nearly full 48- / 64-row ones, or when the weights are what the 16-row tiles spend their time streaming.
  int64_t wbytes = 0;
  for (int j = 0; j < p.nl - 1; ++j) wbytes += (int64_t)round_up(p.K[j], 32) * p.M[j] * 4;
  if (knobs().fwd_wide == 1 && rounds > 1 && R < 44 && wbytes < 768 * 1024) return false;
  memset(wp, 0, sizeof(*wp));
  wp->R = (int)R;
  wp->buf[0] = 0;
  wp->buf[1] = (int)(R + 1) * w[0];
  wp->pv = (int)(R + 1) * (w[0] + w[1]);
  for (int j = 0; j < p.nl - 1; ++j) {
    const int nks = round_up(p.K[j], 32) / 32, nch = p.M[j] / 32;
    int ks = nch >= 16 ? 1 : 16 / nch;
    if (ks > nks) ks = nks;
    const int len = (nks + ks - 1) / ks;
    wp->kslen[j] = len;
    wp->ksplit[j] = (nks + len - 1) / len;
  }
  *lds_bytes = (size_t)(wp->pv + p.pv_total + 64) * sizeof(float);
  return true;
}

// dnn_bwdw_kernel: legal when the layer-0 shortcut is on, every LayerNorm of the layers >= 1 is at most 512 wide and every dgrad
// product has its split-half copy; rows per workgroup as in fwd_wide_plan (LDS: the dz planes, the du tile that also holds the
// column partials of sixteen waves, 128 floats of per-row scalars)
static bool bwd_wide_plan(const DnnPlan& p, int64_t N, WideBwd* wb, size_t* lds_bytes) {
  if (knobs().bwd_wide == 0 || knobs().bwd_h3 == 0 || p.no_h3 || p.nl < 2 || !p.sw_ok) return false;
  if (knobs().bwd_r != 0 || knobs().bwd_nw != 8 || knobs().big_bwd >= 2) return false;  // another kernel was asked for explicitly
  if (p.sv_total * 4 >= ((int64_t)1 << 31)) return false;
  const int top = p.nl - 1;
  int wdz = 0, wdu = 0, cpmax = 16 * 3 * p.K[top];
  for (int j = 1; j <= top; ++j) {
    if (p.K[j] > 512 || p.K[j] % 32 != 0) return false;
    if (j < top) {
      if (p.h3b[j] == 0) return false;
      if (p.K[j] + 8 > wdu) wdu = p.K[j] + 8;
      if (p.M[j] + 8 > wdz) wdz = p.M[j] + 8;
      if (16 * 2 * p.K[j] > cpmax) cpmax = 16 * 2 * p.K[j];
    }
  }
  auto floats = [&](int64_t R) {
    const int64_t du = (R + 1) * wdu > cpmax ? (R + 1) * wdu : cpmax;
    return (R + 1) * wdz + du + 128;
  };
  int64_t rmax = 64;  // four row tiles at most
  while (rmax > 16 && floats(rmax) * 4 > 160 * 1024) --rmax;
  if (rmax < 17) return false;
  const int64_t cus = dnn_device_cus();
  const int64_t rounds = (N + cus * rmax - 1) / (cus * rmax);
  const int64_t R = (N + cus * rounds - 1) / (cus * rounds);
  if (R <= 16) return false;
  memset(wb, 0, sizeof(*wb));
  wb->R = (int)R;
  wb->dz = 0;
  wb->du = (int)((R + 1) * wdz);
  wb->ds = (int)(floats(R) - 128);
  for (int j = 1; j < top; ++j) {
    const int nks = p.M[j] / 32, nch = p.K[j] / 32;
    int ks = nch >= 16 ? 1 : 16 / nch;
    if (ks > nks) ks = nks;
    const int len = (nks + ks - 1) / ks;
    wb->kslen[j] = len;
    wb->ksplit[j] = (nks + len - 1) / len;
  }
  *lds_bytes = (size_t)floats(R) * sizeof(float);
  return true;
}

template <typename KernelT>
static hipError_t set_lds(KernelT k, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

extern "C" int ultr_dnn_forward(const ultr_dnn_desc* d, const float* params, const float* wt, const float* features,
                                int64_t n_docs, const int32_t* docids, int32_t batch, int32_t list_size, float* scores,
                                void* saved, void* stream) {
  if (!params || !docids || !scores || batch <= 0 || list_size <= 0 || n_docs < 0 || (n_docs > 0 && !features))
    return ULTR_E_BADARG;
  const int64_t N = (int64_t)batch * list_size;
  DnnPlan p;
  if (!ultr_make_dnn_plan(d, N, &p)) return ULTR_E_BADARG;
  const int R = fwd_rows_per_wg(p, N);
  const size_t lds = fwd_lds_bytes(p, R);
  const int nw = knobs().fwd_nw;
  const int vm = vecmask_for(p, params, features);
  const dim3 grid((unsigned)((N + R - 1) / R));
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipSuccess;
  UltrProfScope prof(ULTR_K_FWD, st);
  // the fast path needs the k-major weight copy (ultr_dnn_build_wt / kept current by ultr_apply_update)
  const bool av = all_vec(p, vm, N, n_docs) && knobs().no_vec == 0 && (wt != nullptr || p.nl == 1) &&
                  ((uintptr_t)wt & 15) == 0;
  // training forward of a big batch: one pass per layer (needs `saved` for the activations between the passes)
  if (saved != nullptr && wt != nullptr && av && ultr_dnn_big_ok(p, N, n_docs) &&
      knobs().big_fwd != 0 && (big_fwd_wanted(p, N, lds) || lds > 160 * 1024))
    return ultr_dnn_big_forward(p, params, wt, features, n_docs, docids, (int)batch, (int)list_size, scores, (float*)saved, st,
                                prof.on ? prof.a : nullptr, prof.on ? prof.b : nullptr, knobs().fwd_h3 != 0 && !p.no_h3 && wt != nullptr);
  {
    WidePlan wp;
    size_t wlds = 0;
    if (av && wt != nullptr && fwd_wide_plan(p, N, &wp, &wlds)) {
      const int rt = (wp.R + 15) / 16;
      const dim3 wgrid((unsigned)((N + wp.R - 1) / wp.R));
#define LAUNCH_FWDW(RTT)                                                                                                        \
  do {                                                                                                                          \
    e = set_lds(dnn_fwdw_kernel<RTT>, wlds);                                                                                    \
    if (e != hipSuccess) return (int)e;                                                                                         \
    ULTR_LAUNCH(prof, (dnn_fwdw_kernel<RTT>), wgrid, dim3(1024), wlds, st, p, wp, features, n_docs, docids, (int)batch,         \
                (int)list_size, scores, (float*)saved, wt);                                                                     \
  } while (0)
      if (rt == 2) LAUNCH_FWDW(2);
      else if (rt == 3) LAUNCH_FWDW(3);
      else LAUNCH_FWDW(4);
#undef LAUNCH_FWDW
      return (int)hipGetLastError();
    }
  }
  if (lds > 160 * 1024) return ULTR_E_UNSUPPORTED;
#define LAUNCH_FWD(RR, NWW, VV)                                                                                     \
  do {                                                                                                              \
    e = set_lds(dnn_fwd_kernel<RR, NWW, VV>, lds);                                                                  \
    if (e != hipSuccess) return (int)e;                                                                             \
    ULTR_LAUNCH(prof, (dnn_fwd_kernel<RR, NWW, VV>), grid, dim3(NWW * 64), lds, st, p, params, features, n_docs,    \
                       docids, (int)batch, (int)list_size, scores, (float*)saved, wt, vm);                         \
  } while (0)
#define LAUNCH_FWD2(RR, NWW) \
  do {                       \
    if (av) LAUNCH_FWD(RR, NWW, true); \
    else LAUNCH_FWD(RR, NWW, false);   \
  } while (0)
  bool q4 = false;  // one workgroup per CU anyway and a layer that takes 64-column chunks: the 16-byte-load build
  if (av && R == 16 && nw == 8 && knobs().fwd_q4 && lds > 80 * 1024)
    for (int j = 0; j < p.nl - 1; ++j) q4 = q4 || (p.M[j] % 64 == 0 && p.M[j] >= 512);
  if (q4) {
    e = set_lds(dnn_fwd_kernel<16, 8, true, true>, lds);
    if (e != hipSuccess) return (int)e;
    ULTR_LAUNCH(prof, (dnn_fwd_kernel<16, 8, true, true>), grid, dim3(512), lds, st, p, params, features, n_docs, docids,
                (int)batch, (int)list_size, scores, (float*)saved, wt, vm);
  } else if (R == 16 && nw == 4) LAUNCH_FWD2(16, 4);
  else if (R == 16 && nw == 16) LAUNCH_FWD2(16, 16);
  else if (R == 16) LAUNCH_FWD2(16, 8);
  else if (nw == 4) LAUNCH_FWD2(32, 4);
  else LAUNCH_FWD2(32, 8);
#undef LAUNCH_FWD2
#undef LAUNCH_FWD
  return (int)hipGetLastError();
}

extern "C" int32_t ultr_dnn_forward_tile_rows(const ultr_dnn_desc* d, int64_t n_rows, int32_t training) {
  DnnPlan p;
  if (n_rows <= 0 || !ultr_make_dnn_plan(d, n_rows, &p)) return -1;
  const int R = fwd_rows_per_wg(p, n_rows);
  const size_t lds = fwd_lds_bytes(p, R);
  if (training && ultr_dnn_big_ok(p, n_rows, n_rows) && knobs().big_fwd != 0 && (big_fwd_wanted(p, n_rows, lds) || lds > 160 * 1024)) return 0;
  WidePlan wp;
  size_t wlds = 0;
  if (knobs().no_vec == 0 && fwd_wide_plan(p, n_rows, &wp, &wlds)) return 1000 + wp.R;
  return R;
}

extern "C" int32_t ultr_dnn_backward_tile_rows(const ultr_dnn_desc* d, int64_t n_rows) {
  DnnPlan p;
  BwdPlan bp;
  if (n_rows <= 0 || !ultr_make_dnn_plan(d, n_rows, &p) || !ultr_make_bwd_plan(p, n_rows, &bp)) return -1;
  WideBwd wb;
  size_t wl = 0;
  if (knobs().no_vec == 0 && bwd_wide_plan(p, n_rows, &wb, &wl)) return 1000 + wb.R;
  if (knobs().big_bwd != 0 && ultr_dnn_big_ok(p, n_rows, n_rows) && (big_bwd_wanted(p, n_rows) || bwd_lds_bytes(p, bp.rblk) > 160 * 1024)) return 0;
  return bp.rblk;
}

bool ultr_wgrad_h3_geometry(int64_t T, int M, int K, int* nsplit, int* rows_per_split) {
  if (T <= 0 || M <= 0 || K <= 0 || M % 4 != 0 || K % 4 != 0) return false;
  if (T * (int64_t)(M > K ? M : K) * 4 >= ((int64_t)1 << 31)) return false;  // 32-bit buffer offsets
  const int tiles2 = ((M + 127) / 128) * ((K + 127) / 128);
  int ns = 256 / tiles2;
  if (ns < 1) ns = 1;
  int64_t rps = (T + ns - 1) / ns;
  rps = (rps + 31) / 32 * 32;
  if (rps < 64) rps = 64;
  *rows_per_split = (int)rps;
  *nsplit = (int)((T + rps - 1) / rps);
  return true;
}
int ultr_wgrad_h3_plain(const float* dY, const float* X, int64_t T, int M, int K, float* slabs, hipStream_t st) {
  int ns = 0, rps = 0;
  if (!dY || !X || !slabs || !ultr_wgrad_h3_geometry(T, M, K, &ns, &rps)) return ULTR_E_BADARG;
  if ((((uintptr_t)dY | (uintptr_t)X | (uintptr_t)slabs) & 15) != 0) return ULTR_E_UNSUPPORTED;
  // a one-layer plan around the operands: dz = ws + 0 with ws = dY, the ready-made operand = saved + 0 with saved = X (wg_prenorm),
  // slabs at their distance from dY
  DnnPlan p;
  BwdPlan bp;
  memset(&p, 0, sizeof(p));
  memset(&bp, 0, sizeof(bp));
  p.nl = 2;
  p.M[0] = M; p.K[0] = K;
  bp.N = T;
  bp.wg_prenorm = 1;
  bp.wg_h3 = 1;
  WgradLayer& w = bp.wl[0];
  w.M = M; w.K = K;
  w.nmb = (M + 63) / 64; w.nkb = (K + 63) / 64;
  w.nmb2 = (M + 127) / 128; w.nkb2 = (K + 127) / 128;
  w.nsplit = ns; w.rows_per_split = rps; w.blk_begin = 0; w.vec = 1;
  w.dz_off = 0;
  w.slab_off = (int64_t)(slabs - dY);
  bp.wg_tiles2 = w.nmb2 * w.nkb2;
  bp.wg_live = bp.wg_tiles2 * ns;
  bp.wg_chunk = (bp.wg_live + 7) / 8;
  bp.wgrad_blocks = 8 * bp.wg_chunk;
  hipError_t e = set_lds(dnn_wgrad_h3_kernel, (size_t)WH_LDS_BYTES);
  if (e != hipSuccess) return (int)e;
  EarlyReport er = {nullptr, 0u, 0, 1.0f};
  CommDev cd;
  memset(&cd, 0, sizeof(cd));
  hipLaunchKernelGGL(dnn_wgrad_h3_kernel, dim3((unsigned)bp.wgrad_blocks), dim3(512), (size_t)WH_LDS_BYTES, st, p, bp, (const float*)nullptr,
                     (const float*)nullptr, (int64_t)0, (const int32_t*)nullptr, 1, 1, X, const_cast<float*>(dY), (float*)nullptr,
                     (const float*)nullptr, 0, 0, er, cd);
  return (int)hipGetLastError();
}

static int backward_impl(const ultr_dnn_desc* d, const float* params, const float* features, int64_t n_docs,
                         const int32_t* docids, int32_t batch, int32_t list_size, const void* saved, const float* dscores,
                         const void* loss_ws, void* bwd_ws, float* grads, void* stream, FusedSoftmax fl,
                         int fused_rb = 0 /* > 0: the fused forward+backward kernel ran with this many rows per block;
                                             only the weight gradients and the reduction remain */) {
  if (!params || !docids || !saved || (!dscores && !fl.scores) || !bwd_ws || !grads || batch <= 0 || list_size <= 0 ||
      n_docs < 0 || (n_docs > 0 && !features))
    return ULTR_E_BADARG;
  const int64_t N = (int64_t)batch * list_size;
  DnnPlan p;
  BwdPlan bp;
  if (!ultr_make_dnn_plan(d, N, &p) || !ultr_make_bwd_plan(p, N, &bp)) return ULTR_E_BADARG;
  const bool l0g_ok = p.nl >= 2;  // the layer-0 shortcut whenever there is a hidden layer (its A/B knob of round 1 is gone: the explicit du_0 path had rotted)
  const int tail = (int)ultr_tail_len(list_size);
  if (tail > 4096) return ULTR_E_UNSUPPORTED;
  const size_t lds = bwd_lds_bytes(p, bp.rblk);
  const int nw = knobs().bwd_nw;
  const int vm = vecmask_for(p, params, features);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipSuccess;
  float* ws = (float*)bwd_ws;
  const bool av = all_vec(p, vm, N, n_docs) && knobs().no_vec == 0;
  if (bp.wg_h3 && !av) ultr_make_bwd_plan(p, N, &bp, 0);  // the split-half weight gradients take 16-byte paths only
  if (fused_rb > 0) bp.nrb = (int)((N + fused_rb - 1) / fused_rb);  // vector slabs / loss partials: one per fused block
  const bool big = fused_rb == 0 && dscores != nullptr && av && l0g_ok && ultr_dnn_big_ok(p, N, n_docs) &&
                   knobs().big_bwd != 0 && (big_bwd_wanted(p, N) || lds > 160 * 1024);
  if (lds > 160 * 1024 && !big) return ULTR_E_UNSUPPORTED;
#define LAUNCH_BWD(RR, NWW, VV)                                                                                        \
  do {                                                                                                                 \
    e = set_lds(dnn_bwd_kernel<RR, NWW, VV>, lds);                                                                     \
    if (e != hipSuccess) return (int)e;                                                                                \
    ULTR_LAUNCH(prof, (dnn_bwd_kernel<RR, NWW, VV>), dim3(bp.nrb), dim3(NWW * 64), lds, st, p, bp, params, features,   \
                       n_docs, docids, (int)batch, (int)list_size, (const float*)saved, dscores, ws, vm, fl);         \
  } while (0)
#define LAUNCH_BWD2(RR, NWW) \
  do {                       \
    if (av) LAUNCH_BWD(RR, NWW, true); \
    else LAUNCH_BWD(RR, NWW, false);   \
  } while (0)
  // fast variant: aligned shapes, K_j <= 512, 8 waves, LDS budget (see dnn_bwd2_kernel)
  const size_t lds2 = bwd2_lds_floats(p, bp.rblk, 8) * sizeof(float);
  const bool v2 = av && nw == 8 && p.maxdim <= 512 && lds2 <= 160 * 1024 && p.sv_total * 4 < ((int64_t)1 << 31) &&
                  p.P * 4 < ((int64_t)1 << 31);
#define LAUNCH_BWDV2(RR, XX)                                                                                           \
  do {                                                                                                                 \
    e = set_lds(dnn_bwd2_kernel<RR, 8, XX>, lds2);                                                                     \
    if (e != hipSuccess) return (int)e;                                                                                \
    ULTR_LAUNCH(prof, (dnn_bwd2_kernel<RR, 8, XX>), dim3(bp.nrb), dim3(512), lds2, st, p, bp, params, features,        \
                       n_docs, docids, (int)batch, (int)list_size, (const float*)saved, dscores, ws, fl, g_ultr_step_wt); \
  } while (0)
  bp.l0g = l0g_ok ? 1 : 0;  // every backward kernel skips du_0; the wgrad launch makes up for it
  bp.wg_prenorm = (fused_rb > 0) ? 1 : 0;  // the fused kernel left the ready-made wgrad operands in `saved`
  WideBwd wb;
  size_t wblds = 0;
  const bool wide = fused_rb == 0 && dscores != nullptr && av && l0g_ok && g_ultr_step_wt != nullptr && ((uintptr_t)g_ultr_step_wt & 15) == 0 &&
                    bwd_wide_plan(p, N, &wb, &wblds);
  if (fused_rb > 0) {
    // the row-local half already ran inside dnn_fb_kernel
  } else if (wide) {
    UltrProfScope prof(ULTR_K_BWD, st);
    bp.rblk = wb.R;
    bp.nrb = (int)((N + wb.R - 1) / wb.R);  // one vector slab per workgroup
#define LAUNCH_BWDW(RTT)                                                                                                      \
  do {                                                                                                                        \
    e = set_lds(dnn_bwdw_kernel<RTT>, wblds);                                                                                 \
    if (e != hipSuccess) return (int)e;                                                                                       \
    ULTR_LAUNCH(prof, (dnn_bwdw_kernel<RTT>), dim3(bp.nrb), dim3(1024), wblds, st, p, bp, wb, (const float*)saved, dscores, ws, \
                g_ultr_step_wt);                                                                                              \
  } while (0)
    if (wb.R <= 32) LAUNCH_BWDW(2);
    else if (wb.R <= 48) LAUNCH_BWDW(3);
    else LAUNCH_BWDW(4);
#undef LAUNCH_BWDW
  } else if (big) {
    UltrProfScope prof(ULTR_K_BWD, st);
    bp.nrb = (int)((N + ULTR_BIG_ROWS - 1) / ULTR_BIG_ROWS);  // one vector slab per row block of the row kernels
    const int rc = ultr_dnn_big_backward(p, bp, params, (const float*)saved, dscores, ws, st, prof.on ? prof.a : nullptr,
                                         prof.on ? prof.b : nullptr, knobs().bwd_h3 != 0 && !p.no_h3 && g_ultr_step_wt != nullptr);
    if (rc) return rc;
  } else if (v2) {
    UltrProfScope prof(ULTR_K_BWD, st);
    const int xc = p.maxdim <= 256 ? 1 : 2;
    if (bp.rblk == 16 && xc == 1) LAUNCH_BWDV2(16, 1);
    else if (bp.rblk == 16) LAUNCH_BWDV2(16, 2);
    else if (xc == 1) LAUNCH_BWDV2(32, 1);
    else LAUNCH_BWDV2(32, 2);
  } else {
    UltrProfScope prof(ULTR_K_BWD, st);
    if (bp.rblk == 16 && nw == 4) LAUNCH_BWD2(16, 4);
    else if (bp.rblk == 16 && nw == 16) LAUNCH_BWD2(16, 16);
    else if (bp.rblk == 16) LAUNCH_BWD2(16, 8);
    else if (nw == 4) LAUNCH_BWD2(32, 4);
    else LAUNCH_BWD2(32, 8);
  }
#undef LAUNCH_BWD2
#undef LAUNCH_BWDV2
#undef LAUNCH_BWD
  e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  const float* lp = (const float*)loss_ws;
  const int nlp = fl.scores ? bp.nrb : (int)ultr_loss_parts(batch);
  if (nlp > 1024) {
    int len = (nlp + 63) / 64;
    len = len < 1024 ? 1024 : (len + 31) / 32 * 32;
    bp.lf_len = len;
    bp.lf_chunks = (nlp + len - 1) / len;
  }
  if (bp.wg_h3) {
    UltrProfScope prof(ULTR_K_WGRAD, st);
    e = set_lds(dnn_wgrad_h3_kernel, (size_t)WH_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    const dim3 wgrid(bp.wgrad_blocks + bp.vred_blocks + (bp.lf_chunks > 0 ? bp.lf_chunks : 1));
    EarlyReport er = g_ultr_early;
    CommDev cd;
    memset(&cd, 0, sizeof(cd));
    if (g_ultr_step_xchg.comm != nullptr && g_ultr_step_xchg.er.host != nullptr && bp.lf_chunks == 0 &&
        ultr_comm_dev(g_ultr_step_xchg.comm, g_ultr_step_xchg.step, p.P + tail, &cd)) {
      er = g_ultr_step_xchg.er;  // (see the register kernel's launch below)
      g_ultr_step_xchg.er.host = nullptr;
    } else {
      cd.world = 0;
    }
    if (bp.lf_chunks > 0) er.host = nullptr;
    ULTR_LAUNCH(prof, dnn_wgrad_h3_kernel, wgrid, dim3(512), (size_t)WH_LDS_BYTES, st, p, bp, params, features, n_docs, docids, (int)batch,
                (int)list_size, (const float*)saved, ws, grads, lp, nlp, tail, er, cd);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  } else {
    UltrProfScope prof(ULTR_K_WGRAD, st);
    int maxrps = 0;
    for (int j = 0; j < p.nl - 1; ++j) maxrps = bp.wl[j].rows_per_split > maxrps ? bp.wl[j].rows_per_split : maxrps;
    const size_t wlds = (size_t)(4 * 64 * 64 + 4 * 64 + maxrps) * sizeof(float);
    e = av ? set_lds(dnn_wgrad_kernel<true>, wlds) : set_lds(dnn_wgrad_kernel<false>, wlds);
    if (e != hipSuccess) return (int)e;
    const dim3 wgrid(bp.wgrad_blocks + bp.vred_blocks + (bp.lf_chunks > 0 ? bp.lf_chunks : 1));
    EarlyReport er = g_ultr_early;
    CommDev cd;
    memset(&cd, 0, sizeof(cd));  // world 0: not a data-parallel step
    if (g_ultr_step_xchg.comm != nullptr && g_ultr_step_xchg.er.host != nullptr && bp.lf_chunks == 0 &&
        ultr_comm_dev(g_ultr_step_xchg.comm, g_ultr_step_xchg.step, p.P + tail, &cd)) {
      // data-parallel step: the workgroup that folds the loss partials exchanges the head of the tail with the peers and reports
      // the loss NOW; the gradient exchange behind this launch then has nothing to report
      er = g_ultr_step_xchg.er;
      g_ultr_step_xchg.er.host = nullptr;
    } else {
      cd.world = 0;
    }
    if (bp.lf_chunks > 0) er.host = nullptr;  // two-level fold of > 1024 partials: the loss is only final in the reduction launch
    if (av)
      ULTR_LAUNCH(prof, dnn_wgrad_kernel<true>, wgrid, dim3(256), wlds, st, p, bp, params, features, n_docs,
                         docids, (int)batch, (int)list_size, (const float*)saved, ws, (vm >> 31) & 1, grads, lp, nlp, tail, er, cd);
    else
      ULTR_LAUNCH(prof, dnn_wgrad_kernel<false>, wgrid, dim3(256), wlds, st, p, bp, params, features, n_docs,
                         docids, (int)batch, (int)list_size, (const float*)saved, ws, (vm >> 31) & 1, grads, lp, nlp, tail, er, cd);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  RedPlan rp;
  ultr_make_red_plan(p, bp, &rp);
  const int nblk = (int)ultr_red_blocks(p.P, tail);
  UltrProfScope prof(ULTR_K_REDUCE, st);
  // one thread per element (full_sum: any number of slabs, the same bits as the cooperating groups) as long as the BIG segments -
  // the weight slabs - have at most 32 parts; the layer-0 gamma / beta partials (2 K_0 elements, one part per 64-row block of W_0
  // and row split: 56 at config 4 with the split-half launch's 7 splits) may have up to 128
  int maxparts = 1;
  for (int k = 0; k < rp.nseg; ++k) {
    const int np = rp.seg[k].len > 4096 ? rp.seg[k].nparts : (rp.seg[k].nparts + 3) / 4;
    maxparts = np > maxparts ? np : maxparts;
  }
  if (g_ultr_step_xchg.comm != nullptr && maxparts <= 32 && bp.lf_chunks == 0) {
    // data-parallel step: this launch exchanges its own output (grad_reduce_xchg_kernel); ultr_train_step then skips the exchange kernel
    CommDev cd;
    if (ultr_comm_dev(g_ultr_step_xchg.comm, g_ultr_step_xchg.step, p.P + tail, &cd)) {
      const dim3 xg((unsigned)((p.P + tail + 255) / 256));
#define XCHG_LAUNCH(WW) \
  ULTR_LAUNCH(prof, grad_reduce_xchg_kernel<WW>, xg, dim3(256), 0, st, rp, p.P, tail, (const float*)ws, grads, ws + bp.sumsq_off, nblk, cd, g_ultr_step_xchg.er, ws + ultr_sumsq2_off(p.P))
      switch (cd.world) {
        case 1: XCHG_LAUNCH(1); break;
        case 2: XCHG_LAUNCH(2); break;
        case 3: XCHG_LAUNCH(3); break;
        case 4: XCHG_LAUNCH(4); break;
        case 5: XCHG_LAUNCH(5); break;
        case 6: XCHG_LAUNCH(6); break;
        case 7: XCHG_LAUNCH(7); break;
        default: XCHG_LAUNCH(8); break;
      }
#undef XCHG_LAUNCH
      g_ultr_step_xchg.done = true;
      g_ultr_step_nsq2 = (int)xg.x;
      return (int)hipGetLastError();
    }
  }
  if (maxparts <= 32) {
    // (level-2 partials only without the extra loss-fold workgroup: its index would be a level-2 slot)
    float* s2 = bp.lf_chunks == 0 ? ws + ultr_sumsq2_off(p.P) : nullptr;
    ULTR_LAUNCH(prof, grad_reduce_kernel<true>, dim3((nblk + 3) / 4 + (bp.lf_chunks > 0 ? 1 : 0)), dim3(256), 0, st, rp, p.P, tail,
                (const float*)ws, (const float*)(ws + bp.lfold_off), bp.lf_chunks, grads, ws + bp.sumsq_off, nblk, s2);
    if (s2 != nullptr) g_ultr_step_nsq2 = (nblk + 3) / 4;
  } else {
    ULTR_LAUNCH(prof, grad_reduce_kernel<false>, dim3(nblk + (bp.lf_chunks > 0 ? 1 : 0)), dim3(256), 0, st, rp, p.P, tail,
                (const float*)ws, (const float*)(ws + bp.lfold_off), bp.lf_chunks, grads, ws + bp.sumsq_off, nblk, (float*)nullptr);
  }
  return (int)hipGetLastError();
}

extern "C" int ultr_dnn_backward(const ultr_dnn_desc* d, const float* params, const float* features, int64_t n_docs,
                                 const int32_t* docids, int32_t batch, int32_t list_size, const void* saved,
                                 const float* dscores, const void* loss_ws, void* bwd_ws, float* grads, void* stream) {
  FusedSoftmax fl = {nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr};
  return backward_impl(d, params, features, n_docs, docids, batch, list_size, saved, dscores, loss_ws, bwd_ws, grads, stream, fl);
}

extern "C" int ultr_dnn_backward_softmax(const ultr_dnn_desc* d, const float* params, const float* features, int64_t n_docs,
                                         const int32_t* docids, int32_t batch, int32_t list_size, const void* saved,
                                         const float* scores, const float* labels, const float* pw, const float* ipw_table,
                                         int32_t n_ipw, float* dscores_out, void* loss_ws, void* bwd_ws, float* grads,
                                         void* stream) {
  if (!scores || !labels || !loss_ws || (ipw_table && n_ipw <= 0)) return ULTR_E_BADARG;
  {
    // big batches take the per-layer backward, which wants the loss as its own stage
    DnnPlan p;
    const int64_t N = (int64_t)batch * list_size;
    // ... and so do shapes whose row tile does not fit the LDS (a layer wider than 512 at a small batch): the same condition
    // backward_impl applies to the other algorithms - without it this entry point returned ULTR_E_UNSUPPORTED there
    BwdPlan bp0;
    const bool planned = dscores_out && batch > 0 && list_size > 0 && ultr_make_dnn_plan(d, N, &p) && ultr_make_bwd_plan(p, N, &bp0);
    // ... and the wide-tile backward of ultr_train_step (dnn_bwdw_kernel), which takes dscores too
    WideBwd wb0;
    size_t wl0 = 0;
    const bool wide = planned && g_ultr_step_wt != nullptr && knobs().no_vec == 0 && bwd_wide_plan(p, N, &wb0, &wl0);
    if (planned && (wide || ((big_bwd_wanted(p, N) || bwd_lds_bytes(p, bp0.rblk) > 160 * 1024) && knobs().big_bwd != 0 &&
        knobs().no_vec == 0 && ultr_dnn_big_ok(p, N, n_docs)))) {
      const int rc = ultr_softmax_ce(scores, labels, pw, ipw_table, n_ipw, batch, list_size, dscores_out, loss_ws, stream);
      if (rc) return rc;
      FusedSoftmax none = {nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr};
      return backward_impl(d, params, features, n_docs, docids, batch, list_size, saved, dscores_out, loss_ws, bwd_ws, grads, stream, none);
    }
  }
  FusedSoftmax fl = {scores, labels, pw, ipw_table, (int)n_ipw, dscores_out, (float*)loss_ws};
  return backward_impl(d, params, features, n_docs, docids, batch, list_size, saved, nullptr, loss_ws, bwd_ws, grads, stream, fl);
}

// internal (ultr_train_step): forward + NA/IPW loss + backward for a small batch through dnn_fb_kernel, then the weight
// gradients + reduction.  Returns ULTR_E_UNSUPPORTED when the shape does not qualify - the caller then issues the
// separate forward / backward calls.
int ultr_fused_step_softmax(const ultr_dnn_desc* d, const float* params, const float* wt, const float* features, int64_t n_docs,
                            const int32_t* docids, int32_t batch, int32_t list_size, float* scores, void* saved,
                            const float* labels, const float* pw, const float* ipw_table, int32_t n_ipw, float* dscores_out,
                            void* loss_ws, void* bwd_ws, float* grads, void* stream) {
  if (!params || !wt || !docids || !scores || !saved || !labels || !loss_ws || !bwd_ws || !grads || batch <= 0 ||
      list_size <= 0 || n_docs < 0 || (n_docs > 0 && !features) || (ipw_table && n_ipw <= 0))
    return ULTR_E_BADARG;
  if (knobs().no_fused_fb != 0) return ULTR_E_UNSUPPORTED;
  const int L = list_size;
  if (L > 16) return ULTR_E_UNSUPPORTED;
  const int64_t N = (int64_t)batch * L;
  DnnPlan p;
  BwdPlan bp;
  if (!ultr_make_dnn_plan(d, N, &p) || !ultr_make_bwd_plan(p, N, &bp)) return ULTR_E_BADARG;
  const int lpb = 16 / L, rb = lpb * L;
  const int64_t nblk = (batch + lpb - 1) / lpb;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      cus = 256;
  }
  const int vm = vecmask_for(p, params, features);
  const size_t lds = fb_lds_floats(p) * sizeof(float);
  const bool ok = all_vec(p, vm, N, n_docs) && knobs().no_vec == 0 && ((uintptr_t)wt & 15) == 0 &&
                  p.nl >= 2 && p.maxdim <= 512 && p.pv_total <= 3 * 512 * 4 && lds <= 160 * 1024 &&
                  nblk <= (int64_t)knobs().fb_max_wg_per_cu * cus &&  // one round of workgroups (tools/fused_threshold.py:
                                                                          // B=256 62 vs 68 us, B=288 94 vs 70 us)
                  nblk <= (N + 8) / 9 + 1 &&                               // vector-slab allocation (>= 9 live rows / block)
                  p.sv_total * 4 < ((int64_t)1 << 31) && p.P * 4 < ((int64_t)1 << 31);
  if (!ok) return ULTR_E_UNSUPPORTED;
  bp.nrb = (int)nblk;
  bp.l0g = p.nl >= 2 ? 1 : 0;  // must match backward_impl's choice
  bp.wg_prenorm = 1;
  hipStream_t st = (hipStream_t)stream;
  FusedSoftmax fl = {nullptr, labels, pw, ipw_table, (int)n_ipw, dscores_out, (float*)loss_ws};
  float* ws = (float*)bwd_ws;
  FbPlan fp;
  memset(&fp, 0, sizeof(fp));
  for (int j = 0; j < p.nl; ++j) {
    int* r = fp.rec[j];
    auto put64 = [&](int k, int64_t v) { r[k] = (int)(uint32_t)(uint64_t)v; r[k + 1] = (int)(uint32_t)((uint64_t)v >> 32); };
    r[FbPlan::K] = p.K[j]; r[FbPlan::M] = p.M[j]; r[FbPlan::PV_OFF] = p.pv_off[j];
    r[FbPlan::KSPLIT] = p.fl[j].ksplit; r[FbPlan::KLEN] = p.fl[j].klen; r[FbPlan::NCH] = p.fl[j].nch;
    r[FbPlan::BWD_NCH] = p.bwd_nch[j]; r[FbPlan::BWD_MSPLIT] = p.bwd_msplit[j]; r[FbPlan::BWD_MODE] = p.bwd_mode[j];
    r[FbPlan::BWD_MLEN] = p.bwd_mlen[j]; r[FbPlan::VOFF_G] = bp.voff_g[j]; r[FbPlan::VOFF_B] = bp.voff_b[j];
    put64(FbPlan::WSF_OFF, p.wsf_off[j]); put64(FbPlan::WSB_OFF, p.wsb_off[j]); put64(FbPlan::SV_X, p.sv_x[j]);
    put64(FbPlan::OFF_W, p.off_w[j]); put64(FbPlan::SV_MEAN, p.sv_mean[j]); put64(FbPlan::SV_RSTD, p.sv_rstd[j]);
    put64(FbPlan::DZ_OFF, j < p.nl - 1 ? bp.dz_off[j] : 0); put64(FbPlan::WT_OFF, p.wt_off[j]);
    put64(FbPlan::WHF_OFF, p.whf_off[j]); put64(FbPlan::WHB_OFF, p.whb_off[j]);
  }
  hipError_t e;
  {
    UltrProfScope prof(ULTR_K_FUSED, st);
#define LAUNCH_FB(XX, HH)                                                                                                      \
  do {                                                                                                                         \
    e = set_lds(dnn_fb_kernel<XX, HH>, lds);                                                                                   \
    if (e != hipSuccess) return (int)e;                                                                                        \
    ULTR_LAUNCH(prof, (dnn_fb_kernel<XX, HH>), dim3((unsigned)nblk), dim3(512), lds, st, p, bp, params, wt, features, n_docs,  \
                docids, (int)batch, L, lpb, scores, (float*)saved, ws, fl, fp);                                                \
  } while (0)
    // products on the fp16 matrix cores with split operands where the plan has the split-half copies (ULTR_FB_H3=0: fp32 MFMAs)
    const bool h3 = p.fb_h3 != 0;
    if (p.maxdim <= 256) {
      if (h3) LAUNCH_FB(1, true);
      else LAUNCH_FB(1, false);
    } else {
      if (h3) LAUNCH_FB(2, true);
      else LAUNCH_FB(2, false);
    }
#undef LAUNCH_FB
  }
  e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  fl.scores = scores;  // marks "loss partials come one per row block" for the reduction
  return backward_impl(d, params, features, n_docs, docids, batch, list_size, saved, nullptr, loss_ws, bwd_ws, grads, stream, fl,
                       rb);
}

extern "C" int ultr_grad_sumsq(float* grads, int64_t n_params, int32_t list_size, void* bwd_ws, void* stream) {
  if (!grads || !bwd_ws || n_params <= 0) return ULTR_E_BADARG;
  const int tail = (int)ultr_tail_len(list_size);
  const int nblk = (int)ultr_red_blocks(n_params, tail);
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3(nblk), dim3(64), 0, (hipStream_t)stream, n_params, (const float*)grads,
                     (float*)bwd_ws);  // sumsq partials live at offset 0 of bwd_ws
  return (int)hipGetLastError();
}
