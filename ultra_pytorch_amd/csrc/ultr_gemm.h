// ultr_gemm.h — LDS-tiled fp32 GEMM on the matrix cores for the THROUGHPUT regime (many rows per launch).
//
//   C[R x N] = epilogue( A'[R x K] . B[K x N] )        v_mfma_f32_16x16x4_f32, exact fp32
//
// The fused per-workgroup kernels of ultr_dnn.hip own 16 rows end to end and re-stream every weight matrix per 16 rows
// (8 flops per weight byte): right for a few thousand rows, where latency decides, but capped near 38 % of the matrix
// peak at tens of thousands of rows (round-1 profiles).  Here a workgroup owns a 128 x BN tile of ONE layer's output:
// every A and B element staged in LDS feeds 128 / BN outputs, the activations make one HBM round trip per layer
// (fine: >= 100 flops per byte at these shapes), and the per-row work that needs whole rows (LayerNorm statistics,
// LayerNorm backward) lives in separate HBM-bound row kernels (ultr_dnn_big.hip).
//
// Structure (one k-tile = 32 contraction steps):
//   * 256 threads = 4 waves as WM x WN; a wave owns (16 RT) x 64 outputs in RT x 4 accumulator tiles;
//   * A tile [BM][32 (+8 pad)] and B tile in LDS, double-buffered; the NEXT tile's global loads are issued into
//     registers before the MFMAs of the current one and written to the other LDS buffer after them: ONE barrier per
//     k-tile, global latency hidden behind 128 MFMAs per wave;
//   * fragments: A by ds_read_b128 along k (lane (i, q) holds A[row i][k0 + 4q .. +3] = four k-steps); B either
//       k-major  B[k][n]  (NN: weights stored [K][N]; the dgrad's W[M][K] with M as the contraction):  one b128 read
//                         B[k0 + 4q + s][n0 + 4i .. +3] per k-step s feeds FOUR interleaved column tiles, or
//       n-major  B[n][k]  (NT: nn.Linear's weight [out][in] read as it is stored):  one b128 read B[n0 + 16t + i][k0 + 4q ..]
//                         per column tile t feeds four k-steps;
//     either way 2 RT.. 8 LDS reads per 16 RT MFMAs;
//   * the A operand is produced by a functor at staging time (plain rows, rows gathered by document id, LayerNorm
//     applied on the fly from per-row statistics), the epilogue is a functor applied to float4 pieces of the output
//     tile after a transposing trip through LDS (coalesced 16-byte stores for both B layouts; bias / activation /
//     mask / accumulate ride there);
//   * out-of-range rows / columns / contraction indices are zeros at staging (buffer-resource bounds checks), stores
//     are masked: any R, N, K (K % 4 == 0 and 16-byte aligned rows for the vector loads; checked by the launcher).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <type_traits>

#include "ultr_device.h"

namespace ugemm {

#ifdef UGEMM_DEBUG_CYC
__device__ unsigned long long g_ugemm_cyc[4096], g_ugemm_xcc[4096];
#endif
#ifndef UGEMM_STAGES
// LDS ring depth.  3 = fragments of the next half k-tile are read while the current half's MFMAs run (register
// double-buffered fragments; a chunk starts with two tiles).  Measured (tools/gemm_tile_ubench.hip): it helps only where the
// k loop is long (16384 x 4096 x 512: 98 -> 103 TFLOP/s) and loses at the shapes this library runs (12800 x 700 x 512: 86 -> 78,
// 102400 x 256 x 256: 84 -> 77: the second tile of every chunk prologue and one workgroup fewer per CU cost more than the
// LDS round trip per barrier that it hides) - kept selectable, default 2.
#define UGEMM_STAGES 2
#endif
// 1 = all ten fragment reads of a k-tile in front of its MFMAs (counted lgkmcnt waits) instead of hipcc's own order (two
// reads, wait, eight MFMAs - four LDS round trips per k-tile but 52 registers).  Measured: 102 registers -> two workgroups
// per CU instead of three, 101 -> 91 TFLOP/s at 12288 x 704 x 512: occupancy beats the shorter dependency chain.  Default 0.
constexpr int BK = 32;        // contraction steps per k-tile
constexpr int LDA = BK + 8;   // LDS row stride of k-contiguous tiles (A, n-major B): stride = 8 (mod 16) floats is the
                              // conflict-free one for ds_read_b128 with 16 rows x 4 k-groups per wave (its 4 x 16 lane groups)

// ---- A-operand producers ------------------------------------------------------------------------------------------------
// A thread stages the same rows (and the same four contraction columns k4 of every k-tile) throughout the k loop:
//   row(r, rend)      once per staged row of a chunk (rows >= rend read as zeros): whatever is per-row (source offset,
//                     LayerNorm statistics);
//   raw(ctx, k)       ISSUES the global load of [k, k+4) of that row for one k-tile (no arithmetic on the result here:
//                     the load must stay in flight behind the current tile's MFMAs);
//   cols(k) / finish  per-column operands and the arithmetic, applied when the registers are written to LDS.
struct APlain {  // A[r][k], row stride lda; rows >= R and k >= K read as zero
  const float* a;
  int64_t R;
  int K, lda;
  struct Row { unsigned off; };
  struct Cols {};
  __device__ __forceinline__ Row row(int64_t r, int64_t rend) const { return Row{r < rend ? (unsigned)(r * lda * 4) : ULTR_OOB}; }
  __device__ __forceinline__ float4 raw(const Row& c, int k) const {
    return buf_ld4(make_src(a, R * lda), (c.off != ULTR_OOB && k < K) ? c.off + (unsigned)k * 4u : ULTR_OOB);
  }
  __device__ __forceinline__ Cols cols(int) const { return Cols{}; }
  __device__ __forceinline__ float4 finish(const Row&, const Cols&, int, float4 v) const { return v; }
};
// LayerNorm applied on the fly: (x - mean[r]) * rstd[r] * gamma[k] + beta[k]; x rows either consecutive (ids == nullptr)
// or gathered through the feed's position-major document ids (row r = b * L + l reads x[ids[l * B + b]]; PAD = zero row)
struct ALayerNorm {
  const float* x;         // [x_rows][K]
  const float* gb;        // the flat parameter vector (gamma at g_off, beta at b_off), gb_floats long
  int64_t x_rows, gb_floats;
  const float* mean;      // [R]
  const float* rstd;      // [R]
  const int32_t* ids;     // nullptr: row r of x
  int64_t R, n_docs, g_off, b_off;
  int K, B, L;
  struct Row { unsigned off; float mean, rstd; bool in; };
  struct Cols { float4 g, b; };
  __device__ __forceinline__ Row row(int64_t r, int64_t rend) const {
    Row c;
    c.in = r < rend;
    c.mean = c.in ? mean[r] : 0.f;
    c.rstd = c.in ? rstd[r] : 0.f;
    int64_t src = r;
    bool live = c.in;
    if (ids != nullptr && c.in) {
      const uint32_t rr = (uint32_t)r;
      const int32_t id = ids[(int64_t)(rr % (uint32_t)L) * B + rr / (uint32_t)L];
      live = id >= 0 && id < n_docs;
      src = id;
    }
    c.off = live ? (unsigned)(src * K * 4) : ULTR_OOB;
    return c;
  }
  __device__ __forceinline__ float4 raw(const Row& c, int k) const {
    return buf_ld4(make_src(x, x_rows * K), (c.off != ULTR_OOB && k < K) ? c.off + (unsigned)k * 4u : ULTR_OOB);
  }
  __device__ __forceinline__ Cols cols(int k) const {
    Cols o;
    const Src g = make_src(gb, gb_floats);
    o.g = buf_ld4(g, k < K ? (unsigned)((g_off + k) * 4) : ULTR_OOB);
    o.b = buf_ld4(g, k < K ? (unsigned)((b_off + k) * 4) : ULTR_OOB);
    return o;
  }
  // rows past R and columns past K must come out as exact zeros (they meet finite B values): gamma = beta = 0 past K,
  // rstd = 0 and beta masked past R
  __device__ __forceinline__ float4 finish(const Row& c, const Cols& o, int, float4 v) const {
    const float m = c.in ? 1.f : 0.f;
    return make_float4((v.x - c.mean) * c.rstd * o.g.x + m * o.b.x, (v.y - c.mean) * c.rstd * o.g.y + m * o.b.y,
                       (v.z - c.mean) * c.rstd * o.g.z + m * o.b.z, (v.w - c.mean) * c.rstd * o.g.w + m * o.b.w);
  }
};

// x * gamma[k] + beta[k] on consecutive rows (x = an already normalised input, e.g. xhat_0 written by the statistics pass):
// no per-row operands, no gather
struct AAffine {
  const float* x;   // [R][K]
  const float* gb;  // the flat parameter vector (gamma at g_off, beta at b_off), gb_floats long
  int64_t R, gb_floats, g_off, b_off;
  int K;
  struct Row { unsigned off; bool in; };
  struct Cols { float4 g, b; };
  __device__ __forceinline__ Row row(int64_t r, int64_t rend) const {
    Row c;
    c.in = r < rend;
    c.off = c.in ? (unsigned)(r * K * 4) : ULTR_OOB;
    return c;
  }
  __device__ __forceinline__ float4 raw(const Row& c, int k) const {
    return buf_ld4(make_src(x, R * K), (c.off != ULTR_OOB && k < K) ? c.off + (unsigned)k * 4u : ULTR_OOB);
  }
  __device__ __forceinline__ Cols cols(int k) const {
    Cols o;
    const Src g = make_src(gb, gb_floats);
    o.g = buf_ld4(g, k < K ? (unsigned)((g_off + k) * 4) : ULTR_OOB);
    o.b = buf_ld4(g, k < K ? (unsigned)((b_off + k) * 4) : ULTR_OOB);
    return o;
  }
  __device__ __forceinline__ float4 finish(const Row& c, const Cols& o, int, float4 v) const {
    const float m = c.in ? 1.f : 0.f;
    return make_float4(v.x * o.g.x + m * o.b.x, v.y * o.g.y + m * o.b.y, v.z * o.g.z + m * o.b.z, v.w * o.g.w + m * o.b.w);
  }
};

// ---- epilogues: called with the global row, the first of four consecutive columns (all < N unless noted) -------------
// y = act(c + bias); act: -1 none, 0 elu, 1 relu.  n_valid = columns of this piece that exist (1..4)
struct EBiasAct {
  float* y;
  const float* bias;  // may be nullptr
  int ldy, act;
  struct Pre {};
  __device__ __forceinline__ Pre prefetch(int64_t, int, int) const { return Pre{}; }
  __device__ __forceinline__ void operator()(int64_t r, int c, float4 v, int n_valid, const Pre&) const { (*this)(r, c, v, n_valid); }
  __device__ __forceinline__ void operator()(int64_t r, int c, float4 v, int n_valid) const {
    float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < n_valid) {
        float t = o[j] + (bias != nullptr ? bias[c + j] : 0.f);
        o[j] = (act < 0) ? t : act_fwd(t, act);
      }
    }
    float* dst = y + r * ldy + c;
    if (n_valid == 4 && ((ldy & 3) == 0)) st4_out(dst, make_float4(o[0], o[1], o[2], o[3]));
    else
      for (int j = 0; j < n_valid; ++j) dst[j] = o[j];
  }
};
// y = (c + bias) + res: a Linear's output added to the residual stream (the pre-LayerNorm sum of a transformer block); the
// association matches "a + (b + bias)" of a separate residual pass bit for bit
struct EBiasRes {
  float* y;
  const float* bias;  // may be nullptr
  const float* res;   // same shape as y
  int ldy;
  // the residual piece is REQUESTED before the accumulators go through LDS (n-major epilogue) and consumed behind the barrier
  struct Pre { float4 r; };
  __device__ __forceinline__ Pre prefetch(int64_t r, int c, int n_valid) const {
    Pre p;
    const float* rs = res + r * ldy + c;
    if (n_valid == 4 && ((ldy & 3) == 0)) p.r = ld4(rs);
    else p.r = make_float4(rs[0], n_valid > 1 ? rs[1] : 0.f, n_valid > 2 ? rs[2] : 0.f, 0.f);
    return p;
  }
  __device__ __forceinline__ void operator()(int64_t r, int c, float4 v, int n_valid, const Pre& p) const {
    float o[4] = {v.x, v.y, v.z, v.w};
    const float rv[4] = {p.r.x, p.r.y, p.r.z, p.r.w};
    float* dst = y + r * ldy + c;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < n_valid) o[j] = (o[j] + (bias != nullptr ? bias[c + j] : 0.f)) + rv[j];
    if (n_valid == 4 && ((ldy & 3) == 0)) st4_out(dst, make_float4(o[0], o[1], o[2], o[3]));
    else
      for (int j = 0; j < n_valid; ++j) dst[j] = o[j];
  }
  __device__ __forceinline__ void operator()(int64_t r, int c, float4 v, int n_valid) const { (*this)(r, c, v, n_valid, prefetch(r, c, n_valid)); }
};
// y = (accumulate ? y : 0) + c, then (mask != nullptr) zeroed where mask <= 0 (ReLU backward through the saved output)
struct EStore {
  float* y;
  const float* mask;  // same shape as y, or nullptr
  int ldy, accumulate;
  struct Pre {};
  __device__ __forceinline__ Pre prefetch(int64_t, int, int) const { return Pre{}; }
  __device__ __forceinline__ void operator()(int64_t r, int c, float4 v, int n_valid, const Pre&) const { (*this)(r, c, v, n_valid); }
  __device__ __forceinline__ void operator()(int64_t r, int c, float4 v, int n_valid) const {
    float o[4] = {v.x, v.y, v.z, v.w};
    float* dst = y + r * ldy + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < n_valid) {
        if (accumulate) o[j] += dst[j];
        if (mask != nullptr && !(mask[r * ldy + c + j] > 0.f)) o[j] = 0.f;
      }
    }
    if (n_valid == 4 && ((ldy & 3) == 0)) st4_out(dst, make_float4(o[0], o[1], o[2], o[3]));
    else
      for (int j = 0; j < n_valid; ++j) dst[j] = o[j];
  }
};

struct Dims {
  int64_t R;  // rows of A and C
  int N, K;   // columns of C, contraction length
  int ldb;    // row stride of B: N-ish for k-major ([K][ldb]), K-ish for n-major ([N][ldb])
};

template <int BM, int BN, int WM, int WN, bool B_NMAJOR>
struct Cfg {
  static constexpr int STAGES = UGEMM_STAGES;
  static constexpr int NT = WM * WN * 64;
  static constexpr int RT = BM / WM / 16;
  static constexpr int CT = BN / WN / 16;
  static_assert(CT == 4, "a wave owns 64 output columns (four column tiles)");
  static constexpr int LDB = B_NMAJOR ? LDA : BN;           // floats per LDS row of the B tile
  static constexpr int A_FLOATS = BM * LDA;
  static constexpr int B_FLOATS = B_NMAJOR ? BN * LDA : BK * BN;
  static constexpr int LDC = BN + 4;
  static constexpr int STAGE_FLOATS = STAGES * (A_FLOATS + B_FLOATS);
  static constexpr int C_FLOATS = BM * LDC;
  static constexpr int LDS_FLOATS = (B_NMAJOR && C_FLOATS > STAGE_FLOATS) ? C_FLOATS : STAGE_FLOATS;
  static constexpr int A_LOADS = BM * (BK / 4) / NT;        // float4 per thread per k-tile
  static constexpr int B_LOADS = (B_NMAJOR ? BN * (BK / 4) : BK * (BN / 4)) / NT;
  static_assert(BM * (BK / 4) % NT == 0 && (BK * BN / 4) % NT == 0, "staging divides evenly");
};

// Balanced persistent schedule.  The unit of work is 16 output rows x BN columns; a workgroup owns a CONTIGUOUS range of
// units (column-block-major: its chunks mostly share one B panel) and walks it in chunks of up to BM / 16 units.  launch()
// starts exactly the workgroups the GPU holds at once, so every workgroup gets total / grid units +- 1: no tail round of
// half-empty compute units, and with RT = 1 (a wave = one 16-row tile) a short last chunk only issues its own MFMAs.  The next chunk's first k-tile is requested before the epilogue of the current
// one (k-major B: the epilogue needs no LDS).
template <int BM, int BN, int WM, int WN, bool B_NMAJOR, class AProd, class Epi>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel(Dims d, AProd aprod, const float* __restrict__ Bg, Epi epi) {
  using C = Cfg<BM, BN, WM, WN, B_NMAJOR>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                                // [STAGES][BM][LDA]
  float* Bs = smem + C::STAGES * C::A_FLOATS;      // [STAGES][...]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int wr = (wave / WN) * (BM / WM), wc = (wave % WN) * (BN / WN);  // wave origin inside the chunk
  const int ncb = (d.N + BN - 1) / BN;
  const int64_t nru = (d.R + 15) / 16;      // 16-row units per column block
  const int64_t U = nru * ncb;
  // consecutive block ids are dealt round-robin to the 8 XCDs: give every XCD a contiguous band of the unit range (its
  // workgroups share B panels and neighbouring A rows in that XCD's L2)
  int64_t w = blockIdx.x;
  const int64_t G = gridDim.x;
  if ((G & 7) == 0) w = (w & 7) * (G >> 3) + (w >> 3);
  int64_t u = w * U / G;
  const int64_t u_end = (w + 1) * U / G;
  if (u >= u_end) return;
  const Src bsrc = make_src(Bg, B_NMAJOR ? (int64_t)d.N * d.ldb : (int64_t)d.K * d.ldb);
  const int nk = (d.K + BK - 1) / BK;

  // ---- the chunk being staged / computed ------------------------------------------------------------------------------
  int64_t r0 = 0, rend = 0;  // rows [r0, rend) of the chunk
  int n0 = 0, m_units = 0;
  constexpr int NSET = C::STAGES == 3 ? 2 : 1;  // register sets of staged tiles (the 3-stage ring starts a chunk with two tiles)
  float4 areg[NSET][C::A_LOADS], breg[NSET][C::B_LOADS];
  const int k4 = (tid & 7) * 4;  // a thread stages rows (tid >> 3) + (NT / 8) j, always columns k4 .. k4 + 3 of a k-tile
  typename AProd::Row arow[C::A_LOADS];
  typename AProd::Cols acol[NSET];
  auto begin_chunk = [&](int64_t uu) {
    const int64_t cb = uu / nru, ru = uu - cb * nru;
    int64_t m = u_end - uu;
    if (m > BM / 16) m = BM / 16;
    if (m > nru - ru) m = nru - ru;
    m_units = (int)m;
    r0 = ru * 16;
    rend = r0 + 16 * m < d.R ? r0 + 16 * m : d.R;
    n0 = (int)cb * BN;
#pragma unroll
    for (int j = 0; j < C::A_LOADS; ++j) arow[j] = aprod.row(r0 + ((tid + C::NT * j) >> 3), rend);
  };
  auto load_tile = [&](int k0, auto SET) {
    constexpr int set = decltype(SET)::value;
    acol[set] = aprod.cols(k0 + k4);
#pragma unroll
    for (int j = 0; j < C::A_LOADS; ++j) areg[set][j] = aprod.raw(arow[j], k0 + k4);
#pragma unroll
    for (int j = 0; j < C::B_LOADS; ++j) {
      const int idx = tid + C::NT * j;
      if constexpr (B_NMAJOR) {
        const int n = idx >> 3, kk4 = (idx & 7) * 4;
        const bool ok = n0 + n < d.N && k0 + kk4 < d.K;
        breg[set][j] = buf_ld4(bsrc, ok ? (unsigned)(((int64_t)(n0 + n) * d.ldb + k0 + kk4) * 4) : ULTR_OOB);
      } else {
        const int kk = idx / (BN / 4), c4 = (idx - kk * (BN / 4)) * 4;
        const bool ok = k0 + kk < d.K && n0 + c4 < d.N;
        breg[set][j] = buf_ld4(bsrc, ok ? (unsigned)(((int64_t)(k0 + kk) * d.ldb + n0 + c4) * 4) : ULTR_OOB);
      }
    }
    // keep the staging loads ABOVE the MFMAs / the epilogue that follow (hipcc otherwise sinks each load to its first use)
    __builtin_amdgcn_sched_barrier(0);
  };
  auto store_tile = [&](int buf, int k0, auto SET) {
    constexpr int set = decltype(SET)::value;
    float* Ab = As + buf * C::A_FLOATS;
    float* Bb = Bs + buf * C::B_FLOATS;
#pragma unroll
    for (int j = 0; j < C::A_LOADS; ++j) {
      const int idx = tid + C::NT * j;
      st4(Ab + (idx >> 3) * LDA + k4, aprod.finish(arow[j], acol[set], k0 + k4, areg[set][j]));
    }
#pragma unroll
    for (int j = 0; j < C::B_LOADS; ++j) {
      const int idx = tid + C::NT * j;
      if constexpr (B_NMAJOR) st4(Bb + (idx >> 3) * LDA + (idx & 7) * 4, breg[set][j]);
      else {
        const int kk = idx / (BN / 4), c4 = (idx - kk * (BN / 4)) * 4;
        st4(Bb + kk * C::LDB + c4, breg[set][j]);
      }
    }
  };
  constexpr std::integral_constant<int, 0> S0{};
  constexpr std::integral_constant<int, NSET - 1> S1{};
  f32x4 acc[C::RT][C::CT];
  // fragments of one half k-tile (16 contraction steps): a[rt] = A[row i of tile rt][16 h + 4 q ..+3];
  // k-major B: b[s] = B[16 h + 4 q + s][wc + 4 i ..+3] (four interleaved column tiles), n-major B: b[t] = B[col i of tile t][16 h + 4 q ..+3]
  auto read_frags = [&](int buf, int h, float4 (&a)[C::RT], float4 (&b)[4]) {
    const float* Ab = As + buf * C::A_FLOATS + (wr + i) * LDA + 4 * q + 16 * h;
    const float* Bb = Bs + buf * C::B_FLOATS;
#pragma unroll
    for (int rt = 0; rt < C::RT; ++rt) a[rt] = ld4(Ab + rt * 16 * LDA);
    if constexpr (B_NMAJOR) {
#pragma unroll
      for (int t = 0; t < C::CT; ++t) b[t] = ld4(Bb + (wc + 16 * t + i) * LDA + 16 * h + 4 * q);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) b[s] = ld4(Bb + (16 * h + 4 * q + s) * C::LDB + wc + 4 * i);
    }
  };
  auto mfma_half = [&](const float4 (&a)[C::RT], const float4 (&b)[4], int live_rt) {  // live_rt: wave-uniform
    if constexpr (B_NMAJOR) {
#pragma unroll
      for (int rt = 0; rt < C::RT; ++rt) {
        if (rt < live_rt) {
          const float av[4] = {a[rt].x, a[rt].y, a[rt].z, a[rt].w};
#pragma unroll
          for (int t = 0; t < C::CT; ++t) {
            const float bv[4] = {b[t].x, b[t].y, b[t].z, b[t].w};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[rt][t] = mfma16(av[s], bv[s], acc[rt][t]);
          }
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float bv[4] = {b[s].x, b[s].y, b[s].z, b[s].w};
#pragma unroll
        for (int rt = 0; rt < C::RT; ++rt) {
          if (rt < live_rt) {
            const float as = (s == 0) ? a[rt].x : (s == 1) ? a[rt].y : (s == 2) ? a[rt].z : a[rt].w;
#pragma unroll
            for (int t = 0; t < C::CT; ++t) acc[rt][t] = mfma16(as, bv[t], acc[rt][t]);
          }
        }
      }
    }
  };

  begin_chunk(u);
  load_tile(0, S0);
  if constexpr (C::STAGES == 3) {
    if (nk > 1) load_tile(BK, S1);
  }
  store_tile(0, 0, S0);
  if constexpr (C::STAGES == 3) {
    if (nk > 1) store_tile(1, BK, S1);
  }
  lds_barrier();
  for (;;) {
#pragma unroll
    for (int rt = 0; rt < C::RT; ++rt)
#pragma unroll
      for (int t = 0; t < C::CT; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int live_rt = (16 * m_units - wr + 15) / 16;  // this wave's row tiles that hold rows of the chunk
    live_rt = live_rt < 0 ? 0 : (live_rt > C::RT ? C::RT : live_rt);
    if constexpr (C::STAGES == 3) {
      // Three LDS stages, fragments double-buffered by half k-tile: while the MFMAs of one half run, the LDS reads of the
      // next half (the other half of this tile, then the first half of the NEXT tile - already in LDS since the previous
      // barrier) are in flight, so a wave's MFMA stream does not stop for an LDS round trip at every barrier.  Tile t + 2
      // goes global -> registers at the top of iteration t and registers -> LDS (stage of tile t - 1) at its bottom.
      float4 fa0[C::RT], fb0[4], fa1[C::RT], fb1[4];
      if (live_rt > 0) read_frags(0, 0, fa0, fb0);
      int st = 0;
      for (int t = 0; t < nk; ++t) {
        const int st1 = st == 2 ? 0 : st + 1, st2 = st1 == 2 ? 0 : st1 + 1;
        if (t + 2 < nk) load_tile((t + 2) * BK, S0);
        if (live_rt > 0) {
          read_frags(st, 1, fa1, fb1);
          __builtin_amdgcn_sched_barrier(0);
          mfma_half(fa0, fb0, live_rt);
          __builtin_amdgcn_sched_barrier(0);
          if (t + 1 < nk) read_frags(st1, 0, fa0, fb0);
          __builtin_amdgcn_sched_barrier(0);
          mfma_half(fa1, fb1, live_rt);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (t + 2 < nk) store_tile(st2, (t + 2) * BK, S0);
        lds_barrier();
        st = st1;
      }
    } else {
      for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) load_tile((t + 1) * BK, S0);
        if (live_rt > 0) {
          // every fragment of the k-tile is requested before the first MFMA: one LDS round trip per 32 contraction steps
          float4 fa0[C::RT], fb0[4], fa1[C::RT], fb1[4];
          read_frags(t & 1, 0, fa0, fb0);
          read_frags(t & 1, 1, fa1, fb1);
          mfma_half(fa0, fb0, live_rt);
          mfma_half(fa1, fb1, live_rt);
        }
        if (t + 1 < nk) store_tile((t + 1) & 1, (t + 1) * BK, S0);
        lds_barrier();
      }
    }
    // ---- this chunk's output; the next chunk's first k-tile(s) go in flight before it (k-major B) ------------------------
    const int64_t e_r0 = r0, e_rend = rend;
    const int e_n0 = n0;
    const int64_t un = u + m_units;
    const bool more = un < u_end;
    auto load_first = [&]() {
      load_tile(0, S0);
      if constexpr (C::STAGES == 3) {
        if (nk > 1) load_tile(BK, S1);
      }
    };
    if (more) {
      begin_chunk(un);
      if constexpr (!B_NMAJOR) load_first();
    }
    if constexpr (!B_NMAJOR) {
      // a lane already holds four consecutive output columns of a row (the four interleaved column tiles): straight to
      // the functor, 256 contiguous bytes per 16 lanes
#pragma unroll
      for (int rt = 0; rt < C::RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = e_r0 + wr + 16 * rt + 4 * q + r;
          const int c = e_n0 + wc + 4 * i;
          if (row < e_rend && c < d.N) {
            const int nv = d.N - c < 4 ? d.N - c : 4;
            epi(row, c, make_float4(acc[rt][0][r], acc[rt][1][r], acc[rt][2][r], acc[rt][3][r]), nv);
          }
        }
    } else {
      // n-major B: a lane holds one column of four tiles; transpose through LDS (the staging buffers are free: every wave
      // passed the barrier behind the last k-tile) into float4 pieces
      float* Cs = smem;
      constexpr int PIECES = BM * (BN / 4);
      constexpr int PPT = PIECES / C::NT;  // pieces per thread
      static_assert(PIECES % C::NT == 0, "epilogue pieces divide evenly");
      // whatever the functor reads besides the accumulators (a residual tile) is requested NOW, in front of the LDS transpose
      typename Epi::Pre pre[PPT];
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const int idx = tid + C::NT * k;
        const int row = idx / (BN / 4), c4 = (idx - row * (BN / 4)) * 4;
        const int64_t r = e_r0 + row;
        const int c = e_n0 + c4;
        if (r < e_rend && c < d.N) pre[k] = epi.prefetch(r, c, d.N - c < 4 ? d.N - c : 4);
      }
#pragma unroll
      for (int rt = 0; rt < C::RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wr + 16 * rt + 4 * q + r;
#pragma unroll
          for (int t = 0; t < C::CT; ++t) Cs[row * C::LDC + wc + 16 * t + i] = acc[rt][t][r];
        }
      lds_barrier();
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const int idx = tid + C::NT * k;
        const int row = idx / (BN / 4), c4 = (idx - row * (BN / 4)) * 4;
        const int64_t r = e_r0 + row;
        const int c = e_n0 + c4;
        if (r < e_rend && c < d.N) {
          const int nv = d.N - c < 4 ? d.N - c : 4;
          epi(r, c, ld4(Cs + row * C::LDC + c4), nv, pre[k]);
        }
      }
      if (more) {
        lds_barrier();  // the tile in LDS has been read
        load_first();
      }
    }
    if (!more) break;
    store_tile(0, 0, S0);
    if constexpr (C::STAGES == 3) {
      if (nk > 1) store_tile(1, BK, S1);
    }
    lds_barrier();
    u = un;
  }
}

// ---- split-half mode (round 4): the same GEMM on the fp16 matrix cores with fp32-grade results -------------------------------------
// C = epi(A' . W^T) with W given as TWO fp16 planes (hi = fp16(2^8 w), lo = fp16(2^8 w - hi); n-major [N][ldb halves], ldb a multiple of
// 32, zero-padded - ultr_split_planes writes them once per step) and A split WHEN IT IS STAGED: the 32 contraction steps of a k-tile
// of one row are scaled by a power of two that brings their largest magnitude just under 2^14, then written to LDS as hi / lo fp16
// planes next to the row's inverse scale.  A k-tile is ONE 16x16x32 step: ah.wh + ah.wl + al.wh into a fresh accumulator (three
// v_mfma_f32_16x16x32_f16, 48 cycles where the fp32 path issues eight v_mfma_f32_16x16x4_f32 = 256), which is then added to the
// running fp32 sum with the row's scale (x 2^-8): per-TILE scaling, so the caller supplies no row statistics and rows whose
// magnitude varies along k lose nothing.  Error budget as DESIGN section 4 (split-half products): 2^-22 relative per product term.
// Same persistent schedule, staging pattern and LDS-transposed epilogue as the n-major fp32 kernel.
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
#ifndef UGEMM_TRACE_STAMP  // phase tracing (-DULTR_TRACE builds of ultr_setrank.hip define it): s_memtime of wave 0 of every 32nd workgroup
#define UGEMM_TRACE_STAMP(slot) do {} while (0)
#endif
constexpr int LDH = BK + 8;  // halves per LDS row of a plane (80 bytes: 16 rows x b128 reads fall into distinct bank groups)
#define UGEMM_H3_WSCALE 256.0f

template <int BM, int BN, int WM, int WN, class AProd, class Epi>
__global__ __launch_bounds__(WM * WN * 64) void gemm_h3_kernel(Dims d, AProd aprod, const _Float16* __restrict__ Bhi,
                                                               const _Float16* __restrict__ Blo, Epi epi) {
  using C = Cfg<BM, BN, WM, WN, true>;
  static_assert(C::STAGES == 2, "split-half mode: two LDS stages");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // per stage: A planes [2][BM][LDH] halves | B planes [2][BN][LDH] halves  (= the fp32 tiles' bytes); then the scales [2][BM]
  _Float16* Ah = reinterpret_cast<_Float16*>(smem);
  _Float16* Bh = reinterpret_cast<_Float16*>(smem + C::STAGES * C::A_FLOATS);
  float* Sc = smem + C::STAGES * (C::A_FLOATS + C::B_FLOATS);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int wr = (wave / WN) * (BM / WM), wc = (wave % WN) * (BN / WN);
  const int ncb = (d.N + BN - 1) / BN;
  const int64_t nru = (d.R + 15) / 16;
  const int64_t U = nru * ncb;
  int64_t w = blockIdx.x;
  const int64_t G = gridDim.x;
  if ((G & 7) == 0) w = (w & 7) * (G >> 3) + (w >> 3);
  int64_t u = w * U / G;
  const int64_t u_end = (w + 1) * U / G;
  if (u >= u_end) return;
  const int64_t bplane = (int64_t)d.N * d.ldb;  // halves per plane
  const __amdgpu_buffer_rsrc_t bhs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(Bhi), 0, (int)(bplane * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t bls = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(Blo), 0, (int)(bplane * 2), 0x00020000);
  const int nk = (d.K + BK - 1) / BK;

  constexpr int B_PIECES = BN * 4 / C::NT;  // 16-byte pieces (8 halves) per thread per plane per k-tile
  static_assert(BN * 4 % C::NT == 0 && B_PIECES >= 1, "B staging divides evenly");
  const int k4 = (tid & 7) * 4;
  // The operand stream.  Tiles are requested TWO ahead of the one being multiplied and the stream runs on across chunk boundaries
  // (the next chunk's first two tiles are in flight while this chunk's epilogue runs): with one tile of look-ahead inside a chunk
  // and none across chunks every 32-deep contraction step waited for an HBM round trip - 2 800 cycles per step around 190 of MFMAs
  // (profiles/r04: tools/prof_big_fwd.sh).  Two register slots; contraction tiles are walked in pairs so that the slot of a tile
  // is static (an odd tile count is padded with one all-zero tile: its loads are out of range).
  struct Chunk {
    int64_t u, r0, rend;
    int n0, m_units;
  };
  struct Slot {
    float4 areg[C::A_LOADS];
    typename AProd::Row arow[C::A_LOADS];
    typename AProd::Cols acol;
    u32x4 bhr[B_PIECES], blr[B_PIECES];
  };
  auto chunk_at = [&](int64_t uu) {
    Chunk c;
    const int64_t cb = uu / nru, ru = uu - cb * nru;
    int64_t m = u_end - uu;
    if (m > BM / 16) m = BM / 16;
    if (m > nru - ru) m = nru - ru;
    c.u = uu;
    c.m_units = (int)m;
    c.r0 = ru * 16;
    c.rend = c.r0 + 16 * m < d.R ? c.r0 + 16 * m : d.R;
    c.n0 = (int)cb * BN;
    return c;
  };
  const int nkp = (nk + 1) & ~1;
  // loader state: the chunk and tile the next request belongs to, the chunk's row descriptors (once per chunk: a producer's row()
  // may read per-row statistics)
  Chunk lc = chunk_at(u);
  int lkt = 0;
  bool lok = true;
  typename AProd::Row lrow[C::A_LOADS];
  auto loader_rows = [&]() {
#pragma unroll
    for (int j = 0; j < C::A_LOADS; ++j) lrow[j] = aprod.row(lc.r0 + ((tid + C::NT * j) >> 3), lc.rend);
  };
  loader_rows();
  // (no control flow around the loads: a load behind a branch makes hipcc's wait-count pass drain vmcnt(0) at every use - seen in this
  // kernel's ISA as vmcnt(2) / (1) / (0) in front of EVERY tile, i.e. no look-ahead at all; past the end of the stream the offsets
  // are out of range instead)
  auto load_next = [&](Slot& sl) {
    const int k0 = lkt * BK;
    const int ka = lok ? k0 + k4 : 0x40000000;
    sl.acol = aprod.cols(k0 + k4);
#pragma unroll
    for (int j = 0; j < C::A_LOADS; ++j) {
      sl.arow[j] = lrow[j];
      sl.areg[j] = aprod.raw(lrow[j], ka);
    }
#pragma unroll
    for (int j = 0; j < B_PIECES; ++j) {
      const int idx = tid + C::NT * j;
      const int n = idx >> 2, kp = (idx & 3) * 8;
      const bool ok = lok && lc.n0 + n < d.N && k0 + kp < d.ldb;
      const unsigned off = ok ? (unsigned)(((int64_t)(lc.n0 + n) * d.ldb + k0 + kp) * 2) : ULTR_OOB;
      sl.bhr[j] = __builtin_amdgcn_raw_buffer_load_b128(bhs, off, 0, 0);
      sl.blr[j] = __builtin_amdgcn_raw_buffer_load_b128(bls, off, 0, 0);
    }
    if (lok && ++lkt == nkp) {
      lkt = 0;
      const int64_t un = lc.u + lc.m_units;
      if (un < u_end) {
        lc = chunk_at(un);
        loader_rows();
      } else {
        lok = false;
      }
    }
  };
  auto store_tile = [&](const Slot& sl, int buf, int k0) {
    _Float16* Ahb = Ah + (size_t)buf * (2 * C::A_FLOATS);          // (A_FLOATS floats = 2 * A_FLOATS halves per stage)
    _Float16* Alb = Ahb + BM * LDH;
    _Float16* Bhb = Bh + (size_t)buf * (2 * C::B_FLOATS);
    _Float16* Blb = Bhb + BN * LDH;
#pragma unroll
    for (int j = 0; j < C::A_LOADS; ++j) {
      const int idx = tid + C::NT * j, row = idx >> 3;
      const float4 v = aprod.finish(sl.arow[j], sl.acol, k0 + k4, sl.areg[j]);
      // the largest magnitude of the row's 32 steps: the 8 threads of a row are 8 consecutive lanes
      float am = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
      am = fmaxf(am, dpp_or<0xb1>(am, am));
      am = fmaxf(am, dpp_or<0x4e>(am, am));
      am = fmaxf(am, dpp_or<0x141>(am, am));  // row_half_mirror: lane i <-> 7 - i of its group of eight
      int se = 267 - (int)((__float_as_uint(am) >> 23) & 0xffu);  // am * 2^(se - 127) < 2^14
      se = se < 1 ? 1 : (se > 253 ? 253 : se);
      const float rs = __uint_as_float((unsigned)se << 23);
      const float a4[4] = {v.x * rs, v.y * rs, v.z * rs, v.w * rs};
      h4v hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hi[e] = (_Float16)a4[e];
        lo[e] = (_Float16)(a4[e] - (float)hi[e]);
      }
      *reinterpret_cast<h4v*>(Ahb + row * LDH + k4) = hi;
      *reinterpret_cast<h4v*>(Alb + row * LDH + k4) = lo;
      if ((tid & 7) == 0) Sc[buf * BM + row] = __uint_as_float((unsigned)(254 - se) << 23) * (1.0f / UGEMM_H3_WSCALE);
    }
#pragma unroll
    for (int j = 0; j < B_PIECES; ++j) {
      const int idx = tid + C::NT * j;
      const int n = idx >> 2, kp = (idx & 3) * 8;
      *reinterpret_cast<u32x4*>(Bhb + n * LDH + kp) = sl.bhr[j];
      *reinterpret_cast<u32x4*>(Blb + n * LDH + kp) = sl.blr[j];
    }
  };
  f32x4 acc[C::RT][C::CT];
  int live_rt = 0;
  auto multiply = [&](int buf) {
    if (live_rt > 0) {
      const _Float16* Ahb = Ah + (size_t)buf * (2 * C::A_FLOATS) + (wr + i) * LDH + 8 * q;
      const _Float16* Bhb = Bh + (size_t)buf * (2 * C::B_FLOATS) + (wc + i) * LDH + 8 * q;
      h8v bh[C::CT], bl[C::CT];
#pragma unroll
      for (int ct = 0; ct < C::CT; ++ct) {
        bh[ct] = *reinterpret_cast<const h8v*>(Bhb + ct * 16 * LDH);
        bl[ct] = *reinterpret_cast<const h8v*>(Bhb + BN * LDH + ct * 16 * LDH);
      }
#pragma unroll
      for (int rt = 0; rt < C::RT; ++rt) {
        if (rt < live_rt) {
          const h8v ah = *reinterpret_cast<const h8v*>(Ahb + rt * 16 * LDH);
          const h8v al = *reinterpret_cast<const h8v*>(Ahb + BM * LDH + rt * 16 * LDH);
          const float4 sc = ld4(Sc + buf * BM + wr + 16 * rt + 4 * q);  // the scales of this lane's four output rows
          const float scv[4] = {sc.x, sc.y, sc.z, sc.w};
#pragma unroll
          for (int ct = 0; ct < C::CT; ++ct) {
            f32x4 tmp = {0.f, 0.f, 0.f, 0.f};
            tmp = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[ct], tmp, 0, 0, 0);
            tmp = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[ct], tmp, 0, 0, 0);
            tmp = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[ct], tmp, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[rt][ct][r] = fmaf(tmp[r], scv[r], acc[rt][ct][r]);
          }
        }
      }
    }
  };

  Slot s0, s1;
  load_next(s0);
  load_next(s1);
  Chunk cur = chunk_at(u);
  for (;;) {
#pragma unroll
    for (int rt = 0; rt < C::RT; ++rt)
#pragma unroll
      for (int t = 0; t < C::CT; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    live_rt = (16 * cur.m_units - wr + 15) / 16;
    live_rt = live_rt < 0 ? 0 : (live_rt > C::RT ? C::RT : live_rt);
    UGEMM_TRACE_STAMP(0);
    store_tile(s0, 0, 0);
    load_next(s0);
    lds_barrier();
    UGEMM_TRACE_STAMP(1);
    for (int t = 0; t < nkp; t += 2) {
      if (t < 4) UGEMM_TRACE_STAMP(2 + 3 * t);
      multiply(0);
      if (t < 4) UGEMM_TRACE_STAMP(3 + 3 * t);
      store_tile(s1, 1, (t + 1) * BK);
      load_next(s1);
      if (t < 4) UGEMM_TRACE_STAMP(4 + 3 * t);
      lds_barrier();
      if (t < 4) UGEMM_TRACE_STAMP(5 + 3 * t);
      multiply(1);
      if (t + 2 < nkp) {
        store_tile(s0, 0, (t + 2) * BK);
        load_next(s0);
      }
      if (t < 4) UGEMM_TRACE_STAMP(6 + 3 * t);
      lds_barrier();
      if (t < 4) UGEMM_TRACE_STAMP(7 + 3 * t);
    }
    UGEMM_TRACE_STAMP(20);
    // ---- the chunk's output through LDS (a lane holds one column of four tiles), as the n-major fp32 kernel ---------------
    // (s0 / s1 now hold the next chunk's first two tiles, in flight)
    const int64_t e_r0 = cur.r0, e_rend = cur.rend;
    const int e_n0 = cur.n0;
    const int64_t un = cur.u + cur.m_units;
    const bool more = un < u_end;
    float* Cs = smem;
    constexpr int PIECES = BM * (BN / 4);
    constexpr int PPT = PIECES / C::NT;
    static_assert(PIECES % C::NT == 0, "epilogue pieces divide evenly");
    typename Epi::Pre pre[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const int idx = tid + C::NT * k;
      const int row = idx / (BN / 4), c4 = (idx - row * (BN / 4)) * 4;
      const int64_t r = e_r0 + row;
      const int c = e_n0 + c4;
      if (r < e_rend && c < d.N) pre[k] = epi.prefetch(r, c, d.N - c < 4 ? d.N - c : 4);
    }
#pragma unroll
    for (int rt = 0; rt < C::RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wr + 16 * rt + 4 * q + r;
#pragma unroll
        for (int t = 0; t < C::CT; ++t) Cs[row * C::LDC + wc + 16 * t + i] = acc[rt][t][r];
      }
    lds_barrier();
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const int idx = tid + C::NT * k;
      const int row = idx / (BN / 4), c4 = (idx - row * (BN / 4)) * 4;
      const int64_t r = e_r0 + row;
      const int c = e_n0 + c4;
      if (r < e_rend && c < d.N) {
        const int nv = d.N - c < 4 ? d.N - c : 4;
        epi(r, c, ld4(Cs + row * C::LDC + c4), nv, pre[k]);
      }
    }
    UGEMM_TRACE_STAMP(21);
    if (!more) break;
    lds_barrier();  // the tile in LDS has been read
    cur = chunk_at(un);
  }
}

#ifdef UGEMM_DEBUG_GRID
static int g_ugemm_grid_mode = 0;
#endif
// per-DEVICE caches (a process may drive several GPUs; function attributes and occupancy belong to the device current at
// the call): index = hipGetDevice(), guarded by atomics - the worst a race does is set the same value twice
constexpr int UGEMM_MAX_DEV = 64;
inline int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= UGEMM_MAX_DEV) dev = 0;
  return dev;
}
inline int device_cus() {
  static std::atomic<int> cus[UGEMM_MAX_DEV];
  const int dev = current_device();
  int c = cus[dev].load(std::memory_order_relaxed);
  if (c == 0) {
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
    cus[dev].store(c, std::memory_order_relaxed);
  }
  return c;
}

template <int BM, int BN, int WM, int WN, bool B_NMAJOR, class AProd, class Epi>
inline hipError_t launch(const Dims& d, const AProd& aprod, const float* B, const Epi& epi, hipStream_t st, hipEvent_t ev_start = nullptr,
                         hipEvent_t ev_stop = nullptr) {
  using C = Cfg<BM, BN, WM, WN, B_NMAJOR>;
  auto kern = gemm_kernel<BM, BN, WM, WN, B_NMAJOR, AProd, Epi>;
  const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
  const int dev = current_device();
  static std::atomic<bool> attr[UGEMM_MAX_DEV];  // one set per template instance and device
  if (lds > 64 * 1024 && !attr[dev].load(std::memory_order_acquire)) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr[dev].store(true, std::memory_order_release);
  }
  // resident workgroups per CU: asked from the runtime once per template instance (LDS, wave slots and the registers
  // the compiler actually used all bind: 53 KB of LDS -> 3 at <= 80 registers, 2 above)
  static std::atomic<int> per_cu_dev[UGEMM_MAX_DEV];
  int per_cu = per_cu_dev[dev].load(std::memory_order_relaxed);
  if (per_cu == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), C::NT, lds) != hipSuccess || nb < 1) {
      nb = (int)((160 * 1024) / lds);
      const int by_waves = 32 / (WM * WN);
      if (nb > by_waves) nb = by_waves;
      if (nb < 1) nb = 1;
    }
    per_cu = nb;
    per_cu_dev[dev].store(nb, std::memory_order_relaxed);
  }
  const int64_t slots = (int64_t)per_cu * device_cus();
  const int64_t nru = (d.R + 15) / 16;
  const int ncb = (d.N + BN - 1) / BN;
  // One full chunk per workgroup when everything fits at once; otherwise exactly the workgroups the GPU holds, each with
  // total / grid units +- 1.  Measured (tools/gemm_tile_ubench.hip, grid modes): at 102400 x 256 x 256 this ties one
  // workgroup per chunk (157 us both; 3200 chunks on 768 slots), at 12800 x 700 x 512 (800 chunks: 1.04 rounds) it wins
  // 108 vs 114 us; "whole rounds" grids (floor(chunks / slots) x slots workgroups of a chunk and a bit) lost at the
  // large shape (184 us: the second, short chunk of a workgroup pays the full k-loop latency).
  const int64_t chunks = ((nru + BM / 16 - 1) / (BM / 16)) * ncb;
  int64_t grid = chunks <= slots ? chunks : slots;
#ifdef UGEMM_DEBUG_GRID
  if (g_ugemm_grid_mode == 1) grid = chunks;
  if (g_ugemm_grid_mode == 2) grid = chunks <= slots ? chunks : slots;
  if (g_ugemm_grid_mode >= 16) grid = chunks <= slots ? chunks : (int64_t)g_ugemm_grid_mode * device_cus();
#endif
  if (ev_start != nullptr || ev_stop != nullptr)  // timed by the dispatch packet's own timestamps (ultr_prof.h)
    hipExtLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C::NT), (uint32_t)lds, st, ev_start, ev_stop, 0, d, aprod, B, epi);
  else hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C::NT), lds, st, d, aprod, B, epi);
  return hipGetLastError();
}

template <int BM, int BN, int WM, int WN, class AProd, class Epi>
inline hipError_t launch_h3(const Dims& d, const AProd& aprod, const _Float16* Bhi, const _Float16* Blo, const Epi& epi, hipStream_t st) {
  using C = Cfg<BM, BN, WM, WN, true>;
  auto kern = gemm_h3_kernel<BM, BN, WM, WN, AProd, Epi>;
  constexpr int stage_floats = C::STAGES * (C::A_FLOATS + C::B_FLOATS) + C::STAGES * BM;
  const size_t lds = (size_t)(C::C_FLOATS > stage_floats ? C::C_FLOATS : stage_floats) * sizeof(float);
  const int dev = current_device();
  static std::atomic<bool> attr[UGEMM_MAX_DEV];
  if (lds > 64 * 1024 && !attr[dev].load(std::memory_order_acquire)) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr[dev].store(true, std::memory_order_release);
  }
  static std::atomic<int> per_cu_dev[UGEMM_MAX_DEV];
  int per_cu = per_cu_dev[dev].load(std::memory_order_relaxed);
  if (per_cu == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), C::NT, lds) != hipSuccess || nb < 1) {
      nb = (int)((160 * 1024) / lds);
      const int by_waves = 32 / (WM * WN);
      if (nb > by_waves) nb = by_waves;
      if (nb < 1) nb = 1;
    }
    per_cu = nb;
    per_cu_dev[dev].store(nb, std::memory_order_relaxed);
  }
  const int64_t slots = (int64_t)per_cu * device_cus();
  const int64_t nru = (d.R + 15) / 16;
  const int ncb = (d.N + BN - 1) / BN;
  const int64_t chunks = ((nru + BM / 16 - 1) / (BM / 16)) * ncb;
  const int64_t grid = chunks <= slots ? chunks : slots;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C::NT), lds, st, d, aprod, Bhi, Blo, epi);
  return hipGetLastError();
}

template <class AProd, class Epi>
inline hipError_t run_h3(const Dims& d, const AProd& aprod, const _Float16* Bhi, const _Float16* Blo, const Epi& epi, hipStream_t st) {
  if (d.N > 64) return launch_h3<64, 128, 4, 2>(d, aprod, Bhi, Blo, epi, st);
  return launch_h3<64, 64, 4, 1>(d, aprod, Bhi, Blo, epi, st);
}

// C = epi(A' . B).  Tile choice (tools/gemm_tile_ubench.hip): 64 x 128 with 8 waves (a wave = one 16-row tile x 64
// columns, 3 workgroups = 24 waves per CU) for wide outputs, 64 x 64 with 4 waves for narrow ones: more, shorter waves
// per CU hide each other's staging, barriers and epilogues better than the classic 128 x 128 / 4-wave shape here
// (83 -> 90 TFLOP/s at 102400 x 256 x 256 before the persistent schedule), and RT = 1 makes short chunks cheap
template <bool B_NMAJOR, class AProd, class Epi>
inline hipError_t run(const Dims& d, const AProd& aprod, const float* B, const Epi& epi, hipStream_t st, hipEvent_t ev_start = nullptr,
                      hipEvent_t ev_stop = nullptr) {
  if (d.N > 64) return launch<64, 128, 4, 2, B_NMAJOR>(d, aprod, B, epi, st, ev_start, ev_stop);
  return launch<64, 64, 4, 1, B_NMAJOR>(d, aprod, B, epi, st, ev_start, ev_stop);
}

}  // namespace ugemm
