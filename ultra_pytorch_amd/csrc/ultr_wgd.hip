// ultr_wgd.hip — DIRECT weight gradients for small batches (reference: what loss.backward() leaves in .grad of every
// DNN.sequential parameter, base_algorithm.py:208-226 / DNN.py:41-55).
//
// dnn_wgrad_kernel (ultr_dnn.hip) cuts every dW_j into 64 x 64 tiles x row splits, writes one slab per split and needs a second
// launch (grad_reduce_kernel) to fold the slabs, the vector slabs and the sum-of-squares partials: at config 2 (2 560 rows) that
// is 11.8 + 4.9 us of a 48 us step for 0.5 GFLOP, and 6.6 MB of slab round trips.  With so few rows ONE workgroup can contract
// all of them for a small output tile, and then what it holds is the final gradient:
//
//   * a workgroup = 8 waves owns a 16 (m) x 32 (k) tile of dW_j = dz_j^T u_j; wave w takes the 4-row groups w, w + 8, ..
//     (the eight waves walk the same 32 rows together: every 128-byte line of dz / u they touch is shared with the
//     other tiles of the same XCD); per 4-row step a lane issues one 4-byte load (dz[n + q][m0 + i]) and one 8-byte load
//     (u[n + q][k0 + 2 j .. + 1]) feeding two v_mfma_f32_16x16x4_f32 - exact fp32, as everywhere in the weight gradients
//     (the contraction runs over rows: the split-half trick does not apply);
//   * a RING of D steps per wave keeps 2 D loads in flight (the operands were written by the previous launch with
//     write-through stores and come from the memory-side cache);
//   * epilogue: the eight partial tiles meet in LDS (fixed order), the layer-0 shortcut (BwdPlan::l0g) is applied, the tile goes
//     straight into the flat gradient and ONE sum-of-squares partial per workgroup into the slots the update launch sums;
//   * layer 0's d gamma / d beta need a column sum over ALL m-tiles: every tile leaves its 2 x 32 column partials as 16-byte
//     pieces [v0 v1 v2 | launch number] - ONE write-through store per piece, no fence, no flag word, no counter to reset - and
//     exits; one spare workgroup per k-tile polls the pieces until each carries this launch's number (a 16-byte aligned store
//     lands as a unit) and folds them in tile order;
//   * more spare workgroups fold the per-row-block vector slabs (LayerNorm gamma / beta of layers >= 1, the scorer) into final
//     gradients, and the last one folds the loss partials into the step tail (+ the early loss report) and zeroes the unused
//     sum-of-squares slots;
//   * every hidden layer gets its own group of XCDs (block ids go round-robin over the 8 XCDs) and, inside the group, every XCD
//     a rectangular sub-grid of the layer's tiles: an XCD's L2 pulls a (1 / gm) slice of dz_j and a (1 / gk) slice of u_j of ONE
//     layer (config 2: 18 MB from the fabric per launch; with every layer spread over all 8 XCDs it was 35 MB).
// Deterministic: every sum has a fixed order.  Taken by backward_impl for the fused small-batch step (operands ready-made in
// `saved`: BwdPlan::wg_prenorm) when ULTR_WGD=1 and the batch is small enough (ULTR_WGD_MAX_ROWS); everything else keeps the slab path.
// STATUS (round 4): correct (parity tests run it through the knob) but NOT faster at config 2 - 17.7 us against 12.3 + 5.0 us for the
// two slab launches: the 80-step loop is bound by the L2 -> CU stream (208 workgroups x 0.5 MB, every dz / u line wanted by 8 - 16
// workgroups) at ~9.5 us where the matrix cores need 4.3, and more loads in flight only slow it down (ring 6 / 12 / 24: 20.8 /
// 22.6 / 25.0 us; sixteen waves: 20.4).  Off by default; every measurement is in profiles/r04_cfg2_attempts.md.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <utility>

#include "../../include/ultr_hip.h"
#include "ultr_device.h"
#include "ultr_plan.h"
#include "ultr_prof.h"

#ifndef WGD_D
#define WGD_D 6  // steps in flight per wave (measured at config 2: 6 -> 20.8 us, 12 -> 22.6, 24 -> 25.0 before the rotation)
#endif

template <int N_, class F, int... I>
__device__ __forceinline__ void wgd_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N_, class F>
__device__ __forceinline__ void wgd_for(F&& f) {
  wgd_for_impl<N_>(f, std::make_integer_sequence<int, N_>{});
}

__device__ __forceinline__ f32x2 wgd_ld2(const Src& s, unsigned byte_off) {
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(s.rs, byte_off, 0, 0);
  return (f32x2){__uint_as_float(v.x), __uint_as_float(v.y)};
}
__device__ __forceinline__ uint32_t coh_ldu(const Src& s, unsigned byte_off) { return __builtin_amdgcn_raw_buffer_load_b32(s.rs, byte_off, 0, ULTR_SC1); }
__device__ __forceinline__ void coh_stu(const Src& s, unsigned byte_off, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b32(v, s.rs, byte_off, 0, ULTR_SC1); }

// fixed-order sum of one value per wave through LDS; every thread gets the total
template <int NW>
__device__ __forceinline__ float wgd_block_sum(float v, float* smw) {
  asm volatile("" : "+v"(v));  // the caller's product is rounded before the first cross-lane add (see grad_reduce_kernel)
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) smw[threadIdx.x >> 6] = v;
  lds_barrier();
  float t = smw[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) t += smw[w];
  lds_barrier();
  return t;
}

template <int NW>  // waves per workgroup: 8 or 16 (sixteen = four per SIMD: the steps of a wave are a chain of load -> LDS -> MFMA latencies)
__global__ __launch_bounds__(NW * 64) void dnn_wgd_kernel(DnnPlan p, BwdPlan bp, WgdPlan wp, const float* __restrict__ params,
                                                      const float* __restrict__ saved, float* __restrict__ ws,
                                                      float* __restrict__ grads, const float* __restrict__ loss_part,
                                                      int n_loss_part, int tail, int nsq, EarlyReport er) {
  constexpr int NT = NW * 64;
  __shared__ __attribute__((aligned(16))) float red[NW][16 * 32];  // the waves' scratch, then their partial tiles; reused by the folds
  __shared__ float bred[NW][16];
  __shared__ float sm8[NW];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* sumsq_part = ws + bp.sumsq_off;
  const int blk = (int)blockIdx.x;

  if (blk >= wp.ntile_blocks) {
    int sb = blk - wp.ntile_blocks;
    if (sb < wp.l[0].nkt) {
      // ---- layer-0 fold of k-tile sb: poll the pieces [v0 v1 v2 | seq] every m-tile's workgroup leaves, fold in tile order -----
      const WgdLayer wl = wp.l[0];
      const int kt = sb, npiece = wl.nmt * 22;
      const Src part = make_src(ws + wp.l0part_off, (int64_t)wl.nkt * wl.nmt * 96);
      float* fold = &red[0][0];  // [nmt][66]
      bool lost = false;
      for (int idx = tid; idx < npiece; idx += NT) {
        const int mtile = idx / 22, pc = idx - 22 * mtile;
        const unsigned off = (unsigned)(((kt * wl.nmt + mtile) * 24 + pc) * 16);
        float4 v = coh_ld4(part, off);
        for (int spin = 0; __float_as_uint(v.w) != wp.seq && spin < (1 << 22); ++spin) {
          __builtin_amdgcn_s_sleep(4);
          v = coh_ld4(part, off);
        }
        lost = lost || __float_as_uint(v.w) != wp.seq;
        fold[mtile * 66 + 3 * pc + 0] = v.x;
        fold[mtile * 66 + 3 * pc + 1] = v.y;
        fold[mtile * 66 + 3 * pc + 2] = v.z;
      }
      lds_barrier();
      float v = 0.f;
      if (tid < 64) {
        for (int t = 0; t < wl.nmt; ++t) v += fold[t * 66 + tid];
        if (lost) v = __uint_as_float(0x7fc00000u);  // a tile never reported (cannot happen within one launch): make it loud
        const int k = kt * 32 + (tid & 31);
        if (k < wl.K) grads[(tid < 32 ? p.off_lnw[0] : p.off_lnb[0]) + k] = v;
        else v = 0.f;
      }
      if (wave == 0) {
        float vv = v * v;
        asm volatile("" : "+v"(vv));
        const float sq = wave_sum(vv);
        if (lane == 0) sumsq_part[wp.ntile_blocks + kt] = sq;
      }
      return;
    }
    sb -= wp.l[0].nkt;
    if (sb < wp.nvec_blocks) {
      // ---- the nrb per-row-block vector slabs -> final gradients of 64 vector parameters: wave w sums slabs w, w + 8, .. in
      // chunks of 8 loads, the eight wave sums are added in wave order
      const int e = sb * 64 + lane;
      float part = 0.f;
      if (e < bp.vlen) {
        const float* src = ws + bp.vslab_off + e;
        for (int k = wave; k < bp.nrb && wave < 8; k += 64) {
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = (k + 8 * u < bp.nrb) ? src[(int64_t)(k + 8 * u) * bp.vlen] : 0.f;
          part += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
      }
      if (wave < 8) red[0][wave * 64 + lane] = part;
      lds_barrier();
      float g = 0.f;
      if (wave == 0 && e < bp.vlen) {
        g = ((((((red[0][lane] + red[0][64 + lane]) + red[0][128 + lane]) + red[0][192 + lane]) + red[0][256 + lane]) +
              red[0][320 + lane]) + red[0][384 + lane]) + red[0][448 + lane];
        // element e of a vector slab -> its place in the flat gradient (layer 0's gamma / beta come from the layer-0 fold
        // when the layer-0 shortcut is on: their slab entries are not written by the backward kernels)
        int64_t dst = -1;
        const int top = p.nl - 1;
        for (int j = 0; j < p.nl; ++j) {
          if (j == 0 && bp.l0g) continue;
          if (e >= bp.voff_g[j] && e < bp.voff_g[j] + p.K[j]) dst = p.off_lnw[j] + (e - bp.voff_g[j]);
          if (e >= bp.voff_b[j] && e < bp.voff_b[j] + p.K[j]) dst = p.off_lnb[j] + (e - bp.voff_b[j]);
        }
        if (e >= bp.voff_wk && e < bp.voff_wk + p.K[top]) dst = p.off_w[top] + (e - bp.voff_wk);
        if (e == bp.voff_bk) dst = p.off_b[top];
        if (dst >= 0) grads[dst] = g;
        else g = 0.f;
      }
      if (wave == 0) {
        float gg = g * g;
        asm volatile("" : "+v"(gg));
        const float sq = wave_sum(gg);
        if (lane == 0) sumsq_part[wp.ntile_blocks + wp.l[0].nkt + sb] = sq;
      }
      return;
    }
    // ---- last spare workgroup: loss partials -> step tail grads[P ..] (+ the early loss report), unused sum-of-squares slots = 0
    for (int k = wp.nsq_used + tid; k < nsq; k += NT) sumsq_part[k] = 0.f;
    const int grp = wave;  // waves 0..3 = the four cooperating groups of strided_sum; waves 4..7 only keep the barriers company
    float* out = grads + p.P;
    float head = 0.f;  // group 0, lanes 0..3: loss_sum, D, loss2_sum, D2 of the whole batch
    for (int t0 = 0; t0 < tail; t0 += 64) {
      const int t = t0 + lane;
      if (grp < 4) red[0][grp * 64 + lane] = (t < tail && loss_part != nullptr) ? strided_sum(loss_part + t, tail, n_loss_part, grp) : 0.f;
      lds_barrier();
      if (grp == 0 && t < tail && loss_part != nullptr) {
        const float v = ((red[0][lane] + red[0][64 + lane]) + red[0][128 + lane]) + red[0][192 + lane];
        out[t] = v;
        if (t0 == 0) head = v;
      }
      lds_barrier();
    }
    if (er.host != nullptr && grp == 0 && loss_part != nullptr) {
      // the same expressions as update_body (ultr_update.hip): the update kernel's later report carries the same bits
      const float loss_sum = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(head), 0));
      const float D = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(head), 1));
      const float loss2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(head), 2));
      const float D2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(head), 3));
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

  // ---- tile workgroups ---------------------------------------------------------------------------------------------------
  // block -> (XCD, index on that XCD) -> the layer whose XCD group holds this XCD -> tile of the XCD's sub-grid
  const int xcd = blk & 7, tt = blk >> 3;
  int j = 0;
  while (j + 1 < wp.nl1 && xcd >= wp.l[j + 1].x0) ++j;
  const WgdLayer wl = wp.l[j];
  const int xl = xcd - wl.x0;  // position inside the layer's group: (xl / gk, xl % gk) of the gm x gk XCD grid
  const int mt = (xl / wl.gk) * wl.pm + tt / wl.pk, kt = (xl % wl.gk) * wl.pk + tt % wl.pk;
  if (xl >= wl.gm * wl.gk || tt >= wl.pm * wl.pk || mt >= wl.nmt || kt >= wl.nkt) {
    // padding (block-uniform): an XCD with fewer tiles than the busiest one, or the ragged edge of a sub-grid - its slot must be 0
    if (tid == 0) sumsq_part[blk] = 0.f;
    return;
  }
  const int M = wl.M, K = wl.K, m0 = mt * 16, k0 = kt * 32;
  const int64_t N = bp.N;
  const int i = lane & 15, q = lane >> 4;
  const bool l0g = (j == 0) && bp.l0g != 0;
  // epilogue operands of the layer-0 shortcut, requested now (thread = one output (m, kk) of the tile)
  const int em = (tid >> 5) & 15, ek = tid & 31;  // (threads 512.. of a sixteen-wave workgroup mirror 0..511 and write nothing)
  const bool eok = tid < 512 && (m0 + em < M) && (k0 + ek < K);
  float w0 = 0.f, gam0 = 1.f, bet0 = 0.f;
  if (l0g && eok) {
    w0 = params[wl.off_w + (int64_t)(m0 + em) * K + k0 + ek];
    gam0 = params[p.off_lnw[0] + k0 + ek];
    bet0 = params[p.off_lnb[0] + k0 + ek];
  }

  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  float bsum = 0.f;
  // One step = 4 rows of the batch for this wave (row group 8 s + w of 4-row groups: rows 32 s + 4 w .. + 3).  The vector memory
  // path returns a quad of lanes per clock whatever the load width, so 4-byte / 8-byte operand loads in MFMA layout moved a
  // quarter / half of what the same instruction slots can (measured: 21 us for this kernel).  Instead ONE 16-byte load per step:
  // lanes 0..15 fetch the step's dz pieces (row L >> 2, columns m0 + 4 (L & 3) ..), lanes 16..47 its u pieces (row (L - 16) >> 3,
  // columns k0 + 4 ((L - 16) & 7) ..), and a wave-private LDS scratch turns the step into MFMA operand order:
  //   A[i][q] = dz[n + q][m0 + i] = scratch float q * 16 + i;   B[q][2 i + t] = u[n + q][k0 + 2 i + t] = scratch float 64 + q * 32 + 2 i + t
  // (two scratch slots per wave: step s is written while step s - 1 is read - same-wave LDS traffic is ordered, no barrier).
  const bool isA = lane < 16, isB = lane >= 16 && lane < 48;
  const int lrow = isA ? (lane >> 2) : ((lane - 16) >> 3);
  const int lcol = isA ? (m0 + 4 * (lane & 3)) : (k0 + 4 * ((lane - 16) & 7));
  const bool l_ok = isA ? (lcol < M) : (isB && lcol < K);   // (M and K are multiples of 4: a 16-byte piece is all in or all out)
  // per-lane 64-bit addresses (the two halves of the wave read different tensors: one buffer resource cannot describe both);
  // lanes / rows outside the tensors read the tensor's first bytes instead and are zeroed when staged
  const float* lbase = isA ? (ws + wl.dz_off) : (saved + wl.x_off);
  constexpr int RPS = 4 * NW;  // rows per step of the workgroup
  const int64_t lstride = RPS * (int64_t)(isA ? M : K);
  const int nsteps = (int)((N + RPS - 1) / RPS);
  // Every workgroup walks the 32-row steps in ROTATED order (its first step depends on its position inside the XCD): tiles that
  // march through the rows in lockstep hit the same few L2 channels at every moment (measured: deeper prefetch made the kernel
  // SLOWER, loads alone took twice as long as loads + compute).  An XCD's share of the operands (~3.5 MB at config 2) fits its
  // 4 MB L2, so the lines are still fetched from the fabric once.  The order is fixed per tile: results stay deterministic.
#ifndef WGD_ROT
#define WGD_ROT 0  // measured at config 2: the loads alone 45.8 -> 20.1 us, the whole kernel 21.9 -> 22.6 (no gain: kept off)
#endif
  const int r0 = WGD_ROT ? (int)(((unsigned)tt * 2654435761u >> 8) % (unsigned)(nsteps > 0 ? nsteps : 1)) : 0;
  int step_i = r0;
  int64_t nrow = RPS * (int64_t)r0 + 4 * wave + lrow;
  const float* lp = lbase + nrow * (isA ? M : K) + lcol;
  const float* lp_wrap = lbase + (int64_t)(4 * wave + lrow) * (isA ? M : K) + lcol;
  float4 ring[WGD_D];
  bool rok[WGD_D];
  auto issue = [&](float4& v, bool& ok) {
    ok = l_ok && nrow < N;
    v = ld4(ok ? lp : lbase);
    ++step_i;
    const bool wrap = step_i == nsteps;  // (wave-uniform)
    step_i = wrap ? 0 : step_i;
    nrow = wrap ? (int64_t)(4 * wave + lrow) : nrow + RPS;
    lp = wrap ? lp_wrap : lp + lstride;
  };
  float* scr = &red[wave][0];  // 2 slots x 256 floats = this wave's 512 (the tile partials land here only after the loop)
  // (branch-free: a select on the loaded value becomes control flow and the loads lose their counted waits; slot stride 256 floats
  // so that all 64 lanes store - lanes 48..63 hold zeros nobody reads)
  auto stage = [&](int slot, const float4& v, bool ok) {
    const uint32_t m = ok ? 0xffffffffu : 0u;
    st4(scr + slot * 256 + 4 * lane, make_float4(__uint_as_float(__float_as_uint(v.x) & m), __uint_as_float(__float_as_uint(v.y) & m),
                                                 __uint_as_float(__float_as_uint(v.z) & m), __uint_as_float(__float_as_uint(v.w) & m)));
  };
  auto consume = [&](int slot) {
    const float a = scr[slot * 256 + q * 16 + i];
    const float2 bq = *reinterpret_cast<const float2*>(scr + slot * 256 + 64 + q * 32 + 2 * i);
    bsum += a;
    acc0 = mfma16(a, bq.x, acc0);
    acc1 = mfma16(a, bq.y, acc1);
  };
  wgd_for<WGD_D>([&](auto I) { issue(ring[decltype(I)::value], rok[decltype(I)::value]); });
  // software pipeline over the scratch: step t is staged, then step t - 1 is consumed
  int s = 0;
  for (; s + WGD_D <= nsteps; s += WGD_D) {
    wgd_for<WGD_D>([&](auto I) {
      constexpr int U = decltype(I)::value;
      stage(U & 1, ring[U], rok[U]);
      issue(ring[U], rok[U]);  // step s + U + WGD_D (past the batch: a dummy line, zeroed when staged)
      if (s + U > 0) consume((U + 1) & 1);
    });
  }
  static_assert(WGD_D % 2 == 0, "the scratch slot of a step is its ring index's parity");
  wgd_for<WGD_D>([&](auto I) {
    constexpr int U = decltype(I)::value;
    if (s + U < nsteps) {
      stage(U & 1, ring[U], rok[U]);
      if (s + U > 0) consume((U + 1) & 1);
    }
  });
  if (nsteps > 0) consume((nsteps - 1) & 1);
  lds_barrier();  // every wave is done with its scratch before the partial tiles overwrite `red`

  // ---- the eight partial tiles meet in LDS: lane holds D_t[m = 4 q + r][col = i] = dW[m0 + 4 q + r][k0 + 2 i + t] -------------
#pragma unroll
  for (int r = 0; r < 4; ++r) *reinterpret_cast<float2*>(&red[wave][(4 * q + r) * 32 + 2 * i]) = make_float2(acc0[r], acc1[r]);
  {
    float sb = bsum;  // S[m0 + i] = sum over the four row groups q (lanes i, i + 16, i + 32, i + 48)
    sb += __shfl_xor(sb, 16, 64);
    sb += __shfl_xor(sb, 32, 64);
    if (q == 0) bred[wave][i] = sb;
  }
  lds_barrier();
  float G = 0.f, Sm = 0.f;
  {
    const int e = em * 32 + ek;
    G = red[0][e];
    Sm = bred[0][em];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      G += red[w][e];
      Sm += bred[w][em];
    }
  }
  float g = G;
  if (l0g) g = gam0 * G + bet0 * Sm;  // dW_0 = gamma o G + S (x) beta  (G = dz_0^T xhat_0; ultr_plan.h BwdPlan::l0g)
  if (!eok) g = 0.f;
  if (eok) grads[wl.off_w + (int64_t)(m0 + em) * K + k0 + ek] = g;
  float gb = 0.f;
  if (tid < 512 && kt == 0 && ek == 0 && m0 + em < M) {  // the bias gradient db_j[m] = sum_r dz_j[r][m], once per m-tile
    gb = Sm;
    grads[wl.off_b + m0 + em] = gb;
  }
  {
    float gg = g * g;
    asm volatile("" : "+v"(gg));
    float g2 = gb * gb;
    asm volatile("" : "+v"(g2));
    const float tot = wgd_block_sum<NW>(gg + g2, sm8);
    if (tid == 0) sumsq_part[blk] = tot;
  }
  if (!l0g) return;

  // ---- layer 0: column partials of d gamma_0 = sum_m W_0[m, :] o G[m, :] and d beta_0 = sum_m W_0[m, :] S[m] over this tile's rows,
  // left for the k-tile's fold workgroup as 22 pieces [v0 v1 v2 | seq]: one 16-byte write-through store each, then exit
  float* cs = &red[0][0];  // [2][16][32] (everyone is past its reads of `red`: wgd_block_sum ended with a barrier)
  if (tid < 512) {
    cs[em * 32 + ek] = eok ? w0 * G : 0.f;
    cs[512 + em * 32 + ek] = eok ? w0 * Sm : 0.f;
  }
  lds_barrier();
  float* colsum = &red[4][0];  // [66]
  if (tid < 64) {
    const float* src = cs + (tid >> 5) * 512 + (tid & 31);
    float a = 0.f;
#pragma unroll
    for (int m = 0; m < 16; ++m) a += src[m * 32];
    colsum[tid] = a;
  }
  if (tid == 64 || tid == 65) colsum[tid] = 0.f;
  lds_barrier();
  if (tid < 22) {
    const Src part = make_src(ws + wp.l0part_off, (int64_t)wl.nkt * wl.nmt * 96);
    coh_st4(part, (unsigned)(((kt * wl.nmt + mt) * 24 + tid) * 16),
            make_float4(colsum[3 * tid], colsum[3 * tid + 1], colsum[3 * tid + 2], __uint_as_float(wp.seq)));
  }
}

static std::atomic<uint32_t> g_wgd_seq{1u};

int ultr_wgd_launch(const DnnPlan& p, const BwdPlan& bp, const float* params, const float* saved, float* ws, float* grads,
                    const float* loss_part, int n_loss_part, int tail, int nsq, hipStream_t st) {
  if (p.nl < 2 || !bp.wg_prenorm || bp.lf_chunks > 0 || n_loss_part > 1024) return ULTR_E_UNSUPPORTED;
  WgdPlan wp;
  memset(&wp, 0, sizeof(wp));
  wp.nl1 = p.nl - 1;
  if (wp.nl1 > 8) return ULTR_E_UNSUPPORTED;
  // XCDs per layer (powers of two, sum <= 8): start with one each, keep doubling the group of the layer with the most tiles per XCD
  int nx[ULTR_MAXL];
  for (int j = 0; j < wp.nl1; ++j) {
    WgdLayer& l = wp.l[j];
    l.M = p.M[j];
    l.K = p.K[j];
    if (l.M % 4 != 0 || l.K % 4 != 0) return ULTR_E_UNSUPPORTED;
    l.nmt = (l.M + 15) / 16;
    l.nkt = (l.K + 31) / 32;
    nx[j] = 1;
  }
  for (int used = wp.nl1;;) {
    int best = -1;
    for (int j = 0; j < wp.nl1; ++j) {
      if (used + nx[j] > 8) continue;
      if (best < 0 || (int64_t)wp.l[j].nmt * wp.l[j].nkt * nx[best] > (int64_t)wp.l[best].nmt * wp.l[best].nkt * nx[j]) best = j;
    }
    if (best < 0) break;
    used += nx[best];
    nx[best] *= 2;
  }
  int x0 = 0, maxt = 0;
  for (int j = 0; j < wp.nl1; ++j) {
    WgdLayer& l = wp.l[j];
    // the gm x gk = nx[j] grid of XCDs that needs the fewest operand columns per XCD
    int best = -1, bestcols = 0;
    for (int gm = 1; gm <= nx[j]; gm *= 2) {
      const int gk = nx[j] / gm, pm = (l.nmt + gm - 1) / gm, pk = (l.nkt + gk - 1) / gk;
      const int cols = pm * 16 + pk * 32 + (pm * gm - l.nmt + pk * gk - l.nkt);  // (ties: the split with less padding)
      if (best < 0 || cols < bestcols) {
        best = gm;
        bestcols = cols;
      }
    }
    l.gm = best;
    l.gk = nx[j] / best;
    l.pm = (l.nmt + l.gm - 1) / l.gm;
    l.pk = (l.nkt + l.gk - 1) / l.gk;
    l.x0 = x0;
    x0 += nx[j];
    maxt = l.pm * l.pk > maxt ? l.pm * l.pk : maxt;
    l.dz_off = bp.dz_off[j];
    l.x_off = p.sv_x[j];
    l.off_w = p.off_w[j];
    l.off_b = p.off_b[j];
  }
  wp.tiles_per_xcd = maxt;
  wp.ntile_blocks = 8 * maxt;
  wp.nvec_blocks = (bp.vlen + 63) / 64;
  wp.nsq_used = wp.ntile_blocks + wp.l[0].nkt + wp.nvec_blocks;
  if (wp.nsq_used > nsq || wp.l[0].nmt * 66 > 8 * 512 || !bp.l0g) return ULTR_E_UNSUPPORTED;
  wp.l0part_off = bp.wgd_part_off;
  uint32_t seq = g_wgd_seq.fetch_add(1u);
  if (seq == 0u) seq = g_wgd_seq.fetch_add(1u);  // (0 is what a fresh, zeroed workspace holds)
  wp.seq = seq;
  UltrProfScope prof(ULTR_K_WGRAD, st);
  const dim3 grid((unsigned)(wp.ntile_blocks + wp.l[0].nkt + wp.nvec_blocks + 1));
#ifndef WGD_NW
#define WGD_NW 8  // (sixteen waves measured slower at config 2: 20.4 against 17.7 us)
#endif
  ULTR_LAUNCH(prof, (dnn_wgd_kernel<WGD_NW>), grid, dim3(WGD_NW * 64), 0, st, p, bp, wp, params, saved, ws, grads, loss_part, n_loss_part,
              tail, nsq, g_ultr_early);
  return (int)hipGetLastError();
}
