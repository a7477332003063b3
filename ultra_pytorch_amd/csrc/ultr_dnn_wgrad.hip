// ultr_dnn_wgrad.hip - weight gradients of the DNN ranking model's hidden Linears and the reduction of their slabs (reference
// base_algorithm.py:208-226: loss.backward() + clip_grad_norm_'s norm):
//   dnn_wgrad_kernel     dW = dz^T u with the contraction over the N rows; 64 x 64 output blocks x row splits, LayerNorm re-applied to
//                        the B operand on the fly, deterministic partial slabs (no atomics); v_mfma_f32_16x16x4_f32
//   dnn_wgrad_h3_kernel  the same on the fp16 matrix cores (split hi / lo operands) from a few thousand rows up
//   grad_reduce_kernel / grad_reduce_xchg_kernel   fixed-order slab reduction -> flat gradient + step tail + sum-of-squares partials
//                        (the second one exchanges its output with the data-parallel peers in the same launch);  grad_sumsq_kernel
// and their launchers.
#include "ultr_dnn_kernels.h"
#include "ultr_comm.h"

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


ULTR_TRACE_READER(ultr_trace_read_wgrad)

int ultr_launch_dnn_wgrad(UltrProfScope& prof, const DnnPlan& p, const BwdPlan& bp, bool av, bool h3, size_t wlds, dim3 wgrid, hipStream_t st,
                          const float* params, const float* features, int64_t n_docs, const int32_t* docids, int batch, int list_size,
                          const float* saved, float* ws, int l0_vec, float* grads, const float* lp, int nlp, int tail, const EarlyReport& er,
                          const CommDev& cd) {
  hipError_t e;
  if (h3) {
    e = set_lds(dnn_wgrad_h3_kernel, (size_t)WH_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    ULTR_LAUNCH(prof, dnn_wgrad_h3_kernel, wgrid, dim3(512), (size_t)WH_LDS_BYTES, st, p, bp, params, features, n_docs, docids, batch,
                list_size, saved, ws, grads, lp, nlp, tail, er, cd);
    return (int)hipGetLastError();
  }
  e = av ? set_lds(dnn_wgrad_kernel<true>, wlds) : set_lds(dnn_wgrad_kernel<false>, wlds);
  if (e != hipSuccess) return (int)e;
  if (av)
    ULTR_LAUNCH(prof, dnn_wgrad_kernel<true>, wgrid, dim3(256), wlds, st, p, bp, params, features, n_docs, docids, batch, list_size, saved, ws,
                l0_vec, grads, lp, nlp, tail, er, cd);
  else
    ULTR_LAUNCH(prof, dnn_wgrad_kernel<false>, wgrid, dim3(256), wlds, st, p, bp, params, features, n_docs, docids, batch, list_size, saved, ws,
                l0_vec, grads, lp, nlp, tail, er, cd);
  return (int)hipGetLastError();
}

// one thread per element (full_sum) while the big segments have at most 32 parts, the cooperating groups otherwise; level-2
// sum-of-squares partials only without the extra loss-fold workgroup (its index would be a level-2 slot): *nsq2_out = their count
int ultr_launch_grad_reduce(UltrProfScope& prof, const RedPlan& rp, const DnnPlan& p, const BwdPlan& bp, int tail, int nblk, int maxparts,
                            hipStream_t st, float* ws, float* grads, int* nsq2_out) {
  *nsq2_out = 0;
  if (maxparts <= 32) {
    float* s2 = bp.lf_chunks == 0 ? ws + ultr_sumsq2_off(p.P) : nullptr;
    ULTR_LAUNCH(prof, grad_reduce_kernel<true>, dim3((nblk + 3) / 4 + (bp.lf_chunks > 0 ? 1 : 0)), dim3(256), 0, st, rp, p.P, tail,
                (const float*)ws, (const float*)(ws + bp.lfold_off), bp.lf_chunks, grads, ws + bp.sumsq_off, nblk, s2);
    if (s2 != nullptr) *nsq2_out = (nblk + 3) / 4;
  } else {
    ULTR_LAUNCH(prof, grad_reduce_kernel<false>, dim3(nblk + (bp.lf_chunks > 0 ? 1 : 0)), dim3(256), 0, st, rp, p.P, tail,
                (const float*)ws, (const float*)(ws + bp.lfold_off), bp.lf_chunks, grads, ws + bp.sumsq_off, nblk, (float*)nullptr);
  }
  return (int)hipGetLastError();
}

// data-parallel step: the reduction launch exchanges its own output with the peers (ultr_train_step then skips the exchange kernel)
int ultr_launch_grad_reduce_xchg(UltrProfScope& prof, const RedPlan& rp, const DnnPlan& p, const BwdPlan& bp, int tail, int nblk, const CommDev& cd,
                                 const EarlyReport& er, hipStream_t st, float* ws, float* grads, int* nblocks_out) {
  const dim3 xg((unsigned)((p.P + tail + 255) / 256));
#define XCHG_LAUNCH(WW) \
  ULTR_LAUNCH(prof, grad_reduce_xchg_kernel<WW>, xg, dim3(256), 0, st, rp, p.P, tail, (const float*)ws, grads, ws + bp.sumsq_off, nblk, cd, er, ws + ultr_sumsq2_off(p.P))
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
  *nblocks_out = (int)xg.x;
  return (int)hipGetLastError();
}

// dW [M, K] = dY^T X for a caller outside the DNN (SetRank's d x d weight gradients): one-layer plan around the operands
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

extern "C" int ultr_grad_sumsq(float* grads, int64_t n_params, int32_t list_size, void* bwd_ws, void* stream) {
  if (!grads || !bwd_ws || n_params <= 0) return ULTR_E_BADARG;
  const int tail = (int)ultr_tail_len(list_size);
  const int nblk = (int)ultr_red_blocks(n_params, tail);
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3(nblk), dim3(64), 0, (hipStream_t)stream, n_params, (const float*)grads,
                     (float*)bwd_ws);  // sumsq partials live at offset 0 of bwd_ws
  return (int)hipGetLastError();
}
