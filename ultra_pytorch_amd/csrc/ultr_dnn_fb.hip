// ultr_dnn_fb.hip - forward + NA / IPW loss + row-local backward of the DNN ranking model in ONE launch for small batches
// (dnn_fb_kernel: BASELINE config 2's dominant kernel) and its launcher.  Reference: DNN.py:58-88, base_algorithm.py:118-154, 309-330,
// ipw_rank.py:102-182.
#include "ultr_dnn_kernels.h"

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


ULTR_TRACE_READER(ultr_trace_read_fb)

int ultr_launch_dnn_fb(UltrProfScope& prof, const DnnPlan& p, const BwdPlan& bp, size_t lds, int64_t nblk, hipStream_t st, const float* params,
                       const float* wt, const float* features, int64_t n_docs, const int32_t* docids, int batch, int L, int lpb, float* scores,
                       float* saved, float* ws, const FusedSoftmax& fl, const FbPlan& fp) {
  hipError_t e = hipSuccess;
#define LAUNCH_FB(XX, HH)                                                                                                      \
  do {                                                                                                                         \
    e = set_lds(dnn_fb_kernel<XX, HH>, lds);                                                                                   \
    if (e != hipSuccess) return (int)e;                                                                                        \
    ULTR_LAUNCH(prof, (dnn_fb_kernel<XX, HH>), dim3((unsigned)nblk), dim3(512), lds, st, p, bp, params, wt, features, n_docs,  \
                docids, batch, L, lpb, scores, saved, ws, fl, fp);                                                             \
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
  return (int)hipGetLastError();
}
