// ultr_dnn_fwd.hip - the DNN ranking model's forward kernels (reference ultra/ranking_model/DNN.py:58-88, base_algorithm.py:118-154):
//   dnn_fwd_kernel    gather + [LayerNorm -> Linear -> act] x k + LayerNorm -> Linear(., 1), one launch; a workgroup owns R = 16 / 32
//                     document rows end to end: activations never leave LDS between layers, weights stream from L2 straight into MFMA
//                     B-fragments;
//   dnn_fwdw_kernel   the same with 17 .. 64 rows per workgroup behind ONE split-half weight stream (round 5);
// and their launchers (ultr_dnn_forward in ultr_dnn.hip plans, these launch).
#include "ultr_dnn_kernels.h"

// ------------------------------------------------------------------------------------------------
// Forward
// ------------------------------------------------------------------------------------------------
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


ULTR_TRACE_READER(ultr_trace_read_fwd)

int ultr_launch_dnn_fwd(UltrProfScope& prof, const DnnPlan& p, int R, int nw, bool av, bool q4, size_t lds, hipStream_t st, const float* params,
                        const float* features, int64_t n_docs, const int32_t* docids, int batch, int list_size, float* scores, float* saved,
                        const float* wt, int vm) {
  hipError_t e = hipSuccess;
  const int64_t N = (int64_t)batch * list_size;
  const dim3 grid((unsigned)((N + R - 1) / R));
#define LAUNCH_FWD(RR, NWW, VV)                                                                                     \
  do {                                                                                                              \
    e = set_lds(dnn_fwd_kernel<RR, NWW, VV>, lds);                                                                  \
    if (e != hipSuccess) return (int)e;                                                                             \
    ULTR_LAUNCH(prof, (dnn_fwd_kernel<RR, NWW, VV>), grid, dim3(NWW * 64), lds, st, p, params, features, n_docs,    \
                       docids, batch, list_size, scores, saved, wt, vm);                                           \
  } while (0)
#define LAUNCH_FWD2(RR, NWW) \
  do {                       \
    if (av) LAUNCH_FWD(RR, NWW, true); \
    else LAUNCH_FWD(RR, NWW, false);   \
  } while (0)
  if (q4) {
    e = set_lds(dnn_fwd_kernel<16, 8, true, true>, lds);
    if (e != hipSuccess) return (int)e;
    ULTR_LAUNCH(prof, (dnn_fwd_kernel<16, 8, true, true>), grid, dim3(512), lds, st, p, params, features, n_docs, docids,
                batch, list_size, scores, saved, wt, vm);
  } else if (R == 16 && nw == 4) LAUNCH_FWD2(16, 4);
  else if (R == 16 && nw == 16) LAUNCH_FWD2(16, 16);
  else if (R == 16) LAUNCH_FWD2(16, 8);
  else if (nw == 4) LAUNCH_FWD2(32, 4);
  else LAUNCH_FWD2(32, 8);
#undef LAUNCH_FWD2
#undef LAUNCH_FWD
  return (int)hipGetLastError();
}

int ultr_launch_dnn_fwdw(UltrProfScope& prof, const DnnPlan& p, const WidePlan& wp, size_t wlds, hipStream_t st, const float* features,
                         int64_t n_docs, const int32_t* docids, int batch, int list_size, float* scores, float* saved, const float* wt) {
  hipError_t e = hipSuccess;
  const int64_t N = (int64_t)batch * list_size;
  const int rt = (wp.R + 15) / 16;
  const dim3 wgrid((unsigned)((N + wp.R - 1) / wp.R));
#define LAUNCH_FWDW(RTT)                                                                                                        \
  do {                                                                                                                          \
    e = set_lds(dnn_fwdw_kernel<RTT>, wlds);                                                                                    \
    if (e != hipSuccess) return (int)e;                                                                                         \
    ULTR_LAUNCH(prof, (dnn_fwdw_kernel<RTT>), wgrid, dim3(1024), wlds, st, p, wp, features, n_docs, docids, batch,              \
                list_size, scores, saved, wt);                                                                                  \
  } while (0)
  if (rt == 2) LAUNCH_FWDW(2);
  else if (rt == 3) LAUNCH_FWDW(3);
  else LAUNCH_FWDW(4);
#undef LAUNCH_FWDW
  return (int)hipGetLastError();
}
