// ultr_dnn_bwd.hip - the row-local half of the DNN ranking model's backward (the autograd backward of DNN.build, reference
// base_algorithm.py:208-226): dgrad chain + LayerNorm backward + act', one launch per step; emits dz_j to HBM for the weight gradients
// and per-row-block partial sums for every vector parameter (LayerNorm gamma / beta, the M = 1 scorer).
//   dnn_bwd_kernel   any widths;  dnn_bwd2_kernel  aligned shapes, K_j <= 512 (the fast schedule);  dnn_bwdw_kernel  17 .. 48 rows per
//   workgroup behind one split-half weight stream (round 5) - and their launchers.
#include "ultr_dnn_kernels.h"

// ------------------------------------------------------------------------------------------------
// Backward, row-local half
// ------------------------------------------------------------------------------------------------

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


ULTR_TRACE_READER(ultr_trace_read_bwd)

int ultr_launch_dnn_bwd(UltrProfScope& prof, const DnnPlan& p, const BwdPlan& bp, int nw, bool av, size_t lds, hipStream_t st, const float* params,
                        const float* features, int64_t n_docs, const int32_t* docids, int batch, int list_size, const float* saved,
                        const float* dscores, float* ws, int vm, const FusedSoftmax& fl) {
  hipError_t e = hipSuccess;
#define LAUNCH_BWD(RR, NWW, VV)                                                                                        \
  do {                                                                                                                 \
    e = set_lds(dnn_bwd_kernel<RR, NWW, VV>, lds);                                                                     \
    if (e != hipSuccess) return (int)e;                                                                                \
    ULTR_LAUNCH(prof, (dnn_bwd_kernel<RR, NWW, VV>), dim3(bp.nrb), dim3(NWW * 64), lds, st, p, bp, params, features,   \
                       n_docs, docids, batch, list_size, saved, dscores, ws, vm, fl);                                 \
  } while (0)
#define LAUNCH_BWD2(RR, NWW) \
  do {                       \
    if (av) LAUNCH_BWD(RR, NWW, true); \
    else LAUNCH_BWD(RR, NWW, false);   \
  } while (0)
  if (bp.rblk == 16 && nw == 4) LAUNCH_BWD2(16, 4);
  else if (bp.rblk == 16 && nw == 16) LAUNCH_BWD2(16, 16);
  else if (bp.rblk == 16) LAUNCH_BWD2(16, 8);
  else if (nw == 4) LAUNCH_BWD2(32, 4);
  else LAUNCH_BWD2(32, 8);
#undef LAUNCH_BWD2
#undef LAUNCH_BWD
  return (int)hipGetLastError();
}

int ultr_launch_dnn_bwd2(UltrProfScope& prof, const DnnPlan& p, const BwdPlan& bp, size_t lds2, hipStream_t st, const float* params,
                         const float* features, int64_t n_docs, const int32_t* docids, int batch, int list_size, const float* saved,
                         const float* dscores, float* ws, const FusedSoftmax& fl, const float* wt) {
  hipError_t e = hipSuccess;
#define LAUNCH_BWDV2(RR, XX)                                                                                           \
  do {                                                                                                                 \
    e = set_lds(dnn_bwd2_kernel<RR, 8, XX>, lds2);                                                                     \
    if (e != hipSuccess) return (int)e;                                                                                \
    ULTR_LAUNCH(prof, (dnn_bwd2_kernel<RR, 8, XX>), dim3(bp.nrb), dim3(512), lds2, st, p, bp, params, features,        \
                       n_docs, docids, batch, list_size, saved, dscores, ws, fl, wt);                                 \
  } while (0)
  const int xc = p.maxdim <= 256 ? 1 : 2;
  if (bp.rblk == 16 && xc == 1) LAUNCH_BWDV2(16, 1);
  else if (bp.rblk == 16) LAUNCH_BWDV2(16, 2);
  else if (xc == 1) LAUNCH_BWDV2(32, 1);
  else LAUNCH_BWDV2(32, 2);
#undef LAUNCH_BWDV2
  return (int)hipGetLastError();
}

int ultr_launch_dnn_bwdw(UltrProfScope& prof, const DnnPlan& p, const BwdPlan& bp, const WideBwd& wb, size_t wblds, hipStream_t st,
                         const float* saved, const float* dscores, float* ws, const float* wt) {
  hipError_t e = hipSuccess;
#define LAUNCH_BWDW(RTT)                                                                                                      \
  do {                                                                                                                        \
    e = set_lds(dnn_bwdw_kernel<RTT>, wblds);                                                                                 \
    if (e != hipSuccess) return (int)e;                                                                                       \
    ULTR_LAUNCH(prof, (dnn_bwdw_kernel<RTT>), dim3(bp.nrb), dim3(1024), wblds, st, p, bp, wb, saved, dscores, ws, wt);        \
  } while (0)
  if (wb.R <= 32) LAUNCH_BWDW(2);
  else if (wb.R <= 48) LAUNCH_BWDW(3);
  else LAUNCH_BWDW(4);
#undef LAUNCH_BWDW
  return (int)hipGetLastError();
}
