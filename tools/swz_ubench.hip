// Micro-benchmark (GPU box): the forward GEMM phase of dnn_fb_kernel (16-row A tile in LDS, W streamed from L2 by 8 waves,
// 32 output columns per wave) with
//   base : production GemmPipe<1, 2, D, 0> over the k-major copy WT [K][M] - buffer_load_dwordx2, 16 lanes x 8 B x 4 rows
//   swz  : a fragment-major ("pre-swizzled") copy - one buffer_load_dwordx4 per lane = 2 k-steps x 2 column tiles, a wave
//          instruction reads 1 KiB contiguous: [chunk of 32 columns][trip of 32 k][u = 0..3][lane][ka.c0 ka.c1 kb.c0 kb.c1]
//          with ka = 32 trip + 16 (u / 2) + 4 q + 2 (u % 2), kb = ka + 1, columns c0 + 2 i + {0, 1}
// Results are compared (same arithmetic, same order) and cycles per GEMM reported.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/swz_ubench.hip -o tools/bin/swz_ub && tools/bin/swz_ub [G]
#include "../ultra_pytorch_amd/csrc/ultr_dnn.hip"
#include <cstdio>
#include <vector>

template <int D, int MODE = 0>  // MODE 0: sched_barrier between fetch and consume (true depth D); 1: compiler's order; 2: no loads; 3: no MFMA
struct PipeSw {
  float4 b[D][4];
  unsigned of;
  int left;
  template <int S>
  __device__ __forceinline__ void fetch(const Src& W) {
    const bool ok = left > 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if constexpr (MODE == 2) b[S][u] = make_float4(1.f, 1.f, 1.f, 1.f);
      else b[S][u] = buf_ld4(W, ok ? (of + (unsigned)u * 1024u) : ULTR_OOB);
    }
    --left;
    of += 4096u;
  }
  __device__ __forceinline__ void begin(const Src& W, int chunk, int ntrips, bool valid, int lane) {
    of = ((unsigned)chunk * (unsigned)ntrips * 256u + (unsigned)lane) * 16u;
    left = valid ? ntrips : 0;
    if constexpr (D > 1) fetch<0>(W);
    if constexpr (D > 2) fetch<1>(W);
    if constexpr (D > 3) fetch<2>(W);
  }
  template <int S>
  __device__ __forceinline__ void consume(const float* __restrict__ ap, f32x4 (&acc)[2]) {
    const float4 a0 = ld4(ap), a1 = ld4(ap + 16);
    if constexpr (MODE == 3) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc[0][u] += b[S][u].x * a0.x + b[S][u].z * a1.x;
        acc[1][u] += b[S][u].y * a0.y + b[S][u].w * a1.y;
      }
      return;
    }
    acc[0] = mfma16(a0.x, b[S][0].x, acc[0]);
    acc[1] = mfma16(a0.x, b[S][0].y, acc[1]);
    acc[0] = mfma16(a0.y, b[S][0].z, acc[0]);
    acc[1] = mfma16(a0.y, b[S][0].w, acc[1]);
    acc[0] = mfma16(a0.z, b[S][1].x, acc[0]);
    acc[1] = mfma16(a0.z, b[S][1].y, acc[1]);
    acc[0] = mfma16(a0.w, b[S][1].z, acc[0]);
    acc[1] = mfma16(a0.w, b[S][1].w, acc[1]);
    acc[0] = mfma16(a1.x, b[S][2].x, acc[0]);
    acc[1] = mfma16(a1.x, b[S][2].y, acc[1]);
    acc[0] = mfma16(a1.y, b[S][2].z, acc[0]);
    acc[1] = mfma16(a1.y, b[S][2].w, acc[1]);
    acc[0] = mfma16(a1.z, b[S][3].x, acc[0]);
    acc[1] = mfma16(a1.z, b[S][3].y, acc[1]);
    acc[0] = mfma16(a1.w, b[S][3].z, acc[0]);
    acc[1] = mfma16(a1.w, b[S][3].w, acc[1]);
  }
  __device__ __forceinline__ void run(const float* __restrict__ As, int lda, const Src& W, int ntrips, f32x4 (&acc)[2], int lane) {
    const int i = lane & 15, q = lane >> 4;
    const float* ap = As + i * lda + 4 * q;
    auto step = [&](auto uc) {
      constexpr int U = decltype(uc)::value;
      fetch<(U + D - 1) % D>(W);
      if constexpr (MODE != 1) __builtin_amdgcn_sched_barrier(0);
      consume<U>(ap, acc);
      ap += 32;
    };
    int t = 0;
    for (; t + D <= ntrips; t += D) {
      step(std::integral_constant<int, 0>());
      step(std::integral_constant<int, 1>());
      if constexpr (D > 2) step(std::integral_constant<int, 2>());
      if constexpr (D > 3) step(std::integral_constant<int, 3>());
    }
    if (t < ntrips) { consume<0>(ap, acc); ap += 32; }
    if constexpr (D > 2) if (t + 1 < ntrips) { consume<1>(ap, acc); ap += 32; }
    if constexpr (D > 3) if (t + 2 < ntrips) { consume<2>(ap, acc); ap += 32; }
  }
};

template <int VARIANT, int D, int MODE = 0>
__global__ __launch_bounds__(512) void ub_kernel(const float* __restrict__ W, const float* __restrict__ Wsw, int Kc, int Mo,
                                                 float* __restrict__ out, unsigned long long* __restrict__ cyc, int reps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NW = 8;
  const int ld = fwd_ld(Kc > Mo ? Kc : Mo);
  float* X = smem;
  float* Y = smem + 16 * ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < 16 * ld; e += NW * 64) {
    const int c = e % ld;
    X[e] = (c < Kc) ? 0.001f * (float)((e * 7 + blockIdx.x) % 97) : 0.f;
  }
  __syncthreads();
  const int K32 = round_up(Kc, 32), ntrips = K32 / 32;
  const Src Wt = make_src(W, (int64_t)Kc * Mo);
  const Src Ws = make_src(Wsw, (int64_t)K32 * Mo);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; ++rep) {
    if constexpr (VARIANT == 0) {
      GemmPipe<1, 2, D, 0> pipe;
      pipe.begin(Wt, Mo, 0, Kc, wave * 32, wave * 32 < Mo, 0, lane);
      for (int chn = wave; chn * 32 < Mo; chn += NW) {
        f32x4 acc[1][2];
        for (int t = 0; t < 2; ++t) acc[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        pipe.run(X, ld, Wt, 0, Kc, 0, acc, lane);
        store_nn<1, 2>(acc, Y, ld, Mo, chn * 32, lane, false);
      }
    } else {
      PipeSw<D, MODE> pipe;
      pipe.begin(Ws, wave, ntrips, wave * 32 < Mo, lane);
      for (int chn = wave; chn * 32 < Mo; chn += NW) {
        f32x4 acc[1][2];
        for (int t = 0; t < 2; ++t) acc[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        pipe.run(X, ld, Ws, ntrips, acc[0], lane);
        store_nn<1, 2>(acc, Y, ld, Mo, chn * 32, lane, false);
      }
    }
    lds_barrier();
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[blockIdx.x] = (t1 - t0) / reps;
  if (blockIdx.x == 0)
    for (int e = tid; e < 16 * Mo; e += NW * 64) out[e] = Y[(e / Mo) * ld + (e % Mo)];
}

static int g_G = 256;
template <int V, int D, int MODE = 0>
double run(const char* name, const float* dW, const float* dWs, int Kc, int Mo, float* dout, unsigned long long* dcyc, std::vector<float>* res) {
  const int G = g_G, reps = 200;
  const size_t lds = (size_t)2 * 16 * fwd_ld(Kc > Mo ? Kc : Mo) * sizeof(float);
  hipFuncSetAttribute(reinterpret_cast<const void*>(ub_kernel<V, D, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((ub_kernel<V, D, MODE>), dim3(G), dim3(512), lds, 0, dW, dWs, Kc, Mo, dout, dcyc, reps);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((ub_kernel<V, D, MODE>), dim3(G), dim3(512), lds, 0, dW, dWs, Kc, Mo, dout, dcyc, reps);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> c(G);
  hipMemcpy(c.data(), dcyc, G * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : c) s += (double)v;
  res->resize((size_t)16 * Mo);
  hipMemcpy(res->data(), dout, res->size() * sizeof(float), hipMemcpyDeviceToHost);
  const double mfma = (double)16 * round_up(Kc, 32) * Mo / 1024.0 * 32.0 / 4.0;
  printf("%-28s K=%3d M=%3d G=%3d  wall %.2f us/GEMM  memtime avg %.0f  MFMA floor %.0f cyc  %.1f B/clk/CU\n", name, Kc,
         Mo, G, ms * 1e3 / reps, s / G, mfma, (double)Kc * Mo * 4 / (ms * 1e-3 / reps * 2.4e9));
  return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
  if (argc > 1) g_G = atoi(argv[1]);
  for (int Kc : {256, 136}) {
    const int Mo = 256, K32 = round_up(Kc, 32), ntrips = K32 / 32;
    std::vector<float> h((size_t)Kc * Mo), hs((size_t)K32 * Mo, 0.f);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.001f * (float)((i * 13) % 101) - 0.05f;
    for (int ch = 0; ch < Mo / 32; ++ch)
      for (int t = 0; t < ntrips; ++t)
        for (int u = 0; u < 4; ++u)
          for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 15, q = lane >> 4;
            const int ka = 32 * t + 16 * (u / 2) + 4 * q + 2 * (u % 2);
            for (int e = 0; e < 4; ++e) {
              const int k = ka + (e >> 1), c = ch * 32 + 2 * i + (e & 1);
              hs[((((size_t)ch * ntrips + t) * 4 + u) * 64 + lane) * 4 + e] = (k < Kc) ? h[(size_t)k * Mo + c] : 0.f;
            }
          }
    float *dW, *dWs, *dout;
    unsigned long long* dcyc;
    hipMalloc(&dW, h.size() * 4);
    hipMalloc(&dWs, hs.size() * 4);
    hipMalloc(&dout, 16 * Mo * 4);
    hipMalloc(&dcyc, 1024 * sizeof(unsigned long long));
    hipMemcpy(dW, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dWs, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> r0, r1;
    run<0, 2>("base dwordx2 D=2", dW, dWs, Kc, Mo, dout, dcyc, &r0);
    run<0, 3>("base dwordx2 D=3", dW, dWs, Kc, Mo, dout, dcyc, &r1);
    run<0, 4>("base dwordx2 D=4", dW, dWs, Kc, Mo, dout, dcyc, &r1);
    for (int rep = 0; rep < 1; ++rep) {
      run<1, 2>("swizzled dwordx4 D=2", dW, dWs, Kc, Mo, dout, dcyc, &r1);
      double md = 0;
      for (size_t i = 0; i < r0.size(); ++i) md = fmax(md, fabs((double)r0[i] - r1[i]));
      printf("   max |base - swizzled| = %.3e (same products, same order: expect 0)\n", md);
      run<1, 3>("swizzled dwordx4 D=3", dW, dWs, Kc, Mo, dout, dcyc, &r1);
      run<1, 4>("swizzled dwordx4 D=4", dW, dWs, Kc, Mo, dout, dcyc, &r1);
      run<1, 2, 1>("swizzled D=2 compiler order", dW, dWs, Kc, Mo, dout, dcyc, &r1);
      run<1, 2, 2>("swizzled D=2 NO LOADS", dW, dWs, Kc, Mo, dout, dcyc, &r1);
      run<1, 2, 3>("swizzled D=2 NO MFMA", dW, dWs, Kc, Mo, dout, dcyc, &r1);
      run<1, 3, 3>("swizzled D=3 NO MFMA", dW, dWs, Kc, Mo, dout, dcyc, &r1);
    }
    hipFree(dW), hipFree(dWs), hipFree(dout), hipFree(dcyc);
  }
  return 0;
}
