// Micro-benchmark of the per-workgroup GEMM core (GPU box): 160 workgroups x 512 threads, A tile [16 x K] in LDS,
// W [M x K'] streamed from L2, cycles per GEMM via s_memtime.   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_ubench.hip -o /tmp/ub && /tmp/ub
#include "../ultra_pytorch_amd/csrc/ultr_dnn.hip"
#include <cstdio>
#include <vector>

template <int VARIANT, int NW>
__global__ __launch_bounds__(NW * 64) void ub_kernel(const float* __restrict__ W, int Kc /*contraction*/, int Mo /*outputs*/,
                                                     float* __restrict__ out, unsigned long long* __restrict__ cyc, int reps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ld = fwd_ld(Kc > Mo ? Kc : Mo);
  float* X = smem;
  float* Y = smem + 16 * ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < 16 * ld; e += NW * 64) X[e] = 0.001f * (float)((e * 7 + blockIdx.x) % 97);
  __syncthreads();
  const Src Wt = make_src(W, (int64_t)Kc * Mo);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; ++rep) {
    const int nch = (Mo + 63) >> 6;
    int ksplit = 1;
    if (VARIANT == 0) while (ksplit * 2 * nch <= NW) ksplit *= 2;
    if (VARIANT == 0 || VARIANT == 1) {
      // 0: 64-col chunks x k-split (production); 1: same but only first nch waves work (no split)
      const int klen = round_up((Kc + ksplit - 1) / ksplit, 16);
      const bool has = wave < nch * ksplit;
      const int ch = wave % nch, ks = wave / nch;
      f32x4 acc[1][4];
      for (int t = 0; t < 4; ++t) acc[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (has) {
        const int kb = ks * klen, ke = (kb + klen < Kc) ? (kb + klen) : Kc;
        if (kb < ke) gemm_nn<1, 4, true>(X, ld, Wt, Mo, kb, ke, ch * 64, acc, lane);
      }
      for (int r = 0; r < ksplit; ++r) {
        if (has && ks == r) store_nn<1, 4>(acc, Y, ld, Mo, ch * 64, lane, r > 0);
        __syncthreads();
      }
    } else if (VARIANT == 3) {
      // 32-column chunks, whole contraction per wave, depth-1 pipeline (gemm_nn)
      for (int chn = wave; chn * 32 < Mo; chn += NW) {
        f32x4 acc[1][2];
        for (int t = 0; t < 2; ++t) acc[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        gemm_nn<1, 2, true>(X, ld, Wt, Mo, 0, Kc, chn * 32, acc, lane);
        store_nn<1, 2>(acc, Y, ld, Mo, chn * 32, lane, false);
      }
      __syncthreads();
    } else if (VARIANT == 4 || VARIANT == 5 || VARIANT == 6 || VARIANT == 7 || VARIANT == 8 || VARIANT == 9 || VARIANT == 10) {
      // GemmPipe depth 4; 5: next repetition's first trips issued before the epilogue + barrier; 6: rotated trip order
      GemmPipe<1, 2, 4, (VARIANT == 9) ? 2 : (VARIANT == 10) ? 3 : (VARIANT >= 7) ? 1 : 0> pipe;
      const int rot = (VARIANT == 6) ? (int)blockIdx.x : 0;
      if ((VARIANT != 5 && VARIANT != 8) || rep == 0) pipe.begin(Wt, Mo, 0, Kc, wave * 32, wave * 32 < Mo, rot, lane);
      for (int chn = wave; chn * 32 < Mo; chn += NW) {
        f32x4 acc[1][2];
        for (int t = 0; t < 2; ++t) acc[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        pipe.run(X, ld, Wt, 0, Kc, rot, acc, lane);
        if (VARIANT == 5 || VARIANT == 8) pipe.begin(Wt, Mo, 0, Kc, wave * 32, wave * 32 < Mo, rot, lane);
        store_nn<1, 2>(acc, Y, ld, Mo, chn * 32, lane, false);
      }
      lds_barrier();
    } else if (VARIANT == 2) {
      // NT form on W [Mo][Kc] (16 rows x 64 B per load instruction), CT = Mo / (16*NW) col tiles per wave
      const int K16 = round_up(Kc, 16);
      if (Mo >= 32 * NW) {
        for (int chn = wave; chn * 32 < Mo; chn += NW) gemm_nt_chunk<1, 2, true>(X, ld, Kc, K16, Wt, W, Mo, chn * 32, 1, Y, ld, nullptr, 16, lane);
      } else {
        for (int chn = wave; chn * 16 < Mo; chn += NW) gemm_nt_chunk<1, 1, true>(X, ld, Kc, K16, Wt, W, Mo, chn * 16, 1, Y, ld, nullptr, 16, lane);
      }
      __syncthreads();
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[blockIdx.x] = (t1 - t0) / reps;
  if (tid < 16) out[blockIdx.x * 16 + tid] = Y[tid * ld + tid];
}

static int g_G = 160;
template <int V, int NW>
void run(const char* name, const float* dW, int Kc, int Mo, float* dout, unsigned long long* dcyc) {
  const int G = g_G, reps = 200;
  const size_t lds = (size_t)2 * 16 * fwd_ld(Kc > Mo ? Kc : Mo) * sizeof(float);
  hipFuncSetAttribute(reinterpret_cast<const void*>(ub_kernel<V, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((ub_kernel<V, NW>), dim3(G), dim3(NW * 64), lds, 0, dW, Kc, Mo, dout, dcyc, reps);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((ub_kernel<V, NW>), dim3(G), dim3(NW * 64), lds, 0, dW, Kc, Mo, dout, dcyc, reps);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double wall_ns = ms * 1e6 / reps;
  std::vector<unsigned long long> c(G);
  hipMemcpy(c.data(), dcyc, G * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double s = 0, mx = 0;
  for (auto v : c) { s += (double)v; if ((double)v > mx) mx = (double)v; }
  const double mfma_cycles = (double)16 * Kc * Mo / 1024.0 * 32.0 / 4.0;  // per SIMD
  printf("%-34s K=%3d M=%3d  wall %.2f us/GEMM  memtime avg %.0f  (MFMA bound %.0f cyc = %.2f us @2.2GHz => %.2fx)\n", name, Kc, Mo,
         wall_ns / 1e3, s / G, mfma_cycles, mfma_cycles / 2200.0, wall_ns / 1e3 / (mfma_cycles / 2200.0));
}

int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  if (argc > 2) g_G = atoi(argv[2]);
  const int KM = 256 * 256;
  std::vector<float> h(KM);
  for (int i = 0; i < KM; ++i) h[i] = 0.001f * (float)(i % 101);
  float *dW, *dout; unsigned long long* dcyc;
  hipMalloc(&dW, KM * sizeof(float)); hipMalloc(&dout, 160 * 16 * sizeof(float)); hipMalloc(&dcyc, 160 * sizeof(unsigned long long));
  hipMemcpy(dW, h.data(), KM * sizeof(float), hipMemcpyHostToDevice);
  if (only < 0 || only == 0) run<0, 8>("nn4 ksplit NW=8", dW, 256, 256, dout, dcyc);
  if (only < 0 || only == 1) run<0, 16>("nn4 ksplit NW=16", dW, 256, 256, dout, dcyc);
  if (only < 0 || only == 2) run<1, 4>("nn4 nosplit NW=4", dW, 256, 256, dout, dcyc);
  if (only < 0 || only == 3) run<2, 8>("nt CT=2 NW=8", dW, 256, 256, dout, dcyc);
  if (only < 0 || only == 4) run<2, 16>("nt CT=1 NW=16", dW, 256, 256, dout, dcyc);
  if (only < 0 || only == 10) run<3, 8>("nn CT=2 nosplit depth1 NW=8", dW, 256, 256, dout, dcyc);
  if (only < 0 || only == 11) run<4, 8>("pipe CT=2 D=4 NW=8", dW, 256, 256, dout, dcyc);
  if (only < 0 || only == 12) run<5, 8>("pipe CT=2 D=4 cross-phase NW=8", dW, 256, 256, dout, dcyc);
  if (only < 0 || only == 17) run<7, 8>("pipe CT=2 D=4 interleaved NW=8", dW, 256, 256, dout, dcyc);
  if (only < 0 || only == 18) run<8, 8>("pipe CT=2 D=4 interleaved cross-phase", dW, 256, 256, dout, dcyc);
  if (only < 0 || only == 19) run<7, 8>("pipe CT=2 D=4 interleaved NW=8", dW, 136, 256, dout, dcyc);
  if (only < 0 || only == 20) run<8, 8>("pipe CT=2 D=4 interleaved cross-phase", dW, 136, 256, dout, dcyc);
  if (only < 0 || only == 21) run<9, 8>("pipe CT=2 D=4 NO LOADS", dW, 256, 256, dout, dcyc);
  if (only < 0 || only == 22) run<10, 8>("pipe CT=2 D=4 NO MFMA", dW, 256, 256, dout, dcyc);
  if (only < 0 || only == 15) run<6, 8>("pipe CT=2 D=4 rotated NW=8", dW, 256, 256, dout, dcyc);
  if (only < 0 || only == 16) run<6, 8>("pipe CT=2 D=4 rotated NW=8", dW, 136, 256, dout, dcyc);
  if (only < 0 || only == 13) run<3, 8>("nn CT=2 nosplit depth1 NW=8", dW, 136, 256, dout, dcyc);
  if (only < 0 || only == 14) run<5, 8>("pipe CT=2 D=4 cross-phase NW=8", dW, 136, 256, dout, dcyc);
  if (only < 0 || only == 5) run<0, 8>("nn4 ksplit NW=8", dW, 136, 256, dout, dcyc);
  if (only < 0 || only == 6) run<2, 8>("nt CT=2 NW=8", dW, 136, 256, dout, dcyc);
  return 0;
}
