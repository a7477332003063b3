// GPU box: correctness + speed of the LDS-tiled fp32 GEMM core (ultr_gemm.h) at the throughput shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_tile_ubench.hip -o tools/bin/gemm_tile_ub && tools/bin/gemm_tile_ub
#include "../ultra_pytorch_amd/csrc/ultr_gemm.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

static float frand() { return (float)rand() / (float)RAND_MAX * 2.f - 1.f; }

template <bool NMAJ, int BM = 0, int BN = 0, int WM = 0, int WN = 0>
void bench(int64_t R, int K, int N) {
  std::vector<float> hA((size_t)R * K), hB((size_t)K * N), hbias(N);
  for (auto& v : hA) v = frand();
  for (auto& v : hB) v = frand() * 0.1f;
  for (auto& v : hbias) v = frand();
  float *dA, *dB, *dC, *dbias;
  (void)hipMalloc(&dA, hA.size() * 4);
  (void)hipMalloc(&dB, hB.size() * 4);
  (void)hipMalloc(&dC, (size_t)R * N * 4);
  (void)hipMalloc(&dbias, N * 4);
  (void)hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);  // NMAJ: interpreted as [N][K]
  (void)hipMemcpy(dbias, hbias.data(), N * 4, hipMemcpyHostToDevice);
  ugemm::Dims d{R, N, K, NMAJ ? K : N};
  ugemm::APlain ap{dA, R, K, K};
  ugemm::EBiasAct ep{dC, dbias, N, 1};
  auto go = [&]() {
    if constexpr (BM == 0) return ugemm::run<NMAJ>(d, ap, dB, ep, 0);
    else return ugemm::launch<BM, BN, WM, WN, NMAJ>(d, ap, dB, ep, 0);
  };
  hipError_t e = go();
  (void)hipDeviceSynchronize();
  if (e != hipSuccess || hipGetLastError() != hipSuccess) { printf("launch failed %d\n", (int)e); return; }
  std::vector<float> hC((size_t)R * N);
  (void)hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int t = 0; t < 4000; ++t) {
    const int64_t r = (t < 200) ? (R - 1 - t % 3) : rand() % R;
    const int n = (t < 200) ? (N - 1 - t % 5) : rand() % N;
    double s = hbias[n];
    for (int k = 0; k < K; ++k) s += (double)hA[r * K + k] * (NMAJ ? hB[(size_t)n * K + k] : hB[(size_t)k * N + n]);
    s = s > 0 ? s : 0;
    maxerr = fmax(maxerr, fabs(s - hC[r * N + n]));
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int reps = 20;
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) (void)go();
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
#ifdef UGEMM_DEBUG_CYC
  {
    static unsigned long long hc[4096], hs[4096];
    (void)hipMemcpyFromSymbol(hc, HIP_SYMBOL(ugemm::g_ugemm_cyc), sizeof(hc));
    (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(ugemm::g_ugemm_xcc), sizeof(hs));
    const int nb = (int)(((R + 127) / 128) * ((N + 127) / 128));
    double sum = 0; unsigned long long mn = ~0ull, mx = 0, smin = ~0ull, emax = 0;
    const int n = nb < 4096 ? nb : 4096;
    for (int b = 0; b < n; ++b) { sum += hc[b]; if (hc[b] < mn) mn = hc[b]; if (hc[b] > mx) mx = hc[b]; if (hs[b] < smin) smin = hs[b]; if (hs[b] + hc[b] > emax) emax = hs[b] + hc[b]; }
    printf("   main loop cycles per workgroup: avg %.0f min %llu max %llu; first start -> last end %llu cycles; per-wave MFMA issue floor %d; wall %.1f us => %.2f GHz\n",
           sum / n, mn, mx, emax - smin, (int)((K + 31) / 32) * 128 * 32, us, (double)(emax - smin) / us / 1e3);
  }
#endif
  printf("%3dx%3d %s R=%7ld K=%4d N=%4d  %8.1f us  %6.1f TFLOP/s (%.0f%% of 157.3)  max|err| %.2e\n", BM, BN, NMAJ ? "NT" : "NN", (long)R, K, N, us,
         2.0 * R * K * N / us / 1e6, 100.0 * 2.0 * R * K * N / us / 1e6 / 157.3, maxerr);
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dbias);
}

int main(int argc, char** argv) {
#ifdef UGEMM_DEBUG_GRID
  if (argc > 2) ugemm::g_ugemm_grid_mode = atoi(argv[2]);
#endif
  if (argc > 1 && argv[1][0] == 'c') {  // counter passes: two tile shapes at a long contraction
    bench<false, 64, 128, 4, 2>(16384, 4096, 512);
    bench<false, 256, 128, 4, 2>(16384, 4096, 512);
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'q') {  // config 4's first GEMM: tile shapes against the 800-chunks-on-768-slots imbalance
    bench<false, 64, 128, 4, 2>(12800, 700, 512);
    bench<false, 128, 128, 4, 2>(12800, 700, 512);
    bench<false, 128, 128, 2, 2>(12800, 700, 512);
    bench<false, 256, 128, 4, 2>(12800, 700, 512);
    bench<false, 64, 128, 4, 2>(12800, 512, 256);
    bench<false, 128, 128, 4, 2>(12800, 512, 256);
    bench<false, 64, 64, 4, 1>(12800, 512, 256);
    bench<false, 64, 128, 4, 2>(12800, 256, 128);
    bench<false, 64, 64, 4, 1>(12800, 256, 128);
    bench<false, 128, 64, 8, 1>(12800, 256, 128);
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'o') {  // balanced for 512 resident workgroups (2 per CU) and for 768 (3 per CU)
    bench<false, 64, 128, 4, 2>(8192, 4096, 512);
    bench<false, 64, 128, 4, 2>(16384, 704, 512);
    bench<false, 64, 128, 4, 2>(65536, 256, 512);
    bench<false, 64, 128, 4, 2>(12288, 4096, 512);
    bench<false, 64, 128, 4, 2>(24576, 704, 512);
    bench<false, 64, 128, 4, 2>(98304, 256, 512);
    bench<false, 64, 128, 4, 2>(12800, 700, 512);
    bench<false, 64, 128, 4, 2>(102400, 256, 256);
    bench<true, 64, 128, 4, 2>(102400, 256, 256);
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'b') {  // exactly one full chunk per resident workgroup (768 = 3 per CU): the steady-state loop, no tail
    bench<false, 64, 128, 4, 2>(49152, 4096, 128);
    bench<false, 64, 128, 4, 2>(12288, 4096, 512);
    bench<false, 64, 128, 4, 2>(24576, 4096, 512);  // two full chunks each
    bench<false, 64, 128, 4, 2>(12288, 704, 512);
    bench<false, 64, 128, 4, 2>(12288, 256, 512);
    bench<false, 64, 128, 4, 2>(24576, 256, 512);
    bench<false, 64, 128, 4, 2>(98304, 256, 512);   // eight full chunks each
    bench<false, 128, 128, 4, 2>(32768, 4096, 512); // 128 x 128: 1 per CU by LDS? (launch decides) - 256 x 4 = 1024 chunks
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'k') {  // long contraction: the k loop's own efficiency (prologue / epilogue amortised)
    bench<false, 64, 128, 4, 2>(16384, 4096, 512);
    bench<false, 64, 128, 2, 2>(16384, 4096, 512);
    bench<false, 128, 128, 4, 2>(16384, 4096, 512);
    bench<false, 128, 128, 2, 2>(16384, 4096, 512);
    bench<false, 256, 128, 4, 2>(16384, 4096, 512);
    bench<true, 64, 128, 4, 2>(16384, 4096, 512);
    bench<true, 128, 128, 2, 2>(16384, 4096, 512);
    bench<false, 64, 128, 4, 2>(16384, 1024, 512);
    bench<false, 128, 128, 2, 2>(16384, 1024, 512);
    return 0;
  }
  if (argc > 1) {  // one shape only (counter passes): tools/bin/gemm_tile_ub 1
    bench<false, 64, 128, 2, 2>(102400, 256, 256);
    bench<false, 64, 128, 4, 2>(102400, 256, 256);
    bench<true, 64, 128, 4, 2>(102400, 256, 256);
    bench<false, 64, 64, 4, 1>(102400, 256, 64);
    bench<false, 64, 128, 4, 2>(12800, 700, 512);
    bench<false, 64, 128, 2, 2>(12800, 700, 512);
    bench<false, 128, 128, 2, 2>(102400, 256, 256);
    bench<false, 128, 128, 4, 2>(102400, 256, 256);
    bench<false, 256, 128, 4, 2>(102400, 256, 256);
    bench<true, 64, 128, 2, 2>(102400, 256, 256);
    bench<true, 128, 128, 4, 2>(102400, 256, 256);
    bench<false, 64, 64, 4, 1>(102400, 256, 64);
    bench<false, 128, 64, 8, 1>(102400, 256, 64);
    bench<true, 64, 64, 4, 1>(102400, 256, 64);
    bench<true, 128, 64, 8, 1>(102400, 256, 64);
    bench<false, 64, 128, 2, 2>(102400, 64, 256);
    bench<false, 128, 128, 4, 2>(102400, 64, 256);
    return 0;
  }
  bench<false>(12800, 700, 512);
  bench<false>(12800, 512, 256);
  bench<false>(12800, 256, 128);
  bench<false>(10240, 136, 512);
  bench<false>(81920, 136, 256);
  bench<false>(81920, 256, 256);
  bench<true>(102400, 256, 256);
  bench<true>(102400, 256, 64);
  bench<true>(102400, 64, 256);
  bench<true>(102400, 220, 64);
  bench<false>(102400, 256, 256);
  bench<false>(102400, 64, 256);
  bench<false>(102400, 256, 64);
  bench<false, 64, 128, 2, 2>(12800, 700, 512);
  bench<false, 64, 128, 2, 2>(12800, 512, 256);
  bench<false, 64, 128, 2, 2>(102400, 256, 256);
  bench<true, 64, 128, 2, 2>(102400, 256, 256);
  bench<false, 64, 128, 2, 2>(102400, 64, 256);
  bench<false, 64, 64, 4, 1>(102400, 256, 64);
  bench<false, 64, 64, 4, 1>(12800, 256, 128);
  bench<false, 128, 64, 4, 1>(12800, 256, 128);
  bench<false, 64, 128, 2, 2>(12800, 256, 128);
  bench<false>(1000, 100, 36);
  bench<true>(1000, 100, 36);
  return 0;
}
