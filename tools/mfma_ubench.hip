// pure MFMA issue-rate probe: cycles per v_mfma_f32_16x16x4_f32 with NACC independent accumulators, W waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(float* out, unsigned long long* cyc, int n) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// whole-chip sustained rate: `blocks` workgroups of 256 threads (one wave per SIMD each), wall clock only
template <int NACC>
void chip(int blocks, int n) {
  float* out; unsigned long long* cyc; hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&cyc, (size_t)blocks * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, cyc, n);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, cyc, n);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * n * NACC * 2048.0;
  unsigned long long* h = new unsigned long long[blocks]; hipMemcpy(h, cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < blocks; ++i) s += (double)h[i];
  printf("  chip: %d workgroups x 4 waves, NACC=%d, n=%d: wall %.1f us -> %.1f TFLOP/s; avg %.0f counter ticks per workgroup -> %.2f ticks/ns\n", blocks, NACC, n,
         ms * 1e3, flops / ms / 1e9, s / blocks, s / blocks / (ms * 1e6));
  delete[] h; hipFree(out); hipFree(cyc);
}
template <int NACC>
void run(int threads) {
  float* out; unsigned long long* cyc; hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
  const int n = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(160), dim3(threads), 0, 0, out, cyc, n);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<NACC>, dim3(160), dim3(threads), 0, 0, out, cyc, n);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  printf("  wall %.1f us ; ", ms * 1e3);
  unsigned long long h[160]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += (double)v;
  const double per_wave = s / 160 / (n * NACC);
  printf("NACC=%d threads=%4d (waves/SIMD %.1f): %.1f cycles per MFMA per wave -> %.1f cycles per MFMA per SIMD\n", NACC, threads,
         threads / 256.0, per_wave, per_wave / (threads / 256.0 < 1 ? 1 : threads / 256.0));
}
int main() {
  chip<4>(256, 20000); chip<4>(512, 20000); chip<4>(1024, 20000); chip<4>(2048, 20000); chip<4>(256, 200000);
  run<1>(256); run<2>(256); run<4>(256); run<4>(512); run<4>(1024); run<1>(1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  return 0;
}
