// GPU box: does v_mfma_f32_16x16x32_f16 keep fp16 SUBNORMAL inputs?  (the lo halves of the split-half products are subnormal for
// elements far below their row's maximum)   hipcc --offload-arch=gfx950 -O2 tools/mfma_f16_denorm_test.hip -o /tmp/t && /tmp/t
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float bval, float* out) {
  h8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)1.0f; b[e] = (_Float16)bval; }
  f4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
  float* d; hipMalloc(&d, 4);
  const float vals[] = {1.0f, 6.103515625e-05f /* 2^-14: smallest normal */, 3.0517578125e-05f /* 2^-15 */, 5.9604644775390625e-08f /* 2^-24: smallest subnormal */};
  for (float v : vals) {
    k<<<1, 64>>>(v, d);
    float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("b = %.10e  -> sum of 32 products = %.10e  (expected %.10e)\n", v, h, 32.0 * (double)(float)(_Float16)v);
  }
  return 0;
}
