// GPU box: how wide is the adder inside v_mfma_f32_16x16x32_f16?  One dot product of 32 fp16 x fp16 products in which ONE product is 2^20 and the
// other 31 are small (s x s); the exact sum needs more than 24 bits below the large product.  Printed: the instruction's result (C = 0), the same
// products summed in fp64, and the same products added one by one in fp32 (what a chain of exact fp32 fmas gives).
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_f16_accum_test.hip -o tools/bin/mfma_f16_accum_test && tools/bin/mfma_f16_accum_test
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const _Float16* a, const _Float16* b, float* out) {
  const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
  h8 av, bv;
  for (int e = 0; e < 8; ++e) {
    av[e] = a[i * 32 + 8 * q + e];   // A[row i][k = 8 q + e]
    bv[e] = b[i * 32 + 8 * q + e];   // B[k][col i] stored per column
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[(4 * q + r) * 16 + i] = c[r];  // D[row 4 q + r][col i]
}
int main() {
  const float smalls[] = {1.0625f, 1.0009765625f, 0.03125f, 3.0f, 0.333251953125f};
  for (float sm : smalls) {
    for (int bigpos : {0, 13, 31}) {
      _Float16 ha[16 * 32], hb[16 * 32];
      for (int r = 0; r < 16; ++r)
        for (int kk = 0; kk < 32; ++kk) {
          ha[r * 32 + kk] = (_Float16)(kk == bigpos ? 1024.0f : sm);
          hb[r * 32 + kk] = (_Float16)(kk == bigpos ? 1024.0f : sm);
        }
      _Float16 *da, *db;
      float* dout;
      hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dout, 256 * 4);
      hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
      k<<<1, 64>>>(da, db, dout);
      float out[256];
      hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
      double exact = 0; float seq = 0.f;
      for (int kk = 0; kk < 32; ++kk) {
        const double p = (double)(float)ha[kk] * (double)(float)hb[kk];
        exact += p; seq += (float)p;
      }
      printf("small %.10g big at k=%2d : mfma %.4f  exact %.6f  (fp32 of exact %.4f)  fp32 sequential %.4f   mfma - exact = %+.4f (ulp at 2^20 = 0.125)\n", sm, bigpos,
             out[0], exact, (float)exact, seq, out[0] - exact);
      hipFree(da); hipFree(db); hipFree(dout);
    }
  }
  return 0;
}
