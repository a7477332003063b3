// GPU box: does v_pk_mul_f32 with op_sel:[0,1] (the HIGH half of src1 broadcast to both lanes of the packed multiply) ever
// read that half as zero while ANOTHER wave of the same SIMD is executing v_mfma_f32_16x16x32_f16?
// Round 3's intermittent wrong result (profiles/r04_h3_rootcause.md): the split-half dgrad epilogue of dnn_bwd2_kernel scaled
// its accumulators by per-row factors read with one ds_read_b128; hipcc multiplies rows 4q+1 of the two column tiles with
// `v_pk_mul_f32 v[a:b], v[a:b], v[o:o+1] op_sel:[0,1]` - and in lanes 48..63 the LOW lane of that instruction came out as
// (value x 0) in some launches, only while other waves of the workgroup were still inside their MFMA loops.
// Half of the waves of a workgroup (waves 0..3 = one per SIMD) run MFMA chains, the other half (waves 4..7, the SIMD
// partners) run the load + packed multiply and check it.
//   hipcc --offload-arch=gfx950 -O3 tools/pkmul_coexec_test.hip -o tools/bin/pkmul_coexec_test && tools/bin/pkmul_coexec_test
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define CK(x)                                        \
  do {                                               \
    hipError_t e_ = (x);                             \
    if (e_ != hipSuccess) {                          \
      printf("%s: %s\n", #x, hipGetErrorString(e_)); \
      exit(1);                                       \
    }                                                \
  } while (0)

// One checked instruction per CASE: text, and what its low / high lane must produce from p = (p0, p1), o = (o0, o1), c = (c0, c1)
#define CASES(X)                                                                                             \
  X(0, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]", p[0] * o[1], p[1] * o[1])                                     \
  X(1, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]", p[0] * o[0], p[1] * o[0])                                  \
  X(2, "v_pk_mul_f32 %0, %1, %2", p[0] * o[0], p[1] * o[1])                                                  \
  X(3, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]", p[1] * o[0], p[1] * o[1])                                     \
  X(4, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]", p[0] * o[1], p[1] * o[0])                     \
  X(5, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1]", p[0] + o[1], p[1] + o[1])                                     \
  X(6, "v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]", p[1] + o[0], p[0] + o[1])                     \
  X(7, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]", fmaf(p[1], o[0], c[0]), fmaf(p[1], o[1], c[1]))         \
  X(8, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]", fmaf(p[0], o[0], c[1]), fmaf(p[1], o[1], c[1]))         \
  X(9, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]", fmaf(p[0], o[0], c[0]), fmaf(p[1], o[1], c[0]))      \
  X(10, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,1]", p[1] * o[1], p[1] * o[1])
#define NCASE 11
static const char* case_text[NCASE] = {
#define X(ID, TXT, LO, HI) TXT,
    CASES(X)
#undef X
};

__device__ __forceinline__ void partner_work(int partner, int iters, int lane, unsigned* __restrict__ bad) {
  h8 a, b;
  for (int k = 0; k < 8; ++k) {
    a[k] = (_Float16)(0.001f * (float)(lane + k));
    b[k] = (_Float16)(0.002f * (float)(lane - k));
  }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0;
  float fa = 0.001f * (float)lane, fb = 1.0001f;
  for (int it = 0; it < iters * 4; ++it) {
    if (partner == 1)
      asm volatile(
          "v_mfma_f32_16x16x32_f16 %0, %6, %7, %0\n\tv_mfma_f32_16x16x32_f16 %1, %6, %7, %1\n\tv_mfma_f32_16x16x32_f16 %2, %6, %7, %2\n\t"
          "v_mfma_f32_16x16x32_f16 %3, %6, %7, %3\n\tv_mfma_f32_16x16x32_f16 %4, %6, %7, %4\n\tv_mfma_f32_16x16x32_f16 %5, %6, %7, %5"
          : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5)
          : "v"(a), "v"(b));
    else if (partner == 2)
      asm volatile(
          "v_mfma_f32_16x16x4_f32 %0, %6, %7, %0\n\tv_mfma_f32_16x16x4_f32 %1, %6, %7, %1\n\tv_mfma_f32_16x16x4_f32 %2, %6, %7, %2\n\t"
          "v_mfma_f32_16x16x4_f32 %3, %6, %7, %3\n\tv_mfma_f32_16x16x4_f32 %4, %6, %7, %4\n\tv_mfma_f32_16x16x4_f32 %5, %6, %7, %5"
          : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5)
          : "v"(fa), "v"(fb));
    else
      asm volatile(
          "v_fma_f32 %0, %0, %2, %1\n\tv_fma_f32 %0, %0, %2, %1\n\tv_fma_f32 %0, %0, %2, %1\n\tv_fma_f32 %0, %0, %2, %1\n\t"
          "v_fma_f32 %0, %0, %2, %1\n\tv_fma_f32 %0, %0, %2, %1\n\tv_fma_f32 %0, %0, %2, %1\n\tv_fma_f32 %0, %0, %2, %1"
          : "+v"(fa) : "v"(fb), "v"(fb));
  }
  if (c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + fa == 12345.678f) bad[31] = 1;  // keep the chains alive
}

// PARTNER (what waves 0..3, the SIMD partners of the checking waves 4..7, do): 0 idle, 1 v_mfma_f32_16x16x32_f16 chains,
// 2 v_mfma_f32_16x16x4_f32 chains, 3 plain VALU fma chains
template <int CASE>
__global__ __launch_bounds__(512) void coexec_kernel(int partner, int iters, unsigned* __restrict__ bad) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < 4) {
    if (partner == 0) return;
    partner_work(partner, iters, lane, bad);
    return;
  }
  unsigned nlo = 0, nhi = 0;
  for (int it = 0; it < iters; ++it) {
    const f32x2 p = {1.0f + (float)((lane * 7 + it) & 63), -2.0f - (float)((lane * 3 + it) & 31)};
    const f32x2 o = {0.5f + 0.125f * (float)(lane >> 2), 0.625f + 0.125f * (float)(lane & 7)};
    const f32x2 c = {3.0f + (float)(it & 3), -5.0f + (float)(lane & 3)};
    f32x2 d;
    float elo, ehi;
#define X(ID, TXT, LO, HI)                                              \
  if constexpr (CASE == ID) {                                           \
    asm volatile(TXT : "=&v"(d) : "v"(p), "v"(o), "v"(c));              \
    elo = LO;                                                           \
    ehi = HI;                                                           \
  }
    CASES(X)
#undef X
    if (d[0] != elo) ++nlo;
    if (d[1] != ehi) ++nhi;
  }
  if (nlo) atomicAdd(&bad[(lane >> 4) * 2 + 0], nlo);
  if (nhi) atomicAdd(&bad[(lane >> 4) * 2 + 1], nhi);
}

// The same instruction with its registers pinned: does the failure depend on the VGPR banks (register number mod 4) of dst / src0 / src1?
// REGCASE(id, d0, d1, a0, a1, b0, b1, modifiers, lo, hi): v_pk_mul_f32 v[d0:d1], v[a0:a1], v[b0:b1] <modifiers>
#define REGCASES(X)                                                          \
  X(0, 40, 41, 44, 45, 48, 49, "op_sel:[0,1]", p[0] * o[1], p[1] * o[1])     \
  X(1, 42, 43, 44, 45, 48, 49, "op_sel:[0,1]", p[0] * o[1], p[1] * o[1])     \
  X(2, 40, 41, 46, 47, 48, 49, "op_sel:[0,1]", p[0] * o[1], p[1] * o[1])     \
  X(3, 40, 41, 44, 45, 50, 51, "op_sel:[0,1]", p[0] * o[1], p[1] * o[1])     \
  X(4, 42, 43, 46, 47, 50, 51, "op_sel:[0,1]", p[0] * o[1], p[1] * o[1])     \
  X(5, 42, 43, 46, 47, 48, 49, "op_sel:[0,1]", p[0] * o[1], p[1] * o[1])     \
  X(6, 44, 45, 44, 45, 48, 49, "op_sel:[0,1]", p[0] * o[1], p[1] * o[1])     \
  X(7, 48, 49, 44, 45, 48, 49, "op_sel:[0,1]", p[0] * o[1], p[1] * o[1])     \
  X(8, 40, 41, 44, 45, 48, 49, "op_sel:[1,0]", p[1] * o[0], p[1] * o[1])     \
  X(9, 42, 43, 46, 47, 50, 51, "op_sel:[1,0]", p[1] * o[0], p[1] * o[1])     \
  X(10, 44, 45, 44, 45, 48, 49, "op_sel:[1,0]", p[1] * o[0], p[1] * o[1])    \
  X(11, 40, 41, 44, 45, 48, 49, "op_sel_hi:[1,0]", p[0] * o[0], p[1] * o[0]) \
  X(12, 48, 49, 44, 45, 48, 49, "op_sel_hi:[1,0]", p[0] * o[0], p[1] * o[0]) \
  X(13, 40, 41, 44, 45, 48, 49, "op_sel_hi:[0,1]", p[0] * o[0], p[0] * o[1]) \
  X(14, 40, 41, 44, 45, 48, 49, "", p[0] * o[0], p[1] * o[1])
#define NREGCASE 15
#define STR_(x) #x
#define STR(x) STR_(x)
static const char* regcase_text[NREGCASE] = {
#define X(ID, D0, D1, A0, A1, B0, B1, MOD, LO, HI) "v_pk_mul_f32 v[" STR(D0) ":" STR(D1) "], v[" STR(A0) ":" STR(A1) "], v[" STR(B0) ":" STR(B1) "] " MOD,
    REGCASES(X)
#undef X
};
template <int CASE>
__global__ __launch_bounds__(512) void coexec_reg_kernel(int partner, int iters, unsigned* __restrict__ bad) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < 4) {
    if (partner == 0) return;
    partner_work(partner, iters, lane, bad);
    return;
  }
  unsigned nlo = 0, nhi = 0;
  for (int it = 0; it < iters; ++it) {
    const f32x2 p = {1.0f + (float)((lane * 7 + it) & 63), -2.0f - (float)((lane * 3 + it) & 31)};
    const f32x2 o = {0.5f + 0.125f * (float)(lane >> 2), 0.625f + 0.125f * (float)(lane & 7)};
    float dlo, dhi, elo, ehi;
#define X(ID, D0, D1, A0, A1, B0, B1, MOD, LO, HI)                                                                                      \
  if constexpr (CASE == ID) {                                                                                                           \
    asm volatile("v_mov_b32 v" STR(A0) ", %2\n\tv_mov_b32 v" STR(A1) ", %3\n\tv_mov_b32 v" STR(B0) ", %4\n\tv_mov_b32 v" STR(B1) ", %5\n\t" \
                 "s_nop 4\n\tv_pk_mul_f32 v[" STR(D0) ":" STR(D1) "], v[" STR(A0) ":" STR(A1) "], v[" STR(B0) ":" STR(B1) "] " MOD "\n\t"      \
                 "s_nop 4\n\tv_mov_b32 %0, v" STR(D0) "\n\tv_mov_b32 %1, v" STR(D1)                                                       \
                 : "=&v"(dlo), "=&v"(dhi)                                                                                               \
                 : "v"(p[0]), "v"(p[1]), "v"(o[0]), "v"(o[1])                                                                           \
                 : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");                                 \
    elo = LO;                                                                                                                           \
    ehi = HI;                                                                                                                           \
  }
    REGCASES(X)
#undef X
    if (dlo != elo) ++nlo;
    if (dhi != ehi) ++nhi;
  }
  if (nlo) atomicAdd(&bad[(lane >> 4) * 2 + 0], nlo);
  if (nhi) atomicAdd(&bad[(lane >> 4) * 2 + 1], nhi);
}
static const char* pn4[4] = {"idle", "mfma_f16_16x16x32", "mfma_f32_16x16x4", "v_fma_f32"};
template <int CASE>
static void run_reg(unsigned* dbad) {
  for (int partner = 0; partner < (CASE == 0 ? 4 : 2); ++partner) {
    CK(hipMemset(dbad, 0, 32 * sizeof(unsigned)));
    coexec_reg_kernel<CASE><<<256, 512>>>(partner, 20000, dbad);
    CK(hipDeviceSynchronize());
    unsigned h[32];
    CK(hipMemcpy(h, dbad, sizeof(h), hipMemcpyDeviceToHost));
    unsigned long long tot = 0;
    for (int k = 0; k < 8; ++k) tot += h[k];
    printf("%-58s partner waves: %-18s wrong %9llu of %llu  (lo, hi) per lane group:", regcase_text[CASE], pn4[partner], tot,
           256ull * 256ull * 2ull * 20000ull);
    for (int q = 0; q < 4; ++q) printf(" q%d[%u %u]", q, h[2 * q], h[2 * q + 1]);
    printf("\n");
    fflush(stdout);
  }
  if constexpr (CASE + 1 < NREGCASE) run_reg<CASE + 1>(dbad);
}

template <int CASE>
static void run(int partner, int grid, int iters, unsigned* dbad) {
  CK(hipMemset(dbad, 0, 32 * sizeof(unsigned)));
  coexec_kernel<CASE><<<grid, 512>>>(partner, iters, dbad);
  CK(hipDeviceSynchronize());
  unsigned h[32];
  CK(hipMemcpy(h, dbad, sizeof(h), hipMemcpyDeviceToHost));
  unsigned long long tot = 0;
  for (int k = 0; k < 8; ++k) tot += h[k];
  static const char* pn[4] = {"idle", "mfma_f16_16x16x32", "mfma_f32_16x16x4", "v_fma_f32"};
  printf("%-58s partner waves: %-18s grid %-4d wrong %9llu of %llu  (lo, hi) per lane group:", case_text[CASE], pn[partner], grid, tot,
         (unsigned long long)grid * 256ull * 2ull * (unsigned long long)iters);
  for (int q = 0; q < 4; ++q) printf(" q%d[%u %u]", q, h[2 * q], h[2 * q + 1]);
  printf("\n");
  fflush(stdout);
}

template <int CASE>
static void run_all(unsigned* dbad) {
  for (int partner = 0; partner < 4; ++partner) run<CASE>(partner, 256, 20000, dbad);
  if constexpr (CASE + 1 < NCASE) run_all<CASE + 1>(dbad);
}

int main() {
  unsigned* dbad;
  CK(hipMalloc(&dbad, 32 * sizeof(unsigned)));
  run_all<0>(dbad);
  printf("--- registers pinned (a VGPR's bank is its number mod 4)\n");
  run_reg<0>(dbad);
  return 0;
}
