// GPU box: does a buffer load that is issued BEHIND a v_mfma_f32_16x16x32_f16 and writes one of that MFMA's SOURCE registers
// (SrcC or SrcB) ever land before the matrix core has read the register?  hipcc's hazard recogniser guards VALU writes
// behind an MFMA's SrcC read (software wait states) but not VMEM writes - a load is assumed to take longer than the
// MFMA's operand reads.  This probe pins the instruction pair in inline asm and counts wrong results per (lane group,
// accumulator register), for out-of-range offsets (the hardware bounds check answers without a memory access) and for
// L1/L2 hits, with 0..N independent MFMAs queued in front and 0..K idle cycles between the MFMA and the load.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_war_test.hip -o tools/bin/mfma_war_test && tools/bin/mfma_war_test
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                              \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                       \
      exit(1);                                                             \
    }                                                                      \
  } while (0)

// WHICH: 0 = the load overwrites SrcC, 1 = the load overwrites SrcB.  PRE = independent MFMAs queued in front.
// NOP: s_nop NOP-1 between the MFMA and the load (0 = nothing).
template <int WHICH, int PRE, int NOP>
__global__ __launch_bounds__(512) void war_kernel(const float* __restrict__ mem, unsigned nbytes, unsigned off_in, int iters,
                                                  unsigned* __restrict__ bad) {
  const int lane = threadIdx.x & 63;
  const u32x4 rs = {(unsigned)(uintptr_t)mem, (unsigned)((uintptr_t)mem >> 32) & 0xffffu, nbytes, 0x00020000u};
  const unsigned off = off_in == 0xffffffffu ? 0x80000000u : (off_in + 16u * (unsigned)lane);
  h8 a, b;
  for (int k = 0; k < 8; ++k) {
    a[k] = (_Float16)1.0f;
    b[k] = (_Float16)1.0f;
  }
  unsigned nbad[4] = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    f32x4 c, d, e0 = {0, 0, 0, 0}, e1 = e0, e2 = e0, e3 = e0;
    for (int r = 0; r < 4; ++r) c[r] = 100.0f + (float)(lane + 64 * r + (it & 7));
    const f32x4 c0 = c;
    h8 bb = b;
    if constexpr (PRE >= 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(e0) : "v"(a), "v"(b));
    if constexpr (PRE >= 2) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(e1) : "v"(a), "v"(b));
    if constexpr (PRE >= 3) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(e2) : "v"(a), "v"(b));
    if constexpr (PRE >= 4) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(e3) : "v"(a), "v"(b));
    if constexpr (WHICH == 0) {
      if constexpr (NOP == 0)
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %1\n\tbuffer_load_dwordx4 %1, %4, %5, 0 offen\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(d), "+v"(c) : "v"(a), "v"(bb), "v"(off), "s"(rs) : "memory");
      else
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %1\n\ts_nop %6\n\tbuffer_load_dwordx4 %1, %4, %5, 0 offen\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(d), "+v"(c) : "v"(a), "v"(bb), "v"(off), "s"(rs), "n"(NOP - 1) : "memory");
    } else {
      if constexpr (NOP == 0)
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %1, %3\n\tbuffer_load_dwordx4 %1, %4, %5, 0 offen\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(d), "+v"(bb) : "v"(a), "v"(c), "v"(off), "s"(rs) : "memory");
      else
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %1, %3\n\ts_nop %6\n\tbuffer_load_dwordx4 %1, %4, %5, 0 offen\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(d), "+v"(bb) : "v"(a), "v"(c), "v"(off), "s"(rs), "n"(NOP - 1) : "memory");
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    for (int r = 0; r < 4; ++r)
      if (d[r] != 32.0f + c0[r]) ++nbad[r];
    // keep the side accumulators alive
    if (e0[0] + e1[0] + e2[0] + e3[0] == 12345.0f) ++nbad[0];
  }
  for (int r = 0; r < 4; ++r)
    if (nbad[r]) atomicAdd(&bad[(lane >> 4) * 4 + r], nbad[r]);
}

template <int WHICH, int PRE, int NOP>
static void run(const char* what, const float* mem, unsigned nbytes, unsigned off, int grid, int iters, unsigned* dbad) {
  CK(hipMemset(dbad, 0, 16 * sizeof(unsigned)));
  war_kernel<WHICH, PRE, NOP><<<grid, 512>>>(mem, nbytes, off, iters, dbad);
  CK(hipDeviceSynchronize());
  unsigned h[16];
  CK(hipMemcpy(h, dbad, sizeof(h), hipMemcpyDeviceToHost));
  unsigned long long tot = 0;
  for (int k = 0; k < 16; ++k) tot += h[k];
  printf("%-5s load=%-4s pre=%d nop=%-2d grid=%-4d : wrong %10llu of %llu   per lane group q (regs r0..r3):", WHICH ? "SrcB" : "SrcC", what,
         PRE, NOP, grid, tot, (unsigned long long)grid * 512ull * 4ull * (unsigned long long)iters);
  for (int q = 0; q < 4; ++q) printf("  q%d[%u %u %u %u]", q, h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
  printf("\n");
  fflush(stdout);
}

int main() {
  const unsigned nbytes = 1u << 20;
  float* mem;
  unsigned* dbad;
  CK(hipMalloc(&mem, nbytes));
  CK(hipMalloc(&dbad, 16 * sizeof(unsigned)));
  float* h = (float*)malloc(nbytes);
  for (unsigned k = 0; k < nbytes / 4; ++k) h[k] = 7.0f;  // as SrcC: 7.0; as SrcB (two halves per float): 0x40e00000 -> (0.0, 2.4375)
  CK(hipMemcpy(mem, h, nbytes, hipMemcpyHostToDevice));
  const int iters = 2000;
  for (int grid : {8, 256, 1024}) {
    run<0, 0, 0>("oob", mem, nbytes, 0xffffffffu, grid, iters, dbad);
    run<0, 4, 0>("oob", mem, nbytes, 0xffffffffu, grid, iters, dbad);
    run<0, 0, 0>("hit", mem, nbytes, 0u, grid, iters, dbad);
    run<0, 4, 0>("hit", mem, nbytes, 0u, grid, iters, dbad);
    run<0, 4, 2>("oob", mem, nbytes, 0xffffffffu, grid, iters, dbad);
    run<0, 4, 4>("oob", mem, nbytes, 0xffffffffu, grid, iters, dbad);
    run<0, 4, 8>("oob", mem, nbytes, 0xffffffffu, grid, iters, dbad);
    run<0, 4, 16>("oob", mem, nbytes, 0xffffffffu, grid, iters, dbad);
  }
  return 0;
}
