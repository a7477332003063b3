// Micro-benchmark (GPU box) of the fused kernels' per-workgroup GEMM phase: 512 threads, A tile [16 x K] in LDS, W^T [K x M]
// streamed from L2, Y [16 x M] to LDS; G workgroups run it `reps` times.  Variants: the 2-trip register ring (GemmPipe, what
// ships) against whole-panel prefetch (PanelGemm, below), an LDS-DMA ring and 16-byte loads with contraction halves.
// Findings (round 2): more loads in flight is SLOWER (panel 7.1 us, DMA ring 6.9 us vs 6.05 us for the 2-trip
// ring; loads-only 4.1 us and MFMA-only 4.0 us do not overlap on a CU), 16-byte loads win 17 % here (5.04 us) but nothing
// inside the real kernel, where every GEMM phase starts cold behind a LayerNorm.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/panel_ubench.hip ultra_pytorch_amd/csrc/ultr_prof.hip -o tools/bin/panel_ub
#include "../ultra_pytorch_amd/csrc/ultr_dnn.hip"
#include <cstdio>
#include <vector>

// The same contraction with the WHOLE W panel of the wave requested up front (NT trips of 32 contraction rows, NT <= 8: 16 * NT
// registers): for the layer widths of the latency regime (K <= 256) a wave's panel is at most 64 loads, so nothing is
// gained by metering them - every byte the wave will need is in flight before its first MFMA and the matrix cores chase
// the arriving data with counted waits (tools/l2stream_ubench: a CU sustains ~60 B/clk from L2 when >= 64 KB are in
// flight, ~23 B/clk with the 2-trip ring's 32 KB).  issue() and run() are separate so that a caller may issue the panel
// ahead of the phase that produces the A tile.
template <int RT, int NT>
struct PanelGemm {
  f32x2 b[NT][8];
  __device__ __forceinline__ void issue(const Src& W, int ldw, int kb, int ke, int c0, bool valid, int lane) {
    const int i = lane & 15, q = lane >> 4;
    const unsigned rs = (unsigned)ldw * 4u;
    const unsigned of = ((unsigned)(kb + 4 * q) * (unsigned)ldw + (unsigned)(c0 + 2 * i)) * 4u;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const bool ok0 = valid && kb + 32 * t < ke, ok1 = valid && kb + 32 * t + 16 < ke;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        b[t][s] = buf_ldv<2>(W, ok0 ? (of + (unsigned)(32 * t + s) * rs) : ULTR_OOB);
        b[t][4 + s] = buf_ldv<2>(W, ok1 ? (of + (unsigned)(32 * t + 16 + s) * rs) : ULTR_OOB);
      }
    }
    // hipcc's scheduler otherwise sinks every load to just above the MFMA that reads it (one exposed round trip each)
    __builtin_amdgcn_sched_barrier(0);
  }
  // As = A tile in LDS (zero beyond the real contraction length up to a multiple of 32); trips past the slice multiply zeros
  __device__ __forceinline__ void run(const float* __restrict__ As, int lda, int kb, f32x4 (&acc)[RT][2], int lane) {
    const int i = lane & 15, q = lane >> 4;
    const float* ap = As + i * lda + kb + 4 * q;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      {
        float4 a0[RT], a1[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          a0[rt] = ld4(ap + rt * 16 * lda + 32 * t);
          a1[rt] = ld4(ap + rt * 16 * lda + 32 * t + 16);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const float av[4] = {a0[rt].x, a0[rt].y, a0[rt].z, a0[rt].w};
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[rt][c] = mfma16(av[s], b[t][s][c], acc[rt][c]);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const float av[4] = {a1[rt].x, a1[rt].y, a1[rt].z, a1[rt].w};
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[rt][c] = mfma16(av[s], b[t][4 + s][c], acc[rt][c]);
        }
      }
    }
  }
};


template <int V, int NT>
__global__ __launch_bounds__(512) void ub_kernel(const float* __restrict__ W, int Kc, int Mo, float* __restrict__ out,
                                                 unsigned long long* __restrict__ cyc, int reps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NW = 8;
  const int ld = fwd_ld(Kc > Mo ? Kc : Mo);
  float* X = smem;
  float* Y = smem + 16 * ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int e = tid; e < 16 * ld; e += NW * 64) X[e] = ((e % ld) < Kc) ? 0.001f * (float)((e * 7 + blockIdx.x) % 97) : 0.f;
  __syncthreads();
  const Src Wt = make_src(W, (int64_t)Kc * Mo);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; ++rep) {
    if constexpr (V == 0) {
      GemmPipe<1, 2, 2, 0> pipe;
      pipe.begin(Wt, Mo, 0, Kc, wave * 32, wave * 32 < Mo, 0, lane);
      for (int cc = wave * 32; cc < Mo; cc += NW * 32) {
        f32x4 acc[1][2];
        for (int t = 0; t < 2; ++t) acc[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        pipe.run(X, ld, Wt, 0, Kc, 0, acc, lane);
        if (cc + NW * 32 < Mo) pipe.begin(Wt, Mo, 0, Kc, cc + NW * 32, true, 0, lane);
        store_nn<1, 2>(acc, Y, ld, Mo, cc, lane, false);
      }
    } else if constexpr (V == 6) {
      // 64-column chunks (16-byte loads, 256 B contiguous per row) x 2 halves of the contraction, register ring of NT trips
      const int ch = wave & 3, kh = wave >> 2;
      const int klen = ((Kc + 1) / 2 + 31) / 32 * 32;
      const int kb = kh * klen, ke = (kb + klen < Kc) ? kb + klen : Kc;
      GemmPipe<1, 4, NT, 0> pipe;
      pipe.begin(Wt, Mo, kb, ke, ch * 64, ch * 64 < Mo && kb < ke, 0, lane);
      f32x4 acc[1][4];
      for (int t = 0; t < 4; ++t) acc[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (kb < ke) pipe.run(X, ld, Wt, kb, ke, 0, acc, lane);
      if (kh == 0) store_nn<1, 4>(acc, Y, ld, Mo, ch * 64, lane, false);
      lds_barrier();
      if (kh == 1) store_nn<1, 4>(acc, Y, ld, Mo, ch * 64, lane, true);
    } else if constexpr (V == 5) {
      // LDS-DMA ring per wave: NS slots of 1 KiB = 8 contraction rows x 32 columns; B fragments by ds_read_b64
      constexpr int NS = NT;  // ring depth (slots)
      float* ring = smem + 2 * 16 * ld + wave * NS * 256;
      const int i = lane & 15, q = lane >> 4;
      const int r8 = lane >> 3, c4 = lane & 7;
      for (int cc = wave * 32; cc < Mo; cc += NW * 32) {
        const float* gsrc = W + (size_t)r8 * Mo + cc + 4 * c4;
        const int nslot = Kc / 8;
        auto dma = [&](int slot_k, int ring_slot) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + (size_t)slot_k * 8 * Mo),
                                           (__attribute__((address_space(3))) void*)(ring + ring_slot * 256), 16, 0, 0);
        };
#pragma unroll
        for (int u = 0; u < NS; ++u) dma(u, u);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const float* ap = X + i * ld + 4 * q;
        // 16 contraction rows (2 slots) per step
        for (int k16 = 0; k16 < nslot / 2; k16 += NS / 2) {
#pragma unroll
          for (int u = 0; u < NS / 2; ++u) {
            const int kk = k16 + u;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS - 2) : "memory");
            const float4 a = ld4(ap + 16 * kk);
            const float* bs = ring + (2 * u + (q >> 1)) * 256 + (4 * (q & 1)) * 32 + 2 * i;
            const float2 b0 = *reinterpret_cast<const float2*>(bs);
            const float2 b1 = *reinterpret_cast<const float2*>(bs + 32);
            const float2 b2 = *reinterpret_cast<const float2*>(bs + 64);
            const float2 b3 = *reinterpret_cast<const float2*>(bs + 96);
            acc0 = mfma16(a.x, b0.x, acc0); acc1 = mfma16(a.x, b0.y, acc1);
            acc0 = mfma16(a.y, b1.x, acc0); acc1 = mfma16(a.y, b1.y, acc1);
            acc0 = mfma16(a.z, b2.x, acc0); acc1 = mfma16(a.z, b2.y, acc1);
            acc0 = mfma16(a.w, b3.x, acc0); acc1 = mfma16(a.w, b3.y, acc1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slots are about to be overwritten
            if (2 * (kk + NS / 2) < nslot) {
              dma(2 * (kk + NS / 2), 2 * u);
              dma(2 * (kk + NS / 2) + 1, 2 * u + 1);
            } else {  // keep the vmcnt arithmetic uniform: dummy refills of the last slots
              dma(nslot - 2, 2 * u);
              dma(nslot - 1, 2 * u + 1);
            }
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f32x4 acc[1][2] = {{acc0, acc1}};
        store_nn<1, 2>(acc, Y, ld, Mo, cc, lane, false);
      }
    } else if constexpr (V == 2 || V == 3) {
      // 2: loads only (consumed by VALU adds); 3: MFMAs only (constant B operands)
      for (int cc = wave * 32; cc < Mo; cc += NW * 32) {
        PanelGemm<1, NT> pg;
        if constexpr (V == 2) pg.issue(Wt, Mo, 0, Kc, cc, true, lane);
        else {
          for (int t = 0; t < NT; ++t)
            for (int u = 0; u < 8; ++u) pg.b[t][u] = (f32x2){1.0f + (float)lane, 2.0f};
        }
        f32x4 acc[1][2];
        for (int t = 0; t < 2; ++t) acc[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (V == 2) {
          const float* ap = X + (lane & 15) * ld + 4 * (lane >> 4);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const float4 a0 = ld4(ap + 32 * t), a1 = ld4(ap + 32 * t + 16);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              acc[0][0][u & 3] += pg.b[t][u][0] * a0.x;
              acc[0][1][u & 3] += pg.b[t][u][1] * a1.x;
            }
          }
        } else {
          pg.run(X, ld, 0, acc, lane);
        }
        store_nn<1, 2>(acc, Y, ld, Mo, cc, lane, false);
      }
    } else {
      for (int cc = wave * 32; cc < Mo; cc += NW * 32) {
        PanelGemm<1, NT> pg;
        pg.issue(Wt, Mo, 0, Kc, cc, true, lane);
        f32x4 acc[1][2];
        for (int t = 0; t < 2; ++t) acc[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        pg.run(X, ld, 0, acc, lane);
        store_nn<1, 2>(acc, Y, ld, Mo, cc, lane, false);
      }
    }
    lds_barrier();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[blockIdx.x] = (t1 - t0) / reps;
  if (tid < 16) out[blockIdx.x * 16 + tid] = Y[tid * ld + tid];
}

static int g_G = 256;
template <int V, int NT>
void run(const char* name, const float* dW, int Kc, int Mo, float* dout, unsigned long long* dcyc) {
  const int G = g_G, reps = 200;
  const size_t lds = (size_t)2 * 16 * fwd_ld(Kc > Mo ? Kc : Mo) * sizeof(float) + (V == 5 ? (size_t)8 * NT * 1024 : 0);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ub_kernel<V, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((ub_kernel<V, NT>), dim3(G), dim3(512), lds, 0, dW, Kc, Mo, dout, dcyc, reps);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((ub_kernel<V, NT>), dim3(G), dim3(512), lds, 0, dW, Kc, Mo, dout, dcyc, reps);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> c(G);
  (void)hipMemcpy(c.data(), dcyc, G * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::vector<float> o(G * 16);
  (void)hipMemcpy(o.data(), dout, G * 16 * sizeof(float), hipMemcpyDeviceToHost);
  double s = 0, chk = 0;
  for (auto v : c) s += (double)v;
  for (auto v : o) chk += v;
  const double mfma_cycles = (double)16 * ((Kc + 31) / 32 * 32) * Mo / 1024.0 * 32.0 / 4.0;  // per SIMD
  printf("%-30s K=%3d M=%3d G=%3d  wall %.2f us/GEMM  memtime avg %6.0f cyc  (MFMA issue bound %5.0f cyc)  checksum %.4f\n", name, Kc, Mo, G,
         ms * 1e3 / reps, s / G, mfma_cycles, chk);
}

int main(int argc, char** argv) {
  if (argc > 1) g_G = atoi(argv[1]);
  const int KM = 256 * 256;
  std::vector<float> h(KM);
  for (int i = 0; i < KM; ++i) h[i] = 0.001f * (float)(i % 101);
  float *dW, *dout;
  unsigned long long* dcyc;
  (void)hipMalloc(&dW, KM * sizeof(float));
  (void)hipMalloc(&dout, 1024 * 16 * sizeof(float));
  (void)hipMalloc(&dcyc, 1024 * sizeof(unsigned long long));
  (void)hipMemcpy(dW, h.data(), KM * sizeof(float), hipMemcpyHostToDevice);
  run<0, 8>("ring D=2 (round 1)", dW, 256, 256, dout, dcyc);
  run<1, 8>("panel NT=8", dW, 256, 256, dout, dcyc);
  run<2, 8>("panel NT=8 loads only", dW, 256, 256, dout, dcyc);
  run<3, 8>("panel NT=8 MFMA only", dW, 256, 256, dout, dcyc);
  run<5, 8>("LDS-DMA ring 8 slots/wave", dW, 256, 256, dout, dcyc);
  run<6, 2>("x4 loads, 64-col x 2 k-halves, D=2", dW, 256, 256, dout, dcyc);
  run<6, 3>("x4 loads, 64-col x 2 k-halves, D=3", dW, 256, 256, dout, dcyc);
  run<6, 4>("x4 loads, 64-col x 2 k-halves, D=4", dW, 256, 256, dout, dcyc);
  run<6, 2>("x4 loads, 64-col x 2 k-halves, D=2", dW, 136, 256, dout, dcyc);
  run<0, 8>("ring D=2 (round 1)", dW, 136, 256, dout, dcyc);
  run<1, 5>("panel NT=5", dW, 136, 256, dout, dcyc);
  return 0;
}
