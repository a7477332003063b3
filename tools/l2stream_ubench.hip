// Micro-benchmark (GPU box): how fast can ONE compute unit stream an L2-resident matrix (the weight stream of the fused
// forward/backward kernels), by access shape?   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/l2stream_ubench.hip -o /tmp/l2s && /tmp/l2s
//   V0  buffer_load_dwordx2, 16 lanes x 8 B contiguous (128 B) x 4 rows per instruction   (GemmPipe CT=2 today)
//   V1  buffer_load_dwordx4, 16 lanes x 16 B contiguous (256 B) x 4 rows per instruction  (CT=4)
//   V2  buffer_load_dwordx4, 64 lanes x 16 B = 1 KiB contiguous per instruction
//   V3  global_load_lds_dwordx4 (LDS-DMA), 1 KiB contiguous per instruction, no VGPR destination
//   V4  buffer_load_dword, 64 lanes x 4 B = 256 B contiguous
// G workgroups x NW waves all stream the SAME `bytes` (L2-resident after the first touch); reports B/clk/CU from
// s_memtime and the wall-clock aggregate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int V, int NW, int U, int AUX = 0>
__global__ __launch_bounds__(NW * 64) void stream_kernel(const float* __restrict__ W, int nfloats, int rowlen, int reps,
                                                         float* __restrict__ out, unsigned long long* __restrict__ cyc) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, nfloats * 4, 0x00020000);
  float acc = 0.f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; ++rep) {
    if constexpr (V == 0 || V == 1) {
      // a wave owns a 16*CT-column panel; a trip = 32 rows = 8 loads (rows 4q+s and 16+4q+s)
      constexpr int CT = (V == 0) ? 2 : 4;
      const int i = lane & 15, q = lane >> 4;
      const int npanel = rowlen / (16 * CT);
      const int nrows = nfloats / rowlen;
      for (int pn = wave; pn < npanel; pn += NW) {
        const unsigned col = pn * 16 * CT + CT * i;
        for (int m0 = 0; m0 < nrows; m0 += 32 * U / 8) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int row = m0 + 4 * q + (u & 3) + 16 * (u >> 2);
            const unsigned off = ((unsigned)row * rowlen + col) * 4u;
            if constexpr (V == 0) {
              const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, AUX);
              acc += __uint_as_float(v.x) + __uint_as_float(v.y);
            } else {
              const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX);
              acc += __uint_as_float(v.x) + __uint_as_float(v.w);
            }
          }
        }
      }
    } else if constexpr (V == 2 || V == 4) {
      constexpr int BPL = (V == 2) ? 16 : 4;
      const int per_wave_instr = 64 * BPL;  // bytes
      const int total = nfloats * 4;
      for (int o = wave * per_wave_instr * U; o < total; o += NW * per_wave_instr * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const unsigned off = (unsigned)(o + u * per_wave_instr + lane * BPL);
          if constexpr (V == 2) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX);
            acc += __uint_as_float(v.x) + __uint_as_float(v.w);
          } else {
            acc += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
          }
        }
      }
    } else if constexpr (V == 3) {
      // LDS-DMA: each wave fills its own ring of U KiB-slots
      const int total = nfloats * 4;
      float* ring = smem + wave * U * 256;
      for (int o = wave * 1024 * U; o < total; o += NW * 1024 * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const char* g = reinterpret_cast<const char*>(W) + o + u * 1024 + lane * 16;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g),
                                           (__attribute__((address_space(3))) void*)(ring + u * 256), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += ring[lane];
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if (tid == 0) cyc[blockIdx.x] = (t1 - t0) / reps;
  out[blockIdx.x * NW * 64 + tid] = acc;
}

template <int V, int NW, int U, int AUX = 0>
void run(const char* name, const float* dW, int nfloats, int rowlen, int G, float* dout, unsigned long long* dcyc) {
  const int reps = 100;
  const size_t lds = (V == 3) ? (size_t)NW * U * 1024 : 0;
  if (lds > 64 * 1024)
    hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<V, NW, U, AUX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((stream_kernel<V, NW, U, AUX>), dim3(G), dim3(NW * 64), lds, 0, dW, nfloats, rowlen, 2, dout, dcyc);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((stream_kernel<V, NW, U, AUX>), dim3(G), dim3(NW * 64), lds, 0, dW, nfloats, rowlen, reps, dout, dcyc);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> c(G);
  hipMemcpy(c.data(), dcyc, G * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : c) s += (double)v;
  const double cyc = s / G;
  const double bytes = (double)nfloats * 4;
  printf("%-44s G=%3d NW=%2d U=%2d  %8.0f cyc/pass  %6.1f B/clk/CU   wall %.2f us/pass  aggregate %.2f TB/s\n", name, G, NW, U, cyc,
         bytes / cyc, ms * 1e3 / reps, bytes * G / (ms * 1e-3 / reps) / 1e12);
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 256;
  const bool policy = argc > 2;  // cache-policy sweep (aux bits of the buffer loads: 1 sc0, 2 nt, 16 sc1)
  const int rowlen = 256, nrows = 256, nfloats = rowlen * nrows;  // 256 KB: one 256x256 layer
  std::vector<float> h(nfloats);
  for (int i = 0; i < nfloats; ++i) h[i] = 0.001f * (float)(i % 101);
  float *dW, *dout;
  unsigned long long* dcyc;
  hipMalloc(&dW, nfloats * sizeof(float));
  hipMalloc(&dout, (size_t)G * 1024 * sizeof(float));
  hipMalloc(&dcyc, G * sizeof(unsigned long long));
  hipMemcpy(dW, h.data(), nfloats * sizeof(float), hipMemcpyHostToDevice);
  if (policy) {
    run<0, 8, 8, 0>("V0 dwordx2 4 rows  aux 0", dW, nfloats, rowlen, G, dout, dcyc);
    run<0, 8, 8, 1>("V0 dwordx2 4 rows  aux sc0", dW, nfloats, rowlen, G, dout, dcyc);
    run<0, 8, 8, 2>("V0 dwordx2 4 rows  aux nt", dW, nfloats, rowlen, G, dout, dcyc);
    run<0, 8, 8, 16>("V0 dwordx2 4 rows  aux sc1", dW, nfloats, rowlen, G, dout, dcyc);
    run<0, 8, 8, 17>("V0 dwordx2 4 rows  aux sc0 sc1", dW, nfloats, rowlen, G, dout, dcyc);
    run<0, 8, 8, 18>("V0 dwordx2 4 rows  aux nt sc1", dW, nfloats, rowlen, G, dout, dcyc);
    run<0, 8, 8, 3>("V0 dwordx2 4 rows  aux sc0 nt", dW, nfloats, rowlen, G, dout, dcyc);
    run<1, 4, 8, 0>("V1 dwordx4 4 rows  aux 0", dW, nfloats, rowlen, G, dout, dcyc);
    run<1, 4, 8, 1>("V1 dwordx4 4 rows  aux sc0", dW, nfloats, rowlen, G, dout, dcyc);
    run<1, 4, 8, 2>("V1 dwordx4 4 rows  aux nt", dW, nfloats, rowlen, G, dout, dcyc);
    run<1, 4, 8, 16>("V1 dwordx4 4 rows  aux sc1", dW, nfloats, rowlen, G, dout, dcyc);
    run<1, 4, 8, 17>("V1 dwordx4 4 rows  aux sc0 sc1", dW, nfloats, rowlen, G, dout, dcyc);
    run<1, 4, 8, 18>("V1 dwordx4 4 rows  aux nt sc1", dW, nfloats, rowlen, G, dout, dcyc);
    run<2, 8, 8, 0>("V2 dwordx4 1KiB  aux 0", dW, nfloats, rowlen, G, dout, dcyc);
    run<2, 8, 8, 1>("V2 dwordx4 1KiB  aux sc0", dW, nfloats, rowlen, G, dout, dcyc);
    run<2, 8, 8, 2>("V2 dwordx4 1KiB  aux nt", dW, nfloats, rowlen, G, dout, dcyc);
    run<2, 8, 8, 16>("V2 dwordx4 1KiB  aux sc1", dW, nfloats, rowlen, G, dout, dcyc);
    run<2, 8, 8, 17>("V2 dwordx4 1KiB  aux sc0 sc1", dW, nfloats, rowlen, G, dout, dcyc);
    return 0;
  }
  run<0, 8, 8>("V0 dwordx2 16x8B x4rows (CT=2)", dW, nfloats, rowlen, G, dout, dcyc);
  run<0, 8, 16>("V0 dwordx2 16x8B x4rows (CT=2)", dW, nfloats, rowlen, G, dout, dcyc);
  run<0, 8, 32>("V0 dwordx2 16x8B x4rows (CT=2)", dW, nfloats, rowlen, G, dout, dcyc);
  run<0, 8, 64>("V0 dwordx2 16x8B x4rows (CT=2)", dW, nfloats, rowlen, G, dout, dcyc);
  run<1, 4, 8>("V1 dwordx4 16x16B x4rows (CT=4)", dW, nfloats, rowlen, G, dout, dcyc);
  run<1, 4, 16>("V1 dwordx4 16x16B x4rows (CT=4)", dW, nfloats, rowlen, G, dout, dcyc);
  run<2, 8, 4>("V2 dwordx4 1KiB contiguous", dW, nfloats, rowlen, G, dout, dcyc);
  run<2, 8, 8>("V2 dwordx4 1KiB contiguous", dW, nfloats, rowlen, G, dout, dcyc);
  run<2, 8, 16>("V2 dwordx4 1KiB contiguous", dW, nfloats, rowlen, G, dout, dcyc);
  run<2, 4, 16>("V2 dwordx4 1KiB contiguous", dW, nfloats, rowlen, G, dout, dcyc);
  run<2, 16, 8>("V2 dwordx4 1KiB contiguous", dW, nfloats, rowlen, G, dout, dcyc);
  run<3, 8, 4>("V3 LDS-DMA dwordx4 1KiB", dW, nfloats, rowlen, G, dout, dcyc);
  run<3, 8, 8>("V3 LDS-DMA dwordx4 1KiB", dW, nfloats, rowlen, G, dout, dcyc);
  run<3, 4, 8>("V3 LDS-DMA dwordx4 1KiB", dW, nfloats, rowlen, G, dout, dcyc);
  run<3, 16, 4>("V3 LDS-DMA dwordx4 1KiB", dW, nfloats, rowlen, G, dout, dcyc);
  run<4, 8, 16>("V4 dword 256B contiguous", dW, nfloats, rowlen, G, dout, dcyc);
  return 0;
}
