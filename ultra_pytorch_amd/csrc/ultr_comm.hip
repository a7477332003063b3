// ultr_comm.hip — data-parallel gradient exchange over xGMI without a collective library call in the step.
//
// The reference is single-process; row 8(e) shards the queries of a batch over the GPUs of one node and
// needs ONE sum per step of the flat vector  [ unscaled gradients (P) | step tail ]  (0.41 MB at BASELINE config 2).
// At a ~55 us step an RCCL all-reduce (launch + protocol latency, then a separate sum-of-squares pass) is a large
// fixed cost, so the exchange is ONE kernel here ("one-shot" all-reduce, every rank reads every peer):
//
//   * every rank owns an exchange buffer in FINE-GRAINED device memory, exported with hipIpcGetMemHandle and mapped by
//     every peer (xGMI peer-to-peer loads/stores; one process per GPU, handles travel over the host process group);
//   * a workgroup owns a 256-float slice (round 4; 1024 before - the slab-reduction launch that can run this exchange on its own
//     output, grad_reduce_xchg_kernel in ultr_dnn.hip, works in blocks of 256): it PUBLISHES its slice of the local vector into the local exchange slot with
//     system-scope write-through stores, waits for them (vmcnt), then raises its per-(slice, rank) flag in EVERY peer's
//     flag array (one 4-byte posted store per peer); it then polls its OWN flag row until every peer's slice of the
//     same step has landed and sums the W slices IN RANK ORDER with system-scope loads (bitwise identical on all
//     ranks, no atomics), writes the reduced slice to local memory and the sum-of-squares partials ultr_apply_update
//     consumes (the geometry of ultr_grad_sumsq: one partial per 64 gradient elements);
//   * two slots alternate by step parity: a peer's flag for step s+1 can only be raised after it finished reading step s,
//     so slot (s & 1) is free again at step s+2 without a second synchronisation;
//   * flags carry the step number (monotonic, never reset); every wait is bounded (wall clock) and a timeout is
//     reported through ultr_comm_status instead of hanging the GPU.
//
// Nothing here is torch- or RCCL-specific: the host side (parallel.py) moves the 64-byte handles with whatever process
// group exists (gloo in tests, nccl = RCCL in bench.py) and falls back to torch.distributed.all_reduce when the
// self-test of this path fails.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <new>

#include "../../include/ultr_hip.h"
#include "ultr_comm.h"

struct ultr_comm {
  int rank, world;
  int64_t cap;         // floats per slot (multiple of COMM_SLICE)
  int nslice;          // cap / COMM_SLICE
  void* base;          // local allocation: [flags: nslice * world u32, padded to 4 KB][status 4 KB][slot 0][slot 1]
  size_t bytes, flag_bytes;
  void* peer_base[ULTR_COMM_MAX_WORLD];  // mapped peer allocations (peer_base[rank] = base)
  bool mapped[ULTR_COMM_MAX_WORLD];
  int device;
};

static inline size_t round4k(size_t v) { return (v + 4095) & ~(size_t)4095; }

// src [n] local vector -> out [n] = sum over ranks; sumsq_part[k] = sum of out[e]^2 over e in [64k, 64k+64), e < n_params.
// One wave per 256-float slice (float4 per lane).
template <int W>
__global__ __launch_bounds__(64) void comm_allreduce_kernel(CommDev c, int64_t n, int64_t n_params, const float* __restrict__ src,
                                                            float* __restrict__ out, float* __restrict__ sumsq_part, int nsq,
                                                            EarlyReport er) {
  __shared__ int sm_fail;
  __shared__ float sm_head[4];
  const int tid = threadIdx.x;
  const int64_t e4 = ((int64_t)blockIdx.x * 64 + tid) * 4;
  if (tid == 0) sm_fail = 0;
  // ---- publish this workgroup's slice of the local vector (system-scope write-through) ------------------------------
  float4 mine = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e4 + 3 < n) {
    mine = ld4(src + e4);
  } else if (e4 < n) {
    mine.x = src[e4];
    if (e4 + 1 < n) mine.y = src[e4 + 1];
    if (e4 + 2 < n) mine.z = src[e4 + 2];
  }
  bool landed = true;
  if constexpr (W > 1) {
    sys_st4(sys_rsrc(c.x_local, c.cap), (unsigned)(e4 * 4), mine);
    landed = comm_flags_and_wait<W>(c, blockIdx.x, &sm_fail);
  }
  // ---- sum the W slices in rank order ---------------------------------------------------------------------------------
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (W > 1) {
    float4 v[W];
#pragma unroll
    for (int p = 0; p < W; ++p) v[p] = sys_ld4(sys_rsrc(c.x[p], c.cap), (unsigned)(e4 * 4));
#pragma unroll
    for (int p = 0; p < W; ++p) {
      s.x += v[p].x; s.y += v[p].y; s.z += v[p].z; s.w += v[p].w;
    }
    if (!landed) s = mine;  // timed out: the local vector stays; the status word (raised on every rank) makes the update
                            // launches no-ops from here on and the host's next read of the loss raises
  } else {
    s = mine;
  }
  if (e4 + 3 < n) {
    st4(out + e4, s);
  } else if (e4 < n) {
    out[e4] = s.x;
    if (e4 + 1 < n) out[e4 + 1] = s.y;
    if (e4 + 2 < n) out[e4 + 2] = s.z;
  }
  // ---- early loss report (EarlyReport, ultr_plan.h): the workgroup whose slice holds the head of the step tail
  // (loss_sum, D, loss2_sum, D2 at out[n_params .. n_params + 4)) now has the GLOBAL sums: the loss is final here, one
  // launch before the update reports everything else.
  {
    const int64_t b0 = (int64_t)blockIdx.x * COMM_SLICE;
    if (er.host != nullptr && n_params >= b0 && n_params + 4 <= b0 + COMM_SLICE && n_params + 4 <= n) {  // block-uniform
      const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t idx = e4 + k - n_params;
        if (idx >= 0 && idx < 4) sm_head[idx] = sv[k];
      }
      __syncthreads();
      // not after a timed-out wait in THIS slice - nor when another slice's wait, or a peer, already raised this rank's status
      // word (the update behind this launch is then a no-op and reports NaN + the status: the loss must not look final)
      bool failed = false;
      if constexpr (W > 1)
        failed = !landed || __hip_atomic_load(c.status[c.rank], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
      if (tid == 0 && !failed) comm_early_report(er, sm_head[0], sm_head[1], sm_head[2], sm_head[3]);
    }
  }
  // ---- sum-of-squares partials of the reduced gradient (64 elements = 16 lanes) ---------------------------------------
  // The SAME bits as grad_reduce_kernel / grad_reduce_xchg_kernel / grad_sumsq_kernel produce for this partial (ranks may
  // take different exchange paths within one step - unequal local batches - and a one-ulp difference in a partial is a different
  // clip coefficient, i.e. diverging replicas).  Those kernels hold ONE element per lane and run wave_sum: a balanced binary tree over
  // the 64 elements in index order (xor 1, xor 2, row_ror 4 / 8 read from the row's last lane, then the rows).  Here a lane
  // holds four consecutive elements = the two lowest levels of that tree, and the 16 lanes of a row finish it: read from lane 15.
  // Every square is rounded before it is added (no fma contraction - the empty asm, as in grad_reduce_kernel).
  float qx = (e4 < n_params) ? s.x * s.x : 0.f, qy = (e4 + 1 < n_params) ? s.y * s.y : 0.f;
  float qz = (e4 + 2 < n_params) ? s.z * s.z : 0.f, qw = (e4 + 3 < n_params) ? s.w * s.w : 0.f;
  asm volatile("" : "+v"(qx), "+v"(qy), "+v"(qz), "+v"(qw));
  float q = (qx + qy) + (qz + qw);
  q += dpp_or<0xb1>(0.f, q);
  q += dpp_or<0x4e>(0.f, q);
  q += dpp_or<0x124>(0.f, q);
  q += dpp_or<0x128>(0.f, q);
  const int64_t k = e4 >> 6;
  if ((tid & 15) == 15 && k < nsq) sumsq_part[k] = q;
}

extern "C" int ultr_comm_create(int32_t rank, int32_t world, int64_t n_floats, ultr_comm** out) {
  if (!out || world < 1 || world > ULTR_COMM_MAX_WORLD || rank < 0 || rank >= world || n_floats <= 0 ||
      n_floats * 4 >= ((int64_t)1 << 31))
    return ULTR_E_BADARG;
  ultr_comm* c = new (std::nothrow) ultr_comm;
  if (!c) return ULTR_E_WORKSPACE;
  memset(c, 0, sizeof(*c));
  c->rank = rank;
  c->world = world;
  c->cap = (n_floats + COMM_SLICE - 1) / COMM_SLICE * COMM_SLICE;
  c->nslice = (int)(c->cap / COMM_SLICE);
  c->flag_bytes = round4k((size_t)c->nslice * world * sizeof(uint32_t));
  c->bytes = c->flag_bytes + 4096 + 2 * round4k((size_t)c->cap * sizeof(float));
  hipError_t e = hipGetDevice(&c->device);
  if (e == hipSuccess) e = hipExtMallocWithFlags(&c->base, c->bytes, hipDeviceMallocFinegrained);
  if (e == hipSuccess) e = hipMemset(c->base, 0, c->bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    if (c->base) (void)hipFree(c->base);
    delete c;
    return (int)e;
  }
  c->peer_base[rank] = c->base;
  c->mapped[rank] = true;
  *out = c;
  return 0;
}

extern "C" int ultr_comm_export(ultr_comm* c, void* handle_out) {
  if (!c || !handle_out) return ULTR_E_BADARG;
  hipIpcMemHandle_t h;
  const hipError_t e = hipIpcGetMemHandle(&h, c->base);
  if (e != hipSuccess) return (int)e;
  static_assert(sizeof(hipIpcMemHandle_t) <= ULTR_COMM_HANDLE_BYTES, "handle size");
  memset(handle_out, 0, ULTR_COMM_HANDLE_BYTES);
  memcpy(handle_out, &h, sizeof(h));
  return 0;
}

extern "C" int ultr_comm_import(ultr_comm* c, int32_t peer, const void* handle) {
  if (!c || !handle || peer < 0 || peer >= c->world || peer == c->rank) return ULTR_E_BADARG;
  if (c->mapped[peer]) return 0;
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  void* p = nullptr;
  const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) return (int)e;
  c->peer_base[peer] = p;
  c->mapped[peer] = true;
  return 0;
}

static bool comm_ready(const ultr_comm* c) {
  for (int p = 0; p < c->world; ++p)
    if (!c->mapped[p]) return false;
  return true;
}

extern "C" int ultr_comm_allreduce(ultr_comm* c, uint64_t step, const float* src, int64_t n, int64_t n_params, float* out,
                                   void* sumsq_ws, int32_t sumsq_parts, void* stream) {
  const EarlyReport none = {nullptr, 0u, 0, 1.0f};
  return ultr_comm_allreduce_ex(c, step, src, n, n_params, out, sumsq_ws, sumsq_parts, stream, none);
}

// library-internal (ultr_train_step): the same with an early loss report from the workgroup that reduces the step tail
int ultr_comm_allreduce_ex(ultr_comm* c, uint64_t step, const float* src, int64_t n, int64_t n_params, float* out, void* sumsq_ws,
                           int32_t sumsq_parts, void* stream, EarlyReport er) {
  if (!c || !src || !out || !sumsq_ws || n <= 0 || n > c->cap || n_params < 0 || n_params > n || sumsq_parts < 0)
    return ULTR_E_BADARG;
  CommDev d;
  if (!ultr_comm_dev(c, step, n, &d)) return ULTR_E_UNSUPPORTED;
  const int nblk = (int)((n + COMM_SLICE - 1) / COMM_SLICE);
  hipStream_t st = (hipStream_t)stream;
#define COMM_LAUNCH(WW) \
  hipLaunchKernelGGL(comm_allreduce_kernel<WW>, dim3(nblk), dim3(64), 0, st, d, n, n_params, src, out, (float*)sumsq_ws, (int)sumsq_parts, er)
  switch (c->world) {
    case 1: COMM_LAUNCH(1); break;
    case 2: COMM_LAUNCH(2); break;
    case 3: COMM_LAUNCH(3); break;
    case 4: COMM_LAUNCH(4); break;
    case 5: COMM_LAUNCH(5); break;
    case 6: COMM_LAUNCH(6); break;
    case 7: COMM_LAUNCH(7); break;
    default: COMM_LAUNCH(8); break;
  }
#undef COMM_LAUNCH
  return (int)hipGetLastError();
}

bool ultr_comm_dev(ultr_comm* c, uint64_t step, int64_t n, CommDev* out) {
  if (!c || !out || n <= 0 || n > c->cap || !comm_ready(c)) return false;
  CommDev& d = *out;
  memset(&d, 0, sizeof(d));
  d.rank = c->rank;
  d.world = c->world;
  const size_t slot_bytes = round4k((size_t)c->cap * sizeof(float));
  const size_t slot_off = c->flag_bytes + 4096 + (size_t)(step & 1) * slot_bytes;
  for (int p = 0; p < c->world; ++p) {
    d.flags[p] = reinterpret_cast<uint32_t*>(c->peer_base[p]);
    d.x[p] = reinterpret_cast<const float*>(reinterpret_cast<char*>(c->peer_base[p]) + slot_off);
    d.status[p] = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(c->peer_base[p]) + c->flag_bytes);
    d.early[p] = reinterpret_cast<float*>(reinterpret_cast<char*>(c->peer_base[p]) + c->flag_bytes + 256);  // 2 x 8 ranks x 32 bytes
  }
  d.x_local = reinterpret_cast<float*>(reinterpret_cast<char*>(c->base) + slot_off);
  d.timeout_ticks = 300000000LL;  // 3 s of the 100 MHz wall clock
  d.cap = c->cap;
  d.epoch = (uint32_t)(step + 1);
  return true;
}

// library-internal: the device address of this rank's status word (ultr_update_desc::guard of the update behind the exchange)
const uint32_t* ultr_comm_status_word(const ultr_comm* c) {
  return c ? reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(c->base) + c->flag_bytes) : nullptr;
}

extern "C" int ultr_comm_status(ultr_comm* c, void* stream) {
  if (!c) return ULTR_E_BADARG;
  uint32_t v = 0;
  hipError_t e = hipStreamSynchronize((hipStream_t)stream);
  if (e == hipSuccess)
    e = hipMemcpy(&v, reinterpret_cast<char*>(c->base) + c->flag_bytes, sizeof(v), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return (int)e;
  return v ? ULTR_E_COMM_TIMEOUT : 0;
}

extern "C" int ultr_comm_destroy(ultr_comm* c) {
  if (!c) return 0;
  (void)hipDeviceSynchronize();
  for (int p = 0; p < c->world; ++p)
    if (p != c->rank && c->mapped[p] && c->peer_base[p]) (void)hipIpcCloseMemHandle(c->peer_base[p]);
  if (c->base) (void)hipFree(c->base);
  delete c;
  return 0;
}
