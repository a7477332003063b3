// ultr_comm.h — device-side pieces of the data-parallel gradient exchange shared by the stand-alone exchange kernel
// (ultr_comm.hip) and the slab-reduction launch that exchanges its own output (ultr_dnn.hip, grad_reduce_xchg_kernel).
// Protocol (see ultr_comm.hip): a workgroup owns a slice of COMM_SLICE floats of the vector; it publishes the slice into the
// rank's exchange slot with system-scope write-through stores, waits for them, raises flag (slice, rank) = epoch in EVERY rank's
// flag array, polls its OWN flag row until every rank's flag carries the epoch, and adds the W slots IN RANK ORDER.  Both kernels
// use the same slice geometry, flags, slots and epochs: ranks may mix them freely within one step (the sum-of-squares partials they emit are bit-identical: same tree, tests/test_gpu_dp.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ultr_hip.h"
#include "ultr_device.h"
#include "ultr_plan.h"

#define ULTR_SYS 0x11      // cache policy sc0 | sc1: system scope (write-through stores, loads served from memory)
#define COMM_SLICE 256     // floats per workgroup slice (round 4: was 1024 - the slab-reduction launch works in blocks of 256)

struct CommDev {
  int rank, world;
  float* x_local;                          // local slot of this step
  const float* x[ULTR_COMM_MAX_WORLD];     // every rank's slot of this step (x[rank] = x_local)
  uint32_t* flags[ULTR_COMM_MAX_WORLD];    // every rank's flag array [nslice][world]
  uint32_t* status[ULTR_COMM_MAX_WORLD];   // every rank's status word: != 0 after a timed-out wait on ANY rank (the rank
                                           // that times out raises it everywhere, so no replica applies an update its
                                           // peers did not)
  long long timeout_ticks;                 // wall_clock64 ticks (100 MHz)
  float* early[ULTR_COMM_MAX_WORLD];       // every rank's early-loss area (inside its status page): [parity][rank][2 pieces of 16 bytes]
  int64_t cap;                             // floats per slot
  uint32_t epoch;                          // step + 1
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t sys_rsrc(const float* p, int64_t nfloats) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(nfloats * 4), 0x00020000);
}
__device__ __forceinline__ void sys_st4(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, float4 v) {
  const u32x4 d = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(d, rs, byte_off, 0, ULTR_SYS);
}
__device__ __forceinline__ float4 sys_ld4(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, ULTR_SYS);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void sys_st1(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, byte_off, 0, ULTR_SYS);
}
__device__ __forceinline__ float sys_ld1(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, ULTR_SYS));
}

// After this workgroup's slice has been stored into c.x_local (system scope): wait for the stores, raise the flags, wait for the
// peers.  Called by ALL threads of the workgroup (it contains barriers); `slice` = index of the slice, `sm_fail` a shared int that
// thread 0 zeroed before the first barrier of the caller.  Returns true when every peer's slice has landed (false: timed out; the
// status word of every rank has been raised).
template <int W>
__device__ __forceinline__ bool comm_flags_and_wait(const CommDev& c, int64_t slice, int* sm_fail) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the slice is acknowledged at system scope
  __syncthreads();
  const int tid = threadIdx.x;
  if (tid < W) {
    // raise flag (slice, rank) in rank tid's array; then wait for rank tid's flag in ours
    uint32_t* dst = c.flags[tid] + (slice * W + c.rank);
    __hip_atomic_store(dst, c.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t* mineflag = c.flags[c.rank] + (slice * W + tid);
    const long long t0 = wall_clock64();
    bool ok = true;
    // flags carry the step number; a peer may already be one step ahead (>=); the difference is taken modulo 2^32
    while ((int32_t)(__hip_atomic_load(mineflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - c.epoch) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > c.timeout_ticks) {
        ok = false;
        break;
      }
    }
    if (!ok) {
      *sm_fail = 1;
      for (int p = 0; p < W; ++p) __hip_atomic_store(c.status[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __syncthreads();
  // acquire at system scope behind the flag reads: nothing this workgroup cached before the peers published (L1 / non-local
  // L2 lines) may serve the slice loads below.  (The publishing side needs no L2-wide release: its slice went out with
  // write-through sc0|sc1 stores that were waited for - vmcnt(0) - before the flag stores were issued.)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  return *sm_fail == 0;
}

// the early loss report of the data-parallel step (EarlyReport, ultr_plan.h) from the four GLOBAL sums at the head of the step tail
__device__ __forceinline__ void comm_early_report(const EarlyReport& er, float loss_sum, float D, float loss2, float D2) {
  float loss = loss_sum / D;
  if (er.algo == ULTR_ALGO_DLA) loss = loss2 / D2 + er.rlw * (loss_sum / D);
  else if (er.algo == ULTR_ALGO_PAIRDEBIAS) loss = loss_sum;
  __hip_atomic_store(er.host, loss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __hip_atomic_store(reinterpret_cast<uint32_t*>(er.host) + 10, er.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Early exchange of the HEAD of the step tail (loss_sum, D, loss2_sum, D2) between the ranks, ahead of the gradient exchange: the
// loss of a data-parallel step needs the global sums, and reporting it only behind the exchange (round 3) left the host waiting
// until ~7 us before the step's end - it could not queue the next step in the shadow of this one (+5 us per synced step at world
// size 1).  Called by ONE wavefront (the workgroup of the weight-gradient launch that folds the loss partials); lanes 0..3 hold
// the local head on entry.  Every rank writes two self-validating 16-byte pieces [a, b, epoch, epoch] into EVERY rank's area,
// then polls its own area (bounded: 5 ms) and adds the ranks' values in rank order - the order of the gradient exchange, so the
// update launch's later report of the same step carries the same bits.  Returns false on a timeout (no early report then; the
// gradient exchange behind it has its own, reported, timeout).
__device__ __forceinline__ bool comm_early_head(const CommDev& c, float head_lane, float (&out)[4]) {
  const int lane = threadIdx.x & 63;
  const float h0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(head_lane), 0));
  const float h1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(head_lane), 1));
  const float h2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(head_lane), 2));
  const float h3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(head_lane), 3));
  if (c.world <= 1) {
    out[0] = 0.f + h0; out[1] = 0.f + h1; out[2] = 0.f + h2; out[3] = 0.f + h3;
    return true;
  }
  const float ep = __uint_as_float(c.epoch);
  const int slot = (int)(c.epoch & 1u) * ULTR_COMM_MAX_WORLD;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  bool ok = true;
  if (lane < c.world) {
    float* dst = c.early[lane] + (slot + c.rank) * 8;  // my pieces in rank `lane`'s area
    const __amdgpu_buffer_rsrc_t rd = sys_rsrc(dst, 8);
    sys_st4(rd, 0u, make_float4(h0, h1, ep, ep));
    sys_st4(rd, 16u, make_float4(h2, h3, ep, ep));
    const __amdgpu_buffer_rsrc_t rs = sys_rsrc(c.early[c.rank] + (slot + lane) * 8, 8);  // rank `lane`'s pieces in my area
    const long long t0 = wall_clock64();
    a = sys_ld4(rs, 0u);
    b = sys_ld4(rs, 16u);
    while (__float_as_uint(a.z) != c.epoch || __float_as_uint(a.w) != c.epoch || __float_as_uint(b.z) != c.epoch || __float_as_uint(b.w) != c.epoch) {
      if (wall_clock64() - t0 > 500000LL) {  // 5 ms of the 100 MHz wall clock
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
      a = sys_ld4(rs, 0u);
      b = sys_ld4(rs, 16u);
    }
  }
  if (__builtin_amdgcn_ballot_w64(lane < c.world && !ok) != 0ull) return false;
  out[0] = out[1] = out[2] = out[3] = 0.f;
  for (int p = 0; p < c.world; ++p) {  // rank order (wave-uniform loop)
    out[0] += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a.x), p));
    out[1] += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a.y), p));
    out[2] += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b.x), p));
    out[3] += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b.y), p));
  }
  return true;
}

// library-internal (ultr_comm.hip): the device view of communicator c for step `step`; false when not every peer is mapped
struct ultr_comm;
bool ultr_comm_dev(ultr_comm* c, uint64_t step, int64_t n, CommDev* out);
