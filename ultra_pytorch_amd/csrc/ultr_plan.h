// ultr_plan.h — host/device shared "plan" structs for the gfx950 hot path.
//
// A plan is plain-old-data computed on the host from (ultr_dnn_desc, n_rows) and passed BY VALUE
// as a kernel argument, so kernels never chase pointers for their geometry.  All offsets are in
// floats.  Row n of the batch is document (b = n / L, l = n % L); its id is docids[l*B + b].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ULTR_MAXL 8  // Linear layers (hidden + final)

// Parameter layout = the reference's DNN.sequential state_dict order (DNN.py:41-55).
struct DnnPlan {
  int nl;                 // Linear layers, = n_hidden + 1; layer nl-1 is the M=1 scorer
  int act;                // 0 elu, 1 relu
  int K[ULTR_MAXL];       // in-features of Linear_j   (K[0] = F)
  int M[ULTR_MAXL];       // out-features of Linear_j  (M[nl-1] = 1)
  int64_t off_lnw[ULTR_MAXL], off_lnb[ULTR_MAXL], off_w[ULTR_MAXL], off_b[ULTR_MAXL];
  int64_t P;              // total parameters
  // k-major copy of the hidden Linear weights (WT_j = W_j^T, [K_j, M_j] row-major), j < nl-1: the forward's MFMA
  // B-fragments are then 256-byte contiguous per 16 lanes (W_j itself gives 16 rows x 64 B per load instruction,
  // which the texture addresser processes ~4x slower); maintained by the update kernel
  int64_t wt_off[ULTR_MAXL];
  // ... followed, in the same buffer, by the "PV image": every vector parameter packed in the order the forward
  // kernel keeps them in LDS - per layer gamma[K_j] | beta[K_j] | bias[M_j], then the scorer's weight row
  // [K_last] - so a workgroup stages all of them with a handful of contiguous 16-byte loads
  int64_t wt_pv_off;
  int pv_off[ULTR_MAXL];  // offset of layer j's gamma inside the image
  int pv_wlast;           // offset of the scorer's weight row
  int pv_total;           // floats, padded to a multiple of 4
  int64_t wt_total;       // weights + image + fragment-major copies
  // ... followed by the FRAGMENT-MAJOR ("swizzled") copies the fused small-batch kernel streams (sw_ok: every hidden width a
  // multiple of 32): a wave's MFMA B operands of one 32-deep trip of the contraction are four 1-KiB-contiguous
  // buffer_load_dwordx4 - [chunk of 32 output columns][trip of 32][u = 0..3][lane = 16 q + i][ka.c0 ka.c1 kb.c0 kb.c1] with
  // ka = 32 trip + 16 (u / 2) + 4 q + 2 (u % 2), kb = ka + 1, columns 32 chunk + 2 i + {0, 1} (ultr_sw_index).  The k-major
  // copy gives 16 lanes x 8 B x 4 rows per instruction; measured on the forward product of config 2's second layer
  // (tools/swz_ubench.hip): 6.25 -> 4.65 us, with the matrix cores alone at 3.9.
  //   wsf_off[j]: W_j^T (contraction k, output m), j < nl-1        - forward  Y = U . W_j^T
  //   wsb_off[j]: W_j   (contraction m, output k), 1 <= j < nl-1   - dgrad    du = dz . W_j
  int64_t wsf_off[ULTR_MAXL], wsb_off[ULTR_MAXL];
  int64_t ws_begin;       // first float of the fragment-major region (zero-filled by ultr_dnn_build_wt: it is padded)
  int sw_ok;
  // ... and the SPLIT-HALF fragment copies of the same matrices (h3_ok: every hidden width a multiple of 32 and >= 256), for
  // the fused small-batch kernel's products on the fp16 matrix cores: every weight, scaled by 2^8, as hi = fp16(w) and
  // lo = fp16(w - hi) (22 bits of mantissa between them; 4 bytes per weight, as fp32).  A product a . w is evaluated as
  // ah.wh + ah.wl + al.wh with fp32 accumulation on v_mfma_f32_16x16x32_f16 (16x the fp32 MFMA rate, three products): the
  // result differs from the fp32 MFMA chain by less than that chain's own rounding (measured on config 2's reference
  // fixture: scores 8.3e-7 vs 7.5e-7 from the reference, gradients at 0.26 vs 0.24 of the parity tolerance).  Layout
  // (ultr_h3_index): [chunk of 32 output columns][step of 32 along the contraction][4 loads: tile0 hi, tile0 lo, tile1 hi,
  // tile1 lo][64 lanes = 16 q + j][8 halves: contraction 32 s + 8 q + e, column 32 c + 2 j + t] - the B-operand fragment
  // order of the MFMA, 1 KiB contiguous per wave load.  Offsets in FLOATS into the same buffer (two halves per float).
  int64_t whf_off[ULTR_MAXL], whb_off[ULTR_MAXL];
  int h3_ok;              // every hidden layer has both copies (the fused kernel's condition)
  // per layer: h3f[j] - the forward product of layer j has its split-half copy (M_j a multiple of 32, >= 256: eight 32-column
  // chunks); h3b[j] - so has the dgrad product du_j = dz_j . W_j (j >= 1, K_j >= 256, both widths multiples of 32)
  int h3f[ULTR_MAXL], h3b[ULTR_MAXL];
  int fb_h3;              // the fused kernel takes the split-half build (h3_ok and the ULTR_FB_H3 knob)
  int bwd_h3;             // dnn_bwd2_kernel runs at least one dgrad product on them (changes the row stride of its dz tile)
  int64_t h3_flag_off;    // one word behind the copies (every model with a hidden layer has it): ULTR_H3_FLAG_OVER | ULTR_H3_FLAG_NEAR
                          // over ALL hidden weights (status ULTR_STATUS_H3_RANGE / _NEAR)
  int no_h3;              // ultr_dnn_desc::flags & ULTR_MODEL_FP32_PRODUCTS: no split-half products for this model
  int h3_watch;           // some split-half product may read this model's weights (a ULTR_*_H3 knob is on and !no_h3): the range is reported
  int fwd_h3;             // dnn_fwd_kernel runs at least one layer on the split-half copies (changes its LDS row stride)
  int maxdim;             // max over all K_j (and M_j)
  // work map of the update kernel when it maintains the copies above: 16x16 tiles over every hidden W_j (a tile is
  // read row-major and written k-major through an LDS transpose: 64-byte segments both ways instead of a 4-byte
  // scatter), then the vector parameters as segments (parameter offset, length, position in the image)
  int upd_tile_begin[ULTR_MAXL + 1];  // first tile of layer j (prefix sums), j < nl-1
  int upd_ntk[ULTR_MAXL];             // tiles along k of layer j
  int n_vs;
  int64_t vs_off[2 * ULTR_MAXL + 2];
  int vs_len[2 * ULTR_MAXL + 2], vs_pv[2 * ULTR_MAXL + 2], vs_begin[2 * ULTR_MAXL + 3];
  // GEMM work split of the fast kernels (8 waves), precomputed: integer divisions in the kernel cost ~1k cycles per
  // layer.  Forward Y_j = X.W_j^T: 32-column chunks of M_j; fwd_ksplit > 1 = chunks x slices of the contraction.
  // Backward du_j = dz_j.W_j: bwd_mode 1 = 32-column chunks of K_j, 2 = 64-column chunks, 3 = 64-column chunks x
  // bwd_msplit slices of the contraction M_j (slice length bwd_mlen).
  int fwd_ksplit[ULTR_MAXL], fwd_klen[ULTR_MAXL], fwd_nch[ULTR_MAXL];
  // everything the forward kernel needs about layer j, in ONE 64-byte record: `const FwdLayer l = p.fl[j]` is a single
  // s_load_dwordx16 at the top of the layer (scattered p.X[j] reads were a dependent scalar load + wait each, ~1k
  // cycles per layer in the latency-bound phases)
  struct FwdLayer {
    int K, M, ksplit, klen, nch, pad;
    int64_t wt_off, sv_mean, sv_rstd, sv_x_next, off_w;
  } fl[ULTR_MAXL];
  int bwd_mode[ULTR_MAXL], bwd_msplit[ULTR_MAXL], bwd_mlen[ULTR_MAXL], bwd_nch[ULTR_MAXL];
  // saved-for-backward workspace (floats): xs[j] = input of LayerNorm_j, j >= 1; stats for all j
  int64_t sv_x[ULTR_MAXL];     // [N, K_j]   (j >= 1: activations; j = 0: only written in wg_prenorm mode)
  int64_t sv_mean[ULTR_MAXL];  // [N]
  int64_t sv_rstd[ULTR_MAXL];  // [N]
  int64_t sv_total;
};
// per-layer forward with split-half GEMMs (ultr_dnn_big.hip): hi | lo planes of 2^8 W_j, [M_j][ldK_j] halves (ldK_j = K_j rounded up to 32,
// zero-padded), rebuilt by every such forward behind the saved activations.  Host-side offsets (the plans travel as kernel arguments:
// nothing a kernel does not read belongs in them): float offset of the region inside `saved`, layer j's planes in HALVES from there
__host__ static inline int64_t ultr_fwp_off(const DnnPlan& p) { return (p.sv_total + 4 + 7) & ~(int64_t)7; }
__host__ static inline int64_t ultr_fwp_layer(const DnnPlan& p, int j) {
  int64_t h = 0;
  for (int i = 0; i < j; ++i) h += 2 * (int64_t)p.M[i] * ((p.K[i] + 31) / 32 * 32);
  return h;
}
__host__ static inline int64_t ultr_fwp_halves(const DnnPlan& p) { return ultr_fwp_layer(p, p.nl - 1); }
// ... and of 2^8 W_j^T ([K_j][ldM_j], 1 <= j < nl-1) for the split-half dgrad GEMMs of the per-layer backward, at BwdPlan::dgp_off
__host__ static inline int64_t ultr_dgp_layer(const DnnPlan& p, int j) {
  int64_t h = 0;
  for (int i = 1; i < j; ++i) h += 2 * (int64_t)p.K[i] * ((p.M[i] + 31) / 32 * 32);
  return h;
}

// position of element (output column c, contraction index k) of a fragment-major matrix with `ntrips` = ceil(Kc / 32) trips
__host__ __device__ inline int64_t ultr_sw_index(int c, int k, int ntrips) {
  const int chunk = c >> 5, i = (c & 31) >> 1, e_lo = c & 1;
  const int trip = k >> 5, kk = k & 31, h = kk >> 4, r = kk & 15, q = r >> 2, s = r & 3;
  const int u = 2 * h + (s >> 1), e_hi = s & 1;
  return ((((int64_t)chunk * ntrips + trip) * 4 + u) * 64 + (q * 16 + i)) * 4 + (e_hi * 2 + e_lo);
}

// HALF index (2-byte units) of element (output column c, contraction index k, plane hl = 0 hi / 1 lo) of a split-half
// fragment matrix with `nks` = ceil(Kc / 32) steps per chunk
#define ULTR_H3_WSCALE 256.0f  // weights are stored x 2^8: |w| up to 255, lo parts of typical weights stay normal fp16 numbers
#define ULTR_H3_WMAX 32768.0f  // scaled weights must stay below this (fp16 overflows at 65504): |w| < 128, else ULTR_STATUS_H3_RANGE
#define ULTR_H3_WNEAR 16384.0f // |w| >= 64: half of the range is used up - ULTR_STATUS_H3_NEAR, the host switches to the fp32 products
                               // while the copies are still exact (an optimizer step moves a weight by at most lr x the clip norm)
// the flag word behind the copies (DnnPlan::h3_flag_off): ULTR_H3_FLAG_OVER = a copy overflowed, ULTR_H3_FLAG_NEAR = a weight is
// near the edge (include/ultr_hip.h)
__host__ __device__ inline int64_t ultr_h3_index(int c, int k, int nks, int hl) {
  const int chunk = c >> 5, t = c & 1, j = (c & 31) >> 1;  // the two column tiles of a chunk interleave: column 32 chunk + 2 j + t
  const int s = k >> 5, q = (k & 31) >> 3, e = k & 7;      // (the accumulator layout the epilogues of the fp32 paths expect)
  return ((((int64_t)chunk * nks + s) * 4 + (2 * t + hl)) * 64 + (q * 16 + j)) * 8 + e;
}

// position of parameter e inside the PV image (DnnPlan::pv_*), or -1 when e is a hidden Linear weight
__host__ __device__ inline int ultr_pv_index(const DnnPlan& p, int64_t e) {
  for (int j = 0; j < p.nl; ++j) {
    const int K = p.K[j], M = p.M[j];
    if (e >= p.off_lnw[j] && e < p.off_lnw[j] + 2 * (int64_t)K) return p.pv_off[j] + (int)(e - p.off_lnw[j]);  // gamma | beta
    if (e >= p.off_b[j] && e < p.off_b[j] + M) return p.pv_off[j] + 2 * K + (int)(e - p.off_b[j]);
  }
  const int64_t w = p.off_w[p.nl - 1];
  if (e >= w && e < w + p.K[p.nl - 1]) return p.pv_wlast + (int)(e - w);
  return -1;
}

// One matrix-gradient segment (Linear_j, j < nl-1) for the wgrad kernel.
struct WgradLayer {
  int M, K;
  int nmb, nkb;        // 64x64 output blocks
  int nsplit;          // row (n) splits -> partial slabs
  int rows_per_split;  // multiple of 16; each of the 4 waves takes rows_per_split/4
  int blk_begin;       // first blockIdx of this layer
  int vec;             // 1: float4 path legal (M%4==0, K%4==0, aligned bases)
  int nmb2, nkb2;      // BwdPlan::wg_h3: 128 x 128 output blocks of dnn_wgrad_h3_kernel (nmb / nkb keep counting 64 x 64 blocks:
                       // the layer-0 column partials and the slab layout are indexed by those)
  int64_t dz_off;      // dz_j in bwd_ws  [N, M]
  int64_t slab_off;    // slabs in bwd_ws [nsplit][M*K + M]
};

struct BwdPlan {
  int64_t N;
  int rblk;                 // rows per workgroup of the dgrad-chain kernel
  int nrb;                  // number of row blocks = ceil(N / rblk)
  int vlen;                 // floats per vector slab
  int voff_g[ULTR_MAXL];    // dgamma_j offset inside a vector slab
  int voff_b[ULTR_MAXL];    // dbeta_j
  int voff_wk, voff_bk;     // final layer weight [K_last], bias [1]
  int64_t vslab_off;        // [nrb][vlen] in bwd_ws
  int64_t vred_off;         // [vlen]: the vector slabs pre-reduced (by spare workgroups of the wgrad launch)
  int vred_blocks;          // ceil(vlen / 64)
  int64_t dz_off[ULTR_MAXL];// dz_j [N, M_j], j < nl-1
  int64_t du_off;           // [N, max K_j over 1 <= j < nl-1]: du_j scratch of the big-batch backward (ultr_dnn_big.hip)
  WgradLayer wl[ULTR_MAXL];
  int wgrad_blocks;
  // Layer-0 shortcut of the fast backward kernels: the dgrad du_0 = dz_0 . W_0 exists only to feed LayerNorm_0's gamma/beta
  // gradients.  Instead the wgrad launch contracts dz_0 with the NORMALISED input xhat_0 (G = dz_0^T xhat_0, S = sum_r dz_0) and
  // its epilogue emits  dW_0 = gamma o G + S (x) beta  (exact algebra, no division) plus partial column sums of
  // d gamma_0 = sum_m W_0[m,:] o G[m,:]  and  d beta_0 = sum_m W_0[m,:] S[m]  per (row block of W_0, row split).
  int l0g;                  // 1: shortcut active (set by the launcher together with the kernel that skips du_0)
  // 1: `saved` holds what the wgrad launch contracts with, ready-made (the fused forward+backward kernel keeps x_j on chip, so
  // its copies in `saved` serve the weight gradients only): u_j = LayerNorm_j output for j >= 1, xhat_0 (l0g) or u_0 for j = 0.
  // The wgrad loop then issues two loads per 16 MFMAs instead of four and applies no transform.
  int wg_prenorm;
  // 1: the geometry of wl[] (blocks, row splits, slabs) is the one of dnn_wgrad_h3_kernel - weight gradients on the fp16 matrix
  // cores with split (hi / lo) operands, 128 x 128 blocks staged through LDS (ultr_dnn.hip).  Chosen by ultr_make_bwd_plan from
  // the shapes and ULTR_WG_H3; the launcher re-plans with wg_mode 0 when the pointers do not allow the 16-byte paths.
  int wg_h3;
  // wg_h3 block numbering: block b works on item  lin = (b % 8) * wg_chunk + b / 8  (consecutive block ids go round-robin to the 8
  // XCDs, so every XCD owns ONE contiguous range of items), items ordered row split first: split = lin / wg_tiles2, tile of all
  // layers = lin % wg_tiles2 (wl[j].blk_begin = first tile of layer j).  An XCD's L2 then serves ~nsplit / 8 row ranges of every
  // layer whatever the split count is - which is chosen to fill the 2 x 256 workgroup slots ONCE (544 workgroups for 512 slots
  // ran as two rounds: 113 us where 510 take 6x us at config 4).  Items >= wg_live are padding blocks.
  int wg_tiles2, wg_live, wg_chunk;
  int64_t l0part_off;       // [nmb_0 * nsplit_0][2][K_0]
  int64_t lfold_off;        // [64][tail <= 4096]: first level of the loss-partial fold when there are more than 1024 partials
  int lf_chunks, lf_len;    // 0: one workgroup folds all; else lf_chunks workgroups x lf_len partials (set by the launcher)
  int64_t dgp_off;          // per-layer backward with split-half dgrad GEMMs (ultr_dnn_big.hip): the planes of W_j^T (ultr_dgp_layer)
  int64_t sumsq_off;        // [n_red_blocks]
  int n_red_blocks;
  int64_t total;            // floats in bwd_ws
};

struct EarlyReport;
// ultr_dnn.hip: dnn_wgrad_h3_kernel for a PLAIN product (SetRank's Linears): slabs[nsplit][M*K + M] = per-row-split partials of
// dW[M, K] = dY[T, M]^T X[T, K] and of the column sums of dY; the caller folds them (same layout as sr_wgrad_kernel's)
bool ultr_wgrad_h3_geometry(int64_t T, int M, int K, int* nsplit, int* rows_per_split);
int ultr_wgrad_h3_plain(const float* dY, const float* X, int64_t T, int M, int K, float* slabs, hipStream_t st);

// Everything dnn_fb_kernel needs about layer j of BOTH loops, one 128-byte record per layer (32 ints; 64-bit offsets as lo, hi).
// The plans travel as kernel arguments in HBM; runtime-indexed reads of them are scalar loads whose first touch of a cache
// line is a miss on the critical path at the top of every phase.  The kernel copies these records into LDS with one vector
// load per thread at its start and reads a record with ONE ds_read per wave (lane = field), fields by v_readlane.
struct FbPlan {
  enum { K = 0, M, PV_OFF, KSPLIT, KLEN, NCH, BWD_NCH, BWD_MSPLIT, BWD_MODE, BWD_MLEN, VOFF_G, VOFF_B,
         WSF_OFF = 12, WSB_OFF = 14, SV_X = 16, OFF_W = 18, SV_MEAN = 20, SV_RSTD = 22, DZ_OFF = 24, WT_OFF = 26, WHF_OFF = 28,
         WHB_OFF = 30, NFIELD = 32 };
  int rec[ULTR_MAXL][NFIELD];
};

// Segment table for the deterministic slab reduction: grads[off .. off+len) =
//   sum_{s < nparts} ws[base + s*stride + e]
struct RedSeg {
  int64_t off, base, stride;
  int len, nparts;
};
struct RedPlan {
  int nseg;
  RedSeg seg[4 * ULTR_MAXL];
};

struct ultr_dnn_desc;
bool ultr_make_dnn_plan(const ultr_dnn_desc* d, int64_t N, DnnPlan* p);  // ultr_dnn.hip
bool ultr_make_bwd_plan(const DnnPlan& p, int64_t N, BwdPlan* bp, int wg_mode = -1);  // ultr_dnn.hip; wg_mode 0: never the split-half weight gradients
void ultr_make_red_plan(const DnnPlan& p, const BwdPlan& bp, RedPlan* rp); // ultr_dnn.hip

// library-internal, ultr_dnn_big.hip: the per-layer (un-fused) training forward / row-local backward for big batches
#define ULTR_BIG_ROWS 32  // rows per workgroup of its row kernels = rows per vector slab
bool ultr_dnn_big_ok(const DnnPlan& p, int64_t N, int64_t n_docs);
int ultr_dnn_big_forward(const DnnPlan& p, const float* params, const float* wt, const float* features, int64_t n_docs,
                         const int32_t* docids, int B, int L, float* scores, float* saved, hipStream_t st, hipEvent_t ev_start,
                         hipEvent_t ev_stop, bool split_half);
int ultr_dnn_big_backward(const DnnPlan& p, const BwdPlan& bp, const float* params, const float* saved, const float* dscores, float* ws,
                          hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop, bool split_half);

// library-internal: the small-batch NA/IPW step as one fused forward+loss+backward launch (+ weight gradients +
// reduction); ULTR_E_UNSUPPORTED = shape does not qualify, use the separate calls
int ultr_fused_step_softmax(const ultr_dnn_desc* d, const float* params, const float* wt, const float* features, int64_t n_docs,
                            const int32_t* docids, int32_t batch, int32_t list_size, float* scores, void* saved,
                            const float* labels, const float* pw, const float* ipw_table, int32_t n_ipw, float* dscores_out,
                            void* loss_ws, void* bwd_ws, float* grads, void* stream);

// library-internal, ultr_comm.hip: device address of a communicator's status word (the guard of the update behind the exchange)
struct ultr_comm;
const uint32_t* ultr_comm_status_word(const ultr_comm* c);

// Early loss report (single GPU, l2_loss = 0): the spare workgroup of the weight-gradient launch that folds the loss partials
// into the step tail also writes the step's loss to the host-mapped report (ultr_update_desc::host_scalars) - [0] = loss, then
// [10] = seq - i.e. as soon as the loss is FINAL (behind forward + loss), while the weight gradients, the reduction and the
// update of the same step are still running.  The host's read of the loss (the reference's loss.item()) then returns ~20 us
// before the step's last kernel ends and the next step is queued behind it without a bubble.  ultr_train_step sets this
// around its backward call; every other caller leaves host == nullptr (the update kernel's report is the only one then).
struct EarlyReport {
  float* host;
  uint32_t seq;
  int algo;
  float rlw;
};
extern thread_local EarlyReport g_ultr_early;  // ultr_step.hip
// the weight copies of the step in flight (ultr_step_args::wt) for the backward kernels ultr_train_step launches: the public
// ultr_dnn_backward has no such argument (nullptr there: the row-major parameters are streamed)
extern thread_local const float* g_ultr_step_wt;  // ultr_step.hip
// ... and in the data-parallel step the loss needs the GLOBAL sums: the exchange kernel's workgroup that reduces the head of
// the step tail reports it (ultr_comm.hip), one launch ahead of the update
struct ultr_comm;
int ultr_comm_allreduce_ex(ultr_comm* c, uint64_t step, const float* src, int64_t n, int64_t n_params, float* out, void* sumsq_ws,
                           int32_t sumsq_parts, void* stream, EarlyReport er);

// ... and the data-parallel step hands its communicator to the backward it launches: where the backward ends in the slab
// reduction, that launch runs the exchange on its own output (grad_reduce_xchg_kernel) and says so (`done`); ultr_train_step then
// goes straight to the guarded update.  nullptr everywhere else.
struct StepXchg {
  ultr_comm* comm;
  uint64_t step;
  EarlyReport er;
  bool done;
};
extern thread_local StepXchg g_ultr_step_xchg;  // ultr_step.hip
extern thread_local int g_ultr_step_nsq2;       // ultr_step.hip: level-2 sum-of-squares partials this step's reduction launch wrote (0: none)
int ultr_apply_update_ex(const ultr_update_desc* u, const ultr_dnn_desc* d, float* params, float* wt, float* state, const float* grads,
                         float* aux, const void* bwd_ws, float* scalars_out, int nsq2, void* stream);  // ultr_update.hip

#define ULTR_TAIL_FIXED 4
__host__ __device__ static inline int64_t ultr_tail_len(int L) { return ULTR_TAIL_FIXED + 2 * (int64_t)L; }

// sum-of-squares partials: one per 64 gradient elements (grad_reduce_kernel / grad_sumsq_kernel geometry)
__host__ __device__ static inline int64_t ultr_red_blocks(int64_t P, int tail) { return (P + tail + 63) / 64; }
// ... and their sums per 256 elements ("level 2": ((p0 + p1) + p2) + p3 of the four partials of a 256-element block), written by the
// slab-reduction launches of ultr_train_step behind the level-1 partials at a fixed offset of bwd_ws.  The update sums THESE when
// the step's reduction wrote them (a quarter of the words: every update workgroup reads all of them - 401 workgroups x 1601 words
// on 50 cache lines was a hot spot at the top of the launch, config 2: update 6.9 -> 6.3 us without it)
__host__ __device__ static inline int64_t ultr_sumsq2_off(int64_t P) { return (ultr_red_blocks(P, 4096) + 3) & ~(int64_t)3; }
__host__ __device__ static inline int64_t ultr_sumsq2_len(int64_t P) { return (ultr_red_blocks(P, 4096) + 3) / 4 + 4; }

// loss workspace: [0] int n_partials (as float bits unused) ; partials [MAXPART][tail]
#define ULTR_LOSS_LISTS_PER_WG 1  // one list per workgroup: a step has only `batch` lists, spread them over the CUs
__host__ __device__ static inline int64_t ultr_loss_parts(int64_t B) { return (B + ULTR_LOSS_LISTS_PER_WG - 1) / ULTR_LOSS_LISTS_PER_WG; }
