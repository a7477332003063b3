// ultr_dnn.hip - the DNN ranking model (reference ultra/ranking_model/DNN.py) on gfx950: plans, knobs and the C entry points.
// The kernels live next to their launchers (round 6: this file was 5 300 lines):
//   ultr_dnn_fwd.hip    dnn_fwd_kernel, dnn_fwdw_kernel            gather + [LayerNorm -> Linear -> act] x k + LayerNorm -> Linear(., 1)
//   ultr_dnn_bwd.hip    dnn_bwd_kernel, dnn_bwd2_kernel, dnn_bwdw_kernel   the row-local half of the backward
//   ultr_dnn_fb.hip     dnn_fb_kernel                               forward + NA / IPW loss + backward in one launch (small batches)
//   ultr_dnn_wgrad.hip  dnn_wgrad_kernel, dnn_wgrad_h3_kernel, grad_reduce_kernel, grad_reduce_xchg_kernel, grad_sumsq_kernel
//   ultr_dnn_kernels.h  LDS strides, the matrix-core GEMM building blocks, plan structs, the launchers' declarations
//   ultr_dnn_big.hip    the per-layer launches for shapes whose row tile does not fit LDS
// Here: ultr_make_dnn_plan / ultr_make_bwd_plan (what every kernel is handed), the planners that pick a kernel family by shape
// (fwd_wide_plan, bwd_wide_plan, big_*_wanted), the weight-copy builder, ultr_dnn_forward / _backward / _backward_softmax and the fused
// step's host side.
//
// Matrix math: v_mfma_f32_16x16x4_f32 (exact fp32) - and, for layers with >= 256 outputs / inputs where the plan carries split-half
// weight copies (DnnPlan::h3f / h3b; knobs ULTR_FB_H3 / ULTR_FWD_H3 / ULTR_BWD_H3, default on), three v_mfma_f32_16x16x32_f16 on
// hi / lo fp16 halves of both operands with fp32 accumulation (PipeH3: fp32-grade results).  Wave = 64 lanes.
// No packed fp32 VALU instructions anywhere (build.py NO_PACKED_FP32: a gfx950 hazard next to the f16 MFMAs).
#include "ultr_dnn_kernels.h"

#ifdef ULTR_TRACE
// the phase-trace arrays of the kernel units, added up (each unit stamps its own banks; the others stay zero)
extern "C" int ultr_trace_read(unsigned long long* host_out) {
  static unsigned long long part[3 * 64 * 32];
  memset(host_out, 0, sizeof(part));
  int (*readers[4])(unsigned long long*) = {ultr_trace_read_fwd, ultr_trace_read_bwd, ultr_trace_read_fb, ultr_trace_read_wgrad};
  for (int k = 0; k < 4; ++k) {
    const int rc = readers[k](part);
    if (rc) return rc;
    for (int i = 0; i < 3 * 64 * 32; ++i) host_out[i] += part[i];
  }
  return 0;
}
#endif

// ================================================================================================
// Host side: plans and launches
// ================================================================================================
// Tuning / experiment knobs (README.md): read from the environment ONCE (first use), not on every step - a getenv() walk
// per knob per launch is host time on the critical path of a ~50 us step.  ultr_config_reload() re-reads them (tests and
// the A/B tools flip knobs inside one process).
static Knobs g_knobs = {};
static int env_read(const char* name, int dflt) {
  const char* s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}
static void knobs_load() {
  Knobs k;
  k.fwd_r = env_read("ULTR_FWD_R", 0);
  k.bwd_r = env_read("ULTR_BWD_R", 0);
  k.wgrad_wgs = env_read("ULTR_WGRAD_WGS", -1);  // -1: the size-dependent default
  k.fwd_nw = env_read("ULTR_FWD_NW", 8);
  k.bwd_nw = env_read("ULTR_BWD_NW", 8);
  k.no_vec = env_read("ULTR_NO_VEC", 0);
  k.no_fused_fb = env_read("ULTR_NO_FUSED_FB", 0);
  k.fb_max_wg_per_cu = env_read("ULTR_FB_MAX_WG_PER_CU", 1);
  k.fwd_q4 = env_read("ULTR_FWD_Q4", 1);
  // the per-layer big-batch path (ultr_dnn_big.hip): 0 never, 1 by the measured rule (big_*_wanted), 2 whenever legal
  k.big_fwd = env_read("ULTR_BIG_FWD", 1);
  k.big_bwd = env_read("ULTR_BIG_BWD", 1);
  // fused small-batch kernel: products as three fp16 MFMAs on hi / lo operand splits (1, default) or fp32 MFMAs (0)
  k.fb_h3 = env_read("ULTR_FB_H3", 1);
  k.fwd_h3 = env_read("ULTR_FWD_H3", 1);
  k.bwd_h3 = env_read("ULTR_BWD_H3", 1);
  k.wg_h3 = env_read("ULTR_WG_H3", 1);                    // weight gradients on the fp16 matrix cores (split-half operands); 2: any batch size
  k.wg_h3_min_rows = env_read("ULTR_WG_H3_MIN_ROWS", 4096);
  k.wg_h3_wgs = env_read("ULTR_WG_H3_WGS", 0);
  k.bwd_wide = env_read("ULTR_BWD_WIDE", 1);  // dnn_bwdw_kernel, the same for the row-local backward of ultr_train_step
  k.fwd_wide_rmax = env_read("ULTR_FWD_WIDE_RMAX", 64);  // most rows per workgroup of dnn_fwdw_kernel (17 .. 64)
  k.fwd_wide = env_read("ULTR_FWD_WIDE", 1);  // dnn_fwdw_kernel (17 .. 64 rows per workgroup): 0 never, 1 by the measured rule (fwd_wide_plan), 2 whenever legal
  k.loaded = true;
  g_knobs = k;
}
static inline const Knobs& knobs() {
  if (!g_knobs.loaded) knobs_load();
  return g_knobs;
}
const Knobs& ultr_knobs() { return knobs(); }
void ultr_setrank_knobs_reload();  // ultr_setrank.hip
extern "C" int ultr_config_reload(void) {
  knobs_load();
  ultr_setrank_knobs_reload();
  return 0;
}

extern "C" int ultr_abi_version(void) { return ULTR_ABI_VERSION; }

static bool desc_ok(const ultr_dnn_desc* d) {
  if (!d || d->feature_size <= 0 || d->n_hidden < 0 || d->n_hidden > ULTR_MAX_HIDDEN) return false;
  for (int j = 0; j < d->n_hidden; ++j)
    if (d->hidden[j] <= 0) return false;
  return d->activation >= ULTR_ACT_ELU && d->activation <= ULTR_ACT_SIGMOID;
}

bool ultr_make_dnn_plan(const ultr_dnn_desc* d, int64_t N, DnnPlan* p) {
  if (!desc_ok(d)) return false;
  memset(p, 0, sizeof(*p));
  p->nl = d->n_hidden + 1;
  p->act = d->activation;
  p->no_h3 = (d->flags & ULTR_MODEL_FP32_PRODUCTS) ? 1 : 0;
  p->h3_watch = (!p->no_h3 && (knobs().fb_h3 != 0 || knobs().fwd_h3 != 0 || knobs().bwd_h3 != 0)) ? 1 : 0;
  int k = d->feature_size;
  int64_t off = 0;
  p->maxdim = k;
  for (int j = 0; j < p->nl; ++j) {
    const int m = (j < d->n_hidden) ? d->hidden[j] : 1;
    p->K[j] = k;
    p->M[j] = m;
    p->off_lnw[j] = off; off += k;
    p->off_lnb[j] = off; off += k;
    p->off_w[j] = off;   off += (int64_t)m * k;
    p->off_b[j] = off;   off += m;
    if (m > p->maxdim) p->maxdim = m;
    k = m;
  }
  p->P = off;
  const int NWP = 8;  // waves of the fast kernels
  for (int j = 0; j < p->nl - 1; ++j) {
    const int K = p->K[j], M = p->M[j];
    {
      const int nch = (M + 31) >> 5;
      int ks = 1;
      while (ks * 2 * nch <= NWP) ks *= 2;
      p->fwd_nch[j] = nch;
      p->fwd_ksplit[j] = ks;
      p->fwd_klen[j] = (ks > 1) ? round_up((K + ks - 1) / ks, 32) : K;
    }
    {
      const int nch = (K + 63) >> 6;
      int ms = 1;
      while (ms * 2 * nch <= NWP) ms *= 2;
      p->bwd_nch[j] = nch;
      p->bwd_msplit[j] = ms;
      p->bwd_mlen[j] = round_up((M + ms - 1) / ms, 32);
      // 32-column chunks over the whole contraction as soon as they occupy more than half of the waves: no partial-
      // tile rounds, and for a ragged width (136 = 4 x 32 + 8) far fewer wasted columns than 64-column chunks
      p->bwd_mode[j] = (ms > 1 && 2 * ((K + 31) >> 5) > NWP) ? 1 : (ms == 1 ? 2 : 3);
    }
  }
  int64_t wt = 0;
  for (int j = 0; j < p->nl - 1; ++j) {
    p->wt_off[j] = wt;
    wt += (int64_t)p->K[j] * p->M[j];
    wt = (wt + 3) & ~(int64_t)3;
  }
  p->wt_pv_off = wt;
  int pv = 0;
  for (int j = 0; j < p->nl; ++j) {
    p->pv_off[j] = pv;
    pv += 2 * p->K[j] + p->M[j];
  }
  p->pv_wlast = pv;
  pv += p->K[p->nl - 1];
  p->pv_total = (pv + 3) & ~3;
  p->wt_total = wt + p->pv_total;
  {  // fragment-major copies (DnnPlan::wsf_off / wsb_off)
    bool ok = p->nl >= 2;
    for (int j = 0; j < p->nl - 1; ++j) ok = ok && (p->M[j] % 32 == 0);
    p->sw_ok = ok ? 1 : 0;
    int64_t o = (p->wt_total + 255) & ~(int64_t)255;  // 1 KiB aligned
    p->ws_begin = o;
    if (ok) {
      for (int j = 0; j < p->nl - 1; ++j) {
        const int64_t n = (int64_t)round_up(p->K[j], 32) * round_up(p->M[j], 32);
        p->wsf_off[j] = o;
        o += n;
        if (j >= 1) {
          p->wsb_off[j] = o;
          o += n;
        }
      }
      // split-half copies (DnnPlan::whf_off / whb_off), per layer: same element counts, two halves per float
      bool h3 = true;
      for (int j = 0; j < p->nl - 1; ++j) {
        const int64_t n = (int64_t)round_up(p->K[j], 32) * round_up(p->M[j], 32);
        // 1: eight or more 32-column chunks (the 8-wave kernels take the layer on the split-half stream); 2: fewer - only the wide-tile
        // forward (dnn_fwdw_kernel: 16 waves, chunks x slices of the contraction) reads that copy
        p->h3f[j] = (p->M[j] >= 256) ? 1 : 2;
        p->h3b[j] = (j >= 1 && p->K[j] % 32 == 0) ? (p->K[j] >= 256 ? 1 : 2) : 0;  // (2: only dnn_bwdw_kernel reads that copy)
        h3 = h3 && p->h3f[j] == 1 && (j == 0 || p->h3b[j] == 1);
        if (p->h3f[j]) {
          p->whf_off[j] = o;
          o += n;
        }
        if (p->h3b[j]) {
          p->whb_off[j] = o;
          o += n;
        }
      }
      p->h3_ok = h3 ? 1 : 0;
      p->fb_h3 = (h3 && knobs().fb_h3 && !p->no_h3) ? 1 : 0;
      p->bwd_h3 = 0;
      if (knobs().bwd_h3 && !p->no_h3)
        for (int j = 1; j < p->nl - 1; ++j)
          if (p->h3b[j] == 1) p->bwd_h3 = 1;
      p->fwd_h3 = 0;
      if (knobs().fwd_h3 && !p->no_h3)
        for (int j = 0; j < p->nl - 1; ++j)
          if (p->h3f[j] == 1 && round_up(p->K[j], 32) <= 768) p->fwd_h3 = 1;
    }
    // the range word of the hidden weights (every model with a hidden layer: the per-layer big-batch path builds split-half planes of
    // ANY hidden layer, ultr_dnn_big.hip); zeroed by ultr_dnn_build_wt
    if (p->nl >= 2) {
      p->h3_flag_off = o;
      o += 4;
      p->wt_total = o;
    }
  }
  int64_t sv = 0;
  for (int j = 1; j < p->nl; ++j) {
    p->sv_x[j] = sv;
    sv += N * p->K[j];
    sv = (sv + 3) & ~(int64_t)3;
  }
  p->sv_x[0] = sv;  // normalised layer-0 input for the weight gradients (fused kernel, BwdPlan::wg_prenorm)
  sv += N * p->K[0];
  sv = (sv + 3) & ~(int64_t)3;
  for (int j = 0; j < p->nl; ++j) {
    p->sv_mean[j] = sv; sv += N;
    p->sv_rstd[j] = sv; sv += N;
  }
  p->sv_total = sv;
  {  // update-kernel work map
    int t = 0;
    for (int j = 0; j < p->nl - 1; ++j) {
      p->upd_tile_begin[j] = t;
      p->upd_ntk[j] = (p->K[j] + 15) / 16;
      t += ((p->M[j] + 15) / 16) * p->upd_ntk[j];
    }
    p->upd_tile_begin[p->nl - 1] = t;
    int n = 0, v = 0;
    auto seg = [&](int64_t o, int len, int pvpos) {
      p->vs_off[n] = o; p->vs_len[n] = len; p->vs_pv[n] = pvpos; p->vs_begin[n] = v;
      v += len; ++n;
    };
    for (int j = 0; j < p->nl; ++j) {
      seg(p->off_lnw[j], 2 * p->K[j], p->pv_off[j]);  // gamma | beta are adjacent in both layouts
      if (j < p->nl - 1) {
        seg(p->off_b[j], p->M[j], p->pv_off[j] + 2 * p->K[j]);
      } else {
        seg(p->off_w[j], p->K[j], p->pv_wlast);
        seg(p->off_b[j], 1, p->pv_off[j] + 2 * p->K[j]);
      }
    }
    p->n_vs = n;
    p->vs_begin[n] = v;
  }
  for (int j = 0; j < p->nl; ++j) {
    DnnPlan::FwdLayer& l = p->fl[j];
    l.K = p->K[j]; l.M = p->M[j];
    l.ksplit = p->fwd_ksplit[j]; l.klen = p->fwd_klen[j]; l.nch = (p->M[j] + 31) >> 5; l.pad = 0;
    l.wt_off = p->wt_off[j]; l.sv_mean = p->sv_mean[j]; l.sv_rstd = p->sv_rstd[j];
    l.sv_x_next = (j + 1 < p->nl) ? p->sv_x[j + 1] : 0;
    l.off_w = p->off_w[j];
  }
  return true;
}

static size_t fwd_pv_floats(const DnnPlan& p) { return (size_t)p.pv_total; }
static size_t fwd_lds_bytes(const DnnPlan& p, int R) {
  return ((size_t)2 * R * fwd_ld_of(p.maxdim, p.fwd_h3) + fwd_pv_floats(p)) * sizeof(float);
}
static int fwd_rows_per_wg(const DnnPlan& p, int64_t N) {
  int r = knobs().fwd_r;
  if (r == 16 || r == 32) return r;
  // 16-row tiles everywhere: 32-row tiles halve the W stream per row, but their LDS footprint leaves one workgroup per CU and
  // the grid quantises badly (measured at cfg3 / B=1024: 114 -> 100 us and 57 -> 43 us with 16 rows); ULTR_FWD_R=32 forces them
  (void)N;
  return 16;
}
static size_t bwd_lds_bytes(const DnnPlan& p, int R) {
  return ((size_t)R * (2 * bwd_ldu(p.maxdim) + bwd_ldz(p.maxdim)) + 2 * (size_t)bwd_ldu(p.maxdim) + 5 * (size_t)R) * sizeof(float) + (size_t)R * sizeof(int64_t);
}
static int bwd_rows_per_wg(const DnnPlan& p, int64_t N) {
  int r = knobs().bwd_r;
  if (r == 16 || r == 32) return r;
  return ((N + 15) / 16 > 512 && bwd_lds_bytes(p, 32) <= 160 * 1024) ? 32 : 16;
}

bool ultr_make_bwd_plan(const DnnPlan& p, int64_t N, BwdPlan* bp, int wg_mode) {
  memset(bp, 0, sizeof(*bp));
  bp->N = N;
  bp->rblk = bwd_rows_per_wg(p, N);
  bp->nrb = (int)((N + bp->rblk - 1) / bp->rblk);
  int v = 0;
  for (int j = 0; j < p.nl; ++j) {
    bp->voff_g[j] = v; v += p.K[j];
    bp->voff_b[j] = v; v += p.K[j];
  }
  bp->voff_wk = v; v += p.K[p.nl - 1];
  bp->voff_bk = v; v += 1;
  bp->vlen = v;
  int64_t off = 0;
  // sum-of-squares partials first (fixed, small)
  const int64_t tail_max = 4096;  // generous: tail is 4 + 2L floats
  bp->n_red_blocks = (int)ultr_red_blocks(p.P, (int)tail_max);
  bp->sumsq_off = off; off += bp->n_red_blocks; off = (off + 3) & ~(int64_t)3;
  off += ultr_sumsq2_len(p.P);  // level-2 partials at ultr_sumsq2_off(P) (= here: sumsq_off is 0)
  off = (off + 3) & ~(int64_t)3;
  // sized for the finest row blocking any kernel uses (the fused forward+backward kernel owns >= 9 live rows per block)
  const int64_t nrb_alloc = (N + 8) / 9 + 1 > bp->nrb ? (N + 8) / 9 + 1 : bp->nrb;
  bp->vslab_off = off; off += nrb_alloc * bp->vlen; off = (off + 3) & ~(int64_t)3;
  bp->vred_off = off; off += bp->vlen; off = (off + 3) & ~(int64_t)3;
  bp->vred_blocks = (bp->vlen + 63) / 64;
  for (int j = 0; j < p.nl - 1; ++j) {
    bp->dz_off[j] = off; off += N * p.M[j]; off = (off + 3) & ~(int64_t)3;
  }
  {
    int kmax = 0;
    for (int j = 1; j < p.nl - 1; ++j) kmax = p.K[j] > kmax ? p.K[j] : kmax;
    bp->du_off = off; off += N * kmax; off = (off + 3) & ~(int64_t)3;
  }
  // wgrad geometry
  int tiles = 0, tiles2 = 0;
  bool h3w = wg_mode != 0 && knobs().wg_h3 != 0 && p.nl >= 2 && (knobs().wg_h3 >= 2 || N >= knobs().wg_h3_min_rows);
  for (int j = 0; j < p.nl - 1; ++j) {
    tiles += ((p.M[j] + 63) / 64) * ((p.K[j] + 63) / 64);
    tiles2 += ((p.M[j] + 127) / 128) * ((p.K[j] + 127) / 128);
    if (p.M[j] % 4 != 0 || p.K[j] % 4 != 0 || p.off_w[j] % 4 != 0) h3w = false;
  }
  bp->wg_h3 = h3w ? 1 : 0;
  // workgroups over all hidden Linears: about one per CU for a small batch (every workgroup is a chain of latencies and a
  // second one on the CU only slows both), about two per CU otherwise - measured (tools/sweep_wgrad.sh, bench_configs.py):
  // N = 2560 rows: 224 -> 12.3 us, 392 -> 12.7, 448 -> 13.0;  N = 10240: 224 -> 36, 392 -> 33, 448 -> 30 us
  // Split-half launch (wg_h3): 128 x 128 blocks, a quarter of the tiles - every row split is one more slab of P floats to write and
  // to fold, so the target stays near one workgroup per CU
  const int target = h3w ? (knobs().wg_h3_wgs > 0 ? knobs().wg_h3_wgs : 256) : knobs().wgrad_wgs > 0 ? knobs().wgrad_wgs : (N < 4096 ? 224 : 448);
  const int64_t rps_cap = h3w ? 2048 : 4096;  // the per-row tables of a split live in LDS
  int blk = 0;
  for (int j = 0; j < p.nl - 1; ++j) {
    WgradLayer& w = bp->wl[j];
    w.M = p.M[j]; w.K = p.K[j];
    w.nmb = (w.M + 63) / 64; w.nkb = (w.K + 63) / 64;
    w.nmb2 = (w.M + 127) / 128; w.nkb2 = (w.K + 127) / 128;
    const int tl = h3w ? tiles2 : tiles;
    int nsplit = tl > 0 ? (h3w ? target / tl : (target + tl - 1) / tl) : 1;
    if (nsplit < 1) nsplit = 1;
    const int by_cap = (int)((N + rps_cap - 1) / rps_cap);  // the doc-id table of a split lives in LDS: at most 4096 rows
    if (nsplit < by_cap) nsplit = by_cap;
    // XCD-friendly split count.  Blocks are numbered split-fastest and consecutive block ids go round-robin to the 8 XCDs, so
    // with nsplit a divisor or a multiple of 8 each XCD works on ONE row chunk of a layer at a time and the blocks that
    // re-read the same dz / x rows (every 64 x 64 tile of that chunk) meet in one L2.  Misaligned counts fetch every operand
    // once per tile from HBM: config 4 (rocprofv3) 311 MB per launch at nsplit = 4, 115 us - nsplit 3 / 5: 154 / 158 us;
    // config 3 at nsplit = 7: 342 MB for 74 MB of operands, HBM-bound at 5.6 TB/s.
    if (knobs().wgrad_wgs <= 0 && !h3w) nsplit = nsplit <= 1 ? 1 : nsplit <= 2 ? 2 : nsplit <= 5 ? 4 : nsplit <= 11 ? 8 : (nsplit + 4) / 8 * 8;
    if (nsplit < by_cap) nsplit = h3w ? by_cap : (by_cap + 7) / 8 * 8;
    int64_t rps = (N + nsplit - 1) / nsplit;
    rps = (rps + 31) / 32 * 32;  // 8 rows per wave-trip
    if (rps < 64) rps = 64;
    if (rps > rps_cap) rps = rps_cap;
    w.rows_per_split = (int)rps;
    w.nsplit = (int)((N + rps - 1) / rps);
    w.blk_begin = blk;  // (wg_h3: the first TILE of the layer - see BwdPlan::wg_chunk)
    blk += h3w ? w.nmb2 * w.nkb2 : w.nmb * w.nkb * w.nsplit;
    w.vec = (w.M % 4 == 0 && w.K % 4 == 0) ? 1 : 0;
    w.dz_off = bp->dz_off[j];
    w.slab_off = off; off += (int64_t)w.nsplit * ((int64_t)w.M * w.K + w.M); off = (off + 3) & ~(int64_t)3;
  }
  bp->wgrad_blocks = blk;
  bp->wg_tiles2 = bp->wg_live = bp->wg_chunk = 0;
  if (h3w) {
    bp->wg_tiles2 = tiles2;
    bp->wg_live = tiles2 * bp->wl[0].nsplit;  // (every layer got the same split count: one formula, one N)
    bp->wg_chunk = (bp->wg_live + 7) / 8;
    bp->wgrad_blocks = 8 * bp->wg_chunk;
  }
  bp->lfold_off = off; off += 64 * tail_max;  // second level of the loss-partial fold (more than 1024 partials)
  bp->l0g = 0;
  bp->l0part_off = off;
  if (p.nl >= 2) off += ((int64_t)bp->wl[0].nmb * bp->wl[0].nsplit * 2 * p.K[0] + 3) & ~(int64_t)3;
  {
    const int64_t h = ultr_dgp_layer(p, p.nl - 1);
    off = (off + 7) & ~(int64_t)7;
    bp->dgp_off = off; off += (h + 1) / 2; off = (off + 3) & ~(int64_t)3;
  }
  bp->total = off;
  return true;
}


void ultr_make_red_plan(const DnnPlan& p, const BwdPlan& bp, RedPlan* rp) {
  int s = 0;
  for (int j = 0; j < p.nl; ++j) {
    const bool last = (j == p.nl - 1);
    if (j == 0 && bp.l0g) {
      const int np0 = bp.wl[0].nmb * bp.wl[0].nsplit;
      rp->seg[s++] = RedSeg{p.off_lnw[0], bp.l0part_off, 2 * (int64_t)p.K[0], p.K[0], np0};
      rp->seg[s++] = RedSeg{p.off_lnb[0], bp.l0part_off + p.K[0], 2 * (int64_t)p.K[0], p.K[0], np0};
    } else {
      rp->seg[s++] = RedSeg{p.off_lnw[j], bp.vred_off + bp.voff_g[j], 0, p.K[j], 1};
      rp->seg[s++] = RedSeg{p.off_lnb[j], bp.vred_off + bp.voff_b[j], 0, p.K[j], 1};
    }
    if (last) {
      rp->seg[s++] = RedSeg{p.off_w[j], bp.vred_off + bp.voff_wk, 0, p.K[j], 1};
      rp->seg[s++] = RedSeg{p.off_b[j], bp.vred_off + bp.voff_bk, 0, 1, 1};
    } else {
      const WgradLayer& w = bp.wl[j];
      const int64_t stride = (int64_t)w.M * w.K + w.M;
      rp->seg[s++] = RedSeg{p.off_w[j], w.slab_off, stride, w.M * w.K, w.nsplit};
      rp->seg[s++] = RedSeg{p.off_b[j], w.slab_off + (int64_t)w.M * w.K, stride, w.M, w.nsplit};
    }
  }
  rp->nseg = s;
}

// true when every hot-loop load may take the aligned branch-free float4 path
static bool all_vec(const DnnPlan& p, int vm, int64_t N, int64_t n_docs) {
  // buffer-resource offsets are 32-bit: every described tensor must stay below 2 GiB
  const int64_t lim = (int64_t)1 << 31;
  if (n_docs * p.K[0] * 4 >= lim || N * p.maxdim * 4 >= lim) return false;
  for (int j = 0; j < p.nl; ++j)
    if (!((vm >> j) & 1)) return false;
  for (int j = 0; j < p.nl - 1; ++j)
    if (p.M[j] % 4 != 0) return false;
  return (vm >> 31) & 1;
}

static int vecmask_for(const DnnPlan& p, const float* params, const float* features) {
  int mask = 0;
  const bool pa = ((uintptr_t)params & 15) == 0;
  for (int j = 0; j < p.nl; ++j)
    if (pa && p.K[j] % 4 == 0 && p.off_w[j] % 4 == 0) mask |= (1 << j);
  if (features != nullptr && ((uintptr_t)features & 15) == 0 && p.K[0] % 4 == 0) mask |= (1u << 31);
  return mask;
}

extern "C" int64_t ultr_dnn_param_count(const ultr_dnn_desc* d) {
  DnnPlan p;
  return ultr_make_dnn_plan(d, 0, &p) ? p.P : 0;
}
extern "C" int ultr_dnn_param_offsets(const ultr_dnn_desc* d, int64_t* offsets) {
  DnnPlan p;
  if (!offsets || !ultr_make_dnn_plan(d, 0, &p)) return ULTR_E_BADARG;
  for (int j = 0; j < p.nl; ++j) {
    offsets[4 * j + 0] = p.off_lnw[j];
    offsets[4 * j + 1] = p.off_lnb[j];
    offsets[4 * j + 2] = p.off_w[j];
    offsets[4 * j + 3] = p.off_b[j];
  }
  return 0;
}
extern "C" int64_t ultr_dnn_saved_bytes(const ultr_dnn_desc* d, int64_t n_rows) {
  DnnPlan p;
  if (n_rows < 0 || !ultr_make_dnn_plan(d, n_rows, &p)) return 0;
  return (ultr_fwp_off(p) + (ultr_fwp_halves(p) + 1) / 2 + 4) * (int64_t)sizeof(float);
}
extern "C" int64_t ultr_dnn_bwd_workspace_bytes(const ultr_dnn_desc* d, int64_t n_rows) {
  DnnPlan p;
  BwdPlan bp;
  if (n_rows < 0 || !ultr_make_dnn_plan(d, n_rows, &p)) return 0;
  // worst case over the env-tunable geometry: size for both row-block choices
  ultr_make_bwd_plan(p, n_rows, &bp);
  int64_t t = bp.total;
  if (bp.wg_h3) {  // the launcher may fall back to the register kernel's geometry (unaligned pointers)
    BwdPlan b0;
    ultr_make_bwd_plan(p, n_rows, &b0, 0);
    t = b0.total > t ? b0.total : t;
  }
  const int64_t extra = ((n_rows + 15) / 16) * (int64_t)bp.vlen;  // if rblk were 16
  return (t + extra + 1024) * (int64_t)sizeof(float);
}
extern "C" int64_t ultr_step_tail_floats(int32_t list_size) { return ultr_tail_len(list_size); }

extern "C" int64_t ultr_dnn_wt_floats(const ultr_dnn_desc* d) {
  DnnPlan p;
  if (!ultr_make_dnn_plan(d, 0, &p)) return 0;
  return p.wt_total > 0 ? p.wt_total : 4;
}

// WT_j[k, m] = W_j[m, k] for every hidden Linear (coalesced reads, strided writes; ~100k elements) + the PV image
__global__ __launch_bounds__(256) void wt_build_kernel(DnnPlan p, const float* __restrict__ params, float* __restrict__ wt) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= p.P) {
    const int64_t idx = p.pv_wlast + p.K[p.nl - 1] + (e - p.P);  // zero the image's padding (< 4 floats)
    if (idx < p.pv_total) wt[p.wt_pv_off + idx] = 0.f;
    return;
  }
  const int pv = ultr_pv_index(p, e);
  if (pv >= 0) {
    wt[p.wt_pv_off + pv] = params[e];
    return;
  }
  for (int j = 0; j < p.nl - 1; ++j) {
    const int64_t r = e - p.off_w[j];
    if (r >= 0 && r < (int64_t)p.M[j] * p.K[j]) {
      const int m = (int)(r / p.K[j]), k = (int)(r % p.K[j]);
      const float v = params[e];
      wt[p.wt_off[j] + (int64_t)k * p.M[j] + m] = v;
      if (p.sw_ok) {
        wt[p.wsf_off[j] + ultr_sw_index(m, k, (p.K[j] + 31) >> 5)] = v;
        if (j >= 1) wt[p.wsb_off[j] + ultr_sw_index(k, m, (p.M[j] + 31) >> 5)] = v;
      }
      const float sv = v * ULTR_H3_WSCALE;
      if (!(fabsf(sv) < ULTR_H3_WNEAR))  // every hidden weight is watched: the per-layer path builds planes of any layer (NaN counts as out of range)
        flag_or(reinterpret_cast<uint32_t*>(wt + p.h3_flag_off), !(fabsf(sv) < ULTR_H3_WMAX) ? (ULTR_H3_FLAG_OVER | ULTR_H3_FLAG_NEAR) : ULTR_H3_FLAG_NEAR);
      if (p.h3f[j] || p.h3b[j]) {
        const _Float16 hi = (_Float16)sv, lo = (_Float16)(sv - (float)hi);
        if (p.h3f[j]) {
          _Float16* hf = reinterpret_cast<_Float16*>(wt + p.whf_off[j]);
          hf[ultr_h3_index(m, k, (p.K[j] + 31) >> 5, 0)] = hi;
          hf[ultr_h3_index(m, k, (p.K[j] + 31) >> 5, 1)] = lo;
        }
        if (p.h3b[j]) {
          _Float16* hb = reinterpret_cast<_Float16*>(wt + p.whb_off[j]);
          hb[ultr_h3_index(k, m, (p.M[j] + 31) >> 5, 0)] = hi;
          hb[ultr_h3_index(k, m, (p.M[j] + 31) >> 5, 1)] = lo;
        }
      }
      return;
    }
  }
}

extern "C" int ultr_dnn_build_wt(const ultr_dnn_desc* d, const float* params, float* wt, void* stream) {
  DnnPlan p;
  if (!params || !wt || !ultr_make_dnn_plan(d, 0, &p)) return ULTR_E_BADARG;
  const int64_t n = p.P + 4;
  if (p.wt_total > p.ws_begin) {  // the fragment-major copies are padded to whole 32 x 32 blocks: zeros there (and the range word)
    const hipError_t e = hipMemsetAsync(wt + p.ws_begin, 0, (size_t)(p.wt_total - p.ws_begin) * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(wt_build_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, params, wt);
  return (int)hipGetLastError();
}

extern "C" int ultr_dnn_wt_range(const ultr_dnn_desc* d, const float* wt, void* stream) {
  DnnPlan p;
  if (!wt || !ultr_make_dnn_plan(d, 0, &p)) return ULTR_E_BADARG;
  if (p.h3_flag_off <= 0) return 0;  // no hidden layer: nothing is ever split
  uint32_t f = 0;
  hipError_t e = hipMemcpyAsync(&f, wt + p.h3_flag_off, sizeof(f), hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) return -(int)e - 100;
  return (f & ULTR_H3_FLAG_OVER) ? 2 : ((f & ULTR_H3_FLAG_NEAR) ? 1 : 0);
}

// Which shapes take the per-layer path (ultr_dnn_big.hip) instead of the row-tile kernels below; mode = the knob (0 never,
// 1 the measured rule, 2 whenever legal).  Measured ("big-batch path"; us, row-tile -> per-layer):
//   forward   slower in general: config 3 85 -> 120, 81 920 rows x [256,256] 258 -> 297 (the tiled GEMM core runs at 63-85
//             TFLOP/s, the row-tile forward at 61-70 with no activation round trips) - taken when the row-tile kernel's LDS
//             footprint does not fit, and in the case big_fwd_wanted describes (config 4: 220 -> 210);
//   backward  wins when dnn_bwd2_kernel does not apply (a layer wider than 512: config 4 128 -> 85) and for big batches
//             (81 920 x [256,256] 202 -> 182, 163 840 x [512,256,128] 980 -> 830); a tie at config 3 (10 240 rows: 74 / 74).
static int dnn_device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
  }
  return cus;
}
// The one case where the per-layer forward wins: a wide gathered input (K_0 > 512: the statistics pass then writes xhat_0
// contiguously and the first GEMM reads plain rows, ultr_dnn_big.hip) AND both staircases line up - the row tiles would
// run >= 3 full rounds of resident workgroups plus a last round less than a quarter full (their time is a staircase in the
// tile count: config 4 = 800 tiles of a one-per-CU workgroup: 172 us at 750 tiles, 221 us at 800), while the first GEMM's
// 64 x 128 chunks still fit about one round of ITS resident workgroups (3 per CU).  Measured, row tiles -> per-layer (us),
// F700 [512,256,128]: 257 tiles 111 -> 133 (fixed cost of seven launches), 750 tiles 176 -> 193, 800 tiles 221 -> 209,
// 938 tiles 224 -> 215, 1063 tiles 274 -> 281 (the GEMM's own second round).
static bool big_fwd_wanted(const DnnPlan& p, int64_t N, size_t row_tile_lds) {
  if (knobs().big_fwd != 1) return knobs().big_fwd >= 2;
  if (p.K[0] <= 512 || p.nl < 2) return false;
  // round 3: with its first layer on the split-half copies the row-tile forward wins that case too (config 4, 800 tiles:
  // 218 us per-layer, 225 row tiles in fp32, 151 row tiles with the split-half products)
  if (p.fwd_h3 && p.h3f[0] == 1 && round_up(p.K[0], 32) <= 768) return false;
  const int cus = dnn_device_cus();
  const int per_cu = row_tile_lds > 80 * 1024 ? 1 : (row_tile_lds > 53 * 1024 ? 2 : 3);
  const int64_t slots = (int64_t)per_cu * cus, tiles = (N + 15) / 16, full = tiles / slots, rem = tiles % slots;
  const int64_t chunks0 = ((N + 63) / 64) * ((p.M[0] + 127) / 128), gslots = 3 * (int64_t)cus;
  return full >= 3 && rem != 0 && rem * 4 <= slots && chunks0 <= gslots + gslots / 8;
}
static bool big_bwd_wanted(const DnnPlan& p, int64_t N) {
  const int mode = knobs().big_bwd;
  if (mode != 1) return mode >= 2;
  const bool v2 = knobs().bwd_nw == 8 && p.maxdim <= 512 &&
                  bwd2_lds_floats(p, bwd_rows_per_wg(p, N), 8) * sizeof(float) <= 160 * 1024;
  return N >= (v2 ? 16384 : 4096);
}

// dnn_fwdw_kernel: which shapes take it, and with how many rows per workgroup.  Legal: every hidden layer has its split-half
// forward copy (widths multiples of 32) and no LayerNorm is wider than 768 (three float4 per lane).  Rows: as many as the LDS holds
// (two buffers sized per layer parity + the parameter image), at most 64; then the smallest R that still covers the batch in the
// same number of rounds of one workgroup per CU.  Taken when that is more than the 16 rows of dnn_fwd_kernel.
static bool fwd_wide_plan(const DnnPlan& p, int64_t N, WidePlan* wp, size_t* lds_bytes) {
  if (knobs().fwd_wide == 0 || knobs().fwd_h3 == 0 || p.no_h3 || p.nl < 2 || !p.sw_ok || p.pv_total > 2 * 1024 * 4) return false;
  if (knobs().fwd_r != 0 || knobs().fwd_nw != 8) return false;  // an explicit tile geometry of the row-tile kernel was asked for
  int w[2] = {0, 0};
  for (int j = 0; j < p.nl; ++j) {
    const int k16 = round_up(p.K[j], 32);
    if (k16 > 768 || (j < p.nl - 1 && p.h3f[j] == 0)) return false;
    if (k16 + 8 > w[j & 1]) w[j & 1] = k16 + 8;
  }
  const int64_t fixed = ((int64_t)p.pv_total + 64) * 4, per_row = (int64_t)(w[0] + w[1]) * 4;
  int64_t rmax = (160 * 1024 - fixed) / per_row - 1;  // the buffers hold R + 1 rows (dnn_fwdw_kernel: the rows of the last MFMA tile beyond R land in row R)
  const int cap = knobs().fwd_wide_rmax < 17 ? 17 : (knobs().fwd_wide_rmax > 64 ? 64 : knobs().fwd_wide_rmax);  // four MFMA row tiles at most
  if (rmax > cap) rmax = cap;
  if (rmax < 17) return false;
  const int64_t cus = dnn_device_cus();
  const int64_t rounds = (N + cus * rmax - 1) / (cus * rmax);
  const int64_t R = (N + cus * rounds - 1) / (cus * rounds);
  if (R <= 16) return false;
  // Several rounds of small workgroups behind a small weight set are what the 16-row kernel (two or three workgroups per CU, their
  // phases overlapping) does well: validation of config 2's model at 100 candidates (25 600 rows, 0.4 MB of weights), forward us:
  // 16-row tiles 63.7; wide 59.2 at 2 rounds x 50 rows, 67.3 at 3 x 34, 71.2 at 4 x 25.  Against that: config 3 (1 round x 40 rows,
  // 0.94 MB) 60 -> 41, config 4 (2 x 25 rows, 2.1 MB) 151 -> 95.  Wide tiles when one round covers the batch, when the tiles are
  // nearly full 48- / 64-row ones, or when the weights are what the 16-row tiles spend their time streaming.
  int64_t wbytes = 0;
  for (int j = 0; j < p.nl - 1; ++j) wbytes += (int64_t)round_up(p.K[j], 32) * p.M[j] * 4;
  if (knobs().fwd_wide == 1 && rounds > 1 && R < 44 && wbytes < 768 * 1024) return false;
  memset(wp, 0, sizeof(*wp));
  wp->R = (int)R;
  wp->buf[0] = 0;
  wp->buf[1] = (int)(R + 1) * w[0];
  wp->pv = (int)(R + 1) * (w[0] + w[1]);
  for (int j = 0; j < p.nl - 1; ++j) {
    const int nks = round_up(p.K[j], 32) / 32, nch = p.M[j] / 32;
    int ks = nch >= 16 ? 1 : 16 / nch;
    if (ks > nks) ks = nks;
    const int len = (nks + ks - 1) / ks;
    wp->kslen[j] = len;
    wp->ksplit[j] = (nks + len - 1) / len;
  }
  *lds_bytes = (size_t)(wp->pv + p.pv_total + 64) * sizeof(float);
  return true;
}

// dnn_bwdw_kernel: legal when the layer-0 shortcut is on, every LayerNorm of the layers >= 1 is at most 512 wide and every dgrad
// product has its split-half copy; rows per workgroup as in fwd_wide_plan (LDS: the dz planes, the du tile that also holds the
// column partials of sixteen waves, 128 floats of per-row scalars)
static bool bwd_wide_plan(const DnnPlan& p, int64_t N, WideBwd* wb, size_t* lds_bytes) {
  if (knobs().bwd_wide == 0 || knobs().bwd_h3 == 0 || p.no_h3 || p.nl < 2 || !p.sw_ok) return false;
  if (knobs().bwd_r != 0 || knobs().bwd_nw != 8 || knobs().big_bwd >= 2) return false;  // another kernel was asked for explicitly
  if (p.sv_total * 4 >= ((int64_t)1 << 31)) return false;
  const int top = p.nl - 1;
  int wdz = 0, wdu = 0, cpmax = 16 * 3 * p.K[top];
  for (int j = 1; j <= top; ++j) {
    if (p.K[j] > 512 || p.K[j] % 32 != 0) return false;
    if (j < top) {
      if (p.h3b[j] == 0) return false;
      if (p.K[j] + 8 > wdu) wdu = p.K[j] + 8;
      if (p.M[j] + 8 > wdz) wdz = p.M[j] + 8;
      if (16 * 2 * p.K[j] > cpmax) cpmax = 16 * 2 * p.K[j];
    }
  }
  auto floats = [&](int64_t R) {
    const int64_t du = (R + 1) * wdu > cpmax ? (R + 1) * wdu : cpmax;
    return (R + 1) * wdz + du + 128;
  };
  int64_t rmax = 64;  // four row tiles at most
  while (rmax > 16 && floats(rmax) * 4 > 160 * 1024) --rmax;
  if (rmax < 17) return false;
  const int64_t cus = dnn_device_cus();
  const int64_t rounds = (N + cus * rmax - 1) / (cus * rmax);
  const int64_t R = (N + cus * rounds - 1) / (cus * rounds);
  if (R <= 16) return false;
  memset(wb, 0, sizeof(*wb));
  wb->R = (int)R;
  wb->dz = 0;
  wb->du = (int)((R + 1) * wdz);
  wb->ds = (int)(floats(R) - 128);
  for (int j = 1; j < top; ++j) {
    const int nks = p.M[j] / 32, nch = p.K[j] / 32;
    int ks = nch >= 16 ? 1 : 16 / nch;
    if (ks > nks) ks = nks;
    const int len = (nks + ks - 1) / ks;
    wb->kslen[j] = len;
    wb->ksplit[j] = (nks + len - 1) / len;
  }
  *lds_bytes = (size_t)floats(R) * sizeof(float);
  return true;
}


extern "C" int ultr_dnn_forward(const ultr_dnn_desc* d, const float* params, const float* wt, const float* features,
                                int64_t n_docs, const int32_t* docids, int32_t batch, int32_t list_size, float* scores,
                                void* saved, void* stream) {
  if (!params || !docids || !scores || batch <= 0 || list_size <= 0 || n_docs < 0 || (n_docs > 0 && !features))
    return ULTR_E_BADARG;
  const int64_t N = (int64_t)batch * list_size;
  DnnPlan p;
  if (!ultr_make_dnn_plan(d, N, &p)) return ULTR_E_BADARG;
  const int R = fwd_rows_per_wg(p, N);
  const size_t lds = fwd_lds_bytes(p, R);
  const int nw = knobs().fwd_nw;
  const int vm = vecmask_for(p, params, features);
  hipStream_t st = (hipStream_t)stream;
  UltrProfScope prof(ULTR_K_FWD, st);
  // the fast path needs the k-major weight copy (ultr_dnn_build_wt / kept current by ultr_apply_update)
  const bool av = all_vec(p, vm, N, n_docs) && knobs().no_vec == 0 && (wt != nullptr || p.nl == 1) &&
                  ((uintptr_t)wt & 15) == 0;
  // training forward of a big batch: one pass per layer (needs `saved` for the activations between the passes)
  if (saved != nullptr && wt != nullptr && av && ultr_dnn_big_ok(p, N, n_docs) &&
      knobs().big_fwd != 0 && (big_fwd_wanted(p, N, lds) || lds > 160 * 1024))
    return ultr_dnn_big_forward(p, params, wt, features, n_docs, docids, (int)batch, (int)list_size, scores, (float*)saved, st,
                                prof.on ? prof.a : nullptr, prof.on ? prof.b : nullptr, knobs().fwd_h3 != 0 && !p.no_h3 && wt != nullptr);
  {
    WidePlan wp;
    size_t wlds = 0;
    if (av && wt != nullptr && fwd_wide_plan(p, N, &wp, &wlds)) {
      return ultr_launch_dnn_fwdw(prof, p, wp, wlds, st, features, n_docs, docids, (int)batch, (int)list_size, scores, (float*)saved, wt);
    }
  }
  if (lds > 160 * 1024) return ULTR_E_UNSUPPORTED;
  bool q4 = false;  // one workgroup per CU anyway and a layer that takes 64-column chunks: the 16-byte-load build
  if (av && R == 16 && nw == 8 && knobs().fwd_q4 && lds > 80 * 1024)
    for (int j = 0; j < p.nl - 1; ++j) q4 = q4 || (p.M[j] % 64 == 0 && p.M[j] >= 512);
  return ultr_launch_dnn_fwd(prof, p, R, nw, av, q4, lds, st, params, features, n_docs, docids, (int)batch, (int)list_size, scores, (float*)saved, wt,
                             vm);
}

// validation(): ultr_dnn_forward (saved = NULL) + ultr_ndcg_report of its scores behind ONE host call.  (Round 6 also built the ONE-LAUNCH
// form - dnn_fwd_kernel with list-aligned tiles and ndcg_list_kernel's code as its epilogue, bit-identical - and measured it slower: 24.8 us
// against 15.05 + 8.58 us of kernel time, 34.5 - 36.4 against 32.5 - 33.8 us per batch read on the host.  The metric's cost is its chain of
// dependent memory round trips - labels, write-through per-list values, the arrival counter, the last wave's read-back, the host report -
// not its launch, and behind the forward that chain is exposed in full instead of overlapping the forward's tail.  Removed)
extern "C" int ultr_dnn_forward_ndcg(const ultr_dnn_desc* d, const float* params, const float* wt, const float* features, int64_t n_docs,
                                     const int32_t* docids, const float* labels, int32_t batch, int32_t list_size, float* scores,
                                     const int32_t* topn, int32_t n_topn, float* ndcg_out, int32_t* order_out, float* masked_out,
                                     float* ndcg_ws, uint32_t* counter, float* host_report, uint32_t seq, void* stream) {
  if (!labels || !topn || !ndcg_out || !ndcg_ws || !counter || n_topn <= 0) return ULTR_E_BADARG;
  const int rc = ultr_dnn_forward(d, params, wt, features, n_docs, docids, batch, list_size, scores, nullptr, stream);
  if (rc != 0) return rc;
  return ultr_ndcg_report(scores, labels, docids, n_docs, batch, list_size, topn, n_topn, ndcg_out, order_out, masked_out, ndcg_ws, counter,
                          host_report, seq, stream);
}

extern "C" int32_t ultr_dnn_forward_tile_rows(const ultr_dnn_desc* d, int64_t n_rows, int32_t training) {
  DnnPlan p;
  if (n_rows <= 0 || !ultr_make_dnn_plan(d, n_rows, &p)) return -1;
  const int R = fwd_rows_per_wg(p, n_rows);
  const size_t lds = fwd_lds_bytes(p, R);
  if (training && ultr_dnn_big_ok(p, n_rows, n_rows) && knobs().big_fwd != 0 && (big_fwd_wanted(p, n_rows, lds) || lds > 160 * 1024)) return 0;
  WidePlan wp;
  size_t wlds = 0;
  if (knobs().no_vec == 0 && fwd_wide_plan(p, n_rows, &wp, &wlds)) return 1000 + wp.R;
  return R;
}

extern "C" int32_t ultr_dnn_backward_tile_rows(const ultr_dnn_desc* d, int64_t n_rows) {
  DnnPlan p;
  BwdPlan bp;
  if (n_rows <= 0 || !ultr_make_dnn_plan(d, n_rows, &p) || !ultr_make_bwd_plan(p, n_rows, &bp)) return -1;
  WideBwd wb;
  size_t wl = 0;
  if (knobs().no_vec == 0 && bwd_wide_plan(p, n_rows, &wb, &wl)) return 1000 + wb.R;
  if (knobs().big_bwd != 0 && ultr_dnn_big_ok(p, n_rows, n_rows) && (big_bwd_wanted(p, n_rows) || bwd_lds_bytes(p, bp.rblk) > 160 * 1024)) return 0;
  return bp.rblk;
}

bool ultr_wgrad_h3_geometry(int64_t T, int M, int K, int* nsplit, int* rows_per_split) {
  if (T <= 0 || M <= 0 || K <= 0 || M % 4 != 0 || K % 4 != 0) return false;
  if (T * (int64_t)(M > K ? M : K) * 4 >= ((int64_t)1 << 31)) return false;  // 32-bit buffer offsets
  const int tiles2 = ((M + 127) / 128) * ((K + 127) / 128);
  int ns = 256 / tiles2;
  if (ns < 1) ns = 1;
  int64_t rps = (T + ns - 1) / ns;
  rps = (rps + 31) / 32 * 32;
  if (rps < 64) rps = 64;
  *rows_per_split = (int)rps;
  *nsplit = (int)((T + rps - 1) / rps);
  return true;
}

static int backward_impl(const ultr_dnn_desc* d, const float* params, const float* features, int64_t n_docs,
                         const int32_t* docids, int32_t batch, int32_t list_size, const void* saved, const float* dscores,
                         const void* loss_ws, void* bwd_ws, float* grads, void* stream, FusedSoftmax fl,
                         int fused_rb = 0 /* > 0: the fused forward+backward kernel ran with this many rows per block;
                                             only the weight gradients and the reduction remain */) {
  if (!params || !docids || !saved || (!dscores && !fl.scores) || !bwd_ws || !grads || batch <= 0 || list_size <= 0 ||
      n_docs < 0 || (n_docs > 0 && !features))
    return ULTR_E_BADARG;
  const int64_t N = (int64_t)batch * list_size;
  DnnPlan p;
  BwdPlan bp;
  if (!ultr_make_dnn_plan(d, N, &p) || !ultr_make_bwd_plan(p, N, &bp)) return ULTR_E_BADARG;
  const bool l0g_ok = p.nl >= 2;  // the layer-0 shortcut whenever there is a hidden layer (its A/B knob of round 1 is gone: the explicit du_0 path had rotted)
  const int tail = (int)ultr_tail_len(list_size);
  if (tail > 4096) return ULTR_E_UNSUPPORTED;
  const size_t lds = bwd_lds_bytes(p, bp.rblk);
  const int nw = knobs().bwd_nw;
  const int vm = vecmask_for(p, params, features);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipSuccess;
  float* ws = (float*)bwd_ws;
  const bool av = all_vec(p, vm, N, n_docs) && knobs().no_vec == 0;
  if (bp.wg_h3 && !av) ultr_make_bwd_plan(p, N, &bp, 0);  // the split-half weight gradients take 16-byte paths only
  if (fused_rb > 0) bp.nrb = (int)((N + fused_rb - 1) / fused_rb);  // vector slabs / loss partials: one per fused block
  const bool big = fused_rb == 0 && dscores != nullptr && av && l0g_ok && ultr_dnn_big_ok(p, N, n_docs) &&
                   knobs().big_bwd != 0 && (big_bwd_wanted(p, N) || lds > 160 * 1024);
  if (lds > 160 * 1024 && !big) return ULTR_E_UNSUPPORTED;
  // fast variant: aligned shapes, K_j <= 512, 8 waves, LDS budget (see dnn_bwd2_kernel)
  const size_t lds2 = bwd2_lds_floats(p, bp.rblk, 8) * sizeof(float);
  const bool v2 = av && nw == 8 && p.maxdim <= 512 && lds2 <= 160 * 1024 && p.sv_total * 4 < ((int64_t)1 << 31) &&
                  p.P * 4 < ((int64_t)1 << 31);
  bp.l0g = l0g_ok ? 1 : 0;  // every backward kernel skips du_0; the wgrad launch makes up for it
  bp.wg_prenorm = (fused_rb > 0) ? 1 : 0;  // the fused kernel left the ready-made wgrad operands in `saved`
  WideBwd wb;
  size_t wblds = 0;
  const bool wide = fused_rb == 0 && dscores != nullptr && av && l0g_ok && g_ultr_step_wt != nullptr && ((uintptr_t)g_ultr_step_wt & 15) == 0 &&
                    bwd_wide_plan(p, N, &wb, &wblds);
  if (fused_rb > 0) {
    // the row-local half already ran inside dnn_fb_kernel
  } else if (wide) {
    UltrProfScope prof(ULTR_K_BWD, st);
    bp.rblk = wb.R;
    bp.nrb = (int)((N + wb.R - 1) / wb.R);  // one vector slab per workgroup
    const int rcw = ultr_launch_dnn_bwdw(prof, p, bp, wb, wblds, st, (const float*)saved, dscores, ws, g_ultr_step_wt);
    if (rcw) return rcw;
  } else if (big) {
    UltrProfScope prof(ULTR_K_BWD, st);
    bp.nrb = (int)((N + ULTR_BIG_ROWS - 1) / ULTR_BIG_ROWS);  // one vector slab per row block of the row kernels
    const int rc = ultr_dnn_big_backward(p, bp, params, (const float*)saved, dscores, ws, st, prof.on ? prof.a : nullptr,
                                         prof.on ? prof.b : nullptr, knobs().bwd_h3 != 0 && !p.no_h3 && g_ultr_step_wt != nullptr);
    if (rc) return rc;
  } else if (v2) {
    UltrProfScope prof(ULTR_K_BWD, st);
    const int rc2 = ultr_launch_dnn_bwd2(prof, p, bp, lds2, st, params, features, n_docs, docids, (int)batch, (int)list_size, (const float*)saved,
                                         dscores, ws, fl, g_ultr_step_wt);
    if (rc2) return rc2;
  } else {
    UltrProfScope prof(ULTR_K_BWD, st);
    const int rc1 = ultr_launch_dnn_bwd(prof, p, bp, nw, av, lds, st, params, features, n_docs, docids, (int)batch, (int)list_size, (const float*)saved,
                                        dscores, ws, vm, fl);
    if (rc1) return rc1;
  }
  e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  const float* lp = (const float*)loss_ws;
  const int nlp = fl.scores ? bp.nrb : (int)ultr_loss_parts(batch);
  if (nlp > 1024) {
    int len = (nlp + 63) / 64;
    len = len < 1024 ? 1024 : (len + 31) / 32 * 32;
    bp.lf_len = len;
    bp.lf_chunks = (nlp + len - 1) / len;
  }
  if (bp.wg_h3) {
    UltrProfScope prof(ULTR_K_WGRAD, st);
    const dim3 wgrid(bp.wgrad_blocks + bp.vred_blocks + (bp.lf_chunks > 0 ? bp.lf_chunks : 1));
    EarlyReport er = g_ultr_early;
    CommDev cd;
    memset(&cd, 0, sizeof(cd));
    if (g_ultr_step_xchg.comm != nullptr && g_ultr_step_xchg.er.host != nullptr && bp.lf_chunks == 0 &&
        ultr_comm_dev(g_ultr_step_xchg.comm, g_ultr_step_xchg.step, p.P + tail, &cd)) {
      er = g_ultr_step_xchg.er;  // (see the register kernel's launch below)
      g_ultr_step_xchg.er.host = nullptr;
    } else {
      cd.world = 0;
    }
    if (bp.lf_chunks > 0) er.host = nullptr;
    const int rcg = ultr_launch_dnn_wgrad(prof, p, bp, av, true, 0, wgrid, st, params, features, n_docs, docids, (int)batch, (int)list_size,
                                          (const float*)saved, ws, 0, grads, lp, nlp, tail, er, cd);
    if (rcg) return rcg;
  } else {
    UltrProfScope prof(ULTR_K_WGRAD, st);
    int maxrps = 0;
    for (int j = 0; j < p.nl - 1; ++j) maxrps = bp.wl[j].rows_per_split > maxrps ? bp.wl[j].rows_per_split : maxrps;
    const size_t wlds = (size_t)(4 * 64 * 64 + 4 * 64 + maxrps) * sizeof(float);
    const dim3 wgrid(bp.wgrad_blocks + bp.vred_blocks + (bp.lf_chunks > 0 ? bp.lf_chunks : 1));
    EarlyReport er = g_ultr_early;
    CommDev cd;
    memset(&cd, 0, sizeof(cd));  // world 0: not a data-parallel step
    if (g_ultr_step_xchg.comm != nullptr && g_ultr_step_xchg.er.host != nullptr && bp.lf_chunks == 0 &&
        ultr_comm_dev(g_ultr_step_xchg.comm, g_ultr_step_xchg.step, p.P + tail, &cd)) {
      // data-parallel step: the workgroup that folds the loss partials exchanges the head of the tail with the peers and reports
      // the loss NOW; the gradient exchange behind this launch then has nothing to report
      er = g_ultr_step_xchg.er;
      g_ultr_step_xchg.er.host = nullptr;
    } else {
      cd.world = 0;
    }
    if (bp.lf_chunks > 0) er.host = nullptr;  // two-level fold of > 1024 partials: the loss is only final in the reduction launch
    const int rcg = ultr_launch_dnn_wgrad(prof, p, bp, av, false, wlds, wgrid, st, params, features, n_docs, docids, (int)batch, (int)list_size,
                                          (const float*)saved, ws, (vm >> 31) & 1, grads, lp, nlp, tail, er, cd);
    if (rcg) return rcg;
  }
  RedPlan rp;
  ultr_make_red_plan(p, bp, &rp);
  const int nblk = (int)ultr_red_blocks(p.P, tail);
  UltrProfScope prof(ULTR_K_REDUCE, st);
  // one thread per element (full_sum: any number of slabs, the same bits as the cooperating groups) as long as the BIG segments -
  // the weight slabs - have at most 32 parts; the layer-0 gamma / beta partials (2 K_0 elements, one part per 64-row block of W_0
  // and row split: 56 at config 4 with the split-half launch's 7 splits) may have up to 128
  int maxparts = 1;
  for (int k = 0; k < rp.nseg; ++k) {
    const int np = rp.seg[k].len > 4096 ? rp.seg[k].nparts : (rp.seg[k].nparts + 3) / 4;
    maxparts = np > maxparts ? np : maxparts;
  }
  if (g_ultr_step_xchg.comm != nullptr && maxparts <= 32 && bp.lf_chunks == 0) {
    // data-parallel step: this launch exchanges its own output (grad_reduce_xchg_kernel); ultr_train_step then skips the exchange kernel
    CommDev cd;
    if (ultr_comm_dev(g_ultr_step_xchg.comm, g_ultr_step_xchg.step, p.P + tail, &cd)) {
      int nx = 0;
      const int rcx = ultr_launch_grad_reduce_xchg(prof, rp, p, bp, tail, nblk, cd, g_ultr_step_xchg.er, st, ws, grads, &nx);
      if (rcx) return rcx;
      g_ultr_step_xchg.done = true;
      g_ultr_step_nsq2 = nx;
      return 0;
    }
  }
  int nsq2 = 0;
  const int rcr = ultr_launch_grad_reduce(prof, rp, p, bp, tail, nblk, maxparts, st, ws, grads, &nsq2);
  if (rcr) return rcr;
  if (nsq2 > 0) g_ultr_step_nsq2 = nsq2;
  return (int)hipGetLastError();
}

extern "C" int ultr_dnn_backward(const ultr_dnn_desc* d, const float* params, const float* features, int64_t n_docs,
                                 const int32_t* docids, int32_t batch, int32_t list_size, const void* saved,
                                 const float* dscores, const void* loss_ws, void* bwd_ws, float* grads, void* stream) {
  FusedSoftmax fl = {nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr};
  return backward_impl(d, params, features, n_docs, docids, batch, list_size, saved, dscores, loss_ws, bwd_ws, grads, stream, fl);
}

extern "C" int ultr_dnn_backward_softmax(const ultr_dnn_desc* d, const float* params, const float* features, int64_t n_docs,
                                         const int32_t* docids, int32_t batch, int32_t list_size, const void* saved,
                                         const float* scores, const float* labels, const float* pw, const float* ipw_table,
                                         int32_t n_ipw, float* dscores_out, void* loss_ws, void* bwd_ws, float* grads,
                                         void* stream) {
  if (!scores || !labels || !loss_ws || (ipw_table && n_ipw <= 0)) return ULTR_E_BADARG;
  {
    // big batches take the per-layer backward, which wants the loss as its own stage
    DnnPlan p;
    const int64_t N = (int64_t)batch * list_size;
    // ... and so do shapes whose row tile does not fit the LDS (a layer wider than 512 at a small batch): the same condition
    // backward_impl applies to the other algorithms - without it this entry point returned ULTR_E_UNSUPPORTED there
    BwdPlan bp0;
    const bool planned = dscores_out && batch > 0 && list_size > 0 && ultr_make_dnn_plan(d, N, &p) && ultr_make_bwd_plan(p, N, &bp0);
    // ... and the wide-tile backward of ultr_train_step (dnn_bwdw_kernel), which takes dscores too
    WideBwd wb0;
    size_t wl0 = 0;
    const bool wide = planned && g_ultr_step_wt != nullptr && knobs().no_vec == 0 && bwd_wide_plan(p, N, &wb0, &wl0);
    if (planned && (wide || ((big_bwd_wanted(p, N) || bwd_lds_bytes(p, bp0.rblk) > 160 * 1024) && knobs().big_bwd != 0 &&
        knobs().no_vec == 0 && ultr_dnn_big_ok(p, N, n_docs)))) {
      const int rc = ultr_softmax_ce(scores, labels, pw, ipw_table, n_ipw, batch, list_size, dscores_out, loss_ws, stream);
      if (rc) return rc;
      FusedSoftmax none = {nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr};
      return backward_impl(d, params, features, n_docs, docids, batch, list_size, saved, dscores_out, loss_ws, bwd_ws, grads, stream, none);
    }
  }
  FusedSoftmax fl = {scores, labels, pw, ipw_table, (int)n_ipw, dscores_out, (float*)loss_ws};
  return backward_impl(d, params, features, n_docs, docids, batch, list_size, saved, nullptr, loss_ws, bwd_ws, grads, stream, fl);
}

// internal (ultr_train_step): forward + NA/IPW loss + backward for a small batch through dnn_fb_kernel, then the weight
// gradients + reduction.  Returns ULTR_E_UNSUPPORTED when the shape does not qualify - the caller then issues the
// separate forward / backward calls.
int ultr_fused_step_softmax(const ultr_dnn_desc* d, const float* params, const float* wt, const float* features, int64_t n_docs,
                            const int32_t* docids, int32_t batch, int32_t list_size, float* scores, void* saved,
                            const float* labels, const float* pw, const float* ipw_table, int32_t n_ipw, float* dscores_out,
                            void* loss_ws, void* bwd_ws, float* grads, void* stream) {
  if (!params || !wt || !docids || !scores || !saved || !labels || !loss_ws || !bwd_ws || !grads || batch <= 0 ||
      list_size <= 0 || n_docs < 0 || (n_docs > 0 && !features) || (ipw_table && n_ipw <= 0))
    return ULTR_E_BADARG;
  if (knobs().no_fused_fb != 0) return ULTR_E_UNSUPPORTED;
  const int L = list_size;
  if (L > 16) return ULTR_E_UNSUPPORTED;
  const int64_t N = (int64_t)batch * L;
  DnnPlan p;
  BwdPlan bp;
  if (!ultr_make_dnn_plan(d, N, &p) || !ultr_make_bwd_plan(p, N, &bp)) return ULTR_E_BADARG;
  const int lpb = 16 / L, rb = lpb * L;
  const int64_t nblk = (batch + lpb - 1) / lpb;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      cus = 256;
  }
  const int vm = vecmask_for(p, params, features);
  const size_t lds = fb_lds_floats(p) * sizeof(float);
  const bool ok = all_vec(p, vm, N, n_docs) && knobs().no_vec == 0 && ((uintptr_t)wt & 15) == 0 &&
                  p.nl >= 2 && p.maxdim <= 512 && p.pv_total <= 3 * 512 * 4 && lds <= 160 * 1024 &&
                  nblk <= (int64_t)knobs().fb_max_wg_per_cu * cus &&  // one round of workgroups (tools/fused_threshold.py:
                                                                          // B=256 62 vs 68 us, B=288 94 vs 70 us)
                  nblk <= (N + 8) / 9 + 1 &&                               // vector-slab allocation (>= 9 live rows / block)
                  p.sv_total * 4 < ((int64_t)1 << 31) && p.P * 4 < ((int64_t)1 << 31);
  if (!ok) return ULTR_E_UNSUPPORTED;
  bp.nrb = (int)nblk;
  bp.l0g = p.nl >= 2 ? 1 : 0;  // must match backward_impl's choice
  bp.wg_prenorm = 1;
  hipStream_t st = (hipStream_t)stream;
  FusedSoftmax fl = {nullptr, labels, pw, ipw_table, (int)n_ipw, dscores_out, (float*)loss_ws};
  float* ws = (float*)bwd_ws;
  FbPlan fp;
  memset(&fp, 0, sizeof(fp));
  for (int j = 0; j < p.nl; ++j) {
    int* r = fp.rec[j];
    auto put64 = [&](int k, int64_t v) { r[k] = (int)(uint32_t)(uint64_t)v; r[k + 1] = (int)(uint32_t)((uint64_t)v >> 32); };
    r[FbPlan::K] = p.K[j]; r[FbPlan::M] = p.M[j]; r[FbPlan::PV_OFF] = p.pv_off[j];
    r[FbPlan::KSPLIT] = p.fl[j].ksplit; r[FbPlan::KLEN] = p.fl[j].klen; r[FbPlan::NCH] = p.fl[j].nch;
    r[FbPlan::BWD_NCH] = p.bwd_nch[j]; r[FbPlan::BWD_MSPLIT] = p.bwd_msplit[j]; r[FbPlan::BWD_MODE] = p.bwd_mode[j];
    r[FbPlan::BWD_MLEN] = p.bwd_mlen[j]; r[FbPlan::VOFF_G] = bp.voff_g[j]; r[FbPlan::VOFF_B] = bp.voff_b[j];
    put64(FbPlan::WSF_OFF, p.wsf_off[j]); put64(FbPlan::WSB_OFF, p.wsb_off[j]); put64(FbPlan::SV_X, p.sv_x[j]);
    put64(FbPlan::OFF_W, p.off_w[j]); put64(FbPlan::SV_MEAN, p.sv_mean[j]); put64(FbPlan::SV_RSTD, p.sv_rstd[j]);
    put64(FbPlan::DZ_OFF, j < p.nl - 1 ? bp.dz_off[j] : 0); put64(FbPlan::WT_OFF, p.wt_off[j]);
    put64(FbPlan::WHF_OFF, p.whf_off[j]); put64(FbPlan::WHB_OFF, p.whb_off[j]);
  }
  hipError_t e;
  {
    UltrProfScope prof(ULTR_K_FUSED, st);
    const int rcf = ultr_launch_dnn_fb(prof, p, bp, lds, nblk, st, params, wt, features, n_docs, docids, (int)batch, L, lpb, scores, (float*)saved, ws,
                                       fl, fp);
    if (rcf) return rcf;
  }
  e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  fl.scores = scores;  // marks "loss partials come one per row block" for the reduction
  return backward_impl(d, params, features, n_docs, docids, batch, list_size, saved, nullptr, loss_ws, bwd_ws, grads, stream, fl,
                       rb);
}

