// ultr_sr_bwd.h - the row-local half of SetRank's backward as fused launches (round 6; kernels in ultr_sr_bwd.hip, called from
// ultr_setrank_backward).  All offsets are FLOAT offsets into the named buffer unless they say HALVES.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// the shapes the fused kernels are written for (BASELINE config 5's widths); everything else keeps the separate launches
#define SR_BWD_D 256
#define SR_BWD_DFF 64
#define SR_BWD_MAXWG 256  // persistent workgroups (one per CU): the partial regions are sized for this many

// sr_bwd_ffn_kernel: LayerNorm_2 backward + the FFN's backward of one encoder block (SetRank.py:103-111 backwards)
//   d s2 = LN2'(d x'),  d Wf2 += d s2^T f,  d f = (d s2 Wf2) o [f > 0],  d out1 = d s2 + d f Wf1
// per-workgroup partial (stride part_stride): [d Wf2 (d x dff) | d bf2 (d) | d g2 (d) | d b2 (d)]
struct SrBwdFfnArgs {
  int R, d, dff, ntiles;
  int64_t T;
  int64_t dy, dF, dx;        // workspace: d x' [T, d] in;  d f [T, dff] out;  d out1 [T, d] out
  int64_t s, mean, rstd, f;  // saved: s2, its statistics, f
  int64_t gamma;             // parameters: g2
  int64_t gt2, gt1;          // HALVES into the planes: fragment copies of Wf2^T (out dff, contraction d) and Wf1^T (out d, contraction dff)
  int64_t part, part_stride; // workspace
  int p0, p1, p2, os;        // dynamic LDS
};
// sr_bwd_proj_kernel: LayerNorm_1 backward + the attention projection's backward (SetRank.py:92-101 backwards)
//   d s1 = LN1'(d out1),  d Wf1 += d f^T out1 (out1 recomputed from s1),  d A = d s1 Wd
// per-workgroup partial: [d Wf1 (dff x d) | d bf1 (dff) | d bd (d) | d g1 (d) | d b1 (d)]
struct SrBwdProjArgs {
  int R, d, dff, ntiles;
  int64_t T;
  int64_t dy, dF, ds, dx;    // workspace: d out1 [T, d] in;  d f [T, dff] in;  d s1 [T, d] out;  d A [T, d] out (may be dy: in place)
  int64_t s, mean, rstd;     // saved: s1, its statistics
  int64_t gamma, beta;       // parameters: g1, b1
  int64_t gtd;               // HALVES: fragment copy of Wd^T
  int64_t part, part_stride;
  int p0, p1, pf, os;
};
// sr_bwd_head_kernel: the output FFN's backward; per-workgroup partial [d Wo1 (dff x d) | d bo1 (dff) | d wo2 (dff) | d bo2 | pad]
struct SrBwdHeadArgs {
  int R, d, dff, ntiles;
  int64_t T;
  int64_t dx;                // workspace: d x_nl [T, d] out
  int64_t x, oh;             // saved: x_nl, oh
  int64_t wo2;               // parameters
  int64_t gto1;              // HALVES: fragment copy of Wo1^T (out d, contraction dff)
  int64_t part, part_stride;
  int p0, p1, pf, os;
};
// sr_bwd_embed_kernel: the embedding FFN's and the input LayerNorm's backward; per-workgroup partial in the parameter vector's order
// [d g_in (F) | d b_in (F) | d W1 (dff x F) | d b1 (dff) | d W2 (d x dff) | d b2 (d)]
struct SrBwdEmbedArgs {
  int R, d, dff, F, ntiles;
  int64_t T;
  int64_t dy;                       // workspace: d x_0 [T, d] in
  int64_t h0, xg, mean, rstd;       // saved: h0, the gathered feature rows and their statistics
  int64_t g_in, b_in;               // parameters
  int64_t gt2, gt1;                 // HALVES: fragment copies of W2^T (out dff, contraction d) and W1^T (out F, contraction dff)
  int64_t part, part_stride;
  int p0, p1, p2, os;
};
// sr_fwd_block_kernel (ultr_sr_fwd.hip): everything of an encoder block behind the attention - the forward - in the same persistent geometry
struct SrFwdBlockArgs {
  int R, d, dff, ntiles, head, skip_out1;
  int64_t T;
  int64_t bd, bf1, bf2, g1, b1, g2, b2, bo1, wo2, bo2;   // parameters
  int64_t gd, gf1, gf2, go1;                             // HALVES: forward fragment copies of Wd, Wf1, Wf2, Wo1
  int64_t A, x, s1, m1, r1, out1, f, s2, m2, r2, xn, oh; // saved
  int p0, p1, p2, os;
};
int sr_fwd_block_launch(SrFwdBlockArgs a, int nwg, const float* params, const _Float16* planes, float* sv, float* scores, hipStream_t st);
// rows per tile / tiles / workgroups / LDS bytes for T rows on `cus` compute units; false: T too small or too large
bool sr_bwd_geometry(int64_t T, int cus, int* R, int* ntiles, int* nwg);
int sr_bwd_ffn_launch(SrBwdFfnArgs a, int nwg, const float* params, const _Float16* planes, const float* sv, float* ws, hipStream_t st);
int sr_bwd_head_launch(SrBwdHeadArgs a, int nwg, const float* params, const _Float16* planes, const float* sv, const float* dscores, float* ws,
                       hipStream_t st);
int sr_bwd_embed_launch(SrBwdEmbedArgs a, int nwg, const float* params, const _Float16* planes, const float* sv, float* ws, hipStream_t st);
int sr_bwd_proj_launch(SrBwdProjArgs a, int nwg, const float* params, const _Float16* planes, const float* sv, float* ws, hipStream_t st);
