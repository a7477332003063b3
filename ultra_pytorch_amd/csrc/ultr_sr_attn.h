// ultr_sr_attn.h - SetRank's self-attention launches (kernels in ultr_sr_attn.hip; called from ultr_setrank_forward / _backward).
// No projections: q = k = v = x[:, head slice] (SetRank.py:33-37, 54-66, 159-195).
#pragma once
#include <hip/hip_runtime.h>

struct SrAttnShape {
  int d, dh, H;   // d_model, head depth, heads
  int att_f16;    // ultr_setrank_desc.attention_dtype == ULTR_ATTN_FP16
};
// the matrix-core kernels take this shape (list_size <= 128, head depth 16 / 32 / 64, 16-byte aligned head slices); else the scalar kernels
bool sr_attn_mfma_ok(const SrAttnShape& s, int L);
// 0 or ULTR_E_UNSUPPORTED: can the forward (backward = 0) / the backward (backward = 1) run at this list size
int sr_attn_supported(const SrAttnShape& s, int L, int backward);
// A[T, d] = softmax(x x^T / sqrt(dh)) x per (list, head), lse[T, H] the row statistic the backward reads
int sr_attn_forward(const SrAttnShape& s, const float* x, int batch, int L, float* A, float* lse, hipStream_t st);
// dx += the attention path's gradient; split_half: the caller's knobs (ULTR_SR_ATTN_H3 and its per-layer mask) allow the split-half kernel
int sr_attn_backward(const SrAttnShape& s, bool split_half, const float* x, const float* dA, const float* Aout, const float* lse, int batch, int L,
                     float* dx, hipStream_t st);
