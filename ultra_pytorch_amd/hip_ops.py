"""Tensor-level wrappers over the C ABI: torch supplies device memory and the stream, nothing else.

Every function takes CUDA(ROCm) float32 / int32 tensors, passes raw pointers + the current
HIP stream to libultr_hip.so and returns immediately (asynchronous w.r.t. the host).
"""
import ctypes
import os
import warnings
import weakref

import torch

from . import _lib
from ._lib import check


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def raw_stream():
    """The current HIP stream's handle as an int: torch's own C getter (what torch.cuda.current_stream().cuda_stream resolves to,
    without building a Stream object and normalising the device argument: ~3 us -> ~0.3 us per call)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _stream():
    return ctypes.c_void_p(raw_stream())


def _req(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise TypeError("%s must be a contiguous %s tensor on the GPU" % (name, dtype))
    return t


class DnnShape:
    """Host-side geometry of one DNN (desc + parameter layout), mirrors DNN.__init__ (DNN.py:25-55)."""

    def __init__(self, feature_size, hidden, activation="elu"):
        self.lib = _lib.load()
        self.feature_size = int(feature_size)
        self.hidden = [int(h) for h in (hidden or [])]
        self.activation = activation
        self.desc = _lib.make_desc(feature_size, self.hidden, activation)
        self.n_params = int(self.lib.ultr_dnn_param_count(ctypes.byref(self.desc)))
        if self.n_params <= 0:
            raise ValueError("bad DNN description")
        nl = len(self.hidden) + 1
        offs = (ctypes.c_int64 * (4 * nl))()
        check(self.lib.ultr_dnn_param_offsets(ctypes.byref(self.desc), offs), "ultr_dnn_param_offsets")
        self.offsets = list(offs)
        dims, k = [], self.feature_size
        for m in self.hidden + [1]:
            dims.append((k, m))
            k = m
        self.dims = dims

    def layout(self):
        """[(state_dict key, shape, offset)] in the reference's parameter order."""
        out = []
        for j, (k, m) in enumerate(self.dims):
            o = self.offsets[4 * j:4 * j + 4]
            out += [("sequential.layer_norm%d.weight" % j, (k,), o[0]), ("sequential.layer_norm%d.bias" % j, (k,), o[1]),
                    ("sequential.linear%d.weight" % j, (m, k), o[2]), ("sequential.linear%d.bias" % j, (m,), o[3])]
        return out

    def saved_bytes(self, n_rows):
        return int(self.lib.ultr_dnn_saved_bytes(ctypes.byref(self.desc), n_rows))

    def bwd_workspace_bytes(self, n_rows):
        return int(self.lib.ultr_dnn_bwd_workspace_bytes(ctypes.byref(self.desc), n_rows))


class SetRankShape:
    """Host-side geometry of one SetRank model: descriptor + flat parameter layout in the reference's
    state_dict order (SetRank.py:95-103, 130-141)."""

    def __init__(self, feature_size, d_model=256, num_heads=8, num_layers=2, dff=64, attention_dtype="fp32"):
        self.lib = _lib.load()
        self.feature_size, self.d_model, self.num_heads = int(feature_size), int(d_model), int(num_heads)
        self.num_layers, self.dff = int(num_layers), int(dff)
        if attention_dtype not in _lib.ATTN_DTYPE:
            raise ValueError("attention_dtype must be one of %s (got %r)" % (sorted(_lib.ATTN_DTYPE), attention_dtype))
        self.attention_dtype = attention_dtype
        self.desc = _lib.SetRankDesc(self.feature_size, self.d_model, self.num_heads, self.num_layers, self.dff,
                                     _lib.ATTN_DTYPE[attention_dtype])
        self.n_params = int(self.lib.ultr_setrank_param_count(ctypes.byref(self.desc)))
        if self.n_params <= 0:
            raise ValueError("bad SetRank description (d_model must be a multiple of num_heads, 1..8 layers)")

    def layout(self):
        F, d, dff = self.feature_size, self.d_model, self.dff
        out, off = [], 0

        def add(name, shape):
            nonlocal off
            out.append((name, tuple(shape), off))
            n = 1
            for v in shape:
                n *= v
            off += n

        e = "Encoder_layer."
        add(e + "input_layer_norm.weight", (F,)); add(e + "input_layer_norm.bias", (F,))
        add(e + "input_embedding.0.weight", (dff, F)); add(e + "input_embedding.0.bias", (dff,))
        add(e + "input_embedding.2.weight", (d, dff)); add(e + "input_embedding.2.bias", (d,))
        add(e + "output_layer.0.weight", (dff, d)); add(e + "output_layer.0.bias", (dff,))
        add(e + "output_layer.2.weight", (1, dff)); add(e + "output_layer.2.bias", (1,))
        for i in range(self.num_layers):
            l = e + "enc_layers.encoder%d." % i
            add(l + "mha.dense.weight", (d, d)); add(l + "mha.dense.bias", (d,))
            add(l + "ffn.0.weight", (dff, d)); add(l + "ffn.0.bias", (dff,))
            add(l + "ffn.2.weight", (d, dff)); add(l + "ffn.2.bias", (d,))
            add(l + "layernorm1.weight", (d,)); add(l + "layernorm1.bias", (d,))
            add(l + "layernorm2.weight", (d,)); add(l + "layernorm2.bias", (d,))
        assert off == self.n_params
        return out

    def saved_bytes(self, n_rows):
        return int(self.lib.ultr_setrank_saved_bytes(ctypes.byref(self.desc), n_rows))

    def workspace_bytes(self, n_rows):
        return int(self.lib.ultr_setrank_workspace_bytes(ctypes.byref(self.desc), n_rows))

    def range_flag_offset(self, n_rows):
        """Float offset in `saved` of the range word of the split-half planes (ultr_update_desc::range_flag)."""
        return int(self.lib.ultr_setrank_range_flag_offset(ctypes.byref(self.desc), n_rows))


def setrank_forward(shape, params, features, n_docs, docids, B, L, scores, saved):
    check(shape.lib.ultr_setrank_forward(ctypes.byref(shape.desc), _p(params), _p(features), int(n_docs), _p(docids), int(B),
                                         int(L), _p(scores), _p(saved), _stream()), "ultr_setrank_forward")


def setrank_backward(shape, params, B, L, saved, dscores, loss_ws, n_loss_parts, ws, grads):
    check(shape.lib.ultr_setrank_backward(ctypes.byref(shape.desc), _p(params), int(B), int(L), _p(saved), _p(dscores), _p(loss_ws),
                                          int(n_loss_parts), _p(ws), _p(grads), _stream()), "ultr_setrank_backward")


def tail_floats(L):
    return int(_lib.load().ultr_step_tail_floats(int(L)))


def loss_workspace_bytes(B, L):
    return int(_lib.load().ultr_loss_workspace_bytes(int(B), int(L)))


def loss_part_count(B):
    """Step-tail partials a stand-alone loss kernel writes for B lists."""
    return int(_lib.load().ultr_loss_part_count(int(B)))


H3_KNOBS = ("ULTR_FB_H3", "ULTR_FWD_H3", "ULTR_BWD_H3")
SR_H3_KNOBS = ("ULTR_SR_H3",)


def knob_on(name, default=1):
    """An integer knob as the LIBRARY reads it (csrc: atoi of the environment string - 'false' or 'off' parse as 0 there, so they
    do here; an unset or empty variable is the default)."""
    v = os.environ.get(name, "")
    if v == "":
        return default != 0
    v = v.strip()
    sign = -1 if v.startswith("-") else 1
    digits = ""
    for ch in v.lstrip("+-"):
        if not ch.isdigit():
            break
        digits += ch
    return (sign * int(digits) if digits else 0) != 0


def split_half_enabled(shape=None):
    """Does any split-half (fp16 hi / lo) product read THIS model's weights?  Process-wide knobs AND the model's own switch
    (ultr_dnn_desc / ultr_setrank_desc ::flags)."""
    if shape is not None and (int(shape.desc.flags) & _lib.MODEL_FP32_PRODUCTS):
        return False
    knobs = SR_H3_KNOBS if isinstance(shape, SetRankShape) else H3_KNOBS
    return any(knob_on(k) for k in knobs)


def fall_back_to_fp32_products(shape, why):
    """Switch THIS model's products from the split-half (fp16 hi / lo) weight copies to the fp32 matrix cores - a flag in the
    model's descriptor (ULTR_MODEL_FP32_PRODUCTS), read by every later call that takes the descriptor; other models of the
    process keep their plan and nothing is written to the environment.  The reference computes in fp32 at any weight
    magnitude (DNN.py:58-88, base_algorithm.py:208-226); the copies cover |w| < 128 only.  In a data-parallel run the replicas
    are bit-identical and read their step reports in lockstep, so every rank switches at the same step.  Returns True when
    the model was on the split-half plan until now."""
    if not split_half_enabled(shape):
        return False
    shape.desc.flags = int(shape.desc.flags) | _lib.MODEL_FP32_PRODUCTS
    warnings.warn("ultra_pytorch_amd: %s - the products of this model now run on the fp32 matrix cores (ULTR_MODEL_FP32_PRODUCTS): "
                  "same results to the 1e-5 parity bar, a few per cent slower steps" % why, RuntimeWarning, stacklevel=3)
    return True


class WeightCopy:
    """The k-major copy of the hidden Linear weights the fast forward reads (ultr_dnn_build_wt).  Rebuilt whenever
    torch reports an in-place modification of `params` (load_state_dict, .copy_, init); ultr_apply_update keeps it
    current on its own (it writes through raw pointers, which does not bump the tensor version)."""

    def __init__(self, shape):
        self.shape = shape
        self.n = int(shape.lib.ultr_dnn_wt_floats(ctypes.byref(shape.desc)))
        self.wt, self.key, self.ref = None, None, None
        self.check = os.environ.get("ULTR_CHECK_WT", "0") == "1"

    def invalidate(self):
        """Force a rebuild on the next use.  torch's version counter sees every in-place tensor op (copy_, load_state_dict,
        optimizers), but NOT writes through `.data`, raw pointers (other than this library's own update kernel, which
        maintains the copy itself) or DLPack aliases: whoever writes the parameters that way must call this
        (`ranking_model.DNN.invalidate_weight_copy()`); ULTR_CHECK_WT=1 verifies the copy on every use (debug, synchronises)."""
        self.key = None

    def _verify(self, params):
        fresh = torch.empty_like(self.wt)
        check(self.shape.lib.ultr_dnn_build_wt(ctypes.byref(self.shape.desc), _p(params), _p(fresh), _stream()), "ultr_dnn_build_wt")
        if not torch.equal(fresh, self.wt):
            raise RuntimeError("stale k-major weight copy: the parameters were written behind torch's version counter "
                               "(.data / raw pointer); call invalidate_weight_copy() after such writes")

    def get(self, params):
        # identity of the tensor OBJECT (weakref: a freed tensor's address and version 0 can both be reused)
        key = (params.data_ptr(), params._version)
        same = self.ref is not None and self.ref() is params and key == self.key
        if self.wt is None or self.wt.device != params.device:
            self.wt = torch.zeros(self.n, dtype=torch.float32, device=params.device)  # alignment gaps stay zero
            same = False
        if not same:
            check(self.shape.lib.ultr_dnn_build_wt(ctypes.byref(self.shape.desc), _p(params), _p(self.wt), _stream()),
                  "ultr_dnn_build_wt")
            self.key, self.ref = key, weakref.ref(params)
            # parameters that arrived from outside (checkpoint, init, a test) may be out of the split-half copies' range: look
            # once (a stream synchronisation - this branch runs after external writes only) and switch to the fp32 products
            # BEFORE a kernel reads the copies; training drift is caught by the step report instead (engine.StepEngine)
            if split_half_enabled(self.shape):
                rng = int(self.shape.lib.ultr_dnn_wt_range(ctypes.byref(self.shape.desc), _p(self.wt), _stream()))
                if rng < 0:
                    check(rng, "ultr_dnn_wt_range")
                if rng > 0:
                    fall_back_to_fp32_products(self.shape, "a hidden weight of magnitude >= 64 was loaded (the split-half weight "
                                               "copies cover |w| < 128)")
        elif self.check:
            self._verify(params)
        return self.wt


def weight_copy(shape):
    if getattr(shape, "_wcopy", None) is None:
        shape._wcopy = WeightCopy(shape)
    return shape._wcopy


def dnn_forward(shape, params, features, n_docs, docids, B, L, scores, saved=None, wt=None):
    lib = shape.lib
    _req(params, torch.float32, "params"), _req(docids, torch.int32, "docids"), _req(scores, torch.float32, "scores")
    if n_docs > 0:
        _req(features, torch.float32, "features")
    if wt is None:
        wt = weight_copy(shape).get(params)
    check(lib.ultr_dnn_forward(ctypes.byref(shape.desc), _p(params), _p(wt), _p(features) if n_docs > 0 else None,
                               int(n_docs), _p(docids), int(B), int(L), _p(scores), _p(saved), _stream()), "ultr_dnn_forward")


def dnn_backward(shape, params, features, n_docs, docids, B, L, saved, dscores, loss_ws, bwd_ws, grads):
    lib = shape.lib
    check(lib.ultr_dnn_backward(ctypes.byref(shape.desc), _p(params), _p(features) if n_docs > 0 else None, int(n_docs),
                                _p(docids), int(B), int(L), _p(saved), _p(dscores), _p(loss_ws), _p(bwd_ws), _p(grads),
                                _stream()), "ultr_dnn_backward")


def dnn_backward_softmax(shape, params, features, n_docs, docids, B, L, saved, scores, labels, loss_ws, bwd_ws, grads,
                         pw=None, ipw_table=None, dscores_out=None):
    n_ipw = 0 if ipw_table is None else int(ipw_table.numel())
    check(shape.lib.ultr_dnn_backward_softmax(ctypes.byref(shape.desc), _p(params), _p(features) if n_docs > 0 else None,
                                              int(n_docs), _p(docids), int(B), int(L), _p(saved), _p(scores), _p(labels), _p(pw),
                                              _p(ipw_table), n_ipw, _p(dscores_out), _p(loss_ws), _p(bwd_ws), _p(grads),
                                              _stream()), "ultr_dnn_backward_softmax")


def grad_sumsq(grads, n_params, L, bwd_ws):
    check(_lib.load().ultr_grad_sumsq(_p(grads), int(n_params), int(L), _p(bwd_ws), _stream()), "ultr_grad_sumsq")


def softmax_ce(scores, labels, B, L, dscores, loss_ws, pw=None, ipw_table=None):
    n_ipw = 0 if ipw_table is None else int(ipw_table.numel())
    check(_lib.load().ultr_softmax_ce(_p(scores), _p(labels), _p(pw), _p(ipw_table), n_ipw, int(B), int(L), _p(dscores),
                                      _p(loss_ws), _stream()), "ultr_softmax_ce")


def dla_loss(scores, labels, prop_params, logits_to_prob, B, L, dscores, loss_ws):
    check(_lib.load().ultr_dla_loss(_p(scores), _p(labels), _p(prop_params), int(logits_to_prob), int(B), int(L),
                                    _p(dscores), _p(loss_ws), _stream()), "ultr_dla_loss")


def pairdebias_loss(scores, labels, t_plus, t_minus, B, L, batch_total, dscores, loss_ws):
    check(_lib.load().ultr_pairdebias_loss(_p(scores), _p(labels), _p(t_plus), _p(t_minus), int(B), int(L),
                                           int(batch_total), _p(dscores), _p(loss_ws), _stream()), "ultr_pairdebias_loss")


def lambdarank_loss(scores, labels, t_plus, t_minus, sigma, B, L, dscores, loss_ws):
    check(_lib.load().ultr_lambdarank_loss(_p(scores), _p(labels), _p(t_plus), _p(t_minus), float(sigma), int(B), int(L),
                                           _p(dscores), _p(loss_ws), _stream()), "ultr_lambdarank_loss")


def regem_loss(scores, labels, propensity, B, L, dscores, loss_ws, uniforms=None, seed=0, step=0, pseudo_out=None):
    """RegressionEM estimation + BCE loss; uniforms [B, L] teacher-forces the Bernoulli draw."""
    check(_lib.load().ultr_regem_loss(_p(scores), _p(labels), _p(propensity), _p(uniforms), int(seed), int(step), int(B),
                                      int(L), _p(dscores), _p(pseudo_out), _p(loss_ws), _stream()), "ultr_regem_loss")


def apply_update(shape, udesc, params, state, grads, aux, bwd_ws, scalars):
    wt = weight_copy(shape).get(params)  # the update kernel writes the new weights into both layouts
    check(shape.lib.ultr_apply_update(ctypes.byref(udesc), ctypes.byref(shape.desc), _p(params), _p(wt), _p(state), _p(grads),
                                      _p(aux), _p(bwd_ws), _p(scalars), _stream()), "ultr_apply_update")


def ndcg(scores, labels, docids, n_docs, B, L, topn, ndcg_out, ndcg_ws, order_out=None, masked_out=None):
    arr = (ctypes.c_int32 * len(topn))(*[int(t) for t in topn])
    check(_lib.load().ultr_ndcg(_p(scores), _p(labels), _p(docids), int(n_docs), int(B), int(L), arr, len(topn),
                                _p(ndcg_out), _p(order_out), _p(masked_out), _p(ndcg_ws), _stream()), "ultr_ndcg")
