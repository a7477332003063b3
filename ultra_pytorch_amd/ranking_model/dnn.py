"""DNN / Linear ranking models — drop-in for ultra.ranking_model.DNN / ultra.ranking_model.Linear.

Same constructor `(hparams_str, feature_size)`, same `build(input_list, ...)` contract and the same
`state_dict()` keys (`sequential.layer_norm{j}.{weight,bias}`, `sequential.linear{j}.{weight,bias}`,
reference DNN.py:41-55) so checkpoints interchange with the reference.  All parameters are views into ONE
flat fp32 tensor (`flat_params`) laid out in that order — the layout the HIP kernels consume — so the
optimizer kernel updates the module in place with no copies.  `build()` runs the fused HIP forward
(ultr_dnn_forward); there is no CPU path.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import hip_ops
from ..utils.hparams import HParams


def init_flat_params(shape, seed=None):
    """nn.LayerNorm / nn.Linear default initialisation (what DNN.__init__ gets, DNN.py:44-52: gamma 1, beta 0,
    weight and bias U(-1/sqrt(fan_in), 1/sqrt(fan_in))) written into the flat layout.  Parity tests never rely
    on it (they load golden weights)."""
    g = None  # seed None: the global torch RNG, like nn.Linear's own initialisation (torch.manual_seed controls it)
    if seed is not None:
        g = torch.Generator()
        g.manual_seed(int(seed))
    flat = torch.empty(shape.n_params, dtype=torch.float32)
    for j, (k, m) in enumerate(shape.dims):
        o_g, o_b, o_w, o_c = shape.offsets[4 * j:4 * j + 4]
        bound = 1.0 / math.sqrt(k)
        flat[o_g:o_g + k] = 1.0
        flat[o_b:o_b + k] = 0.0
        flat[o_w:o_w + m * k] = (torch.rand(m * k, generator=g) * 2 - 1) * bound
        flat[o_c:o_c + m] = (torch.rand(m, generator=g) * 2 - 1) * bound
    return flat


class DNN(nn.Module):
    """[LayerNorm -> Linear -> act] x k -> LayerNorm -> Linear(., 1)   (LayerNorm precedes EVERY Linear)."""

    DEFAULT_HIDDEN = [512, 256, 128]

    def __init__(self, hparams_str, feature_size):
        super().__init__()
        self.hparams = HParams(hidden_layer_sizes=list(self.DEFAULT_HIDDEN), activation_func="elu", norm="layer")
        self.hparams.parse(hparams_str)
        if self.hparams.norm != "layer":
            raise NotImplementedError("norm=%r: only 'layer' is supported (the reference's 'batch' branch builds "
                                      "BatchNorm2d on 2-D input and cannot run, DNN.py:48-50)" % self.hparams.norm)
        if self.hparams.activation_func == "selu":  # a plain function in ACT_FUNC_DIC: nn.Sequential.add_module raises (DNN.py:52-53)
            raise TypeError("activation_func='selu' raises in the reference too (base_ranking_model.selu is not a Module subclass)")
        if self.hparams.activation_func not in ("elu", "relu", "tanh", "sigmoid"):
            raise NotImplementedError("activation_func=%r: base_ranking_model.py:63-69 knows elu, relu, tanh, sigmoid"
                                      % self.hparams.activation_func)
        self.feature_size = int(feature_size)
        self._init_shape(list(self.hparams.hidden_layer_sizes))

    def _init_shape(self, hidden):
        self.shape = hip_ops.DnnShape(self.feature_size, hidden, self.hparams.activation_func)
        self.output_sizes = hidden + [1]
        flat = init_flat_params(self.shape)
        self._bind(flat)
        self._fwd_cache = {}

    def _bind(self, flat):
        """(Re)create the parameter views over `flat` (device moves re-bind)."""
        self.flat_params = flat
        self.sequential = nn.Module()
        mods = {}
        for name, shp, off in self.shape.layout():
            _, mod, leaf = name.split(".")
            n = int(np.prod(shp))
            p = nn.Parameter(flat[off:off + n].view(*shp), requires_grad=False)
            mods.setdefault(mod, nn.Module()).register_parameter(leaf, p)
        for mod, m in mods.items():
            self.sequential.add_module(mod, m)

    def _apply(self, fn, *a, **k):
        # keep every parameter a view of ONE flat tensor across .to()/.cuda()
        flat = fn(self.flat_params.detach())
        self._bind(flat.contiguous().to(torch.float32))
        self._fwd_cache = {}
        return self

    def invalidate_weight_copy(self):
        """Call after writing parameters behind torch's back (`.data`, raw pointers, DLPack): the forward kernels read a
        k-major COPY of the hidden weights that is rebuilt only when torch's version counter moves (hip_ops.WeightCopy)."""
        hip_ops.weight_copy(self.shape).invalidate()

    def load_state_dict(self, state_dict, strict=True):
        own = dict(self.state_dict())
        missing = [k for k in own if k not in state_dict]
        unexpected = [k for k in state_dict if k not in own]
        if strict and (missing or unexpected):
            raise RuntimeError("load_state_dict: missing %s unexpected %s" % (missing, unexpected))
        with torch.no_grad():
            for k, v in state_dict.items():
                if k in own:
                    own[k].copy_(v.to(own[k].device, torch.float32))
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def build(self, input_list, noisy_params=None, noise_rate=0.05, is_training=False, **kwargs):
        """Scores for a list of L tensors [B, F] -> sequence of L tensors [B, 1]   (DNN.py:58-88)."""
        if noisy_params is not None:
            raise NotImplementedError("noisy_params is the online-learning path of the reference (DNN.py:76-86)")
        if not self.flat_params.is_cuda:
            raise RuntimeError("ultra_pytorch_amd.ranking_model.DNN.build needs the model on the GPU "
                               "(model.cuda()); there is no CPU fallback")
        dev = self.flat_params.device
        L, B = len(input_list), int(input_list[0].shape[0])
        x = torch.cat([t.to(dev, torch.float32) for t in input_list], dim=0).contiguous()  # position-major rows
        docids = torch.arange(L * B, dtype=torch.int32, device=dev)  # row l*B+b == docids[l, b]
        scores = torch.empty(B, L, dtype=torch.float32, device=dev)
        hip_ops.dnn_forward(self.shape, self.flat_params, x, L * B, docids, B, L, scores, None)
        return torch.split(scores.t().contiguous().view(L * B, 1), B, dim=0)


class Linear(DNN):
    """ultra.ranking_model.Linear: LayerNorm -> Linear(F, 1), the k = 0 case of the DNN."""

    def __init__(self, hparams_str, feature_size):
        nn.Module.__init__(self)
        self.hparams = HParams(activation_func="elu", norm="layer")
        self.hparams.parse(hparams_str)
        self.feature_size = int(feature_size)
        self._init_shape([])
