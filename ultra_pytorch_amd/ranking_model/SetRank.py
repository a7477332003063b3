"""SetRank — permutation-invariant ranking model (Pang et al., SIGIR 2020); drop-in for
ultra.ranking_model.SetRank.SetRank (reference ranking_model/SetRank.py:197-255).

Same constructor `(hparams_str, feature_size)`, hparams (`d_model`, `num_heads`, `num_layers`, `diff`, `rate`,
`initializer`), `build(input_list, ...)` contract and `state_dict()` keys (`Encoder_layer.*`), so checkpoints
interchange.  All parameters are views into ONE flat fp32 tensor in state_dict order - the layout the HIP path
(ultr_setrank_forward / ultr_setrank_backward) consumes.  Quirks kept: no Q/K/V projections, no mask (PAD documents
attend and are attended), LayerNorm eps 1e-6, the shuffled index list of `build` is computed by the reference but never
used (SetRank.py:245-246) - dropped.  `rate` must stay 0.0 (the default): dropout is not implemented.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import engine, hip_ops
from ..utils.hparams import HParams


def init_setrank_params(shape, seed=None):
    """nn.LayerNorm / nn.Linear default initialisation in the flat layout (parity tests load golden weights)."""
    g = None  # seed None: the global torch RNG, like nn.Linear's own initialisation (torch.manual_seed controls it)
    if seed is not None:
        g = torch.Generator()
        g.manual_seed(int(seed))
    flat = torch.empty(shape.n_params, dtype=torch.float32)
    pending_bound = None
    for name, shp, off in shape.layout():
        n = int(np.prod(shp))
        if "layer_norm" in name or "layernorm" in name:
            flat[off:off + n] = 1.0 if name.endswith("weight") else 0.0
        elif name.endswith("weight"):
            pending_bound = 1.0 / math.sqrt(shp[1])
            flat[off:off + n] = (torch.rand(n, generator=g) * 2 - 1) * pending_bound
        else:
            flat[off:off + n] = (torch.rand(n, generator=g) * 2 - 1) * pending_bound
    return flat


class _Node(nn.Module):
    """Bare container that accepts numeric child names ("0", "2") the way nn.Sequential exposes them."""


class SetRank(nn.Module):
    step_engine_cls = engine.SetRankStepEngine
    eval_engine_cls = engine.SetRankEvalEngine

    def __init__(self, hparams_str, feature_size=None):
        super().__init__()
        print("build SetRank")
        # attention_dtype is this package's one extension of the reference's hparams: "fp32" (default, the 1e-5 parity
        # path) or "fp16" (fp16 matrix-core operands in the self-attention, ordering-level parity; BASELINE config 5)
        self.hparams = HParams(d_model=256, num_heads=8, num_layers=2, diff=64, rate=0.0, initializer=None,
                               attention_dtype="fp32")
        self.hparams.parse(hparams_str)
        if float(self.hparams.rate) != 0.0:
            raise NotImplementedError("rate=%r: dropout is not implemented (the reference's default is 0.0)" % self.hparams.rate)
        self.feature_size = int(feature_size)
        self.shape = hip_ops.SetRankShape(self.feature_size, self.hparams.d_model, self.hparams.num_heads,
                                          self.hparams.num_layers, self.hparams.diff,
                                          attention_dtype=str(self.hparams.attention_dtype))
        self._bind(init_setrank_params(self.shape))
        self._fwd_saved = {}

    def _bind(self, flat):
        self.flat_params = flat
        root = _Node()
        for name, shp, off in self.shape.layout():
            parts = name.split(".")
            node = root
            for part in parts[:-1]:
                if part not in node._modules:
                    node.add_module(part, _Node())
                node = node._modules[part]
            n = int(np.prod(shp))
            node.register_parameter(parts[-1], nn.Parameter(flat[off:off + n].view(*shp), requires_grad=False))
        self.Encoder_layer = root._modules["Encoder_layer"]

    def _apply(self, fn, *a, **k):
        flat = fn(self.flat_params.detach())
        self._bind(flat.contiguous().to(torch.float32))
        self._fwd_saved = {}
        return self

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
        """Scores for a list of L tensors [B, F] -> list of L tensors [B, 1]   (SetRank.py:229-255)."""
        if noisy_params is not None:
            raise NotImplementedError("SetRank has no support for noisy parameters (SetRank.py:231)")
        if not self.flat_params.is_cuda:
            raise RuntimeError("ultra_pytorch_amd.ranking_model.SetRank.build needs the model on the GPU; there is no CPU fallback")
        dev = self.flat_params.device
        L, B = len(input_list), int(input_list[0].shape[0])
        engine._setrank_draw(L)  # SetRank.py:245-246 consumes the global `random` stream on every forward
        x = torch.cat([t.to(dev, torch.float32) for t in input_list], dim=0).contiguous()  # position-major rows
        docids = torch.arange(L * B, dtype=torch.int32, device=dev)
        scores = torch.empty(B, L, dtype=torch.float32, device=dev)
        key = (B, L)
        if key not in self._fwd_saved:
            self._fwd_saved[key] = torch.empty(max(self.shape.saved_bytes(B * L) // 4, 1), dtype=torch.float32, device=dev)
        hip_ops.setrank_forward(self.shape, self.flat_params, x, L * B, docids, B, L, scores, self._fwd_saved[key])
        return list(torch.split(scores.t().contiguous().view(L * B, 1), B, dim=0))
