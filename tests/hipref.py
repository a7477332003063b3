"""Helpers shared by the GPU parity tests: run the HIP path stage by stage from numpy inputs."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    return d, json.loads(str(d["meta"]))


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


class HipRun:
    """One configuration on the GPU; methods mirror the C-ABI stages and return numpy."""

    def __init__(self, F, hidden, B, L, algo="softmax", act="elu", **kw):
        from ultra_pytorch_amd import engine, hip_ops
        self.hip_ops = hip_ops
        self.shape = hip_ops.DnnShape(F, hidden or [], act)
        self.eng = engine.StepEngine(self.shape, B, L, torch.device("cuda"), algo=algo, **kw)
        self.B, self.L, self.F = B, L, F

    def set_inputs(self, features, docids, labels):
        feats = np.asarray(features, np.float32).reshape(-1, self.F)
        self.n_docs = feats.shape[0]
        self.features = dev(feats) if self.n_docs > 0 else None
        self.docids = dev(docids, torch.int32)
        self.labels = dev(labels, torch.float32)

    def forward(self, params, train=True):
        self.params = dev(params, torch.float32)
        self.eng.forward(self.params, self.features, self.n_docs, self.docids, train=train)
        torch.cuda.synchronize()
        return self.eng.scores.cpu().numpy()

    def loss(self, aux=None, ipw_table=None, pw=None, scores=None, uniforms=None):
        if scores is not None:
            self.eng.scores.copy_(dev(scores, torch.float32).view(self.B, self.L))
        self.aux = None if aux is None else dev(aux, torch.float32)
        self.ipw = None if ipw_table is None else dev(np.asarray(ipw_table, np.float32))
        self.pw = None if pw is None else dev(pw, torch.float32)
        kw = {}
        if uniforms is not None:
            self.uniforms = dev(uniforms, torch.float32)
            kw["uniforms"] = self.uniforms
        self.eng.loss(self.labels, aux=self.aux, ipw_table=self.ipw, pw=self.pw, **kw)
        torch.cuda.synchronize()
        tail = self.eng.tail
        parts = self.eng.loss_ws[: self.hip_ops.loss_part_count(self.B) * tail].view(-1, tail).cpu().numpy()
        return self.eng.dscores.cpu().numpy(), parts.sum(0)

    def backward(self, dscores=None):
        if dscores is not None:
            self.eng.dscores.copy_(dev(dscores, torch.float32).view(self.B, self.L))
        self.eng.backward(self.params, self.features, self.n_docs, self.docids)
        torch.cuda.synchronize()
        g = self.eng.grads.cpu().numpy()
        return g[: self.shape.n_params], g[self.shape.n_params:]

    def update(self, state):
        self.state = None if state is None else dev(state, torch.float32)
        self.eng.update(self.params, self.state, self.aux)
        torch.cuda.synchronize()
        return (self.params.cpu().numpy(), None if self.state is None else self.state.cpu().numpy(),
                None if self.aux is None else self.aux.cpu().numpy(), self.eng.scalars.cpu().numpy())
