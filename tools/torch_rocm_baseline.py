"""Stock PyTorch-ROCm baseline for bench.py (`torch_rocm_baseline`): the reference's training step written with plain torch
ops - nn.LayerNorm / nn.Linear / autograd / clip_grad_norm_ / torch.optim.Adagrad - with every tensor on the GPU, i.e. what a
user gets from the reference by running it with device='cuda' (base_algorithm.py `.to(self.cuda)` sites) on this MI355X.
CONTEXT ONLY: it says what the hand-written path buys over the stock kernels on the same part.  Not product code, not the
oracle; the formulas follow oracle/ultr_oracle.py (which is pinned to the reference) and are checked against it in
tests/test_quirks_cpu.py::test_torch_baseline_matches_oracle on CPU.

Structure notes: PairDebias uses the VECTORISED pair loss (the reference's 2450-iteration Python loop would only be slower);
DLA builds its two optimizers once (the reference rebuilds them every step, dla.py:153-154) - stateless Adagrad is emulated by
zeroing the accumulators.  Both choices favour the baseline.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class DNN(nn.Module):
    """DNN.py:41-55: [LayerNorm -> Linear -> ELU] x k -> LayerNorm -> Linear(., 1)."""

    def __init__(self, feature_size, hidden):
        super().__init__()
        self.sequential = nn.Sequential()
        k = feature_size
        outs = list(hidden) + [1]
        for j, m in enumerate(outs):
            self.sequential.add_module("layer_norm%d" % j, nn.LayerNorm(k))
            self.sequential.add_module("linear%d" % j, nn.Linear(k, m))
            if j != len(outs) - 1:
                self.sequential.add_module("act%d" % j, nn.ELU())
            k = m

    def load_flat(self, flat):
        off = 0
        with torch.no_grad():
            for p in self.parameters():  # state_dict order = the flat vector's order
                n = p.numel()
                p.copy_(torch.as_tensor(flat[off:off + n]).view_as(p))
                off += n
        assert off == len(flat)

    def scores(self, feats, ids):  # feats [n_docs + 1, F] with the zero PAD row appended, ids [L, B] int64
        L, B = ids.shape
        return self.sequential(feats[ids.reshape(-1)]).view(L, B).t()


class SetRank(nn.Module):
    """SetRank.py:23-255 with its defaults (no Q/K/V projections, no mask, dropout 0, LayerNorm eps 1e-6)."""

    def __init__(self, F_, d_model=256, heads=8, layers=2, dff=64):
        super().__init__()
        self.h, self.d = heads, d_model
        self.ln0 = nn.LayerNorm(F_, eps=1e-6)
        self.emb = nn.Sequential(nn.Linear(F_, dff), nn.ReLU(), nn.Linear(dff, d_model))
        self.out = nn.Sequential(nn.Linear(d_model, dff), nn.ReLU(), nn.Linear(dff, 1))
        self.dense = nn.ModuleList([nn.Linear(d_model, d_model) for _ in range(layers)])
        self.ffn = nn.ModuleList([nn.Sequential(nn.Linear(d_model, dff), nn.ReLU(), nn.Linear(dff, d_model)) for _ in range(layers)])
        self.ln1 = nn.ModuleList([nn.LayerNorm(d_model, eps=1e-6) for _ in range(layers)])
        self.ln2 = nn.ModuleList([nn.LayerNorm(d_model, eps=1e-6) for _ in range(layers)])

    def scores(self, feats, ids):
        L, B = ids.shape
        x = feats[ids.reshape(-1)].view(L, B, -1).permute(1, 0, 2)
        x = self.emb(self.ln0(x))
        dep = self.d // self.h
        for i in range(len(self.dense)):
            q = x.reshape(B, L, self.h, dep).permute(0, 2, 1, 3)
            att = torch.softmax((q @ q.transpose(-1, -2)) / math.sqrt(dep), dim=-1) @ q
            att = self.dense[i](att.permute(0, 2, 1, 3).reshape(B, L, self.d))
            o1 = self.ln1[i](x + att)
            x = self.ln2[i](o1 + self.ffn[i](o1))
        return self.out(x)[..., 0]


def softmax_loss(output, labels, pw=None):  # base_algorithm.py:309-330
    w = (labels + 0.0000001) * (torch.ones_like(labels) if pw is None else pw)
    dis = torch.nan_to_num(w / torch.sum(w, 1, keepdim=True))
    loss = torch.sum(-dis * F.log_softmax(output, -1), -1) * torch.sum(w, 1)
    return torch.sum(loss) / torch.sum(w)


def pairdebias_loss(scores, clicks_LB, tp, tm):  # pairwise_debias.py:142-157 vectorised, incl. the x B broadcast
    B, L = scores.shape
    c = clicks_LB.t()
    mask = torch.clamp(F.relu(c.unsqueeze(2) - c.unsqueeze(1)), max=1.0)
    pair = F.softplus(scores.unsqueeze(1) - scores.unsqueeze(2))
    PL = float(B) * (mask * pair).sum(0) * (1.0 - torch.eye(L, device=scores.device))
    return (PL / tp.view(-1, 1) / tm.view(1, -1)).sum(), (PL / tm.view(1, -1)).sum(1), (PL / tp.view(-1, 1)).sum(0)


def lambdarank_loss(scores, labels, tp, tm, sigma=1.0):  # lambda_rank.py:116-135, 247-291
    B, L = scores.shape
    dev = scores.device
    ps, inds = torch.sort(scores, dim=1, descending=True)
    labs = torch.gather(labels, 1, inds)
    Pbar = 0.5 * (1.0 + torch.clamp(labs.unsqueeze(2) - labs.unsqueeze(1), -1.0, 1.0))
    p_ij = 1.0 / (torch.exp(-sigma * (ps.unsqueeze(2) - ps.unsqueeze(1))) + 1.0)
    ideal, _ = torch.sort(labels, dim=1, descending=True)
    idcg = torch.sum((torch.pow(2.0, ideal) - 1.0) / torch.log(torch.arange(1, L + 1, dtype=torch.float32, device=dev) + 1))
    gains = (torch.pow(2.0, labs) - 1.0) / idcg
    disc = 1.0 / torch.log2(torch.arange(L, dtype=torch.float32, device=dev) + 2.0)
    delta = torch.abs(gains.unsqueeze(2) - gains.unsqueeze(1)) * torch.abs(disc.view(1, L, 1) - disc.view(1, 1, L))
    PL = F.binary_cross_entropy_with_logits(p_ij, Pbar, weight=delta, reduction="none").sum(0)
    den = tp.view(-1, 1) * tm.view(1, -1)
    loss = torch.where(den == 0, torch.zeros_like(PL), PL / den).sum()
    return loss, (PL / tm.view(1, -1)).sum(1), (PL.t() / tp.view(1, -1)).sum(1)


class Stepper:
    """One training step of `algo` ('softmax' = IPW, 'dla', 'pairdebias', 'lambdarank') on `device`; step(batch) returns the
    loss as a Python float (the reference's loss.item(): one host sync per step)."""

    def __init__(self, cfg, params0, ipw_list, device, lr, clip=5.0):
        self.dev, self.algo, self.L, self.clip = device, cfg["algo"], cfg["L"], clip
        if cfg["model"] == "setrank":
            torch.manual_seed(0)
            self.model = SetRank(cfg["F"]).to(device)  # random init of the same architecture (timing only)
        else:
            self.model = DNN(cfg["F"], cfg["hidden"])
            self.model.load_flat(params0)
            self.model.to(device)
        self.opt = torch.optim.Adagrad(self.model.parameters(), lr=lr)
        L = self.L
        if ipw_list is not None:
            t = [ipw_list[min(l, len(ipw_list) - 1)] for l in range(L)]
            self.ipw = torch.tensor(t, dtype=torch.float32, device=device).view(1, L)
        if self.algo == "dla":
            self.prop = nn.Linear(L, 1).to(device)  # DenoisingNet (dla.py:24-48)
            with torch.no_grad():
                self.prop.weight.zero_(), self.prop.bias.zero_()
            self.opt_p = torch.optim.Adagrad(self.prop.parameters(), lr=lr)
            self.eye = torch.eye(L, device=device)
        if self.algo in ("pairdebias", "lambdarank"):
            self.tp = torch.ones(L, device=device)
            self.tm = torch.ones(L, device=device)

    def stage(self, batch):
        """host numpy batch -> device tensors (done OUTSIDE the timed loop, like the HIP path's resident pool)."""
        feats, ids, y = batch
        f = torch.from_numpy(np.concatenate([feats, np.zeros((1, feats.shape[1]), np.float32)])).to(self.dev)
        return f, torch.from_numpy(ids.astype(np.int64)).to(self.dev), torch.from_numpy(y).to(self.dev)

    def step(self, staged):
        f, ids, y_LB = staged
        scores = self.model.scores(f, ids)
        labels = y_LB.t()
        if self.algo == "softmax":
            loss = softmax_loss(scores, labels, torch.where(labels > 0, self.ipw.expand_as(labels), torch.zeros_like(labels)))
        elif self.algo == "dla":
            propensity = F.elu(self.prop(self.eye)).view(1, self.L).expand_as(scores)
            with torch.no_grad():
                pr = torch.softmax(propensity, -1)
                pw = pr[:, :1] / pr
                rr = torch.softmax(scores, -1)
                rw = rr[:, :1] / rr
            loss = softmax_loss(propensity, labels, rw) + softmax_loss(scores, labels, pw)
        elif self.algo == "pairdebias":
            loss, tpl, tml = pairdebias_loss(scores, y_LB, self.tp, self.tm)
        else:
            loss, tpl, tml = lambdarank_loss(scores, labels, self.tp, self.tm)
        self.opt.zero_grad(set_to_none=True)
        if self.algo == "dla":
            self.opt_p.zero_grad(set_to_none=True)
        loss.backward()
        if self.algo == "dla":
            nn.utils.clip_grad_norm_(self.prop.parameters(), self.clip)
            for o in (self.opt, self.opt_p):  # fresh optimizers every step (dla.py:153-154) = empty accumulators
                for st in o.state.values():
                    st["sum"].zero_()
        nn.utils.clip_grad_norm_(self.model.parameters(), self.clip)
        self.opt.step()
        if self.algo == "dla":
            self.opt_p.step()
        if self.algo in ("pairdebias", "lambdarank"):
            with torch.no_grad():
                self.tp = 0.95 * self.tp + 0.05 * torch.sqrt(tpl / tpl[0])
                self.tm = 0.95 * self.tm + 0.05 * torch.sqrt(tml / tml[0])
        return loss.item()
