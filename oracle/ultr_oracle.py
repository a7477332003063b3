"""CPU oracle for the ULTRA_pytorch hot path  (TEST INFRASTRUCTURE — not product code).

A from-scratch, vectorised torch-CPU / numpy restatement of the reference's
`model.train(input_feed)` / `model.validation(input_feed)` path for the DNN ranking
model with the NA / IPW / DLA / PairDebias / LambdaRank losses.  Every function cites
the reference file:line it restates (paths relative to the ULTRA_pytorch repo root).

Rules:
  * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import
    this module.  The product package `ultra_pytorch_amd` never imports it and has no
    CPU fallback: it raises if the HIP library is missing.
  * Parity is PINNED: tests/test_oracle_golden.py checks every function here against
    golden vectors captured by running the reference itself in the build container
    (tests/golden/make_golden.py -> tests/golden/*.npz).  The reference's own tests
    hold no numerical fixtures for this path, so those captured vectors
    are the pin.
  * Arithmetic is fp32 torch-CPU — the same library the reference computes with.
  * Quirk-exact: SURVEY.md Appendix A items 1-11 are reproduced on purpose.

Layouts: `features` [n_docs, F] f32; `docids` [L, B] int (position-major, pad id ==
n_docs -> all-zero row); `labels` [L, B] f32 (clicks or relevance); scores [B, L].
Parameters travel as ONE flat f32 vector in `state_dict()` order of the reference's
`DNN.sequential`: for j in 0..k: layer_norm{j}.weight[K_j], layer_norm{j}.bias[K_j],
linear{j}.weight[M_j, K_j] (row-major), linear{j}.bias[M_j]   (DNN.py:41-55).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-5  # nn.LayerNorm default (DNN.py:46)
PADDING_SCORE = -100000.0  # base_algorithm.py:36


# --------------------------------------------------------------------------------------
# parameter layout
# --------------------------------------------------------------------------------------
def layer_dims(feature_size: int, hidden: Sequence[int]) -> List[Tuple[int, int]]:
    """[(K_j, M_j)] for Linear_j, j = 0..k.  DNN.py:38,41-55: output_sizes = hidden + [1]."""
    outs = list(hidden) + [1]
    dims, k = [], feature_size
    for m in outs:
        dims.append((k, m))
        k = m
    return dims


def param_layout(feature_size: int, hidden: Sequence[int]) -> List[Tuple[str, Tuple[int, ...], int]]:
    """[(state_dict key, shape, flat offset)] in reference parameter order (DNN.py:41-55)."""
    out, off = [], 0
    for j, (k, m) in enumerate(layer_dims(feature_size, hidden)):
        for name, shape in (("sequential.layer_norm%d.weight" % j, (k,)),
                            ("sequential.layer_norm%d.bias" % j, (k,)),
                            ("sequential.linear%d.weight" % j, (m, k)),
                            ("sequential.linear%d.bias" % j, (m,))):
            out.append((name, shape, off))
            off += int(np.prod(shape))
    return out


def num_params(feature_size: int, hidden: Sequence[int]) -> int:
    name, shape, off = param_layout(feature_size, hidden)[-1]
    return off + int(np.prod(shape))


def unflatten(flat: torch.Tensor, feature_size: int, hidden: Sequence[int]) -> Dict[str, torch.Tensor]:
    return {n: flat[o:o + int(np.prod(s))].view(*s) for n, s, o in param_layout(feature_size, hidden)}


def init_params(feature_size: int, hidden: Sequence[int], seed: int = 0) -> np.ndarray:
    """nn.LayerNorm / nn.Linear default init (what DNN.__init__ gets, DNN.py:44-52).
    Parity tests never rely on this (they load golden weights); bench/smoke use it."""
    g = torch.Generator().manual_seed(seed)
    parts = []
    for k, m in layer_dims(feature_size, hidden):
        bound = 1.0 / math.sqrt(k)  # kaiming_uniform(a=sqrt(5)) on [m,k] == U(-1/sqrt(k), 1/sqrt(k))
        parts += [torch.ones(k), torch.zeros(k),
                  (torch.rand(m, k, generator=g) * 2 - 1) * bound,
                  (torch.rand(m, generator=g) * 2 - 1) * bound]
    return torch.cat([p.reshape(-1) for p in parts]).numpy().astype(np.float32)


# --------------------------------------------------------------------------------------
# a1/a2: marshal + gather    (base_algorithm.py:134-154, 169-186)
# --------------------------------------------------------------------------------------
def gather_rows(features: np.ndarray, docids: np.ndarray) -> torch.Tensor:
    """np.take over features ++ zero PAD row, position-major rows (row = l*B + b).
    base_algorithm.py:148-153 + DNN.py:72-73 (cat dim 0, cast to f32)."""
    feats = np.asarray(features, dtype=np.float32).reshape(-1, features.shape[-1] if np.ndim(features) == 2 else 0)
    pad = np.zeros((1, feats.shape[1]), dtype=np.float32)
    table = np.concatenate((feats, pad), axis=0)
    ids = np.asarray(docids).astype(np.int64).reshape(-1)
    return torch.from_numpy(np.take(table, ids, axis=0))


# --------------------------------------------------------------------------------------
# a3: DNN forward   (DNN.py:41-55, 58-88)
# --------------------------------------------------------------------------------------
def _act(name: str):
    # base_ranking_model.py:63-69 ACT_FUNC_DIC; elu alpha = 1.  ('selu' is in the table as a plain function, which
    # nn.Sequential.add_module rejects with a TypeError: DNN.py:52-53 - not a usable option of the reference)
    return {"elu": F.elu, "relu": F.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid}[name]


def l2_term(flat: torch.Tensor, l2_loss: float, layout) -> torch.Tensor:
    """`for p in params: loss += l2_loss * self.l2_loss(p)` with l2_loss(p) = sum(p ** 2) / 2 (ipw_rank.py:154-157,
    base_algorithm.py:332-333), one term per parameter TENSOR in state_dict order.  `layout` = [(name, shape, offset)]."""
    t = torch.zeros((), dtype=torch.float32)
    for _, shape, off in layout:
        n = int(np.prod(shape))
        t = t + l2_loss * (torch.sum(flat[off:off + n] ** 2) / 2)
    return t


def dnn_forward(params: torch.Tensor, feature_size: int, hidden: Sequence[int], x: torch.Tensor,
                act: str = "elu") -> torch.Tensor:
    """[LayerNorm -> Linear -> act] x k, LayerNorm -> Linear(.,1).  LayerNorm precedes
    EVERY Linear (Appendix A.1; DNN.py:43-47: self.layer_norm is never set)."""
    p = unflatten(params, feature_size, hidden)
    dims = layer_dims(feature_size, hidden)
    h = x
    for j, (k, m) in enumerate(dims):
        h = F.layer_norm(h, (k,), p["sequential.layer_norm%d.weight" % j], p["sequential.layer_norm%d.bias" % j], LN_EPS)
        h = F.linear(h, p["sequential.linear%d.weight" % j], p["sequential.linear%d.bias" % j])
        if j != len(dims) - 1:
            h = _act(act)(h)
    return h  # [N, 1]


def ranking_scores(params: torch.Tensor, feature_size: int, hidden: Sequence[int], features: np.ndarray,
                   docids: np.ndarray, act: str = "elu") -> torch.Tensor:
    """base_algorithm.py:118-132: split position-major output into L x [B,1], cat dim 1 -> [B, L]."""
    L, B = docids.shape
    out = dnn_forward(params, feature_size, hidden, gather_rows(features, docids), act)
    return out.view(L, B).t()


def dnn_backward_manual(params: np.ndarray, feature_size: int, hidden: Sequence[int], x: np.ndarray,
                        dscore: np.ndarray, act: str = "elu", abs_terms: bool = False) -> np.ndarray:
    """Closed-form backward of dnn_forward (what autograd does for DNN.py:41-55) in numpy
    float64-free fp32 — this is the written-out spec the HIP backward kernels implement.
    Returns the flat gradient.
    abs_terms=True: the same walk in float64, returning per entry the SUM OF THE ABSOLUTE VALUES of the per-row terms the
    entry is a sum of (|dz|^T |u| for a weight, sum |dz| for a bias, sum |du xhat| / sum |du| for the LayerNorm pair) - the
    scale accumulation-order differences of an fp32 evaluation are proportional to (tests/test_gpu_full_size.py)."""
    assert act in ("elu", "relu", "tanh", "sigmoid")
    if abs_terms:
        return _dnn_backward_abs_terms(params, feature_size, hidden, x, dscore, act)
    x = np.asarray(x, np.float32)
    dims = layer_dims(feature_size, hidden)
    lay = {n: (s, o) for n, s, o in param_layout(feature_size, hidden)}

    def get(n):
        s, o = lay[n]
        return params[o:o + int(np.prod(s))].reshape(s)

    xs, xhats, rstds, us = [], [], [], []
    h = x
    for j, (k, m) in enumerate(dims):
        mu = h.mean(axis=1, keepdims=True, dtype=np.float32)
        var = ((h - mu) ** 2).mean(axis=1, keepdims=True, dtype=np.float32)
        r = (1.0 / np.sqrt(var + np.float32(LN_EPS))).astype(np.float32)
        xhat = (h - mu) * r
        u = xhat * get("sequential.layer_norm%d.weight" % j) + get("sequential.layer_norm%d.bias" % j)
        z = u @ get("sequential.linear%d.weight" % j).T + get("sequential.linear%d.bias" % j)
        xs.append(h), xhats.append(xhat), rstds.append(r), us.append(u)
        if j != len(dims) - 1:
            h = {"elu": lambda: np.where(z > 0, z, np.expm1(np.minimum(z, 0))), "relu": lambda: np.maximum(z, 0),
                 "tanh": lambda: np.tanh(z), "sigmoid": lambda: 1.0 / (1.0 + np.exp(-z))}[act]().astype(np.float32)
    grads = np.zeros_like(params)

    def put(n, g):
        s, o = lay[n]
        grads[o:o + int(np.prod(s))] = g.reshape(-1)

    dz = np.asarray(dscore, np.float32).reshape(-1, 1)
    for j in reversed(range(len(dims))):
        W = get("sequential.linear%d.weight" % j)
        put("sequential.linear%d.weight" % j, dz.T @ us[j])
        put("sequential.linear%d.bias" % j, dz.sum(axis=0))
        du = dz @ W
        put("sequential.layer_norm%d.weight" % j, (du * xhats[j]).sum(axis=0))
        put("sequential.layer_norm%d.bias" % j, du.sum(axis=0))
        if j == 0:
            break
        dxhat = du * get("sequential.layer_norm%d.weight" % j)
        c1 = dxhat.mean(axis=1, keepdims=True)
        c2 = (dxhat * xhats[j]).mean(axis=1, keepdims=True)
        dx = rstds[j] * (dxhat - c1 - xhats[j] * c2)
        a = xs[j]  # = act(z_{j-1})
        dact = {"elu": lambda: np.where(a > 0, 1.0, a + 1.0), "relu": lambda: (a > 0).astype(np.float32),
                "tanh": lambda: 1.0 - a * a, "sigmoid": lambda: a * (1.0 - a)}[act]()
        dz = (dx * dact).astype(np.float32)
    return grads


def _dnn_backward_abs_terms(params, feature_size, hidden, x, dscore, act):
    f8 = np.float64
    params = np.asarray(params, f8)
    dims = layer_dims(feature_size, hidden)
    lay = {n: (s, o) for n, s, o in param_layout(feature_size, hidden)}
    get = lambda n: params[lay[n][1]:lay[n][1] + int(np.prod(lay[n][0]))].reshape(lay[n][0])
    xs, xhats, rstds, us = [], [], [], []
    h = np.asarray(x, f8)
    for j, (k, m) in enumerate(dims):
        mu = h.mean(axis=1, keepdims=True)
        r = 1.0 / np.sqrt(((h - mu) ** 2).mean(axis=1, keepdims=True) + LN_EPS)
        xhat = (h - mu) * r
        u = xhat * get("sequential.layer_norm%d.weight" % j) + get("sequential.layer_norm%d.bias" % j)
        z = u @ get("sequential.linear%d.weight" % j).T + get("sequential.linear%d.bias" % j)
        xs.append(h), xhats.append(xhat), rstds.append(r), us.append(u)
        if j != len(dims) - 1:
            h = {"elu": lambda: np.where(z > 0, z, np.expm1(np.minimum(z, 0))), "relu": lambda: np.maximum(z, 0),
                 "tanh": lambda: np.tanh(z), "sigmoid": lambda: 1.0 / (1.0 + np.exp(-z))}[act]()
    out = np.zeros(params.shape, f8)

    def put(n, g):
        s, o = lay[n]
        out[o:o + int(np.prod(s))] = g.reshape(-1)

    dz = np.asarray(dscore, f8).reshape(-1, 1)
    for j in reversed(range(len(dims))):
        W = get("sequential.linear%d.weight" % j)
        put("sequential.linear%d.weight" % j, np.abs(dz).T @ np.abs(us[j]))
        put("sequential.linear%d.bias" % j, np.abs(dz).sum(axis=0))
        du = dz @ W
        put("sequential.layer_norm%d.weight" % j, np.abs(du * xhats[j]).sum(axis=0))
        put("sequential.layer_norm%d.bias" % j, np.abs(du).sum(axis=0))
        if j == 0:
            break
        dxhat = du * get("sequential.layer_norm%d.weight" % j)
        dx = rstds[j] * (dxhat - dxhat.mean(axis=1, keepdims=True) - xhats[j] * (dxhat * xhats[j]).mean(axis=1, keepdims=True))
        a = xs[j]
        dact = {"elu": lambda: np.where(a > 0, 1.0, a + 1.0), "relu": lambda: (a > 0).astype(f8),
                "tanh": lambda: 1.0 - a * a, "sigmoid": lambda: a * (1.0 - a)}[act]()
        dz = dx * dact
    return out


# --------------------------------------------------------------------------------------
# a4: listwise softmax cross entropy   (base_algorithm.py:18-30, 309-330)
# --------------------------------------------------------------------------------------
def softmax_loss(output: torch.Tensor, labels: torch.Tensor, propensity_weights: Optional[torch.Tensor] = None) -> torch.Tensor:
    if propensity_weights is None:
        propensity_weights = torch.ones_like(labels)
    weighted_labels = (labels + 0.0000001) * propensity_weights  # the 1e-7 smoothing (Appendix A.2)
    label_dis = weighted_labels / torch.sum(weighted_labels, 1, keepdim=True)
    label_dis = torch.nan_to_num(label_dis)
    loss = torch.sum(-label_dis * F.log_softmax(output, -1), -1) * torch.sum(weighted_labels, 1)
    return torch.sum(loss) / torch.sum(weighted_labels)  # GLOBAL normaliser


def softmax_loss_closed_form(scores: np.ndarray, labels: np.ndarray, pw: Optional[np.ndarray]):
    """Closed form used as the kernel spec:  w=(y+1e-7)pw, S_b=sum_l w, D=sum w,
    loss=sum_b( -sum_l w*log_softmax(s) )/D,  dloss/ds = (softmax(s)*S_b - w)/D."""
    s = torch.as_tensor(scores, dtype=torch.float32)
    y = torch.as_tensor(labels, dtype=torch.float32)
    w = (y + 0.0000001) * (torch.ones_like(y) if pw is None else torch.as_tensor(pw, dtype=torch.float32))
    Sb = w.sum(1, keepdim=True)
    D = w.sum()
    lsm = F.log_softmax(s, -1)
    loss = (-(w * lsm).sum()) / D
    ds = (lsm.exp() * Sb - w) / D
    return float(loss), ds.numpy(), float(D)


# --------------------------------------------------------------------------------------
# a8: IPW weights   (propensity_estimator.py:22-42, ipw_rank.py:115-138)
# --------------------------------------------------------------------------------------
def ipw_weights(clicks_LB: np.ndarray, ipw_list: Sequence[float]) -> torch.Tensor:
    """pw[b,l] = IPW_list[min(l, len-1)] if click[l,b] > 0 else 0  ->  f32 [B, L]."""
    L, B = clicks_LB.shape
    table = np.asarray([ipw_list[l] if l < len(ipw_list) else ipw_list[-1] for l in range(L)], dtype=np.float64)
    pw = np.where(np.asarray(clicks_LB).T > 0, table[None, :], 0.0)
    return torch.as_tensor(pw.tolist())  # list of python floats -> f32, as ipw_rank.py:138


# --------------------------------------------------------------------------------------
# a5/a6: clip + optimizer   (base_algorithm.py:208-226; torch.optim.Adagrad / SGD)
# --------------------------------------------------------------------------------------
def clip_coef(total_norm: float, max_norm: float) -> float:
    """torch.nn.utils.clip_grad_norm_: coef = max_norm/(norm+1e-6) clamped to 1."""
    return min(1.0, max_norm / (total_norm + 1e-6))


def grad_norm(g: torch.Tensor) -> torch.Tensor:
    """clip_grad_norm_ takes the norm of the per-tensor norms (short fp32 sums, accurate to ~1e-7); a flat fp32
    vector_norm over 250k elements is off by 1e-5 (measured on the SetRank fixture), so accumulate in fp64."""
    return torch.linalg.vector_norm(g.double(), 2).float()


def adagrad_update(p: torch.Tensor, g: torch.Tensor, state_sum: torch.Tensor, lr: float, eps: float = 1e-10):
    """torch.optim.Adagrad (lr_decay 0, weight_decay 0, initial_accumulator 0): s += g*g; p -= lr*g/(sqrt(s)+eps)."""
    s = state_sum + g * g
    return p - lr * g / (s.sqrt() + eps), s


def apply_update(p, g, state_sum, lr, max_norm, strategy="ada", stateless=False):
    """opt_step (base_algorithm.py:208-226): clip_grad_norm_(max_norm) then optimizer.step().
    stateless=True restates DLA's per-step optimizer re-construction (dla.py:153-154): the
    Adagrad accumulator is always empty when step() runs."""
    n = grad_norm(g)
    if max_norm > 0:
        g = g * min(1.0, float(max_norm / (n + 1e-6)))
    if strategy == "sgd":
        return p - lr * g, state_sum, n, g
    if stateless:
        state_sum = torch.zeros_like(p)
    p2, s2 = adagrad_update(p, g, state_sum, lr)
    return p2, s2, n, g


# --------------------------------------------------------------------------------------
# a7/a8: NA and IPW train steps   (navie_algorithm.py:76-120, ipw_rank.py:102-182)
# --------------------------------------------------------------------------------------
def train_step_softmax(params, state_sum, F_, hidden, features, docids, labels_LB, ipw_list=None, lr=0.05,
                       max_norm=5.0, strategy="ada", act="elu", l2_loss=0.0):
    """One NA (ipw_list None) / IPW step.  Returns dict(loss, scores, grads, norm, params, state).
    l2_loss > 0 (ipw_rank.py:154-159, navie_algorithm.py:109-116): the L2 loop exhausts the `params` generator that is then
    handed to clip_grad_norm_, so the clip sees NO parameters and nothing is clipped."""
    p = torch.as_tensor(params, dtype=torch.float32).clone().requires_grad_(True)
    scores = ranking_scores(p, F_, hidden, features, docids, act)
    labels = torch.from_numpy(np.ascontiguousarray(np.transpose(labels_LB))).float()  # base_algorithm.py:182
    pw = None if ipw_list is None else ipw_weights(labels_LB, ipw_list)
    loss = softmax_loss(scores, labels, pw)
    if l2_loss > 0:
        loss = loss + l2_term(p, l2_loss, param_layout(F_, hidden))
        max_norm = 0.0
    (g,) = torch.autograd.grad(loss, p)
    with torch.no_grad():
        p2, s2, n, gc = apply_update(p.detach(), g, torch.as_tensor(state_sum, dtype=torch.float32), lr, max_norm, strategy)
    return dict(loss=float(loss.detach()), scores=scores.detach().numpy(), grads=g.numpy(), norm=float(n),
                params=p2.numpy(), state=s2.numpy(), pw=None if pw is None else pw.numpy())


# --------------------------------------------------------------------------------------
# a9: DLA   (dla.py:24-48, 141-177, 179-266, 287-306)
# --------------------------------------------------------------------------------------
def denoising_net(prop_params: torch.Tensor, B: int, L: int) -> torch.Tensor:
    """DenoisingNet.forward (dla.py:33-48): one-hot(l) -> Linear(L,1) -> ELU, i.e.
    propensity[b,l] = ELU(W[0,l] + bias), batch-independent.  prop_params = [W(L) | bias(1)]."""
    w, b = prop_params[:L], prop_params[L]
    return F.elu(w + b).unsqueeze(0).expand(B, L)


def normalized_weights(prob: torch.Tensor) -> torch.Tensor:
    """get_normalized_weights (dla.py:287-306): w[:,l] = prob[:,0]/prob[:,l]; the
    max_propensity_weight clamp acts on a grad-less tensor's .grad -> inert (Appendix A.5)."""
    return prob[:, :1] / prob


def logits_to_prob(x: torch.Tensor, kind: str = "softmax") -> torch.Tensor:
    if kind == "sigmoid":  # dla.py:21-22
        return torch.sigmoid(x - torch.mean(x, -1, keepdim=True))
    return torch.softmax(x, dim=-1)


def _fresh_adagrad_step(flat, grad, F_, hidden, lr, max_norm, split):
    """DLA.separate_gradient_update in the reference's own STRUCTURE (dla.py:141-177): one leaf tensor per parameter,
    clip_grad_norm_ over them, a torch.optim.Adagrad object constructed for this step only, .step().  Numerically the
    stateless update of apply_update(stateless=True); used for the bench's "reference structure" CPU timing and checked
    against it in tests/test_quirks_cpu.py."""
    if split:
        leaves = [flat[o:o + int(np.prod(s))].detach().clone().view(*s).requires_grad_(True) for _, s, o in param_layout(F_, hidden)]
        grads = [grad[o:o + int(np.prod(s))].view(*s) for _, s, o in param_layout(F_, hidden)]
    else:
        leaves, grads = [flat.detach().clone().requires_grad_(True)], [grad]
    for p, g in zip(leaves, grads):
        p.grad = g.clone()
    opt = torch.optim.Adagrad(leaves, lr=lr)
    norm = torch.nn.utils.clip_grad_norm_(leaves, max_norm)
    opt.step()
    return torch.cat([p.detach().reshape(-1) for p in leaves]), float(norm)


def dla_step(params, prop_params, F_, hidden, features, docids, labels_LB, lr=0.05, prop_lr=None, max_norm=5.0,
             ranker_loss_weight=1.0, strategy="ada", l2p="softmax", act="elu", fresh_optimizers=False, l2_loss=0.0):
    prop_lr = lr if prop_lr is None or prop_lr < 0 else prop_lr
    L, B = docids.shape
    p = torch.as_tensor(params, dtype=torch.float32).clone().requires_grad_(True)
    q = torch.as_tensor(prop_params, dtype=torch.float32).clone().requires_grad_(True)
    scores = ranking_scores(p, F_, hidden, features, docids, act)
    labels = torch.from_numpy(np.ascontiguousarray(np.transpose(labels_LB))).float()
    propensity = denoising_net(q, B, L)
    with torch.no_grad():
        pw = normalized_weights(logits_to_prob(propensity, l2p))  # dla.py:200-202
    rank_loss = softmax_loss(scores, labels, pw)  # dla.py:203
    with torch.no_grad():
        rw = normalized_weights(logits_to_prob(scores, l2p))  # dla.py:217-219
    exam_loss = softmax_loss(propensity, labels, rw)  # dla.py:221-224
    if l2_loss > 0:  # dla.py:146-150: on the ranking model only, INSIDE rank_loss; both clips stay active (fresh .parameters())
        rank_loss = rank_loss + l2_term(p, l2_loss, param_layout(F_, hidden))
    loss = exam_loss + ranker_loss_weight * rank_loss  # dla.py:237
    gp, gq = torch.autograd.grad(loss, (p, q))
    if fresh_optimizers and strategy == "ada":
        q2, nq = _fresh_adagrad_step(q.detach(), gq, F_, hidden, prop_lr, max_norm, split=False)
        p2, np_ = _fresh_adagrad_step(p.detach(), gp, F_, hidden, lr, max_norm, split=True)
    else:
        with torch.no_grad():
            # separate clips (dla.py:161-163), fresh optimizers (dla.py:153-154) => stateless Adagrad
            q2, _, nq, _ = apply_update(q.detach(), gq, torch.zeros_like(gq), prop_lr, max_norm, strategy, stateless=True)
            p2, _, np_, _ = apply_update(p.detach(), gp, torch.zeros_like(gp), lr, max_norm, strategy, stateless=True)
    return dict(loss=float(loss.detach()), rank_loss=float(rank_loss.detach()), exam_loss=float(exam_loss.detach()), scores=scores.detach().numpy(),
                grads=gp.numpy(), norm=float(np_), prop_grads=gq.numpy(), prop_norm=float(nq), params=p2.numpy(),
                prop_params=q2.numpy(), propensity_weights=pw.numpy(), relevance_weights=rw.numpy())


# --------------------------------------------------------------------------------------
# a10: PairDebias   (pairwise_debias.py:106-174, base_algorithm.py:228-248)
# --------------------------------------------------------------------------------------
def pairdebias_loss(scores: torch.Tensor, clicks_LB: torch.Tensor, t_plus: torch.Tensor, t_minus: torch.Tensor):
    """Vectorised restatement of the 2-level Python pair loop (pairwise_debias.py:142-157).
    PL[i,j] = B * sum_b min(1,relu(c_i-c_j)) * (-log_softmax([s_i,s_j])[0]);  the factor B is the
    reference's [B]*[B,1] broadcast (Appendix A.6).  Returns (loss, PL[L,L], t_plus_loss[L], t_minus_loss[L])."""
    B, L = scores.shape
    c = clicks_LB.t()  # [B, L]
    mask = torch.minimum(torch.ones(()), F.relu(c.unsqueeze(2) - c.unsqueeze(1)))  # [B, i, j]
    pair = F.softplus(scores.unsqueeze(1) - scores.unsqueeze(2))  # [B,i,j] = log(1+exp(s_j - s_i))
    offdiag = 1.0 - torch.eye(L)
    PL = float(B) * (mask * pair).sum(0) * offdiag  # [L, L]
    tp, tm = t_plus.view(-1), t_minus.view(-1)
    t_plus_loss = (PL / tm.unsqueeze(0)).sum(1)
    t_minus_loss = (PL / tp.unsqueeze(1)).sum(0)
    loss = (PL / tp.unsqueeze(1) / tm.unsqueeze(0)).sum()
    return loss, PL, t_plus_loss, t_minus_loss


def pairdebias_loss_loops(scores: torch.Tensor, clicks_LB: torch.Tensor, t_plus: torch.Tensor, t_minus: torch.Tensor):
    """The same quantities in the reference's own STRUCTURE (pairwise_debias.py:142-157): a Python loop over the L*(L-1)
    ordered position pairs, each a handful of small tensor ops, with the [B] x [B,1] broadcast of
    pairwise_cross_entropy_loss (base_algorithm.py:242-248) left in place (it is where the factor B comes from).
    Used for the bench's "reference structure" CPU timing; equal to pairdebias_loss (tests/test_quirks_cpu.py)."""
    B, L = scores.shape
    cols = [scores[:, l:l + 1] for l in range(L)]  # [B, 1] each, as torch.split gives the reference
    tp, tm = t_plus.view(-1), t_minus.view(-1)
    tpl = [scores.new_zeros(()) for _ in range(L)]
    tml = [scores.new_zeros(()) for _ in range(L)]
    PL = [[scores.new_zeros(()) for _ in range(L)] for _ in range(L)]
    loss = scores.new_zeros(())
    for i in range(L):
        for j in range(L):
            if i == j:
                continue
            mask = torch.minimum(torch.ones_like(clicks_LB[i]), F.relu(clicks_LB[i] - clicks_LB[j]))  # [B]
            logits = torch.cat([cols[i], cols[j]], dim=1)
            ce = -(torch.log_softmax(logits, dim=-1)[:, 0])  # [B]: label distribution (1, 0), base_algorithm.py:18-30
            ce = ce * torch.ones_like(cols[i])  # [B] * [B, 1] -> [B, B]: every pair loss counted B times (:242-248)
            pl = torch.sum(mask * ce)
            PL[i][j] = pl
            tpl[i] = tpl[i] + pl / tm[j]
            tml[j] = tml[j] + pl / tp[i]
            loss = loss + pl / tp[i] / tm[j]
    return loss, torch.stack([torch.stack(r) for r in PL]), torch.stack(tpl), torch.stack(tml)


def em_update(t, t_loss, alpha, p, safe=False):
    """pairwise_debias.py:160-163 (plain /) and lambda_rank.py:138-142 (_safe_div)."""
    ratio = torch.where(t_loss[0] == 0, torch.zeros_like(t_loss), t_loss / t_loss[0]) if safe else t_loss / t_loss[0]
    return (1 - alpha) * t + alpha * torch.pow(ratio, 1.0 / (p + 1)).view_as(t)


def pairdebias_step(params, state_sum, t_plus, t_minus, F_, hidden, features, docids, labels_LB, lr=0.005,
                    max_norm=5.0, em_step=0.05, reg_p=1, strategy="ada", act="elu", loops=False, l2_loss=0.0):
    p = torch.as_tensor(params, dtype=torch.float32).clone().requires_grad_(True)
    tp = torch.as_tensor(t_plus, dtype=torch.float32)
    tm = torch.as_tensor(t_minus, dtype=torch.float32)
    scores = ranking_scores(p, F_, hidden, features, docids, act)
    loss_fn = pairdebias_loss_loops if loops else pairdebias_loss
    loss, PL, tpl, tml = loss_fn(scores, torch.as_tensor(labels_LB, dtype=torch.float32), tp, tm)
    if l2_loss > 0:  # pairwise_debias.py:166-171: same exhausted-generator quirk, the clip is skipped
        loss = loss + l2_term(p, l2_loss, param_layout(F_, hidden))
        max_norm = 0.0
    (g,) = torch.autograd.grad(loss, p)
    with torch.no_grad():
        tp2 = em_update(tp, tpl, em_step, reg_p)
        tm2 = em_update(tm, tml, em_step, reg_p)
        p2, s2, n, _ = apply_update(p.detach(), g, torch.as_tensor(state_sum, dtype=torch.float32), lr, max_norm, strategy)
    return dict(loss=float(loss.detach()), scores=scores.detach().numpy(), grads=g.numpy(), norm=float(n), params=p2.numpy(),
                state=s2.numpy(), t_plus=tp2.numpy(), t_minus=tm2.numpy(), pair_loss=PL.detach().numpy())


# --------------------------------------------------------------------------------------
# a11: LambdaRank   (lambda_rank.py:96-216, 247-291)
# --------------------------------------------------------------------------------------
def _safe_div(n: torch.Tensor, d: torch.Tensor) -> torch.Tensor:
    """metrics.py:156-170."""
    return torch.where(torch.eq(d, 0), torch.zeros_like(n), torch.div(n, d))


def lambdarank_loss(scores: torch.Tensor, labels: torch.Tensor, t_plus: torch.Tensor, t_minus: torch.Tensor, sigma: float = 1.0):
    B, L = scores.shape
    preds_sorted, inds = torch.sort(scores, dim=1, descending=True)  # :116
    labs = torch.gather(labels, 1, inds)  # :117
    S = torch.clamp(labs.unsqueeze(2) - labs.unsqueeze(1), -1.0, 1.0)  # :119-121
    Pbar = 0.5 * (1.0 + S)
    s_ij = preds_sorted.unsqueeze(2) - preds_sorted.unsqueeze(1)
    p_ij = 1.0 / (torch.exp(-sigma * s_ij) + 1.0)  # :125
    ideal, _ = torch.sort(labels, dim=1, descending=True)
    # dcg(): batch-GLOBAL scalar, natural log (lambda_rank.py:247-266; Appendix A.7)
    pos = torch.arange(1, L + 1, dtype=torch.float32)
    idcg = torch.sum(_safe_div(torch.pow(torch.tensor(2.0), ideal) - 1.0, torch.log(pos + 1)))
    gains = (torch.pow(2.0, labs) - 1.0) / idcg  # :280-282
    disc = 1.0 / torch.log2(torch.arange(L, dtype=torch.float32) + 2.0)
    delta = torch.abs(gains.unsqueeze(2) - gains.unsqueeze(1)) * torch.abs(disc.view(1, L, 1) - disc.view(1, 1, L))
    # BCE-with-LOGITS applied to the probability p_ij (lambda_rank.py:128; Appendix A.7)
    l = F.binary_cross_entropy_with_logits(p_ij, Pbar, weight=delta, reduction="none")
    PL = l.sum(0)  # [L, L] incl. diagonal (delta = 0 there)
    tp, tm = t_plus.view(-1), t_minus.view(-1)
    t_plus_loss = (PL / tm.unsqueeze(0)).sum(1)  # :130
    t_minus_loss = (PL.t() / tp.unsqueeze(0)).sum(1)  # :131-132
    loss = _safe_div(PL, tp.unsqueeze(1) * tm.unsqueeze(0)).sum()  # :133-135
    return loss, PL, t_plus_loss, t_minus_loss


def lambdarank_step(params, state_sum, t_plus, t_minus, F_, hidden, features, docids, labels_LB, lr=0.05, max_norm=5.0,
                    em_step=0.05, reg_p=1, sigma=1.0, strategy="ada", act="elu"):
    p = torch.as_tensor(params, dtype=torch.float32).clone().requires_grad_(True)
    tp = torch.as_tensor(t_plus, dtype=torch.float32)
    tm = torch.as_tensor(t_minus, dtype=torch.float32)
    scores = ranking_scores(p, F_, hidden, features, docids, act)
    labels = torch.from_numpy(np.ascontiguousarray(np.transpose(labels_LB))).float()
    loss, PL, tpl, tml = lambdarank_loss(scores, labels, tp, tm, sigma)
    (g,) = torch.autograd.grad(loss, p)
    with torch.no_grad():
        tp2 = em_update(tp, tpl, em_step, reg_p, safe=True)
        tm2 = em_update(tm, tml, em_step, reg_p, safe=True)
        p2, s2, n, _ = apply_update(p.detach(), g, torch.as_tensor(state_sum, dtype=torch.float32), lr, max_norm, strategy)
    return dict(loss=float(loss.detach()), scores=scores.detach().numpy(), grads=g.numpy(), norm=float(n), params=p2.numpy(),
                state=s2.numpy(), t_plus=tp2.numpy(), t_minus=tm2.numpy(), pair_loss=PL.detach().numpy())


# --------------------------------------------------------------------------------------
# next row 8f.3: RegressionEM  (regression_EM.py:108-193)
# --------------------------------------------------------------------------------------
def regression_em_estimation(scores: torch.Tensor, labels: torch.Tensor, propensity: torch.Tensor):
    """E-step (regression_EM.py:130-145): gamma = sigmoid(s + sigmoid_prob_b) with sigmoid_prob_b == 0 (a plain tensor,
    never trained: :102-104); posteriors of (examined, not relevant) and (not examined, relevant) given no click;
    p_r1 = c + (1 - c) * P(e=0, r=1 | c=0)."""
    gamma = torch.sigmoid(scores)
    den = 1 - propensity * gamma
    p_e1_r0_c0 = propensity * (1 - gamma) / den
    p_e0_r1_c0 = (1 - propensity) * gamma / den
    p_r1 = labels + (1 - labels) * p_e0_r1_c0
    return p_e1_r0_c0, p_r1


def regression_em_step(params, state_sum, propensity, uniforms, F_, hidden, features, docids, labels_LB, lr=0.05,
                       max_norm=5.0, em_step=0.05, strategy="ada", act="elu", l2_loss=0.0):
    """One RegressionEM.train step with the Bernoulli uniforms INJECTED (the reference draws them from an unseeded
    torch.rand: get_bernoulli_sample, regression_EM.py:20-34, sample = ceil(p - u)).  Loss = BCEWithLogits(scores,
    pseudo-labels), mean over all B*L elements (:149-151); persistent Adagrad + global-norm clip (:165-176); M-step
    propensity <- (1-a) propensity + a * mean_b(c + (1-c) P(e=1, r=0 | c=0)) with the PRE-update scores (:180-183)."""
    p = torch.as_tensor(params, dtype=torch.float32).clone().requires_grad_(True)
    prop = torch.as_tensor(propensity, dtype=torch.float32).reshape(1, -1)
    u = torch.as_tensor(uniforms, dtype=torch.float32)
    scores = ranking_scores(p, F_, hidden, features, docids, act)
    labels = torch.from_numpy(np.ascontiguousarray(np.transpose(labels_LB))).float()
    with torch.no_grad():
        p_e1_r0_c0, p_r1 = regression_em_estimation(scores.detach(), labels, prop)
        ranker_labels = torch.ceil(p_r1 - u)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(scores, ranker_labels)
    if l2_loss > 0:  # regression_EM.py:166-176: same exhausted-generator quirk, the clip is skipped
        loss = loss + l2_term(p, l2_loss, param_layout(F_, hidden))
        max_norm = 0.0
    (g,) = torch.autograd.grad(loss, p)
    with torch.no_grad():
        prop2 = (1 - em_step) * prop + em_step * torch.mean(labels + (1 - labels) * p_e1_r0_c0, dim=0, keepdim=True)
        p2, s2, n, _ = apply_update(p.detach(), g, torch.as_tensor(state_sum, dtype=torch.float32), lr, max_norm, strategy)
    return dict(loss=float(loss.detach()), scores=scores.detach().numpy(), grads=g.numpy(), norm=float(n), params=p2.numpy(),
                state=s2.numpy(), propensity=prop2.numpy(), ranker_labels=ranker_labels.numpy())


# --------------------------------------------------------------------------------------
# next row 8f.1: the SetRank ranking model  (ranking_model/SetRank.py:23-255)
# --------------------------------------------------------------------------------------
def setrank_layout(feature_size: int, d_model: int, num_layers: int, dff: int) -> List[Tuple[str, Tuple[int, ...], int]]:
    """Flat parameter layout = SetRank.state_dict() order: Encoder.__init__ registers input_layer_norm, input_embedding,
    output_layer, then enc_layers (SetRank.py:130-141); an EncoderLayer registers mha.dense, ffn, layernorm1, layernorm2
    (:95-103)."""
    out, off = [], 0

    def add(name, shape):
        nonlocal off
        out.append((name, tuple(shape), off))
        off += int(np.prod(shape))

    e = "Encoder_layer."
    add(e + "input_layer_norm.weight", (feature_size,))
    add(e + "input_layer_norm.bias", (feature_size,))
    add(e + "input_embedding.0.weight", (dff, feature_size))
    add(e + "input_embedding.0.bias", (dff,))
    add(e + "input_embedding.2.weight", (d_model, dff))
    add(e + "input_embedding.2.bias", (d_model,))
    add(e + "output_layer.0.weight", (dff, d_model))
    add(e + "output_layer.0.bias", (dff,))
    add(e + "output_layer.2.weight", (1, dff))
    add(e + "output_layer.2.bias", (1,))
    for i in range(num_layers):
        l = e + "enc_layers.encoder%d." % i
        add(l + "mha.dense.weight", (d_model, d_model))
        add(l + "mha.dense.bias", (d_model,))
        add(l + "ffn.0.weight", (dff, d_model))
        add(l + "ffn.0.bias", (dff,))
        add(l + "ffn.2.weight", (d_model, dff))
        add(l + "ffn.2.bias", (d_model,))
        add(l + "layernorm1.weight", (d_model,))
        add(l + "layernorm1.bias", (d_model,))
        add(l + "layernorm2.weight", (d_model,))
        add(l + "layernorm2.bias", (d_model,))
    return out


def setrank_forward(params: torch.Tensor, feature_size: int, d_model: int, num_heads: int, num_layers: int, dff: int,
                    features: np.ndarray, docids: np.ndarray) -> torch.Tensor:
    """SetRank.build + Encoder.forward (SetRank.py:143-156, 229-255) -> scores [B, L].  No Q/K/V projections (the
    heads are slices of x itself, :57-66), no mask, dropout rate 0.0 (default hparam) = identity, LayerNorm eps 1e-6,
    softmax(q k^T / sqrt(depth)) v per head (:159-195), then mha.dense; post-LN residual blocks (:105-117)."""
    P = {n: params[o:o + int(np.prod(sh))].reshape(sh) for n, sh, o in setrank_layout(feature_size, d_model, num_layers, dff)}
    e = "Encoder_layer."
    x = gather_rows(features, docids)  # [L, B, F] rows by (position, query)
    L, B = docids.shape
    x = x.reshape(L, B, feature_size).permute(1, 0, 2).float()  # [B, L, F]
    x = torch.nn.functional.layer_norm(x, (feature_size,), P[e + "input_layer_norm.weight"], P[e + "input_layer_norm.bias"], 1e-6)
    x = torch.relu(x @ P[e + "input_embedding.0.weight"].T + P[e + "input_embedding.0.bias"])
    x = x @ P[e + "input_embedding.2.weight"].T + P[e + "input_embedding.2.bias"]
    depth = d_model // num_heads
    for i in range(num_layers):
        l = e + "enc_layers.encoder%d." % i
        q = x.reshape(B, L, num_heads, depth).permute(0, 2, 1, 3)  # [B, H, L, depth]
        logits = (q @ q.transpose(-1, -2)) / torch.sqrt(torch.tensor(float(depth)))
        att = torch.softmax(logits, dim=-1) @ q
        att = att.permute(0, 2, 1, 3).reshape(B, L, d_model)
        att = att @ P[l + "mha.dense.weight"].T + P[l + "mha.dense.bias"]
        out1 = torch.nn.functional.layer_norm(x + att, (d_model,), P[l + "layernorm1.weight"], P[l + "layernorm1.bias"], 1e-6)
        f = torch.relu(out1 @ P[l + "ffn.0.weight"].T + P[l + "ffn.0.bias"])
        f = f @ P[l + "ffn.2.weight"].T + P[l + "ffn.2.bias"]
        x = torch.nn.functional.layer_norm(out1 + f, (d_model,), P[l + "layernorm2.weight"], P[l + "layernorm2.bias"], 1e-6)
    o = torch.relu(x @ P[e + "output_layer.0.weight"].T + P[e + "output_layer.0.bias"])
    o = o @ P[e + "output_layer.2.weight"].T + P[e + "output_layer.2.bias"]
    return o[..., 0]  # [B, L]


def train_step_setrank_softmax(params, state_sum, cfg, features, docids, labels_LB, ipw_list=None, lr=0.05, max_norm=5.0,
                               strategy="ada"):
    """NavieAlgorithm / IPWrank.train with ranking_model = SetRank: same loss, clip and optimizer as a7 / a8.
    cfg = (feature_size, d_model, num_heads, num_layers, dff)."""
    p = torch.as_tensor(params, dtype=torch.float32).clone().requires_grad_(True)
    scores = setrank_forward(p, *cfg, features, docids)
    labels = torch.from_numpy(np.ascontiguousarray(np.transpose(labels_LB))).float()
    pw = ipw_weights(labels_LB, ipw_list) if ipw_list is not None else None
    loss = softmax_loss(scores, labels, pw)
    (g,) = torch.autograd.grad(loss, p)
    with torch.no_grad():
        p2, s2, n, _ = apply_update(p.detach(), g, torch.as_tensor(state_sum, dtype=torch.float32), lr, max_norm, strategy)
    return dict(loss=float(loss.detach()), scores=scores.detach().numpy(), grads=g.numpy(), norm=float(n), params=p2.numpy(),
                state=s2.numpy())


# --------------------------------------------------------------------------------------
# a12/a13: validation  (base_algorithm.py:88-116; metrics.py:156-336, 456-495)
# --------------------------------------------------------------------------------------
def mask_padding(scores: torch.Tensor, docids_LB: np.ndarray, n_docs: int) -> torch.Tensor:
    """remove_padding_for_metric_eval: score := -100000 where docid == n_docs (the PAD row)."""
    valid = torch.from_numpy(np.asarray(docids_LB).T != n_docs)
    return torch.where(valid, scores, torch.full_like(scores, PADDING_SCORE))


def _prepare(labels: torch.Tensor, predictions: torch.Tensor, topn: Sequence[int]):
    """metrics.py:224-265 with weights=None."""
    list_size = predictions.shape[1]
    topn = [min(n, list_size) for n in topn]
    ok = labels >= 0.0  # metric_utils.py:44-46
    labels = torch.where(ok, labels, torch.zeros_like(labels))
    predictions = torch.where(ok, predictions, -1e-6 * torch.ones_like(predictions) + torch.min(predictions, dim=1, keepdim=True).values)
    return labels, predictions, topn


def _dcg(prediction: torch.Tensor, labels: torch.Tensor, topn: Sequence[int]) -> torch.Tensor:
    """metrics.py:191-221 with weights = 1."""
    L = labels.shape[1]
    _, idx = prediction.sort(descending=True, dim=-1)
    sl = torch.gather(labels, 1, idx)
    disc = torch.tensor(1) / torch.log2(torch.arange(L, dtype=torch.float) + 2.0)
    gains = torch.pow(torch.tensor(2.0), sl.to(torch.float32)) - 1.0
    cum = torch.cumsum((gains * disc)[:, :int(np.max(topn))], dim=1)
    return cum[:, torch.tensor(topn, dtype=torch.long) - 1]


def ndcg(labels: torch.Tensor, predictions: torch.Tensor, topn: Sequence[int]) -> torch.Tensor:
    """normalized_discounted_cumulative_gain, weights=None branch (metrics.py:486-494)."""
    labels, predictions, topn = _prepare(labels, predictions, topn)
    d = _dcg(predictions, labels, topn)
    i = _dcg(labels, labels, topn)
    return torch.mean(_safe_div(d, i), dim=0)


def argsort_desc(labels: torch.Tensor, predictions: torch.Tensor) -> torch.Tensor:
    """The permutation metrics sort with (metrics.py:208)."""
    labels, predictions, _ = _prepare(labels, predictions, [1])
    return predictions.sort(descending=True, dim=-1)[1]


def mrr(labels: torch.Tensor, predictions: torch.Tensor, topn: Sequence[int]) -> torch.Tensor:
    """mean_reciprocal_rank (metrics.py:268-298)."""
    L = predictions.shape[-1]
    labels, predictions, topn = _prepare(labels, predictions, topn)
    _, idx = predictions.sort(descending=True, dim=-1)
    rel = torch.ge(torch.gather(labels, 1, idx), 1.0).float()
    rr = 1.0 / torch.arange(1, L + 1, dtype=torch.float32)
    return torch.mean(torch.max(rel * rr, dim=1, keepdim=True).values).repeat(len(topn))


def err(labels: torch.Tensor, predictions: torch.Tensor, topn: Sequence[int], max_label: float) -> torch.Tensor:
    """expected_reciprocal_rank (metrics.py:300-336)."""
    labels, predictions, topn = _prepare(labels, predictions, topn)
    _, idx = predictions.sort(descending=True, dim=-1)
    sl = torch.gather(labels, 1, idx)
    L = sl.shape[-1]
    two = torch.as_tensor(2.0)
    rel = (torch.pow(two, sl) - 1) / torch.pow(two, torch.as_tensor(max_label))
    non_rel = torch.cumprod(1.0 - rel, dim=1) / (1.0 - rel)
    rr = 1.0 / torch.arange(1, L + 1, dtype=torch.float32)
    outs = []
    for n in topn:
        m = torch.ge(rr, 1.0 / n).float()
        outs.append(torch.sum(rel * non_rel * rr * m, dim=1, keepdim=True))
    return torch.mean(torch.stack(outs, dim=0), dim=1).view(-1)


def mean_average_precision(labels: torch.Tensor, predictions: torch.Tensor, topn: Sequence[int]) -> torch.Tensor:
    """mean_average_precision (metrics.py:408-453), weights = 1: loops over lists and ranks (small cases only)."""
    labels, predictions, topn = _prepare(labels, predictions, topn)
    _, idx = predictions.sort(descending=True, dim=-1)
    sl = torch.gather(labels, 1, idx).numpy()
    vals = []
    for row in sl:
        hits, acc = 0, 0.0
        for r, l in enumerate(row):
            if l >= 1.0:
                hits += 1
                acc += np.float32(hits) / np.float32(r + 1)
        vals.append(acc / hits if hits else 0.0)
    return torch.tensor(np.mean(np.asarray(vals, np.float32)), dtype=torch.float32).repeat(len(topn))


def ordered_pair_accuracy(labels: torch.Tensor, predictions: torch.Tensor, topn: Sequence[int]) -> torch.Tensor:
    """ordered_pair_accuracy (metrics.py:531-568), weights = 1: correctly ordered valid pairs / (B * L * L)."""
    clean, predictions, topn = _prepare(labels, predictions, topn)
    y, c, s = labels.numpy(), clean.numpy(), predictions.numpy()
    B, L = s.shape
    n = 0
    for b in range(B):
        for i in range(L):
            for j in range(L):
                if y[b, i] == c[b, i] and y[b, j] == c[b, j] and c[b, i] > c[b, j] and s[b, i] > s[b, j]:
                    n += 1
    return torch.tensor(n / float(B * L * L), dtype=torch.float32).repeat(len(topn))


def average_relevance_position(labels: torch.Tensor, predictions: torch.Tensor, topn: Sequence[int]) -> torch.Tensor:
    """average_relevance_position (metrics.py:338-370), weights = 1 (topn only sets the length of the result)."""
    labels, predictions, topn = _prepare(labels, predictions, topn)
    _, idx = predictions.sort(descending=True, dim=-1)
    sl = torch.gather(labels, 1, idx)
    pos = torch.arange(1, sl.shape[1] + 1, dtype=torch.float)
    return torch.mean(_safe_div(torch.sum(pos * sl, 1, keepdim=True), torch.sum(sl, 1, keepdim=True))).repeat(len(topn))


def precision_whole_list(labels: torch.Tensor, predictions: torch.Tensor) -> torch.Tensor:
    """precision (metrics.py:373-405), weights = 1: ONE scalar - the share of relevant documents in the whole list, 0 for lists
    without one, batch mean (topn is not applied by the reference)."""
    labels, predictions, _ = _prepare(labels, predictions, [1])
    rel = torch.ge(labels, 1.0).float()
    per_list = torch.sum(rel, 1, keepdim=True) / float(rel.shape[1])
    has = _safe_div(torch.sum(rel, 1, keepdim=True), torch.sum(rel, 1, keepdim=True))
    return torch.mean(per_list * has)


def validation(params, F_, hidden, features, docids, labels_LB, topn=(1, 3, 5, 10), max_label=4.0, act="elu"):
    """*.validation (ipw_rank.py:184-211): returns UNMASKED scores + metrics on masked scores (Appendix A.10)."""
    with torch.no_grad():
        p = torch.as_tensor(params, dtype=torch.float32)
        scores = ranking_scores(p, F_, hidden, features, docids, act)
        labels = torch.from_numpy(np.ascontiguousarray(np.transpose(labels_LB))).float()
        masked = mask_padding(scores, docids, np.asarray(features).reshape(-1, F_).shape[0])
        out = dict(scores=scores.numpy(), masked=masked.numpy(), ndcg=ndcg(labels, masked, topn).numpy(),
                   mrr=mrr(labels, masked, topn).numpy(), err=err(labels, masked, topn, max_label).numpy(),
                   argsort=argsort_desc(labels, masked).numpy())
    return out
