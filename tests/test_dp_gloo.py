"""Data-parallel protocol on CPU: world_size 2, gloo.  Each rank takes its shard of ONE global batch
(parallel.shard_input_feed), produces the [unscaled grads | step tail] vector the HIP backward produces
(here from the oracle), sum-all-reduces it, and normalises — the result must equal the single-process
full-batch step."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F, HIDDEN, B, L = 12, [8, 4], 6, 5


def make_global(seed):
    rng = np.random.RandomState(seed)
    n_docs = B * L - 2
    feats = rng.uniform(-1, 1, size=(n_docs, F)).astype(np.float32)
    ids = rng.permutation(B * L)
    ids = np.where(ids >= n_docs, n_docs, ids).reshape(L, B)
    clicks = (rng.uniform(size=(L, B)) < 0.4).astype(np.float32)
    clicks[0, :] = 1.0
    names_d = ["docid_input%d" % l for l in range(L)]
    names_l = ["label%d" % l for l in range(L)]
    feed = {"letor_features": feats}
    for l in range(L):
        feed[names_d[l]] = ids[l].astype(np.float32)
        feed[names_l[l]] = clicks[l]
    return feed, names_d, names_l


def local_vector(algo, params, feed, names_d, names_l, aux, batch_total):
    """What ultr_dnn_backward leaves in `grads` on one rank: unscaled gradient + [loss_sum, D, 0, 0 | 2L sums]."""
    from oracle import ultr_oracle as O
    feats = feed["letor_features"]
    ids = np.stack([feed[n] for n in names_d]).astype(np.int64)
    y_LB = np.stack([feed[n] for n in names_l]).astype(np.float32)
    p = torch.tensor(params, requires_grad=True)
    scores = O.ranking_scores(p, F, HIDDEN, feats, ids)
    tail = torch.zeros(4 + 2 * L)
    if algo == "softmax":
        y = torch.tensor(y_LB.T.copy())
        w = (y + 1e-7)
        loss_sum = -(w * torch.log_softmax(scores, -1)).sum()
        tail[0], tail[1] = loss_sum.detach(), w.sum()
        obj = loss_sum
    elif algo == "lambdarank":
        # every term carries 1 / IDCG_batch (lambda_rank.py:247-266, 280-282: ONE scalar over the whole batch): a shard emits the
        # IDCG-free sums (its own loss x its own IDCG) and its IDCG share in tail[1]; the update divides by the reduced IDCG
        tp, tm = torch.tensor(aux[:L]), torch.tensor(aux[L:])
        y = torch.tensor(y_LB.T.copy())
        loss, PL, tpl, tml = O.lambdarank_loss(scores, y, tp, tm)
        ideal, _ = torch.sort(y, dim=1, descending=True)
        idcg = torch.sum(O._safe_div(torch.pow(torch.tensor(2.0), ideal) - 1.0, torch.log(torch.arange(1, L + 1, dtype=torch.float32) + 1)))
        obj = loss * idcg
        tail[0], tail[1] = obj.detach(), idcg
        tail[4:4 + L], tail[4 + L:] = tpl.detach() * idcg, tml.detach() * idcg
    elif algo == "dla":
        # two coupled softmax losses (dla.py:196-237), each with its own batch-global normaliser: [X_rank, D_rank, X_exam, D_exam]
        # and, per position, the un-normalised gradient of the exam loss with respect to the (batch-independent) propensity
        y = torch.tensor(y_LB.T.copy())
        q = torch.tensor(aux[:L + 1])
        prop = O.denoising_net(q, scores.shape[0], L).clone().requires_grad_(True)
        with torch.no_grad():
            pw = O.normalized_weights(O.logits_to_prob(prop))
            rw = O.normalized_weights(O.logits_to_prob(scores))
        w1, w2 = (y + 1e-7) * pw, (y + 1e-7) * rw
        x_rank = -(w1 * torch.log_softmax(scores, -1)).sum()
        x_exam = -(w2 * torch.log_softmax(prop, -1)).sum()
        (gprop,) = torch.autograd.grad(x_exam, prop)
        tail[0], tail[1], tail[2], tail[3] = x_rank.detach(), w1.sum(), x_exam.detach(), w2.sum()
        tail[4:4 + L] = gprop.sum(0)
        obj = x_rank
    else:  # pairdebias: linear in the pair sums; the xB factor uses the GLOBAL batch
        tp, tm = torch.tensor(aux[:L]), torch.tensor(aux[L:])
        loss, PL, tpl, tml = O.pairdebias_loss(scores, torch.tensor(y_LB), tp, tm)
        scale = float(batch_total) / scores.shape[0]
        obj = loss * scale
        tail[0] = obj.detach()
        tail[4:4 + L], tail[4 + L:] = tpl.detach() * scale, tml.detach() * scale
    (g,) = torch.autograd.grad(obj, p)
    return torch.cat([g, tail])


def worker(rank, world, port, algo, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from oracle import ultr_oracle as O
    from ultra_pytorch_amd import parallel
    r, w, _, pg = parallel.init_process_group_from_env(backend="gloo")
    assert (r, w) == (rank, world) and pg is not None
    feed, names_d, names_l = make_global(3)
    params = O.init_params(F, HIDDEN, seed=1)
    aux = np.linspace(0.9, 1.1, 2 * L).astype(np.float32)
    local = parallel.shard_input_feed(feed, "letor_features", names_d, names_l, L, rank, world)
    lo, hi = parallel.shard_bounds(B, rank, world)
    assert np.stack([local[n] for n in names_d]).shape == (L, hi - lo)
    vec = local_vector(algo, params, local, names_d, names_l, aux, B)
    dist.all_reduce(vec, group=pg)  # the ONE collective of the step
    if rank == 0:
        out.put(vec.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("algo", ["softmax", "pairdebias", "lambdarank", "dla"])
def test_sharded_allreduce_equals_single_process(algo):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29611 + ["softmax", "pairdebias", "lambdarank", "dla"].index(algo)
    procs = [ctx.Process(target=worker, args=(r, 2, port, algo, out)) for r in range(2)]
    [p.start() for p in procs]
    vec = out.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    sys.path.insert(0, ROOT)
    from oracle import ultr_oracle as O
    feed, names_d, names_l = make_global(3)
    params = O.init_params(F, HIDDEN, seed=1)
    ids = np.stack([feed[n] for n in names_d]).astype(np.int64)
    y_LB = np.stack([feed[n] for n in names_l]).astype(np.float32)
    P = params.size
    if algo == "softmax":
        ref = O.train_step_softmax(params, np.zeros_like(params), F, HIDDEN, feed["letor_features"], ids, y_LB)
        D = vec[P + 1]
        np.testing.assert_allclose(vec[:P] / D, ref["grads"], rtol=1e-4, atol=1e-6)
        assert abs(vec[P] / D - ref["loss"]) < 1e-5
    elif algo == "lambdarank":
        aux = np.linspace(0.9, 1.1, 2 * L).astype(np.float32)
        ref = O.lambdarank_step(params, np.zeros_like(params), aux[:L], aux[L:], F, HIDDEN, feed["letor_features"], ids, y_LB)
        D = vec[P + 1]  # the batch-global IDCG, summed over the shards
        np.testing.assert_allclose(vec[:P] / D, ref["grads"], rtol=1e-4, atol=1e-5 * np.abs(ref["grads"]).max())
        assert abs(vec[P] / D - ref["loss"]) <= 1e-5 * max(1.0, abs(ref["loss"]))
        tpl, tml = torch.tensor(vec[P + 4:P + 4 + L] / D), torch.tensor(vec[P + 4 + L:] / D)
        np.testing.assert_allclose(O.em_update(torch.tensor(aux[:L]), tpl, 0.05, 1, safe=True).numpy(), ref["t_plus"].ravel(), atol=1e-6)
        np.testing.assert_allclose(O.em_update(torch.tensor(aux[L:]), tml, 0.05, 1, safe=True).numpy(), ref["t_minus"].ravel(), atol=1e-6)
    elif algo == "dla":
        aux = np.linspace(0.9, 1.1, 2 * L).astype(np.float32)
        q = aux[:L + 1]
        ref = O.dla_step(params, q, F, HIDDEN, feed["letor_features"], ids, y_LB)
        d1, d2 = vec[P + 1], vec[P + 3]
        np.testing.assert_allclose(vec[:P] / d1, ref["grads"], rtol=1e-4, atol=1e-6)
        assert abs(vec[P] / d1 - ref["rank_loss"]) < 1e-5 and abs(vec[P + 2] / d2 - ref["exam_loss"]) < 1e-5
        assert abs(vec[P + 2] / d2 + vec[P] / d1 - ref["loss"]) < 1e-5
        # DenoisingNet backward from the reduced per-position sums (what update_kernel's block 0 does): dW_l = ELU'(W_l + b) s_l / D2
        z = q[:L] + q[L]
        gw = (vec[P + 4:P + 4 + L] / d2) * np.where(z > 0, 1.0, np.exp(z))
        np.testing.assert_allclose(np.concatenate([gw, [gw.sum()]]), ref["prop_grads"], rtol=1e-4, atol=1e-6)
    else:
        aux = np.linspace(0.9, 1.1, 2 * L).astype(np.float32)
        ref = O.pairdebias_step(params, np.zeros_like(params), aux[:L], aux[L:], F, HIDDEN, feed["letor_features"], ids, y_LB)
        np.testing.assert_allclose(vec[:P], ref["grads"], rtol=1e-4, atol=1e-4 * np.abs(ref["grads"]).max())
        assert abs(vec[P] - ref["loss"]) <= 1e-5 * abs(ref["loss"])
        # EM update from the reduced per-position sums == the single-process EM update
        tpl, tml = torch.tensor(vec[P + 4:P + 4 + L]), torch.tensor(vec[P + 4 + L:])
        tp2 = O.em_update(torch.tensor(aux[:L]), tpl, 0.05, 1)
        np.testing.assert_allclose(tp2.numpy(), ref["t_plus"].ravel(), atol=1e-6)


def test_shard_bounds_cover_batch():
    from ultra_pytorch_amd import parallel
    for Bt in (1, 5, 8, 256):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(Bt, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == Bt
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
