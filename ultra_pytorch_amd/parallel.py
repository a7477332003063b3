"""Data parallelism over queries: one process per GPU, parameters / optimizer state / EM state
replicated, ONE sum all-reduce (RCCL over xGMI; gloo in CPU tests) per step of the flat buffer
    [ unscaled parameter gradients (P) | loss_sum, D, loss2_sum, D2 | per-position sums (2L) ]
after which every rank applies the identical normalisation, global-norm clip and optimizer step.  Because the
loss kernels keep the global normalisers OUT of the gradients (they emit gradient x D), the sharded sum is exactly
the single-process quantity: no per-shard averaging error.

Host-side helpers only; the collective itself is torch.distributed (backend "nccl" = RCCL on ROCm).
"""
import os

import numpy as np


def init_process_group_from_env(backend=None):
    """Join the job torchrun started (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).  Returns (rank, world, local_rank, group)."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and os.environ.get("ULTR_FORCE_DP", "0") != "1":
        return rank, world, local_rank, None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank, dist.group.WORLD


def shard_bounds(batch, rank, world):
    """Lists [lo, hi) of the global batch that `rank` owns (contiguous, remainder spread over the first ranks)."""
    base, rem = divmod(int(batch), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_input_feed(input_feed, letor_features_name, docid_inputs_name, labels_name, list_size, rank, world):
    """Cut rank's lists out of a GLOBAL input_feed (the reference feed's layout) and compact its feature rows:
    returns a feed with the same keys whose docids index the local feature matrix and whose PAD id is the local
    n_docs — i.e. exactly what a feed built for B/world queries would have produced."""
    feats = np.asarray(input_feed[letor_features_name])
    n_docs = feats.shape[0] if feats.ndim == 2 else 0
    ids = np.stack([np.asarray(input_feed[docid_inputs_name[l]]) for l in range(list_size)]).astype(np.int64)  # [L, B]
    lo, hi = shard_bounds(ids.shape[1], rank, world)
    loc = ids[:, lo:hi]
    used = np.unique(loc[loc < n_docs])
    remap = np.full(n_docs + 1, -1, dtype=np.int64)
    remap[used] = np.arange(used.size)
    remap[n_docs] = used.size  # PAD -> local PAD id
    out = {letor_features_name: feats[used] if n_docs > 0 else feats}
    for l in range(list_size):
        out[docid_inputs_name[l]] = remap[loc[l]].astype(np.float32)
        out[labels_name[l]] = np.asarray(input_feed[labels_name[l]])[lo:hi]
    return out
