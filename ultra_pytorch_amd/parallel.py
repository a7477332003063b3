"""Data parallelism over queries: one process per GPU, parameters / optimizer state / EM state
replicated, ONE sum all-reduce (RCCL over xGMI; gloo in CPU tests) per step of the flat buffer
    [ unscaled parameter gradients (P) | loss_sum, D, loss2_sum, D2 | per-position sums (2L) ]
after which every rank applies the identical normalisation, global-norm clip and optimizer step.  Because the
loss kernels keep the global normalisers OUT of the gradients (they emit gradient x D), the sharded sum is exactly
the single-process quantity: no per-shard averaging error.

The sum itself is ONE kernel of libultr_hip.so (PeerComm below -> ultr_comm_allreduce: every rank reads every peer's
vector over xGMI through hipIpc-mapped exchange buffers, adds in rank order and emits the sum-of-squares partials of the
reduced gradient); torch.distributed (backend "nccl" = RCCL on ROCm, gloo in tests) only moves the 64-byte handles at
start-up and is the fallback collective when the peer path is unavailable or fails its self-test.
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


class PeerComm:
    """ultr_comm_* (include/ultr_hip.h, section e) for one rank: the exchange buffer, the mapped peers, the step counter.

    PeerComm.create returns None (on EVERY rank, by agreement through the process group) when the peer path cannot be
    used - IPC export/import failed, or the start-up self-test (a full all-reduce of a known pattern) did not produce the
    expected sum in bounded time - and the caller then keeps its collective-library path."""

    def __init__(self, lib, handle, rank, world, n_floats):
        self.lib, self.h, self.rank, self.world, self.n = lib, handle, rank, world, int(n_floats)
        self.step = 0

    @classmethod
    def create(cls, pg, n_floats, device, selftest=True):
        import ctypes
        import torch
        import torch.distributed as dist
        from . import _lib
        if os.environ.get("ULTR_DP_COMM", "peer") != "peer":
            return None
        lib = _lib.load()
        rank, world = dist.get_rank(pg), dist.get_world_size(pg)
        if world > _lib.COMM_MAX_WORLD:
            return None
        cpu_pg = dist.get_backend(pg) == "gloo"

        def agree(ok):  # every rank must take the same branch
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if cpu_pg else device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=pg)
            return bool(int(t.item()))

        h = ctypes.c_void_p()
        ok = lib.ultr_comm_create(rank, world, int(n_floats), ctypes.byref(h)) == 0
        blob = ctypes.create_string_buffer(_lib.COMM_HANDLE_BYTES)
        ok = ok and lib.ultr_comm_export(h, blob) == 0
        blobs = [None] * world
        dist.all_gather_object(blobs, blob.raw if ok else None, group=pg)
        ok = ok and all(b is not None for b in blobs)
        if ok:
            for p in range(world):
                if p != rank and lib.ultr_comm_import(h, p, ctypes.create_string_buffer(blobs[p], _lib.COMM_HANDLE_BYTES)) != 0:
                    ok = False
        if not agree(ok):
            if h:
                lib.ultr_comm_destroy(h)
            return None
        self = cls(lib, h, rank, world, n_floats)
        if selftest:
            n = min(self.n, 5000)
            idx = torch.arange(n, dtype=torch.float32, device=device)
            src = (idx % 97.0) * float(rank + 1) + float(rank)
            out = torch.empty_like(src)
            ws = torch.zeros((n + 63) // 64 + 4, dtype=torch.float32, device=device)
            good = False
            try:
                self.allreduce(src, n, n, out, ws)
                if lib.ultr_comm_status(h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0:
                    want = (idx % 97.0) * float(world * (world + 1) // 2) + float(world * (world - 1) // 2)
                    sq = float((want.double() ** 2).sum())
                    good = bool(torch.equal(out, want)) and abs(float(ws[:(n + 63) // 64].double().sum()) - sq) <= 1e-4 * sq
            except Exception:
                good = False
            if not agree(good):
                self.close()
                return None
        return self

    def allreduce(self, src, n, n_params, out, sumsq_ws):
        """out[:n] = sum over ranks of src[:n] (src / out may be the same tensor); sum-of-squares partials -> sumsq_ws."""
        import ctypes
        import torch
        from . import _lib
        _lib.check(self.lib.ultr_comm_allreduce(self.h, self.step, ctypes.c_void_p(src.data_ptr()), int(n), int(n_params),
                                                ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(sumsq_ws.data_ptr()),
                                                (int(n) + 63) // 64,
                                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "ultr_comm_allreduce")
        self.step += 1

    def status(self):
        import ctypes
        import torch
        return int(self.lib.ultr_comm_status(self.h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def close(self):
        if self.h:
            self.lib.ultr_comm_destroy(self.h)
            self.h = None
