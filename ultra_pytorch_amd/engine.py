"""StepEngine — one training / validation step of the hot path on one GPU (rank).

Owns the per-shape device workspaces (torch tensors = plumbing for HBM allocations) and
sequences the C-ABI calls on the current HIP stream:

    train:  ultr_dnn_forward -> ultr_<loss> -> ultr_dnn_backward [-> ultr_comm_allreduce (one kernel over xGMI; or
            the process group's all-reduce + ultr_grad_sumsq)] -> ultr_apply_update
    valid:  ultr_dnn_forward -> ultr_ndcg

Nothing here synchronises with the host; the caller decides when to read `scalars`
(the reference's `loss.item()` is the only sync, base_algorithm.py / ipw_rank.py:182).
"""
import ctypes
import os
import random
import time

import numpy as np
import torch

from . import _lib, hip_ops

ALGOS = {"softmax": _lib.ALGO_SOFTMAX, "dla": _lib.ALGO_DLA, "pairdebias": _lib.ALGO_PAIRDEBIAS,
         "lambdarank": _lib.ALGO_LAMBDARANK, "regem": _lib.ALGO_REGEM}


def _f32(n, device, zero=False):
    n = max(int(n), 1)
    return (torch.zeros if zero else torch.empty)(n, dtype=torch.float32, device=device)


class StepEngine:
    def __init__(self, shape, batch, list_size, device, algo="softmax", optimizer="ada", learning_rate=0.05,
                 max_gradient_norm=5.0, ranker_loss_weight=1.0, propensity_learning_rate=None, em_step_size=0.05,
                 regulation_p=1.0, sigma=1.0, logits_to_prob="softmax", process_group=None, rng_seed=0, l2_loss=0.0,
                 batch_total=None, comm=None, no_peer_comm=False):
        if not torch.cuda.is_available():
            raise RuntimeError("ultra_pytorch_amd needs an MI355X/ROCm GPU: there is no CPU fallback")
        self.shape, self.B, self.L, self.device = shape, int(batch), int(list_size), device
        shape.lib.ultr_config_reload()  # the ULTR_* knobs are read when an engine is built, never per step
        self.N = self.B * self.L
        self.algo = algo
        self.sigma = float(sigma)
        self.l2p = 1 if logits_to_prob == "sigmoid" else 0
        self.pg = process_group
        self.world = 1 if process_group is None else torch.distributed.get_world_size(process_group)
        self.rank = 0 if process_group is None else torch.distributed.get_rank(process_group)
        # RegressionEM's Bernoulli draw when no uniforms are injected: Philox keyed by (seed, step); every shard gets its own
        # key, otherwise list i of every rank would draw the same uniforms
        self.rng_seed, self.rng_step = (int(rng_seed) + 0x9E3779B1 * self.rank) & 0xFFFFFFFFFFFFFFFF, 0
        # PairDebias' xB factor (base_algorithm.py:242-248) is the GLOBAL batch: shards may be uneven (parallel.shard_bounds),
        # so it is the sum of the local batches, not B * world
        # (`batch_total` given: the caller already agreed on it - learning_algorithm.BaseAlgorithm does, per step, so that a
        # cached engine can never sit on a stale value or enter a collective its peers do not)
        self.batch_total = self.B if batch_total is None else int(batch_total)
        if process_group is not None and batch_total is None:
            cpu_pg = torch.distributed.get_backend(process_group) == "gloo"
            t = torch.tensor([self.B], dtype=torch.int64, device="cpu" if cpu_pg else device)
            torch.distributed.all_reduce(t, group=process_group)
            self.batch_total = int(t.item())
        P, tail = shape.n_params, hip_ops.tail_floats(self.L)
        self.P, self.tail = P, tail
        # the gradient exchange: a communicator handed in by the caller (one per process group, shared by every engine of an
        # algorithm object), or this engine's own
        self.comm, self._own_comm = comm, False
        if process_group is not None and comm is None and not no_peer_comm:
            from . import parallel
            self.comm = parallel.PeerComm.create(process_group, P + tail, device)
            self._own_comm = self.comm is not None
        if self.comm is not None and self.comm.n < P + tail:
            raise ValueError("the shared communicator is too small for this model")
        self._alloc(shape, device)
        self.loss_ws = _f32(hip_ops.loss_workspace_bytes(self.B, self.L) // 4, device, zero=True)
        self.scores = _f32(self.N, device).view(self.B, self.L)
        self.dscores = _f32(self.N, device).view(self.B, self.L)
        self.grads = _f32(P + tail, device, zero=True)
        self.scalars = _f32(16, device, zero=True)
        # step report in HOST-mapped pinned memory (ultr_update_desc::host_scalars): the update kernel's block 0 writes the step
        # scalars, the exchange's status word and - last - the step's sequence number there; read_loss() spins on the sequence
        # number instead of a stream synchronisation + device-to-host copy (the reference's loss.item(): 16-18 us -> one PCIe write)
        self._hs = torch.zeros(16, dtype=torch.float32).pin_memory()
        self._hs_f = self._hs.numpy()
        self._hs_u = self._hs_f.view(np.uint32)
        self._seq = 0
        self._host_report = os.environ.get("ULTR_HOST_REPORT", "1") != "0"
        u = _lib.UpdateDesc()
        u.algo = ALGOS[algo]
        u.optimizer = _lib.OPT_SGD if optimizer == "sgd" else _lib.OPT_ADAGRAD
        u.list_size = self.L
        u.logits_to_prob = self.l2p
        u.n_params = P
        u.learning_rate = float(learning_rate)
        u.max_gradient_norm = float(max_gradient_norm)
        u.adagrad_eps = 1e-10
        u.ranker_loss_weight = float(ranker_loss_weight)
        plr = propensity_learning_rate
        u.propensity_learning_rate = float(learning_rate if plr is None or plr < 0 else plr)
        u.em_step_size = float(em_step_size)
        u.regulation_p = float(regulation_p)
        u.l2_loss = float(l2_loss)
        u.guard = None
        u.range_flag = None
        u.host_scalars = self._hs.data_ptr() if self._host_report else None
        u.seq = 0
        self.udesc = u
        self._args = None
        self._cached = (None,) * 8  # the tensor objects behind the pointer fields of self._args (train_step)
        self.next_click_source = None  # a DeviceClickFeed (set per step by the plugin algorithm): its next batch is drawn behind the step

    def _alloc(self, shape, device):
        self.saved = _f32(shape.saved_bytes(self.N) // 4, device)
        self.bwd_ws = _f32(shape.bwd_workspace_bytes(self.N) // 4, device)

    # ---- forward only (validation / DNN.build) -------------------------------------------------
    def forward(self, params, features, n_docs, docids, scores=None, train=False):
        scores = self.scores if scores is None else scores
        hip_ops.dnn_forward(self.shape, params, features, n_docs, docids, self.B, self.L, scores,
                            self.saved if train else None)
        return scores

    # ---- loss stage ------------------------------------------------------------------------------
    def loss(self, labels, aux=None, ipw_table=None, pw=None, uniforms=None):
        B, L = self.B, self.L
        if self.algo == "softmax":
            hip_ops.softmax_ce(self.scores, labels, B, L, self.dscores, self.loss_ws, pw=pw, ipw_table=ipw_table)
        elif self.algo == "dla":
            hip_ops.dla_loss(self.scores, labels, aux, self.l2p, B, L, self.dscores, self.loss_ws)
        elif self.algo == "pairdebias":
            hip_ops.pairdebias_loss(self.scores, labels, aux[:L], aux[L:], B, L, self.batch_total, self.dscores, self.loss_ws)
        elif self.algo == "lambdarank":
            hip_ops.lambdarank_loss(self.scores, labels, aux[:L], aux[L:], self.sigma, B, L, self.dscores, self.loss_ws)
        elif self.algo == "regem":
            hip_ops.regem_loss(self.scores, labels, aux, B, L, self.dscores, self.loss_ws, uniforms=uniforms,
                               seed=self.rng_seed, step=self.rng_step)
            self.rng_step += 1
        else:
            raise ValueError(self.algo)

    def backward(self, params, features, n_docs, docids):
        hip_ops.dnn_backward(self.shape, params, features, n_docs, docids, self.B, self.L, self.saved, self.dscores,
                             self.loss_ws, self.bwd_ws, self.grads)
        self.dp_reduce()

    def dp_reduce(self):
        """Queries shard across ranks: ONE sum of [grads | step tail] per step, then the sum-of-squares partials of the
        reduced gradient for the clip."""
        if self.pg is None:
            return
        if self.comm is not None:  # one kernel: publish, xGMI reads of every peer, fixed-order sum, partials
            self.comm.allreduce(self.grads, self.P + self.tail, self.P, self.grads, self.bwd_ws)
        else:  # the process group's collective (RCCL over xGMI; gloo in tests)
            torch.distributed.all_reduce(self.grads, group=self.pg)
            hip_ops.grad_sumsq(self.grads, self.P, self.L, self.bwd_ws)

    def update(self, params, state, aux=None):
        hip_ops.apply_update(self.shape, self.udesc, params, state, self.grads, aux, self.bwd_ws, self.scalars)

    def _next_seq(self):
        self._seq = (self._seq % 0xFFFFFFFF) + 1  # 1 .. 2^32-1: never 0, the buffer's initial content
        self.udesc.seq = self._seq

    def read_scalars(self, timeout_s=60.0, scheduled=False):
        """The step scalars of the LAST queued step as a numpy array ([0] loss, [1] gradient norm, [2] clip coefficient, [3] D,
        [4] rank_loss, [5] exam_loss, [6] propensity gradient norm, [7] sum g^2) - the reference's `loss.item()`.  Waits for the
        update kernel's report in host-mapped memory (no stream synchronisation); raises if the gradient exchange of a
        data-parallel step timed out on any rank (the update was then NOT applied, ultr_update_desc::guard)."""
        if not self._host_report:
            torch.cuda.current_stream().synchronize()
            if self.comm is not None and self.comm.status() != 0:
                raise _lib.UltrHipError("data-parallel gradient exchange timed out (ULTR_E_COMM_TIMEOUT): the update was not applied")
            return self.scalars[:8].cpu().numpy()
        seq, u, spins, t0 = self._seq, self._hs_u, 0, None
        if seq == 0:
            raise RuntimeError("read_scalars() before the first train_step()")
        while int(u[9]) != seq:
            spins += 1
            if spins & 0x3FF == 0:  # look at the clock every 1024 polls only
                now = time.perf_counter()
                t0 = now if t0 is None else t0
                if now - t0 > timeout_s:
                    raise _lib.UltrHipError("no step report from the GPU within %.0f s (step %d)" % (timeout_s, seq))
        self._raise_on_status(int(u[8]), scheduled)
        return self._hs_f[:8].copy()

    H3_RANGE, H3_NEAR = 0x100, 0x200  # include/ultr_hip.h: ULTR_STATUS_H3_RANGE / ULTR_STATUS_H3_NEAR

    DP_STATUS_CADENCE = 64  # data parallel: every rank examines the step report of the same steps (train_step)

    def _raise_on_status(self, st, scheduled=False):
        """Act on the status word of the latest full step report (host_scalars[8]).
        Data parallel: the switch to the fp32 products changes the bits of the following steps, so every rank must make it at the
        SAME step.  The replicas are bit-identical, hence every rank's update kernel raises the NEAR bit in the report of the same
        step - but ranks read reports when their callers ask for a loss (rank 0 may log every step, the others never).  There the
        switch is only made from train_step's own scheduled look at the report (every DP_STATUS_CADENCE-th step, on every rank
        alike: ADVICE r05); a read in between leaves the NEAR bit alone (the copies are still exact: one step moves a weight by at
        most lr x the clip norm, 64 steps by far less than the 64 .. 128 margin)."""
        if st == 0:
            return
        if self.pg is not None and not scheduled:
            st &= ~self.H3_NEAR
            if st == 0:
                return
        if st & (self.H3_RANGE | self.H3_NEAR):
            # a hidden weight is near (>= 64) or beyond (>= 128) the range of the split-half weight copies.  Near: every copy is
            # still exact and one optimizer step moves a weight by at most lr x the clip norm - switch to the fp32 products now
            # and go on, as the reference would (base_algorithm.py:208-226 trains any weight magnitude).  Beyond, with the
            # split-half products still on: some step already read an overflowed copy.
            switched = hip_ops.fall_back_to_fp32_products(self.shape, "a hidden weight reached |w| >= 64 during training (the "
                                                          "split-half weight copies cover |w| < 128)")
            if (st & self.H3_RANGE) and switched:
                raise _lib.UltrHipError("a hidden weight jumped to |w| >= 128 within the steps between two reads of the step report, "
                                        "outside the range of the split-half (fp16 hi / lo) weight copies (ULTR_STATUS_H3_RANGE): the "
                                        "last steps are not to be trusted - restart from the last checkpoint (this model is on the fp32 "
                                        "matrix-core products now: ULTR_MODEL_FP32_PRODUCTS)")
            st &= ~(self.H3_RANGE | self.H3_NEAR)
            if st == 0:
                return
        raise _lib.UltrHipError("data-parallel gradient exchange timed out on some rank (ULTR_E_COMM_TIMEOUT): this and all "
                                "later updates were NOT applied - restart from the last checkpoint")

    def read_loss(self, timeout_s=60.0):
        """The loss of the last queued step (the reference's `loss.item()`).  Returns as soon as the loss is FINAL: on one GPU
        that is behind forward + loss (the weight-gradient launch reports it, include/ultr_hip.h: host_scalars[10]), while the
        step's reduction and update are still running - later work simply queues behind them on the stream.  The status word
        of the LATEST full report is examined on every call (it is one or two launches older than the loss: a failure -
        exchange timeout, split-half range - surfaces one step late at most; both conditions are sticky on the device)."""
        if not self._host_report:
            return float(self.read_scalars()[0])
        seq, u, spins, t0 = self._seq, self._hs_u, 0, None
        if seq == 0:
            raise RuntimeError("read_loss() before the first train_step()")
        while int(u[10]) != seq:
            spins += 1
            if spins & 0x3FF == 0:
                now = time.perf_counter()
                t0 = now if t0 is None else t0
                if now - t0 > timeout_s:
                    raise _lib.UltrHipError("no loss report from the GPU within %.0f s (step %d)" % (timeout_s, seq))
        self._raise_on_status(int(u[8]))
        return float(self._hs_f[0])

    def close(self):
        if self._own_comm and self.comm is not None:
            self.comm.close()
        self.comm = None

    def train_step(self, params, state, features, n_docs, docids, labels, aux=None, ipw_table=None, pw=None,
                   uniforms=None):
        """One full step through ONE C call (ultr_train_step); returns the device tensor of step scalars ([0] = loss)."""
        a = self._args
        if a is None:
            a = self._args = _lib.StepArgs()
            a.desc = ctypes.pointer(self.shape.desc)
            a.upd = ctypes.pointer(self.udesc)
            a.scores, a.dscores = self.scores.data_ptr(), self.dscores.data_ptr()
            a.saved, a.loss_ws, a.bwd_ws = self.saved.data_ptr(), self.loss_ws.data_ptr(), self.bwd_ws.data_ptr()
            a.grads, a.scalars = self.grads.data_ptr(), self.scalars.data_ptr()
            a.batch, a.list_size = self.B, self.L
            a.sigma = self.sigma
            # data parallel: with the peer exchange the whole sharded step is still ONE C call (backward -> exchange kernel ->
            # update); with the process group's all-reduce the call stops behind the backward and the host issues the rest
            a.skip_update = 1 if (self.pg is not None and self.comm is None) else 0
            a.comm = self.comm.h if self.comm is not None else None
            self._fn = self.shape.lib.ultr_train_step
            self._fn_feed = self.shape.lib.ultr_feed_train_step
            self._aref = ctypes.byref(a)
        a.batch_total = self.batch_total  # per step: uneven data-parallel shards may change it (PairDebias' xB factor)
        # pointer fields are rewritten only when the tensor OBJECT behind them changed (a ctypes field store + data_ptr() is ~0.3 us;
        # the same parameters / state / feature matrix / table come back every step, only the batch tensors alternate)
        c = self._cached
        a.params = params.data_ptr()  # (always: `.data = ...` re-seats a tensor object's storage)
        a.wt = hip_ops.weight_copy(self.shape).get(params).data_ptr()
        a.state = state.data_ptr() if state is not None else None
        if aux is not c[2]:
            a.aux = aux.data_ptr() if aux is not None else None
        if features is not c[3] or n_docs != a.n_docs:
            a.features = features.data_ptr() if n_docs > 0 else None
            a.n_docs = n_docs
        if docids is not c[4]:
            a.docids = docids.data_ptr()
        if labels is not c[5]:
            a.labels = labels.data_ptr()
        if pw is not c[6]:
            a.pw = pw.data_ptr() if pw is not None else None
        if ipw_table is not c[7]:
            a.ipw_table = ipw_table.data_ptr() if ipw_table is not None else None
            a.n_ipw = int(ipw_table.numel()) if ipw_table is not None else 0
        self._cached = (params, state, aux, features, docids, labels, pw, ipw_table)
        if self.algo == "regem":
            a.uniforms = uniforms.data_ptr() if uniforms is not None else None
            a.rng_seed, a.rng_step = self.rng_seed, self.rng_step
            self.rng_step += 1
        if self.comm is not None:
            a.comm_step = self.comm.step
            self.comm.step += 1
        if self.pg is not None and self._host_report and self._seq > 0 and self._seq % self.DP_STATUS_CADENCE == 0:
            self.read_scalars(scheduled=True)  # the same step on every rank: a switch to the fp32 products happens everywhere at once
        self._next_seq()
        src = self.next_click_source
        nxt = src.next_click_args() if src is not None else None
        if nxt is not None and a.skip_update == 0:
            rc = self._fn_feed(self._aref, nxt, hip_ops.raw_stream())
        else:
            rc = self._fn(self._aref, hip_ops.raw_stream())
            if nxt is not None and rc == 0:  # (process-group path: the draw follows the step's first half)
                rc = self.shape.lib.ultr_click_batch_args(nxt, hip_ops.raw_stream())
        if rc != 0:
            _lib.check(rc, "ultr_train_step")
        if self.pg is not None and self.comm is None:
            self.dp_reduce()
            self.update(params, state, aux)
        return self.scalars


class EvalEngine:
    """validation(): forward at max_candidate_num + padding mask + NDCG@topn."""

    def __init__(self, shape, batch, list_size, device, topn=(1, 3, 5, 10)):
        if not torch.cuda.is_available():
            raise RuntimeError("ultra_pytorch_amd needs an MI355X/ROCm GPU: there is no CPU fallback")
        self.shape, self.B, self.L, self.device = shape, int(batch), int(list_size), device
        self.topn = [int(t) for t in topn]
        self.scores = _f32(self.B * self.L, device).view(self.B, self.L)
        self.masked = _f32(self.B * self.L, device).view(self.B, self.L)
        self.order = torch.empty(self.B, self.L, dtype=torch.int32, device=device)
        self.ndcg = _f32(len(self.topn), device)
        self.ndcg_ws = _f32(self.B * len(self.topn), device)
        # the metric vector in HOST-mapped pinned memory + a sequence word (ultr_ndcg_report): read_ndcg() spins on the word instead
        # of a stream synchronisation + device-to-host copy per validation batch
        self._counter = torch.zeros(1, dtype=torch.int32, device=device)
        self._hs = torch.zeros(32, dtype=torch.float32).pin_memory()
        self._hs_f = self._hs.numpy()
        self._hs_u = self._hs_f.view(np.uint32)
        self._seq = 0
        self._topn_arr = (ctypes.c_int32 * len(self.topn))(*self.topn)
        self._lib = shape.lib

    def _ndcg(self, labels, docids, n_docs):
        self._seq = (self._seq % 0xFFFFFFFF) + 1
        _lib.check(self._lib.ultr_ndcg_report(self.scores.data_ptr(), labels.data_ptr(), docids.data_ptr(), int(n_docs), self.B, self.L,
                                              self._topn_arr, len(self.topn), self.ndcg.data_ptr(), self.order.data_ptr(),
                                              self.masked.data_ptr(), self.ndcg_ws.data_ptr(), self._counter.data_ptr(),
                                              self._hs.data_ptr(), self._seq, hip_ops.raw_stream()), "ultr_ndcg_report")

    def run(self, params, features, n_docs, docids, labels):
        """ONE host call (ultr_dnn_forward_ndcg): the forward, then the metric launch with its report in host-mapped memory."""
        self._seq = (self._seq % 0xFFFFFFFF) + 1
        wt = hip_ops.weight_copy(self.shape).get(params)
        _lib.check(self._lib.ultr_dnn_forward_ndcg(ctypes.byref(self.shape.desc), params.data_ptr(), wt.data_ptr() if wt is not None else None,
                                                   features.data_ptr() if n_docs > 0 else None, int(n_docs), docids.data_ptr(),
                                                   labels.data_ptr(), self.B, self.L, self.scores.data_ptr(), self._topn_arr, len(self.topn),
                                                   self.ndcg.data_ptr(), self.order.data_ptr(), self.masked.data_ptr(), self.ndcg_ws.data_ptr(),
                                                   self._counter.data_ptr(), self._hs.data_ptr(), self._seq, hip_ops.raw_stream()),
                   "ultr_dnn_forward_ndcg")
        return self.scores, self.ndcg

    def read_ndcg(self, timeout_s=60.0):
        """NDCG@topn of the LAST run() as a numpy array - the reference's `.item()` per metric (ipw_rank.py:204-210) without a stream
        synchronisation: waits for the launch's report in host-mapped memory."""
        seq, u, spins, t0 = self._seq, self._hs_u, 0, None
        if seq == 0:
            raise RuntimeError("read_ndcg() before the first run()")
        while int(u[16]) != seq:
            spins += 1
            if spins & 0x3FF == 0:
                now = time.perf_counter()
                t0 = now if t0 is None else t0
                if now - t0 > timeout_s:
                    raise _lib.UltrHipError("no NDCG report from the GPU within %.0f s" % timeout_s)
        return self._hs_f[:len(self.topn)].copy()


def _setrank_draw(list_size):
    """The reference's SetRank.build shuffles an index list on EVERY forward and never uses it (SetRank.py:245-246), but the
    call advances Python's global `random` stream - the one ClickSimulationFeed draws queries and clicks from.  Consuming
    the same draws keeps a seeded SetRank run on the reference's batches and clicks."""
    random.shuffle(list(range(int(list_size))))


class SetRankStepEngine(StepEngine):
    """The same step for the SetRank ranking model: ultr_setrank_forward -> ultr_<loss> ->
    ultr_setrank_backward -> [all-reduce] -> ultr_grad_sumsq -> ultr_apply_update.  Stage calls instead of ONE C call:
    at SetRank's size (hundreds of microseconds to milliseconds per step) the host is not on the critical path."""

    def __init__(self, shape, batch, list_size, device, **kw):
        self._sr_shape = shape
        super().__init__(shape, batch, list_size, device, **kw)

    def _alloc(self, shape, device):
        self.saved = _f32(shape.saved_bytes(self.N) // 4, device)
        self.sr_ws = _f32(shape.workspace_bytes(self.N) // 4, device)
        # sum-of-squares partials of ultr_grad_sumsq / ultr_apply_update
        self.bwd_ws = _f32((self.P + self.tail + 63) // 64 + self.P // 4096 + 16, device)
        self._flag_off = shape.range_flag_offset(self.N)
        self.saved[self._flag_off].zero_()  # (every forward zeroes it again: this is for a report before the first forward)

    def forward(self, params, features, n_docs, docids, scores=None, train=False):
        scores = self.scores if scores is None else scores
        _setrank_draw(self.L)
        hip_ops.setrank_forward(self.shape, params, features, n_docs, docids, self.B, self.L, scores, self.saved)
        return scores

    def backward(self, params, features, n_docs, docids):
        hip_ops.setrank_backward(self.shape, params, self.B, self.L, self.saved, self.dscores, self.loss_ws, hip_ops.loss_part_count(self.B),
                                 self.sr_ws, self.grads)
        if self.pg is not None:
            self.dp_reduce()
        else:
            hip_ops.grad_sumsq(self.grads, self.P, self.L, self.bwd_ws)

    def update(self, params, state, aux=None):
        check = _lib.check
        # the range word of this step's split-half planes (raised by ultr_setrank_forward) travels in the step report
        self.udesc.range_flag = self.saved.data_ptr() + 4 * self._flag_off
        check(self.shape.lib.ultr_apply_update(ctypes.byref(self.udesc), None, ctypes.c_void_p(params.data_ptr()), None,
                                               ctypes.c_void_p(state.data_ptr()) if state is not None else None,
                                               ctypes.c_void_p(self.grads.data_ptr()),
                                               ctypes.c_void_p(aux.data_ptr()) if aux is not None else None,
                                               ctypes.c_void_p(self.bwd_ws.data_ptr()), ctypes.c_void_p(self.scalars.data_ptr()),
                                               hip_ops.raw_stream()), "ultr_apply_update")

    def train_step(self, params, state, features, n_docs, docids, labels, aux=None, ipw_table=None, pw=None, uniforms=None):
        self._next_seq()
        self.forward(params, features, n_docs, docids, train=True)
        if self.algo == "regem":
            self.loss(labels, aux=aux, uniforms=uniforms)
        else:
            self.loss(labels, aux=aux, ipw_table=ipw_table, pw=pw)
        self.backward(params, features, n_docs, docids)
        self.update(params, state, aux)
        src = self.next_click_source
        nxt = src.next_click_args() if src is not None else None
        if nxt is not None:
            _lib.check(self.shape.lib.ultr_click_batch_args(nxt, hip_ops.raw_stream()), "ultr_click_batch")
        return self.scalars


class SetRankEvalEngine(EvalEngine):
    def __init__(self, shape, batch, list_size, device, topn=(1, 3, 5, 10)):
        super().__init__(shape, batch, list_size, device, topn=topn)
        self.saved = _f32(shape.saved_bytes(self.B * self.L) // 4, device)
        self._flag = self.saved[shape.range_flag_offset(self.B * self.L):][:1].view(torch.int32)
        self._checked = None  # (data_ptr, version) of the parameters whose split-half planes were last looked at

    def run(self, params, features, n_docs, docids, labels):
        _setrank_draw(self.L)
        hip_ops.setrank_forward(self.shape, params, features, n_docs, docids, self.B, self.L, self.scores, self.saved)
        # the split-half planes of a loaded checkpoint may be out of range (|w| >= 64 / 128): the forward raised the word - look
        # (validation reads its metrics on the host anyway), switch this model to the fp32 products and score again
        # ONCE per parameter version (the word is a function of the weights; a blocking .item() per validation batch put a stream
        # synchronisation between the forward and the NDCG launch - ADVICE r05)
        key = (params.data_ptr(), params._version)
        look = key != self._checked
        self._checked = key
        if look and hip_ops.split_half_enabled(self.shape) and int(self._flag.item()) != 0:
            hip_ops.fall_back_to_fp32_products(self.shape, "a SetRank weight of magnitude >= 64 was loaded (the split-half weight planes "
                                                           "cover |w| < 128)")
            hip_ops.setrank_forward(self.shape, params, features, n_docs, docids, self.B, self.L, self.scores, self.saved)
        self._ndcg(labels, docids, n_docs)
        return self.scores, self.ndcg
