"""BaseAlgorithm — the shared half of the `ultra.learning_algorithm` plugin contract, HIP-backed.

Mirrors what main.py and the input feeds call / read on a learning-algorithm object:
  ctor (data_set, exp_settings); train(input_feed) -> (float loss, None, dict);
  validation(input_feed, is_online_simulation=False) -> (None, Tensor[B, max_candidate_num], dict);
  attributes model, global_step, learning_rate, rank_list_size, max_candidate_num, feature_size,
  letor_features_name, docid_inputs_name[], labels_name[], hparams.
Reference: ultra/learning_algorithm/base_algorithm.py:32-333.

The hot path — gather + DNN forward, the loss, DNN backward, clip, optimizer, NDCG — runs in libultr_hip.so
(engine.StepEngine / EvalEngine).  This file only marshals the feed (host numpy -> pinned -> HBM) and keeps the
reference's bookkeeping.  There is no CPU path: constructing an algorithm without a GPU raises.
"""
import collections

import numpy as np
import torch

from .. import engine
from ..utils import find_class
from ..utils import metrics as metrics_mod


class BaseAlgorithm(object):
    PADDING_SCORE = -100000  # base_algorithm.py:36
    ENGINE_ALGO = "softmax"

    # ---- construction ------------------------------------------------------------------------------
    def _setup(self, data_set, exp_settings):
        if not torch.cuda.is_available():
            raise RuntimeError("ultra_pytorch_amd.learning_algorithm needs an MI355X/ROCm GPU (no CPU fallback)")
        self.cuda = torch.device("cuda", torch.cuda.current_device())
        self.is_cuda_avail = True
        self.train_summary, self.eval_summary = {}, {}
        self.exp_settings = exp_settings
        if "selection_bias_cutoff" in exp_settings:
            self.rank_list_size = int(exp_settings["selection_bias_cutoff"])
        self.max_candidate_num = int(exp_settings["max_candidate_num"])
        self.feature_size = int(data_set.feature_size)
        self.letor_features_name = "letor_features"
        self.letor_features = None
        self.docid_inputs_name = ["docid_input{0}".format(i) for i in range(self.max_candidate_num)]
        self.labels_name = ["label{0}".format(i) for i in range(self.max_candidate_num)]
        self.docid_inputs, self.labels = [], []
        self.global_step = 0
        self.model = self.create_model(self.feature_size).to(self.cuda)
        self.learning_rate = float(self.hparams.learning_rate)
        self.state_sum = torch.zeros_like(self.model.flat_params)  # Adagrad accumulator (initial value 0)
        # per-(batch, list size) workspaces, least recently used first: DirectLabelFeed may return short batches (SURVEY Appendix
        # A.13), each distinct size is a workspace set - at most MAX_ENGINES of them are kept
        self._train_engines, self._eval_engines = collections.OrderedDict(), collections.OrderedDict()
        self._stage, self._stage_events = {}, {}
        self.process_group = exp_settings.get("process_group", None)  # data parallel: one rank per GPU
        self._dp_comm, self._dp_comm_tried = None, False

    def create_model(self, feature_size):
        """base_algorithm.py:156-167: the ranking model is a plugin too."""
        model = find_class(self.exp_settings["ranking_model"])(self.exp_settings["ranking_model_hparams"], feature_size)
        if not hasattr(model, "flat_params") or not hasattr(model, "shape"):
            raise TypeError("ranking_model %r is not a HIP-backed model; use ultra_pytorch_amd.ranking_model.DNN / .Linear"
                            % self.exp_settings["ranking_model"])
        return model

    # ---- feed marshalling (a1: base_algorithm.py:169-186) -------------------------------------------
    def _staging(self, n_words):
        """ONE pinned host buffer + ONE device buffer of 4-byte words for a whole batch: [features f32 | docids i32 | labels f32].
        A batch then costs one host-to-device copy (three separate tensors cost three launches and three events: ~25 us of
        the ~190 us a host-fed step took)."""
        st = self._stage.get("batch")
        if st is None or st[0].numel() < n_words:
            cap = max(int(n_words * 1.25), 1024)
            host = torch.empty(cap, dtype=torch.float32).pin_memory()
            dev = torch.empty(cap, dtype=torch.float32, device=self.cuda)
            st = self._stage["batch"] = (host, dev, host.numpy(), host.numpy().view(np.int32))
        return st

    def create_input_feed(self, input_feed, list_size):
        """numpy feed -> device tensors: features [n_docs,F] f32, docids [L,B] i32, labels [L,B] f32.
        A feed from input_layer.DeviceClickFeed already holds device tensors (resident dataset): nothing to move.
        LIFETIME: self.letor_features / docid_inputs / labels_LB are VIEWS of one reused device staging buffer and the returned
        host labels a view of the reused pinned buffer - valid until the next create_input_feed() of this object (train() and
        validation() consume them before they return; the reference allocates fresh tensors per batch, base_algorithm.py:169-186,
        and nothing of it keeps them either).  Whoever wants to keep a batch must .clone() it."""
        if input_feed.get("device_feed", False):
            self.n_docs, self.batch_size = input_feed["n_docs"], input_feed["batch_size"]
            self.letor_features = input_feed["features"]
            ids, lab = input_feed["docids"], input_feed["labels"]
            whole = ids.shape[0] == list_size  # (a slice is a new tensor object: ~2 us each that a 47 us step does not have)
            self.docid_inputs = ids if whole else ids[:list_size]
            self.labels_LB = lab if whole else lab[:list_size]
            self._feed_obj = input_feed.get("feed_obj")
            return None
        self._feed_obj = None
        feats = np.asarray(input_feed[self.letor_features_name])
        if feats.ndim != 2:
            feats = feats.reshape(0, self.feature_size)
        self.n_docs = int(feats.shape[0])
        F = self.feature_size
        B = int(len(input_feed[self.docid_inputs_name[0]]))
        self.batch_size = B
        nf, nid = self.n_docs * F, list_size * B
        host, dev, hf, hi = self._staging(nf + 2 * nid)
        ev = self._stage_events.get("batch")
        if ev is not None:
            ev.synchronize()  # the previous asynchronous copy out of the staging buffer must have read it
        # casts (f64 features -> f32, f32 ids -> i32) straight into the pinned buffer through its numpy views: torch's
        # cross-dtype copy_ is ~40x slower than numpy's for these sizes (5.4 ms vs 0.14 ms for 2560 x 136 f64 -> f32)
        if nf:
            np.copyto(hf[:nf].reshape(self.n_docs, F), feats, casting="unsafe")
        ids_h = hi[nf:nf + nid].reshape(list_size, B)
        lab_h = hf[nf + nid:nf + 2 * nid].reshape(list_size, B)
        for l in range(list_size):
            np.copyto(ids_h[l], input_feed[self.docid_inputs_name[l]], casting="unsafe")
            np.copyto(lab_h[l], input_feed[self.labels_name[l]], casting="unsafe")
        n = nf + 2 * nid
        dev[:n].copy_(host[:n], non_blocking=True)
        if ev is None:
            ev = self._stage_events["batch"] = torch.cuda.Event()
        ev.record()
        self.letor_features = dev[:nf].view(self.n_docs, F) if nf else None
        self.docid_inputs = dev[nf:nf + nid].view(torch.int32).view(list_size, B)
        self.labels_LB = dev[nf + nid:n].view(list_size, B)
        return lab_h  # [L, B] host view of this batch's labels (valid until the next batch is staged)

    # ---- engines -------------------------------------------------------------------------------------
    def _engine_kwargs(self):
        return {}

    MAX_ENGINES = 8

    def _dp_batch_total(self, B):
        """Data parallel: the GLOBAL batch of this step (shards may be uneven and may change from step to step).  Only
        PairDebias' xB factor consumes it (base_algorithm.py:242-248), so only PairDebias pays for the agreement - one
        integer all-reduce per step, entered by EVERY rank on EVERY step (never from inside a lazily built engine, where one
        rank could enter a collective its peers do not)."""
        if self.process_group is None:
            return B
        import torch.distributed as dist
        if self.ENGINE_ALGO != "pairdebias":
            return B * dist.get_world_size(self.process_group)
        cpu_pg = dist.get_backend(self.process_group) == "gloo"
        t = torch.tensor([B], dtype=torch.int64, device="cpu" if cpu_pg else self.cuda)
        dist.all_reduce(t, group=self.process_group)
        return int(t.item())

    def _train_engine(self, B, L):
        bt = self._dp_batch_total(B)
        if self.process_group is not None and not self._dp_comm_tried:
            # ONE gradient-exchange communicator per algorithm object, created at the first step (all ranks are here
            # together) and shared by every engine built later - engine construction itself runs no collective
            from .. import parallel
            from ..hip_ops import tail_floats
            self._dp_comm_tried = True
            # sized for the LONGEST list this object can see (the step tail is 4 + 2 L floats): a later engine with a longer
            # list - validation-shaped training batches, max_candidate_num > selection_bias_cutoff - must fit the same buffer
            lmax = max(int(L), int(getattr(self, "max_candidate_num", L) or L), int(getattr(self, "rank_list_size", L) or L))
            self._dp_comm = parallel.PeerComm.create(self.process_group, self.model.shape.n_params + tail_floats(lmax), self.cuda)
        key = (B, L)
        eng = self._train_engines.get(key)
        if eng is None:
            kw = dict(learning_rate=self.learning_rate, max_gradient_norm=float(self.hparams.max_gradient_norm),
                      optimizer="sgd" if self.hparams.grad_strategy == "sgd" else "ada",
                      l2_loss=float(getattr(self.hparams, "l2_loss", 0.0)), process_group=self.process_group)
            if self.process_group is not None:
                kw.update(batch_total=bt, comm=self._dp_comm)
                if self._dp_comm is None:
                    kw.update(no_peer_comm=True)
            kw.update(self._engine_kwargs())
            cls = getattr(self.model, "step_engine_cls", engine.StepEngine)  # the ranking model picks its engine
            eng = self._train_engines[key] = cls(self.model.shape, B, L, self.cuda, algo=self.ENGINE_ALGO, **kw)
            while len(self._train_engines) > self.MAX_ENGINES:
                _, old = self._train_engines.popitem(last=False)
                old.close()
        else:
            self._train_engines.move_to_end(key)
        eng.batch_total = bt
        # a DeviceClickFeed batch: the engine draws the feed's NEXT batch behind this step in the same host call
        eng.next_click_source = getattr(self, "_feed_obj", None)
        return eng

    def _check_hparams(self):
        lf = getattr(self.hparams, "loss_func", "softmax_loss")
        if lf in ("sigmoid_loss", "pairwise_loss"):
            raise NotImplementedError("loss_func=%r raises in the reference too (base_algorithm.py:267,307)" % lf)

    # ---- validation (a12/a13: e.g. ipw_rank.py:184-211) ----------------------------------------------
    def validation(self, input_feed, is_online_simulation=False):
        if self.model.training:
            self.model.eval()
        L = self.max_candidate_num
        labels_host = self.create_input_feed(input_feed, L)
        B = self.batch_size
        topn = [int(t) for t in self.exp_settings["metrics_topn"]]
        key = (B, L, tuple(topn))
        if key not in self._eval_engines:
            cls = getattr(self.model, "eval_engine_cls", engine.EvalEngine)
            self._eval_engines[key] = cls(self.model.shape, B, L, self.cuda, topn=topn)
            while len(self._eval_engines) > self.MAX_ENGINES:
                self._eval_engines.popitem(last=False)
        else:
            self._eval_engines.move_to_end(key)
        ev = self._eval_engines[key]
        scores, ndcg = ev.run(self.model.flat_params, self.letor_features, self.n_docs, self.docid_inputs, self.labels_LB)
        self.output = scores.clone()  # the UNMASKED scores are what callers get (base_algorithm.py / main.py:266)
        if not is_online_simulation:
            masked = None
            for metric in self.exp_settings["metrics"]:
                if metric == metrics_mod.RankingMetricKey.NDCG:
                    values = ev.read_ndcg()  # host-mapped report of the NDCG launch (no stream synchronisation, no copy)
                else:
                    if masked is None:
                        masked = ev.masked.cpu()
                        if labels_host is None:  # device feed: the labels live in HBM
                            lab_BL = self.labels_LB.t().contiguous().cpu()
                        else:
                            lab_BL = torch.from_numpy(np.ascontiguousarray(labels_host.T.astype(np.float32)))
                    values = metrics_mod.make_ranking_metric_fn(metric, topn)(lab_BL, masked, None)
                for n, v in zip(topn, values):
                    self.eval_summary["%s_%d" % (metric, n)] = float(v)
        return None, self.output, self.eval_summary
