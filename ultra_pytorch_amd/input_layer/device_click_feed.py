"""DeviceClickFeed — the MI355X-native input layer: dataset resident in HBM, clicks simulated on the device.

Replaces the per-step Python batch construction of ClickSimulationFeed (reference click_simulation_feed.py:101-174,
~15 ms at config 2) by one kernel launch (ultr_click_batch): features never move again after the one-time upload, a
batch is B x L global document ids + clicks.  Same distribution as the reference feed (uniform queries, position-biased or
cascade clicks, click-less lists redrawn, the bias severity eta drifting every `dynamic_bias_step_interval` batches when
`dynamic_bias_eta_change` is set - click_simulation_feed.py:165-172); not the same random stream.  `get_batch` returns a feed of DEVICE tensors that the
plugin algorithms recognise (`device_feed` key) and pass straight to the step kernels.

Round 6: the feed is double-buffered and a plugin algorithm's `train(feed)` draws the NEXT batch behind its own step in the same host
call (`next_click_args` -> ultr_feed_train_step): a batch is a pure function of (seed, batch counter), so drawing it one call early
changes nothing but when the kernel runs - under the step's reduction / update instead of in front of the next step.  `get_batch`
then only hands out the buffer that is already filled (and draws on the spot when nobody drew ahead, or when its arguments differ
from what was assumed).  The feed dictionaries are cached per buffer: no per-step ctypes / dict construction."""
import ctypes
import json
import os

import numpy as np
import torch

from .. import _lib
from .. import hip_ops
from ..utils import HParams
from ..utils import click_models


class ResidentDataset(object):
    """One-time upload of a Raw_data (after .pad(L)) to HBM."""

    def __init__(self, data_set, device):
        feats = np.asarray(data_set.features[:-1] if self._has_pad_row(data_set) else data_set.features, dtype=np.float32)
        self.n_docs = int(feats.shape[0])
        self.features = torch.from_numpy(np.ascontiguousarray(feats)).to(device)
        lmax = max(len(x) for x in data_set.initial_list)
        lists = np.full((len(data_set.initial_list), lmax), -1, dtype=np.int32)
        labels = np.zeros((len(data_set.initial_list), lmax), dtype=np.float32)
        for q, (lst, lab) in enumerate(zip(data_set.initial_list, data_set.labels)):
            lists[q, : len(lst)] = lst
            labels[q, : len(lab)] = lab
        lists[lists >= self.n_docs] = -1
        self.lists = torch.from_numpy(lists).to(device)
        self.labels = torch.from_numpy(labels).to(device)
        self.n_queries, self.lmax = int(lists.shape[0]), int(lmax)

    @staticmethod
    def _has_pad_row(data_set):
        return len(data_set.features) > 0 and not any(data_set.features[-1]) and len(data_set.features) > len(data_set.dids)


class DeviceClickFeed(object):
    def __init__(self, model, batch_size, hparam_str, seed=0):
        self.hparams = HParams(click_model_json="./example/ClickModel/pbm_0.1_1.0_4_1.0.json", max_tries=100,
                               dynamic_bias_eta_change=0.0, dynamic_bias_step_interval=1000)
        self.hparams.parse(hparam_str)
        path = self.hparams.click_model_json
        if not os.path.exists(path):
            alt = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", os.path.basename(path))
            path = alt if os.path.exists(alt) else path
        desc = json.load(open(path))
        if desc["model_name"] not in ("position_biased_model", "cascade_model", "user_browsing_model"):
            raise NotImplementedError("DeviceClickFeed simulates the position-biased, the cascade and the user-browsing model")
        self.click_model = click_models.loadModelFromJson(desc)  # host twin: owns eta and the examination table
        self.model_id = {"position_biased_model": 0, "cascade_model": 1, "user_browsing_model": 2}[desc["model_name"]]
        self.model, self.batch_size = model, int(batch_size)
        self.rank_list_size = model.rank_list_size
        self.device = model.cuda
        self.exam, self.n_exam = self._exam_tensor()
        self.global_batch_count = 0
        self.cprob = torch.tensor(desc["click_prob"], dtype=torch.float32, device=self.device)
        self.seed, self.step = int(seed), 0
        self.lib = _lib.load()
        self._resident = {}
        B, L = self.batch_size, self.rank_list_size
        self._bufs = [(torch.empty(L, B, dtype=torch.int32, device=self.device), torch.empty(L, B, dtype=torch.float32, device=self.device),
                       torch.empty(B, dtype=torch.int32, device=self.device)) for _ in range(2)]
        self._cur = 0          # the buffer of the batch handed out last
        self._pre = None       # (dataset key, max_tries) of a batch drawn AHEAD into the other buffer for counter value self.step
        self._last = None      # (dataset key, max_tries, resident dataset) of the last get_batch: what a draw ahead assumes
        self._cargs = [_lib.ClickArgs(), _lib.ClickArgs()]
        self._cptr = [ctypes.c_void_p(ctypes.addressof(c)) for c in self._cargs]
        self._ckey = [None, None]
        self._feeds = [None, None]
        self.docids, self.clicks, self.qidx = self._bufs[0]

    def _exam_tensor(self):
        """The examination table on the device: [n] for PBM / cascade, a dense [n][n] image of the triangular rank x distance table
        for the user-browsing model (ULTR_CLICK_UBM)."""
        ep = self.click_model.exam_prob
        if self.model_id == 2:
            n = len(ep)
            dense = [[(row[c] if c < len(row) else 0.0) for c in range(n)] for row in ep]
            return torch.tensor(dense, dtype=torch.float32, device=self.device).contiguous(), n
        return torch.tensor(ep, dtype=torch.float32, device=self.device), len(ep)

    @staticmethod
    def preprocess_data(data_set, hparam_str, exp_settings):
        return

    def resident(self, data_set):
        key = id(data_set)
        if key not in self._resident:
            self._resident[key] = ResidentDataset(data_set, self.device)
        return self._resident[key]

    def _fill(self, k, rd, max_tries):
        """Argument block k (= buffer k) for the batch with counter value self.step."""
        c = self._cargs[k]
        key = (id(rd), self.exam.data_ptr(), max_tries)
        if self._ckey[k] != key:
            docids, clicks, qidx = self._bufs[k]
            c.lists, c.labels, c.n_queries, c.lmax, c.n_docs = rd.lists.data_ptr(), rd.labels.data_ptr(), rd.n_queries, rd.lmax, rd.n_docs
            c.exam_prob, c.n_exam, c.click_prob, c.n_rel = self.exam.data_ptr(), self.n_exam, self.cprob.data_ptr(), int(self.cprob.numel())
            c.click_model, c.seed, c.batch, c.list_size, c.max_tries = self.model_id, self.seed, self.batch_size, self.rank_list_size, max_tries
            c.docids, c.clicks, c.query_idx = docids.data_ptr(), clicks.data_ptr(), qidx.data_ptr()
            self._ckey[k] = key
        c.step = self.step
        return self._cptr[k]

    def next_click_args(self):
        """Called by the step engine inside `train(feed)`: the argument block (ultr_click_args*) of the NEXT batch, to be drawn into the
        other buffer behind the step that is being queued - or None when no batch has been handed out yet."""
        if self._last is None or self._pre is not None:
            return None
        key, mt, rd = self._last
        self._pre = (key, mt)
        return self._fill(1 - self._cur, rd, mt)

    def get_batch(self, data_set, check_validation=True, data_format="ULTRA"):
        rd = self.resident(data_set)
        mt = int(self.hparams.max_tries) if check_validation else 1
        key = id(data_set)
        k = 1 - self._cur
        if self._pre != (key, mt):  # nobody drew ahead (or for other arguments: the draw is redone, stream order keeps it consistent)
            _lib.check(self.lib.ultr_click_batch_args(self._fill(k, rd, mt), hip_ops.raw_stream()),
                       "ultr_click_batch")
        self._pre, self._cur, self._last = None, k, (key, mt, rd)
        self.docids, self.clicks, self.qidx = self._bufs[k]
        self.step += 1
        self.global_batch_count += 1
        # drifting bias severity (click_simulation_feed.py:165-172): eta moves, the examination table is rebuilt from the model's
        # ORIGINAL table to the new power (click_models.py:77-81) and uploaded - 10 floats every `interval` batches
        if self.hparams.dynamic_bias_eta_change != 0 and self.global_batch_count % self.hparams.dynamic_bias_step_interval == 0:
            self.click_model.eta += self.hparams.dynamic_bias_eta_change
            self.click_model.setExamProb(self.click_model.eta)
            self.exam, self.n_exam = self._exam_tensor()
        cached = self._feeds[k]
        if cached is None or cached[0] != key:
            feed = {"device_feed": True, "features": rd.features, "n_docs": rd.n_docs, "docids": self.docids,
                    "labels": self.clicks, "batch_size": self.batch_size, "feed_obj": self}
            cached = self._feeds[k] = (key, feed, {"rank_list_idxs": self.qidx})
        return cached[1], cached[2]
