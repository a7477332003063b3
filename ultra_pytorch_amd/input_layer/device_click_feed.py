"""DeviceClickFeed — the MI355X-native input layer: dataset resident in HBM, clicks simulated on the device.

Replaces the per-step Python batch construction of ClickSimulationFeed (reference click_simulation_feed.py:101-174,
~15 ms at config 2) by one kernel launch (ultr_click_batch): features never move again after the one-time upload, a
batch is B x L global document ids + clicks.  Same distribution as the reference feed (uniform queries, position-biased or
cascade clicks, click-less lists redrawn, the bias severity eta drifting every `dynamic_bias_step_interval` batches when
`dynamic_bias_eta_change` is set - click_simulation_feed.py:165-172); not the same random stream.  `get_batch` returns a feed of DEVICE tensors that the
plugin algorithms recognise (`device_feed` key) and pass straight to the step kernels."""
import ctypes
import json
import os

import numpy as np
import torch

from .. import _lib
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
        self.docids = torch.empty(L, B, dtype=torch.int32, device=self.device)
        self.clicks = torch.empty(L, B, dtype=torch.float32, device=self.device)
        self.qidx = torch.empty(B, dtype=torch.int32, device=self.device)

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

    def get_batch(self, data_set, check_validation=True, data_format="ULTRA"):
        rd = self.resident(data_set)
        vp = lambda t: ctypes.c_void_p(t.data_ptr())
        rc = self.lib.ultr_click_batch(vp(rd.lists), vp(rd.labels), rd.n_queries, rd.lmax, rd.n_docs, vp(self.exam),
                                       self.n_exam, vp(self.cprob), int(self.cprob.numel()), self.model_id, self.seed, self.step,
                                       self.batch_size, self.rank_list_size, int(self.hparams.max_tries) if check_validation else 1,
                                       vp(self.docids), vp(self.clicks), vp(self.qidx),
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "ultr_click_batch")
        self.step += 1
        self.global_batch_count += 1
        # drifting bias severity (click_simulation_feed.py:165-172): eta moves, the examination table is rebuilt from the model's
        # ORIGINAL table to the new power (click_models.py:77-81) and uploaded - 10 floats every `interval` batches
        if self.hparams.dynamic_bias_eta_change != 0 and self.global_batch_count % self.hparams.dynamic_bias_step_interval == 0:
            self.click_model.eta += self.hparams.dynamic_bias_eta_change
            self.click_model.setExamProb(self.click_model.eta)
            self.exam, self.n_exam = self._exam_tensor()
        feed = {"device_feed": True, "features": rd.features, "n_docs": rd.n_docs, "docids": self.docids,
                "labels": self.clicks, "batch_size": self.batch_size}
        return feed, {"rank_list_idxs": self.qidx}
