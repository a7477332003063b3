"""DLA — the Dual Learning Algorithm: ranker and per-position propensity model learned jointly.
Drop-in for ultra.learning_algorithm.DLA (reference dla.py:51-306)."""
import math

import torch
import torch.nn as nn

from ..utils import HParams
from .base_algorithm import BaseAlgorithm


class DenoisingNet(nn.Module):
    """dla.py:24-48: Linear(L,1)+ELU on one-hot positions, i.e. propensity_l = ELU(W[0,l] + b).
    Parameters are views into one flat [L+1] tensor (W | b) that the loss / update kernels use directly."""

    def __init__(self, input_vec_size):
        super().__init__()
        self.list_size = int(input_vec_size)
        bound = 1.0 / math.sqrt(self.list_size)  # nn.Linear default init
        self._bind((torch.rand(self.list_size + 1) * 2 - 1) * bound)

    def _bind(self, flat):
        L = self.list_size
        self.flat_params = flat
        self.linear_layer = nn.Module()
        self.linear_layer.register_parameter("weight", nn.Parameter(flat[:L].view(1, L), requires_grad=False))
        self.linear_layer.register_parameter("bias", nn.Parameter(flat[L:L + 1], requires_grad=False))

    def _apply(self, fn, *a, **k):
        self._bind(fn(self.flat_params.detach()).contiguous().to(torch.float32))
        return self


class DLA(BaseAlgorithm):
    ENGINE_ALGO = "dla"

    def __init__(self, data_set, exp_settings):
        print("Build DLA")
        self.hparams = HParams(learning_rate=0.05, max_gradient_norm=5.0, loss_func="softmax_loss",
                               logits_to_prob="softmax", propensity_learning_rate=-1.0, ranker_loss_weight=1.0,
                               l2_loss=0.0, max_propensity_weight=-1, constant_propensity_initialization=False,
                               grad_strategy="ada")
        print(exp_settings["learning_algorithm_hparams"])
        self.hparams.parse(exp_settings["learning_algorithm_hparams"])
        self._check_hparams()
        self._setup(data_set, exp_settings)
        self.propensity_model = DenoisingNet(self.rank_list_size).to(self.cuda)
        plr = float(self.hparams.propensity_learning_rate)
        self.propensity_learning_rate = self.learning_rate if plr < 0 else plr
        print("Loss Function is " + self.hparams.loss_func)

    def _engine_kwargs(self):
        return dict(ranker_loss_weight=float(self.hparams.ranker_loss_weight),
                    propensity_learning_rate=self.propensity_learning_rate,
                    logits_to_prob=self.hparams.logits_to_prob)

    def train(self, input_feed):
        """dla.py:179-266.  Both models get a FRESH optimizer each step in the reference (dla.py:153-154), so the
        Adagrad accumulator never persists: state=None selects the stateless update.  max_propensity_weight is
        inert in the reference (clamps .grad of a grad-less tensor, dla.py:303-305) and is ignored here too."""
        self.rank_list_size = self.exp_settings["selection_bias_cutoff"]
        if not self.model.training:  # (nn.Module.train() walks every submodule: ~10 us a 47 us step does not have)
            self.model.train()
        self.create_input_feed(input_feed, self.rank_list_size)
        eng = self._train_engine(self.batch_size, self.rank_list_size)
        sc = eng.train_step(self.model.flat_params, None, self.letor_features, self.n_docs, self.docid_inputs,
                            self.labels_LB, aux=self.propensity_model.flat_params)
        vals = eng.read_scalars()
        self.loss, self.rank_loss, self.exam_loss = float(vals[0]), float(vals[4]), float(vals[5])
        print(" Loss %f at Global Step %d: " % (self.loss, self.global_step))
        self.global_step += 1
        return self.loss, None, self.train_summary
