"""PairDebias — Unbiased LambdaMART-style pairwise debiasing with EM-estimated position ratios t+ / t-.
Drop-in for ultra.learning_algorithm.PairDebias (reference pairwise_debias.py:27-202)."""
import torch

from ..utils import HParams
from .base_algorithm import BaseAlgorithm


class PairDebias(BaseAlgorithm):
    ENGINE_ALGO = "pairdebias"

    def __init__(self, data_set, exp_settings):
        print("Build Pairwise Debiasing algorithm.")
        self.hparams = HParams(EM_step_size=0.05, learning_rate=0.005, max_gradient_norm=5.0, regulation_p=1,
                               l2_loss=0.0, grad_strategy="ada")
        print(exp_settings["learning_algorithm_hparams"])
        self.hparams.parse(exp_settings["learning_algorithm_hparams"])
        self._check_hparams()
        self._setup(data_set, exp_settings)
        L = self.rank_list_size
        self.t_state = torch.ones(2 * L, dtype=torch.float32, device=self.cuda)  # [t_plus | t_minus], init 1 (:91-99)

    @property
    def t_plus(self):
        return self.t_state[: self.rank_list_size].view(1, -1)

    @property
    def t_minus(self):
        return self.t_state[self.rank_list_size:].view(1, -1)

    def _engine_kwargs(self):
        return dict(em_step_size=float(self.hparams.EM_step_size), regulation_p=float(self.hparams.regulation_p))

    def train(self, input_feed):
        """pairwise_debias.py:106-174: all ordered pairs (i, j), i != j, masked by click_i > click_j, the xB
        broadcast inflation included (Appendix A.6); EM update of t+/t- with the PRE-update values in the sums."""
        if not self.model.training:  # (nn.Module.train() walks every submodule: ~10 us a 47 us step does not have)
            self.model.train()
        self.create_input_feed(input_feed, self.rank_list_size)
        eng = self._train_engine(self.batch_size, self.rank_list_size)
        sc = eng.train_step(self.model.flat_params, self.state_sum, self.letor_features, self.n_docs, self.docid_inputs,
                            self.labels_LB, aux=self.t_state)
        self.loss = eng.read_loss()  # the reference's only host sync: loss.item()
        print(" Loss %f at Global Step %d" % (self.loss, self.global_step))
        self.global_step += 1
        return self.loss, None, self.train_summary
