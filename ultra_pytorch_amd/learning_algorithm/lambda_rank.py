"""LambdaRank — delta-NDCG-weighted pairwise loss on the sorted list with t+/t- EM debiasing.
Drop-in for ultra.learning_algorithm.LambdaRank (reference lambda_rank.py:21-291)."""
import torch

from ..utils import HParams
from .base_algorithm import BaseAlgorithm


class LambdaRank(BaseAlgorithm):
    ENGINE_ALGO = "lambdarank"

    def __init__(self, data_set, exp_settings):
        self.hparams = HParams(EM_step_size=0.05, learning_rate=0.05, max_gradient_norm=5.0, grad_strategy="ada",
                               regulation_p=1, sigma=1.0)
        print(exp_settings["learning_algorithm_hparams"])
        self.hparams.parse(exp_settings["learning_algorithm_hparams"])
        self._check_hparams()
        self._setup(data_set, exp_settings)
        self.sigma = float(self.hparams.sigma)
        L = self.rank_list_size
        self.t_state = torch.ones(2 * L, dtype=torch.float32, device=self.cuda)

    @property
    def t_plus(self):
        return self.t_state[: self.rank_list_size].view(1, -1)

    @property
    def t_minus(self):
        return self.t_state[self.rank_list_size:].view(1, -1)

    def _engine_kwargs(self):
        return dict(em_step_size=float(self.hparams.EM_step_size), regulation_p=float(self.hparams.regulation_p),
                    sigma=float(self.hparams.sigma))

    def train(self, input_feed):
        """lambda_rank.py:96-216 incl. its quirks: BCE-with-logits applied to a probability, batch-global
        natural-log IDCG, all L^2 pairs incl. the (zero-weight) diagonal (Appendix A.7)."""
        self.rank_list_size = self.exp_settings["selection_bias_cutoff"]
        self.global_step += 1
        if not self.model.training:  # (nn.Module.train() walks every submodule: ~10 us a 47 us step does not have)
            self.model.train()
        self.create_input_feed(input_feed, self.rank_list_size)
        eng = self._train_engine(self.batch_size, self.rank_list_size)
        sc = eng.train_step(self.model.flat_params, self.state_sum, self.letor_features, self.n_docs, self.docid_inputs,
                            self.labels_LB, aux=self.t_state)
        self.loss = eng.read_loss()  # the reference's only host sync: loss.item()
        print(" Loss %f at Global Step %d: " % (self.loss, self.global_step))
        return self.loss, None, self.train_summary
