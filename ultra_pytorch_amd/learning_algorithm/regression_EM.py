"""RegressionEM — regression-based EM for position-bias estimation (Wang et al., WSDM 2018).
Drop-in for ultra.learning_algorithm.RegressionEM (reference regression_EM.py:36-222)."""
import torch

from ..utils import HParams
from .base_algorithm import BaseAlgorithm


class RegressionEM(BaseAlgorithm):
    ENGINE_ALGO = "regem"

    def __init__(self, data_set, exp_settings):
        print("Build Regression-based EM algorithm.")
        self.hparams = HParams(EM_step_size=0.05, learning_rate=0.05, max_gradient_norm=5.0, l2_loss=0.0,
                               grad_strategy="ada")
        print(exp_settings["learning_algorithm_hparams"])
        self.hparams.parse(exp_settings["learning_algorithm_hparams"])
        self._check_hparams()
        self._setup(data_set, exp_settings)
        # examination propensity per position, initialised to 0.9 (regression_EM.py:97-100); sigmoid_prob_b is a
        # constant zero in the reference (:102-104: a plain tensor, never an optimizer parameter) and is dropped
        self.propensity_state = torch.full((self.rank_list_size,), 0.9, dtype=torch.float32, device=self.cuda)
        self.uniforms = None  # tests inject the Bernoulli draw here ([B, L] device tensor); None = device Philox stream

    @property
    def propensity(self):
        return self.propensity_state.view(1, -1)

    @property
    def propensity_weights(self):
        return 1.0 / self.propensity  # regression_EM.py:185

    def _engine_kwargs(self):
        return dict(em_step_size=float(self.hparams.EM_step_size))

    def train(self, input_feed):
        """regression_EM.py:108-193: E-step posteriors from the current scores, Bernoulli pseudo-labels, BCE-with-logits
        (mean over B*L), clip + Adagrad, M-step on the propensity with the pre-update scores; global_step is
        incremented AFTER the step (:188)."""
        if not self.model.training:  # (nn.Module.train() walks every submodule: ~10 us a 47 us step does not have)
            self.model.train()
        self.create_input_feed(input_feed, self.rank_list_size)
        eng = self._train_engine(self.batch_size, self.rank_list_size)
        sc = eng.train_step(self.model.flat_params, self.state_sum, self.letor_features, self.n_docs, self.docid_inputs,
                            self.labels_LB, aux=self.propensity_state, uniforms=self.uniforms)
        self.loss = eng.read_loss()  # the reference's only host sync: loss.item()
        self.update_propensity_op = self.propensity
        self.global_step += 1
        print("Loss %f at global step %d" % (self.loss, self.global_step))
        return self.loss, None, self.train_summary
