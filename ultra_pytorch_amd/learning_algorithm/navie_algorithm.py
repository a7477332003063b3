"""NavieAlgorithm (sic) — click data used as relevance labels, listwise softmax cross entropy.
Drop-in for ultra.learning_algorithm.NavieAlgorithm (reference navie_algorithm.py:26-149)."""
from ..utils import HParams
from .base_algorithm import BaseAlgorithm


class NavieAlgorithm(BaseAlgorithm):
    ENGINE_ALGO = "softmax"

    def __init__(self, data_set, exp_settings):
        print("Build NavieAlgorithm")
        self.hparams = HParams(learning_rate=0.05, max_gradient_norm=5.0, loss_func="softmax_cross_entropy",
                               l2_loss=0.0, grad_strategy="ada")
        print(exp_settings["learning_algorithm_hparams"])
        self.hparams.parse(exp_settings["learning_algorithm_hparams"])
        self._check_hparams()
        self._setup(data_set, exp_settings)

    def train(self, input_feed):
        """navie_algorithm.py:76-120: forward, softmax_loss(labels), backward, clip, Adagrad."""
        self.global_step += 1
        if not self.model.training:  # (nn.Module.train() walks every submodule: ~10 us a 47 us step does not have)
            self.model.train()
        self.create_input_feed(input_feed, self.rank_list_size)
        eng = self._train_engine(self.batch_size, self.rank_list_size)
        sc = eng.train_step(self.model.flat_params, self.state_sum, self.letor_features, self.n_docs, self.docid_inputs,
                            self.labels_LB)
        self.loss = eng.read_loss()  # the reference's only host sync: loss.item()
        print(" Loss %f at Global Step %d: " % (self.loss, self.global_step))
        return self.loss, None, self.train_summary
