"""Ranking models with the reference's plugin contract (ultra.ranking_model.*), HIP-backed."""
from .dnn import DNN, Linear, init_flat_params  # noqa: F401
from . import SetRank  # noqa: F401,E402  module, as in the reference: the class is ultra.ranking_model.SetRank.SetRank
