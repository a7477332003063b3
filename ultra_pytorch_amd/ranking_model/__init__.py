"""Ranking models with the reference's plugin contract (ultra.ranking_model.*), HIP-backed."""
from .dnn import DNN, Linear, init_flat_params  # noqa: F401
