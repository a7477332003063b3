"""Host-side utilities mirroring ultra.utils (plugin seam, hyper-parameter strings, metrics)."""
from .hparams import HParams  # noqa: F401
from .sys_tools import find_class, create_object  # noqa: F401
from .data_utils import Raw_data, read_data, merge_Summary, output_ranklist  # noqa: F401
from .metrics import make_ranking_metric_fn, RankingMetricKey  # noqa: F401
from . import click_models, propensity_estimator, data_utils, metrics, hparams, sys_tools  # noqa: F401
