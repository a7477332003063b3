"""Learning algorithms with the reference's plugin contract (ultra.learning_algorithm.*), HIP-backed.
Classes are re-exported at package level because the plugin seam resolves "pkg.Class" with
getattr(sys.modules[pkg], Class) (reference sys_tools.py:7-22, ultra/learning_algorithm/__init__.py:2-13)."""
from .base_algorithm import BaseAlgorithm  # noqa: F401
from .navie_algorithm import NavieAlgorithm  # noqa: F401
from .ipw_rank import IPWrank  # noqa: F401
from .dla import DLA, DenoisingNet  # noqa: F401
from .pairwise_debias import PairDebias  # noqa: F401
from .lambda_rank import LambdaRank  # noqa: F401
from .regression_EM import RegressionEM  # noqa: F401
