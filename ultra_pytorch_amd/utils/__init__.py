"""Host-side utilities mirroring ultra.utils (plugin seam, hyper-parameter strings, metrics)."""
from .hparams import HParams  # noqa: F401
from .sys_tools import find_class, create_object  # noqa: F401
