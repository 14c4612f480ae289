"""Reference name `utils.loss_util` (utils/loss_util.py:7-62)."""
from vince_amd.utils.loss_util import similarity_cross_entropy  # noqa: F401
