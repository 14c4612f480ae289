"""Stand-in for dg_util.python_utils.tensorboard_logger on boxes without dg_util (see compat/README.md)."""
from vince_amd.utils.jsonl_logger import Logger  # noqa: F401
