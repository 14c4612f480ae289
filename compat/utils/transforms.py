"""Reference name `utils.transforms` (utils/transforms.py:62-235): the `--transform` registry, GPU batch recipes."""
from vince_amd.utils import transforms as _t
from vince_amd.utils.transforms import *  # noqa: F401,F403

__all__ = list(getattr(_t, "__all__", [n for n in vars(_t) if n.endswith("Transform")]))
