"""Reference name `constants` (constants.py:12-29) -> vince_amd.constants."""
from vince_amd.constants import *  # noqa: F401,F403
from vince_amd.constants import BASE_LOG_DIR, IMAGENET_MEAN, IMAGENET_STD, NONLINEARITY, TIME_STR  # noqa: F401
