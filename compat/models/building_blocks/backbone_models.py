"""Reference name `models.building_blocks.backbone_models` (backbone_models.py:7-81): the `--backbone` registry."""
from vince_amd.models.building_blocks import backbone_models as _b
from vince_amd.models.building_blocks.backbone_models import *  # noqa: F401,F403

__all__ = list(_b.__all__)
