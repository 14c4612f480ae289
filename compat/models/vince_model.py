"""Reference name `models.vince_model` (models/vince_model.py:19-613)."""
from vince_amd.models.vince_model import VinceModel, VinceQueueModel  # noqa: F401
