"""Reference name `models.base_model` (models/base_model.py:8-26)."""
from vince_amd.models.base_model import BaseModel  # noqa: F401
