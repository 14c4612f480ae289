"""Reference name `utils.storage_queue` (utils/storage_queue.py:4-56)."""
from vince_amd.utils.storage_queue import StorageQueue  # noqa: F401
