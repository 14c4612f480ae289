"""vince_amd -- MI355X-native encoder + contrastive hot path of VINCE (danielgordon10/vince).

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); every hot op is a hand-written
HIP kernel in ``vince_amd/csrc`` exposed through the flat C ABI of ``include/vince_hip.h`` and bound with ctypes.
There is no CPU fallback: without ``libvince_hip.so`` or without a GPU the compute entry points raise.

The module layout mirrors the reference's import paths for the path:
``models.vince_model`` (VinceModel, VinceQueueModel), ``models.building_blocks.backbone_models`` (ResNet18, ResNet50),
``utils.loss_util`` (similarity_cross_entropy), ``utils.storage_queue`` (StorageQueue), ``solvers.vince_solver``
(VinceSolver), ``solvers.base_solver`` (BaseSolver), ``solver_runner`` (main), ``arg_parser`` (parse_args).
"""
__version__ = "0.1.0"
