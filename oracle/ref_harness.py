"""Build-container-only harness that imports the *reference* (``/root/reference``) on CPU.

TEST INFRASTRUCTURE.  Used solely by ``oracle/make_golden.py`` to produce ``tests/golden/*.npz``.
It never travels to the GPU box in a usable form (``/root/reference`` does not exist there) and
no test, smoke or bench path imports it.

The reference depends on three things that are not installed here and cannot be installed
(no network): ``dg_util`` (unpinned ``git+https://github.com/danielgordon10/dg_util.git``,
reference ``requirements.txt:21``), ``torchvision==0.5.0`` (``requirements.txt:18``) and ``cv2``.
On the hot path they contribute no arithmetic:

* ``torchvision.models.resnet18/50`` -> satisfied by the reference's own vendored copy
  ``models/building_blocks/resnet.py`` (identical architecture and module names).
* ``dg_util.python_utils.pytorch_util`` -> reshape helpers and an ``nn.Module`` base class whose
  semantics are forced by the reference's call sites; restated below.
* ``cv2`` / ``efficientnet_pytorch`` / ``dg_util.drawing`` -> only names are needed at import time.
"""
import os
import sys
import types

import numpy as np
import torch
from torch import nn

REFERENCE_ROOT = os.environ.get("VINCE_REFERENCE_ROOT", "/root/reference")


def _split_dim(x, dim, d1, d2):
    # pt_util.split_dim: reshape axis `dim` into (d1, d2); -1 allowed (vince_model.py:148,164).
    shape = list(x.shape)
    new_shape = shape[:dim] + [d1, d2] + shape[dim + 1:]
    return x.reshape(new_shape)


def _remove_dim(x, dim):
    # pt_util.remove_dim: merge axis `dim` into axis dim-1; tuple -> highest first (vince_model.py:153-155).
    if isinstance(dim, (tuple, list)):
        for d in sorted(dim, reverse=True):
            x = _remove_dim(x, d)
        return x
    shape = list(x.shape)
    new_shape = shape[: dim - 1] + [shape[dim - 1] * shape[dim]] + shape[dim + 1:]
    return x.reshape(new_shape)


def _expand_new_dim(x, dim, n):
    # pt_util.expand_new_dim: unsqueeze(dim) then expand to n along it (vince_model.py:168).
    x = x.unsqueeze(dim)
    shape = [-1] * x.dim()
    shape[dim] = n
    return x.expand(*shape)


class _RemoveDim(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        return _remove_dim(x, self.dim)


class _BaseModel(nn.Module):
    # pt_util.BaseModel: nn.Module with a .device property and a .saves counter (base_model.py:24-26).
    def __init__(self):
        super().__init__()
        self.saves = 0

    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")

    def restore(self, *a, **k):
        return 0


def _get_data_parallel(module, gpu_ids):
    # pt_util.get_data_parallel: nn.DataParallel for >1 id, else the module (vince_model.py:35).
    return module


def install():
    """Insert the stand-in modules and put the reference on sys.path.  Idempotent."""
    if "dg_util" in sys.modules:
        return
    if not hasattr(np, "bool"):
        np.bool = bool  # vince_model.py:54 uses the removed alias np.bool

    dg = types.ModuleType("dg_util")
    pu = types.ModuleType("dg_util.python_utils")
    pt = types.ModuleType("dg_util.python_utils.pytorch_util")
    pt.split_dim = _split_dim
    pt.remove_dim = _remove_dim
    pt.expand_new_dim = _expand_new_dim
    pt.RemoveDim = _RemoveDim
    pt.BaseModel = _BaseModel
    pt.get_data_parallel = _get_data_parallel
    pt.from_numpy = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    pt.to_numpy = lambda t: t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    pt.save = lambda *a, **k: None
    pt.AttentionPool2D = None
    misc = types.ModuleType("dg_util.python_utils.misc_util")
    misc.get_time_str = lambda: "golden"
    drawing = types.ModuleType("dg_util.python_utils.drawing")
    pu.pytorch_util = pt
    pu.misc_util = misc
    pu.drawing = drawing
    dg.python_utils = pu
    sys.modules.update({
        "dg_util": dg,
        "dg_util.python_utils": pu,
        "dg_util.python_utils.pytorch_util": pt,
        "dg_util.python_utils.misc_util": misc,
        "dg_util.python_utils.drawing": drawing,
    })
    cv2 = types.ModuleType("cv2")
    cv2.INTER_LINEAR = 1
    sys.modules["cv2"] = cv2
    eff = types.ModuleType("efficientnet_pytorch")
    eff.EfficientNet = object
    sys.modules["efficientnet_pytorch"] = eff

    sys.path.insert(0, REFERENCE_ROOT)
    # torchvision.models -> the reference's vendored resnet.py
    import importlib
    my_resnet = importlib.import_module("models.building_blocks.resnet")
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvm.resnet18 = my_resnet.resnet18
    tvm.resnet34 = my_resnet.resnet34
    tvm.resnet50 = my_resnet.resnet50
    tv.models = tvm
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm


def load_reference():
    """Returns the reference modules used on the path."""
    install()
    import importlib
    mods = types.SimpleNamespace()
    mods.vince_model = importlib.import_module("models.vince_model")
    mods.loss_util = importlib.import_module("utils.loss_util")
    mods.storage_queue = importlib.import_module("utils.storage_queue")
    mods.backbone_models = importlib.import_module("models.building_blocks.backbone_models")
    return mods


def make_args(**kw):
    """An argparse-like namespace with the attributes the path reads (SURVEY.md 8b config contract)."""
    install()
    import importlib
    bb = importlib.import_module("models.building_blocks.backbone_models")
    d = dict(
        num_frames=1, backbone="ResNet18", use_attention=False, feature_extractor_gpu_ids=["cpu"],
        pytorch_gpu_ids=["cpu"], vince_embedding_size=64, jigsaw=False, inter_batch_comparison=False,
        self_batch_comparison=False, batch_size=32, vince_queue_size=512, use_imagenet=False,
        use_imagenet_weights=False, vince_temperature=0.07, vince_self_temperature=0.03,
        vince_momentum=0.999, restore=False, save=False, base_lr=0.03,
    )
    d.update(kw)
    if isinstance(d["backbone"], str):
        d["backbone"] = getattr(bb, d["backbone"])
    return types.SimpleNamespace(**d)
