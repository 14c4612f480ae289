"""ResNet-18 / ResNet-50 frame-encoder backbones (reference models/building_blocks/backbone_models.py:21-75).

Same constructor contract as the reference -- ``args.backbone(args, final_layer)`` -- and the same parameter names
(``model.<torchvision resnet names>``) so reference checkpoints load, but the module tree only HOLDS parameters: the
arithmetic is the hand-written HIP trunk engine (vince_amd/csrc/trunk.hip), driven by the owning VinceModel.
Only ``final_layer == -2`` (conv1 .. layer4, what VinceModel uses, vince_model.py:26) is wired to the engine.
"""
import weakref

import torch
from torch import nn

from ...engine import TrunkPlan

__all__ = ["ResNet18", "ResNet50"]


class _Node(nn.Module):
    """Bare container so that parameters get torchvision's dotted names."""


class ResNetParams(nn.Module):
    """Parameters and BatchNorm buffers of a torchvision ResNet, named exactly as torchvision names them."""

    def __init__(self, arch):
        super().__init__()
        self.plan = TrunkPlan(arch)
        self.trunk_params = []    # nn.Parameter list in engine order
        self.bn_buffers = []      # (running_mean, running_var, num_batches_tracked) in engine BN order
        for name, kind, shape, _ in self.plan.params:
            node, leaf = self._descend(name)
            p = nn.Parameter(torch.empty(shape))
            if kind == 0:
                nn.init.kaiming_normal_(p, mode="fan_out", nonlinearity="relu")   # resnet.py:181-183
            elif kind == 1:
                nn.init.constant_(p, 1.0)                                           # resnet.py:184-186
            else:
                nn.init.constant_(p, 0.0)
            node.register_parameter(leaf, p)
            self.trunk_params.append(p)
        for name, c in self.plan.bns:
            node, _ = self._descend(name + ".x")
            node.register_buffer("running_mean", torch.zeros(c))
            node.register_buffer("running_var", torch.ones(c))
            node.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        # torchvision's classifier stays in the state dict although Backbone never runs it (resnet.py:177)
        self.fc = nn.Linear(self.plan.out_channels, 1000)

    def _descend(self, dotted):
        parts = dotted.split(".")
        node = self
        for part in parts[:-1]:
            if not hasattr(node, part):
                node.add_module(part, _Node())
            node = getattr(node, part)
        return node, parts[-1]

    def bn_nodes(self):
        return [self._descend(name + ".x")[0] for name, _ in self.plan.bns]

    def forward(self, *a, **k):
        raise RuntimeError("ResNetParams holds parameters only; run it through VinceModel / Backbone")


class Backbone(nn.Module):
    def __init__(self, args, arch, final_layer=None):
        super().__init__()
        self.args = args
        self.arch = arch
        self.model = ResNetParams(arch)
        n_children = 10   # conv1 bn1 relu maxpool layer1-4 avgpool fc
        if final_layer is None:
            final_layer = n_children
        if final_layer < 0:
            final_layer = n_children + final_layer
        self.final_layer = final_layer
        self.output_channels = self.model.plan.out_channels
        self._owner = None

    def bind_owner(self, owner):
        self._owner = weakref.ref(owner)

    def forward(self, x, final_layer=None):
        """Runs conv1..layer4 and returns the NCHW spatial features (reference Backbone.forward with final_layer=8)."""
        fl = self.final_layer if final_layer is None else final_layer
        if fl < 0:
            fl = 10 + fl
        if fl != 8:
            raise NotImplementedError("the HIP trunk implements final_layer=-2 (conv1..layer4) only, got %d" % fl)
        owner = self._owner() if self._owner is not None else None
        if owner is None:
            raise RuntimeError("Backbone is not attached to a VinceModel")
        return owner.extract_features(x, run_average_layer=False)["spatial_features"]


class ResNet18(Backbone):
    def __init__(self, args, final_layer=None):
        if getattr(args, "use_imagenet_weights", False):
            raise NotImplementedError("pretrained torchvision weights need network access; load a checkpoint instead")
        super().__init__(args, "ResNet18", final_layer)


class ResNet50(Backbone):
    def __init__(self, args, final_layer=None):
        if getattr(args, "use_imagenet_weights", False):
            raise NotImplementedError("pretrained torchvision weights need network access; load a checkpoint instead")
        super().__init__(args, "ResNet50", final_layer)
