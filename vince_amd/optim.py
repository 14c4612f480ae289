"""SGD with momentum and weight decay over VinceModel's flat parameter buffer (reference: optim.SGD(lr, weight_decay=
0.0001, momentum=0.9) on model.parameters(), solvers/vince_solver.py:253-259, stepped at :469).

Keeps the small part of torch.optim's surface the reference's driver touches: ``param_groups`` (solver_runner.py:36-43
rewrites ``pg["lr"]`` during warm-up; base_solver.py:107-129 reads ``initial_lr``), ``zero_grad()``, ``step()``.
One fused kernel per contiguous role range (trunk / projection head / jigsaw head); a range whose parameters
received no gradient in this step is skipped, as torch skips parameters with ``grad is None`` (App. D item 10).
Parameters outside the flat buffer (the optional ImageNet side decoders) go through a plain torch SGD.
"""
import torch

from . import ops


class FlatSGD:
    def __init__(self, model, lr, momentum=0.9, weight_decay=1e-4):
        self.model = model
        self.momentum, self.weight_decay = momentum, weight_decay
        self.param_groups = [{"params": list(model.parameters()), "lr": lr, "initial_lr": lr, "momentum": momentum,
                              "weight_decay": weight_decay}]
        flat, grad, n_train, _ = model.flat_parameters()
        self.momentum_buffer = torch.zeros(n_train, dtype=torch.float32, device=flat.device)
        self.grad_scale = 1.0   # 1/world_size when gradients were SUM-all-reduced
        extra = [p for p in (model.imagenet_decoders.parameters() if hasattr(model, "imagenet_decoders") else [])]
        self._extra = torch.optim.SGD(extra, lr=lr, momentum=momentum, weight_decay=weight_decay) if extra else None

    def zero_grad(self, set_to_none=True):
        self.model.zero_grad()

    def step(self):
        lr = float(self.param_groups[0]["lr"])
        flat, grad, n_train, _ = self.model.flat_parameters()
        if self.momentum_buffer.device != flat.device:
            self.momentum_buffer = self.momentum_buffer.to(flat.device)
        for role, (a, b) in self.model._segments.items():
            if b > a and self.model._touched.get(role, False):
                ops.sgd_flat(flat[a:b], grad[a:b], self.momentum_buffer[a:b], lr, self.momentum, self.weight_decay,
                             self.grad_scale)
        self.model._touch()
        if self._extra is not None:
            for g in self._extra.param_groups:
                g["lr"] = lr
            self._extra.step()

    def state_dict(self):
        return {"momentum_buffer": self.momentum_buffer, "param_groups": [{k: v for k, v in g.items() if k != "params"}
                                                                          for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.momentum_buffer.copy_(sd["momentum_buffer"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)

    def __repr__(self):
        g = self.param_groups[0]
        return "FlatSGD(lr=%g, momentum=%g, weight_decay=%g, flat_params=%d)" % (
            g["lr"], self.momentum, self.weight_decay, self.momentum_buffer.numel())
