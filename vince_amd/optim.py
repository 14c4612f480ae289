"""SGD with momentum and weight decay over VinceModel's flat parameter buffer (reference: optim.SGD(lr, weight_decay=
0.0001, momentum=0.9) on model.parameters(), solvers/vince_solver.py:253-259, stepped at :469).

Keeps the small part of torch.optim's surface the reference's driver touches: ``param_groups`` (solver_runner.py:36-43
rewrites ``pg["lr"]`` during warm-up; base_solver.py:107-129 reads ``initial_lr``), ``zero_grad()``, ``step()``.
One fused kernel per contiguous role range (trunk / projection head / jigsaw head).  A range that has NEVER received a
gradient is skipped, as torch skips parameters whose ``.grad`` is None (App. D item 10: the unused ``resnet.fc``, a head
that was never run).  Once a range has been used it is stepped on every later step, idle or not: the reference pins
torch==1.4.0 (requirements.txt:17), whose ``optimizer.zero_grad()`` zeroes gradients in place and never resets them to
None, so under ``--jigsaw`` alternation the idle head keeps receiving weight decay and momentum with a zero gradient.
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
        self._ever_touched = set()   # roles that have received a gradient at least once (their .grad is a tensor in torch 1.4)
        extra = [p for p in (model.imagenet_decoders.parameters() if hasattr(model, "imagenet_decoders") else [])]
        self._extra = torch.optim.SGD(extra, lr=lr, momentum=momentum, weight_decay=weight_decay) if extra else None

    def zero_grad(self, set_to_none=True):
        self.model.zero_grad()

    def step(self, defer_stem=False):
        """defer_stem=True (VinceSolver's loop; only meaningful when the model runs with defer_stem_join): every parameter but
        conv1.weight is stepped now -- beside the stem's weight gradient, still running on the engine's side stream -- and
        conv1.weight's own step is left as model._deferred_step, which VinceQueueModel.param_update / the next forward / zero_grad
        finish behind the stem event.  The default completes everything before returning."""
        lr = float(self.param_groups[0]["lr"])
        flat, grad, n_train, _ = self.model.flat_parameters()
        if self.momentum_buffer.device != flat.device:
            self.momentum_buffer = self.momentum_buffer.to(flat.device)
        if getattr(self.model, "_grad_zero_pending", False):   # zero_grad() with no backward since: gradients are zero
            grad.zero_()
            self.model._grad_zero_pending = False
        deferred_trunk = None
        for role, (a, b) in self.model._segments.items():
            if self.model._touched.get(role, False):
                self._ever_touched.add(role)
            # an idle range's flat gradient is all zeros here (the whole buffer is cleared before backward): the kernel then
            # applies exactly weight decay + momentum, which is what torch 1.4 does with an in-place-zeroed .grad
            if b > a and role in self._ever_touched:
                if role == "trunk" and getattr(self.model, "_stem_pending", False):
                    # conv1.weight is the first tensor of the flat buffer: [a, n1); its gradient is final behind the stem event only.
                    # Data parallel: the reducer holds back the whole last gradient bucket (model._late = (its end, a waiter)).
                    late = getattr(self.model, "_late", None)
                    n1 = late[0] if late else self.model._offs[1]
                    ops.sgd_flat(flat[n1:b], grad[n1:b], self.momentum_buffer[n1:b], lr, self.momentum, self.weight_decay,
                                 self.grad_scale)

                    early = [False]

                    def finish(a=a, n1=n1, lr=lr, scale=self.grad_scale, late=late):
                        if late:
                            late[1]()                        # the last bucket's all-reduce (which waited for the stem event)
                            self.model._late = None
                        self.model.finish_stem_grad()
                        self.model._deferred_split = None
                        f, g, _, _ = self.model.flat_parameters()
                        ops.sgd_flat(f[a:n1], g[a:n1], self.momentum_buffer[a:n1], lr, self.momentum, self.weight_decay, scale)
                        if early[0] and self.model.prepare_weights_early(2):
                            self.model.weights_current()     # (runs after step() has bumped the parameter version)
                    self.model._deferred_split = n1
                    if defer_stem:
                        self.model._deferred_step = finish
                        deferred_trunk = early
                    else:
                        finish()
                    continue
                ops.sgd_flat(flat[a:b], grad[a:b], self.momentum_buffer[a:b], lr, self.momentum, self.weight_decay,
                             self.grad_scale)
        if deferred_trunk is not None and self.model._deferred_split == self.model._offs[1]:
            # every parameter but conv1.weight has its new value: rebuild their compute copies now, beside the stem's weight gradient
            # (the head ranges were stepped by the launches above; the trunk's table entry 0 follows in finish())
            deferred_trunk[0] = self.model.prepare_weights_early(1)
        self.model._touch()
        if self._extra is not None:
            for g in self._extra.param_groups:
                g["lr"] = lr
            self._extra.step()

    def state_dict(self):
        return {"momentum_buffer": self.momentum_buffer, "ever_touched": sorted(self._ever_touched), "param_groups": [{k: v for k, v in g.items() if k != "params"}
                                                                          for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.momentum_buffer.copy_(sd["momentum_buffer"])
        self._ever_touched = set(sd.get("ever_touched", ()))
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)

    def __repr__(self):
        g = self.param_groups[0]
        return "FlatSGD(lr=%g, momentum=%g, weight_decay=%g, flat_params=%d)" % (
            g["lr"], self.momentum, self.weight_decay, self.momentum_buffer.numel())
