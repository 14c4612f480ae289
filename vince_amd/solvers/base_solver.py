"""Solver template: the surface `solver_runner.main` drives (reference: solvers/base_solver.py:20-167).

Same attribute and method names as the reference's `BaseSolver`, organised differently: the learning-rate schedule is a
small value object, the per-epoch meters are built by one helper, and the hooks a concrete solver must provide are
declared in one table.
"""
import math

import tqdm

from .meters import RollingAverageMeter

_TIME_METERS = ("total_time", "data_cache_time", "forward_time", "metrics_time", "backward_time")
# hooks a concrete solver implements (base_solver.py:60-84, :131-140); `end` is optional
_REQUIRED_HOOKS = ("setup_dataloader", "setup_other", "get_batch", "setup_model", "setup_optimizer", "run_train_iteration",
                   "run_val")


class LearningRateSchedule:
    """base_solver.py:107-119 -- cosine over `epochs`, or x0.1 at every passed milestone."""

    def __init__(self, kind, epochs, milestones):
        self.kind, self.epochs, self.milestones = kind, epochs, tuple(milestones or ())

    def factor(self, epoch):
        if self.kind == "cos":
            return 0.5 * (1.0 + math.cos(math.pi * epoch / self.epochs))
        return 0.1 ** sum(1 for m in self.milestones if epoch >= m)

    def describe(self, start_lr, epoch):
        if self.kind == "cos":
            return ("Cosine learning rate schedule Start epoch %s End epoch %s\nStart LR %s End LR %s"
                    % (epoch, self.epochs, start_lr, start_lr * self.factor(self.epochs - 1)))
        return ("Step learning rate schedule Steps %s\nStart LR %s End LR %s"
                % (list(self.milestones), start_lr, start_lr * 0.1 ** len(self.milestones)))


def _missing_hook(name):
    def hook(self, *a, **k):
        raise NotImplementedError("%s.%s" % (type(self).__name__, name))
    hook.__name__ = name
    return hook


class BaseSolver:
    def __init__(self, args, train_logger=None, val_logger=None):
        self.args, self.input_size = args, args.input_size
        self.use_apex = False   # NVIDIA Apex is CUDA-only; mixed precision is args.compute_dtype here
        quiet = bool(args.debug)
        self.train_logger, self.val_logger = (None, None) if quiet else (train_logger, val_logger)
        self.model = self.optimizer = None
        self.iteration = self.epoch = self.logger_iteration = 0
        self.time_meters, self.metric_meters, self.loss_meters = {}, {}, {}
        self.freeze_feature_extractor = getattr(args, "freeze_feature_extractor", False)
        for stage in ("setup_dataloader", "setup_other", "setup_model", "setup_optimizer"):   # base_solver.py:38-41 order
            getattr(self, stage)()

    # ---- names ----------------------------------------------------------------------------------
    device = property(lambda self: self.args.pytorch_gpu_ids[0])
    solver_name = property(lambda self: type(self).__name__)
    model_name = property(lambda self: type(self.model).__name__ if self.model is not None else "Unknown")
    full_name = property(lambda self: "%s_%s" % (self.solver_name, self.model_name))

    @property
    def lr_schedule(self):
        """Built from `args` on every use: callers may edit the schedule fields between epochs, as with the reference."""
        a = self.args
        return LearningRateSchedule(a.lr_decay_type, a.epochs, getattr(a, "lr_step_schedule", ()))

    @property
    def iterations_per_epoch(self):
        raise NotImplementedError

    def end(self):
        """Optional tear-down hook."""

    # ---- learning rate ----------------------------------------------------------------------------
    def adjust_learning_rate(self):
        """Sets every param group's lr from its `initial_lr` and the schedule; returns the lr of the base-lr group."""
        k = self.lr_schedule.factor(self.epoch)
        shown = self.args.base_lr
        for group in self.optimizer.param_groups:
            group["lr"] = group["initial_lr"] * k
            if group["initial_lr"] == self.args.base_lr:
                shown = group["lr"]
        if self.train_logger is not None:
            for tag, value in (("epoch", self.epoch), ("lr", shown)):
                self.train_logger.scalar_summary("metrics/%s/%s" % (self.full_name, tag), value, step=self.iteration,
                                                 increment_counter=False)
        print("Epoch", self.epoch, "Learning rate", shown)
        return shown

    def print_optimizer(self):
        print("optimizer", self.optimizer)
        print(self.lr_schedule.describe(self.adjust_learning_rate(), self.epoch))

    # ---- epochs -----------------------------------------------------------------------------------
    def _fresh_meters(self):
        window = self.args.log_frequency
        new = lambda: RollingAverageMeter(window)   # noqa: E731
        self.time_meters.update({name: new() for name in _TIME_METERS})
        self.metric_meters.update({name: new() for name in self.model.get_metrics(None)})
        self.loss_meters.update({name: new() for name in self.model.loss(None)})
        if len(self.loss_meters) > 1:
            self.loss_meters["total_loss"] = new()

    def reset_epoch(self):
        self.logger_iteration = 0
        self._fresh_meters()
        self.adjust_learning_rate()
        self.model.train()
        summarise = getattr(self.train_logger, "network_conv_summary", None)
        if summarise is not None:
            summarise(self.model, self.iteration)

    def run_n_train_iterations(self, num_iterations: int):
        self.reset_epoch()
        for _ in tqdm.tqdm(range(num_iterations)):
            self.run_train_iteration()

    def save(self, num_to_keep=-1, sync=False):
        self.model.save(self.iteration, num_to_keep)


for _name in _REQUIRED_HOOKS:
    setattr(BaseSolver, _name, _missing_hook(_name))
