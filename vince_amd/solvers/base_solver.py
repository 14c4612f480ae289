"""Solver template (reference solvers/base_solver.py:20-167): the surface solver_runner.main drives."""
import abc
from typing import Dict

import numpy as np
import torch
import tqdm

from .meters import RollingAverageMeter


class BaseSolver(abc.ABC):
    def __init__(self, args, train_logger=None, val_logger=None):
        self.args = args
        self.use_apex = False   # NVIDIA Apex is CUDA-only; mixed precision is args.compute_dtype here
        self.input_size = args.input_size
        self.model = None
        self.logger_iteration = 0
        self.train_logger = None
        self.val_logger = None
        if not args.debug:
            self.train_logger = train_logger
            self.val_logger = val_logger
        self.time_meters = {}
        self.metric_meters = {}
        self.loss_meters = {}
        self.iteration = 0
        self.epoch = 0
        self.optimizer = None
        self.freeze_feature_extractor = getattr(self.args, "freeze_feature_extractor", False)
        self.setup_dataloader()
        self.setup_other()
        self.setup_model()
        self.setup_optimizer()

    @property
    def device(self):
        return self.args.pytorch_gpu_ids[0]

    @property
    def model_name(self):
        return "Unknown" if self.model is None else type(self.model).__name__

    @property
    def solver_name(self):
        return type(self).__name__

    @property
    def full_name(self):
        return self.solver_name + "_" + self.model_name

    def setup_dataloader(self):
        raise NotImplementedError

    @property
    def iterations_per_epoch(self):
        raise NotImplementedError

    def setup_other(self):
        raise NotImplementedError

    def get_batch(self) -> Dict[str, torch.Tensor]:
        raise NotImplementedError

    def setup_model(self):
        raise NotImplementedError

    def setup_optimizer(self):
        raise NotImplementedError

    def end(self):
        pass

    def print_optimizer(self):
        print("optimizer", self.optimizer)
        start_lr = self.adjust_learning_rate()
        if self.args.lr_decay_type == "cos":
            print("Cosine learning rate schedule", "Start epoch", self.epoch, "End epoch", self.args.epochs)
            print("Start LR", start_lr, "End LR",
                  start_lr * 0.5 * (1.0 + np.cos(np.pi * (self.args.epochs - 1) / self.args.epochs)))
        else:
            print("Step learning rate schedule", "Steps", self.args.lr_step_schedule)
            print("Start LR", start_lr, "End LR", start_lr * 0.1 ** len(self.args.lr_step_schedule))

    def adjust_learning_rate(self):
        """Decay the learning rate based on schedule (base_solver.py:107-129)."""
        out_base_lr = self.args.base_lr
        for param_group in self.optimizer.param_groups:
            in_lr = param_group["initial_lr"]
            out_lr = in_lr
            if self.args.lr_decay_type == "cos":
                out_lr *= 0.5 * (1.0 + np.cos(np.pi * self.epoch / self.args.epochs))
            else:
                for milestone in self.args.lr_step_schedule:
                    out_lr *= 0.1 if self.epoch >= milestone else 1.0
            param_group["lr"] = out_lr
            if in_lr == self.args.base_lr:
                out_base_lr = out_lr
        if self.train_logger is not None:
            self.train_logger.scalar_summary("metrics/%s/epoch" % self.full_name, self.epoch, step=self.iteration,
                                             increment_counter=False)
            self.train_logger.scalar_summary("metrics/%s/lr" % self.full_name, out_base_lr, step=self.iteration,
                                             increment_counter=False)
        print("Epoch", self.epoch, "Learning rate", out_base_lr)
        return out_base_lr

    def reset_epoch(self):
        self.logger_iteration = 0
        n = self.args.log_frequency
        self.time_meters.update(dict(total_time=RollingAverageMeter(n), data_cache_time=RollingAverageMeter(n),
                                     forward_time=RollingAverageMeter(n), metrics_time=RollingAverageMeter(n),
                                     backward_time=RollingAverageMeter(n)))
        self.metric_meters.update({m: RollingAverageMeter(n) for m in self.model.get_metrics(None).keys()})
        self.loss_meters.update({k: RollingAverageMeter(n) for k in self.model.loss(None).keys()})
        if len(self.loss_meters) > 1:
            self.loss_meters["total_loss"] = RollingAverageMeter(n)
        self.adjust_learning_rate()
        self.model.train()
        if self.train_logger is not None and hasattr(self.train_logger, "network_conv_summary"):
            self.train_logger.network_conv_summary(self.model, self.iteration)

    def run_train_iteration(self):
        raise NotImplementedError

    def run_n_train_iterations(self, num_iterations: int):
        self.reset_epoch()
        for _ in tqdm.tqdm(range(num_iterations)):
            self.run_train_iteration()

    def run_val(self):
        raise NotImplementedError

    def save(self, num_to_keep=-1):
        self.model.save(self.iteration, num_to_keep)
