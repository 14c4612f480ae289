"""Entry point: epochs of train + validation with the 500-iteration linear learning-rate warm-up of the reference
(solver_runner.py:12-54), driving any solver that exposes the BaseSolver surface.

    python -m vince_amd.solver_runner <reference flags> [--compute-dtype bf16|fp32]
    python -m torch.distributed.run --nproc-per-node N -m vince_amd.solver_runner ...   (one process per GPU)
"""
import os
import traceback

import tqdm

from . import arg_parser

WARMUP_ITERATIONS = 500   # solver_runner.py:36-43


def _join_process_group():
    """One process per GPU under torch.distributed.run; a plain launch stays single-device."""
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 or "RANK" not in os.environ:
        return
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("nccl")


class _Warmup:
    """Linear ramp of every param group's lr from 0 to `peak` over the first WARMUP_ITERATIONS training iterations."""

    def __init__(self, enabled, peak):
        self.left = WARMUP_ITERATIONS if enabled else 0
        self.peak = peak

    def apply(self, optimizer):
        if self.left <= 0:
            return
        done = WARMUP_ITERATIONS - self.left + 1
        for group in optimizer.param_groups:
            group["lr"] = self.peak * min(1.0, done / float(WARMUP_ITERATIONS))
        self.left -= 1


def _train(args, solver):
    if args.test_first:
        print("Running initial Val")
        solver.reset_epoch()
        solver.run_val()
    warmup = _Warmup(args.use_warmup, solver.adjust_learning_rate())
    while solver.epoch < args.epochs:
        solver.reset_epoch()
        print("Running Train")
        for _ in tqdm.tqdm(range(solver.iterations_per_epoch)):
            warmup.apply(solver.optimizer)
            solver.run_train_iteration()
        print("Running Val")
        solver.run_val()
        solver.epoch += 1
    solver.end()


def _make_loggers(args):
    """solver_runner.py:15-20: no loggers with --debug, else one per split under args.tensorboard_dir.  dg_util's
    TensorBoard logger is used when it is installed; otherwise the JSON-lines logger with the same two methods the path
    calls.  Only rank 0 logs (replicas are identical)."""
    if args.debug or int(os.environ.get("RANK", "0")) != 0:
        return None, None
    try:
        from dg_util.python_utils import tensorboard_logger as tb
    except ImportError:
        from .utils import jsonl_logger as tb
    return (tb.Logger(os.path.join(args.tensorboard_dir, "train")), tb.Logger(os.path.join(args.tensorboard_dir, "val")))


def main(argv=None, train_logger=None, val_logger=None):
    """solver_runner.py:12-54.  Loggers: any object with `dict_log(dict, step)` / `scalar_summary(tag, value, step)`; when
    none is passed they are made the way the reference makes them (_make_loggers)."""
    args = arg_parser.parse_args(argv)
    _join_process_group()
    if train_logger is None and val_logger is None:
        train_logger, val_logger = _make_loggers(args)
    solver = args.solver(args, train_logger, val_logger)   # solver_runner.py:22
    try:
        _train(args, solver)
    except Exception:   # the reference prints the traceback and still saves (solver_runner.py:47-54)
        traceback.print_exc()
    finally:
        if args.save:
            print("Saving models")
            solver.save()   # may run on one rank's exception path: the default save holds no collective (VinceSolver.save)


if __name__ == "__main__":
    main()
