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


def main(argv=None):
    args = arg_parser.parse_args(argv)
    _join_process_group()
    # dg_util's TensorBoard logger is not part of the path: a caller that wants logging constructs the solver itself
    solver = args.solver(args, None, None)
    try:
        _train(args, solver)
    except Exception:   # the reference prints the traceback and still saves (solver_runner.py:47-54)
        traceback.print_exc()
    finally:
        if args.save:
            print("Saving models")
            solver.save()


if __name__ == "__main__":
    main()
