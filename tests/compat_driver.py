"""Driver written against the REFERENCE's import names only (run with compat/ on sys.path, see compat/README.md): the
control flow of the reference's entry point (solver_runner.py:12-54 -- loggers unless --debug, solver from the
class registry, optional initial validation, epochs of [reset_epoch, iterations with the 500-step linear warm-up,
run_val], end, save in `finally`) restated as a test harness.  Prints one JSON line with what the loggers received."""
import json
import os
import sys

import arg_parser                                            # reference: `import arg_parser`
from dg_util.python_utils import tensorboard_logger          # reference: solver_runner.py:6 (stand-in on this box)
from solvers.base_solver import BaseSolver                   # reference: solver_runner.py:9


class RecordingLogger(tensorboard_logger.Logger):
    def __init__(self, log_dir):
        super().__init__(log_dir)
        self.dicts, self.scalars = [], []

    def dict_log(self, scalars, step):
        self.dicts.append((int(step), sorted(scalars)))
        super().dict_log(scalars, step)

    def scalar_summary(self, tag, value, step=None, increment_counter=False):
        self.scalars.append((tag, float(value)))
        super().scalar_summary(tag, value, step, increment_counter)


def main():
    args = arg_parser.parse_args()
    train_logger = val_logger = None
    if not args.debug:
        train_logger = RecordingLogger(os.path.join(args.tensorboard_dir, "train"))
        val_logger = RecordingLogger(os.path.join(args.tensorboard_dir, "val"))
    solver: BaseSolver = args.solver(args, train_logger, val_logger)
    lrs = []
    step = 1
    try:
        if args.test_first:
            solver.reset_epoch()
            solver.run_val()
        peak = solver.adjust_learning_rate()
        while solver.epoch < args.epochs:
            solver.reset_epoch()
            for _ in range(solver.iterations_per_epoch):
                if args.use_warmup and step <= 500:
                    for group in solver.optimizer.param_groups:
                        group["lr"] = min(1.0, step / 500.0) * peak
                    step += 1
                lrs.append(solver.optimizer.param_groups[0]["lr"])
                solver.run_train_iteration()
            solver.run_val()
            solver.epoch += 1
        solver.end()
    finally:
        if args.save:
            solver.save()
    print("COMPAT_RESULT " + json.dumps({
        "solver": type(solver).__module__ + "." + type(solver).__name__,
        "model": type(solver.model).__module__,
        "queue": type(solver.vince_queue).__module__,
        "iteration": int(solver.iteration), "tail": int(solver.vince_queue.current_tail), "lrs": lrs, "peak": peak,
        "train_dicts": train_logger.dicts if train_logger else None,
        "train_scalars": train_logger.scalars if train_logger else None,
        "events_file": os.path.exists(os.path.join(args.tensorboard_dir, "train", "events.jsonl"))}))


if __name__ == "__main__":
    main()
