"""Epoch loop + 500-iteration linear warm-up (reference solver_runner.py:12-54), driving the BaseSolver surface."""
import os
import traceback

import tqdm

from . import arg_parser


def main(argv=None):
    args = arg_parser.parse_args(argv)
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    train_logger = val_logger = None   # dg_util's TensorBoard logger is not part of the path; pass your own via the API
    solver = args.solver(args, train_logger, val_logger)

    curr_iteration = 1
    try:
        if args.test_first:
            print("Running initial Val")
            solver.reset_epoch()
            solver.run_val()
        starting_lr = solver.adjust_learning_rate()
        while solver.epoch < args.epochs:
            solver.reset_epoch()
            print("Running Train")
            for ii in tqdm.tqdm(range(solver.iterations_per_epoch)):
                if args.use_warmup:
                    if curr_iteration <= 500:
                        lr_scale = min(1.0, curr_iteration / 500.0)
                        new_lr = lr_scale * starting_lr
                        for pg in solver.optimizer.param_groups:
                            pg["lr"] = new_lr
                        curr_iteration += 1
                solver.run_train_iteration()
            print("Running Val")
            solver.run_val()
            solver.epoch += 1
        solver.end()
    except Exception:
        traceback.print_exc()
    finally:
        if args.save:
            print("Saving models")
            solver.save()


if __name__ == "__main__":
    main()
