"""Debug aid: two data-parallel ranks of the full solver on ONE GPU through the gloo backend (NCCL refuses two ranks on
one device).  Exercises the multi-rank orchestration -- parameter / queue broadcast, bucketed gradient all-reduce behind
the engine's bucket events, key all-gather, replicated enqueue -- and checks the replicas stay identical.
Launch: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/dp2_one_gpu.py"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, ".")
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
from vince_amd.config import make_args                      # noqa: E402
from vince_amd.data_source import SyntheticFrames          # noqa: E402
from vince_amd.solvers.vince_solver import VinceSolver     # noqa: E402

args = make_args(backbone="ResNet18", batch_size=16, vince_queue_size=128, input_size=(64, 64), compute_dtype="bf16",
                 batch_source=SyntheticFrames(16, 64, 64, 1, device="cuda:0", seed=100 + rank),
                 pytorch_gpu_ids=[0], feature_extractor_gpu_ids=[0], log_frequency=10 ** 9)
solver = VinceSolver(args)
solver.reset_epoch()
assert solver.reducer is not None
losses = [float(solver.run_train_iteration()[0]["nce_loss"]) for _ in range(4)]
torch.cuda.synchronize()
bad = False
# replicas must hold identical parameters and identical queues after every rank applied the same reduced gradient
for name, t in (("params", solver.model._flat), ("queue", solver.vince_queue.vector_queue)):
    ref = t.clone()
    dist.broadcast(ref, src=0)
    diff = float((ref - t).abs().max())
    print("rank %d %s max |replica - rank0| = %.3e (max |x| %.3e)" % (rank, name, diff, float(t.abs().max())), flush=True)
    bad = bad or diff != 0.0
assert not bad
# the sample counter counts every rank's samples (a single process at the global batch of world * 16 would show the same number)
assert solver.samples_per_step == world * 16 and solver.iteration == 4 * world * 16, (solver.samples_per_step, solver.iteration)
# 4 steps x world*16 keys: 128 rows written into K = 128 -> the tail sits at the end (it wraps on the next enqueue)
assert solver.vince_queue.current_tail in (0, 128), solver.vince_queue.current_tail
print("rank %d ok: losses %s, tail %d" % (rank, ["%.4f" % l for l in losses], solver.vince_queue.current_tail))
dist.destroy_process_group()
