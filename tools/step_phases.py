"""Where one training step's time goes, phase by phase, WITHOUT a profiler attached: CUDA events at the phase boundaries of
VinceSolver.run_train_iteration (its `phase_marks` hook) -- both forwards (key encoder on its stream beside the query encoder), the
head + InfoNCE + metrics, backward + optimiser, enqueue + EMA.  BASELINE config 3 by default.
Usage: python tools/step_phases.py [steps=20] [dtype=bf16]"""
import sys
import contextlib
import io
import torch
sys.path.insert(0, ".")
import bench
from vince_amd.config import make_args
from vince_amd.solvers.vince_solver import VinceSolver

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda", 0)
pool = bench.PooledFrames(256, 224, 224, 1, dev, pool=4, rank=0, world=1)
args = make_args(backbone="ResNet50", batch_size=256, vince_queue_size=65536, vince_embedding_size=128, vince_temperature=0.2,
                 compute_dtype=dtype, input_size=(224, 224), base_lr=0.03, pytorch_gpu_ids=[0], feature_extractor_gpu_ids=[0],
                 batch_source=pool, log_frequency=10 ** 9, iterations_per_epoch=10 ** 9)
with contextlib.redirect_stdout(io.StringIO()):
    solver = VinceSolver(args)
    solver.reset_epoch()
for _ in range(5):
    solver.run_train_iteration()
torch.cuda.synchronize()
acc = {}
order = []
t0 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(steps):
    solver.phase_marks = []
    solver.run_train_iteration()
    marks = solver.phase_marks
    torch.cuda.synchronize()
    for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
        acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1)
        if n1 not in order:
            order.append(n1)
solver.phase_marks = None
tot = sum(acc.values()) / steps
print("%s step, %d steps (synchronised between steps): %s | sum %.3f ms" %
      (dtype, steps, "  ".join("%s %.3f ms" % (n, acc[n] / steps) for n in order), tot))
