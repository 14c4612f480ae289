"""Soak: N training steps of BASELINE config 3 (and a shorter stretch of the jigsaw mode) -- finite loss, no memory growth,
steady step time.  Usage: soak.py [steps]"""
import sys
import time
import torch
sys.path.insert(0, ".")
from vince_amd.config import make_args
from vince_amd.data_source import SyntheticFrames
from vince_amd.solvers.vince_solver import VinceSolver

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
for mode in ("moco", "vince"):
    kw = dict(num_frames=4, inter_batch_comparison=True, self_batch_comparison=True, jigsaw=True) if mode == "vince" else {}
    args = make_args(backbone="ResNet50", batch_size=256, vince_queue_size=65536, vince_embedding_size=128, vince_temperature=0.2,
                     compute_dtype="bf16", input_size=(224, 224), base_lr=0.03, log_frequency=10 ** 9, iterations_per_epoch=10 ** 9,
                     batch_source=SyntheticFrames(256, 224, 224, 4 if mode == "vince" else 1, device="cuda", seed=3), **kw)
    solver = VinceSolver(args)
    solver.reset_epoch()
    n = steps if mode == "moco" else steps // 4
    marks = {}
    t0 = time.time()
    for i in range(n):
        loss = solver.run_train_iteration()[0]
        if i in (20, n // 2, n - 1) or (mode == "vince" and i % 25 == 0 and i > 20):
            torch.cuda.synchronize()
            v = {k: float(x.detach()) for k, x in loss.items()}
            assert all(abs(x) < 1e4 and x == x for x in v.values()), v
            marks[i] = (torch.cuda.memory_allocated() >> 20, torch.cuda.max_memory_allocated() >> 20, round(time.time() - t0, 2), v)
    for k, m in marks.items():
        print(mode, "step", k, "alloc MiB", m[0], "peak MiB", m[1], "t", m[2], m[3])
    a = [m[0] for m in marks.values()]
    # (in jigsaw mode the live outputs of the last step differ by ~180 MiB between a jigsawed-query and a jigsawed-key step)
    assert max(a) - min(a) < 256 and a[-1] - a[len(a) // 2] < 256, ("memory grows", a)
    print(mode, "alloc MiB over time", a)
    del solver
    torch.cuda.empty_cache()
print("soak ok")
