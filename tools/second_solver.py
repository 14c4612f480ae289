"""Does a SECOND solver in one process run as fast as the first?  (bench.py's extra legs build one solver after another.)
Usage: python tools/second_solver.py [first dtype] [second dtype]"""
import gc
import sys
import time
import torch
sys.path.insert(0, ".")
from vince_amd.config import make_args
from vince_amd.solvers.vince_solver import VinceSolver
from bench import PooledFrames


def run(dtype, steps=8, warm=3, **kw):
    src = PooledFrames(256, 224, 224, 1, "cuda:0", pool=2, rank=0, world=1)
    args = make_args(backbone="ResNet50", batch_size=256, vince_queue_size=65536, vince_embedding_size=128, vince_temperature=0.2,
                     compute_dtype=dtype, base_lr=0.03, input_size=(224, 224), batch_source=src, log_frequency=10 ** 9, **kw)
    s = VinceSolver(args)
    s.reset_epoch()
    for _ in range(warm):
        s.run_train_iteration()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        s.run_train_iteration()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / steps * 1e3
    del s, src
    gc.collect()
    import os
    if os.environ.get("CLEAR_NOGRAD_WS") == "1":
        from vince_amd import engine
        engine._NOGRAD_WS.clear()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    print("reserved MB after free", torch.cuda.memory_reserved() >> 20, file=sys.stderr)
    return ms


a, b = (sys.argv[1:] + ["bf16", "x3"])[:2]
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    r = [run(a), run(b), run(b), run(a)]
print("first %s %.2f ms | then %s %.2f ms | %s again %.2f ms | %s again %.2f ms" % (a, r[0], b, r[1], b, r[2], a, r[3]))
