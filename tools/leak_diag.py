import gc, sys, io, contextlib, weakref
import torch
sys.path.insert(0, ".")
from vince_amd.config import make_args
from vince_amd.solvers.vince_solver import VinceSolver
from vince_amd.models.vince_model import VinceModel
from vince_amd.data_source import SyntheticFrames
src = SyntheticFrames(16, 64, 64, 1, device="cuda:0", seed=1)
args = make_args(backbone="ResNet18", batch_size=16, vince_queue_size=64, input_size=(64, 64), compute_dtype="bf16", batch_source=src, log_frequency=10**9)
with contextlib.redirect_stdout(io.StringIO()):
    s = VinceSolver(args)
    s.reset_epoch()
    for _ in range(2):
        s.run_train_iteration()
torch.cuda.synchronize()
w = weakref.ref(s.model)
wq = weakref.ref(s.queue_model)
del s
gc.collect()
print("model alive:", w() is not None, "queue model alive:", wq() is not None)
for name, obj in (("model", w()), ("queue_model", wq())):
    if obj is None:
        continue
    refs = gc.get_referrers(obj)
    for r in refs:
        t = type(r).__name__
        desc = ""
        if isinstance(r, dict):
            owners = [type(o).__name__ for o in gc.get_referrers(r)][:4]
            keys = [k for k, v in r.items() if v is obj][:4]
            desc = "dict keys=%s owners=%s" % (keys, owners)
        elif hasattr(r, "__class__"):
            desc = repr(r)[:120]
        print(name, "<-", t, desc)
print("---- tensors whose grad_fn chain reaches _EncodeFnBackward")
def reaches(fn, depth=0):
    if fn is None or depth > 12:
        return False
    if "EncodeFn" in type(fn).__name__:
        return True
    return any(reaches(n, depth + 1) for n, _ in fn.next_functions)
for o in gc.get_objects():
    try:
        if isinstance(o, torch.Tensor) and o.grad_fn is not None and reaches(o.grad_fn):
            print("tensor", tuple(o.shape), o.dtype, type(o.grad_fn).__name__)
            for r in gc.get_referrers(o):
                if r is gc.get_objects: continue
                d = ""
                if isinstance(r, dict):
                    d = "keys=%s owners=%s" % ([k for k, v in r.items() if v is o][:3], [type(x).__name__ for x in gc.get_referrers(r)][:4])
                print("    <-", type(r).__name__, d or repr(r)[:100])
    except Exception as e:
        pass
