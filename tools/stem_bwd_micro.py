"""stem_pool_fwd and the fused stem backward (reduce + apply: bn1's backward with the max-pool gradient gathered on the fly) timed alone
at BASELINE config 3's size (256 x 112 x 112 x 64, bf16).  Usage: python tools/stem_bwd_micro.py"""
import sys
import torch
sys.path.insert(0, ".")
from vince_amd import _lib
L = _lib.lib()
N, H, W, C = 256, 112, 112, 64
dev = "cuda"
y = torch.randn(N, H, W, C, device=dev).bfloat16()
sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
out = torch.empty(N, 56, 56, C, device=dev, dtype=torch.bfloat16)
am = torch.empty(N, 56, 56, C, device=dev, dtype=torch.uint8)
dpool = torch.randn(N, 56, 56, C, device=dev).bfloat16()
mean, invstd, gamma = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.ones(C, device=dev)
sums = torch.zeros(16, C, 2, device=dev, dtype=torch.float64)
dy = torch.empty_like(y)
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
st = torch.cuda.current_stream().cuda_stream


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


p = lambda t: t.data_ptr()
fwd = lambda: _lib.check(L.vince_stem_pool_fwd(1, p(y), p(sc), p(sh), p(out), p(am), N, H, W, C, st))
red = lambda: _lib.check(L.vince_stem_bwd_reduce(1, p(dpool), p(am), p(y), p(mean), p(invstd), p(sums), N, H, W, C, st))
app = lambda: _lib.check(L.vince_stem_bwd_apply(1, p(dpool), p(am), p(y), p(mean), p(invstd), p(gamma), p(sums), p(dy), p(dg), p(db), N, H, W, C, st))
fwd()
print("stem_pool_fwd %.1f us | stem_bwd_reduce %.1f us | stem_bwd_apply %.1f us" % (timed(fwd), timed(red), timed(app)))
