"""Layer1's 3x3 input gradient at the benchmark batch (N = 256, 56 x 56, 64 -> 64, bf16) with the fused BatchNorm-backward reduction:
the image-strip kernel (vince_conv3x3_strip_dgrad) against the implicit-GEMM gradient launch it replaces, each timed alone."""
import sys
import torch
sys.path.insert(0, ".")
from vince_amd import ops
N, H = 256, 56
dev = "cuda"
dy = (torch.randn(N, H, 56, 64, device=dev) * 0.1).bfloat16()
w = (torch.randn(64, 9, 64, device=dev) * (2.0 / 576) ** 0.5)
_, wt = ops.prepare_weight(w, torch.bfloat16, want_transposed=True)
y = torch.randn(N, H, 56, 64, device=dev).bfloat16()
mean, invstd = torch.randn(64, device=dev), torch.rand(64, device=dev) + 0.5
msc, msh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.3
dx = torch.empty(N, H, 56, 64, device=dev, dtype=torch.bfloat16)
sums = torch.zeros(ops.STATS_REPLICAS, 64, 2, device=dev, dtype=torch.float64)
br = ops.bn_reduce_arg(y, mean, invstd, sums, mask_scale=msc, mask_shift=msh)
d = ops.dgrad_descs(N, H, 56, 64, 64, 3, 1, 1)[0]


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


print("implicit GEMM + fused reduction: %.1f us" % timed(lambda: ops.conv_igemm(d, dy, wt, dx, bnred=br, replicas=16)))
print("image strip    + fused reduction: %.1f us" % timed(lambda: ops.conv3x3_strip_dgrad(dy, wt, dx, bnred=br, replicas=16)))
print("implicit GEMM, plain: %.1f us" % timed(lambda: ops.conv_igemm(d, dy, wt, dx)))
print("image strip,   plain: %.1f us" % timed(lambda: ops.conv3x3_strip_dgrad(dy, wt, dx)))
