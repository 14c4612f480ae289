"""Image-strip 3x3 kernel against vince_conv_igemm on layer1's conv2 (B=256, 56x56, 64 -> 64, bf16): values, statistics, time."""
import sys
import torch
sys.path.insert(0, ".")
from vince_amd import ops
dev = "cuda"


def t(fn, n=20):
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


N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = 56
x = torch.randn(N, H, 56, 64, device=dev).clamp_(min=0).bfloat16()
w = (torch.randn(64, 9, 64, device=dev) * 0.05).bfloat16()
o1 = torch.full((N, H, 56, 64), 7.0, device=dev).bfloat16()
o2 = torch.empty_like(o1)
s1 = torch.zeros(16, 64, 2, device=dev, dtype=torch.float64)
s2 = torch.zeros(16, 64, 2, device=dev, dtype=torch.float64)
d = ops.conv_desc(N, H, 56, 64, 64, 3, 1, 1)
ops.conv3x3_strip(x, w, o1, stats=s1, replicas=16)
ops.conv_igemm(d, x, w, o2, stats=s2, replicas=16)
torch.cuda.synchronize()
diff = (o1.float() - o2.float()).abs().max().item()
print("max |strip - igemm| = %.3e (max |ref| %.3f); stats rel diff %.3e" % (
    diff, o2.float().abs().max().item(), ((s1.sum(0) - s2.sum(0)).abs().max() / s2.sum(0).abs().max()).item()))
a = t(lambda: ops.conv3x3_strip(x, w, o1, stats=s1, replicas=16))
b = t(lambda: ops.conv_igemm(d, x, w, o2, stats=s2, replicas=16))
fl = 2.0 * N * H * 56 * 64 * 64 * 9
print("N=%d: strip %.1f us (%.0f TF/s, %.2f TB/s in+out) | igemm %.1f us" % (N, a, fl / a / 1e6, 2 * x.numel() * 2 / a / 1e6, b))
