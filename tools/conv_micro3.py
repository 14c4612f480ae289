"""Ablation timing of one 3x3 conv layer (layer3 shape) through vince_conv_igemm."""
import sys
import torch
sys.path.insert(0, ".")
from vince_amd import ops
N, H, W, Ci, Co = 256, 14, 14, 256, 256
dev = "cuda"
x = torch.randn(N, H, W, Ci, device=dev).bfloat16()
w = (torch.randn(Co, 9, Ci, device=dev) * 0.05).bfloat16()
out = torch.empty(N, H, W, Co, device=dev, dtype=torch.bfloat16)
d = ops.conv_desc(N, H, W, Ci, Co, 3, 1, 1)
for _ in range(3):
    ops.conv_igemm(d, x, w, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 30
e0.record()
for _ in range(n):
    ops.conv_igemm(d, x, w, out)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1000 / n
print("%.1f us  %.1f TF/s" % (us, 2.0 * N * H * W * Ci * Co * 9 / us / 1e6))
