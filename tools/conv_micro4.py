"""Times a few representative ResNet-50 conv layers (B=256, bf16) through vince_conv_igemm; with statistics like the
engine's forward.  Usage: conv_micro4.py [label]"""
import os
import sys
import torch
sys.path.insert(0, ".")
from vince_amd import ops
dev = "cuda"
SHAPES = [("l1 1x1 64->256", 56, 64, 256, 1), ("l1 3x3 64", 56, 64, 64, 3), ("l1 1x1 256->64", 56, 256, 64, 1),
          ("l2 3x3 128", 28, 128, 128, 3), ("l3 3x3 256", 14, 256, 256, 3), ("l3 1x1 1024->256", 14, 1024, 256, 1),
          ("l3 1x1 256->1024", 14, 256, 1024, 1), ("l4 3x3 512", 7, 512, 512, 3),
          ("l2 1x1 128->512", 28, 128, 512, 1), ("l2 1x1 512->128", 28, 512, 128, 1), ("l2 1x1 256->128@56", 56, 256, 128, 1),
          ("l4 1x1 512->2048", 7, 512, 2048, 1), ("l4 1x1 2048->512", 7, 2048, 512, 1)]
N = 256
res = []
for name, hw, ci, co, k in SHAPES:
    x = torch.randn(N, hw, hw, ci, device=dev).clamp_(min=0).bfloat16()   # post-ReLU-like input
    w = (torch.randn(co, k * k, ci, device=dev) * 0.05).bfloat16()
    out = torch.empty(N, hw, hw, co, device=dev, dtype=torch.bfloat16)
    stats = None if os.environ.get("NOSTATS") else torch.zeros(ops.STATS_REPLICAS, co, 2, device=dev, dtype=torch.float64)
    d = ops.conv_desc(N, hw, hw, ci, co, k, 1, k // 2)
    for _ in range(3):
        ops.conv_igemm(d, x, w, out, stats=stats)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.conv_igemm(d, x, w, out, stats=stats)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / n
    res.append("%s %.1f" % (name, us))
print(sys.argv[1] if len(sys.argv) > 1 else "", " | ".join(res))
