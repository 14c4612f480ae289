"""One conv layer through vince_conv_igemm, N times (PMC runs).  Usage: conv_one.py hw ci co k [n=20]"""
import sys
import torch
sys.path.insert(0, ".")
from vince_amd import ops
hw, ci, co, k = [int(v) for v in sys.argv[1:5]]
n = int(sys.argv[5]) if len(sys.argv) > 5 else 20
N = 256
x = torch.randn(N, hw, hw, ci, device="cuda").clamp_(min=0).bfloat16()
w = (torch.randn(co, k * k, ci, device="cuda") * 0.05).bfloat16()
out = torch.empty(N, hw, hw, co, device="cuda", dtype=torch.bfloat16)
stats = torch.zeros(ops.STATS_REPLICAS, co, 2, device="cuda", dtype=torch.float64)
d = ops.conv_desc(N, hw, hw, ci, co, k, 1, k // 2)
for _ in range(n):
    ops.conv_igemm(d, x, w, out, stats=stats)
torch.cuda.synchronize()
