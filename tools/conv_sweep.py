"""Tile-quantisation sweep: one conv shape at several batch sizes (workgroup counts).  Usage: conv_sweep.py hw ci co k N1,N2,..."""
import sys
import torch
sys.path.insert(0, ".")
from vince_amd import ops
hw, ci, co, k = [int(v) for v in sys.argv[1:5]]
for N in [int(v) for v in sys.argv[5].split(",")]:
    x = torch.randn(N, hw, hw, ci, device="cuda").clamp_(min=0).bfloat16()
    w = (torch.randn(co, k * k, ci, device="cuda") * 0.05).bfloat16()
    out = torch.empty(N, hw, hw, co, device="cuda", dtype=torch.bfloat16)
    stats = torch.zeros(ops.STATS_REPLICAS, co, 2, device="cuda", dtype=torch.float64)
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
    M = N * hw * hw
    fl = 2.0 * M * co * ci * k * k
    print("hw %d ci %d co %d k %d N %4d: M %7d tiles(256x128) %5d tiles(128x128) %5d  %7.1f us  %6.1f TF/s" %
          (hw, ci, co, k, N, M, (M + 255) // 256 * ((co + 127) // 128), (M + 127) // 128 * ((co + 127) // 128), us, fl / us / 1e6))
