"""Micro-benchmark of vince_conv_igemm for 1x1 convs: time vs (K, Co, stats) at fixed pixel count."""
import sys
import torch
sys.path.insert(0, ".")
from vince_amd import ops

M = 802816
dev = "cuda"
print("%8s %6s %6s %6s %9s %9s %9s" % ("M", "K", "Co", "stats", "us", "TB/s", "TF/s"))
for K, Co in [(64, 64), (64, 128), (64, 256), (64, 512), (128, 256), (256, 256), (256, 64), (512, 64), (512, 128)]:
    x = torch.randn(M, K, device=dev).bfloat16().view(1, M, 1, K)
    w = (torch.randn(Co, 1, K, device=dev) * 0.1).bfloat16()
    out = torch.empty(1, M, 1, Co, device=dev, dtype=torch.bfloat16)
    d = ops.conv_desc(1, M, 1, K, Co, 1, 1, 0)
    for use_stats in (False, True):
        stats = torch.zeros(ops.STATS_REPLICAS, Co, 2, device=dev, dtype=torch.float64) if use_stats else None
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
        gb = (M * K + M * Co) * 2 / 1e9
        print("%8d %6d %6d %6s %9.1f %9.2f %9.1f" % (M, K, Co, use_stats, us, gb / us * 1e-3 * 1e3, 2.0 * M * K * Co / us / 1e6))
