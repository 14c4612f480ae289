"""bn2 + ReLU in the operand path of conv3 (vince_conv_epi.bn_in) against the two launches it replaces, at the benchmark batch:
layer3's 256 -> 1024 at 14x14 and layer4's 512 -> 2048 at 7x7 (N = 256, bf16).  us per call, each route timed alone."""
import sys
import torch
sys.path.insert(0, ".")
from vince_amd import ops
dev = "cuda"


def t(fn, n=30):
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


for name, hw, K, Co in [("layer3 conv3", 14, 256, 1024), ("layer4 conv3", 7, 512, 2048)]:
    N = 256
    rows = N * hw * hw
    y = torch.randn(rows, K, device=dev).bfloat16()
    g, b = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
    w = (torch.randn(Co, K, device=dev) * 0.05).bfloat16()
    st = torch.zeros(4, K, 2, device=dev, dtype=torch.float64)
    st[0] = torch.stack([y.double().sum(0), (y.double() ** 2).sum(0)], 1)
    rm, rv = torch.zeros(K, device=dev), torch.ones(K, device=dev)
    desc = ops.conv_desc(N, hw, hw, K, Co, 1, 1, 0)
    out = torch.empty(N, hw, hw, Co, device=dev, dtype=torch.bfloat16)
    so = torch.zeros(16, Co, 2, device=dev, dtype=torch.float64)
    a = ops.bn_train_apply(y, st, rows, g, b, rm, rv, replicas=4)[0]
    t_apply = t(lambda: ops.bn_train_apply(y, st, rows, g, b, rm, rv, replicas=4))
    t_conv = t(lambda: ops.conv_igemm(desc, a.view(N, hw, hw, K), w.view(Co, 1, K), out, stats=so, replicas=16))
    bt, keep = ops.bn_train_arg(st, rows, g, b, rm, rv, replicas=4)
    t_fused = t(lambda: ops.conv_igemm(desc, y.view(N, hw, hw, K), w.view(Co, 1, K), out, stats=so, replicas=16, bn_in=bt))
    both = t(lambda: (ops.bn_train_apply(y, st, rows, g, b, rm, rv, replicas=4),
                      ops.conv_igemm(desc, a.view(N, hw, hw, K), w.view(Co, 1, K), out, stats=so, replicas=16)))
    if K == 256:
        xo = out.view(rows, Co)
        t_xk = t(lambda: ops.conv_expand_stats(a, w, xo, stats=so, replicas=16))
        t_xkbn = t(lambda: ops.conv_expand_stats(y, w, xo, stats=so, replicas=16, bn_in=bt))
        both_xk = t(lambda: (ops.bn_train_apply(y, st, rows, g, b, rm, rv, replicas=4), ops.conv_expand_stats(a, w, xo, stats=so, replicas=16)))
        print("   streaming kernel (conv_xk): plain %.1f us (%.2f TB/s; with the bn pass back to back %.1f) | with bn_in %.1f us" % (
            t_xk, rows * (K + Co) * 2 / t_xk / 1e6, both_xk, t_xkbn))
    print("%s (%dx%d %d<-%d): bn apply %.1f us + conv %.1f us = %.1f (back to back %.1f) | bn_in conv %.1f us" % (
        name, hw, hw, Co, K, t_apply, t_conv, t_apply + t_conv, both, t_fused))
