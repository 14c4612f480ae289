"""Estimate for the Gram-statistics conv3 fusion: per ResNet-50 stage at B=256 bf16, times
  conv3 with statistics (today), conv3 without, conv3 with bias + in-place residual join + ReLU (the folded epilogue),
  the join bn_apply pass, and the Gram X^T X of the conv3 input through the weight-gradient kernel."""
import sys
import torch
sys.path.insert(0, ".")
from vince_amd import ops
from vince_amd._lib import EPI_ACCUMULATE, EPI_RELU
dev = "cuda"
N = 256


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


for name, hw, w in [("l1", 56, 64), ("l2", 28, 128), ("l3", 14, 256), ("l4", 7, 512)]:
    co = 4 * w
    x = torch.randn(N, hw, hw, w, device=dev).clamp_(min=0).bfloat16()
    wt = (torch.randn(co, 1, w, device=dev) * 0.05).bfloat16()
    out = torch.empty(N, hw, hw, co, device=dev, dtype=torch.bfloat16)
    z = torch.randn(N, hw, hw, co, device=dev).bfloat16()
    rows = N * hw * hw
    tiles = (rows + 127) // 128
    R = 16 if tiles >= 4096 else 4 if tiles >= 1024 else 1
    stats = torch.zeros(ops.STATS_REPLICAS, co, 2, device=dev, dtype=torch.float64)
    bias = torch.randn(co, device=dev)
    d = ops.conv_desc(N, hw, hw, w, co, 1, 1, 0)
    t_stats = t(lambda: ops.conv_igemm(d, x, wt, out, stats=stats, replicas=R))
    t_plain = t(lambda: ops.conv_igemm(d, x, wt, out))
    t_join = t(lambda: ops.conv_igemm(d, x, wt, z, bias=bias, flags=EPI_ACCUMULATE | EPI_RELU))
    sc, sh = torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev)
    idn = torch.randn(rows, co, device=dev).bfloat16()
    y2 = out.view(rows, co)
    t_bn = t(lambda: ops.bn_apply(y2, sc, sh, identity=idn, want_mask=True))
    dg = ops.conv_desc(N, hw, hw, w, w, 1, 1, 0)
    gram = torch.zeros(w, 1, w, device=dev)
    t_gram = t(lambda: ops.conv_wgrad(dg, x, x, gram))
    print("%s w=%d rows=%d: conv3+stats %.1f | conv3 plain %.1f | conv3+bias+join+relu (in place) %.1f | join bn_apply %.1f | gram(wgrad) %.1f"
          "  -> today %.1f, fused no-grad %.1f (+gram)" % (name, w, rows, t_stats, t_plain, t_join, t_bn, t_gram, t_stats + t_bn, t_join + t_gram))
