"""Times the gradient-epilogue launches of a few ResNet-50 layers (B=256, bf16): 1x1 dgrad with the masked residual
join and the fused BatchNorm-backward reduction (block-input gradient), and 1x1 / 3x3 dgrads with the self-mask reduction.
Usage: conv_micro_bwd.py [label]"""
import sys
import torch
sys.path.insert(0, ".")
from vince_amd import ops
dev = "cuda"
N = 256
# (name, hw, K = channels of dy, Co = channels of dx, k, join)
SHAPES = [("l1 conv1 dgrad+join", 56, 64, 256, 1, True), ("l2 conv1 dgrad+join", 28, 128, 512, 1, True),
          ("l3 conv1 dgrad+join", 14, 256, 1024, 1, True), ("l4 conv1 dgrad+join", 7, 512, 2048, 1, True),
          ("l3 conv3 dgrad+red", 14, 1024, 256, 1, False), ("l3 conv2 dgrad+red", 14, 256, 256, 3, False),
          ("l1 conv3 dgrad+red", 56, 256, 64, 1, False)]
res = []
for name, hw, ci, co, k, join in SHAPES:
    dy = torch.randn(N, hw, hw, ci, device=dev).bfloat16()
    wt = (torch.randn(co, k * k, ci, device=dev) * 0.05).bfloat16()       # [Ci_fwd][T][Co_fwd] as the dgrad "weights"
    dx = torch.randn(N, hw, hw, co, device=dev).bfloat16()
    y = torch.randn(N, hw, hw, co, device=dev).bfloat16()
    bits = torch.randint(0, 256, (N * hw * hw * co // 8,), device=dev, dtype=torch.uint8)
    mean, invstd = torch.randn(co, device=dev), torch.rand(co, device=dev) + 0.5
    msc, msh = torch.randn(co, device=dev), torch.randn(co, device=dev)
    sums = torch.zeros(ops.STATS_REPLICAS, co, 2, device=dev, dtype=torch.float64)
    d = ops.conv_desc(N, hw, hw, ci, co, k, 1, k // 2)
    if join:
        br = ops.bn_reduce_arg(y, mean, invstd, sums, mask_bits=bits)
        kw = dict(flags=ops.EPI_ACCUMULATE, acc_mask=bits, bnred=br)
    else:
        br = ops.bn_reduce_arg(y, mean, invstd, sums, mask_scale=msc, mask_shift=msh)
        kw = dict(bnred=br)
    for _ in range(3):
        ops.conv_igemm(d, dy, wt, dx, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.conv_igemm(d, dy, wt, dx, **kw)
    e1.record()
    torch.cuda.synchronize()
    res.append("%s %.1f" % (name, e0.elapsed_time(e1) * 1000 / n))
print(sys.argv[1] if len(sys.argv) > 1 else "", " | ".join(res))
