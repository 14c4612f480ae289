"""The gradient epilogues of vince_conv_igemm at the benchmark batch (N = 256, bf16): layer3's block-input gradient (the expand shape,
256 -> 1024 at 14x14, residual-gradient join through the mask bytes + fused BatchNorm-backward sums) and its 3x3 input gradient on
the 8-wavefront core (fused sums with the ReLU recomputed from the saved conv output).  us per call, timed alone."""
import sys
import torch
sys.path.insert(0, ".")
from vince_amd import ops
from vince_amd._lib import EPI_ACCUMULATE
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


N = 256
for name, hw, K, Co in [("layer3 block-input gradient 256->1024", 14, 256, 1024), ("layer2 block-input gradient 128->512 (igemm route)", 28, 128, 512)]:
    rows = N * hw * hw
    dy = (torch.randn(rows, K, device=dev) * 0.1).bfloat16()
    wt = (torch.randn(Co, K, device=dev) * 0.05).bfloat16()
    out = torch.randn(rows, Co, device=dev).bfloat16()
    ylo = torch.randn(rows, Co, device=dev).bfloat16()
    amask = torch.randint(0, 256, (rows * Co // 8,), device=dev, dtype=torch.uint8)
    mean, invstd = torch.randn(Co, device=dev), torch.rand(Co, device=dev) + 0.5
    sums = torch.zeros(16, Co, 2, device=dev, dtype=torch.float64)
    br = ops.bn_reduce_arg(ylo, mean, invstd, sums, mask_bits=amask)
    d = ops.conv_desc(N, hw, hw, K, Co, 1, 1, 0)
    a = t(lambda: ops.conv_igemm(d, dy.view(N, hw, hw, K), wt.view(Co, 1, K), out.view(N, hw, hw, Co), flags=EPI_ACCUMULATE,
                                 acc_mask=amask, bnred=br, replicas=16))
    b = t(lambda: ops.conv_igemm(d, dy.view(N, hw, hw, K), wt.view(Co, 1, K), out.view(N, hw, hw, Co)))
    print("%s: join + fused sums %.1f us | plain store %.1f us" % (name, a, b))
for name, hw, C in [("layer3 3x3 input gradient (conv_m8)", 14, 256), ("layer2 3x3 input gradient", 28, 128)]:
    rows = N * hw * hw
    dy = (torch.randn(N, hw, hw, C, device=dev) * 0.1).bfloat16()
    wt = (torch.randn(C, 9, C, device=dev) * 0.05).bfloat16()
    out = torch.empty(N, hw, hw, C, device=dev, dtype=torch.bfloat16)
    ylo = torch.randn(rows, C, device=dev).bfloat16()
    mean, invstd = torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5
    msc, msh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    sums = torch.zeros(16, C, 2, device=dev, dtype=torch.float64)
    br = ops.bn_reduce_arg(ylo, mean, invstd, sums, mask_scale=msc, mask_shift=msh)
    d = ops.conv_desc(N, hw, hw, C, C, 3, 1, 1)
    a = t(lambda: ops.conv_igemm(d, dy, wt, out, bnred=br, replicas=16))
    b = t(lambda: ops.conv_igemm(d, dy, wt, out))
    print("%s: fused sums %.1f us | plain store %.1f us" % (name, a, b))
