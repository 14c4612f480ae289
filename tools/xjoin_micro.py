"""Times vince_conv_expand_join against vince_conv_igemm's join epilogue on the layer1 / layer2 shapes (B=256, bf16)."""
import sys
import torch
sys.path.insert(0, ".")
from vince_amd import ops
from vince_amd._lib import EPI_ACCUMULATE, EPI_RELU
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


for name, hw, K in [("l1", 56, 64), ("l2", 28, 128)]:
    N, Co = 256, 4 * K
    rows = N * hw * hw
    x = torch.randn(rows, K, device=dev).clamp_(min=0).bfloat16()
    w = (torch.randn(Co, K, device=dev) * 0.05).bfloat16()
    z = torch.randn(rows, Co, device=dev).bfloat16()
    sc, sh = torch.rand(Co, device=dev) + 0.5, torch.randn(Co, device=dev)
    d = ops.conv_desc(N, hw, hw, K, Co, 1, 1, 0)
    a = t(lambda: ops.conv_expand_join(x, w, sc, sh, z))
    b = t(lambda: ops.conv_igemm(d, x.view(N, hw, hw, K), w.view(Co, 1, K), z.view(N, hw, hw, Co), bias=sh,
                                 flags=EPI_ACCUMULATE | EPI_RELU, out_scale=sc))
    zin = torch.randn(rows, Co, device=dev).bfloat16()
    zout, yraw = torch.empty_like(zin), torch.empty_like(zin)
    mask = torch.empty(rows * Co // 8, device=dev, dtype=torch.uint8)
    c = t(lambda: ops.conv_expand_join(x, w, sc, sh, zin, out=zout, y_raw=yraw, mask_out=mask))
    print("   training forward (y_raw + mask too): %.1f us (%.2f TB/s)" % (c, rows * (K + 3 * Co + Co / 16) * 2 / c / 1e6))
    yo = torch.empty(rows, Co, device=dev, dtype=torch.bfloat16)
    st = torch.zeros(16, Co, 2, device=dev, dtype=torch.float64)
    e = t(lambda: ops.conv_expand_stats(x, w, yo, stats=st, replicas=16))
    f = t(lambda: ops.conv_igemm(d, x.view(N, hw, hw, K), w.view(Co, 1, K), yo.view(N, hw, hw, Co), stats=st, replicas=16))
    print("   plain expand conv + statistics: streaming %.1f us (%.2f TB/s) | igemm %.1f us" % (e, rows * (K + Co) * 2 / e / 1e6, f))
    nbytes = rows * (K + 2 * Co) * 2
    print("%s rows=%d K=%d Co=%d: expand_join %.1f us (%.2f TB/s) | igemm join %.1f us (%.2f TB/s)" % (
        name, rows, K, Co, a, nbytes / a / 1e6, b, nbytes / b / 1e6))

# ---- the block-input gradient (expand-shaped dgrad + residual-gradient join + BatchNorm-backward sums)
for name, hw, K, Co in [("l1 dgrad", 56, 64, 256), ("l2.0 dgrad", 56, 128, 256), ("l2 dgrad", 28, 128, 512)]:
    N = 256
    rows = N * hw * hw
    dy = (torch.randn(rows, K, device=dev) * 0.1).bfloat16()
    wt = (torch.randn(Co, K, device=dev) * 0.05).bfloat16()
    out = torch.randn(rows, Co, device=dev).bfloat16()
    ylo = torch.randn(rows, Co, device=dev).bfloat16()
    amask = torch.randint(0, 256, (rows * Co // 8,), device=dev, dtype=torch.uint8)
    mean, invstd = torch.randn(Co, device=dev), torch.rand(Co, device=dev) + 0.5
    sums = torch.zeros(16, Co, 2, device=dev, dtype=torch.float64)
    br = ops.bn_reduce_arg(ylo, mean, invstd, sums, mask_bits=amask)
    a = t(lambda: ops.conv_expand_dgrad(dy, wt, out, accumulate=True, acc_mask=amask, bnred=br, replicas=16))
    d = ops.conv_desc(N, hw, hw, K, Co, 1, 1, 0)
    b = t(lambda: ops.conv_igemm(d, dy.view(N, hw, hw, K), wt.view(Co, 1, K), out.view(N, hw, hw, Co), flags=EPI_ACCUMULATE,
                                 acc_mask=amask, bnred=br, replicas=16))
    nbytes = rows * (K + 3 * Co + Co / 4) * 2
    print("%s rows=%d K=%d Co=%d: streaming %.1f us (%.2f TB/s) | igemm %.1f us" % (name, rows, K, Co, a, nbytes / a / 1e6, b))
