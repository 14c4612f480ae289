"""bf16 weight gradient of vince_conv_wgrad against torch autograd (fp32 on the same GPU) over a list of shapes.  Usage:
    VINCE_KNOBS=wgrad_tile=256128 python tools/wgrad_check.py"""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from vince_amd import ops
# N, H, W, Ci, Co, k, stride, pad
SHAPES = [(3, 9, 11, 64, 256, 1, 1, 0), (2, 14, 14, 64, 64, 3, 1, 1), (2, 15, 15, 128, 128, 3, 2, 1), (2, 14, 14, 256, 512, 1, 2, 0),
          (2, 7, 7, 512, 512, 3, 1, 1), (3, 14, 14, 256, 256, 3, 1, 1), (5, 9, 9, 512, 256, 1, 1, 0), (16, 14, 14, 256, 256, 3, 1, 1),
          (16, 28, 28, 128, 128, 3, 1, 1), (8, 56, 56, 64, 64, 3, 1, 1), (16, 14, 14, 1024, 256, 1, 1, 0), (16, 14, 14, 256, 1024, 1, 1, 0),
          (8, 56, 56, 64, 256, 1, 1, 0), (8, 56, 56, 256, 64, 1, 1, 0), (16, 28, 28, 256, 512, 1, 2, 0), (64, 7, 7, 512, 2048, 1, 1, 0)]
bad = 0
for N, H, W, Ci, Co, k, s, p in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(N, Ci, H, W, device="cuda", generator=g).bfloat16().float().requires_grad_(False)
    w = torch.zeros(Co, Ci, k, k, device="cuda", requires_grad=True)
    y = F.conv2d(x, w, None, s, p)
    dy = torch.randn(y.shape, device="cuda", generator=g).bfloat16().float()
    y.backward(dy)
    ref = w.grad.permute(0, 2, 3, 1).reshape(Co, k * k, Ci)
    d = ops.conv_desc(N, H, W, Ci, Co, k, s, p)
    dw = torch.zeros(Co, k * k, Ci, device="cuda")
    ops.conv_wgrad(d, x.permute(0, 2, 3, 1).contiguous().bfloat16(), dy.permute(0, 2, 3, 1).contiguous().bfloat16(), dw)
    torch.cuda.synchronize()
    err = ((dw - ref).abs().max() / ref.abs().max()).item()
    flag = "" if err < 1e-4 else "   <-- MISMATCH"
    if flag:
        bad += 1
        e = (dw - ref).abs()
        co, t, ci = [int(v) for v in torch.unravel_index(e.argmax(), e.shape)]
        wrong = (e > 1e-3 * ref.abs().max())
        flag += " worst at co %d tap %d ci %d; wrong co range %s ci range %s taps %s" % (
            co, t, ci, sorted(set((wrong.nonzero()[:, 0] // 32 * 32).tolist()))[:10], sorted(set((wrong.nonzero()[:, 2] // 32 * 32).tolist()))[:10],
            sorted(set(wrong.nonzero()[:, 1].tolist())))
    print("%-40s err %.2e%s" % ((N, H, W, Ci, Co, k, s, p), err, flag))
print("MISMATCHES", bad)
