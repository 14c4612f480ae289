"""Times bn_apply (plain and with the residual join + mask) over the BatchNorm shapes of ResNet-50 at B=256 (bf16) and
prints bytes moved / time.  Usage: bn_micro.py [label]"""
import sys
import torch
sys.path.insert(0, ".")
from vince_amd import ops
dev = "cuda"
SHAPES = [(802816, 64), (802816, 256), (200704, 128), (200704, 512), (50176, 256), (50176, 1024), (12544, 512), (12544, 2048)]
out = []
for rows, C in SHAPES:
    y = torch.randn(rows, C, device=dev).bfloat16()
    idn = torch.randn(rows, C, device=dev).bfloat16()
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    for name, kw, passes in (("plain", dict(), 2), ("join", dict(identity=idn, want_mask=True), 3)):
        if name == "join" and C < 256:
            continue
        for _ in range(3):
            ops.bn_apply(y, sc, sh, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ops.bn_apply(y, sc, sh, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / n
        nbytes = rows * C * 2 * passes + (rows * C // 8 if name == "join" else 0)
        out.append("%dx%d %s %.1fus %.2fTB/s" % (rows, C, name, us, nbytes / us / 1e6))
print(sys.argv[1] if len(sys.argv) > 1 else "", " | ".join(out))

# ---- backward: reduce alone, then reduce + apply (difference = the apply pass: reads dz, y; writes dx) ----
out = []
for rows, C in SHAPES:
    y = torch.randn(rows, C, device=dev).bfloat16()
    dz = torch.randn(rows, C, device=dev).bfloat16()
    mean, invstd, gamma = torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) + 0.5
    msc, msh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    dgamma, dbeta = torch.zeros(C, device=dev), torch.zeros(C, device=dev)

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
    tiles = (rows + 127) // 128
    R = 16 if tiles >= 4096 else 4 if tiles >= 1024 else 1          # csrc/trunk.hip replicas_for
    red = t(lambda: ops.bn_bwd_reduce(dz, y, mean, invstd, mask_scale=msc, mask_shift=msh, replicas=R))
    both = t(lambda: ops.bn_bwd(dz, None, y, mean, invstd, gamma, dgamma, dbeta, mask_scale=msc, mask_shift=msh, replicas=R))
    ap = both - red
    out.append("%dx%d reduce %.1fus %.2fTB/s apply %.1fus %.2fTB/s" % (rows, C, red, rows * C * 4 / red / 1e6, ap, rows * C * 6 / ap / 1e6))
print("bwd", " | ".join(out))
