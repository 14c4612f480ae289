"""The convolutions of ResNet-50 at the benchmark batch (N = 256), timed alone: forward, input gradient (stride-1 layers) and weight
gradient as split-half products (x3: fp32 tensors, f16 hi/lo forward, bf16 hi/lo gradients; x1: the gradient launches of x3f -- single
bfloat16 products on the same fp32 tensors, forward as x3) beside the bf16 and exact-fp32 kernels.
Usage: python tools/x3_micro.py [label]   (X3_SHAPES=0,3 restricts the list; X3_MODES=x3,bf16,fp32; X3_OPS=fwd,dgrad,wgrad)"""
import os
import sys
import torch
sys.path.insert(0, ".")
from vince_amd import ops

N = 256
# hw, ci, co, k
SHAPES = [(56, 64, 64, 3), (56, 64, 256, 1), (56, 256, 64, 1), (28, 128, 128, 3), (28, 128, 512, 1), (28, 512, 128, 1),
          (14, 256, 256, 3), (14, 256, 1024, 1), (14, 1024, 256, 1), (7, 512, 512, 3), (7, 512, 2048, 1), (7, 2048, 512, 1)]
if os.environ.get("X3_SHAPES"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["X3_SHAPES"].split(",")]
MODES = os.environ.get("X3_MODES", "x3,bf16,fp32").split(",")
OPS = os.environ.get("X3_OPS", "fwd,dgrad,wgrad").split(",")
REPS = int(os.environ.get("X3_REPS", "10"))


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / REPS


print(sys.argv[1] if len(sys.argv) > 1 else "", "VINCE_KNOBS=%s" % os.environ.get("VINCE_KNOBS", ""))
print("%-22s %-6s %s" % ("layer", "op", "  ".join("%16s" % m for m in MODES)))
tot = {(o, m): 0.0 for o in OPS for m in MODES}
for hw, ci, co, k in SHAPES:
    gf = 2.0 * N * hw * hw * co * ci * k * k / 1e9
    for op in OPS:
        cells = []
        for mode in MODES:
            dt = torch.bfloat16 if mode == "bf16" else torch.float32
            x3f, x3b = ("h", "b") if mode == "x3" else ("h", "1") if mode == "x1" else (None, None)
            x = torch.randn(N, hw, hw, ci, device="cuda").to(dt)
            w = (torch.randn(co, k * k, ci, device="cuda") * (2.0 / (ci * k * k)) ** 0.5)
            wk, wt = ops.prepare_weight(w, dt, want_transposed=True, x3=(mode in ("x3", "x1")))
            y = torch.empty(N, hw, hw, co, device="cuda", dtype=dt)
            d = ops.conv_desc(N, hw, hw, ci, co, k, 1, k // 2)
            if op == "fwd":
                us = timed(lambda: ops.conv_igemm(d, x, wk, y, x3=x3f))
            elif op == "dgrad":
                dd = ops.dgrad_descs(N, hw, hw, ci, co, k, 1, k // 2)[0]
                dy = torch.randn(N, hw, hw, co, device="cuda").to(dt)
                dx = torch.empty(N, hw, hw, ci, device="cuda", dtype=dt)
                us = timed(lambda: ops.conv_igemm(dd, dy, wt, dx, x3=x3b))
            else:
                dy = torch.randn(N, hw, hw, co, device="cuda").to(dt)
                dw = torch.zeros(co, k * k, ci, device="cuda")
                us = timed(lambda: ops.conv_wgrad(d, x, dy, dw, x3=x3b))
            tot[(op, mode)] += us
            cells.append("%7.1f us %5.0f TF" % (us, gf / us * 1e3))
        print("%-22s %-6s %s" % ("%dx%d %d<-%d k%d" % (hw, hw, co, ci, k), op, "  ".join(cells)))
for op in OPS:
    print("%-22s %-6s %s" % ("sum (one of each)", op, "  ".join("%13.1f us" % tot[(op, m)] for m in MODES)))
