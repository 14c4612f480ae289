"""Weight-gradient launches of the ResNet-50 layers at the benchmark batch, timed alone (bf16).  Usage: wgrad_micro.py [label]
Measurement build: VINCE_HIP_LIB=.../libvince_hip_measure.so VINCE_WGRAD_ABLATE=<bits> (1 no atomics, 2 no main loop, 4 no DMA, 8 no LDS
fragment reads, 16 no MFMA) / VINCE_WGRAD_BLOCKS=n.  WGRAD_SHAPES=0,3 restricts the list; WGRAD_REPS=n launches per shape (PMC runs: no warm-up needed)."""
import sys
import torch
sys.path.insert(0, ".")
from vince_amd import ops
N = 256
# hw, ci, co, k
SHAPES = [(14, 256, 256, 3), (28, 128, 128, 3), (56, 64, 64, 3), (7, 512, 512, 3), (14, 256, 1024, 1), (14, 1024, 256, 1),
          (28, 128, 512, 1), (28, 512, 128, 1), (56, 64, 256, 1), (56, 256, 64, 1), (7, 512, 2048, 1), (7, 2048, 512, 1)]
import os
if os.environ.get("WGRAD_SHAPES"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["WGRAD_SHAPES"].split(",")]
res = []
for hw, ci, co, k in SHAPES:
    x = torch.randn(N, hw, hw, ci, device="cuda").bfloat16()
    dy = torch.randn(N, hw, hw, co, device="cuda").bfloat16()
    dw = torch.zeros(co, k * k, ci, device="cuda")
    d = ops.conv_desc(N, hw, hw, ci, co, k, 1, k // 2)
    det = os.environ.get("WGRAD_DET") == "1"
    scratch = None
    if det:
        import ctypes
        from vince_amd._lib import lib
        need = lib().vince_conv_wgrad_scratch_bytes(ctypes.byref(d), ops.dtype_code(x), ci)
        scratch = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
    run = (lambda: ops.conv_wgrad_det(d, x, dy, dw, scratch=scratch)) if det else (lambda: ops.conv_wgrad(d, x, dy, dw))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = int(os.environ.get("WGRAD_REPS", "20"))
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / n
    res.append("%dx%d %d<-%d k%d %.1f us (%.0f TF/s)" % (hw, hw, co, ci, k, us, 2.0 * N * hw * hw * co * ci * k * k / us / 1e6))
print(sys.argv[1] if len(sys.argv) > 1 else "", " | ".join(res))
