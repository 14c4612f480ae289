"""The stem's weight gradient at the benchmark batch (N = 256, 224x224, bf16: 64 channels x 7 packed row taps), timed alone."""
import sys
import torch
sys.path.insert(0, ".")
from vince_amd import ops
N, H, W = 256, 224, 224
x = torch.randn(N, 3, H, W, device="cuda")
xin = ops.input_nchw_to_rows(x, torch.bfloat16)
d = ops.stem_desc(N, H, W)
dy = torch.randn(N, d.Ho, d.Wo, 64, device="cuda").bfloat16()
dw = torch.zeros(64, 49, 3, device="cuda")
for _ in range(3):
    ops.conv_wgrad(d, xin, dy, dw, ci_dw=3)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.conv_wgrad(d, xin, dy, dw, ci_dw=3)
e1.record()
torch.cuda.synchronize()
print("stem weight gradient: %.1f us" % (e0.elapsed_time(e1) * 1000 / 20))
