"""Times the projection-MLP GEMMs (fp32, 256 rows) through ops.linear_fwd.  Usage: linear_micro.py [label]"""
import sys
import torch
sys.path.insert(0, ".")
from vince_amd import ops
dev = "cuda"
res = []
for rows, cin, cout, relu in [(256, 2048, 2048, True), (256, 2048, 128, False), (256, 512, 512, True)]:
    x = torch.randn(rows, cin, device=dev)
    w = torch.randn(cout, cin, device=dev) * cin ** -0.5
    b = torch.randn(cout, device=dev)
    for _ in range(3):
        ops.linear_fwd(x, w, b, relu=relu)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        ops.linear_fwd(x, w, b, relu=relu)
    e1.record()
    torch.cuda.synchronize()
    res.append("%dx%dx%d %.1f us" % (rows, cout, cin, e0.elapsed_time(e1) * 1000 / n))
print(sys.argv[1] if len(sys.argv) > 1 else "", " | ".join(res))
