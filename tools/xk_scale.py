import sys, torch
sys.path.insert(0, ".")
from vince_amd import ops
dev="cuda"
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1000/n
K, Co = 256, 1024
for tiles in (32, 64, 128, 392, 784):
    rows = tiles*128
    x = torch.randn(rows, K, device=dev).bfloat16(); w = (torch.randn(Co, K, device=dev)*0.05).bfloat16()
    out = torch.empty(rows, Co, device=dev, dtype=torch.bfloat16); so = torch.zeros(16, Co, 2, device=dev, dtype=torch.float64)
    a = t(lambda: ops.conv_expand_stats(x, w, out, stats=so, replicas=16))
    d = ops.conv_desc(1, rows, 1, K, Co, 1, 1, 0)
    b = t(lambda: ops.conv_igemm(d, x.view(1, rows, 1, K), w.view(Co, 1, K), out.view(1, rows, 1, Co), stats=so, replicas=16))
    print("tiles %4d (per WG %.2f): conv_xk %.1f us | igemm %.1f us" % (tiles, tiles/32, a, b))
