"""Measured ceilings on this box next to the spec ones: dense bf16 GEMM through the vendor library
(torch.matmul -> hipBLASLt), fp32 GEMM, and HBM stream copy / read / write.  Calibration only -- not a product path."""
import torch

dev = "cuda"


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


for n in (4096, 8192):
    x = torch.randn(n, n, device=dev).bfloat16()
    y = torch.randn(n, n, device=dev).bfloat16()
    t = timeit(lambda: torch.matmul(x, y))
    print("bf16 GEMM %d^3 (vendor library): %.1f TFLOP/s   [spec dense peak ~2500]" % (n, 2.0 * n ** 3 / t / 1e12))
x = torch.randn(8192, 8192, device=dev)
y = torch.randn(8192, 8192, device=dev)
t = timeit(lambda: torch.matmul(x, y), n=5)
print("fp32 GEMM 8192^3 (vendor library): %.1f TFLOP/s   [spec matrix peak ~157]" % (2.0 * 8192 ** 3 / t / 1e12))
nbytes = 2 << 30
src = torch.empty(nbytes, dtype=torch.uint8, device=dev).random_(0, 255)
dst = torch.empty_like(src)
t = timeit(lambda: dst.copy_(src))
print("HBM stream copy 2 GiB: %.2f TB/s read+write   [spec ~8]" % (2.0 * nbytes / t / 1e12))
t = timeit(lambda: dst.zero_())
print("HBM write-only (fill 2 GiB): %.2f TB/s" % (nbytes / t / 1e12))

# ---- the library's own float4 copy kernel (vince_stream_copy): the streaming ceiling element-wise passes are held against
import sys
sys.path.insert(0, ".")
from vince_amd import _lib
L = _lib.lib()
for nbytes in (256 << 20, 1 << 30, 2 << 30):
    s = src[:nbytes]
    d = dst[:nbytes]
    for nt in (0, 1, 2, 3, 4, 5):     # bit 0: nt loads / stores; bits 1-2: 0 grid-stride 16 B per lane, 1 = 32 B per lane, 2 = block-contiguous
        best = None
        for blocks in (1024, 2048, 4096, 8192, 16384):
            st = torch.cuda.current_stream().cuda_stream
            t = timeit(lambda: _lib.check(L.vince_stream_copy(d.data_ptr(), s.data_ptr(), nbytes, blocks, nt, st)), n=10)
            r = 2.0 * nbytes / t / 1e12
            if best is None or r > best[0]:
                best = (r, blocks)
            print("vince_stream_copy %4d MiB shape=%d nt=%d blocks=%5d: %.2f TB/s read+write" % (nbytes >> 20, nt >> 1, nt & 1, blocks, r))
        print("  best %d MiB shape=%d nt=%d: %.2f TB/s at %d blocks" % (nbytes >> 20, nt >> 1, nt & 1, best[0], best[1]))
