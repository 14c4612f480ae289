"""GPU parity of every HIP kernel behind the C ABI against torch-CPU restatements of the reference ops
(the oracle for float ops is torch CPU fp32).  Run with -m gpu on an MI355X."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import vince_oracle as vo  # noqa: E402


def _ops():
    from vince_amd import ops
    return ops


DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def to_nhwc(x, dtype, cpad=None):
    """CPU NCHW fp32 -> GPU NHWC dtype (channels zero-padded to cpad)."""
    x = x.permute(0, 2, 3, 1).contiguous()
    if cpad is not None and cpad > x.shape[-1]:
        x = F.pad(x, (0, cpad - x.shape[-1]))
    return x.to(DEV).to(dtype).contiguous()


def from_nhwc(x):
    return x.float().cpu().permute(0, 3, 1, 2).contiguous()


def q(x, dtype):
    """Round a CPU fp32 tensor through `dtype` (so that the only GPU/CPU difference is accumulation order)."""
    return x.to(dtype).float()


def tol(dtype, f32=2e-5, bf16=2e-2):
    return f32 if dtype == torch.float32 else bf16


def assert_close(a, b, dtype, f32=2e-5, bf16=2e-2, what=""):
    a, b = a.float().cpu(), b.float().cpu()
    scale = b.abs().max().item() + 1e-12
    err = (a - b).abs().max().item() / scale
    assert err < tol(dtype, f32, bf16), "%s: max err / max|ref| = %.3e (dtype %s)" % (what, err, dtype)


CONVS = [
    # N, H, W, Ci, Co, k, stride, pad
    (2, 14, 14, 64, 64, 3, 1, 1),
    (3, 9, 11, 64, 256, 1, 1, 0),
    (2, 15, 15, 128, 128, 3, 2, 1),
    (2, 14, 14, 256, 512, 1, 2, 0),
    (2, 7, 7, 512, 512, 3, 1, 1),
    (1, 38, 38, 64, 64, 3, 1, 1),
    # shapes the 8-wavefront 256 x 256 core takes when forced (Co and the input gradient's Ci multiples of 256, Ci multiple of 64):
    # several tiles with a ragged last one, a stride-2 3x3 (input gradient = four parity classes with their own tap maps)
    (3, 14, 14, 256, 256, 3, 1, 1),
    (5, 9, 9, 512, 256, 1, 1, 0),
    (3, 13, 15, 256, 512, 3, 2, 1),
    # enough pixels for several pixel-range splits of the weight gradient with a ragged last 32-pixel slice (M = 3703), and a 1x1 whose
    # pixel count is not a multiple of anything (M = 3 * 19 * 21 = 1197)
    (7, 23, 23, 64, 128, 3, 1, 1),
    (3, 19, 21, 128, 64, 1, 1, 0),
]


def weights_krsc(w, dtype, cip=None, x3=False):
    """OIHW fp32 CPU -> GPU [Co][T][Ci] master (fp32) -> compute copies (x3: the split-half layout of VINCE_F32X3)."""
    ops = _ops()
    Co, Ci, kh, kw = w.shape
    master = w.permute(0, 2, 3, 1).reshape(Co, kh * kw, Ci).contiguous().to(DEV)
    return ops.prepare_weight(master, dtype, cip=cip, want_transposed=True, x3=x3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", CONVS)
def test_conv_fwd_stats(cfg, dtype):
    ops = _ops()
    N, H, W, Ci, Co, k, s, p = cfg
    x = q(rnd(N, Ci, H, W, seed=1), dtype)
    w = q(rnd(Co, Ci, k, k, seed=2, scale=(2.0 / (Ci * k * k)) ** 0.5), dtype)
    ref = F.conv2d(x, w, None, s, p)
    wk, _ = weights_krsc(w, dtype)
    d = ops.conv_desc(N, H, W, Ci, Co, k, s, p)
    out = torch.empty(N, d.Ho, d.Wo, Co, device=DEV, dtype=dtype)
    stats = torch.zeros(ops.STATS_REPLICAS, Co, 2, device=DEV, dtype=torch.float64)
    ops.conv_igemm(d, to_nhwc(x, dtype), wk, out, stats=stats)
    stats = stats.sum(0)
    assert_close(from_nhwc(out), ref, dtype, what="conv fwd")
    o = out.float().cpu().reshape(-1, Co).double()
    np.testing.assert_allclose(stats[:, 0].cpu().numpy(), o.sum(0).numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(stats[:, 1].cpu().numpy(), (o * o).sum(0).numpy(), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_stem(dtype):
    ops = _ops()
    N, H, W = 2, 40, 36
    x = q(rnd(N, 3, H, W, seed=3), dtype)
    w = q(rnd(64, 3, 7, 7, seed=4, scale=0.1), dtype)
    ref = F.conv2d(x, w, None, 2, 3)
    cp = 4 if dtype == torch.float32 else 8
    xin = ops.input_nchw_to_nhwc(x.to(DEV), dtype)
    assert xin.shape[-1] == cp
    torch.testing.assert_close(xin[..., :3].float().cpu(), x.permute(0, 2, 3, 1), rtol=0, atol=0)
    assert float(xin[..., 3:].abs().max()) == 0.0
    wk, _ = weights_krsc(w, dtype, cip=cp)
    d = ops.conv_desc(N, H, W, cp, 64, 7, 2, 3)
    out = torch.empty(N, d.Ho, d.Wo, 64, device=DEV, dtype=dtype)
    ops.conv_igemm(d, xin, wk, out)
    assert_close(from_nhwc(out), ref, dtype, what="stem conv")
    # stem wgrad: only the 3 real input channels are written
    dy = q(rnd(N, 64, d.Ho, d.Wo, seed=5), dtype)
    xr = x.clone().requires_grad_(False)
    wr = w.clone().requires_grad_(True)
    F.conv2d(xr, wr, None, 2, 3).backward(dy)
    dw = torch.zeros(64, 49, 3, device=DEV)
    ops.conv_wgrad(d, xin, to_nhwc(dy, dtype), dw, ci_dw=3)
    ref_dw = wr.grad.permute(0, 2, 3, 1).reshape(64, 49, 3)
    assert_close(dw, ref_dw, dtype, f32=1e-4, what="stem wgrad")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hw", [(32, 32), (45, 51), (224, 224)])
def test_conv_stem_packed_rows(dtype, hw):
    """The engine's stem: 7 packed row taps over the [N][H][Wp][4] layout == conv2d(7x7, stride 2, pad 3), fwd + wgrad."""
    ops = _ops()
    H, W = hw
    N = 2
    x = q(rnd(N, 3, H, W, seed=3), dtype)
    w = q(rnd(64, 3, 7, 7, seed=4, scale=(2.0 / 147) ** 0.5), dtype).requires_grad_(True)
    y = F.conv2d(x, w, None, 2, 3)
    dy = q(rnd(*y.shape, seed=5), dtype)
    y.backward(dy)
    xin = ops.input_nchw_to_rows(x.to(DEV), dtype)
    Wp = ops.stem_row_width(W)
    assert xin.shape == (N, H, Wp, 4)
    torch.testing.assert_close(xin[:, :, 3:3 + W, :3].float().cpu(), x.permute(0, 2, 3, 1), rtol=0, atol=0)
    assert float(xin[:, :, :3].abs().max()) == 0.0 and float(xin[:, :, 3 + W:].abs().max()) == 0.0
    assert float(xin[..., 3].abs().max()) == 0.0
    # packed weights [Co][kh][kw*4 + c], zero at kw = 7 and c = 3
    wp = torch.zeros(64, 7, 8, 4)
    wp[:, :, :7, :3] = w.detach().permute(0, 2, 3, 1)
    wk = wp.reshape(64, 7, 32).to(DEV).to(dtype).contiguous()
    d = ops.stem_desc(N, H, W)
    out = torch.empty(N, d.Ho, d.Wo, 64, device=DEV, dtype=dtype)
    stats = torch.zeros(ops.STATS_REPLICAS, 64, 2, device=DEV, dtype=torch.float64)
    ops.conv_igemm(d, xin, wk, out, stats=stats)
    assert_close(from_nhwc(out), y.detach(), dtype, what="packed stem conv")
    dw = torch.zeros(64, 49, 3, device=DEV)
    ops.conv_wgrad(d, xin, to_nhwc(dy, dtype), dw, ci_dw=3)
    ref_dw = w.grad.permute(0, 2, 3, 1).reshape(64, 49, 3)
    assert_close(dw, ref_dw, dtype, f32=1e-4, what="packed stem wgrad")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", CONVS)
def test_conv_dgrad_wgrad(cfg, dtype):
    ops = _ops()
    N, H, W, Ci, Co, k, s, p = cfg
    x = q(rnd(N, Ci, H, W, seed=6), dtype).requires_grad_(True)
    w = q(rnd(Co, Ci, k, k, seed=7, scale=(2.0 / (Ci * k * k)) ** 0.5), dtype).requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    dy = q(rnd(*y.shape, seed=8), dtype)
    y.backward(dy)
    wk, wt = weights_krsc(w.detach(), dtype)
    dyg = to_nhwc(dy, dtype)
    # dgrad (store) then a second pass with ACCUMULATE must double it
    dx = torch.full((N, H, W, Ci), float("nan"), device=DEV, dtype=dtype)
    descs = ops.dgrad_descs(N, H, W, Ci, Co, k, s, p)
    if len(descs) < s * s:
        dx.zero_()
    for d in descs:
        ops.conv_igemm(d, dyg, wt, dx)
    assert_close(from_nhwc(dx), x.grad, dtype, what="dgrad")
    for d in descs:
        ops.conv_igemm(d, dyg, wt, dx, flags=ops.EPI_ACCUMULATE)
    assert_close(from_nhwc(dx), 2 * x.grad, dtype, bf16=3e-2, what="dgrad accumulate")
    if s == 1:
        # masked accumulate (residual join): out = dgrad + out * bit, one byte per 16-byte chunk of `out`
        ch = 4 if dtype == torch.float32 else 8
        old = q(rnd(N, Ci, H, W, seed=21), dtype)
        bits = torch.randint(0, 256, (N, H, W, Ci // ch), dtype=torch.uint8, generator=torch.Generator().manual_seed(3))
        keep = ((bits.unsqueeze(-1) >> torch.arange(ch, dtype=torch.uint8)) & 1).reshape(N, H, W, Ci).permute(0, 3, 1, 2)
        dx = to_nhwc(old, dtype).clone()
        ops.conv_igemm(descs[0], dyg, wt, dx, flags=ops.EPI_ACCUMULATE, acc_mask=bits.to(DEV))
        assert_close(from_nhwc(dx), x.grad + old * keep.float(), dtype, bf16=3e-2, what="dgrad masked accumulate")
    # fused BatchNorm-backward reduction in the dgrad epilogue == the stand-alone reduction over the stored gradient
    yb = to_nhwc(q(rnd(N, Ci, H, W, seed=22), dtype), dtype)
    mean, invstd = rnd(Ci, seed=23).to(DEV), (rnd(Ci, seed=24).abs() + 0.5).to(DEV)
    msc, msh = rnd(Ci, seed=25).to(DEV), rnd(Ci, seed=26, scale=0.3).to(DEV)
    ch = 4 if dtype == torch.float32 else 8
    mbits = torch.randint(0, 256, (N, H, W, Ci // ch), dtype=torch.uint8, generator=torch.Generator().manual_seed(4)).to(DEV)
    for kind in ("none", "bits", "self"):
        kw = {"bits": dict(mask_bits=mbits), "self": dict(mask_scale=msc, mask_shift=msh), "none": {}}[kind]
        sums = torch.zeros(ops.STATS_REPLICAS, Ci, 2, device=DEV, dtype=torch.float64)
        dx = torch.zeros(N, H, W, Ci, device=DEV, dtype=dtype)
        br = ops.bn_reduce_arg(yb, mean, invstd, sums, **kw)
        for d in descs:
            ops.conv_igemm(d, dyg, wt, dx, bnred=br)
        want = ops.bn_bwd_reduce(dx, yb, mean, invstd, **kw)
        got = sums.sum(0)
        scale = float(want.abs().max()) + 1e-6
        assert float((got - want).abs().max()) / scale < 1e-5, ("fused bn reduce", kind, float((got - want).abs().max()), scale)
    # wgrad, both operand-fetch variants
    fd = ops.conv_desc(N, H, W, Ci, Co, k, s, p)
    ref_dw = w.grad.permute(0, 2, 3, 1).reshape(Co, k * k, Ci)
    for variant in ([0, 1] if dtype == torch.bfloat16 else [0]):
        dw = torch.zeros(Co, k * k, Ci, device=DEV)
        ops.conv_wgrad(fd, to_nhwc(x.detach(), dtype), dyg, dw, variant=variant)
        assert_close(dw, ref_dw, dtype, f32=1e-4, what="wgrad variant %d" % variant)
    # the reproducible path (per-split slabs + fixed-order reduction): same values, and two runs agree to the bit
    dets = []
    for _ in range(2):
        dw = torch.zeros(Co, k * k, Ci, device=DEV)
        ops.conv_wgrad_det(fd, to_nhwc(x.detach(), dtype), dyg, dw)
        dets.append(dw)
    assert_close(dets[0], ref_dw, dtype, f32=1e-4, what="deterministic wgrad")
    assert torch.equal(dets[0], dets[1])


def _random_conv_cases(n=14, seed=20260929):
    """Seeded shapes off the ResNet grid: odd images, batch tails, channel counts that do not fill a 64- or 128-wide tile,
    both kernel sizes and strides (multi-tap layers need Ci / 8 to be a power of two: vince_conv_igemm's documented limit)."""
    rs = np.random.RandomState(seed)
    cases = []
    while len(cases) < n:
        k = int(rs.choice([1, 3]))
        s_ = int(rs.choice([1, 2]))
        ci = int(rs.choice([8, 16, 64, 128, 256] if k == 3 else [8, 24, 64, 72, 136, 256, 520]))
        co = int(rs.choice([8, 16, 64, 128, 256] if k == 3 else [8, 24, 64, 80, 128, 200, 264]))   # (the input gradient of a 3x3 reduces over Co)
        h, w = int(rs.randint(5, 34)), int(rs.randint(5, 34))
        cases.append((int(rs.randint(1, 4)), h, w, ci, co, k, s_, k // 2))
    return cases


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", _random_conv_cases())
def test_conv_random_shapes_fwd_dgrad_wgrad(cfg, dtype):
    """Forward (+ statistics), input gradient and weight gradient against torch on the CPU for seeded off-grid shapes."""
    ops = _ops()
    N, H, W, Ci, Co, k, s, p = cfg
    x = q(rnd(N, Ci, H, W, seed=31), dtype).requires_grad_(True)
    w = q(rnd(Co, Ci, k, k, seed=32, scale=(2.0 / (Ci * k * k)) ** 0.5), dtype).requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    dy = q(rnd(*y.shape, seed=33), dtype)
    y.backward(dy)
    wk, wt = weights_krsc(w.detach(), dtype)
    d = ops.conv_desc(N, H, W, Ci, Co, k, s, p)
    out = torch.full((N, d.Ho, d.Wo, Co), float("nan"), device=DEV, dtype=dtype)
    stats = torch.zeros(ops.STATS_REPLICAS, Co, 2, device=DEV, dtype=torch.float64)
    ops.conv_igemm(d, to_nhwc(x.detach(), dtype), wk, out, stats=stats)
    assert_close(from_nhwc(out), y.detach(), dtype, what="conv fwd %s" % (cfg,))
    o = out.float().cpu().reshape(-1, Co).double()
    np.testing.assert_allclose(stats.sum(0)[:, 0].cpu().numpy(), o.sum(0).numpy(), rtol=1e-5, atol=1e-3)
    dyg = to_nhwc(dy, dtype)
    dx = torch.full((N, H, W, Ci), float("nan"), device=DEV, dtype=dtype)
    descs = ops.dgrad_descs(N, H, W, Ci, Co, k, s, p)
    if len(descs) < s * s:
        dx.zero_()
    for dd in descs:
        ops.conv_igemm(dd, dyg, wt, dx)
    assert_close(from_nhwc(dx), x.grad, dtype, what="dgrad %s" % (cfg,))
    dw = torch.zeros(Co, k * k, Ci, device=DEV)
    ops.conv_wgrad(d, to_nhwc(x.detach(), dtype), dyg, dw)
    assert_close(dw, w.grad.permute(0, 2, 3, 1).reshape(Co, k * k, Ci), dtype, f32=1e-4, what="wgrad %s" % (cfg,))


def _rerun_conv_tests(extra_env, select="test_conv_fwd_stats or test_conv_dgrad_wgrad or test_conv_stem"):
    import os
    import subprocess
    import sys
    env = dict(os.environ, **extra_env)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_ops_gpu.py"), "-q", "-m", "gpu",
                        "-k", select, "-p", "no:cacheprovider"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    tail = r.stdout[-1500:] + r.stderr[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail


def test_conv_suite_through_the_register_staged_fallback():
    """Tensors beyond the 31-bit buffer offsets of the direct-to-LDS kernels (> 2 GiB) take the register-staged conv and
    wgrad kernels; no test tensor is that large, so the same parity tests are re-run with those kernels forced."""
    _rerun_conv_tests({"VINCE_KNOBS": "dlds_min_k=1000000000,wgrad_dlds=0"})


def test_conv_suite_through_the_256_pixel_tiles():
    """The 256-pixel tile kernels (128ch x 256px for K >= 1024, 64ch x 256px for the stem / layer1) are only selected at
    benchmark-sized pixel counts; their selection thresholds are read from the environment once per process, so the conv
    parity tests of this file are re-run in a child process that forces both onto every shape."""
    _rerun_conv_tests({"VINCE_KNOBS": "big_min_k=1,big_min_tiles=1,narrow256_min_tiles=1"},
                      "test_conv_fwd_stats or test_conv_dgrad_wgrad or test_conv_stem")


def test_conv_suite_through_the_rotated_main_loop():
    """The rotated main loop (fragment reads one MFMA phase ahead, across the tile barrier; VINCE_KNOBS rot = bit per tile shape) is the
    default only for the 256x128 and the 3-stage tiles, which small test shapes do not reach: the conv parity tests are
    re-run with every bit set, once on the 128-pixel tiles (K thresholds lowered so the 3-stage ring is taken too) and once
    with the 256-pixel tiles forced."""
    _rerun_conv_tests({"VINCE_KNOBS": "rot=15,s3_min_k=512"},
                      "test_conv_fwd_stats or test_conv_dgrad_wgrad or test_conv_stem")
    _rerun_conv_tests({"VINCE_KNOBS": "rot=15,big_min_k=1,big_min_tiles=1,narrow256_min_tiles=1"},
                      "test_conv_fwd_stats or test_conv_dgrad_wgrad or test_conv_stem")


def test_conv_suite_through_the_8_wavefront_core():
    """conv_m8 (256 pixels x 256 channels, 8 wavefronts, csrc/conv_m8.hip) is selected for long bf16 reductions with at least 48
    tiles -- benchmark sizes only.  Re-run the conv parity tests with both thresholds at 1 so that every shape with Co % 256 == 0
    and Ci % 64 == 0 (forward) or the transposed condition (input gradient, with the join / BatchNorm-reduction epilogues) goes
    through it, and once more with it switched off (the default run mixes both)."""
    sel = "test_conv_fwd_stats or test_conv_dgrad_wgrad or test_conv_random_shapes_fwd_dgrad_wgrad"
    _rerun_conv_tests({"VINCE_KNOBS": "m8_min_k=1,m8_min_tiles=1"}, sel)
    _rerun_conv_tests({"VINCE_KNOBS": "m8=0"}, "test_conv_fwd_stats or test_conv_dgrad_wgrad")


def test_wgrad_suite_through_every_tile_of_the_transposing_kernel():
    """conv_wgrad_tr (csrc/conv_wgrad_tr.hip) picks its tile from the layer (128 wide wherever the side divides, else 64): every member
    is re-run by forcing `wgrad_tile = ct * 1000 + nt` onto every test shape the tile divides, and the LDS-DMA kernel it replaced for
    bf16 once with `wgrad_tr=0` (it still serves fp32)."""
    sel = "test_conv_dgrad_wgrad or test_conv_random_shapes_fwd_dgrad_wgrad or test_bn3_backward_algebra_vs_autograd"
    for tile in (64064, 64128, 128064):
        _rerun_conv_tests({"VINCE_KNOBS": "wgrad_tile=%d" % tile}, sel)
    _rerun_conv_tests({"VINCE_KNOBS": "wgrad_tr=0"}, sel)
    # the 1x1 layers run a four-stage ring by default (round 4); the three-stage instantiation they ran before stays selectable
    _rerun_conv_tests({"VINCE_KNOBS": "wgrad_stages4_linear=0"}, sel + " or test_linear_fwd_bwd")


def test_conv_suite_through_whole_line_k_rows():
    """128-byte K rows (KC = 8) are taken by 1x1 reductions of at least kc8_min_k (VINCE_KNOBS) elements (2048 by default): re-run
    with the threshold at 64 so that every 1x1 test shape goes through them, forward and input gradient."""
    _rerun_conv_tests({"VINCE_KNOBS": "kc8_min_k=64"}, "test_conv_fwd_stats or test_conv_dgrad_wgrad or test_linear_fwd_bwd")


@pytest.mark.parametrize("rows,cin,cout", [(37, 512, 64), (256, 2048, 2048), (256, 2048, 128), (64, 1000, 96)])
def test_linear_fwd_bwd(rows, cin, cout):
    # (256, 2048, *) are the ResNet-50 projection-MLP shapes: the launcher takes its split-K route for them
    ops = _ops()
    x = rnd(rows, cin, seed=9).requires_grad_(True)
    w = rnd(cout, cin, seed=10, scale=cin ** -0.5).requires_grad_(True)
    b = rnd(cout, seed=11).requires_grad_(True)
    y = F.relu(F.linear(x, w, b))
    dy = rnd(rows, cout, seed=12)
    y.backward(dy)
    xg, wg, bg = x.detach().to(DEV), w.detach().to(DEV), b.detach().to(DEV)
    out = ops.linear_fwd(xg, wg, bg, relu=True)
    assert_close(out, y.detach(), torch.float32, what="linear fwd")
    dpre = ops.relu_bwd(dy.to(DEV), out)
    dwg, dbg = torch.zeros_like(wg), torch.zeros_like(bg)
    dx = ops.linear_bwd(xg, wg.t().contiguous(), dpre, dwg, dbg)
    assert_close(dx, x.grad, torch.float32, what="linear dx")
    assert_close(dwg, w.grad, torch.float32, f32=1e-4, what="linear dw")
    assert_close(dbg, b.grad, torch.float32, f32=1e-4, what="linear db")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,shape", [(64, (4, 9, 9)), (256, (2, 5, 5)), (2048, (3, 2, 2))])
def test_bn_train_fwd_bwd(C, shape, dtype):
    ops = _ops()
    N, H, W = shape
    y = q(rnd(N, C, H, W, seed=13) * 1.5 + 0.3, dtype).requires_grad_(True)
    idn = q(rnd(N, C, H, W, seed=14), dtype).requires_grad_(True)
    gamma = (1 + 0.1 * rnd(C, seed=15)).requires_grad_(True)
    beta = (0.1 * rnd(C, seed=16)).requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    z = F.relu(F.batch_norm(y, rm, rv, gamma, beta, True, 0.1, 1e-5) + idn)
    dz = q(rnd(N, C, H, W, seed=17), dtype)
    z.backward(dz)
    # GPU: statistics come from the conv epilogue in production; here from the values directly
    yg = to_nhwc(y.detach(), dtype)
    yy = yg.float().reshape(-1, C).double()
    stats = torch.zeros(ops.STATS_REPLICAS, C, 2, device=DEV, dtype=torch.float64)
    stats[3] = torch.stack([yy.sum(0), (yy * yy).sum(0)], 1)
    rmg, rvg = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    consts = ops.bn_finalize(stats, N * H * W, gamma.detach().to(DEV), beta.detach().to(DEV), rmg, rvg, nbt, True)
    np.testing.assert_allclose(rmg.cpu().numpy(), rm.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rvg.cpu().numpy(), rv.numpy(), rtol=1e-4, atol=1e-5)
    assert int(nbt) == 1
    zg = ops.bn_apply(yg, consts[0], consts[1], identity=to_nhwc(idn.detach(), dtype), relu=True)
    assert_close(from_nhwc(zg), z.detach(), dtype, what="bn apply")
    dgam, dbet = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dy, g = ops.bn_bwd(to_nhwc(dz, dtype), zg, yg, consts[2], consts[3], gamma.detach().to(DEV), dgam, dbet, want_g=True)
    assert_close(from_nhwc(dy), y.grad, dtype, f32=1e-4, bf16=3e-2, what="bn dy")
    assert_close(from_nhwc(g), idn.grad, dtype, what="residual g")
    assert_close(dgam, gamma.grad, dtype, f32=1e-4, bf16=3e-2, what="dgamma")
    assert_close(dbet, beta.grad, dtype, f32=1e-4, bf16=3e-2, what="dbeta")
    # the production path reads the ReLU mask as 1 byte per 16-byte chunk written by bn_apply: identical results
    zg2, bits = ops.bn_apply(yg, consts[0], consts[1], identity=to_nhwc(idn.detach(), dtype), relu=True, want_mask=True)
    assert torch.equal(zg2, zg)
    dgam2, dbet2 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dy2, g2 = ops.bn_bwd(to_nhwc(dz, dtype), None, yg, consts[2], consts[3], gamma.detach().to(DEV), dgam2, dbet2, want_g=True,
                         mask_bits=bits)
    assert torch.equal(dy2, dy) and torch.equal(g2, g)
    torch.testing.assert_close(dgam2, dgam, rtol=1e-6, atol=1e-6)
    # finalize fused into the apply launch (vince_bn_train_apply): same outputs, constants, running statistics; only the
    # replicas it is told to fold are read (the statistics sit in replica 3 -> needs replicas >= 4)
    rm2, rv2 = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    nbt2 = torch.zeros((), dtype=torch.int64, device=DEV)
    zg3, bits3, sc3, sh3, mean3, inv3 = ops.bn_train_apply(yg, stats, N * H * W, gamma.detach().to(DEV), beta.detach().to(DEV),
                                                            rm2, rv2, nbt2, identity=to_nhwc(idn.detach(), dtype),
                                                            want_mask=True, replicas=4)
    assert torch.equal(zg3, zg) and torch.equal(bits3, bits)
    for got, want in zip((sc3, sh3, mean3, inv3), consts):
        torch.testing.assert_close(got, want, rtol=0, atol=0)
    torch.testing.assert_close(rm2, rmg, rtol=0, atol=0)
    torch.testing.assert_close(rv2, rvg, rtol=0, atol=0)
    assert int(nbt2) == 1
    # replica-limited backward: sums produced with 2 replicas, folded with 2
    dgam4, dbet4 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dy4, _ = ops.bn_bwd(to_nhwc(dz, dtype), None, yg, consts[2], consts[3], gamma.detach().to(DEV), dgam4, dbet4,
                        mask_bits=bits, replicas=2)
    assert float((dy4.float() - dy.float()).abs().max()) <= 1e-6 * (1 + float(dy.float().abs().max()))
    torch.testing.assert_close(dgam4, dgam, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_bn_bwd_mask_recomputed_from_y(dtype):
    """Plain BN+ReLU (no residual): the mask is the sign of y*scale+shift, recomputed from the conv output."""
    ops = _ops()
    N, C, H, W = 3, 128, 6, 5
    y = q(rnd(N, C, H, W, seed=60) * 1.3 - 0.2, dtype).requires_grad_(True)
    gamma = (1 + 0.2 * rnd(C, seed=61)).requires_grad_(True)
    beta = (0.2 * rnd(C, seed=62)).requires_grad_(True)
    a = F.relu(F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5))
    dz = q(rnd(N, C, H, W, seed=63), dtype)
    a.backward(dz)
    yg = to_nhwc(y.detach(), dtype)
    yy = yg.float().reshape(-1, C).double()
    stats = torch.zeros(ops.STATS_REPLICAS, C, 2, device=DEV, dtype=torch.float64)
    stats[0] = torch.stack([yy.sum(0), (yy * yy).sum(0)], 1)
    consts = ops.bn_finalize(stats, N * H * W, gamma.detach().to(DEV), beta.detach().to(DEV), None, None, None, True)
    dgam, dbet = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dy, _ = ops.bn_bwd(to_nhwc(dz, dtype), None, yg, consts[2], consts[3], gamma.detach().to(DEV), dgam, dbet,
                       mask_scale=consts[0], mask_shift=consts[1])
    assert_close(from_nhwc(dy), y.grad, dtype, f32=1e-4, bf16=3e-2, what="bn dy (mask from y)")
    assert_close(dgam, gamma.grad, dtype, f32=1e-4, bf16=3e-2, what="dgamma")


def test_bn_eval_and_downsample_identity():
    ops = _ops()
    C = 128
    y, yd = rnd(2, C, 4, 4, seed=18), rnd(2, C, 4, 4, seed=19)
    g1, b1, g2, b2 = 1 + 0.1 * rnd(C, seed=20), 0.1 * rnd(C, seed=21), 1 + 0.1 * rnd(C, seed=22), 0.1 * rnd(C, seed=23)
    rm1, rv1, rm2, rv2 = 0.2 * rnd(C, seed=24), 1 + 0.1 * rnd(C, seed=25).abs(), 0.2 * rnd(C, seed=26), 1 + 0.1 * rnd(C, seed=27).abs()
    ref = F.relu(F.batch_norm(y, rm1, rv1, g1, b1, False) + F.batch_norm(yd, rm2, rv2, g2, b2, False))
    c1 = ops.bn_finalize(None, 0, g1.to(DEV), b1.to(DEV), rm1.to(DEV), rv1.to(DEV), None, False)
    c2 = ops.bn_finalize(None, 0, g2.to(DEV), b2.to(DEV), rm2.to(DEV), rv2.to(DEV), None, False)
    out = ops.bn_apply(to_nhwc(y, torch.float32), c1[0], c1[1], identity=to_nhwc(yd, torch.float32), id_scale=c2[0],
                       id_shift=c2[1], relu=True)
    assert_close(from_nhwc(out), ref, torch.float32, what="eval bn + downsample identity")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hw", [(16, 16), (19, 13)])
def test_stem_pool(dtype, hw):
    ops = _ops()
    N, C = 2, 64
    H, W = hw
    y = q(rnd(N, C, H, W, seed=28), dtype).requires_grad_(True)
    scale, shift = 1 + 0.2 * rnd(C, seed=29), 0.3 * rnd(C, seed=30)
    a = F.relu(y * scale[None, :, None, None] + shift[None, :, None, None])
    pooled = F.max_pool2d(a, 3, 2, 1)
    dp = q(rnd(*pooled.shape, seed=31), dtype)
    a.retain_grad()
    pooled.backward(dp)
    out, amax = ops.stem_pool_fwd(to_nhwc(y.detach(), dtype), scale.to(DEV), shift.to(DEV))
    assert_close(from_nhwc(out), pooled.detach(), dtype, bf16=1e-2, what="stem pool fwd")
    g = ops.stem_pool_bwd(to_nhwc(dp, dtype), amax, H, W)
    # reference gradient wrt the BN output (after the ReLU mask) = a.grad * (a > 0)
    ref_g = a.grad * (a.detach() > 0)
    if dtype == torch.float32:
        assert_close(from_nhwc(g), ref_g, dtype, what="stem pool bwd")
    else:  # bf16 rounding of the pre-pool activation can move an argmax between tied neighbours: compare totals
        assert abs(float(g.float().sum()) - float(ref_g.sum())) < 2e-2 * float(ref_g.abs().sum())
    # fused stem backward (pool gather inside the bn1 backward) == stem_pool_bwd followed by the BatchNorm backward
    yg = to_nhwc(y.detach(), dtype)
    mean, invstd = (0.1 * rnd(C, seed=41)).to(DEV), (rnd(C, seed=42).abs() + 0.5).to(DEV)
    gamma = (1 + 0.2 * rnd(C, seed=43)).to(DEV)
    dg1, db1, dg2, db2 = (torch.zeros(C, device=DEV) for _ in range(4))
    want, _ = ops.bn_bwd(g, None, yg, mean, invstd, gamma, dg1, db1)
    got = ops.stem_bwd(to_nhwc(dp, dtype), amax, yg, mean, invstd, gamma, dg2, db2)
    tol_ = 1e-5 if dtype == torch.float32 else 2e-2
    assert float((got.float() - want.float()).abs().max()) <= tol_ * (float(want.float().abs().max()) + 1e-6)
    torch.testing.assert_close(dg2, dg1, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db2, db1, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dtype", DTYPES)
def test_avgpool(dtype):
    ops = _ops()
    x = q(rnd(5, 512, 3, 2, seed=32), dtype)
    out = ops.avgpool_fwd(to_nhwc(x, dtype))
    assert_close(out, x.mean(dim=(2, 3)), dtype, f32=1e-6, bf16=1e-6, what="avgpool")
    dout = rnd(5, 512, seed=33)
    dx = ops.avgpool_bwd(dout.to(DEV), 3, 2, dtype)
    assert_close(from_nhwc(dx), (dout / 6)[:, :, None, None].expand(5, 512, 3, 2), dtype, bf16=5e-3, what="avgpool bwd")


def test_l2norm():
    ops = _ops()
    x = rnd(33, 128, seed=34).requires_grad_(True)
    y = F.normalize(x, dim=1)
    dy = rnd(33, 128, seed=35)
    y.backward(dy)
    out, norms = ops.l2norm_fwd(x.detach().to(DEV))
    assert_close(out, y.detach(), torch.float32, f32=1e-6, what="l2norm")
    dx = ops.l2norm_bwd(x.detach().to(DEV), norms, dy.to(DEV))
    assert_close(dx, x.grad, torch.float32, f32=1e-5, what="l2norm bwd")


def test_jigsaw_and_layouts():
    ops = _ops()
    for hw in [66, 64]:
        x = torch.arange(2 * 3 * hw * hw, dtype=torch.float32).reshape(2, 3, hw, hw) / 100.0
        ref = vo.jigsaw_tile(x)
        out = ops.jigsaw_nchw_to_nhwc(x.to(DEV), torch.float32)
        assert list(out.shape[:3]) == [18, ref.shape[2], ref.shape[3]]
        torch.testing.assert_close(from_nhwc(out)[:, :3], ref, rtol=0, atol=0)
        for dt in DTYPES:   # the packed stem layout of the same tiles: [9N][th][Wp][4], zero margins
            rows = ops.jigsaw_nchw_to_rows(x.to(DEV), dt)
            tw = ref.shape[3]
            assert rows.shape == (18, ref.shape[2], ops.stem_row_width(tw), 4)
            torch.testing.assert_close(rows[:, :, 3:3 + tw, :3].float().cpu().permute(0, 3, 1, 2), q(ref, dt), rtol=0, atol=0)
            assert float(rows[:, :, :3].abs().max()) == 0.0 and float(rows[:, :, 3 + tw:].abs().max()) == 0.0
            assert float(rows[..., 3].abs().max()) == 0.0
    x = rnd(3, 3, 10, 12, seed=36)
    perm = torch.tensor([2, 0, 1])
    out = ops.input_nchw_to_nhwc(x.to(DEV), torch.float32, perm=perm.to(DEV))
    torch.testing.assert_close(from_nhwc(out)[:, :3], x[perm], rtol=0, atol=0)
    y = rnd(2, 5, 6, 16, seed=37)
    torch.testing.assert_close(ops.nhwc_to_nchw_f32(y.to(DEV)).cpu(), y.permute(0, 3, 1, 2).contiguous(), rtol=0, atol=0)


# ---------------------------------------------------------------------------------------------- InfoNCE
def unit_rows(n, d, seed):
    return F.normalize(rnd(n, d, seed=seed), dim=1)


@pytest.mark.parametrize("B,K,D", [(8, 64, 64), (32, 512, 128), (256, 4096, 64), (64, 1000, 128)])
@pytest.mark.parametrize("mode", ["moco", "inter1", "inter4"])
@pytest.mark.parametrize("T", [0.07, 0.2])
def test_infonce_vs_oracle(B, K, D, mode, T):
    ops = _ops()
    frames = 4 if mode == "inter4" else 1
    inter = mode != "moco"
    qq = unit_rows(B, D, 40).requires_grad_(True)
    kk = F.normalize(qq.detach() + 0.5 * unit_rows(B, D, 41), dim=1)
    queue = unit_rows(K, D, 42)
    sims, mask = vo.similarities(qq, kk, queue, inter, frames)
    ld = vo.similarity_cross_entropy(sims, T, mask)
    met = vo.nce_metrics(sims.detach(), mask, ld["softmax_weight"])
    (ld["dist"] * 1.7).backward()
    qg, kg, queueg = qq.detach().to(DEV), kk.to(DEV), queue.to(DEV)
    r = ops.infonce_fwd(qg, kg, queueg, T, frames=frames, offdiag_neg=inter)
    sc = r.scalars.cpu().numpy()
    np.testing.assert_allclose(sc[0], float(ld["dist"]), rtol=1e-4)
    np.testing.assert_allclose(sc[1], float(ld["softmax_weight"]), rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(sc[2], float(met["nce_accuracy_mean"]), atol=1e-6)
    np.testing.assert_allclose(sc[3], float(met["cosine_sim"]), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(sc[4], float(met["cosine_sim_neg_max"]), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(r.dists.cpu().numpy(), ld["dists"].detach().numpy().reshape(B, frames), rtol=1e-3, atol=1e-4)
    dq = torch.zeros(B, D, device=DEV)
    ops.infonce_bwd(r, qg, kg, queueg, torch.tensor([1.7], device=DEV), dq)
    assert_close(dq, qq.grad, torch.float32, f32=2e-4, what="infonce dq")
    # the training route: the forward stores its logits (split-half products: fp32-grade), backward reads them back
    r2 = ops.infonce_fwd(qg, kg, queueg, T, frames=frames, offdiag_neg=inter, save_logits=True)
    assert torch.equal(r2.scalars, r.scalars) and torch.equal(r2.dists, r.dists)
    want = torch.cat([qq.detach() @ kk.t(), qq.detach() @ queue.t()], dim=1).double()
    assert r2.logits.shape == want.shape
    assert float((r2.logits.cpu().double() - want).abs().max()) < 1e-6, "stored logits"
    dq2 = torch.zeros(B, D, device=DEV)
    ops.infonce_bwd(r2, qg, kg, queueg, torch.tensor([1.7], device=DEV), dq2)
    assert_close(dq2, qq.grad, torch.float32, f32=2e-4, what="infonce dq from stored logits")


def test_infonce_self_similarity():
    ops = _ops()
    B, D, T = 32, 64, 0.03
    qq = unit_rows(B, D, 43).requires_grad_(True)
    ssims = qq @ qq.t()
    mask = vo.positive_mask(B, 4, B)
    sl = vo.similarity_cross_entropy(ssims, T, mask)
    sl["dist"].backward()
    qg = qq.detach().to(DEV)
    r = ops.infonce_fwd(qg, qg, None, T, frames=4, offdiag_neg=True)
    np.testing.assert_allclose(float(r.scalars[0]), float(sl["dist"]), rtol=1e-4)
    dq = torch.zeros(B, D, device=DEV)
    wmat = ops.infonce_bwd(r, qg, qg, None, torch.ones(1, device=DEV), dq, want_wmat=True)
    # column-side gradient: dq_j += sum_i w_ij q_i  == wgrad form (rows = i as the reduced axis)
    ops.conv_wgrad(ops.linear_desc(B, D, B), qg, wmat, dq)
    assert_close(dq, qq.grad, torch.float32, f32=2e-4, what="self-sim dq")


# ---------------------------------------------------------------------------------------------- queue / EMA / SGD
@pytest.mark.parametrize("name", ["k512", "k96"])
def test_queue_indices_bit_exact_vs_golden(name, golden_dir):
    import os
    ops = _ops()
    g = np.load(os.path.join(golden_dir, "g1_queue.npz"))
    K = int(g[name + "_K"])
    queue = torch.full((K, 4), -1.0, device=DEV)
    tail, full, nid = 0, False, 0
    for step, n in enumerate(g[name + "_sizes"]):
        n = int(n)
        items = (nid + torch.arange(n, dtype=torch.float32))[:, None].repeat(1, 4).to(DEV)
        tail, full = ops.queue_enqueue(queue, items, tail, full)
        nid += n
        assert tail == int(g[name + "_tails"][step]) and full == bool(g[name + "_fulls"][step])
        np.testing.assert_array_equal(queue[:, 0].cpu().numpy().astype(np.int64), g[name + "_owners"][step])
        np.testing.assert_array_equal(queue[:, 3].cpu().numpy().astype(np.int64), g[name + "_owners"][step])


def test_ema_sgd_flat():
    ops = _ops()
    n = 100003
    k, qv, g = rnd(n, seed=50), rnd(n, seed=51), rnd(n, seed=52)
    kd, qd = {"p": k.clone()}, {"p": qv.clone()}
    vo.ema_update(kd, qd, ["p"], 0.999)
    kg = torch.zeros(n + 1, device=DEV)[:n]
    kg.copy_(k)
    ops.ema_flat(kg, qv.to(DEV), 0.999)
    np.testing.assert_allclose(kg.cpu().numpy(), kd["p"].numpy(), rtol=1e-6, atol=1e-7)
    # momentum 0 -> exact copy (param_update(model, 0), vince_solver.py:296,318)
    ops.ema_flat(kg, qv.to(DEV), 0.0)
    torch.testing.assert_close(kg.cpu(), qv, rtol=0, atol=0)
    p = {"p": qv.clone()}
    bufs = {}
    pg, bg = qv.clone().to(DEV), torch.zeros(n, device=DEV)
    for it in range(3):
        gi = g * (it + 1)
        vo.sgd_step(p, {"p": gi}, bufs, 0.03)
        ops.sgd_flat(pg, gi.to(DEV), bg, 0.03)
    np.testing.assert_allclose(pg.cpu().numpy(), p["p"].numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bg.cpu().numpy(), bufs["p"].numpy(), rtol=1e-5, atol=1e-6)


def test_cpu_tensor_is_rejected():
    ops = _ops()
    with pytest.raises(RuntimeError):
        ops.l2norm_fwd(torch.randn(4, 64))


# ---------------------------------------------------------------------------------------------- Gram-statistics residual join
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,hw,K,Co,ds", [(4, 14, 64, 256, False), (3, 9, 128, 512, True), (2, 5, 64, 256, True), (3, 14, 256, 1024, False)])
def test_gram_statistics_join_vs_separate_passes(dtype, N, hw, K, Co, ds):
    """bn3(conv3(a)) + identity + ReLU (resnet.py:125-133) three ways: torch fp32/fp64 on the CPU, the engine's separate
    passes, and the Gram route -- column sums from the pass that writes a, Gram matrix through the weight-gradient kernel,
    vince_bn_gram_finalize, conv3 with the join in its epilogue (in place on the identity)."""
    ops = _ops()
    from vince_amd._lib import EPI_ACCUMULATE, EPI_RELU
    rows = N * hw * hw
    y2 = rnd(rows, K, seed=1) * (0.5 + torch.rand(K, generator=torch.Generator().manual_seed(2))) + rnd(K, seed=3) * 0.5
    g2, b2 = torch.rand(K, generator=torch.Generator().manual_seed(4)) + 0.5, rnd(K, seed=5) * 0.3
    w3 = rnd(Co, K, seed=6) * (2.0 / Co) ** 0.5
    g3, b3 = torch.rand(Co, generator=torch.Generator().manual_seed(7)) + 0.5, rnd(Co, seed=8) * 0.3
    idn = rnd(rows, Co, seed=9)
    isc, ish = torch.rand(Co, generator=torch.Generator().manual_seed(10)) + 0.5, rnd(Co, seed=11) * 0.2
    # ---- CPU reference in fp64 on the dtype-rounded operands
    y2q, w3q, idq = q(y2, dtype).double(), q(w3, dtype).double(), q(idn, dtype).double()
    m2, v2 = y2q.mean(0), y2q.var(0, unbiased=False)
    a = q(torch.relu((y2q - m2) / torch.sqrt(v2 + 1e-5) * g2.double() + b2.double()).float(), dtype).double()
    y3 = a @ w3q.t()
    m3, v3 = y3.mean(0), y3.var(0, unbiased=False)
    ident = idq * isc.double() + ish.double() if ds else idq
    want = torch.relu((y3 - m3) / torch.sqrt(v3 + 1e-5) * g3.double() + b3.double() + ident)
    # ---- GPU: bn2 apply with column sums
    y2g, w3g = y2.to(DEV).to(dtype), w3.to(DEV).to(dtype).contiguous()
    st2 = torch.stack([y2g.double().sum(0), (y2g.double() ** 2).sum(0)], 1)[None].contiguous()     # double[1][K][2]
    colsum = torch.zeros(4, K, device=DEV, dtype=torch.float64)
    ag, _, _, _, _, _ = ops.bn_train_apply(y2g, st2, rows, g2.to(DEV), b2.to(DEV), replicas=1, out_sum=colsum)
    np.testing.assert_allclose(colsum.sum(0).cpu().numpy(), ag.double().sum(0).cpu().numpy(), rtol=1e-6, atol=1e-6)
    assert_close(ag, a.float(), dtype, f32=2e-5, bf16=2e-2, what="a")
    gram = torch.zeros(K, 1, K, device=DEV)
    x4 = ag.view(N, hw, hw, K)
    ops.conv_wgrad(ops.conv_desc(N, hw, hw, K, K, 1, 1, 0), x4, x4, gram)
    rm, rv = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
    nbt = torch.zeros(1, device=DEV, dtype=torch.int64)
    consts = ops.bn_gram_finalize(gram.view(K, K), colsum, rows, w3g, g3.to(DEV), b3.to(DEV), rm, rv, nbt)
    # statistics of the UNROUNDED conv output of the GPU's own a
    y3g = ag.double() @ w3g.double().t()
    mg, vg = y3g.mean(0), y3g.var(0, unbiased=False)
    np.testing.assert_allclose(consts[2].cpu().numpy(), mg.cpu().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(consts[3].cpu().numpy(), (1.0 / torch.sqrt(vg + 1e-5)).cpu().numpy(), rtol=2e-5)
    np.testing.assert_allclose(rm.cpu().numpy(), 0.1 * mg.cpu().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(rv.cpu().numpy(), (0.9 + 0.1 * vg * rows / (rows - 1)).cpu().numpy(), rtol=2e-5)
    assert int(nbt) == 1
    # ---- conv3 with the join in its epilogue, in place on the identity
    z = idn.to(DEV).to(dtype).view(N, hw, hw, Co).contiguous()
    ops.conv_igemm(ops.conv_desc(N, hw, hw, K, Co, 1, 1, 0), x4, w3g.view(Co, 1, K), z, bias=consts[1], flags=EPI_ACCUMULATE | EPI_RELU,
                   out_scale=consts[0], id_scale=isc.to(DEV) if ds else None, id_shift=ish.to(DEV) if ds else None)
    assert_close(z.view(rows, Co), want.float(), dtype, f32=5e-5, bf16=3e-2, what="join")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("K,Co", [(64, 256), (96, 200), (128, 512), (256, 1024), (512, 64)])
def test_bn_gram_finalize_alone_vs_fp64(dtype, K, Co):
    """vince_bn_gram_finalize on a given Gram matrix: batch statistics of y = W a from G = a^T a and the column sums, against fp64 on
    the CPU -- the bottleneck shapes (K = 64 / 128 / 256: one column of G per thread) and other K (one wavefront per channel)."""
    ops = _ops()
    rows = 5000
    a = q(torch.relu(rnd(rows, K, seed=21) + 0.3), dtype).double()
    w = q(rnd(Co, K, seed=22) * (2.0 / K) ** 0.5, dtype)
    gram = (a.t() @ a).float().to(DEV).contiguous()
    colsum = torch.zeros(4, K, dtype=torch.float64)
    colsum[1] = a.sum(0)
    y = a @ w.double().t()
    rm, rv = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
    consts = ops.bn_gram_finalize(gram, colsum.to(DEV), rows, w.to(DEV).to(dtype).contiguous(), torch.ones(Co, device=DEV),
                                  torch.zeros(Co, device=DEV), rm, rv, None)
    np.testing.assert_allclose(consts[2].cpu().numpy(), y.mean(0).numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(consts[3].cpu().numpy(), (1.0 / torch.sqrt(y.var(0, unbiased=False) + 1e-5)).numpy(), rtol=2e-5)
    np.testing.assert_allclose(rv.cpu().numpy(), (0.9 + 0.1 * y.var(0, unbiased=True)).numpy(), rtol=2e-5)


@pytest.mark.parametrize("rows,K", [(4 * 14 * 14, 64), (3 * 9 * 9, 128), (64 * 37 + 5, 64), (64 * 700 + 17, 128), (33, 128), (200704, 128)])
def test_bn_train_apply_gram_vs_separate_launches(rows, K):
    """vince_bn_train_apply_gram (csrc/bn_gram.hip): a = relu(bn2(y)) written once, with the column sums and the Gram matrix a^T a of
    exactly the stored bf16 values -- against vince_bn_train_apply (bit-equal output, constants, running statistics), fp64 sums of the
    stored tensor, and the Gram matrix the weight-gradient kernel computes from it; ragged row counts, one to 512 workgroups, and
    two runs that agree to the bit (slabs added in a fixed order)."""
    ops = _ops()
    y = (rnd(rows, K, seed=1) * (0.5 + torch.rand(K, generator=torch.Generator().manual_seed(2))) + rnd(K, seed=3) * 0.5).to(DEV).bfloat16()
    g, b = (torch.rand(K, generator=torch.Generator().manual_seed(4)) + 0.5).to(DEV), (rnd(K, seed=5) * 0.3).to(DEV)
    st = torch.stack([y.double().sum(0), (y.double() ** 2).sum(0)], 1)[None].contiguous()     # double[1][K][2]
    rm0, rv0 = torch.zeros(K, device=DEV), torch.ones(K, device=DEV)
    nbt0 = torch.zeros(1, device=DEV, dtype=torch.int64)
    cs0 = torch.zeros(4, K, device=DEV, dtype=torch.float64)
    want, _, sc0, sh0, mu0, is0 = ops.bn_train_apply(y, st, rows, g, b, rm0, rv0, nbt0, replicas=1, out_sum=cs0)
    outs = []
    for _ in range(2):
        rm, rv = torch.zeros(K, device=DEV), torch.ones(K, device=DEV)
        nbt = torch.zeros(1, device=DEV, dtype=torch.int64)
        cs = torch.zeros(4, K, device=DEV, dtype=torch.float64)
        gram = torch.zeros(K, K, device=DEV)
        a, sc, sh, mu, isd = ops.bn_train_apply_gram(y, st, rows, g, b, gram, cs, rm, rv, nbt, replicas=1)
        outs.append((a, gram, cs.sum(0)))
        assert torch.equal(a.view(torch.int16), want.view(torch.int16))
        for got, ref in ((sc, sc0), (sh, sh0), (mu, mu0), (isd, is0), (rm, rm0), (rv, rv0)):
            assert torch.equal(got, ref)
        assert int(nbt) == 1
        np.testing.assert_allclose(cs.sum(0).cpu().numpy(), a.double().sum(0).cpu().numpy(), rtol=1e-6, atol=1e-6)
        ref_gram = a.double().t() @ a.double()
        scale = float(ref_gram.abs().max())
        assert float((gram.double() - ref_gram).abs().max()) / scale < 2e-6, float((gram.double() - ref_gram).abs().max()) / scale
    assert torch.equal(outs[0][1], outs[1][1])
    # and next to the route it replaces: the weight-gradient kernel over the stored tensor
    if rows % (14 * 14) == 0:
        n = rows // 196
        old = torch.zeros(K, 1, K, device=DEV)
        x4 = outs[0][0].view(n, 14, 14, K)
        ops.conv_wgrad(ops.conv_desc(n, 14, 14, K, K, 1, 1, 0), x4, x4, old)
        assert float((old.view(K, K) - outs[0][1]).abs().max()) / float(outs[0][1].abs().max()) < 1e-5


@pytest.mark.parametrize("rows,K,Co,ds", [(4 * 14 * 14, 64, 256, False), (1000, 64, 256, True), (3 * 9 * 9, 128, 512, True),
                                          (128 * 40 + 5, 128, 512, False), (37, 64, 512, False)])
def test_conv_expand_join_streaming_kernel(rows, K, Co, ds):
    """vince_conv_expand_join (csrc/conv_xjoin.hip): relu(scale * (x W^T) + shift + identity') against fp64 on the same bf16
    operands, in place on the identity; ragged row counts, both K, one and two channel groups, identity read through the
    downsample BatchNorm's affine; and against vince_conv_igemm's join epilogue (bit-equal up to the last bf16 rounding)."""
    ops = _ops()
    from vince_amd._lib import EPI_ACCUMULATE, EPI_RELU
    x = rnd(rows, K, seed=1).clamp_(min=0)
    w = rnd(Co, K, seed=2) * (2.0 / Co) ** 0.5
    idn = rnd(rows, Co, seed=3)
    sc, sh = torch.rand(Co, generator=torch.Generator().manual_seed(4)) + 0.5, rnd(Co, seed=5) * 0.3
    isc, ish = torch.rand(Co, generator=torch.Generator().manual_seed(6)) + 0.5, rnd(Co, seed=7) * 0.2
    xq, wq, iq = q(x, torch.bfloat16).double(), q(w, torch.bfloat16).double(), q(idn, torch.bfloat16).double()
    ident = iq * isc.double() + ish.double() if ds else iq
    want = torch.relu((xq @ wq.t()) * sc.double() + sh.double() + ident)
    xg, wg = x.to(DEV).bfloat16(), w.to(DEV).bfloat16().contiguous()
    z = idn.to(DEV).bfloat16().contiguous()
    guard = torch.full((4096,), 7.0, device=DEV).bfloat16()          # (allocated right behind z most of the time: tail writes show)
    ops.conv_expand_join(xg, wg, sc.to(DEV), sh.to(DEV), z, id_scale=isc.to(DEV) if ds else None, id_shift=ish.to(DEV) if ds else None)
    assert_close(z, want.float(), torch.bfloat16, bf16=1e-2, what="expand join")
    assert float((guard.float() - 7.0).abs().max()) == 0.0
    z2 = idn.to(DEV).bfloat16().view(1, rows, 1, Co).contiguous()
    ops.conv_igemm(ops.conv_desc(1, rows, 1, K, Co, 1, 1, 0), xg.view(1, rows, 1, K), wg.view(Co, 1, K), z2, bias=sh.to(DEV),
                   flags=EPI_ACCUMULATE | EPI_RELU, out_scale=sc.to(DEV), id_scale=isc.to(DEV) if ds else None,
                   id_shift=ish.to(DEV) if ds else None)
    # the streaming kernel applies scale / shift to the fp32 accumulator, the igemm epilogue to its bf16-rounded copy
    assert_close(z, z2.view(rows, Co).float(), torch.bfloat16, bf16=1.5e-2, what="vs igemm join")
    # not in place
    out = torch.empty_like(z)
    ops.conv_expand_join(xg, wg, sc.to(DEV), sh.to(DEV), idn.to(DEV).bfloat16().contiguous(), out=out,
                         id_scale=isc.to(DEV) if ds else None, id_shift=ish.to(DEV) if ds else None)
    assert torch.equal(out, z)
    # the training forward's extra outputs: the raw convolution output (bf16) and the ReLU mask bytes of `out`
    yraw = torch.empty_like(z)
    mask = torch.zeros(rows * Co // 8, device=DEV, dtype=torch.uint8)
    out2 = torch.empty_like(z)
    ops.conv_expand_join(xg, wg, sc.to(DEV), sh.to(DEV), idn.to(DEV).bfloat16().contiguous(), out=out2, y_raw=yraw, mask_out=mask,
                         id_scale=isc.to(DEV) if ds else None, id_shift=ish.to(DEV) if ds else None)
    assert torch.equal(out2, z)
    assert_close(yraw, (xq @ wq.t()).float(), torch.bfloat16, bf16=1e-2, what="raw conv output")
    bits = ((mask.view(rows, Co // 8, 1).int() >> torch.arange(8, device=DEV).view(1, 1, 8)) & 1).view(rows, Co).bool()
    pre = (xq @ wq.t()) * sc.double() + sh.double() + ident                      # pre-ReLU value, fp64
    sure = pre.abs() > 2e-2 * pre.abs().max()                                    # away from the rounding band around zero
    assert torch.equal(bits.cpu()[sure], (pre > 0)[sure])
    assert torch.equal(bits, out2 > 0) or float((bits != (out2 > 0)).float().mean()) < 1e-3   # mask == (relu output > 0) up to -0 / tiny


@pytest.mark.parametrize("rows,ds,save,C2", [(4 * 56 * 56, False, 0, 64), (128 * 37 + 5, True, 0, 64), (3 * 56 * 56 + 17, False, 2, 64), (1000, True, 1, 64),
                                             (37, False, 2, 64), (4 * 56 * 56, False, 0, 128), (128 * 37 + 5, False, 1, 128), (3 * 56 * 56 + 17, False, 2, 128),
                                             (37, False, 0, 128)])
def test_conv_expand_join_next_block_conv1_fused(rows, ds, save, C2):
    """vince_conv_expand_join_next (csrc/conv_xjoin.hip, NEXT): the join of a layer1 bottleneck + the following block's conv1
    (resnet.py:117: 1x1, 256 -> 64; or layer2's first block behind layer1's last, 256 -> 128) on the block output while it is in LDS.  Everything the join alone writes is unchanged, bit for
    bit; y_next equals vince_conv_igemm's output for that layer on the stored block output, bit for bit (same MFMA steps in the same
    order); its BatchNorm statistics equal the sums of the stored values; ragged row counts, the affine identity, all three save modes."""
    ops = _ops()
    K, Co = 64, 256
    x = rnd(rows, K, seed=1).clamp_(min=0)
    w = rnd(Co, K, seed=2) * (2.0 / Co) ** 0.5
    w2 = rnd(C2, Co, seed=8) * (2.0 / C2) ** 0.5
    idn = rnd(rows, Co, seed=3)
    sc, sh = torch.rand(Co, generator=torch.Generator().manual_seed(4)) + 0.5, rnd(Co, seed=5) * 0.3
    isc, ish = torch.rand(Co, generator=torch.Generator().manual_seed(6)) + 0.5, rnd(Co, seed=7) * 0.2
    xg, wg, w2g = x.to(DEV).bfloat16(), w.to(DEV).bfloat16().contiguous(), w2.to(DEV).bfloat16().contiguous()
    kw = dict(id_scale=isc.to(DEV) if ds else None, id_shift=ish.to(DEV) if ds else None)

    def extras():
        yraw = torch.zeros(rows, Co, device=DEV).bfloat16() if save == 2 else None
        mask = torch.zeros(rows * Co // 8, device=DEV, dtype=torch.uint8) if save else None
        return yraw, mask

    # the two launches it replaces
    out_a = torch.empty(rows, Co, device=DEV).bfloat16()
    yraw_a, mask_a = extras()
    ops.conv_expand_join(xg, wg, sc.to(DEV), sh.to(DEV), idn.to(DEV).bfloat16().contiguous(), out=out_a, y_raw=yraw_a, mask_out=mask_a, **kw)
    y_a = torch.empty(1, rows, 1, C2, device=DEV).bfloat16()
    st_a = torch.zeros(ops.STATS_REPLICAS, C2, 2, device=DEV, dtype=torch.float64)
    ops.conv_igemm(ops.conv_desc(1, rows, 1, Co, C2, 1, 1, 0), out_a.view(1, rows, 1, Co), w2g.view(C2, 1, Co), y_a, stats=st_a)
    # the fused launch
    out_b = torch.empty(rows, Co, device=DEV).bfloat16()
    yraw_b, mask_b = extras()
    y_b = torch.full((rows + 64, C2), 7.0, device=DEV).bfloat16()          # 64 guard rows behind the tensor
    st_b = torch.zeros(ops.STATS_REPLICAS, C2, 2, device=DEV, dtype=torch.float64)
    ops.conv_expand_join_next(xg, wg, sc.to(DEV), sh.to(DEV), idn.to(DEV).bfloat16().contiguous(), w2g, y_b, out=out_b, y_raw=yraw_b,
                              mask_out=mask_b, stats_next=st_b, **kw)
    torch.cuda.synchronize()
    assert torch.equal(out_b, out_a)
    if save == 2:
        assert torch.equal(yraw_b, yraw_a)
    if save:
        assert torch.equal(mask_b, mask_a)
    assert float((y_b[rows:].float() - 7.0).abs().max()) == 0.0
    assert torch.equal(y_b[:rows], y_a.view(rows, C2))
    yf = y_b[:rows].double()
    got = st_b.sum(0)
    np.testing.assert_allclose(got[:, 0].cpu().numpy(), yf.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(got[:, 1].cpu().numpy(), (yf * yf).sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(got.cpu().numpy(), st_a.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3)
    # in place on the identity (the no-grad forward's arrangement)
    z = idn.to(DEV).bfloat16().contiguous()
    y_c = torch.empty(rows, C2, device=DEV).bfloat16()
    ops.conv_expand_join_next(xg, wg, sc.to(DEV), sh.to(DEV), z, w2g, y_c, **kw)
    assert torch.equal(z, out_a) and torch.equal(y_c, y_a.view(rows, C2))
    # the folded-inference epilogue: y_next = relu(conv + bias) with the implicit-GEMM kernel's roundings
    from vince_amd._lib import EPI_RELU
    b2 = (rnd(C2, seed=9) * 0.5).to(DEV)
    y_d = torch.empty(1, rows, 1, C2, device=DEV).bfloat16()
    ops.conv_igemm(ops.conv_desc(1, rows, 1, Co, C2, 1, 1, 0), out_a.view(1, rows, 1, Co), w2g.view(C2, 1, Co), y_d, bias=b2, flags=EPI_RELU)
    z = idn.to(DEV).bfloat16().contiguous()
    y_e = torch.empty(rows, C2, device=DEV).bfloat16()
    ops.conv_expand_join_next(xg, wg, sc.to(DEV), sh.to(DEV), z, w2g, y_e, bias_next=b2, relu_next=True, **kw)
    assert torch.equal(z, out_a) and torch.equal(y_e, y_d.view(rows, C2))


@pytest.mark.parametrize("C2", [64, 128])
def test_conv_expand_join_next_is_stable_beside_a_busy_stream(C2):
    """The fused join keeps a pixel half's stripes in a SHARED LDS buffer between four barriers per tile: a missing barrier would show as
    a run-to-run difference under load.  Twenty repetitions at layer1's size (64 frames) beside a stream that keeps the machine busy:
    outputs equal the separate launches bit for bit every time, the statistics are bitwise identical run to run
    (tools/stress_xjoin_next.py: 150 repetitions at 256 frames)."""
    ops = _ops()
    rows, K, Co = 64 * 56 * 56, 64, 256
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(rows, K, device=DEV, generator=g).clamp_(min=0).bfloat16()
    w = (torch.randn(Co, K, device=DEV, generator=g) * 0.1).bfloat16().contiguous()
    w2 = (torch.randn(C2, Co, device=DEV, generator=g) * 0.1).bfloat16().contiguous()
    idn = torch.randn(rows, Co, device=DEV, generator=g).bfloat16()
    sc, sh = torch.rand(Co, device=DEV, generator=g) + 0.5, torch.randn(Co, device=DEV, generator=g) * 0.3
    out_a = torch.empty(rows, Co, device=DEV).bfloat16()
    ops.conv_expand_join(x, w, sc, sh, idn, out=out_a)
    y_a = torch.empty(1, rows, 1, C2, device=DEV).bfloat16()
    ops.conv_igemm(ops.conv_desc(1, rows, 1, Co, C2, 1, 1, 0), out_a.view(1, rows, 1, Co), w2.view(C2, 1, Co), y_a)
    side, junk, first = torch.cuda.Stream(), torch.randn(32 * 1024 * 1024, device=DEV), None
    for it in range(20):
        with torch.cuda.stream(side):
            junk.mul_(1.0001)
        out_b = torch.empty(rows, Co, device=DEV).bfloat16()
        y_b = torch.empty(rows, C2, device=DEV).bfloat16()
        st_b = torch.zeros(ops.STATS_REPLICAS, C2, 2, device=DEV, dtype=torch.float64)
        mask = torch.zeros(rows * Co // 8, device=DEV, dtype=torch.uint8) if it % 2 else None
        ops.conv_expand_join_next(x, w, sc, sh, idn, w2, y_b, out=out_b, stats_next=st_b, mask_out=mask)
        assert torch.equal(out_b, out_a) and torch.equal(y_b, y_a.view(rows, C2)), it
        first = st_b.sum(0) if first is None else first
        assert torch.equal(st_b.sum(0), first), it
    torch.cuda.synchronize()


def test_similarity_cross_entropy_unequal_positives_use_float():
    """utils/loss_util.py:25-36,46-48 on the HIP row kernel, against the reference's own numbers (tests/golden/g2u_loss_unequal.npz)
    incl. the process-wide cached decision (App. D item 2) and the failure the reference has when an equal-count mask came
    first."""
    from vince_amd.utils import loss_util
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g2u_loss_unequal.npz"))
    sims, mask, eq = vo.g2u_inputs()
    loss_util.USE_FLOAT = None
    for tag, m in (("uneq", mask), ("eq_after", eq)):
        s = sims.clone().to(DEV).requires_grad_(True)
        r = loss_util.similarity_cross_entropy(s, 0.2, 6, 1, m.to(DEV))
        assert loss_util.USE_FLOAT is True
        r["dist"].backward()
        np.testing.assert_allclose(r["dists"].detach().cpu().numpy(), g[tag + "_dists"], rtol=2e-5, atol=1e-5)
        np.testing.assert_allclose(float(r["dist"]), float(g[tag + "_dist"]), rtol=1e-5)
        np.testing.assert_allclose(r["softmax_weights"].detach().cpu().numpy(), g[tag + "_softmax_weights"], rtol=2e-5, atol=1e-8)
        np.testing.assert_allclose(float(r["softmax_weight"]), float(g[tag + "_softmax_weight"]), rtol=1e-5)
        np.testing.assert_allclose(s.grad.cpu().numpy(), g[tag + "_dsims"], rtol=1e-4, atol=1e-7)
    loss_util.USE_FLOAT = None
    r = loss_util.similarity_cross_entropy(sims.to(DEV), 0.2, 6, 1, eq.to(DEV))          # equal counts first: compacted path, cached
    assert loss_util.USE_FLOAT is False and r["dists"].shape == (6, 1, 2)
    np.testing.assert_allclose(float(r["dist"]), float(g["eq_after_dist"]), rtol=1e-5)
    with pytest.raises(RuntimeError):
        loss_util.similarity_cross_entropy(sims.to(DEV), 0.2, 6, 1, mask.to(DEV))
    loss_util.USE_FLOAT = None


@pytest.mark.parametrize("N,H", [(1, 4), (3, 56), (5, 8), (2, 20)])
def test_conv3x3_image_strip_kernel(N, H):
    """vince_conv3x3_strip (layer1's 3x3: 64 -> 64 channels, 56 wide): output BIT-identical to vince_conv_igemm's on the same
    operands, statistics equal to the sums over the stored output, and both against torch on the CPU; also with a permuted tap
    map (what an input gradient would pass)."""
    ops = _ops()
    x = rnd(N, H, 56, 64, seed=41).clamp_(min=0).to(DEV).bfloat16()
    w = (rnd(64, 9, 64, seed=42) * (2.0 / 576) ** 0.5).to(DEV).bfloat16().contiguous()
    out = torch.full((N, H, 56, 64), 5.0, device=DEV).bfloat16()
    stats = torch.zeros(4, 64, 2, device=DEV, dtype=torch.float64)
    ops.conv3x3_strip(x, w, out, stats=stats, replicas=4)
    ref = torch.empty_like(out)
    ops.conv_igemm(ops.conv_desc(N, H, 56, 64, 64, 3, 1, 1), x, w, ref)
    assert torch.equal(out, ref)
    o = out.double().reshape(-1, 64)
    st = stats.sum(0)
    np.testing.assert_allclose(st[:, 0].cpu().numpy(), o.sum(0).cpu().numpy(), rtol=2e-6, atol=1e-3)
    np.testing.assert_allclose(st[:, 1].cpu().numpy(), (o * o).sum(0).cpu().numpy(), rtol=2e-6, atol=1e-3)
    want = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.float().cpu().view(64, 3, 3, 64).permute(0, 3, 1, 2), None, 1, 1)
    assert_close(out.permute(0, 3, 1, 2), want, torch.bfloat16, bf16=1e-2, what="strip conv vs torch")
    flip = list(range(8, -1, -1))                            # kernel rotated by 180 degrees
    ops.conv3x3_strip(x, w, out, tap_map=flip)
    wantf = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.float().cpu().view(64, 3, 3, 64).flip(1, 2).permute(0, 3, 1, 2), None, 1, 1)
    assert_close(out.permute(0, 3, 1, 2), wantf, torch.bfloat16, bf16=1e-2, what="strip conv, flipped taps")


@pytest.mark.parametrize("N,H", [(3, 56), (5, 8), (1, 4)])
def test_conv3x3_image_strip_kernel_bias_relu_epilogue(N, H):
    """vince_conv3x3_strip_bias -- layer1's 3x3 in the BatchNorm-folded inference forward: out = relu(conv + bias) -- BIT-identical to
    vince_conv_igemm's bias + ReLU epilogue on the same operands, and against torch on the CPU; without ReLU and without a bias too."""
    ops = _ops()
    from vince_amd._lib import EPI_RELU
    x = rnd(N, H, 56, 64, seed=43).clamp_(min=0).to(DEV).bfloat16()
    w = (rnd(64, 9, 64, seed=44) * (2.0 / 576) ** 0.5).to(DEV).bfloat16().contiguous()
    b = (rnd(64, seed=45) * 0.3).to(DEV)
    d = ops.conv_desc(N, H, 56, 64, 64, 3, 1, 1)
    for bias, relu in ((b, True), (b, False), (None, True)):
        out = torch.full((N, H, 56, 64), 5.0, device=DEV).bfloat16()
        ops.conv3x3_strip_bias(x, w, out, bias=bias, relu=relu)
        ref = torch.empty_like(out)
        ops.conv_igemm(d, x, w, ref, bias=bias, flags=EPI_RELU if relu else 0)
        assert torch.equal(out, ref), (bias is not None, relu)
    want = torch.relu(F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.float().cpu().view(64, 3, 3, 64).permute(0, 3, 1, 2), b.cpu(), 1, 1))
    ops.conv3x3_strip_bias(x, w, out, bias=b, relu=True)
    assert_close(out.permute(0, 3, 1, 2), want, torch.bfloat16, bf16=1e-2, what="strip conv + bias + relu vs torch")


@pytest.mark.parametrize("N,H", [(3, 56), (5, 8)])
def test_conv3x3_image_strip_kernel_input_gradient_with_fused_bn_reduction(N, H):
    """vince_conv3x3_strip_dgrad (layer1's 3x3 input gradient through the image-strip kernel, taps flipped over the [Ci][tap][Co] weight
    copy): dx equal to vince_conv_igemm's input-gradient launch up to summation order, the fused BatchNorm-backward sums (sum g, sum g * xhat of the
    stored gradient, gated by the sign of y * scale + shift) equal to that launch's epilogue and to the stand-alone reduction, and dx
    against torch autograd on the CPU."""
    ops = _ops()
    x = rnd(N, 64, H, 56, seed=51).requires_grad_(True)
    w = (rnd(64, 64, 3, 3, seed=52) * (2.0 / 576) ** 0.5).bfloat16().float().requires_grad_(True)
    dy = (rnd(N, 64, H, 56, seed=53) * 0.1).bfloat16().float()
    F.conv2d(x, w, None, 1, 1).backward(dy)
    _, wt = weights_krsc(w.detach(), torch.bfloat16)
    dyg = to_nhwc(dy, torch.bfloat16)
    yb = to_nhwc(rnd(N, 64, H, 56, seed=54), torch.bfloat16)
    mean, invstd = rnd(64, seed=55).to(DEV), (rnd(64, seed=56).abs() + 0.5).to(DEV)
    msc, msh = rnd(64, seed=57).to(DEV), rnd(64, seed=58, scale=0.3).to(DEV)
    res = {}
    for which in ("strip", "igemm"):
        sums = torch.zeros(ops.STATS_REPLICAS, 64, 2, device=DEV, dtype=torch.float64)
        dx = torch.full((N, H, 56, 64), float("nan"), device=DEV, dtype=torch.bfloat16)
        br = ops.bn_reduce_arg(yb, mean, invstd, sums, mask_scale=msc, mask_shift=msh)
        if which == "strip":
            ops.conv3x3_strip_dgrad(dyg, wt, dx, bnred=br, replicas=4)
        else:
            ops.conv_igemm(ops.dgrad_descs(N, H, 56, 64, 64, 3, 1, 1)[0], dyg, wt, dx, bnred=br, replicas=4)
        res[which] = (dx, sums.sum(0))
    # (not bit-identical: the strip kernel walks the kernel positions in raster order, the implicit GEMM its own tap enumeration -- the
    # fp32 sums differ in their last bits and a few stored values by one bf16 rounding)
    assert_close(res["strip"][0], res["igemm"][0].float(), torch.bfloat16, bf16=4e-3, what="strip vs implicit-GEMM input gradient")
    want = ops.bn_bwd_reduce(res["strip"][0], yb, mean, invstd, mask_scale=msc, mask_shift=msh)
    scale = float(want.abs().max()) + 1e-6
    assert float((res["strip"][1] - want).abs().max()) / scale < 1e-5          # the sums of what THIS launch stored
    assert float((res["strip"][1] - res["igemm"][1]).abs().max()) / scale < 2e-3
    assert_close(from_nhwc(res["strip"][0]), x.grad, torch.bfloat16, bf16=1e-2, what="strip input gradient vs autograd")
    plain = torch.empty_like(res["strip"][0])
    ops.conv3x3_strip_dgrad(dyg, wt, plain)                 # without the reduction
    assert torch.equal(plain, res["strip"][0])


def test_conv3x3_image_strip_kernel_several_images_per_workgroup():
    """The ring restarts per image: with fewer workgroups than images (VINCE_KNOBS strip_grid, read once per process) every workgroup
    walks several images -- re-run the parity test above in a child process with 2 workgroups."""
    _rerun_conv_tests({"VINCE_KNOBS": "strip_grid=2"}, "test_conv3x3_image_strip_kernel and not several")


@pytest.mark.parametrize("rows,K,Co", [(4 * 14 * 14, 64, 256), (128 * 9 + 77, 64, 512), (3 * 28 * 28, 128, 512), (50, 128, 256)])
def test_conv_expand_stats_streaming_kernel(rows, K, Co):
    """vince_conv_expand_stats: the expand convolution on its own through the streaming kernel -- output equal to
    vince_conv_igemm's, statistics equal to the sums over the stored output (what the BatchNorm finalize consumes)."""
    ops = _ops()
    x = rnd(rows, K, seed=11).clamp_(min=0).to(DEV).bfloat16()
    w = (rnd(Co, K, seed=12) * (2.0 / Co) ** 0.5).to(DEV).bfloat16().contiguous()
    out = torch.full((rows, Co), 3.0, device=DEV).bfloat16()
    stats = torch.zeros(4, Co, 2, device=DEV, dtype=torch.float64)
    ops.conv_expand_stats(x, w, out, stats=stats, replicas=4)
    ref = torch.empty(1, rows, 1, Co, device=DEV, dtype=torch.bfloat16)
    ops.conv_igemm(ops.conv_desc(1, rows, 1, K, Co, 1, 1, 0), x.view(1, rows, 1, K), w.view(Co, 1, K), ref)
    assert_close(out, ref.view(rows, Co).float(), torch.bfloat16, bf16=1e-5, what="expand conv output")
    o = out.double()
    st = stats.sum(0)
    np.testing.assert_allclose(st[:, 0].cpu().numpy(), o.sum(0).cpu().numpy(), rtol=2e-6, atol=1e-3)
    np.testing.assert_allclose(st[:, 1].cpu().numpy(), (o * o).sum(0).cpu().numpy(), rtol=2e-6, atol=1e-3)
    want = x.double() @ w.double().t()
    assert_close(out, want.float(), torch.bfloat16, bf16=1e-2, what="vs fp64")


@pytest.mark.parametrize("rows,K,Co,mode", [(4 * 14 * 14, 64, 256, "mask+red"), (128 * 7 + 50, 64, 512, "plain"),
                                            (3 * 28 * 28, 128, 512, "mask+red"), (2 * 56 * 56, 128, 256, "acc+red"), (90, 64, 256, "mask+red")])
def test_conv_expand_dgrad_streaming_kernel(rows, K, Co, mode):
    """vince_conv_expand_dgrad against vince_conv_igemm's gradient instantiation on the same operands: the residual-gradient join
    through the mask bytes (or unmasked, or none) in place, and the fused BatchNorm-backward sums of the consumer BatchNorm."""
    ops = _ops()
    from vince_amd._lib import EPI_ACCUMULATE
    dy = (rnd(rows, K, seed=21) * 0.1).to(DEV).bfloat16()
    wt = (rnd(Co, K, seed=22) * (2.0 / Co) ** 0.5).to(DEV).bfloat16().contiguous()          # W^T: [Co][K]
    old = rnd(rows, Co, seed=23).to(DEV).bfloat16()
    ylo = rnd(rows, Co, seed=24).to(DEV).bfloat16()
    g = torch.Generator().manual_seed(25)
    amask = torch.randint(0, 256, (rows * Co // 8,), generator=g, dtype=torch.uint8).to(DEV)
    lbits = torch.randint(0, 256, (rows * Co // 8,), generator=g, dtype=torch.uint8).to(DEV)
    mean, invstd = rnd(Co, seed=26).to(DEV), (torch.rand(Co, generator=g) + 0.5).to(DEV)
    accumulate = mode != "plain"
    use_mask = mode == "mask+red"
    use_red = mode != "plain"
    res = {}
    for which in ("stream", "igemm"):
        out = old.clone()
        sums = torch.zeros(4, Co, 2, device=DEV, dtype=torch.float64)
        br = ops.bn_reduce_arg(ylo, mean, invstd, sums, mask_bits=lbits) if use_red else None
        if which == "stream":
            ops.conv_expand_dgrad(dy, wt, out, accumulate=accumulate, acc_mask=amask if use_mask else None, bnred=br, replicas=4)
        else:
            ops.conv_igemm(ops.conv_desc(1, rows, 1, K, Co, 1, 1, 0), dy.view(1, rows, 1, K), wt.view(Co, 1, K), out.view(1, rows, 1, Co),
                           flags=EPI_ACCUMULATE if accumulate else 0, acc_mask=amask if use_mask else None, bnred=br, replicas=4)
        res[which] = (out, sums.sum(0))
    a, b = res["stream"], res["igemm"]
    assert_close(a[0], b[0].float(), torch.bfloat16, bf16=8e-3, what="dx")          # (the join adds in fp32 in both; one bf16 rounding)
    # fp64 reference of the join
    bits = ((amask.view(rows, Co // 8, 1).int() >> torch.arange(8, device=DEV).view(1, 1, 8)) & 1).view(rows, Co).bool()
    want = dy.double() @ wt.double().t()
    if accumulate:
        want = want + (torch.where(bits, old.double(), torch.zeros_like(old.double())) if use_mask else old.double())
    assert_close(a[0], want.float(), torch.bfloat16, bf16=1e-2, what="dx vs fp64")
    if use_red:
        lb = ((lbits.view(rows, Co // 8, 1).int() >> torch.arange(8, device=DEV).view(1, 1, 8)) & 1).view(rows, Co).bool()
        gg = torch.where(lb, a[0].double(), torch.zeros_like(a[0].double()))
        xhat = (ylo.double() - mean.double()) * invstd.double()
        np.testing.assert_allclose(a[1][:, 0].cpu().numpy(), gg.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(a[1][:, 1].cpu().numpy(), (gg * xhat).sum(0).cpu().numpy(), rtol=1e-5, atol=2e-3)


def test_bf16_stores_round_to_nearest_even_like_the_reference():
    """Every bf16 tensor the library stores goes through pack_bf16x2 = v_cvt_pk_bf16_f32 (csrc/common.h).  Its rounding against torch's
    fp32 -> bfloat16 conversion (round to nearest even: what the reference's autocast-free bf16 tensors would hold), bit for bit, on random
    bit patterns, every halfway case neighbourhood, denormals, the largest finite values (which round to infinity) and infinities --
    through vince_bn_apply with y = 0, scale = 1, shift = the test values."""
    ops = _ops()
    g = torch.Generator().manual_seed(77)
    bits = torch.randint(0, 2 ** 32, (1 << 16,), generator=g, dtype=torch.int64)
    hi = torch.randint(0, 2 ** 16, (4096,), generator=g, dtype=torch.int64) << 16
    edge = torch.cat([hi | lo for lo in (0x7fff, 0x8000, 0x8001, 0xffff, 0x0001, 0x0000)])
    special = torch.tensor([0x00000001, 0x00008000, 0x00018000, 0x007fffff, 0x7f7fffff, 0x7f7f8000, 0x7f7f7fff, 0xff7fffff, 0x7f800000,
                            0xff800000, 0x3f800000, 0x3f808000, 0x3f818000], dtype=torch.int64)
    allbits = torch.cat([bits, edge, special])
    vals = (allbits & 0xffffffff).to(torch.int64)
    vals = torch.where(vals >= 2 ** 31, vals - 2 ** 32, vals).to(torch.int32).view(torch.float32)
    vals = vals[~torch.isnan(vals)]
    vals = vals[: vals.numel() // 8 * 8]
    want = (torch.zeros_like(vals) + vals).bfloat16().view(torch.int16)                  # (0 + x: a negative zero enters as +0)
    C = vals.numel()
    y = torch.zeros(64, C, device=DEV, dtype=torch.bfloat16)
    out = ops.bn_apply(y, torch.ones(C, device=DEV), vals.to(DEV), relu=False)
    got = out.cpu().view(torch.int16)
    assert torch.equal(got[0], want), int((got[0] != want).sum())
    assert torch.equal(got[63], want)


@pytest.mark.parametrize("rows,cin,cout", [(16, 512, 512), (144, 512, 64), (16, 4608, 512), (256, 2048, 2048), (256, 2048, 128)])
def test_head_linear_split_half_vs_fp64(rows, cin, cout):
    """The projection head's Linear layers (vince_model.py:38-42) as split-half products of bfloat16 halves -- what a bf16-trunk model
    runs (models/vince_model.py _HeadCopies) -- against fp64: forward (bias + ReLU, the tiny-M reduction split over workgroups with
    atomics) within 2e-5 of max |y|, both gradients within 5e-5; and operands far outside the IEEE-half range stay finite (the reason
    the head takes bfloat16 halves, not the trunk's IEEE-half ones)."""
    ops = _ops()
    x = torch.relu(rnd(rows, cin, seed=31)).to(DEV)
    w = (rnd(cout, cin, seed=32) * (2.0 / cin) ** 0.5).to(DEV)
    b = (rnd(cout, seed=33) * 0.1).to(DEV)
    wk, wt = ops.prepare_weight(w.view(cout, 1, cin), torch.float32, want_transposed=True, x3="b")
    y = ops.linear_fwd(x, wk, b, relu=True, x3="b")
    ref = torch.relu(x.double() @ w.double().t() + b.double())
    assert float((y.double() - ref).abs().max() / ref.abs().max()) < 2e-5
    assert torch.isfinite(ops.linear_fwd(x * 1e6, wk, b, relu=True, x3="b")).all()
    dy = (rnd(rows, cout, seed=34) * 1e-3).to(DEV)
    dw, db = torch.zeros(cout, cin, device=DEV), torch.zeros(cout, device=DEV)
    dx = ops.linear_bwd(x, wt, dy, dw, db, x3=True)
    rdx, rdw = dy.double() @ w.double(), dy.double().t() @ x.double()
    assert float((dx.double() - rdx).abs().max() / rdx.abs().max()) < 5e-5
    assert float((dw.double() - rdw).abs().max() / rdw.abs().max()) < 5e-5
    np.testing.assert_allclose(db.cpu().numpy(), dy.sum(0).cpu().numpy(), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("pieces", [1, 2, 3, 1023, 4097, 65537])
def test_stream_copy_copies_every_byte(pieces):
    """The library's own copy kernel (the streaming ceiling of bench.py's roofline leg) in every shape, on piece counts that are odd /
    below one wavefront / ragged against the grid: each shape must move ALL bytes (the 32-bytes-per-lane shape dropped
    the last 16-byte piece of an odd count)."""
    from vince_amd import _lib
    L = _lib.lib()
    nbytes = pieces * 16
    src = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    for mode in range(6):        # bit 0 = non-temporal; bits 1-2 = shape
        for blocks in (1, 7, 64):
            dst = torch.zeros_like(src)
            _lib.check(L.vince_stream_copy(dst.data_ptr(), src.data_ptr(), nbytes, blocks, mode, st))
            assert torch.equal(dst, src), (mode, blocks)


def test_nonfinite_loss_latch():
    """The per-iteration finite-loss assertion of the reference (solvers/vince_solver.py:446-452) as a device-side latch: finite
    values leave it alone, NaN / +-inf count and keep the FIRST offending step."""
    ops = _ops()
    latch = torch.zeros(2, dtype=torch.int64, device=DEV)
    for step, v in enumerate([1.5, -3.0e38, 0.0, float("inf"), 2.0, float("nan"), float("-inf")]):
        ops.nonfinite_latch(torch.tensor([v], device=DEV), step, latch)
    assert latch.tolist() == [3, 3 + 1]
    from vince_amd.solvers.vince_solver import VinceSolver
    import types
    stub = types.SimpleNamespace(_loss_latch=latch)
    with pytest.raises(AssertionError, match="first at iteration 3"):
        VinceSolver.check_loss_latch(stub)
    VinceSolver.check_loss_latch(types.SimpleNamespace(_loss_latch=torch.zeros(2, dtype=torch.int64, device=DEV)))


def test_bn3_algebra_masked_pixel_sums_at_engine_scale():
    """What the BatchNorm BELOW the algebra reduces are sums of the input gradient over ITS ReLU mask (a > 0): cancellations of a few
    hundred to one, in which anything that is the same at every pixel shows.  At the engine's size (28 x 28 x 256 pixels, w = 64, correlated
    activations, an upstream gradient with a per-channel mean) against fp64: the mixed mode's form (fp32 master weights, matrices in
    bfloat16 hi + lo parts, the constant on the fp32 accumulators) is as good as the separate passes (3.6e-3 emulated); single bf16
    matrices are not (5e-2), and the constant added to the staged bf16 value was 1.6e-1 (rounds 3-5; tools/alg_op_probe.py)."""
    ops = _ops()
    from vince_amd._lib import ConvDesc
    rows, w = 200704, 64
    Co = 4 * w
    g0 = torch.Generator(device=DEV).manual_seed(3)

    def rn(*shape):
        return torch.randn(*shape, generator=g0, device=DEV)
    a = torch.relu(rn(rows, 16) @ (rn(16, w) * 0.5) + 0.5 * rn(rows, w) + 0.3)
    W = rn(Co, w) * (2.0 / w) ** 0.5
    gamma, beta = torch.rand(Co, generator=g0, device=DEV) + 0.5, rn(Co) * 0.2
    ident, G = rn(rows, Co), rn(rows, Co) * 0.1 + 0.02
    a64, W64 = a.double(), W.double()
    y = a64 @ W64.t()
    mu, var = y.mean(0), y.var(0, unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    xhat = (y - mu) * invstd
    gb = (G * ((xhat * gamma.double() + beta.double() + ident.double()) > 0)).bfloat16()
    g = gb.double()
    s = gamma.double() * invstd
    da_true = (s * (g - g.mean(0) - xhat * (g * xhat).mean(0))) @ W64
    m2 = (a64 > 0).double()
    ref = (da_true * m2).sum(0)
    ab = a.bfloat16()
    R = torch.zeros(Co, 1, w, device=DEV)
    ops.conv_wgrad(ops.conv_desc(1, rows, 1, w, Co, 1, 1, 0), ab.view(1, rows, 1, w), gb.view(1, rows, 1, Co), R)
    colsum = torch.zeros(4, w, device=DEV, dtype=torch.float64)
    colsum[0] = a64.sum(0)
    gs = torch.zeros(ops.STATS_REPLICAS, Co, 2, device=DEV, dtype=torch.float64)
    gs[0, :, 0] = g.sum(0)
    err = {}
    for name, wt, split in (("single", W.bfloat16(), False), ("nq", W, "nq"), ("split", W, True)):
        coef, w2, nr = ops.bn3_bwd_prepare(R.view(Co, w), wt.contiguous(), gs, mu.float(), invstd.float(), gamma, rows,
                                           torch.zeros(Co, device=DEV), torch.zeros(Co, device=DEV), colsum=colsum, split=split)
        taps = 3 if split is True else 2
        dd = ConvDesc(N=1, Hi=rows, Wi=1, Ci=Co, Ho=rows, Wo=1, Co=w, sh=1, sw=1, TA=1, TB=taps, dh0=0, dhs=1, dw0=0, dws=0, wt0=0, wta=0,
                      wtb=1, WT=taps, OH=rows, OW=1, osh=1, osw=1, oh0=0, ow0=0)
        da = torch.empty(rows, w, device=DEV, dtype=torch.bfloat16)
        ops.conv_igemm(dd, gb.view(1, rows, 1, Co), w2, da.view(1, rows, 1, w), bias=nr, in2=ab.view(1, rows, 1, w), in2_repeat=2 if split else 0)
        err[name] = float(((da.double() * m2).sum(0) - ref).abs().max() / ref.abs().max())
        assert float((da.double() - da_true).norm() / da_true.norm()) < 5e-3
    print("bn3 algebra at 200704 x 64: masked pixel sums vs fp64 %.2e (fp32 weights, hi + lo matrices) %.2e (nq alone in two parts: the engine's default) "
          "%.2e (single bf16 matrices)" % (err["split"], err["nq"], err["single"]))
    assert err["split"] < 6e-3 and err["nq"] < 3e-2 and err["single"] < 1.2e-1


@pytest.mark.parametrize("w,rows", [(64, 3000), (128, 1111)])
def test_bn3_backward_algebra_vs_autograd(w, rows):
    """csrc/bn_algebra.hip: BatchNorm backward THROUGH a bottleneck's last 1x1 convolution without that convolution's output
    (resnet.py:123-133).  Oracle: torch autograd in fp64 on the CPU of  z = relu(bn3(W a) + identity)  with the SAME bf16-valued W and a
    (so the only differences are the bf16 rounding of the derived dgrad weights and of the stored input gradient).  Checked: the masked
    gradient hand-off (out_mask + sums, both producers), dgamma / dbeta, the weight gradient, the input gradient."""
    ops = _ops()
    Co = 4 * w
    g = torch.Generator().manual_seed(5 + w)
    a = torch.relu(torch.randn(rows, w, generator=g) + 0.3).bfloat16().float()
    W = (torch.randn(Co, w, generator=g) * (2.0 / w) ** 0.5).bfloat16().float()
    gamma = torch.rand(Co, generator=g) + 0.5
    beta = torch.randn(Co, generator=g) * 0.2
    ident = torch.randn(rows, Co, generator=g).bfloat16().float()
    G = (torch.randn(rows, Co, generator=g) * 0.1 + 0.02).bfloat16().float()          # upstream gradient of the block output
    # ---- oracle (fp64 autograd)
    a64, W64 = a.double().requires_grad_(True), W.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y = a64 @ W64.t()
    mu, var = y.mean(0), y.var(0, unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    u = (y - mu) * invstd * g64 + b64
    z = torch.relu(u + ident.double())
    z.backward(G.double())
    keep = (z.detach() > 0)
    gm = (G.double() * keep).float()                                                    # g = G gated by the block's ReLU
    # ---- the hand-off: producers store (dgrad + old) gated by the mask bits and sum it per channel
    bits = (keep.reshape(rows, Co // 8, 8).long() << torch.arange(8)).sum(-1).to(torch.uint8)
    K2 = 64
    dy2 = (torch.randn(rows, K2, generator=g) * 0.1).bfloat16()
    wt2 = (torch.randn(Co, K2, generator=g) * 0.1).bfloat16()
    old = G.bfloat16()
    want = ((dy2.float() @ wt2.float().t() + old.float()) * keep).bfloat16()
    for route in ("xjoin", "igemm"):
        out = old.clone().to(DEV)
        sums = torch.zeros(ops.STATS_REPLICAS, Co, 2, device=DEV, dtype=torch.float64)
        if route == "xjoin":
            ops.conv_expand_dgrad_masked(dy2.to(DEV), wt2.to(DEV), out, bits.to(DEV), sums, accumulate=True)
        else:
            d = ops.conv_desc(1, rows, 1, K2, Co, 1, 1, 0)
            ops.conv_igemm(d, dy2.to(DEV).view(1, rows, 1, K2), wt2.to(DEV).view(Co, 1, K2), out.view(1, rows, 1, Co), stats=sums,
                           flags=ops.EPI_ACCUMULATE, out_mask=bits.to(DEV))
        err = (out.float().cpu() - want.float()).abs().max() / want.float().abs().max()
        assert err < 1e-2, (route, float(err))
        assert bool(((out.float().cpu() != 0) <= keep).all()), route                  # nothing leaks through a closed gate
        s_want = out.float().cpu().double().sum(0)
        np.testing.assert_allclose(sums.sum(0)[:, 0].cpu().numpy(), s_want.numpy(), rtol=1e-6, atol=1e-4)
    # ---- the algebra itself, fed with the exact g
    gb = gm.bfloat16().to(DEV)
    ab = a.bfloat16().to(DEV)
    Wb = W.bfloat16().to(DEV)
    d3 = ops.conv_desc(1, rows, 1, w, Co, 1, 1, 0)
    R = torch.zeros(Co, 1, w, device=DEV)
    ops.conv_wgrad(d3, ab.view(1, rows, 1, w), gb.view(1, rows, 1, Co), R)
    dgr = ops.conv_desc(1, rows, 1, w, w, 1, 1, 0)
    gram = torch.zeros(w, 1, w, device=DEV)
    ops.conv_wgrad(dgr, ab.view(1, rows, 1, w), ab.view(1, rows, 1, w), gram)
    colsum = torch.zeros(4, w, device=DEV, dtype=torch.float64)
    colsum[0] = ab.double().sum(0)
    gs = torch.zeros(ops.STATS_REPLICAS, Co, 2, device=DEV, dtype=torch.float64)
    gs[1, :, 0] = gb.double().sum(0)
    dgam, dbet = torch.zeros(Co, device=DEV), torch.zeros(Co, device=DEV)
    mean32, invstd32 = mu.detach().float().to(DEV), invstd.detach().float().to(DEV)
    # (first the form of rounds 3-5 -- the forward's mean, nr from the unrounded coefficients -- for the pixel-sum comparison below)
    _, w2_old, nr_old = ops.bn3_bwd_prepare(R.view(Co, w), Wb, gs, mean32, invstd32, gamma.to(DEV), rows, torch.zeros_like(dgam), torch.zeros_like(dbet))
    # the engine's form: colsum given -> the implied mean W colsum / n (here W and a ARE what the forward multiplied, so it equals mu to
    # rounding) and nr formed from the rounded wd / nq
    coef, w2, nr = ops.bn3_bwd_prepare(R.view(Co, w), Wb, gs, mean32, invstd32, gamma.to(DEV), rows, dgam, dbet, colsum=colsum)
    assert coef.shape == (5, Co) and float((coef[4].cpu() - mu.detach().float()).abs().max()) < 1e-5 * float(mu.detach().abs().max()) + 1e-6
    assert float((w2.float() - w2_old.float()).abs().max()) <= 2.0 ** -7 * float(w2_old.float().abs().max())   # (the mean moved in its last bits)
    mean32 = coef[4].contiguous()
    # sum over pixels of da = wd_r G + nq_r A + n nr: BatchNorm backward hands conv3 a dy whose pixel sums vanish, so this is 0 in exact
    # arithmetic; with nr formed from the ROUNDED matrices it is 0 to fp32 rounding, with the unrounded form the bf16 residue (the same
    # sign at every pixel) stays -- and lands in the bias gradient of the BatchNorm below
    Gs, As = gb.double().sum(0).cpu(), ab.double().sum(0).cpu()
    wd64, nq64 = w2[:, 0].double().cpu(), w2[:, 1, :w].double().cpu()
    scale = (wd64.abs() @ Gs.abs() + nq64.abs() @ As.abs())
    resid = ((wd64 @ Gs + nq64 @ As + rows * nr.double().cpu()).abs() / scale).max()
    resid_old = ((wd64 @ Gs + nq64 @ As + rows * nr_old.double().cpu()).abs() / scale).max()
    print("bn3 algebra w=%d: pixel-sum residue of da relative to its terms: %.2e (nr from the unrounded coefficients: %.2e)" % (w, resid, resid_old))
    assert resid < 2e-6 and resid_old > 10 * resid
    # the accumulate-into form (what the engine uses: R stays scratch, the finished gradient is ADDED to a buffer that already holds one)
    R0 = R.clone()
    acc = torch.full((Co, w), 0.5, device=DEV)
    ops.bn3_bwd_finish_dw(R.view(Co, w), Wb, gram.view(w, w), colsum, coef, mean32, invstd32, dw_accum=acc)
    assert torch.equal(R, R0)
    ops.bn3_bwd_finish_dw(R.view(Co, w), Wb, gram.view(w, w), colsum, coef, mean32, invstd32)       # in place
    assert float((acc - 0.5 - R.view(Co, w)).abs().max()) <= 1e-6 * float(R.abs().max()) + 1e-7

    def relerr(x, ref):
        return float((x.double().cpu() - ref).abs().max() / ref.abs().max())
    assert relerr(dgam, g64.grad) < 2e-3 and relerr(dbet, b64.grad) < 2e-3, (relerr(dgam, g64.grad), relerr(dbet, b64.grad))
    assert relerr(R.view(Co, w), W64.grad) < 5e-3, relerr(R.view(Co, w), W64.grad)
    # the input gradient in ONE launch: the reduction runs over g's Co channels (tap 0) and then over a's w channels (tap 1 = in2)
    da = torch.empty(rows, w, device=DEV, dtype=torch.bfloat16)
    from vince_amd._lib import ConvDesc
    dd = ConvDesc(N=1, Hi=rows, Wi=1, Ci=Co, Ho=rows, Wo=1, Co=w, sh=1, sw=1, TA=1, TB=2, dh0=0, dhs=1, dw0=0, dws=0, wt0=0, wta=0,
                  wtb=1, WT=2, OH=rows, OW=1, osh=1, osw=1, oh0=0, ow0=0)
    ops.conv_igemm(dd, gb.view(1, rows, 1, Co), w2, da.view(1, rows, 1, w), bias=nr, in2=ab.view(1, rows, 1, w))
    # ... equals the two launches it replaces (wd g + nr, then += nq a) up to one bf16 rounding of the intermediate
    da2 = torch.empty_like(da)
    ops.conv_igemm(ops.conv_desc(1, rows, 1, Co, w, 1, 1, 0), gb.view(1, rows, 1, Co), w2[:, 0].contiguous().view(w, 1, Co),
                   da2.view(1, rows, 1, w), bias=nr)
    ops.conv_igemm(dgr, ab.view(1, rows, 1, w), w2[:, 1, :w].contiguous().view(w, 1, w), da2.view(1, rows, 1, w), flags=ops.EPI_ACCUMULATE)
    assert relerr(da.float(), da2.float().double().cpu()) < 1.2e-2
    e_da = relerr(da.float(), a64.grad)
    assert e_da < 2e-2, e_da
    # ---- the mixed mode's form (round 6): fp32 master weights, both matrices of the input gradient in bfloat16 hi + lo parts, reduced as
    # [wd_hi | wd_lo] g + [nq_hi | nq_lo] a (vince_conv_epi.in2_repeat).  What it is for: the sums the BatchNorm below takes over ITS ReLU
    # mask (a > 0) -- heavy cancellations in which the single-bf16 matrices' rounding, the same sign at every pixel, shows.
    dg3, db3 = torch.zeros(Co, device=DEV), torch.zeros(Co, device=DEV)
    coef3, w3, nr3 = ops.bn3_bwd_prepare(R0.view(Co, w), W.to(DEV), gs, mean32, invstd32, gamma.to(DEV), rows, dg3, db3, colsum=colsum, split=True)
    assert w3.shape == (w, 3, Co) and relerr(dg3, g64.grad) < 2e-3 and relerr(db3, b64.grad) < 2e-3
    hi, lo = w3[:, 0].float(), w3[:, 1].float()
    assert float(lo.abs().max()) <= 2.0 ** -8 * float(hi.abs().max()) and float(lo.abs().max()) > 0      # a remainder, not a copy
    wd_exact = (W * coef3[0].cpu()[:, None]).t()
    assert relerr(hi + lo, wd_exact.double()) < 2.0 ** -15
    Rf = R0.clone()
    ops.bn3_bwd_finish_dw(Rf.view(Co, w), W.to(DEV), gram.view(w, w), colsum, coef3, coef3[4].contiguous(), invstd32)
    assert relerr(Rf.view(Co, w), W64.grad) < 5e-3
    da3 = torch.empty(rows, w, device=DEV, dtype=torch.bfloat16)
    dd3 = ConvDesc(N=1, Hi=rows, Wi=1, Ci=Co, Ho=rows, Wo=1, Co=w, sh=1, sw=1, TA=1, TB=3, dh0=0, dhs=1, dw0=0, dws=0, wt0=0, wta=0,
                   wtb=1, WT=3, OH=rows, OW=1, osh=1, osw=1, oh0=0, ow0=0)
    ops.conv_igemm(dd3, gb.view(1, rows, 1, Co), w3, da3.view(1, rows, 1, w), bias=nr3, in2=ab.view(1, rows, 1, w), in2_repeat=2)
    # (the same reduction as three plain launches: wd_hi g + nr, += wd_lo g, += (nq_hi + nq_lo) a)
    w3f = w3.float()
    want3 = (gb.float() @ (w3f[:, 0] + w3f[:, 1]).t() + ab.float() @ (w3f[:, 2, :w] + w3f[:, 2, w:2 * w]).t() + nr3).double().cpu()
    assert relerr(da3.float(), want3) < 6e-3            # (one bf16 rounding of the stored value)
    e_da3 = relerr(da3.float(), a64.grad)
    m2 = (a > 0).double()

    def masked(dax):
        got, ref = (dax.double().cpu() * m2).sum(0), (a64.grad * m2).sum(0)
        return float((got - ref).abs().max() / ref.abs().max())
    print("bn3 algebra w=%d: input gradient vs fp64 %.2e (split) %.2e (single); masked pixel sums %.2e (split) %.2e (single)"
          % (w, e_da3, e_da, masked(da3.float()), masked(da.float())))
    assert e_da3 < 1.2e-2
    # the engine fuses the reduction of the BatchNorm BELOW (bn2: its own ReLU mask from its input) into this launch: the sums must be those
    # of the stand-alone reduction over the stored gradient, bias and second tensor included
    y2 = torch.randn(rows, w, generator=g).bfloat16().to(DEV)
    m2c, i2c = (torch.randn(w, generator=g) * 0.1).to(DEV), (torch.rand(w, generator=g) + 0.5).to(DEV)
    msc, msh = (torch.rand(w, generator=g) + 0.5).to(DEV), (torch.randn(w, generator=g) * 0.3).to(DEV)
    for desc, wts, nrr, rep in ((dd, w2, nr, 0), (dd3, w3, nr3, 2)):
        sums = torch.zeros(ops.STATS_REPLICAS, w, 2, device=DEV, dtype=torch.float64)
        out = torch.empty(rows, w, device=DEV, dtype=torch.bfloat16)
        br = ops.bn_reduce_arg(y2.view(1, rows, 1, w), m2c, i2c, sums, mask_scale=msc, mask_shift=msh)
        ops.conv_igemm(desc, gb.view(1, rows, 1, Co), wts, out.view(1, rows, 1, w), bias=nrr, in2=ab.view(1, rows, 1, w), in2_repeat=rep, bnred=br,
                       replicas=4)
        assert torch.equal(out, da3 if rep else da)
        want = ops.bn_bwd_reduce(out.view(1, rows, 1, w), y2.view(1, rows, 1, w), m2c, i2c, mask_scale=msc, mask_shift=msh)
        got = sums.sum(0)
        assert float((got - want).abs().max()) / (float(want.abs().max()) + 1e-6) < 1e-5, ("fused reduce behind the algebra's launch", rep)
