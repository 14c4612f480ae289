"""Split-half products (VINCE_F32X3H / VINCE_F32X3B, compute_dtype="x3"): fp32 tensors whose convolutions run as three half-precision
MFMAs of hi / lo halves.  Op level against fp64 convolutions on the CPU -- held to what the claim is: forward launches (IEEE half
halves) at fp32's own error, gradient launches (bfloat16 halves) at 2^-16 -- and the model against the reference's goldens at the
north-star bar (1e-3 on embeddings and loss).  Run with -m gpu on an MI355X."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import vince_oracle as vo  # noqa: E402
from tests.test_ops_gpu import CONVS, DEV, from_nhwc, rnd, to_nhwc, weights_krsc  # noqa: E402


def _ops():
    from vince_amd import ops
    return ops


def _err(got, want):
    got, want = got.double().cpu(), want.double().cpu()
    return float((got - want).abs().max() / (want.abs().max() + 1e-300))


@pytest.mark.parametrize("xscale", [1.0, 1e-3, 300.0])
@pytest.mark.parametrize("cfg", CONVS)
def test_x3h_forward_conv_is_as_good_as_fp32(cfg, xscale):
    """The forward convolution through IEEE-half hi / lo halves against an fp64 convolution of the same fp32 operands: within 4e-6 of
    max |y| (the exact-fp32 MFMA kernel of the same layer is measured beside it: ~1e-6), at activations of order one, at 1e-3 (the
    lo halves of the activations are subnormal half numbers there: the matrix pipe must not flush them) and at 300."""
    ops = _ops()
    N, H, W, Ci, Co, k, s, p = cfg
    x = rnd(N, Ci, H, W, seed=1) * xscale
    w = rnd(Co, Ci, k, k, seed=2, scale=(2.0 / (Ci * k * k)) ** 0.5)
    ref = F.conv2d(x.double(), w.double(), None, s, p)
    wk, _ = weights_krsc(w, torch.float32)
    wk3, _ = weights_krsc(w, torch.float32, x3=True)     # the split-half weight layout (IEEE half pairs)
    d = ops.conv_desc(N, H, W, Ci, Co, k, s, p)
    xg = to_nhwc(x, torch.float32)
    out = torch.empty(N, d.Ho, d.Wo, Co, device=DEV)
    stats = torch.zeros(ops.STATS_REPLICAS, Co, 2, device=DEV, dtype=torch.float64)
    ops.conv_igemm(d, xg, wk3, out, stats=stats, x3="h")
    e3 = _err(from_nhwc(out), ref)
    out32 = torch.empty_like(out)
    ops.conv_igemm(d, xg, wk, out32)
    e32 = _err(from_nhwc(out32), ref)
    print("x3h conv %s x%g: err %.2e (fp32 MFMA kernel %.2e)" % (cfg, xscale, e3, e32))
    assert e3 < 4e-6, (e3, e32)
    o = out.cpu().reshape(-1, Co).double()
    np.testing.assert_allclose(stats.sum(0)[:, 0].cpu().numpy(), o.sum(0).numpy(), rtol=1e-5, atol=1e-3 * xscale)


def test_x3h_operands_past_the_half_range_fail_loudly():
    """The IEEE-half split scales activations by 2^4 and weights by 2^8 (csrc/common.h X3_XSHIFT / X3_WSHIFT).  Just inside the range
    (|x| = 4000, |w| = 250) the product is still fp32-grade; past it (|x| >= 4095, |w| * 2^8 >= 65520 -- e.g. a folded eval weight
    w * gamma * invstd with a near-zero running variance) the result must be NaN / inf, never a finite saturated number."""
    ops = _ops()
    N, H, W, Ci, Co, k, s, p = 2, 8, 8, 64, 64, 1, 1, 0
    d = ops.conv_desc(N, H, W, Ci, Co, k, s, p)
    w = rnd(Co, Ci, k, k, seed=2, scale=(2.0 / Ci) ** 0.5)
    x = rnd(N, Ci, H, W, seed=1)
    x = x / x.abs().max()

    def run(xx, ww):
        wk3, _ = weights_krsc(ww, torch.float32, x3=True)
        out = torch.empty(N, d.Ho, d.Wo, Co, device=DEV)
        ops.conv_igemm(d, to_nhwc(xx, torch.float32), wk3, out, x3="h")
        return from_nhwc(out)

    ok = run(x * 4000.0, w)
    assert _err(ok, F.conv2d(x.double() * 4000.0, w.double())) < 4e-6
    assert not torch.isfinite(run(x * 4200.0, w)).all(), "an activation past 65520 / 2^4 must not saturate silently"
    wbig = w / w.abs().max()
    okw = run(x, wbig * 250.0)
    assert _err(okw, F.conv2d(x.double(), wbig.double() * 250.0)) < 4e-6
    assert not torch.isfinite(run(x, wbig * 300.0)).all(), "a weight past 65520 / 2^8 must not saturate silently"


@pytest.mark.parametrize("cfg", CONVS)
def test_x3b_gradients_vs_fp64(cfg):
    """Input and weight gradients through bfloat16 hi / lo halves (fp32's exponent range: dy of order 1e-6 here) against fp64 autograd:
    2^-16 per product, held at 1e-4 of the largest entry; the accumulate / masked epilogues run on the same instantiation."""
    ops = _ops()
    N, H, W, Ci, Co, k, s, p = cfg
    x = rnd(N, Ci, H, W, seed=6).double().requires_grad_(True)
    w = rnd(Co, Ci, k, k, seed=7, scale=(2.0 / (Ci * k * k)) ** 0.5).double().requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    dy = rnd(*y.shape, seed=8) * 1e-6
    y.backward(dy.double())
    wk, wt = weights_krsc(w.detach().float(), torch.float32, x3=True)     # wt: the split-half layout, bfloat16 pairs
    dyg = to_nhwc(dy, torch.float32)
    dx = torch.full((N, H, W, Ci), float("nan"), device=DEV)
    descs = ops.dgrad_descs(N, H, W, Ci, Co, k, s, p)
    if len(descs) < s * s:
        dx.zero_()
    for d in descs:
        ops.conv_igemm(d, dyg, wt, dx, x3="b")
    e = _err(from_nhwc(dx), x.grad)
    for d in descs:
        ops.conv_igemm(d, dyg, wt, dx, flags=ops.EPI_ACCUMULATE, x3="b")
    e2 = _err(from_nhwc(dx), 2 * x.grad)
    fd = ops.conv_desc(N, H, W, Ci, Co, k, s, p)
    ref_dw = w.grad.permute(0, 2, 3, 1).reshape(Co, k * k, Ci)
    dw = torch.zeros(Co, k * k, Ci, device=DEV)
    ops.conv_wgrad(fd, to_nhwc(x.detach().float(), torch.float32), dyg, dw, x3="b")
    ew = _err(dw, ref_dw)
    dets = []
    for _ in range(2):
        dwd = torch.zeros(Co, k * k, Ci, device=DEV)
        ops.conv_wgrad_det(fd, to_nhwc(x.detach().float(), torch.float32), dyg, dwd, x3="b")
        dets.append(dwd)
    print("x3b %s: dgrad %.2e (accumulated %.2e), wgrad %.2e, deterministic wgrad %.2e" % (cfg, e, e2, ew, _err(dets[0], ref_dw)))
    assert e < 1e-4 and e2 < 1e-4 and ew < 1e-4 and _err(dets[0], ref_dw) < 1e-4
    assert torch.equal(dets[0], dets[1])


@pytest.mark.parametrize("hw", [(32, 32), (45, 51)])
def test_x3_stem_packed_rows(hw):
    """The engine's stem (7 packed row taps over the [N][H][Wp][4] layout) through the split-half forward and weight gradient."""
    ops = _ops()
    H, W = hw
    N = 2
    x = rnd(N, 3, H, W, seed=3).double()
    w = rnd(64, 3, 7, 7, seed=4, scale=(2.0 / 147) ** 0.5).double().requires_grad_(True)
    y = F.conv2d(x, w, None, 2, 3)
    dy = rnd(*y.shape, seed=5)
    y.backward(dy.double())
    xin = ops.input_nchw_to_rows(x.float().to(DEV), torch.float32)
    wp = torch.zeros(64, 7, 8, 4)
    wp[:, :, :7, :3] = w.detach().float().permute(0, 2, 3, 1)
    wk, _ = ops.prepare_weight(wp.reshape(64, 7, 32).to(DEV).contiguous(), torch.float32, want_transposed=False, x3=True)
    d = ops.stem_desc(N, H, W)
    out = torch.empty(N, d.Ho, d.Wo, 64, device=DEV)
    ops.conv_igemm(d, xin, wk, out, x3="h")
    assert _err(from_nhwc(out), y.detach()) < 4e-6
    dw = torch.zeros(64, 49, 3, device=DEV)
    ops.conv_wgrad(d, xin, to_nhwc(dy, torch.float32), dw, ci_dw=3, x3="b")
    assert _err(dw, w.grad.permute(0, 2, 3, 1).reshape(64, 49, 3)) < 1e-4


@pytest.mark.parametrize("cfg", CONVS)
def test_x1b_single_product_gradients_vs_fp64(cfg):
    """VINCE_F32X1B (the gradient launches of compute_dtype "x3f"): fp32 tensors, ONE bfloat16 MFMA per block -- both operands rounded
    to 8 significand bits, fp32 accumulation.  Input and weight gradients against fp64 autograd: rounding errors of 2^-9 per operand
    average out over the reduction; held at 6e-3 of the largest entry (measured 1e-3 ... 3e-3), and -- the point of the mode -- the
    SAME launches with three products (x3b) are three orders tighter on the same operands."""
    ops = _ops()
    N, H, W, Ci, Co, k, s, p = cfg
    x = rnd(N, Ci, H, W, seed=6).double().requires_grad_(True)
    w = rnd(Co, Ci, k, k, seed=7, scale=(2.0 / (Ci * k * k)) ** 0.5).double().requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    dy = rnd(*y.shape, seed=8) * 1e-6
    y.backward(dy.double())
    _, wt = weights_krsc(w.detach().float(), torch.float32, x3=True)     # the bfloat16-pair layout; the lo halves are not read
    dyg = to_nhwc(dy, torch.float32)
    res = {}
    for tag in ("1", "b"):
        dx = torch.full((N, H, W, Ci), float("nan"), device=DEV)
        descs = ops.dgrad_descs(N, H, W, Ci, Co, k, s, p)
        if len(descs) < s * s:
            dx.zero_()
        for d in descs:
            ops.conv_igemm(d, dyg, wt, dx, x3=tag)
        fd = ops.conv_desc(N, H, W, Ci, Co, k, s, p)
        dw = torch.zeros(Co, k * k, Ci, device=DEV)
        ops.conv_wgrad(fd, to_nhwc(x.detach().float(), torch.float32), dyg, dw, x3=tag)
        res[tag] = (_err(from_nhwc(dx), x.grad), _err(dw, w.grad.permute(0, 2, 3, 1).reshape(Co, k * k, Ci)))
    print("x1b %s: dgrad %.2e wgrad %.2e (x3b on the same operands: %.2e / %.2e)" % ((cfg,) + res["1"] + res["b"]))
    assert res["1"][0] < 6e-3 and res["1"][1] < 6e-3
    assert res["b"][0] < 1e-4 and res["b"][1] < 1e-4
    assert res["1"][0] > 10 * res["b"][0]        # (the single-product launches really are a different arithmetic)


@pytest.mark.parametrize("cfg", [(2, 14, 14, 64, 64, 3, 1, 1), (2, 15, 15, 128, 128, 3, 2, 1), (3, 14, 14, 256, 256, 3, 1, 1), (3, 9, 11, 64, 256, 1, 1, 0)])
def test_x3h_stored_half_pairs_equal_the_in_loop_split(cfg):
    """A BatchNorm pass that writes its output as stored IEEE-half pairs (vince_bn_train.out_half_pairs) followed by a split-half
    convolution that multiplies the pairs as they are (VINCE_EPI_IN_HALF_PAIRS) computes, bit for bit, what the fp32 output followed by
    the in-loop split computes: the stored halves ARE the halves the kernel would make (round 6: bn1 -> 3x3 of every block in the
    forwards whose activations no fp32 reader needs)."""
    ops = _ops()
    N, H, W, Ci, Co, k, s, p = cfg
    y = to_nhwc(rnd(N, Ci, H, W, seed=11) * 3.0 + 0.5, torch.float32)          # the raw output of the convolution in front
    rows = N * H * W
    stats = torch.zeros(ops.STATS_REPLICAS, Ci, 2, device=DEV, dtype=torch.float64)
    yf = y.reshape(rows, Ci).double()
    stats[0, :, 0], stats[0, :, 1] = yf.sum(0), (yf * yf).sum(0)
    gamma = (1.0 + 0.1 * rnd(Ci, seed=12)).to(DEV)
    beta = (0.1 * rnd(Ci, seed=13)).to(DEV)
    a32 = ops.bn_train_apply(y, stats, rows, gamma, beta)[0]
    ahp = ops.bn_train_apply(y, stats, rows, gamma, beta, half_pairs=True)[0]
    assert not torch.equal(a32, ahp)                                            # (another byte layout altogether)
    w = rnd(Co, Ci, k, k, seed=2, scale=(2.0 / (Ci * k * k)) ** 0.5)
    wk3, _ = weights_krsc(w, torch.float32, x3=True)
    d = ops.conv_desc(N, H, W, Ci, Co, k, s, p)
    out32 = torch.empty(N, d.Ho, d.Wo, Co, device=DEV)
    outhp = torch.full_like(out32, float("nan"))
    st32 = torch.zeros(ops.STATS_REPLICAS, Co, 2, device=DEV, dtype=torch.float64)
    sthp = torch.zeros_like(st32)
    ops.conv_igemm(d, a32, wk3, out32, stats=st32, x3="h")
    ops.conv_igemm(d, ahp, wk3, outhp, stats=sthp, flags=ops.EPI_IN_HALF_PAIRS, x3="h")
    assert torch.equal(out32, outhp), float((out32 - outhp).abs().max())
    assert float(out32.abs().max()) > 0.1
    # refused where it cannot work: other dtypes
    with pytest.raises(RuntimeError):
        ops.conv_igemm(d, ahp, wk3, outhp, flags=ops.EPI_IN_HALF_PAIRS, x3="b")
