"""CPU emulation of where bf16 rounding enters the ResNet-50 trunk and what each storage choice costs in embedding error against the
fp32 trunk (is an fp32 residual stream enough for the 1e-3 bar?).  Roundings: w (weights), x (input), y (conv outputs
before BatchNorm), a (bottleneck-internal activations), z (block outputs = the residual stream), y3 (conv3's output only).
Usage: python tools/precision_emulation.py [batch=16] [hw=96]"""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from oracle import vince_oracle as vo

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 96
ARCH = "ResNet50"
spec = vo.model_spec(ARCH, 128)
sd0 = vo.seeded_state(spec, 11)
x0 = vo.structured_frames(B, HW, HW, seed=77)


def rb(t, on):
    """on: falsy = keep fp32; 1 / "bf16" = round to bfloat16 (8 significand bits); "f16" = round to IEEE half (11 bits; every tensor of a
    BatchNorm network sits well inside its range -- the emulation reports an overflow if one does not)."""
    if not on:
        return t
    if on == "f16":
        h = t.half()
        assert torch.isfinite(h).all(), "IEEE-half overflow"
        return h.float()
    return t.bfloat16().float()


def bn(sd, p, x):
    return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.1, 1e-5)


def trunk(sd, x, r):
    pre = "feature_extractor.model."
    W = lambda n: rb(sd[n], r["w"])
    x = rb(x, r["x"])
    x = rb(F.conv2d(x, W(pre + "conv1.weight"), None, 2, 3), r["y"])
    x = rb(F.max_pool2d(F.relu(bn(sd, pre + "bn1", x)), 3, 2, 1), r["z"] or r["a"])
    for li, nb in enumerate(vo.ARCH[ARCH]["layers"]):
        for bi in range(nb):
            stride = 2 if (li > 0 and bi == 0) else 1
            p = "%slayer%d.%d." % (pre, li + 1, bi)
            xin = rb(x, r["a"])   # what the convs read is always the compute dtype
            out = rb(F.conv2d(xin, W(p + "conv1.weight")), r["y"])
            out = rb(F.relu(bn(sd, p + "bn1", out)), r["a"])
            out = rb(F.conv2d(out, W(p + "conv2.weight"), None, stride, 1), r["y"])
            out = rb(F.relu(bn(sd, p + "bn2", out)), r["a"])
            out = rb(F.conv2d(out, W(p + "conv3.weight")), r["y3"])
            out = bn(sd, p + "bn3", out)
            idn = x
            if (p + "downsample.0.weight") in sd:
                idn = rb(F.conv2d(xin, W(p + "downsample.0.weight"), None, stride), r["y"])
                idn = bn(sd, p + "downsample.1", idn)
            x = rb(F.relu(out + idn), r["z"])
    return x


HEAD_SHIFT = None      # the centred-head state of fixtures G11c / G12: the head's output bias moved by minus the batch mean of the pre-norm
                       # features, so that the embeddings are spread over the sphere and the L2 normalisation hides nothing


def embed(sd, x, r):
    f = trunk(sd, x, r).mean(dim=(2, 3))
    h = F.relu(F.linear(f, sd["embedding.0.weight"], sd["embedding.0.bias"]))
    pre = F.linear(h, sd["embedding.2.weight"], sd["embedding.2.bias"])
    if HEAD_SHIFT is not None:
        pre = pre + HEAD_SHIFT
    return F.normalize(pre, dim=1), f


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


none = dict(w=False, x=False, y=False, a=False, z=False, y3=False)
with torch.no_grad():
    e_ref, f_ref = embed(sd0, x0, none)
    cases = [("all bf16 (separate passes: w x y a z y3)", dict(w=1, x=1, y=1, a=1, z=1, y3=1)),
             ("bf16, y3 unrounded (fused join)", dict(w=1, x=1, y=1, a=1, z=1, y3=0)),
             ("bf16r32: z fp32 (y3 rounded)", dict(w=1, x=1, y=1, a=1, z=0, y3=1)),
             ("bf16r32 + fused join: z fp32, y3 unrounded", dict(w=1, x=1, y=1, a=1, z=0, y3=0)),
             ("only weights + input bf16", dict(w=1, x=1, y=0, a=0, z=0, y3=0)),
             ("only weights bf16", dict(w=1, x=0, y=0, a=0, z=0, y3=0)),
             ("only y (conv outputs) bf16", dict(w=0, x=0, y=1, a=0, z=0, y3=1)),
             ("only a bf16", dict(w=0, x=0, y=0, a=1, z=0, y3=0)),
             ("only z bf16", dict(w=0, x=0, y=0, a=0, z=1, y3=0)),
             # VERDICT r4 next #2b: IEEE half (fp16 MFMA = the bf16 rate; 11 vs 8 significand bits) for the FORWARD tensors
             ("all IEEE half (w x y a z y3)", dict(w="f16", x="f16", y="f16", a="f16", z="f16", y3="f16")),
             ("IEEE half, y3 unrounded (fused join)", dict(w="f16", x="f16", y="f16", a="f16", z="f16", y3=0)),
             ("IEEE-half activations, bf16 weights", dict(w=1, x="f16", y="f16", a="f16", z="f16", y3=0)),
             ("only weights IEEE half", dict(w="f16", x=0, y=0, a=0, z=0, y3=0))]
    print("ResNet-50 random init, B=%d, %dx%d; cosine between embeddings of different frames: %.4f" %
          (B, HW, HW, float((e_ref @ e_ref.t()).fill_diagonal_(0).sum() / (B * (B - 1)))))
    for state in ("random init", "centred head"):
        if state == "centred head":
            h0 = F.relu(F.linear(f_ref, sd0["embedding.0.weight"], sd0["embedding.0.bias"]))
            HEAD_SHIFT = -F.linear(h0, sd0["embedding.2.weight"], sd0["embedding.2.bias"]).mean(0, keepdim=True)
            e_ref, f_ref = embed(sd0, x0, none)
            print("-- centred head: mean pairwise cosine %.4f" % float((e_ref @ e_ref.t()).fill_diagonal_(0).sum() / (B * (B - 1))))
        for name, r in cases:
            e, f = embed(sd0, x0, {**none, **r})
            print("%-52s embeddings %.3e   pooled features %.3e   min cos %.5f" %
                  (name, rel(e, e_ref), rel(f, f_ref), float((e * e_ref).sum(1).min())))
