"""Diagnostic: the BatchNorm-backward algebra's input gradient at engine scale (rows x w, correlated activations, an upstream gradient with a
per-channel mean), single bf16 matrices against hi + lo parts, measured by the MASKED pixel sums the BatchNorm below reduces -- against fp64.
Usage: python tools/alg_op_probe.py [rows] [w] [gmean]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vince_amd import ops            # noqa: E402
from vince_amd._lib import ConvDesc  # noqa: E402

DEV = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200704
w = int(sys.argv[2]) if len(sys.argv) > 2 else 64
gmean = float(sys.argv[3]) if len(sys.argv) > 3 else 0.02
Co = 4 * w
g0 = torch.Generator(device=DEV).manual_seed(3)
rn = lambda *s: torch.randn(*s, generator=g0, device=DEV)
mix = rn(16, w) * 0.5
a = torch.relu(rn(rows, 16) @ mix + 0.5 * rn(rows, w) + 0.3)                 # fp32 activations (what the x3 forward holds)
W = rn(Co, w) * (2.0 / w) ** 0.5                                             # fp32 master weights
gamma, beta = torch.rand(Co, generator=g0, device=DEV) + 0.5, rn(Co) * 0.2
ident = rn(rows, Co)
G = rn(rows, Co) * 0.1 + gmean
a64, W64 = a.double(), W.double()
y = a64 @ W64.t()
mu, var = y.mean(0), y.var(0, unbiased=False)
invstd = (var + 1e-5).rsqrt()
xhat = (y - mu) * invstd
keep = (xhat * gamma.double() + beta.double() + ident.double()) > 0
gb = (G * keep).bfloat16()                                                     # the stored gated gradient
g = gb.double()
s = gamma.double() * invstd
c1, c2 = g.mean(0), (g * xhat).mean(0)
da_true = (s * (g - c1 - xhat * c2)) @ W64
m2 = (a64 > 0).double()
ref = (da_true * m2).sum(0)
refx = (da_true * m2 * a64).sum(0)
ab = a.bfloat16()


def score(name, da):
    got, gotx = (da.double() * m2).sum(0), (da.double() * m2 * a64).sum(0)
    print("%-34s masked sums: max err / max |ref| %.2e   moment %.2e   field %.2e" % (
        name, float((got - ref).abs().max() / ref.abs().max()), float((gotx - refx).abs().max() / refx.abs().max()),
        float((da.double() - da_true).norm() / da_true.norm())), flush=True)


# the separate passes as the twin runs them: centred bf16 y, dy rounded to bf16, bf16 weights
ycb = (y - mu).float().bfloat16().double()
xh = ycb * invstd
c1b, c2b = g.mean(0), (g * xh).mean(0)
dyb = (s * (g - c1b - xh * c2b)).float().bfloat16()
score("separate passes (emulated)", (dyb.double() @ W.bfloat16().double()).float().bfloat16())
score("separate passes, unrounded da", dyb.double() @ W.bfloat16().double())

d3 = ops.conv_desc(1, rows, 1, w, Co, 1, 1, 0)
R = torch.zeros(Co, 1, w, device=DEV)
ops.conv_wgrad(d3, ab.view(1, rows, 1, w), gb.view(1, rows, 1, Co), R)
colsum = torch.zeros(4, w, device=DEV, dtype=torch.float64)
colsum[0] = a64.sum(0)
gs = torch.zeros(ops.STATS_REPLICAS, Co, 2, device=DEV, dtype=torch.float64)
gs[0, :, 0] = g.sum(0)
for name, wt, split in (("algebra, bf16 W, single matrices", W.bfloat16(), False), ("algebra, fp32 W, single matrices", W, False),
                        ("algebra, bf16 W, hi + lo", W.bfloat16(), True), ("algebra, fp32 W, nq alone hi + lo", W, "nq"),
                        ("algebra, fp32 W, hi + lo", W, True)):
    dg, db = torch.zeros(Co, device=DEV), torch.zeros(Co, device=DEV)
    coef, w2, nr = ops.bn3_bwd_prepare(R.view(Co, w), wt.contiguous(), gs, mu.float(), invstd.float(), gamma, rows, dg, db, colsum=colsum, split=split)
    taps = 3 if split is True else 2
    dd = ConvDesc(N=1, Hi=rows, Wi=1, Ci=Co, Ho=rows, Wo=1, Co=w, sh=1, sw=1, TA=1, TB=taps, dh0=0, dhs=1, dw0=0, dws=0, wt0=0, wta=0,
                  wtb=1, WT=taps, OH=rows, OW=1, osh=1, osw=1, oh0=0, ow0=0)
    da = torch.empty(rows, w, device=DEV, dtype=torch.bfloat16)
    ops.conv_igemm(dd, gb.view(1, rows, 1, Co), w2, da.view(1, rows, 1, w), bias=nr, in2=ab.view(1, rows, 1, w), in2_repeat=2 if split else 0)
    score(name, da)
    w2f = w2.double()
    if split is True:
        full = g @ (w2f[:, 0] + w2f[:, 1]).t() + ab.double() @ (w2f[:, 2, :w] + w2f[:, 2, w:2 * w]).t() + nr.double()
    elif split:
        full = g @ w2f[:, 0].t() + ab.double() @ (w2f[:, 1, :w] + w2f[:, 1, w:2 * w]).t() + nr.double()
    else:
        full = g @ w2f[:, 0].t() + ab.double() @ w2f[:, 1, :w].t() + nr.double()
    score("   ... the same, unrounded da", full)
    print("       coef: c1 err %.1e  c2 err %.1e (rel. to max)" % (float((coef[1].double() - c1).abs().max() / c1.abs().max()),
                                                                    float((coef[2].double() - c2).abs().max() / c2.abs().max())))
