"""Parity at BASELINE.json's FULL sizes (config C3: B=256 per GPU, K=65536, D=128, 224x224 frames) through checks that stay
cheap at that size: the CPU oracle where the full-size problem is a few GFLOP (similarity / InfoNCE, queue index
arithmetic), and size-independent properties for the conv kernels (exact power-of-two homogeneity in bf16, agreement of the
two tile configurations) where a CPU convolution would take minutes."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vince_oracle as vo  # noqa: E402  (test infrastructure)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from vince_amd import ops
    return ops


def unit_rows(n, d, seed):
    return torch.nn.functional.normalize(torch.randn(n, d, generator=torch.Generator().manual_seed(seed)), dim=1)


def test_infonce_full_size_vs_oracle():
    """B=256, K=65536, D=128, T=0.2 (MoCo mode): loss, metrics, per-row distances and dq against the CPU oracle."""
    ops = _ops()
    B, K, D, T = 256, 65536, 128, 0.2
    qq = unit_rows(B, D, 140).requires_grad_(True)
    kk = torch.nn.functional.normalize(qq.detach() + 0.5 * unit_rows(B, D, 141), dim=1)
    queue = unit_rows(K, D, 142)
    sims, mask = vo.similarities(qq, kk, queue, False, 1)
    ld = vo.similarity_cross_entropy(sims, T, mask)
    met = vo.nce_metrics(sims.detach(), mask, ld["softmax_weight"])
    ld["dist"].backward()
    r = ops.infonce_fwd(qq.detach().to(DEV), kk.to(DEV), queue.to(DEV), T, frames=1, offdiag_neg=False)
    sc = r.scalars.cpu().numpy()
    np.testing.assert_allclose(sc[0], float(ld["dist"]), rtol=1e-4)
    np.testing.assert_allclose(sc[2], float(met["nce_accuracy_mean"]), atol=1e-6)
    np.testing.assert_allclose(sc[3], float(met["cosine_sim"]), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(sc[4], float(met["cosine_sim_neg_max"]), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(r.dists.cpu().numpy(), ld["dists"].detach().numpy().reshape(B, 1), rtol=1e-3, atol=1e-4)
    dq = torch.zeros(B, D, device=DEV)
    ops.infonce_bwd(r, qq.detach().to(DEV), kk.to(DEV), queue.to(DEV), torch.tensor([1.0], device=DEV), dq)
    ref = qq.grad
    assert float((dq.cpu() - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-8
    # size-independent property: the loss does not depend on the order of the queue rows (up to fp32 summation order)
    perm = torch.randperm(K, generator=torch.Generator().manual_seed(7))
    r2 = ops.infonce_fwd(qq.detach().to(DEV), kk.to(DEV), queue[perm].to(DEV), T, frames=1, offdiag_neg=False)
    np.testing.assert_allclose(float(r2.scalars[0]), float(r.scalars[0]), rtol=2e-6)


def test_queue_full_size_indices_bit_exact_vs_oracle():
    """K=65536 with the per-step enqueue sizes of 1, 2 and 8 GPUs (256, 512, 2048 rows) plus ragged tails: tail, full flag
    and row ownership equal the oracle's exactly after every enqueue, across several laps of the ring."""
    ops = _ops()
    K = 65536
    queue = torch.full((K, 4), -1.0, device=DEV)
    oq = vo.OracleQueue(K, 4)
    tail, full, nid = 0, False, 0
    sizes = [256] * 100 + [2048] * 40 + [512] * 30 + [100, 65536, 7, 70000, 255, 65535, 1]
    for n in sizes:
        ids = nid + torch.arange(n, dtype=torch.float32)
        items = ids[:, None].repeat(1, 4)
        oq.enqueue(items.numpy())
        tail, full = ops.queue_enqueue(queue, items.to(DEV), tail, full)
        nid += n
        assert tail == oq.current_tail and full == oq.full
    np.testing.assert_array_equal(queue[:, 0].cpu().numpy().astype(np.int64), oq.owner)
    np.testing.assert_array_equal(queue.cpu().numpy(), oq.vectors)


# (name, H = W, Ci, Co, k) at N = 256: layer3's 3x3 and 1x1 (128ch x 256px tile), layer1's 3x3 (64ch x 256px tile)
FULL_LAYERS = [("layer3 3x3", 14, 256, 256, 3), ("layer3 1x1 1024->256", 14, 1024, 256, 1), ("layer1 3x3", 56, 64, 64, 3)]


@pytest.mark.parametrize("name,hw,ci,co,k", FULL_LAYERS)
def test_conv_full_size_homogeneity_and_statistics(name, hw, ci, co, k):
    """Full-batch layers (N=256): conv(4x) == 4*conv(x) and conv(x; w/2) == conv(x)/2 BIT-EXACTLY (powers of two commute
    with bf16 rounding and fp32 accumulation), and the fused BatchNorm statistics equal the sums of the stored output."""
    ops = _ops()
    N = 256
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(N, hw, hw, ci, device=DEV, generator=g).clamp_(min=0).bfloat16()
    w = (torch.randn(co, k * k, ci, device=DEV, generator=g) * (2.0 / (ci * k * k)) ** 0.5).bfloat16()
    d = ops.conv_desc(N, hw, hw, ci, co, k, 1, k // 2)
    outs = []
    for xs, ws in ((x, w), (x * 4, w), (x, w * 0.5)):
        out = torch.empty(N, hw, hw, co, device=DEV, dtype=torch.bfloat16)
        stats = torch.zeros(ops.STATS_REPLICAS, co, 2, device=DEV, dtype=torch.float64)
        ops.conv_igemm(d, xs.contiguous(), ws.contiguous(), out, stats=stats)
        outs.append((out, stats.sum(0)))
    base, st = outs[0]
    assert torch.equal(outs[1][0].float(), base.float() * 4)
    assert torch.equal(outs[2][0].float(), base.float() * 0.5)
    o = base.float().reshape(-1, co).double()
    np.testing.assert_allclose(st[:, 0].cpu().numpy(), o.sum(0).cpu().numpy(), rtol=1e-6, atol=1e-2)
    np.testing.assert_allclose(st[:, 1].cpu().numpy(), (o * o).sum(0).cpu().numpy(), rtol=1e-6, atol=1e-2)
    # a random sample of output pixels against an fp64 dot product of the same bf16 operands
    sel = torch.randint(0, N * hw * hw, (64,), generator=torch.Generator().manual_seed(3))
    xp = torch.nn.functional.pad(x.float().permute(0, 3, 1, 2), (k // 2,) * 4).cpu().double()
    wf = w.float().cpu().double().reshape(co, k, k, ci)
    for pix in sel.tolist():
        n, rem = divmod(pix, hw * hw)
        h, ww = divmod(rem, hw)
        patch = xp[n, :, h:h + k, ww:ww + k].permute(1, 2, 0)            # [k][k][ci]
        want = (wf * patch[None]).sum(dim=(1, 2, 3))
        got = base[n, h, ww].float().cpu().double()
        assert float((got - want).abs().max()) <= 1.5e-2 * float(want.abs().max()) + 1e-3, (name, pix)


def test_full_batch_trunk_eval_folded_vs_separate_passes(monkeypatch):
    """ResNet-50 at the benchmark batch (256 frames of 224 x 224, bf16): the engine at full size -- 256-pixel tiles, 4- and
    16-way statistic replicas, the residual join in the conv epilogue -- through the one comparison that needs no oracle:
    the BatchNorm-folded inference path against the eval-mode forward that runs the BatchNorm passes separately."""
    from vince_amd.config import make_args
    from vince_amd.models import vince_model as vm
    args = make_args(backbone="ResNet50", vince_embedding_size=128, compute_dtype="bf16")
    model = vm.VinceModel(args)
    model.load_state_dict(vo.seeded_state(vo.model_spec("ResNet50", 128, False), 21))
    model.to(DEV)
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(256, 3, 224, 224, device=DEV, generator=g)
    model.train()
    with torch.no_grad():
        model.get_embeddings({"data": x})            # one train-mode pass: realistic running statistics
    outs = {}
    for fold in (True, False):
        monkeypatch.setattr(vm, "FOLD_BN", fold)
        model.eval()
        with torch.no_grad():
            outs[fold] = model.extract_features(x)["extracted_features"].float().cpu()
    a, b = outs[True], outs[False]
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    rel = float((a - b).abs().max() / b.abs().max())
    cos = float(torch.nn.functional.cosine_similarity(a, b, dim=1).min())
    assert rel < 8e-2 and cos > 0.995, (rel, cos)
