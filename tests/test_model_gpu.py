"""GPU parity of the assembled path (VinceModel / VinceQueueModel / StorageQueue / FlatSGD / VinceSolver) against the
reference-generated golden vectors (tests/golden, G3-G6) and, teacher-forced step by step, against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vince_oracle as vo  # noqa: E402

DEV = "cuda:0"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def build(arch, embed, dtype, seed, **kw):
    from vince_amd.config import make_args
    from vince_amd.models.vince_model import VinceModel
    args = make_args(backbone=arch, vince_embedding_size=embed, compute_dtype=dtype, **kw)
    model = VinceModel(args)
    model.load_state_dict(vo.seeded_state(vo.model_spec(arch, embed, kw.get("jigsaw", False)), seed))
    model.to(DEV)
    return args, model


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


# ------------------------------------------------------------------------------------------ G3 / G4 trunk + head
@pytest.mark.parametrize("arch,embed", [("ResNet18", 64), ("ResNet50", 128)])
@pytest.mark.parametrize("hw", [64, 224])
@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("dtype", ["fp32", "x3"])
def test_g3_trunk_head_fp32(arch, embed, hw, train, dtype):
    """fp32 trunk, and the same fp32 tensors with every convolution as split-half products (compute_dtype="x3"), at the same bounds."""
    g = load("g3_trunk.npz")
    p = "%s_%d_%s_" % (arch, hw, "train" if train else "eval")
    _, model = build(arch, embed, dtype, 11)
    model.train(train)
    x = vo.structured_frames(2, hw, hw, seed=500 + hw).to(DEV)
    with torch.no_grad():
        o = model.get_embeddings({"data": x})
    # north_star bar: embeddings within 1e-3 relative of the fp32 reference; fp32 kernels land far inside it
    # (2-image batches leave 8 samples per channel in layer4's BatchNorm, which amplifies rounding ~100x)
    assert rel(o["embeddings"].cpu(), g[p + "embeddings"]) < 5e-4
    assert rel(o["prenorm_features"].cpu(), g[p + "prenorm"]) < 5e-4
    assert rel(o["extracted_features"].cpu(), g[p + "extracted"]) < 5e-4
    sp = o["spatial_features"].float().cpu()
    assert list(sp.shape) == [2, vo.ARCH[arch]["out_channels"], hw // 32, hw // 32]
    if hw == 64:
        assert rel(sp, g[p + "spatial"]) < 5e-4
    sd = model.state_dict()
    for bn in ["feature_extractor.model.bn1", "feature_extractor.model.layer4.1.bn2",
               "feature_extractor.model.layer2.0.downsample.1"]:
        np.testing.assert_allclose(sd[bn + ".running_mean"].cpu().numpy(), g[p + bn + ".running_mean"], rtol=1e-3, atol=1e-5)
        np.testing.assert_allclose(sd[bn + ".running_var"].cpu().numpy(), g[p + bn + ".running_var"], rtol=1e-3, atol=1e-5)
        assert int(sd[bn + ".num_batches_tracked"]) == int(g[p + bn + ".num_batches_tracked"])


@pytest.mark.parametrize("arch,embed", [("ResNet18", 64), ("ResNet50", 128)])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_eval_bn_folding_matches_unfolded(arch, embed, dtype, monkeypatch):
    """the inference path folds every BatchNorm into its convolution (conv + bias [+ ReLU] [+ residual join
    in the epilogue]); it must reproduce the eval-mode forward that runs the BatchNorm passes separately, including after
    a train-mode forward has moved the running statistics (the folded cache has to notice)."""
    from vince_amd.models import vince_model as vm
    _, model = build(arch, embed, dtype, 12)
    x = vo.structured_frames(4, 96, 96, seed=77).to(DEV)
    tol_ = 2e-5 if dtype == "fp32" else 6e-2

    def eval_out(fold):
        monkeypatch.setattr(vm, "FOLD_BN", fold)
        model.eval()
        with torch.no_grad():
            o = model.extract_features(x)
        return {k: o[k].float().cpu() for k in ("extracted_features", "spatial_features")}

    for _ in range(2):
        a, b = eval_out(True), eval_out(False)
        for k in a:
            assert rel(a[k], b[k]) < tol_, (k, rel(a[k], b[k]))
        model.train()
        with torch.no_grad():
            model.get_embeddings({"data": x})    # moves the running statistics


def test_eval_bn_folding_past_the_streaming_joins_descriptor_limit(monkeypatch):
    """which blocks keep conv3 unfolded in the inference cache is a property of the architecture (one cache serves every
    trunk of a model, whatever its batch); a launch whose tensor is past the streaming join's 31-bit offsets runs the implicit-GEMM join
    epilogue on the SAME cache (bn3's scale as out_scale).  Forced here through `xjoin_folded_max_bytes`: a trunk on the fallback and a
    trunk of another batch on the streaming kernel share one cache, and both reproduce the separate-pass eval forward."""
    from vince_amd.models import vince_model as vm
    _, model = build("ResNet50", 128, "bf16", 12)
    model.eval()
    xs = {n: vo.structured_frames(n, 96, 96, seed=70 + n).to(DEV) for n in (4, 2)}

    def out(x, fold):
        monkeypatch.setattr(vm, "FOLD_BN", fold)
        with torch.no_grad():
            return model.extract_features(x)["extracted_features"].float().cpu()

    want = {n: out(x, False) for n, x in xs.items()}
    monkeypatch.setenv("VINCE_KNOBS", "xjoin_folded_max_bytes=1")        # batch 4: every join on the fallback; builds the cache
    a4 = out(xs[4], True)
    monkeypatch.delenv("VINCE_KNOBS")                                       # batch 2: the streaming kernel, same cache
    a2 = out(xs[2], True)
    monkeypatch.setenv("VINCE_KNOBS", "xjoin_folded_max_bytes=1")
    b2 = out(xs[2], True)
    assert rel(a4, want[4]) < 6e-2 and rel(a2, want[2]) < 6e-2 and rel(b2, want[2]) < 6e-2, (rel(a4, want[4]), rel(a2, want[2]), rel(b2, want[2]))
    assert rel(a2, b2) < 2e-2     # the two join kernels round differently, nothing more


def test_x3_folded_inference_weights_past_the_half_range_are_refused():
    """an x3 model's inference cache folds gamma / sqrt(running_var + eps) into the weights BEFORE the IEEE-half split; a
    near-zero running variance (1 / sqrt(1e-5) = 316) can push a folded weight past 255.9 x 2^8 -- the model must say which layer,
    not hand back NaN features (nor, as before this round, finite wrong ones)."""
    _, model = build("ResNet18", 64, "x3", 12)
    x = vo.structured_frames(2, 64, 64, seed=77).to(DEV)
    model.eval()
    with torch.no_grad():
        ok = model.extract_features(x)["extracted_features"]
    assert torch.isfinite(ok).all()
    res = model.feature_extractor.model
    gi = next(i for i, (_, kind, _, bn) in enumerate(res.plan.params) if kind == 1 and bn == 3)
    with torch.no_grad():
        res.bn_nodes()[3].running_var.zero_()          # 1 / sqrt(eps) = 316 ...
        res.trunk_params[gi].data.fill_(8.0)           # ... x gamma 8 x max |w| ~ 0.2: far past 255.9
    model._touch()
    with torch.no_grad(), pytest.raises(RuntimeError, match="IEEE-half range"):
        model.extract_features(x)


@pytest.mark.parametrize("arch,embed", [("ResNet18", 64), ("ResNet50", 128)])
def test_g3_trunk_head_bf16_reported(arch, embed, record_property):
    """bf16 trunk against the fp32 reference.  53 stacked bf16 layers cannot hold 1e-3 on raw embeddings in general
    (SURVEY.md 7.4-4); the miss is reported as a number, and bounded."""
    g = load("g3_trunk.npz")
    p = "%s_224_train_" % arch
    _, model = build(arch, embed, "bf16", 11)
    model.train(True)
    x = vo.structured_frames(2, 224, 224, seed=500 + 224).to(DEV)
    with torch.no_grad():
        o = model.get_embeddings({"data": x})
    err = rel(o["embeddings"].cpu(), g[p + "embeddings"])
    cos = float((o["embeddings"].cpu() * torch.from_numpy(g[p + "embeddings"])).sum(1).min())
    record_property("bf16_embedding_rel_err", err)
    print("bf16 %s: embedding max rel err %.3e, min cosine to reference %.6f" % (arch, err, cos))
    # measured on MI355X, batch of 2 (8-98 samples per BN channel): ResNet18 ~1e-2, ResNet50 ~8e-2, cosine > 0.996
    assert err < 0.2 and cos > 0.99


def test_bf16_vs_fp32_at_a_training_batch(record_property):
    """The benchmark configuration's precision, measured where BatchNorm has a realistic sample count: ResNet-50, 32 frames
    of 224 x 224 (1 568 - 401 408 samples per channel), bf16 trunk against this build's fp32 trunk (itself within 5e-4 of
    the reference goldens), same weights and inputs; InfoNCE loss against the same queue.  Reported, and bounded."""
    from vince_amd.utils import loss_util
    x = (vo.structured_frames(32, 224, 224, seed=901) + 0.25 * vo.gaussian_frames(32, 224, 224, 902)).to(DEV)
    xk = (vo.structured_frames(32, 224, 224, seed=901) + 0.25 * vo.gaussian_frames(32, 224, 224, 903)).to(DEV)
    queue = torch.nn.functional.normalize(torch.randn(4096, 128, generator=torch.Generator().manual_seed(9)), dim=1).to(DEV)
    out = {}
    for dt in ("fp32", "bf16"):
        _, model = build("ResNet50", 128, dt, 11)
        model.train(True)
        with torch.no_grad():
            q = model.get_embeddings({"data": x})["embeddings"]
            k = model.get_embeddings({"data": xk})["embeddings"]
            sims = torch.cat([(q * k).sum(1, keepdim=True), q @ queue.t()], 1)
            mask = torch.zeros_like(sims, dtype=torch.bool)
            mask[:, 0] = True
            loss = loss_util.similarity_cross_entropy(sims.contiguous(), 0.2, sims.shape[0], 1, mask=mask)["dist"]
        out[dt] = (q.cpu(), float(loss))
        del model
    err = rel(out["bf16"][0], out["fp32"][0])
    cos = float((out["bf16"][0] * out["fp32"][0]).sum(1).min())
    loss_rel = abs(out["bf16"][1] / out["fp32"][1] - 1.0)
    record_property("bf16_vs_fp32_embedding_rel_err_b32", err)
    record_property("bf16_vs_fp32_loss_rel_err_b32", loss_rel)
    print("bf16 vs fp32, ResNet50 B=32: embedding max rel err %.3e, min cosine %.6f, InfoNCE loss rel err %.3e"
          % (err, cos, loss_rel))
    # measured on MI355X: embedding max rel err 1.2e-1 (random-init encoder: nearly collapsed embeddings, cosine 0.9946),
    # InfoNCE loss rel err 2.2e-4 -- the loss meets the north-star 1e-3 bar in bf16, raw embedding elements do not
    assert err < 0.25 and cos > 0.99 and loss_rel < 1e-3


# ------------------------------------------------------------------------------------------ training step, teacher forced
def oracle_trainer(mode, lr=0.03):
    return vo.OracleTrainer("ResNet18", 64, 512, 32, 0.07, lr, inter_batch=mode == "vince",
                            num_frames=4 if mode == "vince" else 1, self_batch=mode == "vince", seed=5)


def gpu_stack(mode, lr=0.03, dtype="fp32"):
    from vince_amd.models.vince_model import VinceQueueModel
    from vince_amd.optim import FlatSGD
    from vince_amd.utils.storage_queue import StorageQueue
    args, model = build("ResNet18", 64, dtype, 5, batch_size=32, vince_queue_size=512, vince_temperature=0.07,
                        num_frames=4 if mode == "vince" else 1, inter_batch_comparison=mode == "vince",
                        self_batch_comparison=mode == "vince", base_lr=lr)
    model.train()
    qm = VinceQueueModel(args, model)
    qm.to(DEV)
    qm.train()
    queue = StorageQueue(512, 64, device=DEV)
    opt = FlatSGD(model, lr=lr)
    return args, model, qm, queue, opt


def load_oracle_state(tr, model, qm, queue, opt):
    """Teacher forcing: copy the oracle's complete training state into the GPU objects."""
    model.load_state_dict({k: v.detach() for k, v in tr.q.items()})
    qm.queue_network.load_state_dict({k: v.detach() for k, v in tr.k.items()})
    queue.vector_queue.copy_(torch.from_numpy(tr.queue.vectors))
    queue.current_tail, queue.full = tr.queue.current_tail, tr.queue.full
    names = [n for n in tr.pnames if not n.startswith("feature_extractor.model.fc.")]
    sd = dict(model.named_parameters())
    opt.momentum_buffer.zero_()
    for n in names:
        if n in tr.bufs:
            # momentum buffers live in the flat layout: write through a view shaped like the parameter
            p = sd[n]
            off = (p.data_ptr() - model._flat.data_ptr()) // 4
            view = opt.momentum_buffer[off:off + p.numel()]
            src = tr.bufs[n]
            if p.dim() == 4:
                src = src.permute(0, 2, 3, 1)
            view.copy_(src.reshape(-1))


def gpu_step(args, model, qm, queue, opt, data, qdata, mode):
    F_ = 4 if mode == "vince" else 1
    batch = {"data": data.to(DEV), "queue_data": qdata.to(DEV), "batch_types": ["images"], "batch_sizes": [32],
             "data_source": ["XX"], "num_frames": [F_]}
    qb = qm(batch, shuffle=True)
    outs = model.get_embeddings(batch, shuffle=True)
    ib = model.split_dict_by_type(batch["batch_types"], batch["batch_sizes"], batch)
    output = outs[0]
    output.update(queue.dequeue())
    output.update(ib[0])
    output.update(qb[0])
    output.update(model(output))
    ld = model.loss(output)
    met = model.get_metrics(output)
    total = sum(w * v for w, v in ld.values())
    opt.zero_grad()
    total.backward()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    opt.step()
    queue.enqueue(output["queue_embeddings"], None, "XX")
    qm.vince_update(model)
    return output, ld, met, grads


def step_inputs(it):
    data = vo.structured_frames(32, 64, 64, seed=1000 + it)
    qdata = vo.structured_frames(32, 64, 64, seed=1000 + it) + 0.25 * vo.gaussian_frames(32, 64, 64, 2000 + it)
    return data, qdata


@pytest.mark.parametrize("mode", ["moco", "vince"])
def test_g5_first_step_vs_reference_golden(mode):
    """Iteration 0 of config C1 against the REFERENCE's own numbers (tests/golden/g5_step.npz)."""
    g = load("g5_step.npz")
    stack = gpu_stack(mode)
    args, model, qm, queue, opt = stack
    qi = torch.nn.functional.normalize(torch.randn(512, 64, generator=torch.Generator().manual_seed(5 + 77)), dim=-1)
    queue.vector_queue.copy_(qi)
    data, qdata = step_inputs(0)
    output, ld, met, grads = gpu_step(*stack, data, qdata, mode)
    pre = "%s_it0_" % mode
    np.testing.assert_allclose(float(ld["nce_loss"][1]), float(g[pre + "loss_nce_loss"]), rtol=1e-3)
    if mode == "vince":
        np.testing.assert_allclose(float(ld["nce_loss_self"][1]), float(g[pre + "loss_nce_loss_self"]), rtol=1e-3)
    assert rel(output["embeddings"].detach().cpu(), g[pre + "embeddings"]) < 1e-3
    assert rel(output["queue_embeddings"].cpu(), g[pre + "queue_embeddings"]) < 1e-3
    for kk in ["nce_accuracy_mean", "cosine_sim", "cosine_sim_neg_max"]:
        np.testing.assert_allclose(float(met[kk]), float(g[pre + "m_" + kk]), rtol=1e-3, atol=1e-5)
    assert queue.current_tail == int(g[pre + "tail"]) and queue.full == bool(g[pre + "full"])
    gb = grads["feature_extractor.model.bn1.weight"].cpu().numpy()
    assert rel(gb, g[pre + "grad_bn1w"]) < 2e-2
    for name, key in [("feature_extractor.model.conv1.weight", "grad_conv1"), ("embedding.2.weight", "grad_emb2"),
                      ("feature_extractor.model.layer4.1.conv2.weight", "grad_l4")]:
        cs = vo.tensor_checksum(grads[name].cpu().contiguous())
        np.testing.assert_allclose(cs[2], g[pre + key][2], rtol=2e-2)
    pcs = np.array([vo.tensor_checksum(p.detach().cpu().contiguous())[2] for _, p in model.named_parameters()])
    # Stem-adjacent gradients of a freshly initialised (nearly collapsed) encoder are ill-conditioned: two fp32 summation
    # orders of the SAME stem convolution (outputs equal to 3e-6) move conv1 / bn1 / layer1.0 gradients by ~5e-3
    # element-wise while layer4 and head gradients stay within 1e-5 (tools/debug_stem.py, VINCE_KNOBS=stem_packed=0 vs 1).  The
    # updated conv1 / bn1 parameters are therefore held to 2e-3, everything else to 1e-4.
    names = [n for n, _ in model.named_parameters()]
    stem = np.array([n.startswith("feature_extractor.model.conv1") or n.startswith("feature_extractor.model.bn1") for n in names])
    want = g[pre + "param_checksums"][:, 2]
    np.testing.assert_allclose(pcs[~stem], want[~stem], rtol=1e-4)
    np.testing.assert_allclose(pcs[stem], want[stem], rtol=2e-3)
    kcs = np.array([vo.tensor_checksum(p.detach().cpu().contiguous())[2] for _, p in qm.queue_network.named_parameters()])
    np.testing.assert_allclose(kcs, g[pre + "key_checksums"][:, 2], rtol=1e-5)
    np.testing.assert_allclose(vo.tensor_checksum(queue.vector_queue.cpu()), g[pre + "queue_checksum"], rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize("mode", ["moco", "vince"])
def test_three_steps_teacher_forced_vs_oracle(mode):
    """Every iteration starts from the ORACLE's state (the free-running problem is chaotic, see test_oracle_golden)."""
    tr = oracle_trainer(mode)
    stack = gpu_stack(mode)
    args, model, qm, queue, opt = stack
    for it in range(3):
        load_oracle_state(tr, model, qm, queue, opt)
        data, qdata = step_inputs(it)
        r = tr.step(data, qdata)
        output, ld, met, grads = gpu_step(*stack, data, qdata, mode)
        np.testing.assert_allclose(float(ld["nce_loss"][1]), r["nce_loss"], rtol=1e-3)
        if mode == "vince":
            np.testing.assert_allclose(float(ld["nce_loss_self"][1]), r["nce_loss_self"], rtol=1e-3)
        assert rel(output["embeddings"].detach().cpu(), r["embeddings"]) < 1e-3
        assert rel(output["queue_embeddings"].cpu(), r["queue_embeddings"]) < 1e-3
        for kk in ["nce_accuracy_mean", "cosine_sim", "cosine_sim_neg_max"]:
            np.testing.assert_allclose(float(met[kk]), r[kk], rtol=1e-3, atol=1e-5)
        assert queue.current_tail == r["tail"] and queue.full == r["full"]
        # Gradient conditioning (CPU oracle in fp32 vs fp64, same code): at iteration 0 the head and layer4 gradients agree to
        # 6e-6 and the stem to 2e-3; from iteration 1 on -- the queue holds the encoder's own nearly collapsed keys -- the
        # head still agrees to 2e-4 but layer4 differs by 0.24 and the stem by 0.09.  So: head every iteration, trunk
        # gradients tightly at iteration 0 only, afterwards just a direction check.
        for name in ["embedding.2.weight", "embedding.0.bias"]:
            assert rel(grads[name].cpu(), r["grads"][name]) < 5e-3, name
        for name in ["feature_extractor.model.layer4.1.conv2.weight", "feature_extractor.model.layer4.1.bn2.weight"]:
            got, want = grads[name].cpu().flatten().double(), r["grads"][name].flatten().double()
            if it == 0:
                assert rel(got, want) < 5e-3, name
            else:
                assert float(torch.nn.functional.cosine_similarity(got, want, dim=0)) > 0.9, name
        if it == 0:
            for name in ["feature_extractor.model.conv1.weight", "feature_extractor.model.bn1.weight",
                         "feature_extractor.model.layer2.0.downsample.0.weight"]:
                assert rel(grads[name].cpu(), r["grads"][name]) < 2e-2, name
        # queue contents after this step's enqueue, key encoder after EMA
        np.testing.assert_allclose(queue.vector_queue.cpu().numpy(), tr.queue.vectors, rtol=1e-3, atol=2e-4)
        ksd = qm.queue_network.state_dict()
        for name in ["embedding.2.weight", "feature_extractor.model.layer4.1.conv2.weight", "feature_extractor.model.fc.weight"]:
            assert rel(ksd[name].cpu(), tr.k[name]) < 1e-4


# ------------------------------------------------------------------------------------------ G11: off the freshly initialised state
def _g11_stack(dtype):
    from vince_amd.models.vince_model import VinceQueueModel
    from vince_amd.optim import FlatSGD
    from vince_amd.utils.storage_queue import StorageQueue
    c = vo.G11
    args, model = build(c["arch"], c["embed"], dtype, c["seed"], batch_size=c["B"], vince_queue_size=c["K"], vince_temperature=c["T"],
                        base_lr=c["lr"])
    model.train()
    qm = VinceQueueModel(args, model)
    qm.to(DEV)
    qm.train()
    return args, model, qm, StorageQueue(c["K"], c["embed"], device=DEV), FlatSGD(model, lr=c["lr"])


def _g11_step(stack, data, qdata):
    args, model, qm, queue, opt = stack
    B = vo.G11["B"]
    batch = {"data": data.to(DEV), "queue_data": qdata.to(DEV), "batch_types": ["images"], "batch_sizes": [B], "data_source": ["XX"],
             "num_frames": [1]}
    qb = qm(batch, shuffle=True)
    output = model.get_embeddings(batch, shuffle=True)[0]
    output.update(queue.dequeue())
    output.update(model.split_dict_by_type(batch["batch_types"], batch["batch_sizes"], batch)[0])
    output.update(qb[0])
    output.update(model(output))
    ld = model.loss(output)
    met = model.get_metrics(output)
    opt.zero_grad()
    sum(w * v for w, v in ld.values()).backward()
    grads = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
    return output, float(ld["nce_loss"][1]), met, grads


# bf16 bounds: 1.5 x the value measured on MI355X (the numbers are printed by the tests; DESIGN.md section 3 holds the table)
# measured: centred head -- loss 2.5e-2 .. 3.3e-2, embeddings 5.8e-1 of max|e|, min cosine 0.77 (B=16 at 64x64 leaves 64 samples per
# layer4 BatchNorm channel: the harshest case); after 20 steps -- loss 5.5e-4 (inside the 1e-3 bar), embeddings 4.3e-3, cosine 0.99999
G11_BF16_BOUNDS = {"centred": dict(loss=5e-2, emb=0.9, cos=0.65), "after20": dict(loss=1e-3, emb=7e-3, cos=0.9999)}


@pytest.mark.parametrize("dtype", ["fp32", "x3", "bf16"])
def test_g11_centred_head_vs_reference_golden(dtype):
    """G11c (oracle/make_golden_g11.py): ONE iteration of the REFERENCE from the seeded ResNet-50 whose head bias is shifted so that the
    batch's embeddings are spread over the sphere (mean pairwise cosine -0.06): nothing hides trunk error behind the L2 normalisation.
    fp32 trunk: the north-star bars (loss, embeddings 1e-3).  bf16 trunk: reported, bounded at 1.5 x the measured values."""
    g = load("g11_after20.npz")
    stack = _g11_stack(dtype)
    args, model, qm, queue, opt = stack
    shift = torch.from_numpy(g["c_shift"]).to(DEV)
    with torch.no_grad():
        dict(model.named_parameters())["embedding.2.bias"].add_(shift)
        dict(qm.queue_network.named_parameters())["embedding.2.bias"].add_(shift)
    queue.vector_queue.copy_(vo.g11_queue(78))
    output, loss, met, grads = _g11_step(stack, *vo.g11_inputs(100))
    e_loss = abs(loss / float(g["c_loss"]) - 1.0)
    e_emb = rel(output["embeddings"].detach().cpu(), g["c_embeddings"])
    e_key = rel(output["queue_embeddings"].cpu(), g["c_queue_embeddings"])
    cos = float((output["embeddings"].detach().cpu() * torch.from_numpy(g["c_embeddings"])).sum(1).min())
    names = list(g["c_grad_names"])
    worst = max(abs(vo.tensor_checksum(grads[n])[2] / g["c_grad_checksums"][names.index(n)][2] - 1.0) for n in names if n in grads)
    print("G11 centred head, %s trunk: loss rel %.3e  embeddings %.3e  keys %.3e  min cosine %.5f  worst sum|grad| rel %.3e"
          % (dtype, e_loss, e_emb, e_key, cos, worst))
    if dtype in ("fp32", "x3"):   # x3 (split-half products on fp32 tensors) is held to the SAME bars as the fp32 trunk
        assert e_loss < 1e-3 and e_emb < 1e-3 and e_key < 1e-3
        np.testing.assert_allclose([float(met[k]) for k in ("nce_accuracy_mean", "cosine_sim", "cosine_sim_neg_max")], g["c_metrics"],
                                   rtol=1e-3, atol=1e-4)
        assert worst < 3e-2
        assert rel(grads["embedding.2.weight"][:8], g["c_grad_embedding.2.weight"]) < 5e-3
    else:
        b = G11_BF16_BOUNDS["centred"]
        assert e_loss < b["loss"] and e_emb < b["emb"] and cos > b["cos"]


# ------------------------------------------------------------------------------------------ G13: config 5 at its own size
@pytest.mark.parametrize("coin", ["k", "q"])
@pytest.mark.parametrize("dtype", ["fp32", "x3", "x3f", "bf16"])
def test_g13_config5_multiframe_jigsaw_vs_reference_golden(dtype, coin):
    """G13 (oracle/make_golden_g13.py): BASELINE config 5's per-GPU work at ITS OWN size against the imported
    reference -- ResNet-50, 224 x 224, 4 frames per clip (8 clips), inter-batch + self-batch comparison, D=128, T=0.2 / self-T 0.03,
    K=65536, jigsaw head (the jigsawed side padded to 225 -> 9 tiles of 75 x 75 per frame, models/vince_model.py:144-171), ONE full
    iteration per coin outcome (solvers/vince_solver.py:397-403: "k" = key side jigsawed, "q" = query side, whose backward then runs
    through the jigsaw head and the 9x batch); the reference's per-sample tile orders are inputs.  fp32 / x3: loss terms, both sides'
    embeddings and pre-norm features, metrics at the north-star bar; every gradient's sum |g| and sampled head rows.  bf16: reported."""
    from vince_amd.models.vince_model import VinceQueueModel
    g = load("g13_config5.npz")
    c = vo.G13
    args, model = build(c["arch"], c["embed"], dtype, c["seed"], jigsaw=True, batch_size=c["B"], num_frames=c["F"], vince_queue_size=c["K"],
                        vince_temperature=c["T"], vince_self_temperature=c["self_T"], base_lr=c["lr"], inter_batch_comparison=True,
                        self_batch_comparison=True, input_size=(c["hw"], c["hw"]))
    model.train()
    qm = VinceQueueModel(args, model)
    qm.to(DEV)
    qm.train()
    queue = vo.g13_queue().to(DEV)
    data, qdata = vo.g13_inputs()
    p = coin + "_"
    orders = torch.from_numpy(g[p + "orders"]).to(DEV)
    batch = {"data": data.to(DEV), "queue_data": qdata.to(DEV), "batch_types": ["images"], "batch_sizes": [c["B"]], "data_source": ["XX"],
             "num_frames": [c["F"]], ("queue_jigsaw_orders" if coin == "k" else "jigsaw_orders"): orders}
    qb = qm(batch, jigsaw=(coin == "k"), shuffle=True)
    o = model.get_embeddings(batch, jigsaw=(coin == "q"), shuffle=True)[0]
    o.update({"queue_vectors": queue, "queue_images": None, "queue_data_sources": None})
    o.update(model.split_dict_by_type(batch["batch_types"], batch["batch_sizes"], batch)[0])
    o.update(qb[0])
    o.update(model(o))
    ld = model.loss(o)
    met = model.get_metrics(o)
    model.zero_grad()
    sum(w * v for w, v in ld.values()).backward()
    torch.cuda.synchronize()
    assert sorted(ld) == list(g[p + "loss_names"]) and sorted(met) == list(g[p + "metric_names"])
    terms = np.array([float(ld[k][0] * ld[k][1]) for k in sorted(ld)])
    e_terms = float(np.abs(terms / g[p + "loss_terms"] - 1).max())
    errs = {k: rel(o[k].detach().float().cpu(), g[p + k]) for k in ("embeddings", "prenorm_features", "extracted_features")}
    errs.update({"queue_" + k: rel(qb[0]["queue_" + k].float().cpu(), g[p + "queue_" + k]) for k in ("embeddings", "prenorm_features")})
    named = dict(model.named_parameters())
    names = list(g[p + "grad_names"])
    ratios = {n: abs(vo.tensor_checksum(named[n].grad.detach().float().cpu())[2] / g[p + "grad_checksums"][names.index(n)][2] - 1.0)
              for n in names if named[n].grad is not None}
    # the head the step did not touch: no gradient in the reference's fresh model; here its (flat-buffer) gradient exists and is zero
    for n, q in named.items():
        if q.grad is not None and n not in names:
            assert float(q.grad.abs().max()) == 0.0, n
    assert all(named[n].grad is not None for n in names)
    print("G13 coin %s, %s trunk: loss terms rel %.3e  %s  sum|g| rel err median %.2e worst %.2e (%s)"
          % (coin, dtype, e_terms, "  ".join("%s %.2e" % kv for kv in errs.items()), float(np.median(list(ratios.values()))),
             max(ratios.values()), max(ratios, key=ratios.get)))
    if dtype == "bf16":
        # REPORTED, bounded at 1.5 x what was measured on MI355X: loss terms 1.6e-3 (key side jigsawed) /
        # 3.9e-3 (query side), embeddings 1e-1 / 2e-1 of max |e| -- a bf16 config-5 loss that drifts further is a regression
        assert np.isfinite(e_terms) and e_terms < 6e-3, e_terms
        assert max(errs.values()) < 0.3, errs
        return
    assert e_terms < 1e-3 and max(errs.values()) < 1e-3, (e_terms, errs)
    np.testing.assert_allclose([float(met[k]) for k in sorted(met)], g[p + "metrics"], rtol=2e-3, atol=2e-4)
    if dtype == "x3f":    # the forward is x3's; the mixed-precision (bf16 twin engine) backward: sums of |g| at its own, AMP-grade bound
        assert max(ratios.values()) < 4e-2, (max(ratios, key=ratios.get), max(ratios.values()))
        return
    head = "jigsaw_embedding.2.weight" if coin == "q" else "embedding.2.weight"
    assert rel(named[head].grad[:8].cpu(), g[p + "grad_" + head]) < 5e-3
    early = ("feature_extractor.model.conv1", "feature_extractor.model.bn1", "feature_extractor.model.layer1")
    bad = [(n, v) for n, v in ratios.items() if not v < (5e-2 if n.startswith(early) else 2e-2)]
    assert not bad, bad


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_g11_after_twenty_sgd_steps_vs_oracle(dtype):
    """The state after 20 SGD iterations of G11's recipe, reached by replaying them with the CPU oracle (pinned to the reference's
    trajectory by tests/test_oracle_golden.py::test_g11_twenty_sgd_steps_trajectory -- the 100 MB state itself is not a fixture);
    iteration 20 then runs teacher-forced on the GPU against the oracle's iteration 20, and lands inside the reference's record."""
    g = load("g11_after20.npz")
    c = vo.G11
    tr = vo.OracleTrainer(c["arch"], c["embed"], c["K"], c["B"], c["T"], c["lr"], seed=c["seed"])
    for it in range(20):
        tr.step(*vo.g11_inputs(it))
    stack = _g11_stack(dtype)
    load_oracle_state(tr, *stack[1:])
    r = tr.step(*vo.g11_inputs(20))
    output, loss, met, grads = _g11_step(stack, *vo.g11_inputs(20))
    e_loss = abs(loss / r["nce_loss"] - 1.0)
    e_emb = rel(output["embeddings"].detach().cpu(), r["embeddings"])
    cos = float((output["embeddings"].detach().cpu() * r["embeddings"]).sum(1).min())
    e_head = rel(grads["embedding.2.weight"], r["grads"]["embedding.2.weight"])
    print("G11 after 20 SGD steps, %s trunk: loss rel %.3e  embeddings %.3e  min cosine %.5f  head gradient %.3e  (reference loss %.5f)"
          % (dtype, e_loss, e_emb, cos, e_head, float(g["trajectory"][20, 0])))
    assert abs(loss / float(g["trajectory"][20, 0]) - 1.0) < 3e-2          # the reference's own record (chaotic band)
    if dtype == "fp32":
        assert e_loss < 1e-3 and e_emb < 1e-3 and e_head < 1e-2
    else:
        b = G11_BF16_BOUNDS["after20"]
        assert e_loss < b["loss"] and e_emb < b["emb"] and cos > b["cos"]


class _G5Source:
    """batch_source for VinceSolver: the G5 / oracle inputs of iteration `it` in the reference's loader-output layout
    (vince_solver.py:191-199,215-223)."""

    def __init__(self, mode):
        self.mode, self.it = mode, 0

    def __call__(self, loader_id=0):
        data, qdata = step_inputs(self.it)
        self.it += 1
        return {"data": data, "queue_data": qdata, "batch_type": "images", "batch_size": 32, "data_source": "XX",
                "num_frames": 4 if self.mode == "vince" else 1}


def _g5_solver(mode, overlap, dtype="fp32"):
    from vince_amd.config import make_args
    from vince_amd.solvers.vince_solver import VinceSolver
    src = _G5Source(mode)
    args = make_args(backbone="ResNet18", vince_embedding_size=64, compute_dtype=dtype, batch_size=32, vince_queue_size=512,
                     vince_temperature=0.07, num_frames=4 if mode == "vince" else 1, inter_batch_comparison=mode == "vince",
                     self_batch_comparison=mode == "vince", base_lr=0.03, input_size=(64, 64), batch_source=src,
                     log_frequency=1, save=False)
    solver = VinceSolver(args)           # (setup_model's fill_queue_repeat consumes one batch; the state is replaced below)
    solver.overlap_key_encoder = bool(overlap)
    solver.reset_epoch()
    captured = []
    inner = solver.model.get_embeddings

    def spy(*a, **k):
        out = inner(*a, **k)
        captured.append(out)
        return out
    solver.model.get_embeddings = spy
    return solver, src, captured


@pytest.mark.parametrize("overlap", [0, 1])
@pytest.mark.parametrize("mode", ["moco", "vince"])
def test_solver_own_step_vs_reference_golden_and_oracle(mode, overlap):
    """SURVEY 8 row a13 (VERDICT r1 missing #1): `VinceSolver.run_train_iteration` ITSELF -- side-stream key encoder,
    record_stream hand-offs, dequeue-before-enqueue, SGD then enqueue then EMA (solvers/vince_solver.py:386-518) -- against the
    reference's G5 numbers at iteration 0 and against the oracle for 3 teacher-forced iterations, with the key encoder on
    its own stream and inline."""
    g = load("g5_step.npz")
    tr = oracle_trainer(mode)
    solver, src, captured = _g5_solver(mode, overlap)
    model, qm, queue, opt = solver.model, solver.queue_model, solver.vince_queue, solver.optimizer
    for it in range(3):
        load_oracle_state(tr, model, qm, queue, opt)
        src.it = it
        data, qdata = step_inputs(it)
        r = tr.step(data, qdata)
        it0 = solver.iteration
        ld, met = solver.run_train_iteration()
        torch.cuda.synchronize()
        assert solver.iteration == it0 + 32                                   # vince_solver.py:514: counts samples
        np.testing.assert_allclose(float(ld["nce_loss"]), r["nce_loss"], rtol=1e-3)
        if mode == "vince":
            np.testing.assert_allclose(float(ld["nce_loss_self"]), r["nce_loss_self"], rtol=1e-3)
        for kk in ["nce_accuracy_mean", "cosine_sim", "cosine_sim_neg_max"]:
            np.testing.assert_allclose(float(met[kk]), r[kk], rtol=1e-3, atol=1e-5)
        out = captured[-1][0]
        assert rel(out["embeddings"].detach().cpu(), r["embeddings"]) < 1e-3
        assert rel(out["queue_embeddings"].cpu(), r["queue_embeddings"]) < 1e-3
        # the queue was READ before this step's keys were written, and written before the EMA
        assert queue.current_tail == r["tail"] and queue.full == r["full"]
        np.testing.assert_allclose(queue.vector_queue.cpu().numpy(), tr.queue.vectors, rtol=1e-3, atol=2e-4)
        named = dict(model.named_parameters())
        for name in ["embedding.2.weight", "embedding.0.bias"]:
            assert rel(named[name].grad.cpu(), r["grads"][name]) < 5e-3, name
        if it == 0:
            for name in ["feature_extractor.model.layer4.1.conv2.weight", "feature_extractor.model.bn1.weight"]:
                assert rel(named[name].grad.cpu(), r["grads"][name]) < 2e-2, name
        # parameters after SGD (query) and after the EMA that FOLLOWS it (key): theta_k*m + (1-m)*theta_q_new
        # (stem-adjacent gradients are conditioned to ~5e-3 element-wise, see test_g5_first_step...: looser bound for layer1)
        for name, tol0 in [("embedding.2.weight", 1e-4), ("feature_extractor.model.layer4.1.conv2.weight", 1e-4),
                           ("feature_extractor.model.layer1.0.bn1.bias", 1e-3)]:
            assert rel(named[name].detach().cpu(), tr.q[name].detach()) < (tol0 if it == 0 else 3e-3), name
        ksd = qm.queue_network.state_dict()
        for name in ["embedding.2.weight", "feature_extractor.model.layer4.1.conv2.weight", "feature_extractor.model.fc.weight"]:
            assert rel(ksd[name].cpu(), tr.k[name]) < 1e-4, name
        if it == 0:   # the reference's own run (tests/golden/g5_step.npz, iteration 0 starts from the same seeded state)
            pre = "%s_it0_" % mode
            np.testing.assert_allclose(float(ld["nce_loss"]), float(g[pre + "loss_nce_loss"]), rtol=1e-3)
            assert rel(out["embeddings"].detach().cpu(), g[pre + "embeddings"]) < 1e-3
            assert queue.current_tail == int(g[pre + "tail"]) and queue.full == bool(g[pre + "full"])
            kcs = np.array([vo.tensor_checksum(p.detach().cpu().contiguous())[2] for _, p in qm.queue_network.named_parameters()])
            np.testing.assert_allclose(kcs, g[pre + "key_checksums"][:, 2], rtol=1e-5)
            np.testing.assert_allclose(vo.tensor_checksum(queue.vector_queue.cpu()), g[pre + "queue_checksum"], rtol=1e-3, atol=1e-2)
            names = [n for n, _ in model.named_parameters()]
            stem = np.array([n.startswith("feature_extractor.model.conv1") or n.startswith("feature_extractor.model.bn1") for n in names])
            pcs = np.array([vo.tensor_checksum(p.detach().cpu().contiguous())[2] for _, p in model.named_parameters()])
            want = g[pre + "param_checksums"][:, 2]
            np.testing.assert_allclose(pcs[~stem], want[~stem], rtol=1e-4)
            np.testing.assert_allclose(pcs[stem], want[stem], rtol=2e-3)


@pytest.mark.parametrize("mode", ["moco", "vince"])
def test_solver_free_running_two_steps_vs_reference_golden(mode):
    """Two consecutive `run_train_iteration` calls with NO state reload in between, against the reference's free-running
    G5 losses / tails at iterations 0 and 1: an ordering slip (enqueue before dequeue, EMA before enqueue, a missed stream
    join) shows up in the second step's loss, which depends on the queue and key encoder the first step left behind."""
    g = load("g5_step.npz")
    tr = oracle_trainer(mode)
    solver, src, captured = _g5_solver(mode, overlap=1)
    load_oracle_state(tr, solver.model, solver.queue_model, solver.vince_queue, solver.optimizer)
    src.it = 0
    for it in range(2):
        ld, met = solver.run_train_iteration()
        pre = "%s_it%d_" % (mode, it)
        # iteration 1 inherits iteration 0's rounding through SGD / EMA / queue: the reference's own fp32-vs-fp64 band
        # there is ~1e-3 on the loss (oracle/README in DESIGN.md section 3), so 5e-3
        np.testing.assert_allclose(float(ld["nce_loss"]), float(g[pre + "loss_nce_loss"]), rtol=1e-3 if it == 0 else 5e-3)
        assert solver.vince_queue.current_tail == int(g[pre + "tail"]) and solver.vince_queue.full == bool(g[pre + "full"])
        np.testing.assert_allclose(float(met["cosine_sim"]), float(g[pre + "m_cosine_sim"]), rtol=5e-3, atol=1e-4)
    np.testing.assert_allclose(vo.tensor_checksum(solver.vince_queue.vector_queue.cpu()), g["%s_it1_queue_checksum" % mode],
                               rtol=5e-3, atol=5e-2)


# ------------------------------------------------------------------------------------------ G6 jigsaw
@pytest.mark.parametrize("hw", [66, 64])
def test_g6_jigsaw(hw):
    g = load("g6_jigsaw.npz")
    p = "hw%d_" % hw
    _, model = build("ResNet18", 64, "fp32", 21, jigsaw=True)
    model.train()
    x = vo.structured_frames(2, hw, hw, seed=900 + hw).to(DEV)
    with torch.no_grad():
        o = model.get_embeddings({"data": x, "jigsaw_orders": torch.from_numpy(g[p + "orders"])}, jigsaw=True)
    assert rel(o["embeddings"].cpu(), g[p + "embeddings"]) < 1e-3
    assert rel(o["extracted_features"].cpu(), g[p + "extracted"]) < 1e-3


def test_jigsaw_backward_vs_oracle():
    sd = vo.seeded_state(vo.model_spec("ResNet18", 64, jigsaw=True), 21)
    pn = vo.param_names(vo.model_spec("ResNet18", 64, jigsaw=True))
    for n in pn:
        sd[n].requires_grad_(True)
    x = vo.structured_frames(2, 66, 66, seed=966)
    orders = torch.stack([torch.randperm(9, generator=torch.Generator().manual_seed(s)) for s in (1, 2)])
    o = vo.get_embeddings(sd, x, "ResNet18", True, jigsaw=True, jigsaw_orders=orders)
    w = torch.randn(2, 64, generator=torch.Generator().manual_seed(3))
    (o["embeddings"] * w).sum().backward()
    _, model = build("ResNet18", 64, "fp32", 21, jigsaw=True)
    model.train()
    out = model.get_embeddings({"data": x.to(DEV), "jigsaw_orders": orders}, jigsaw=True)
    model.zero_grad()
    (out["embeddings"] * w.to(DEV)).sum().backward()
    named = dict(model.named_parameters())
    for name in ["jigsaw_embedding.2.weight", "jigsaw_embedding.0.weight", "jigsaw_linear.weight",
                 "feature_extractor.model.layer4.1.conv2.weight", "feature_extractor.model.conv1.weight"]:
        assert rel(named[name].grad.cpu(), sd[name].grad) < 2e-2, name
    assert model._touched["jigsaw"] and not model._touched["embedding"]


# ------------------------------------------------------------------------------------------ API surface / solver
def test_materialised_similarity_path_matches_fused():
    from vince_amd.utils.loss_util import similarity_cross_entropy
    stack = gpu_stack("vince")
    args, model, qm, queue, opt = stack
    data, qdata = step_inputs(0)
    batch = {"data": data.to(DEV), "queue_data": qdata.to(DEV), "batch_types": ["images"], "batch_sizes": [32],
             "data_source": ["XX"], "num_frames": [4]}
    qb = qm(batch)
    out = model.get_embeddings(batch)[0]
    out.update(queue.dequeue()); out.update(model.split_dict_by_type(["images"], [32], batch)[0]); out.update(qb[0])
    out.update(model(out))
    fused = model.loss(out)["nce_loss"][1]
    sims = out["vince_similarities"].materialize()
    assert list(sims.shape) == [32, 32 + 512]
    res = similarity_cross_entropy(sims, 0.07, 32, 1, out["vince_similarities_mask"])
    np.testing.assert_allclose(float(res["dist"]), float(fused), rtol=1e-4)
    # and its gradient agrees with the oracle's autograd
    s_cpu = sims.cpu().clone().requires_grad_(True)
    mask = vo.positive_mask(32, 4, 544)
    vo.similarity_cross_entropy(s_cpu, 0.07, mask)["dist"].backward()
    sg = sims.clone().requires_grad_(True)
    similarity_cross_entropy(sg, 0.07, 32, 1, mask.to(DEV))["dist"].backward()
    assert rel(sg.grad.cpu(), s_cpu.grad) < 1e-4


def test_solver_runs_and_advances_queue():
    from vince_amd.config import make_args
    from vince_amd.solvers.vince_solver import VinceSolver
    args = make_args(backbone="ResNet18", batch_size=16, vince_queue_size=64, input_size=(64, 64), compute_dtype="bf16",
                     iterations_per_epoch=3)
    solver = VinceSolver(args)
    solver.reset_epoch()
    losses = []
    for _ in range(5):
        ld, met = solver.run_train_iteration()
        losses.append(float(ld["nce_loss"]))
    assert all(np.isfinite(losses))
    assert solver.vince_queue.current_tail == (5 * 16) % 64 and solver.vince_queue.full
    assert solver.iteration == 5 * 16
    assert set(met.keys()) == {"nce_accuracy_mean", "nce_softmax_weight_mean", "cosine_sim", "cosine_sim_neg_max"}


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_deferred_stem_join_steps_the_same_parameters(dtype, monkeypatch):
    """The solver's default single-process step leaves backward before conv1's weight gradient has landed and steps / averages
    conv1.weight behind an event (engine deferred stem join, FlatSGD.step(defer_stem=True), VinceQueueModel.param_update).
    (1) Inside one run: after the first step the gradient buffer holds the COMPLETE stem gradient g, so conv1.weight must equal
    w - lr (g + wd w) (torch SGD's first step, momentum buffer empty) and the key encoder's copy m k + (1 - m) w_new -- an optimiser
    that ran before the gradient had landed, or an EMA before the step, misses by a visible fraction of the update.
    (2) fp32: the joined order (VINCE_DEFER_STEM=0) from the same seeds gives the same parameters (float atomics: a tolerance; the bf16
    backward is not run-to-run reproducible at this toy size -- its stem gradient varies by percents -- so (1) carries that dtype).
    Later steps only check that nothing is left pending; the free-running golden tests (test_solver_free_running_*) hold the same
    default step to the reference over two iterations."""
    from vince_amd.config import make_args
    from vince_amd.data_source import SyntheticFrames
    from vince_amd.solvers.vince_solver import VinceSolver
    lr, wd, m = 0.03, 1e-4, 0.999

    def run(defer):
        monkeypatch.setenv("VINCE_DEFER_STEM", "1" if defer else "0")
        torch.manual_seed(11)
        args = make_args(backbone="ResNet18", batch_size=16, vince_queue_size=64, input_size=(64, 64), compute_dtype=dtype,
                         batch_source=SyntheticFrames(16, 64, 64, 1, device=DEV, seed=5, iterations=10 ** 6), base_lr=lr)
        solver = VinceSolver(args)
        assert solver.defer_stem == defer and solver.model.defer_stem_join == defer
        solver.reset_epoch()
        conv1 = solver.model.feature_extractor.model.trunk_params[0]
        kconv1 = solver.queue_model.queue_network.feature_extractor.model.trunk_params[0]
        w0, k0 = conv1.detach().clone(), kconv1.detach().clone()
        mom = float(solver.queue_model.vince_momentum)
        solver.run_train_iteration()
        assert solver.model._deferred_step is None and not solver.model._stem_pending     # nothing left for later
        torch.cuda.synchronize()
        g = conv1.grad.detach().clone()            # (a strided view into the flat gradient buffer) complete by now
        want = w0 - lr * (g + wd * w0)
        moved = float((want - w0).abs().max())
        assert moved > 0
        assert float((conv1.detach() - want).abs().max()) < 1e-3 * moved + 1e-9, "conv1.weight was stepped with a partial gradient"
        kwant = mom * k0 + (1.0 - mom) * want
        assert float((kconv1.detach() - kwant).abs().max()) < 1e-3 * (1.0 - mom) * moved + 1e-9, "the key encoder averaged a stale conv1.weight"
        # the compute-dtype weight copies were rebuilt ahead of the next forward in two parts (everything but conv1 beside the stem's
        # weight gradient, conv1 behind its step): they must equal a fresh full rebuild from the stepped parameters, byte for byte
        for net in (solver.model, solver.queue_model.queue_network):
            if defer:
                assert net._wcache_version == net._param_version, "the early rebuild did not declare the cache current"
            have = net._wcache.clone()
            next(iter(net._trunks.values())).prepare_weights(net._param_ptrs, net._wcache)
            torch.cuda.synchronize()
            if defer:
                assert torch.equal(have, net._wcache), "weight cache rebuilt in parts differs from a full rebuild"
        q = {n: p.detach().float().cpu().clone() for n, p in solver.model.named_parameters()}
        k = {n: p.detach().float().cpu().clone() for n, p in solver.queue_model.queue_network.named_parameters()}
        for _ in range(3):
            ld, _ = solver.run_train_iteration()
            assert solver.model._deferred_step is None and not solver.model._stem_pending
        assert np.isfinite(float(ld["nce_loss"]))
        return q, k

    qa, ka = run(True)
    qb, kb = run(False)
    if dtype == "fp32":
        for name in qa:
            assert float((qa[name] - qb[name]).abs().max()) / (float(qb[name].abs().max()) + 1e-12) < 2e-5, name
            assert float((ka[name] - kb[name]).abs().max()) / (float(kb[name].abs().max()) + 1e-12) < 2e-5, name


def test_solver_c5_mode_jigsaw_multiframe_and_val():
    """Config C5's mode at toy size: 4 frames per clip, inter-batch + self-batch positives, jigsaw head, then run_val."""
    from vince_amd.config import make_args
    from vince_amd.data_source import SyntheticFrames
    from vince_amd.solvers.vince_solver import VinceSolver
    val = SyntheticFrames(16, 64, 64, 4, device=DEV, seed=77, iterations=2)
    g = torch.Generator().manual_seed(3)
    knn_set = {"data": torch.randint(0, 256, (48, 3, 64, 64), generator=g).float(), "labels": torch.randint(0, 10, (48,), generator=g)}
    args = make_args(backbone="ResNet18", batch_size=16, vince_queue_size=64, input_size=(64, 64), compute_dtype="bf16",
                     num_frames=4, inter_batch_comparison=True, self_batch_comparison=True, jigsaw=True,
                     val_batch_source=[val], knn_dataset=knn_set)
    solver = VinceSolver(args)
    solver.reset_epoch()
    seen = set()
    for _ in range(6):
        ld, met = solver.run_train_iteration()
        assert np.isfinite(float(ld["nce_loss"])) and np.isfinite(float(ld["nce_loss_self"]))
        seen.add((solver.model._touched["embedding"], solver.model._touched["jigsaw"]))
    assert seen <= {(True, False), (False, True)}            # exactly one head trained per step (vince_solver.py:397-403)
    assert set(met.keys()) == {"nce_accuracy_mean", "nce_softmax_weight_mean", "cosine_sim", "cosine_sim_neg_max",
                               "nce_accuracy_self_mean", "nce_softmax_weight_self_mean", "cosine_self_sim"}
    out = solver.run_val()
    assert np.isfinite(out["nce_loss"]) and 0.0 <= out["nce_accuracy_mean"] <= 1.0
    assert 0.0 <= out["epoch_knn_cifar"] <= 1.0                # the labelled-set k-NN score of vince_solver.py:651-679
    assert solver.model.training                                # run_val restores train mode


def test_knn_eval_matches_kdtree():
    """brute-force GEMM + top-k on the GPU against the reference's recipe (sklearn KDTree, k=11, drop the
    self match, scipy mode) on clustered unit vectors."""
    import scipy.stats
    from sklearn.neighbors import KDTree
    from vince_amd.utils.knn_eval import knn_accuracy, knn_indices
    g = torch.Generator().manual_seed(11)
    n, d, ncls = 3001, 64, 10
    centers = torch.randn(ncls, d, generator=g)
    labels = torch.randint(0, ncls, (n,), generator=g)
    feats = torch.nn.functional.normalize(centers[labels] + 0.9 * torch.randn(n, d, generator=g), dim=1)
    kdt = KDTree(feats.numpy().astype(np.float64), leaf_size=40, metric="euclidean")
    ref_d, ref_i = kdt.query(feats.numpy().astype(np.float64), k=11)
    ref_pred = scipy.stats.mode(labels.numpy()[ref_i[:, 1:]], axis=1, keepdims=True)[0].squeeze(1)
    ref_acc = float(np.mean(ref_pred == labels.numpy()))
    acc, preds, nbrs = knn_accuracy(feats.to(DEV), labels.to(DEV), k=10)
    idx = knn_indices(feats.to(DEV), 11).cpu().numpy()
    assert (idx[:, 0] == np.arange(n)).all()                   # the self match comes first, as KDTree returns it
    # identical neighbour sets except where two distances tie to fp32 rounding at the k-th place
    same = np.array([set(a) == set(b) for a, b in zip(idx, ref_i)])
    assert same.mean() > 0.999
    assert (preds.cpu().numpy() == ref_pred).mean() > 0.999
    assert abs(acc - ref_acc) < 1e-3


def test_bucketed_allreduce_machinery_single_rank():
    """The data-parallel gradient path (engine bucket events -> side stream -> RCCL all-reduce -> SGD grad_scale) on one
    GPU with a single-rank nccl group: results must equal the plain path."""
    import torch.distributed as dist
    from vince_amd.config import make_args
    from vince_amd.data_source import SyntheticFrames
    from vince_amd.solvers.vince_solver import VinceSolver

    def run(force, shuffle_bn=False, defer=True, backbone="ResNet18", dtype="fp32", steps=2):
        torch.manual_seed(0)
        if force:
            os.environ["VINCE_FORCE_DP"] = "1"
        else:
            os.environ.pop("VINCE_FORCE_DP", None)
        os.environ["VINCE_DEFER_STEM"] = "1" if defer else "0"
        args = make_args(backbone=backbone, batch_size=16, vince_queue_size=64, input_size=(64, 64), compute_dtype=dtype,
                         batch_source=SyntheticFrames(16, 64, 64, 1, device=DEV, seed=5), dp_shuffle_bn=shuffle_bn)
        solver = VinceSolver(args)
        solver.model.load_state_dict(vo.seeded_state(vo.model_spec(backbone, 64), 2))
        solver.queue_model.queue_network.load_state_dict(vo.seeded_state(vo.model_spec(backbone, 64), 2))
        solver.vince_queue.vector_queue.copy_(torch.nn.functional.normalize(
            torch.randn(64, 64, generator=torch.Generator().manual_seed(1)), dim=1))
        solver.reset_epoch()
        assert solver.defer_stem == defer
        losses = []
        for _ in range(steps):
            losses.append(float(solver.run_train_iteration()[0]["nce_loss"]))
            # the data-parallel deferred-stem path (dp.GradientReducer.done_most -> model._late -> FlatSGD splitting the step at the last
            # bucket's end -> VinceQueueModel.param_update through _deferred_split) leaves nothing pending behind an iteration
            assert solver.model._late is None and solver.model._deferred_step is None and solver.model._deferred_split is None
            assert not solver.model._stem_pending
        if steps == 1:
            return losses, solver.model._flat.clone(), solver.model._flat_grad.clone()
        return losses, solver.model._flat.clone(), solver.reducer is not None

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        l0, p0, r0 = run(False)
        l1, p1, r1 = run(True)
        l2, p2, _ = run(True, shuffle_bn=True)
        l3, p3, _ = run(True, defer=False)       # the same data-parallel step with the joined order: last bucket stepped with the rest
        # the mixed mode: the bucket events fire from the bf16 TWIN's backward (ResNet-50: the bottleneck tails' algebra route included)
        # (ONE iteration, gradients compared: this start -- loss 1.7e-3, queue of repeats -- has gradients that are the residue of
        # cancellations, 1.8e-2 apart between two plain bf16-grade backwards and chaotic from the second step on, tools/dp_x3f_probe.py;
        # a bucket that was skipped or read early is an O(1) difference)
        l4, p4, g4 = run(False, backbone="ResNet50", dtype="x3f", steps=1)
        l5, p5, g5 = run(True, backbone="ResNet50", dtype="x3f", steps=1)
    finally:
        os.environ.pop("VINCE_FORCE_DP", None)
        os.environ.pop("VINCE_DEFER_STEM", None)
        dist.destroy_process_group()
        from vince_amd._lib import lib
        lib().vince_set_side_streams(2)      # the data-parallel solver lowered the engine's stream budget process-wide
    assert not r0 and r1
    # (two steps only: fp32 atomics make weight gradients order-dependent in the last bits and a freshly initialised
    # encoder amplifies that chaotically from the third step on)
    # (bounds: several times the run-to-run spread of the PLAIN step -- tools/dp_x3f_probe.py: 1e-3 of the parameters after two steps of a
    # freshly initialised encoder; a bucket skipped or read early is an O(1e-1) difference)
    np.testing.assert_allclose(l1, l0, rtol=1e-3, atol=1e-7)
    assert rel(p1.cpu(), p0.cpu()) < 1e-2
    # deferred (stem + layer1 bucket stepped last, behind the stem event) against joined, both data parallel: the same parameters
    np.testing.assert_allclose(l3, l1, rtol=1e-3, atol=1e-7)
    assert rel(p3.cpu(), p1.cpu()) < 1e-2
    # cross-rank shuffle-BN (dp_shuffle_bn): on one rank the key batch is only re-ordered, so the key BatchNorm statistics,
    # the un-permuted keys and hence the losses are unchanged up to summation order
    np.testing.assert_allclose(l2, l0, rtol=1e-3, atol=1e-6)
    assert rel(p2.cpu(), p0.cpu()) < 2e-2
    np.testing.assert_allclose(l5, l4, rtol=1e-5, atol=1e-7)
    assert rel(p5.cpu(), p4.cpu()) < 1e-4 and rel(g5.cpu(), g4.cpu()) < 6e-2


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_uint8_input_stage_equals_float_frames(dtype):
    """SURVEY 8(f)-3, deterministic part: raw uint8 HWC frames with per-frame crop windows and flips through the fused input
    stage (crop + flip + ToTensor/Normalize + stem layout in one kernel) give bit-identical trunk features in eval mode --
    and embeddings / parameter gradients equal up to the atomics' summation order -- to feeding the float NCHW tensor those operations
    produce (same arithmetic, same kernels downstream)."""
    from vince_amd.models.vince_model import U8Frames
    _, model = build("ResNet18", 64, dtype, 31)
    g = torch.Generator().manual_seed(8)
    n, hs, ws, h, w = 6, 80, 96, 64, 64
    frames = torch.randint(0, 256, (n, hs, ws, 3), generator=g, dtype=torch.uint8).to(DEV)
    crop = torch.stack([torch.randint(0, hs - h + 1, (n,), generator=g), torch.randint(0, ws - w + 1, (n,), generator=g)], 1).to(DEV)
    flip = (torch.rand(n, generator=g) < 0.5).to(DEV)
    u8 = U8Frames(frames, (h, w), crop, flip)
    ref = u8.float_reference()
    model.eval()
    with torch.no_grad():
        a = model.get_embeddings({"data": u8})
        b = model.get_embeddings({"data": ref})
    # the trunk is deterministic in eval mode: identical input layout -> bit-identical pooled features; the projection head's
    # split-K GEMM accumulates with fp32 atomics, so embeddings agree to rounding only
    # (fp32: torch's GPU division and the kernel's correctly rounded one may differ in the last bit of an input value)
    if dtype == "bf16":
        assert torch.equal(a["extracted_features"], b["extracted_features"])
    assert rel(a["extracted_features"].cpu(), b["extracted_features"].cpu()) < 1e-5
    assert rel(a["embeddings"].cpu(), b["embeddings"].cpu()) < 1e-5
    # the staged values themselves against a CPU evaluation of (u8 - mean) / std
    cpu_ref = U8Frames(frames.cpu(), (h, w), crop.cpu(), flip.cpu()).float_reference()
    assert rel(ref.cpu(), cpu_ref) < 1e-6
    model.train()
    outs = []
    for data in (u8, ref):
        model.zero_grad()
        e = model.get_embeddings({"data": data})["embeddings"]
        (e * torch.linspace(-1, 1, e.numel(), device=DEV).view_as(e)).sum().backward()
        outs.append((e.detach().clone(), model._flat_grad.clone()))
    # train mode: BatchNorm statistics and weight gradients accumulate with atomics -> equal up to summation order
    assert rel(outs[0][0].cpu(), outs[1][0].cpu()) < (1e-5 if dtype == "fp32" else 2e-2)
    assert rel(outs[0][1].cpu(), outs[1][1].cpu()) < (1e-3 if dtype == "fp32" else 5e-2)


def test_solver_runner_cli_end_to_end(capsys):
    """`python -m vince_amd.solver_runner <reference flags>`: argument parsing, solver construction, the epoch loop with
    the 500-iteration warm-up, validation hook and the final save switch, on the built-in synthetic source."""
    from vince_amd import solver_runner
    solver_runner.main(["--backbone", "ResNet18", "--batch-size", "16", "--vince-queue-size", "64", "--input-width", "64",
                        "--input-height", "64", "--epochs", "2", "--iterations-per-epoch", "4", "--base-lr", "0.03",
                        "--no-save", "--no-restore", "--log-frequency", "2", "--compute-dtype", "bf16", "--debug"])
    out = capsys.readouterr().out
    assert out.count("Running Train") == 2 and out.count("Running Val") == 2
    assert "Traceback" not in out


def test_reference_entry_point_control_flow_through_compat_names(tmp_path):
    """SURVEY 8b / VERDICT r1 missing #2: a driver that uses ONLY the reference's import names (`arg_parser`,
    `solvers.base_solver`, `dg_util...tensorboard_logger`) and restates solver_runner.py:12-54 runs the HIP solver, the
    loggers receive the reference's keys (vince_solver.py:503-512, base_solver.py:121-128), the warm-up ramps the lr."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(root, "compat"), os.path.join(root, "compat", "_standins"), root])
    cmd = [sys.executable, os.path.join(root, "tests", "compat_driver.py"), "--title", "compat", "--description", "t",
           "--solver", "VinceSolver", "--backbone", "ResNet18", "--batch-size", "16", "--vince-queue-size", "64",
           "--vince-embedding-size", "64", "--input-width", "64", "--input-height", "64", "--epochs", "2",
           "--iterations-per-epoch", "3", "--base-lr", "0.03", "--no-save", "--no-restore", "--log-frequency", "1",
           "--base-logdir", str(tmp_path)]
    r = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("COMPAT_RESULT ")][-1]
    out = json.loads(line[len("COMPAT_RESULT "):])
    assert out["solver"] == "vince_amd.solvers.vince_solver.VinceSolver" and out["model"].startswith("vince_amd.models")
    assert out["iteration"] == 6 * 16 and out["tail"] == (6 * 16) % 64
    assert out["events_file"]
    # six logged iterations, each with the reference's three key families
    assert [s for s, _ in out["train_dicts"]] == [16 * i for i in range(6)]
    keys = out["train_dicts"][-1][1]
    full = "VinceSolver_VinceModel"   # base_solver.py:60-62: "%s_%s" % (solver_name, model_name)
    assert "losses/%s/nce_loss" % full in keys
    for k in ("nce_accuracy_mean", "nce_softmax_weight_mean", "cosine_sim", "cosine_sim_neg_max"):
        assert "metrics/%s/%s" % (full, k) in keys
    for k in ("total_time", "forward_time", "backward_time", "data_cache_time", "metrics_time"):
        assert "times/%s/%s" % (full, k) in keys
    tags = [t for t, _ in out["train_scalars"]]
    assert "metrics/%s/lr" % full in tags and "metrics/%s/epoch" % full in tags
    # solver_runner.py:36-43: lr_i = min(1, i/500) * peak for the first 500 iterations
    np.testing.assert_allclose(out["lrs"], [out["peak"] * (i + 1) / 500.0 for i in range(6)], rtol=1e-12)


@pytest.mark.parametrize("world", [2, 8])
def test_g7_per_rank_computation_vs_reference_chunkwise_emulation(world):
    """Golden set G7: what each data-parallel rank computes -- its chunk through the SAME weights with per-chunk BatchNorm
    statistics, its loss against the shared (not yet updated) queue, the rank-mean gradient -- and the replicated enqueue of
    the rank-ordered key block, against the REFERENCE run chunk-wise on CPU (oracle/make_golden_full.py; the cross-rank
    plumbing itself is tests/test_dp_gloo.py::test_g7_*)."""
    from vince_amd.models.vince_model import VinceQueueModel
    from vince_amd.utils.storage_queue import StorageQueue
    g = load("g7_dp.npz")
    p = "w%d_" % world
    b, K = 8, 96
    args, model = build("ResNet18", 64, "fp32", 7, batch_size=b, vince_queue_size=K, vince_temperature=0.07)
    model.train()
    qm = VinceQueueModel(args, model)
    qm.to(DEV)
    qm.train()
    queue = StorageQueue(K, 64, device=DEV)
    queue.vector_queue.copy_(torch.from_numpy(g[p + "queue_before"]))
    queue.current_tail = K - 5
    data, qdata = vo.g7_inputs(world, b, 64)
    keys, embs, losses, grads = [], [], [], None
    for r in range(world):
        sl = slice(r * b, (r + 1) * b)
        batch = {"data": data[sl].to(DEV), "queue_data": qdata[sl].to(DEV), "batch_types": ["images"], "batch_sizes": [b],
                 "data_source": ["XX"], "num_frames": [1]}
        qb = qm(batch, shuffle=True)
        o = model.get_embeddings(batch, shuffle=True)[0]
        o.update(queue.dequeue())
        o.update(model.split_dict_by_type(batch["batch_types"], batch["batch_sizes"], batch)[0])
        o.update(qb[0])
        o.update(model(o))
        loss = sum(w * v for w, v in model.loss(o).values())
        model.zero_grad()
        loss.backward()
        gr = {n: q.grad.detach().clone() for n, q in model.named_parameters() if q.grad is not None}
        grads = gr if grads is None else {n: grads[n] + gr[n] for n in gr}
        keys.append(qb[0]["queue_embeddings"].detach())
        embs.append(o["embeddings"].detach())
        losses.append(float(loss))
    assert rel(torch.cat(keys).cpu(), g[p + "keys"]) < 1e-3
    assert rel(torch.cat(embs).cpu(), g[p + "embeddings"]) < 1e-3
    np.testing.assert_allclose(losses, g[p + "losses"], rtol=2e-3, atol=2e-6)
    queue.enqueue(torch.cat(keys), None, "XX")
    assert queue.current_tail == int(g[p + "tail"]) and queue.full == bool(g[p + "full"])
    np.testing.assert_allclose(queue.vector_queue.cpu().numpy(), g[p + "queue_after"], rtol=1e-3, atol=2e-4)
    # rows the enqueue did not touch are bit-identical (ownership is integer arithmetic)
    same = np.all(g[p + "queue_after"] == g[p + "queue_before"], axis=1)
    np.testing.assert_array_equal(queue.vector_queue.cpu().numpy()[same], g[p + "queue_before"][same])
    for n in ["embedding.2.weight", "feature_extractor.model.layer4.1.bn2.weight"]:
        assert rel((grads[n] / world).cpu(), g[p + "meangrad_" + n]) < 5e-3, n
    assert rel((grads["feature_extractor.model.layer4.1.conv2.weight"] / world)[:4].cpu(),
               g[p + "meangrad_feature_extractor.model.layer4.1.conv2.weight"]) < 5e-3
    assert rel((grads["feature_extractor.model.conv1.weight"] / world).cpu(), g[p + "meangrad_feature_extractor.model.conv1.weight"]) < 3e-2


@pytest.mark.parametrize("dtype,arch", [("fp32", "ResNet50"), ("bf16", "ResNet50")])
def test_gram_statistics_join_equals_separate_passes_in_the_model(monkeypatch, dtype, arch):
    """No-grad train-mode forwards (the key encoder, forward + InfoNCE) take the Gram-statistics residual join in layer1 /
    layer2 (csrc/trunk.hip gram_on); VINCE_KNOBS=gram_join=0 runs the separate passes.  Same embeddings, same running statistics;
    and the fp32 run still reproduces the reference's G3 goldens."""
    g = load("g3_trunk.npz")
    outs, stats = {}, {}
    for mode in ("1", "0"):
        monkeypatch.setenv("VINCE_KNOBS", "gram_join=%s" % mode)
        _, model = build(arch, 128, dtype, 11)
        model.train()
        x = vo.structured_frames(2, 224, 224, seed=500 + 224).to(DEV)
        with torch.no_grad():
            o = model.get_embeddings({"data": x})
        outs[mode] = {k: o[k].float().cpu() for k in ("embeddings", "prenorm_features", "extracted_features")}
        sd = model.state_dict()
        stats[mode] = {k: sd[k].float().cpu() for k in sd if "layer1.1.bn3.running" in k or "layer2.0.bn3.running" in k
                       or "layer2.0.downsample.1.running" in k or "layer1.0.bn3.num_batches" in k}
    tol_e = 1e-4 if dtype == "fp32" else 0.15   # (2-image batch: the hardest case for bf16)
    for k in outs["1"]:
        assert rel(outs["1"][k], outs["0"][k]) < tol_e, k
    for k in stats["1"]:
        assert rel(stats["1"][k], stats["0"][k]) < (1e-5 if dtype == "fp32" else 2e-2), k
    if dtype == "fp32":
        p = "%s_224_train_" % arch
        assert rel(outs["1"]["embeddings"], g[p + "embeddings"]) < 5e-4
        assert rel(outs["1"]["extracted_features"], g[p + "extracted"]) < 5e-4


def test_gram_statistics_join_in_the_training_forward_equals_separate_passes(monkeypatch):
    """Grad-enabled bf16 forwards run layer1 / layer2's conv3 + bn3 + join through the streaming kernel with the Gram
    statistics (VINCE_KNOBS=gram_train=1 with bn3_algebra=0; the algebra route of the next test is the default); backward then reads the y3 / mask / mean / invstd that kernel and the Gram
    finalize left.  Two bf16 arrangements differ from each other by bf16 noise, which a freshly initialised 16-block
    BatchNorm chain amplifies, so both are held against the fp32 trunk: the new arrangement must be as
    close to it as the separate passes are -- embeddings, gradient norms everywhere, gradient direction where the fp32 / bf16
    comparison itself is conditioned (head, layer4: early-layer bf16 gradients of this 16-frame problem have cosine 0.1-0.35 to
    the fp32 ones under EITHER arrangement)."""
    names = ("feature_extractor.model.layer4.2.conv3.weight", "feature_extractor.model.layer2.1.conv3.weight",
             "feature_extractor.model.layer2.1.bn3.weight", "feature_extractor.model.layer1.0.conv3.weight",
             "feature_extractor.model.layer1.0.downsample.0.weight", "feature_extractor.model.layer1.1.bn2.bias",
             "feature_extractor.model.conv1.weight", "embedding.2.weight")
    res = {}
    for tag, dtype, mode in (("ref", "fp32", "1"), ("new", "bf16", "1"), ("old", "bf16", "0")):
        monkeypatch.setenv("VINCE_KNOBS", "bn3_algebra=0,gram_train=%s" % mode)
        _, model = build("ResNet50", 128, dtype, 11)
        model.train()
        x = vo.structured_frames(16, 128, 128, seed=77).to(DEV)
        o = model.get_embeddings({"data": x})
        w = torch.randn(16, 128, generator=torch.Generator().manual_seed(3)).to(DEV)
        model.zero_grad()
        (o["embeddings"] * w).sum().backward()
        named = dict(model.named_parameters())
        res[tag] = {"emb": o["embeddings"].detach().float().cpu(),
                    "grads": {n: named[n].grad.detach().float().cpu().clone() for n in names}}

    def cos(a, b):
        return float(torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0))
    e_new, e_old = rel(res["new"]["emb"], res["ref"]["emb"]), rel(res["old"]["emb"], res["ref"]["emb"])
    print("embeddings vs fp32: gram %.3e  separate %.3e" % (e_new, e_old))
    assert e_new < max(0.2, 1.5 * e_old)
    for n in names:
        c_new, c_old = cos(res["new"]["grads"][n], res["ref"]["grads"][n]), cos(res["old"]["grads"][n], res["ref"]["grads"][n])
        ratio = float(res["new"]["grads"][n].norm() / res["ref"]["grads"][n].norm())
        print("%-60s cos vs fp32: gram %.4f  separate %.4f  norm ratio %.3f" % (n, c_new, c_old, ratio))
        ratio_old = float(res["old"]["grads"][n].norm() / res["ref"]["grads"][n].norm())
        assert 0.5 < ratio < 2.0 and abs(ratio / ratio_old - 1.0) < 0.3, (n, ratio, ratio_old)
        if n.startswith("embedding") or "layer4" in n:
            assert c_new > c_old - 0.05, (n, c_new, c_old)


def test_bf16_trunk_forward_is_reproducible():
    """the Gram-statistics join feeds bn3 from a Gram matrix that used to be summed with fp32 atomics, so the key encoder's
    keys varied from run to run.  The Gram matrices now take the reproducible weight-gradient path (vince_conv_wgrad_det: per-split
    slabs + a fixed-order reduction), and the trunk output of two identical train-mode forwards is identical to the bit -- no-grad
    (key encoder) and grad-enabled (query encoder) alike."""
    _, model = build("ResNet50", 128, "bf16", 11)
    model.train()
    x = vo.structured_frames(8, 96, 96, seed=5).to(DEV)
    outs = []
    for grad in (False, False, True, True):
        with torch.set_grad_enabled(grad):
            o = model.get_embeddings({"data": x})
        outs.append(o["spatial_features"].detach().float().clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[2], outs[3])


def test_gram_matrix_from_the_pass_that_writes_the_tensor_vs_the_weight_gradient_launch(monkeypatch):
    """Since round 3 the Gram matrix of conv3's input comes out of the BatchNorm + ReLU pass that writes it (csrc/bn_gram.hip);
    `VINCE_KNOBS=gram_fused=0` restores the weight-gradient launch over the stored tensor.  Both sum exact bf16 products in fp32, in
    different orders (the op-level test holds them to 1e-5 of each other), and each route is reproducible to the bit on its own.  In
    the model two bf16 arrangements differ by bf16 noise that the freshly initialised BatchNorm chain amplifies, so
    -- as for the Gram join and the algebra above -- both are held against the fp32 trunk: the fused route must sit as close to it as
    the route it replaced, in the key encoder's no-grad forward and in the training forward."""
    x = vo.structured_frames(16, 128, 128, seed=6).to(DEV)
    _, ref_model = build("ResNet50", 128, "fp32", 11)
    ref_model.train()
    _, model = build("ResNet50", 128, "bf16", 11)
    model.train()
    for grad in (False, True):
        with torch.set_grad_enabled(grad):
            ref = ref_model.get_embeddings({"data": x})["embeddings"].detach().float().cpu()
        outs = {}
        for knobs in ("gram_fused=1", "gram_fused=0", "gram_fused=1"):
            monkeypatch.setenv("VINCE_KNOBS", knobs)
            with torch.set_grad_enabled(grad):
                o = model.get_embeddings({"data": x})
            cur = (o["spatial_features"].detach().float().cpu(), o["embeddings"].detach().float().cpu())
            if knobs in outs:
                assert torch.equal(outs[knobs][0], cur[0])     # the trunk; the head's split-K sums are fp32 atomics (1e-6 run to run)
            outs[knobs] = cur
        e_new, e_old = rel(outs["gram_fused=1"][1], ref), rel(outs["gram_fused=0"][1], ref)
        print("grad %s: embeddings vs fp32: fused %.3e, weight-gradient route %.3e; fused vs the other %.3e" %
              (grad, e_new, e_old, rel(outs["gram_fused=1"][1], outs["gram_fused=0"][1])))
        assert e_new < max(0.2, 1.5 * e_old), (e_new, e_old)


def test_bn3_backward_algebra_in_the_model_vs_separate_passes(monkeypatch):
    """The default bf16 training route since round 3 (csrc/bn_algebra.hip; VINCE_KNOBS=bn3_algebra=0 restores the separate passes): layer1 /
    layer2 bottlenecks run conv3 + bn3 + join in one streaming launch that does NOT store conv3's output, and backward gets bn3's
    gradients through the convolution algebraically.  Same three-way comparison as the Gram-join test above: fp32 trunk as the reference,
    the algebra route must sit as close to it as the separate passes do -- embeddings, gradient norms everywhere (incl. layer1 / layer2
    conv3 / bn3 tensors, the ones the algebra produces), gradient direction where the bf16 / fp32 comparison is conditioned."""
    names = ("feature_extractor.model.layer4.2.conv3.weight", "feature_extractor.model.layer2.1.conv3.weight",
             "feature_extractor.model.layer2.1.bn3.weight", "feature_extractor.model.layer2.1.bn3.bias",
             "feature_extractor.model.layer2.0.downsample.1.weight", "feature_extractor.model.layer2.2.conv2.weight",
             "feature_extractor.model.layer1.0.conv3.weight", "feature_extractor.model.layer1.2.bn3.weight",
             "feature_extractor.model.layer1.0.downsample.0.weight", "feature_extractor.model.layer1.1.bn2.bias",
             "feature_extractor.model.conv1.weight", "embedding.2.weight")
    res = {}
    for tag, dtype, mode in (("ref", "fp32", "1"), ("new", "bf16", "1"), ("old", "bf16", "0")):
        monkeypatch.setenv("VINCE_KNOBS", "bn3_algebra=%s" % mode)
        _, model = build("ResNet50", 128, dtype, 11)
        model.train()
        x = vo.structured_frames(16, 128, 128, seed=77).to(DEV)
        o = model.get_embeddings({"data": x})
        w = torch.randn(16, 128, generator=torch.Generator().manual_seed(3)).to(DEV)
        model.zero_grad()
        (o["embeddings"] * w).sum().backward()
        named = dict(model.named_parameters())
        res[tag] = {"emb": o["embeddings"].detach().float().cpu(),
                    "grads": {n: named[n].grad.detach().float().cpu().clone() for n in names}}

    def cos(a, b):
        return float(torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0))
    e_new, e_old = rel(res["new"]["emb"], res["ref"]["emb"]), rel(res["old"]["emb"], res["ref"]["emb"])
    print("embeddings vs fp32: algebra %.3e  separate %.3e" % (e_new, e_old))
    assert e_new < max(0.2, 1.5 * e_old)
    for n in names:
        c_new, c_old = cos(res["new"]["grads"][n], res["ref"]["grads"][n]), cos(res["old"]["grads"][n], res["ref"]["grads"][n])
        ratio = float(res["new"]["grads"][n].norm() / res["ref"]["grads"][n].norm())
        ratio_old = float(res["old"]["grads"][n].norm() / res["ref"]["grads"][n].norm())
        print("%-60s cos vs fp32: algebra %.4f  separate %.4f  norm ratio %.3f / %.3f" % (n, c_new, c_old, ratio, ratio_old))
        assert 0.5 < ratio < 2.0 and abs(ratio / ratio_old - 1.0) < 0.3, (n, ratio, ratio_old)
        if n.startswith("embedding") or "layer4" in n:
            assert c_new > c_old - 0.05, (n, c_new, c_old)


def test_layer1_join_with_the_next_blocks_conv1_in_the_model(monkeypatch):
    """Round 5: a layer1 join launch also runs the FOLLOWING block's conv1 on the block output while it is in LDS
    (vince_conv_expand_join_next; `VINCE_KNOBS=xjoin_next=0` restores the separate launches, `xjoin_next128=0` only the one at the
    layer1 -> layer2 transition).  The convolution output is the same to the bit (op-level test); its BatchNorm statistics are sums of
    the same bf16 values in another fp32 order.  For the 256 -> 64 members the constants come out identical and so does the trunk
    output; the 256 -> 128 member moves a last bit of layer2's first BatchNorm here and there -- bf16 rounding flips that the
    freshly initialised BatchNorm chain amplifies, so that leg is held against the fp32 trunk like the other route
    pairs.  Each route reproduces itself to the bit, and backward (which reads the tensors the fused launch wrote) gives the same
    gradients."""
    x = vo.structured_frames(16, 128, 128, seed=6).to(DEV)
    _, ref_model = build("ResNet50", 128, "fp32", 11)
    ref_model.train()
    _, model = build("ResNet50", 128, "bf16", 11)
    model.train()
    names = ("feature_extractor.model.layer1.1.conv1.weight", "feature_extractor.model.layer1.2.conv1.weight",
             "feature_extractor.model.layer2.0.conv1.weight", "feature_extractor.model.layer2.0.bn1.weight",
             "feature_extractor.model.layer1.1.bn1.weight", "feature_extractor.model.layer1.0.conv3.weight",
             "feature_extractor.model.layer1.2.conv3.weight", "feature_extractor.model.conv1.weight", "embedding.2.weight")
    w = torch.randn(16, 128, generator=torch.Generator().manual_seed(3)).to(DEV)
    SEP, N64, ALL = "xjoin_next=0", "xjoin_next=1,xjoin_next128=0", "xjoin_next=1,xjoin_next128=1"
    # the BatchNorm the 256 -> 128 member feeds: one forward per route from the same running state
    key = "feature_extractor.model.layer2.0.bn1.running_"
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    seen = {}
    for knobs in (SEP, ALL):
        monkeypatch.setenv("VINCE_KNOBS", knobs)
        model.load_state_dict(state0)
        with torch.no_grad():
            model.get_embeddings({"data": x})
        sd = model.state_dict()
        seen[knobs] = (sd[key + "mean"].float().cpu().clone(), sd[key + "var"].float().cpu().clone())
    model.load_state_dict(state0)
    for a, b in zip(seen[SEP], seen[ALL]):
        assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max()) + 1e-9, float((a - b).abs().max())
    for grad in (False, True):
        with torch.set_grad_enabled(grad):
            ref = ref_model.get_embeddings({"data": x})["embeddings"].detach().float().cpu()
        outs, grads = {}, {}
        for knobs in (ALL, SEP, N64, ALL):
            monkeypatch.setenv("VINCE_KNOBS", knobs)
            with torch.set_grad_enabled(grad):
                o = model.get_embeddings({"data": x})
            cur = (o["spatial_features"].detach().float().cpu(), o["embeddings"].detach().float().cpu())
            if knobs in outs:
                assert torch.equal(outs[knobs][0], cur[0])
            outs[knobs] = cur
            if grad:
                model.zero_grad()
                (o["embeddings"] * w).sum().backward()
                named = dict(model.named_parameters())
                grads[knobs] = {n: named[n].grad.detach().float().cpu().clone() for n in names}
        e_new, e_64, e_old = rel(outs[ALL][1], ref), rel(outs[N64][1], ref), rel(outs[SEP][1], ref)
        d64, d = rel(outs[N64][0], outs[SEP][0]), rel(outs[ALL][0], outs[SEP][0])
        print("grad %s: embeddings vs fp32: fused %.3e, 256 -> 64 only %.3e, separate %.3e; trunk output vs separate: %.3e / %.3e"
              % (grad, e_new, e_64, e_old, d, d64))
        assert e_new < max(0.2, 1.5 * e_old) and e_64 < max(0.2, 1.5 * e_old), (e_new, e_64, e_old)
        assert d64 < 1e-3, d64
        assert d < e_old, (d, e_old)       # the two bf16 routes are closer to each other than either is to fp32
        if grad:
            for n in names:
                for tag in (ALL, N64):
                    a, b = grads[tag][n], grads[SEP][n]
                    r = float(a.norm() / b.norm())
                    c = float(torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0))
                    print("%-55s %s / separate: norm ratio %.4f cos %.5f" % (n, "fused" if tag == ALL else "64-only", r, c))
                    # (early-layer bf16 gradients of a fresh encoder at this batch are noise-dominated -- cosine 0.24-0.39 against fp32,
                    # DESIGN section 3 -- so a last-bit change of one BatchNorm constant decorrelates them between two bf16 routes)
                    assert abs(r - 1.0) < 0.1 and c > (0.97 if tag == N64 else 0.5), (n, tag, r, c)


@pytest.mark.parametrize("dtype", ["bf16", "fp32", "x3"])
def test_two_backward_passes_without_zero_grad_accumulate(dtype):
    """Gradient accumulation: a second forward + backward WITHOUT zero_grad must ADD its gradient to the buffer -- for the
    bf16 trunk that includes the layer1 / layer2 bottlenecks whose conv3 / bn3 gradients come out of the BatchNorm-backward algebra
    (the raw weight gradient R lives in scratch; csrc/bn_algebra.hip finish_dw adds the finished one).  Same batch twice: every
    gradient doubles."""
    _, model = build("ResNet50", 128, dtype, 11)
    model.train()
    x = vo.structured_frames(8, 96, 96, seed=79).to(DEV)
    w = torch.randn(8, 128, generator=torch.Generator().manual_seed(5)).to(DEV)

    def fwd_bwd():
        for m in model.modules():            # same BatchNorm state for both passes is irrelevant in train mode (batch statistics)
            pass
        o = model.get_embeddings({"data": x})
        (o["embeddings"] * w).sum().backward()
        torch.cuda.synchronize()
        return {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
    model.zero_grad()
    g1 = fwd_bwd()
    g2 = fwd_bwd()                           # no zero_grad in between
    worst = max(float((g2[n] - 2 * g1[n]).norm()) / (float(g1[n].norm()) * 2 + 1e-30) for n in g1)
    names = [n for n in g1 if ("layer1" in n or "layer2" in n) and ("conv3" in n or "bn3" in n)]
    assert names
    print("accumulated twice vs 2 x once (%s): worst relative L2 difference %.2e" % (dtype, worst))
    # (not bit-exact: fp32 atomics of the split weight gradients and, in bf16, the rounding of re-summed activations' gradients)
    # (bf16: the two passes differ by the same run-to-run noise as two runs of one arrangement -- which branch gradient of a stage entry
    # is stored first decides a rounding of the stored type, and the early layers amplify it: 1e-2 ... 2.03e-2 measured)
    assert worst < (4e-2 if dtype == "bf16" else 2e-4), worst


def test_backward_with_a_deeper_dy_ring_and_the_old_wgrad_kernel_gives_the_same_gradients(monkeypatch):
    """Two engine switches that must not change results: `dy_slots` (how many layer gradients the backward pass keeps alive for the
    weight-gradient stream; 3 by default, up to 8) and `wgrad_tr=0` (bf16 weight gradients through the LDS-DMA kernel that
    conv_wgrad_tr replaced).  Same model, same batch: every parameter gradient agrees to the fp32-atomics noise of the split-K sums."""
    grads = {}
    for tag, knobs in (("base", ""), ("again", ""), ("dy_slots=6", "dy_slots=6"), ("wgrad_tr=0", "wgrad_tr=0")):
        monkeypatch.setenv("VINCE_KNOBS", knobs)
        _, model = build("ResNet50", 128, "bf16", 11)
        model.train()
        x = vo.structured_frames(8, 128, 128, seed=78).to(DEV)
        o = model.get_embeddings({"data": x})
        w = torch.randn(8, 128, generator=torch.Generator().manual_seed(4)).to(DEV)
        model.zero_grad()
        (o["embeddings"] * w).sum().backward()
        torch.cuda.synchronize()
        grads[tag] = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}

    def worst(a, b):    # largest relative L2 difference of any parameter gradient
        return max(float((a[n] - b[n]).norm()) / (float(b[n].norm()) + 1e-12) for n in b)
    # the backward pass is not bit-reproducible (fp32 atomics of the split-K weight gradients, fp64 atomics of the BatchNorm sums below
    # fp32 resolution, then bf16 rounding of every gradient tensor on the way down): two runs of the SAME configuration set the scale
    noise = worst(grads["again"], grads["base"])
    print("run-to-run: %.3e" % noise)
    for tag in ("dy_slots=6", "wgrad_tr=0"):
        d = worst(grads[tag], grads["base"])
        print("%s: %.3e" % (tag, d))
        assert d < max(3 * noise, 1e-3) and d < 0.3, (tag, d, noise)


@pytest.mark.parametrize("ranks", [2, 4])
def test_bench_multi_rank_launch_contract_on_one_gpu(ranks):
    """The driver's N > 1 launch line (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) with N = 2 and 4
    ranks sharing this GPU through gloo (VINCE_BENCH_ONE_GPU=1): rendezvous from the environment, bucketed gradient
    all-reduce, key all-gather, max-over-ranks timing -- and exactly ONE JSON line, printed by rank 0, for the whole job."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VINCE_BENCH_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(ranks), "--steps", "3", "--warmup", "1",
           "--backbone", "ResNet18", "--batch", "16", "--size", "64", "--queue", "512", "--embed", "64", "--no-extras"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["global_batch"] == 16 * ranks and d["config"]["parallelism"] == "dp%d" % ranks
    assert d["config"]["frames_per_step"] == 32 * ranks
    assert d["config"]["workload"].startswith("not a BASELINE configuration")
    assert np.isfinite(d["config"]["final_loss"])


def test_two_ranks_on_one_gpu_stay_identical():
    """The multi-rank path on GPU hardware: two data-parallel ranks of the full solver share this one GPU through the
    gloo backend (NCCL refuses two ranks per device) -- parameter / queue broadcast, bucketed gradient all-reduce behind
    the engine's bucket events, key all-gather, replicated enqueue; the script asserts that both replicas end with
    bit-identical parameters and queues (tools/dp2_one_gpu.py)."""
    import socket
    import subprocess
    import sys
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tools", "dp2_one_gpu.py")]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout


def test_a_discarded_model_is_collected():
    """The autograd node of a grad-enabled forward holds the model (ctx.model); the model must not hold that node's OUTPUT tensors in its
    saved state, or the cycle runs through a C++ object Python's collector cannot see and every discarded model keeps its workspaces
    (tens of GB per solver at the benchmark size; found through bench.py's extra legs)."""
    import gc
    import weakref
    _, model = build("ResNet18", 64, "bf16", 11)
    model.train()
    x = vo.structured_frames(4, 64, 64, seed=80).to(DEV)
    o = model.get_embeddings({"data": x})
    (o["embeddings"] * 0.5).sum().backward()
    o2 = model.get_embeddings({"data": x})         # a second grad-enabled forward whose graph is simply dropped
    torch.cuda.synchronize()
    ref = weakref.ref(model)
    del model, o, o2
    gc.collect()
    assert ref() is None


def test_cpu_model_forward_raises():
    from vince_amd.config import make_args
    from vince_amd.models.vince_model import VinceModel
    model = VinceModel(make_args())
    with pytest.raises(RuntimeError):
        model.get_embeddings({"data": torch.randn(2, 3, 64, 64)})


def test_imagenet_side_decoders_train_beside_the_contrastive_loss():
    """vince_model.py:79-90,244-248,282-288,344-348: with --use-imagenet two linear probes read DETACHED pooled features of
    "IN" batches; their cross-entropy joins the loss dict, their accuracy the metrics, their weights train, and the encoder's
    trajectory is untouched by them."""
    from vince_amd.config import make_args
    from vince_amd.data_source import SyntheticFrames
    from vince_amd.solvers.vince_solver import VinceSolver

    class Labelled(SyntheticFrames):
        labelled = True

        def __call__(self, loader_id=0):
            b = super().__call__(loader_id)
            if self.labelled:     # (like the reference, an "IN" batch without --use-imagenet is an error: vince_model.py:244-246)
                b["data_source"] = "IN"
                b["imagenet_labels"] = (torch.arange(self.batch_size, device=self.device) * 37) % 1000
            return b

    def run(use_imagenet):
        torch.manual_seed(0)
        src = Labelled(16, 64, 64, 1, device=DEV, seed=5)
        src.labelled = use_imagenet
        args = make_args(backbone="ResNet18", batch_size=16, vince_queue_size=64, input_size=(64, 64), compute_dtype="fp32",
                         use_imagenet=use_imagenet, batch_source=src)
        solver = VinceSolver(args)
        sd = vo.seeded_state(vo.model_spec("ResNet18", 64), 2)
        solver.model.load_state_dict(sd, strict=False)
        solver.queue_model.queue_network.load_state_dict(sd, strict=False)
        solver.vince_queue.vector_queue.copy_(torch.nn.functional.normalize(
            torch.randn(64, 64, generator=torch.Generator().manual_seed(1)), dim=1))
        solver.reset_epoch()
        w0 = [p.detach().clone() for p in solver.model.imagenet_decoders.parameters()] if use_imagenet else None
        out, flat1 = [], None
        for it in range(2):
            out.append(solver.run_train_iteration())
            if it == 0:
                flat1 = solver.model._flat[:solver.model._n_train].detach().cpu().clone()
        return solver, out, w0, flat1

    s1, o1, w0, f1 = run(True)
    s0, o0, _, f0 = run(False)
    for it, ((l1, m1), (l0, m0)) in enumerate(zip(o1, o0)):
        assert {"imagenet_loss_0", "imagenet_loss_1"} <= set(l1) and "imagenet_loss_0" not in l0
        assert {"imagenet_accuracy_0", "imagenet_accuracy_1"} <= set(m1)
        for k in ("imagenet_loss_0", "imagenet_loss_1"):
            v = float(l1[k].detach())
            assert np.isfinite(v) and 5.0 < v < 9.0          # ~ln(1000) for an untrained probe
        # (the second loss sits behind one optimiser step of a freshly initialised encoder, which amplifies the fp32 atomics' order
        # chaotically -- two PLAIN runs differ by as much, tools/dp_x3f_probe.py; one run in three failed at 1e-5 on both iterations)
        assert abs(float(l1["nce_loss"].detach()) - float(l0["nce_loss"].detach())) < (1e-5 if it == 0 else 1e-3)
    moved = [float((p.detach() - w).abs().max()) for p, w in zip(s1.model.imagenet_decoders.parameters(), w0)]
    assert all(m > 0 for m in moved)
    # the probes see detached features: encoder parameters evolve exactly as without them -- to atomics-order noise after one step (1e-8
    # measured), within its amplification after two
    assert rel(f1, f0) < 1e-5
    assert rel(s1.model._flat[:s1.model._n_train].cpu(), s0.model._flat[:s0.model._n_train].cpu()) < 2e-2
    assert {"imagenet_loss_0", "imagenet_loss_1"} <= set(s1.model.loss(None)) and "imagenet_accuracy_1" in s1.model.get_metrics(None)


def test_fill_queue_with_distinct_batches():
    """vince_solver.py:293-313 `fill_queue`: K rows from as many different batches as needed; with K not a multiple of the
    batch the last enqueue wraps (storage_queue.py:35-43), so the queue is full and the tail sits at the overshoot."""
    from vince_amd.config import make_args
    from vince_amd.data_source import SyntheticFrames
    from vince_amd.solvers.vince_solver import VinceSolver
    args = make_args(backbone="ResNet18", batch_size=16, vince_queue_size=40, input_size=(64, 64), compute_dtype="fp32",
                     batch_source=SyntheticFrames(16, 64, 64, 1, device=DEV, seed=5))
    solver = VinceSolver(args)
    solver.model.load_state_dict(vo.seeded_state(vo.model_spec("ResNet18", 64), 2))
    solver.reset_epoch()
    solver.fill_queue()
    q = solver.vince_queue
    assert q.full and q.current_tail == 8 and len(q) == 40
    rows = q.dequeue()["queue_vectors"].cpu()
    assert np.allclose(rows.norm(dim=1).numpy(), 1.0, atol=1e-5)
    # three different batches went in: rows written by batch 2 (8..15 kept from batch 0? no -- overwritten) differ from batch 1's
    assert not torch.allclose(rows[0:8], rows[16:24])
    # the key encoder is a hard copy of the encoder after the fill (param_update(model, 0))
    n = solver.model._n_ema
    assert torch.equal(solver.queue_model.queue_network._flat[:n], solver.model._flat[:n])


def test_x3f_mixed_mode_forward_is_x3s_and_backward_runs_on_the_bf16_twin(monkeypatch):
    """compute_dtype "x3f" (round 6): the forward is x3's -- the same split-half kernels, trunk features equal to rounding (bit-identical with
    `gram_shadow=0`) -- and also leaves bfloat16 copies of what
    backward reads in the workspace of a bf16 twin engine (engine.Trunk.set_shadow), which then runs the backward: gradients agree with
    x3's in direction (cosine) and size at bf16 grade.  With VINCE_X3F_HYBRID=0 the backward stays on the fp32 tensors as single bfloat16
    products; a second forward + backward reuses the twin."""
    x = vo.structured_frames(8, 96, 96, seed=31).to(DEV)

    def run(dtype, train=True, steps=1):
        _, model = build("ResNet50", 128, dtype, 12)
        model.train(train)
        for _ in range(steps):
            model.zero_grad()
            o = model.get_embeddings({"data": x})
            (o["embeddings"] * torch.linspace(-1, 1, 128, device=DEV)).sum().backward()
        torch.cuda.synchronize()
        return model, o["extracted_features"].detach().float().cpu(), {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters()
                                                               if p.grad is not None}

    m3, e3, g3 = run("x3")
    # (the query forward of x3f takes the key encoder's Gram route for the bottleneck tails -- bn3's statistics from the Gram matrix of
    # conv3's input, conv3 + join in one launch -- so its features equal x3's grad-enabled forward to rounding; with that switched off
    # the two forwards are the same launches and agree to the bit)
    monkeypatch.setenv("VINCE_KNOBS", "gram_shadow=0")
    _, ef0, _ = run("x3f")
    assert torch.equal(e3, ef0), "without the Gram route the x3f forward must be x3's to the bit"
    monkeypatch.delenv("VINCE_KNOBS")
    mf, ef, gf = run("x3f")
    assert mf.x3f_hybrid and len(mf._twins) == 1 and mf._saved_ws_bf is not None and not m3._twins
    assert rel(ef, e3) < 1e-4      # (3e-5 measured: bn3's statistics from a Gram matrix of split-half products, 2^-17 per term; 1.3e-5 with `gram_x3=0`)
    assert sorted(g3) == sorted(gf)
    worst = {}
    for n in g3:
        a, b = gf[n].double().flatten(), g3[n].double().flatten()
        if float(b.abs().max()) == 0.0:      # (the unused fc)
            assert float(a.abs().max()) == 0.0, n
            continue
        worst[n] = (float(a @ b / (a.norm() * b.norm())), abs(float(a.abs().sum() / b.abs().sum()) - 1.0))
    lo = min(worst, key=lambda n: worst[n][0])
    print("x3f vs x3 gradients (ResNet-50, 8 x 96 x 96): min cosine %.5f (%s), worst sum|g| ratio error %.2e" %
          (worst[lo][0], lo, max(v[1] for v in worst.values())))
    assert worst[lo][0] > 0.98 and max(v[1] for v in worst.values()) < 6e-2
    # the twin's backward takes the BatchNorm-backward algebra on the K = 64 / 128 bottleneck tails (Gram sums copied across instead of a
    # centred copy of conv3's output); `x3f_alg=0`: the separate passes everywhere -- the same forward to the bit, the same gradients
    # to bf16 rounding
    monkeypatch.setenv("VINCE_KNOBS", "x3f_alg=0")
    _, efa, gfa = run("x3f")
    monkeypatch.delenv("VINCE_KNOBS")
    assert torch.equal(efa, ef)
    cos_alg = {n: float(gfa[n].double().flatten() @ gf[n].double().flatten() / (gfa[n].double().norm() * gf[n].double().norm() + 1e-300))
               for n in gf if float(gf[n].abs().max()) > 0}
    lo_alg = min(cos_alg, key=cos_alg.get)
    print("x3f algebra route vs separate passes: min gradient cosine %.5f (%s)" % (cos_alg[lo_alg], lo_alg))
    assert cos_alg[lo_alg] > 0.995 and any(float((gfa[n] - gf[n]).abs().max()) > 0 for n in gf)      # (the switch switches something)
    # two steps: the twin and its workspace are reused, the stale-weight check of its bf16 cache sees the (unchanged) parameter version
    mf2, ef2, gf2 = run("x3f", steps=2)
    assert torch.equal(ef2, ef) and len(mf2._twins) == 1
    # the switch: no twin, the fp32-tensor backward (single bfloat16 products) -- same forward, gradients at x3's bounds
    n = "feature_extractor.model.layer3.2.conv2.weight"
    monkeypatch.setenv("VINCE_X3F_HYBRID", "0")
    mo, eo, go = run("x3f")
    assert not mo.x3f_hybrid and not mo._twins and torch.equal(eo, e3)      # (no twin, no shadow: x3's own grad-enabled forward)
    assert rel(go[n], g3[n]) < 3e-2


def test_backward_through_an_eval_mode_forward_is_refused():
    """The engine's BatchNorm backward is the train-mode one; through an eval-mode (running-statistics) forward it would be the wrong
    function (and overflows on unnormalised inputs: found in round 6).  Refused loudly; the head alone (detached features) still trains."""
    _, model = build("ResNet18", 64, "fp32", 12)
    model.eval()
    x = vo.structured_frames(4, 64, 64, seed=5).to(DEV)
    o = model.get_embeddings({"data": x})
    with pytest.raises(RuntimeError, match="eval-mode trunk forward"):
        o["embeddings"].sum().backward()
