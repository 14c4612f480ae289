"""Pins the CPU oracle (oracle/vince_oracle.py) to the reference's own outputs (tests/golden/*.npz,
produced by oracle/make_golden.py from the imported reference).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import vince_oracle as vo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name))


# ------------------------------------------------------------------------------------------ G1 queue indices: bit-exact
@pytest.mark.parametrize("name", ["k512", "k96"])
def test_g1_queue_indices_bit_exact(name):
    g = load("g1_queue.npz")
    K = int(g[name + "_K"])
    q = vo.OracleQueue(K, 4, init=-np.ones((K, 4), np.float32))
    nid = 0
    for step, n in enumerate(g[name + "_sizes"]):
        items = np.repeat((nid + np.arange(n, dtype=np.float32))[:, None], 4, 1)
        q.enqueue(items)
        nid += int(n)
        assert q.current_tail == int(g[name + "_tails"][step])
        assert q.full == bool(g[name + "_fulls"][step])
        np.testing.assert_array_equal(q.vectors[:, 0].astype(np.int64), g[name + "_owners"][step])
        np.testing.assert_array_equal(q.owner, g[name + "_owners"][step])


def test_probe_case_from_survey():
    # K=512, enqueue 300 then 300 -> tail 88, full True
    segs, tail, wrapped = vo.enqueue_segments(0, 300, 512)
    assert (segs, tail, wrapped) == ([(0, 0, 300)], 300, False)
    segs, tail, wrapped = vo.enqueue_segments(300, 300, 512)
    assert (segs, tail, wrapped) == ([(300, 0, 212), (0, 212, 88)], 88, True)


# ------------------------------------------------------------------------------------------ G2 loss / metrics / dq
def _unit_rows(n, d, seed):
    return torch.nn.functional.normalize(torch.randn(n, d, generator=torch.Generator().manual_seed(seed)), dim=1)


def g2_case_inputs(ci, B, K, D):
    q = _unit_rows(B, D, 100 + ci)
    k = torch.nn.functional.normalize(q + 0.5 * _unit_rows(B, D, 200 + ci), dim=1)
    queue = _unit_rows(K, D, 300 + ci)
    return q, k, queue


def test_g2_loss_metrics_grad():
    g = load("g2_loss.npz")
    for ci in range(int(g["n_cases"])):
        p = "c%d_" % ci
        B, K, D, F_, inter, selfb = [int(v) for v in g[p + "cfg"]]
        T = float(g[p + "T"])
        q, k, queue = g2_case_inputs(ci, B, K, D)
        q.requires_grad_(True)
        sims, mask = vo.similarities(q, k, queue, bool(inter), F_)
        np.testing.assert_allclose(vo.tensor_checksum(sims), g[p + "sims_checksum"], rtol=1e-5, atol=1e-4)
        ld = vo.similarity_cross_entropy(sims, T, mask)
        np.testing.assert_allclose(ld["dists"].detach().numpy(), g[p + "dists"], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(float(ld["dist"]), float(g[p + "dist"]), rtol=1e-5)
        np.testing.assert_allclose(float(ld["softmax_weight"]), float(g[p + "softmax_weight"]), rtol=1e-4, atol=1e-7)
        met = vo.nce_metrics(sims.detach(), mask, ld["softmax_weight"])
        for kk in ["nce_accuracy_mean", "cosine_sim", "cosine_sim_neg_max"]:
            np.testing.assert_allclose(float(met[kk]), float(g[p + "m_" + kk]), rtol=1e-5, atol=1e-6)
        total = ld["dist"]
        if selfb:
            ssims = q @ q.t()
            sl = vo.similarity_cross_entropy(ssims, 0.03, mask[:, :B])
            np.testing.assert_allclose(float(sl["dist"]), float(g[p + "self_dist"]), rtol=1e-5)
            total = total + sl["dist"]
        total.backward()
        np.testing.assert_allclose(q.grad.numpy(), g[p + "dq"], rtol=1e-4, atol=1e-6)


def test_multi_positive_is_not_plain_softmax():
    # per-positive denominators differ from softmax over all columns
    q, k, queue = g2_case_inputs(0, 8, 64, 64)
    sims, mask = vo.similarities(q, k, queue, True, 4)
    ours = float(vo.similarity_cross_entropy(sims, 0.07, mask)["dist"])
    s = sims / 0.07
    plain = -(torch.log_softmax(s, 1)[mask]).mean()
    assert abs(ours - float(plain)) > 1e-4


# ------------------------------------------------------------------------------------------ G3/G4 trunk + head
@pytest.mark.parametrize("arch,embed", [("ResNet18", 64), ("ResNet50", 128)])
@pytest.mark.parametrize("hw", [64, 224])
@pytest.mark.parametrize("train", [True, False])
def test_g3_trunk_head(arch, embed, hw, train):
    g = load("g3_trunk.npz")
    p = "%s_%d_%s_" % (arch, hw, "train" if train else "eval")
    sd = vo.seeded_state(vo.model_spec(arch, embed), 11)
    x = vo.structured_frames(2, hw, hw, seed=500 + hw)
    with torch.no_grad():
        o = vo.get_embeddings(sd, x, arch, train)
    if hw == 64:
        np.testing.assert_allclose(o["spatial_features"].numpy(), g[p + "spatial"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(vo.tensor_checksum(o["spatial_features"]), g[p + "spatial_checksum"], rtol=1e-4, atol=1e-2)
    np.testing.assert_allclose(o["extracted_features"].numpy(), g[p + "extracted"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(o["prenorm_features"].numpy(), g[p + "prenorm"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(o["embeddings"].numpy(), g[p + "embeddings"], rtol=1e-4, atol=1e-6)
    for bn in ["feature_extractor.model.bn1", "feature_extractor.model.layer4.1.bn2",
               "feature_extractor.model.layer2.0.downsample.1"]:
        np.testing.assert_allclose(sd[bn + ".running_mean"].numpy(), g[p + bn + ".running_mean"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(sd[bn + ".running_var"].numpy(), g[p + bn + ".running_var"], rtol=1e-4, atol=1e-6)
        assert int(sd[bn + ".num_batches_tracked"]) == int(g[p + bn + ".num_batches_tracked"])


# ------------------------------------------------------------------------------------------ G5 three training iterations (C1)
@pytest.mark.parametrize("mode", ["moco", "vince"])
@pytest.mark.parametrize("tag,lr", [("", 0.03), ("_lowlr", 0.002)])
def test_g5_three_steps(mode, tag, lr):
    """Free-running three iterations.  From iteration 1 on the queue holds real keys of a randomly initialised
    (nearly collapsed) encoder and the problem is ill-conditioned: an fp64 run of this same oracle departs from
    its fp32 run by ~1e-2 relative in the stem gradients at iteration 1 and by ~1e-3..1e-2 in the embeddings at
    iteration 2, and the reference's own fp32 trajectory sits in the same band.  Iteration 0 is therefore
    checked tightly and later iterations to the conditioning band; the GPU parity tests are teacher-forced
    (one step from an identical state) for the same reason."""
    g = load("g5_step.npz")
    emb_atol = [2e-5, 2e-4, 3e-2]
    grad_tol = [1e-3, 5e-2, 1.0]
    tr = vo.OracleTrainer("ResNet18", 64, 512, 32, 0.07, lr, inter_batch=mode == "vince",
                          num_frames=4 if mode == "vince" else 1, self_batch=mode == "vince", seed=5)
    for it in range(3):
        data = vo.structured_frames(32, 64, 64, seed=1000 + it)
        qdata = vo.structured_frames(32, 64, 64, seed=1000 + it) + 0.25 * vo.gaussian_frames(32, 64, 64, 2000 + it)
        r = tr.step(data, qdata)
        pre = "%s%s_it%d_" % (mode, tag, it)
        np.testing.assert_allclose(r["nce_loss"], float(g[pre + "loss_nce_loss"]), rtol=1e-3)
        if mode == "vince":
            np.testing.assert_allclose(r["nce_loss_self"], float(g[pre + "loss_nce_loss_self"]), rtol=1e-3)
        np.testing.assert_allclose(r["embeddings"].numpy(), g[pre + "embeddings"], rtol=1e-3, atol=emb_atol[it])
        np.testing.assert_allclose(r["queue_embeddings"].numpy(), g[pre + "queue_embeddings"], rtol=1e-3, atol=emb_atol[it])
        for kk in ["nce_accuracy_mean", "cosine_sim", "cosine_sim_neg_max"]:
            np.testing.assert_allclose(r[kk], float(g[pre + "m_" + kk]), rtol=1e-3, atol=0.04 if (it == 2 and kk == "nce_accuracy_mean") else 1e-4)
        assert r["tail"] == int(g[pre + "tail"]) and r["full"] == bool(g[pre + "full"])
        gb = g[pre + "grad_bn1w"]
        tol = grad_tol[it]
        np.testing.assert_allclose(r["grads"]["feature_extractor.model.bn1.weight"].numpy(), gb,
                                   rtol=0, atol=tol * np.abs(gb).max())
        pcs = np.array([vo.tensor_checksum(tr.q[n]) for n in tr.pnames])
        np.testing.assert_allclose(pcs[:, 2], g[pre + "param_checksums"][:, 2], rtol=[1e-5, 1e-3, 2e-2][it])
        kcs = np.array([vo.tensor_checksum(tr.k[n]) for n in tr.pnames])
        np.testing.assert_allclose(kcs[:, 2], g[pre + "key_checksums"][:, 2], rtol=[1e-6, 1e-5, 1e-4][it])
        np.testing.assert_allclose(vo.tensor_checksum(torch.from_numpy(tr.queue.vectors)), g[pre + "queue_checksum"],
                                   rtol=1e-4, atol=[1e-3, 1e-2, 1.0][it])


# ------------------------------------------------------------------------------------------ G6 jigsaw
@pytest.mark.parametrize("hw", [66, 64])
def test_g6_jigsaw(hw):
    g = load("g6_jigsaw.npz")
    p = "hw%d_" % hw
    n = 2
    ramp = torch.arange(n * 3 * hw * hw, dtype=torch.float32).reshape(n, 3, hw, hw)
    tiles = vo.jigsaw_tile(ramp)
    assert list(tiles.shape) == list(g[p + "ramp_tiles_shape"])
    np.testing.assert_array_equal(torch.stack([tiles[:, :, 0, 0], tiles[:, :, -1, -1]]).numpy(), g[p + "ramp_tiles_corner"])
    sd = vo.seeded_state(vo.model_spec("ResNet18", 64, jigsaw=True), 21)
    x = vo.structured_frames(n, hw, hw, seed=900 + hw)
    with torch.no_grad():
        o = vo.get_embeddings(sd, x, "ResNet18", True, jigsaw=True, jigsaw_orders=torch.from_numpy(g[p + "orders"]))
    np.testing.assert_allclose(o["embeddings"].numpy(), g[p + "embeddings"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(o["extracted_features"].numpy(), g[p + "extracted"], rtol=1e-4, atol=1e-5)


def test_g2u_unequal_positives_use_float_branch():
    """The reference's USE_FLOAT branch (utils/loss_util.py:25-36,46-48; VERDICT r1 missing #4): rows with 1, 2 and 3 positives --
    full-width outputs, -2**20 fill, means over the mask entries -- and an equal-count mask sent down the same path afterwards
    (the reference caches the decision process-wide, App. D item 2).  Oracle vs the imported reference's numbers."""
    g = np.load(os.path.join(GOLDEN, "g2u_loss_unequal.npz"))
    sims, mask, eq = vo.g2u_inputs()
    for tag, m in (("uneq", mask), ("eq_after", eq)):
        s = sims.clone().requires_grad_(True)
        r = vo.similarity_cross_entropy(s, 0.2, m, use_float=True)
        r["dist"].backward()
        np.testing.assert_allclose(r["dists"].detach().numpy(), g[tag + "_dists"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(float(r["dist"]), float(g[tag + "_dist"]), rtol=1e-6)
        np.testing.assert_allclose(r["softmax_weights"].numpy(), g[tag + "_softmax_weights"], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(float(r["softmax_weight"]), float(g[tag + "_softmax_weight"]), rtol=1e-6)
        np.testing.assert_allclose(s.grad.numpy(), g[tag + "_dsims"], rtol=1e-5, atol=1e-8)
    # the automatic decision: unequal counts -> float path; equal counts -> compacted path with the same mean
    auto = vo.similarity_cross_entropy(sims, 0.2, mask)
    assert auto["dists"].shape == (6, 1, 20)
    comp = vo.similarity_cross_entropy(sims, 0.2, eq)
    assert comp["dists"].shape == (6, 1, 2)
    np.testing.assert_allclose(float(comp["dist"]), float(g["eq_after_dist"]), rtol=1e-6)


# ------------------------------------------------------------------------------------------ G11: off the freshly initialised state
def _g11_centred_trainer(g):
    """The oracle in the state of G11's centred-head part: seeded ResNet-50 with the head's output bias shifted by the stored vector."""
    c = vo.G11
    tr = vo.OracleTrainer(c["arch"], c["embed"], c["K"], c["B"], c["T"], c["lr"], seed=c["seed"], queue_init=vo.g11_queue(78).numpy())
    with torch.no_grad():
        tr.q["embedding.2.bias"] += torch.from_numpy(g["c_shift"])
        tr.k["embedding.2.bias"] += torch.from_numpy(g["c_shift"])
    return tr


def test_g11_centred_head_iteration():
    """G11c (oracle/make_golden_g11.py): one iteration of the REFERENCE from a state whose embeddings are spread over the sphere (mean
    pairwise cosine -0.06, not 0.98) -- the regime where the L2 normalisation hides nothing (SURVEY 8(c) input caveat)."""
    g = load("g11_after20.npz")
    assert float(g["c_pairwise_cosine"]) < 0.0
    tr = _g11_centred_trainer(g)
    data, qdata = vo.g11_inputs(100)
    r = tr.step(data, qdata)
    np.testing.assert_allclose(r["nce_loss"], float(g["c_loss"]), rtol=1e-4)
    np.testing.assert_allclose([r[k] for k in ("nce_accuracy_mean", "cosine_sim", "cosine_sim_neg_max")], g["c_metrics"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(r["embeddings"].numpy(), g["c_embeddings"], atol=2e-4)
    np.testing.assert_allclose(r["queue_embeddings"].numpy(), g["c_queue_embeddings"], atol=2e-4)
    names = list(g["c_grad_names"])
    for n in ("embedding.2.weight", "embedding.0.bias", "feature_extractor.model.layer4.2.conv3.weight",
              "feature_extractor.model.layer3.0.downsample.0.weight"):
        want = g["c_grad_checksums"][names.index(n)]
        got = vo.tensor_checksum(r["grads"][n])
        assert abs(got[2] - want[2]) <= 2e-2 * abs(want[2]), (n, got, want)     # sum |g|
    np.testing.assert_allclose(r["grads"]["embedding.2.weight"][:8].numpy(), g["c_grad_embedding.2.weight"], rtol=5e-2, atol=1e-5)


def test_g11_twenty_sgd_steps_trajectory():
    """G11: 21 iterations of the reference from the seeded ResNet-50 (B=16, K=256, 64x64, lr 0.03).  The problem is chaotic -- two runs of
    the REFERENCE on this CPU differ by 2e-3 in the loss of iteration 20 (thread-order of its reductions) -- so the oracle is held
    tightly on the first iterations and to the run-to-run band afterwards.  The state the oracle reaches after 20 iterations is what
    the GPU tests enter (tests/test_model_gpu.py::test_g11_*)."""
    g = load("g11_after20.npz")
    c = vo.G11
    tr = vo.OracleTrainer(c["arch"], c["embed"], c["K"], c["B"], c["T"], c["lr"], seed=c["seed"])
    traj = g["trajectory"]
    assert traj.shape == (21, 5)
    for it in range(c["iters"]):
        r = tr.step(*vo.g11_inputs(it))
        tol = 1e-4 if it == 0 else (2e-3 if it < 3 else 2e-2)
        np.testing.assert_allclose(r["nce_loss"], traj[it, 0], rtol=tol, err_msg="iteration %d" % it)
    assert tr.queue.current_tail == int(g["tail"]) and bool(tr.queue.full) == bool(g["full"])
    # iteration 20: same ballpark as the reference's record (direction of the embeddings, metrics)
    e, ge = r["embeddings"].numpy(), g["embeddings"]
    assert float((e * ge).sum(1).min()) > 0.99
    np.testing.assert_allclose(r["cosine_sim"], traj[20, 2], atol=3e-2)


@pytest.mark.parametrize("coin", ["k", "q"])
def test_g13_config5_multiframe_jigsaw_iteration(coin):
    """G13 (oracle/make_golden_g13.py): ONE iteration of the imported reference at BASELINE config 5's own size -- ResNet-50, 224 x 224,
    8 clips x 4 frames, inter-batch + self-batch comparison, jigsaw head (the jigsawed side padded to 225 -> 9 tiles of 75 x 75) -- per
    coin outcome; the oracle replays it with the recorded per-sample tile orders: both loss terms, embeddings of both sides, head gradient."""
    g = load("g13_config5.npz")
    c = vo.G13
    tr = vo.OracleTrainer(c["arch"], c["embed"], c["K"], c["B"], c["T"], c["lr"], inter_batch=True, num_frames=c["F"], self_batch=True,
                          self_temperature=c["self_T"], seed=c["seed"], jigsaw=True, queue_init=vo.g13_queue().numpy())
    data, qdata = vo.g13_inputs()
    p = coin + "_"
    orders = torch.from_numpy(g[p + "orders"])
    r = tr.step(data, qdata, jig_key=(coin == "k"), jig_query=(coin == "q"), orders_key=orders if coin == "k" else None,
                orders_query=orders if coin == "q" else None)
    terms = dict(zip(g[p + "loss_names"], g[p + "loss_terms"]))
    np.testing.assert_allclose(r["nce_loss"], terms["nce_loss"], rtol=1e-4)
    np.testing.assert_allclose(r["nce_loss_self"], terms["nce_loss_self"], rtol=1e-4)
    np.testing.assert_allclose(r["embeddings"].numpy(), g[p + "embeddings"], atol=2e-4)
    np.testing.assert_allclose(r["queue_embeddings"].numpy(), g[p + "queue_embeddings"], atol=2e-4)
    head = "jigsaw_embedding.2.weight" if coin == "q" else "embedding.2.weight"
    np.testing.assert_allclose(r["grads"][head][:8].numpy(), g[p + "grad_" + head], rtol=5e-2, atol=1e-5)
    names = list(g[p + "grad_names"])
    for n in (head, "feature_extractor.model.layer4.2.conv3.weight"):
        want = g[p + "grad_checksums"][names.index(n)]
        assert abs(vo.tensor_checksum(r["grads"][n])[2] - want[2]) <= 2e-2 * abs(want[2]), n


def test_g12_fixture_is_the_centred_head_state_at_config3_size():
    """G12 (oracle/make_golden_g12.py: config 3 at its real size from the centred-head state; 2.5 minutes and 40 GB of host memory for the
    reference, so the CPU suite only checks the fixture's shape and that it IS a spread-out state -- the GPU tests hold the HIP path
    against it): 256 unit embeddings of width 128, mean pairwise cosine far from the collapsed 0.98, every gradient tensor recorded."""
    g = load("g12_full_centred.npz")
    assert g["embeddings"].shape == (256, 128) and g["queue_embeddings"].shape == (256, 128) and g["shift"].shape == (128,)
    np.testing.assert_allclose(np.linalg.norm(g["embeddings"], axis=1), 1.0, atol=1e-5)
    e = g["embeddings"].astype(np.float64)
    pair = ((e @ e.T).sum() - 256) / (256 * 255)
    np.testing.assert_allclose(pair, float(g["pairwise_cosine"]), atol=1e-5)
    assert pair < 0.3 and 0.5 < float(g["m_nce_accuracy_mean"]) <= 1.0
    assert len(g["grad_names"]) == 163 == len(g["grad_checksums"])


def test_g14_config2_full_size_iteration():
    """G14 (oracle/make_golden_g14.py): ONE iteration of the imported reference at BASELINE config 2's OWN size -- ResNet-18 (BasicBlock
    trunk, models/building_blocks/resnet.py:53-92,269), 224 x 224, batch 256, K=4096, D=64, T=0.07, fp32 -- from the centred-head state.
    Cheap enough (half a minute of CPU) for the oracle to replay it here; the GPU tests hold the HIP path against the same fixture."""
    g = load("g14_config2.npz")
    c = vo.G14
    assert g["embeddings"].shape == (c["B"], c["embed"]) and g["shift"].shape == (c["embed"],) and len(g["grad_names"]) == 64
    assert float(g["pairwise_cosine"]) < 0.3
    tr = vo.OracleTrainer(c["arch"], c["embed"], c["K"], c["B"], c["T"], c["lr"], seed=c["seed"], queue_init=vo.g14_queue().numpy())
    with torch.no_grad():
        tr.q["embedding.2.bias"] += torch.from_numpy(g["shift"])
        tr.k["embedding.2.bias"] += torch.from_numpy(g["shift"])
    r = tr.step(*vo.g14_inputs())
    np.testing.assert_allclose(r["nce_loss"], float(g["loss"]), rtol=1e-4)
    np.testing.assert_allclose([r[k] for k in ("nce_accuracy_mean", "cosine_sim", "cosine_sim_neg_max")],
                               [float(g["m_" + k]) for k in ("nce_accuracy_mean", "cosine_sim", "cosine_sim_neg_max")], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(r["embeddings"].numpy(), g["embeddings"], atol=2e-4)
    np.testing.assert_allclose(r["queue_embeddings"].numpy(), g["queue_embeddings"], atol=2e-4)
    names = list(g["grad_names"])
    for n in ("embedding.2.weight", "embedding.0.bias", "feature_extractor.model.layer4.1.conv2.weight",
              "feature_extractor.model.layer2.0.downsample.0.weight", "feature_extractor.model.conv1.weight"):
        want = g["grad_checksums"][names.index(n)]
        got = vo.tensor_checksum(r["grads"][n])
        assert abs(got[2] - want[2]) <= 2e-2 * abs(want[2]), (n, got, want)     # sum |g|
    np.testing.assert_allclose(r["grads"]["embedding.2.weight"][:16].numpy(), g["grad_embedding.2.weight"], rtol=5e-2, atol=1e-5)
