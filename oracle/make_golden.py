"""Generate tests/golden/*.npz by running the REFERENCE (imported from /root/reference) on CPU.

TEST INFRASTRUCTURE; runs only in the build container (needs /root/reference):

    python -m oracle.make_golden

The fixtures hold inputs-by-seed (regenerated through oracle.vince_oracle helpers) and the reference's
outputs.  No reference source text is stored.  Golden sets follow SURVEY.md section 8(c): G1 queue, G2 loss,
G3/G4 trunk + head, G5 three-iteration training step (config C1, both modes), G6 jigsaw.
"""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness as rh  # noqa: E402
from oracle import vince_oracle as vo  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def np_(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def load_seeded(model, arch, embed, seed, jigsaw=False):
    spec = vo.model_spec(arch, embed, jigsaw)
    sd = vo.seeded_state(spec, seed)
    ref_keys = list(model.state_dict().keys())
    assert ref_keys == [n for n, _, _ in spec], "state-dict layout mismatch vs reference"
    model.load_state_dict(sd, strict=True)
    return spec


# ------------------------------------------------------------------------------------------ G1
def g1_queue(ref):
    out = {}
    scripts = {
        "k512": (512, [300, 300, 212, 512, 1, 700, 88]),
        "k96": (96, [32, 32, 32, 40, 96, 200, 7, 89]),
    }
    for name, (K, sizes) in scripts.items():
        torch.manual_seed(3)
        q = ref.storage_queue.StorageQueue(K, 4)
        nid = 0
        tails, fulls, owners = [], [], []
        q.vector_queue[:] = -1
        for n in sizes:
            items = (nid + torch.arange(n, dtype=torch.float32))[:, None].repeat(1, 4)
            q.enqueue(items, [None] * n, "src")
            nid += n
            tails.append(q.current_tail)
            fulls.append(q.full)
            owners.append(np_(q.vector_queue[:, 0]).astype(np.int64).copy())
        out[name + "_K"] = K
        out[name + "_sizes"] = np.array(sizes)
        out[name + "_tails"] = np.array(tails)
        out[name + "_fulls"] = np.array(fulls)
        out[name + "_owners"] = np.stack(owners)
    np.savez_compressed(os.path.join(OUT, "g1_queue.npz"), **out)


# ------------------------------------------------------------------------------------------ G2
def unit_rows(n, d, seed):
    return torch.nn.functional.normalize(torch.randn(n, d, generator=torch.Generator().manual_seed(seed)), dim=1)


def g2_loss(ref):
    out = {}
    cases = []
    ci = 0
    for (B, K, D) in [(8, 64, 64), (32, 512, 128)]:
        for mode in ["moco", "inter1", "inter4", "inter4self"]:
            for T in [0.07, 0.2]:
                cases.append((B, K, D, mode, T))
    for (B, K, D, mode, T) in cases:
        ref.loss_util.USE_FLOAT = None
        F_ = 4 if mode.startswith("inter4") else 1
        args = rh.make_args(batch_size=B, vince_queue_size=K, vince_embedding_size=D, num_frames=F_,
                            inter_batch_comparison=mode != "moco", self_batch_comparison=mode == "inter4self",
                            vince_temperature=T, vince_self_temperature=0.03)
        model = ref.vince_model.VinceModel(args)
        q = unit_rows(B, D, 100 + ci).requires_grad_(True)
        # keys correlated with queries so that positives are informative
        k = torch.nn.functional.normalize(q.detach() + 0.5 * unit_rows(B, D, 200 + ci), dim=1)
        queue = unit_rows(K, D, 300 + ci)
        inputs = dict(embeddings=q, extracted_features=torch.zeros(B, 1), queue_embeddings=k, queue_vectors=queue,
                      data_source="XX", num_frames=F_)
        o = model(inputs)
        losses = model.loss(o)
        metrics = model.get_metrics(o)
        total = sum(w * v for (w, v) in losses.values())
        total.backward()
        p = "c%d_" % ci
        out[p + "cfg"] = np.array([B, K, D, F_, int(mode != "moco"), int(mode == "inter4self")])
        out[p + "T"] = np.array(T)
        out[p + "sims_checksum"] = np.array(vo.tensor_checksum(o["vince_similarities"]))
        out[p + "dists"] = np_(o["vince_loss_dists"])
        out[p + "dist"] = np_(o["vince_loss_dist"])
        out[p + "softmax_weight"] = np_(o["vince_loss_softmax_weight"])
        for kk, v in metrics.items():
            out[p + "m_" + kk] = np_(v)
        if mode == "inter4self":
            out[p + "self_dist"] = np_(o["vince_loss_self_dist"])
        out[p + "dq"] = np_(q.grad)
        ci += 1
    out["n_cases"] = np.array(ci)
    np.savez_compressed(os.path.join(OUT, "g2_loss.npz"), **out)


# ------------------------------------------------------------------------------------------ G3 / G4
def g3_trunk(ref):
    out = {}
    for arch, embed, sizes in [("ResNet18", 64, [64, 224]), ("ResNet50", 128, [64, 224])]:
        for hw in sizes:
            for train in [True, False]:
                args = rh.make_args(backbone=arch, vince_embedding_size=embed)
                model = ref.vince_model.VinceModel(args)
                load_seeded(model, arch, embed, seed=11)
                model.train(train)
                x = vo.structured_frames(2, hw, hw, seed=500 + hw)
                with torch.no_grad():
                    o = model.get_embeddings({"data": x})
                p = "%s_%d_%s_" % (arch, hw, "train" if train else "eval")
                sp = o["spatial_features"]
                out[p + "spatial_stats"] = np.array([float(sp.mean()), float(sp.std()), float(sp.abs().max())])
                out[p + "spatial_checksum"] = np.array(vo.tensor_checksum(sp))
                if hw == 64:
                    out[p + "spatial"] = np_(sp)
                out[p + "extracted"] = np_(o["extracted_features"])
                out[p + "prenorm"] = np_(o["prenorm_features"])
                out[p + "embeddings"] = np_(o["embeddings"])
                sd = model.state_dict()
                for bn in ["feature_extractor.model.bn1", "feature_extractor.model.layer4.1.bn2",
                           "feature_extractor.model.layer2.0.downsample.1"]:
                    out[p + bn + ".running_mean"] = np_(sd[bn + ".running_mean"])
                    out[p + bn + ".running_var"] = np_(sd[bn + ".running_var"])
                    out[p + bn + ".num_batches_tracked"] = np_(sd[bn + ".num_batches_tracked"])
    np.savez_compressed(os.path.join(OUT, "g3_trunk.npz"), **out)


# ------------------------------------------------------------------------------------------ G5
def replay_steps(ref, arch, embed, B, K, hw, T, lr, mode, iters, seed):
    """Replays solvers/vince_solver.py:405-499 with the reference classes on CPU."""
    ref.loss_util.USE_FLOAT = None
    F_ = 4 if mode == "vince" else 1
    args = rh.make_args(backbone=arch, batch_size=B, vince_queue_size=K, vince_embedding_size=embed, num_frames=F_,
                        inter_batch_comparison=mode == "vince", self_batch_comparison=mode == "vince",
                        vince_temperature=T, base_lr=lr)
    model = ref.vince_model.VinceModel(args)
    spec = load_seeded(model, arch, embed, seed)
    model.train()
    queue_model = ref.vince_model.VinceQueueModel(args, model)
    queue_model.train()
    vq = ref.storage_queue.StorageQueue(K, embed)
    g = torch.Generator().manual_seed(seed + 77)
    vq.vector_queue.copy_(torch.nn.functional.normalize(torch.randn(K, embed, generator=g), dim=-1))
    opt = torch.optim.SGD(model.parameters(), lr=lr, weight_decay=0.0001, momentum=0.9)
    recs = []
    for it in range(iters):
        data = vo.structured_frames(B, hw, hw, seed=1000 + it)
        qdata = vo.structured_frames(B, hw, hw, seed=1000 + it) + 0.25 * vo.gaussian_frames(B, hw, hw, 2000 + it)
        batch = {"data": data, "queue_data": qdata, "batch_types": ["images"], "batch_sizes": [B],
                 "data_source": ["XX"], "num_frames": [F_], "queue_data_cpu": qdata}
        queue_batches = queue_model(batch, shuffle=True)
        outputs = model.get_embeddings(batch, shuffle=True)
        image_batches = model.split_dict_by_type(batch["batch_types"], batch["batch_sizes"], batch)
        loss_list, metrics_list = [], []
        for image_batch, queue_batch, output in zip(image_batches, queue_batches, outputs):
            output.update(vq.dequeue())
            output.update(image_batch)
            output.update(queue_batch)
            output.update(model(output))
            loss_dict = model.loss(output)
            metrics = model.get_metrics(output)
            loss_list.append({k: v[0] * v[1] for k, v in loss_dict.items()})
            metrics_list.append(metrics)
        total = sum(loss_list[0].values())
        opt.zero_grad()
        total.backward()
        rec = {"loss_" + k: float(v) for k, v in loss_list[0].items()}
        rec.update({"m_" + k: float(v) for k, v in metrics_list[0].items()})
        rec["embeddings"] = np_(outputs[0]["embeddings"])
        rec["queue_embeddings"] = np_(outputs[0]["queue_embeddings"])
        named = dict(model.named_parameters())
        rec["grad_conv1"] = np.array(vo.tensor_checksum(named["feature_extractor.model.conv1.weight"].grad))
        rec["grad_emb2"] = np.array(vo.tensor_checksum(named["embedding.2.weight"].grad))
        rec["grad_l4"] = np.array(vo.tensor_checksum(named["feature_extractor.model.layer4.1.conv2.weight"].grad))
        rec["grad_bn1w"] = np_(named["feature_extractor.model.bn1.weight"].grad)
        opt.step()
        for image_batch, output in zip(image_batches, outputs):
            vq.enqueue(output["queue_embeddings"], image_batch["queue_data_cpu"], image_batch["data_source"])
        queue_model.vince_update(model)
        rec["tail"], rec["full"] = vq.current_tail, vq.full
        rec["param_checksums"] = np.array([vo.tensor_checksum(p) for _, p in model.named_parameters()])
        rec["key_checksums"] = np.array([vo.tensor_checksum(p) for _, p in queue_model.queue_network.named_parameters()])
        rec["queue_checksum"] = np.array(vo.tensor_checksum(vq.vector_queue))
        recs.append(rec)
    return recs


def g5_step(ref):
    out = {}
    # lr 0.03 is config C1's value; on this synthetic problem the first step at that lr makes the
    # later iterations chaotic (fp32-vs-fp64 of the same code diverge by percent), so a low-lr variant is recorded
    # too for tight multi-step parity.
    for mode in ["moco", "vince"]:
        for tag, lr in [("", 0.03), ("_lowlr", 0.002)]:
            recs = replay_steps(ref, "ResNet18", 64, 32, 512, 64, 0.07, lr, mode, 3, seed=5)
            for it, rec in enumerate(recs):
                for k, v in rec.items():
                    out["%s%s_it%d_%s" % (mode, tag, it, k)] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, "g5_step.npz"), **out)


# ------------------------------------------------------------------------------------------ G6
def g6_jigsaw(ref):
    out = {}
    for hw in [66, 64]:
        args = rh.make_args(backbone="ResNet18", vince_embedding_size=64, jigsaw=True)
        model = ref.vince_model.VinceModel(args)
        load_seeded(model, "ResNet18", 64, seed=21, jigsaw=True)
        model.train()
        n = 2
        ramp = torch.arange(n * 3 * hw * hw, dtype=torch.float32).reshape(n, 3, hw, hw)
        x = vo.structured_frames(n, hw, hw, seed=900 + hw)
        torch.manual_seed(1234)
        orders = torch.stack([torch.randperm(9) for _ in range(n)])
        torch.manual_seed(1234)
        with torch.no_grad():
            o = model.get_embeddings({"data": x}, jigsaw=True)
        p = "hw%d_" % hw
        out[p + "orders"] = np_(orders)
        out[p + "embeddings"] = np_(o["embeddings"])
        out[p + "extracted"] = np_(o["extracted_features"])
        out[p + "spatial_checksum"] = np.array(vo.tensor_checksum(o["spatial_features"]))
        # pure index check of the tiling on a ramp image: replay vince_model.py:144-155 through the reference's own
        # tensor ops by feeding the ramp through a model whose trunk we bypass -> take tiles from the oracle and
        # verify against the reference-computed first/last pixel of every tile
        import torch.nn.functional as Fn
        from dg_util.python_utils import pytorch_util as pt_util
        data = ramp
        if (data.shape[2] % 3) != 0 or (data.shape[3] % 3) != 0:
            data = Fn.pad(data, (0, 3 - data.shape[3] % 3, 0, 3 - data.shape[2] % 3))
        data = pt_util.split_dim(data, 2, 3, data.shape[2] // 3)
        data = pt_util.split_dim(data, 4, 3, data.shape[4] // 3)
        data = data.permute(0, 2, 4, 1, 3, 5).contiguous()
        data = pt_util.remove_dim(data, (1, 2))
        out[p + "ramp_tiles_corner"] = np_(torch.stack([data[:, :, 0, 0], data[:, :, -1, -1]]))
        out[p + "ramp_tiles_shape"] = np.array(data.shape)
    np.savez_compressed(os.path.join(OUT, "g6_jigsaw.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ref = rh.load_reference()
    g1_queue(ref)
    print("g1 done")
    g2_loss(ref)
    print("g2 done")
    g3_trunk(ref)
    print("g3 done")
    g5_step(ref)
    print("g5 done")
    g6_jigsaw(ref)
    print("g6 done")


if __name__ == "__main__":
    main()
