"""Generates the slower reference fixtures (build container only; needs /root/reference):

  G7  data-parallel emulation: the REFERENCE run chunk-wise on CPU to emulate 2 and 8 ranks -- per-chunk
      BatchNorm statistics (what nn.DataParallel gives, models/vince_model.py:35), keys concatenated in rank order, the
      reference's StorageQueue after the replicated enqueue, per-rank losses against the shared queue and the rank-mean
      gradient.
  G2u the loss with UNEQUAL positives per row (the reference's USE_FLOAT branch, utils/loss_util.py:25-36,46-48) and its
      process-wide cached decision.
  G9  BASELINE config 3 at its REAL size: ResNet-50, B=256, 224x224, K=65536, D=128, T=0.2 (vince/train_moco_v2.sh:17-27),
      one full training iteration of the reference on CPU (forward, loss, metrics, backward): loss, metrics, embeddings,
      gradient checksums and sampled gradient rows, BatchNorm running statistics.

TEST INFRASTRUCTURE, like make_golden.py.  Usage: python -m oracle.make_golden_full [g7] [g9]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness as rh          # noqa: E402
from oracle import vince_oracle as vo         # noqa: E402
from oracle.make_golden import load_seeded, np_   # noqa: E402

OUT = os.environ.get("VINCE_GOLDEN_OUT", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))


g7_inputs = vo.g7_inputs   # (shared with the tests)


def g7_dp(ref):
    out = {}
    arch, embed, K, hw, T, b_local = "ResNet18", 64, 96, 64, 0.07, 8
    for world in (2, 8):
        ref.loss_util.USE_FLOAT = None
        args = rh.make_args(backbone=arch, batch_size=b_local, vince_queue_size=K, vince_embedding_size=embed, num_frames=1,
                            vince_temperature=T, base_lr=0.03)
        model = ref.vince_model.VinceModel(args)
        load_seeded(model, arch, embed, seed=7)
        model.train()
        queue_model = ref.vince_model.VinceQueueModel(args, model)
        queue_model.train()
        vq = ref.storage_queue.StorageQueue(K, embed)
        g = torch.Generator().manual_seed(7 + 77)
        vq.vector_queue.copy_(torch.nn.functional.normalize(torch.randn(K, embed, generator=g), dim=-1))
        vq.current_tail = K - 5          # the replicated enqueue of world*b_local rows wraps (and laps the ring at world 8)
        data, qdata = g7_inputs(world, b_local, hw)
        p = "w%d_" % world
        out[p + "queue_before"] = np_(vq.vector_queue).copy()   # (a copy: enqueue writes the tensor in place)
        keys, losses, embs = [], [], []
        grads = None
        named = dict(model.named_parameters())
        for r in range(world):           # one "rank" = one chunk through the SAME weights with its own batch statistics
            sl = slice(r * b_local, (r + 1) * b_local)
            batch = {"data": data[sl], "queue_data": qdata[sl], "batch_types": ["images"], "batch_sizes": [b_local],
                     "data_source": ["XX"], "num_frames": [1]}
            qb = queue_model(batch, shuffle=True)
            o = model.get_embeddings(batch, shuffle=True)[0]
            o.update(vq.dequeue())       # every rank reads the SAME queue: the enqueue happens after all ranks' losses
            o.update(model.split_dict_by_type(batch["batch_types"], batch["batch_sizes"], batch)[0])
            o.update(qb[0])
            o.update(model(o))
            ld = model.loss(o)
            loss = sum(w * v for w, v in ld.values())
            model.zero_grad()
            loss.backward()
            gr = {n: p_.grad.detach().clone() for n, p_ in named.items() if p_.grad is not None}
            grads = gr if grads is None else {n: grads[n] + gr[n] for n in gr}
            keys.append(qb[0]["queue_embeddings"].detach())
            embs.append(o["embeddings"].detach())
            losses.append(float(loss))
        gathered = torch.cat(keys, 0)    # rank order
        vq.enqueue(gathered, [None] * gathered.shape[0], "XX")   # storage_queue.py:32 wants one image slot per key
        out[p + "keys"] = np_(gathered)
        out[p + "embeddings"] = np_(torch.cat(embs, 0))
        out[p + "losses"] = np.array(losses)
        out[p + "queue_after"] = np_(vq.vector_queue)
        out[p + "tail"], out[p + "full"] = np.array(vq.current_tail), np.array(vq.full)
        for n in ["embedding.2.weight", "feature_extractor.model.layer4.1.conv2.weight", "feature_extractor.model.layer4.1.bn2.weight",
                  "feature_extractor.model.conv1.weight"]:
            mg = grads[n] / world
            out[p + "meangrad_" + n] = np_(mg[:4] if mg.dim() == 4 and mg.shape[0] > 64 else mg)   # (first 4 filters of the big one)
    np.savez_compressed(os.path.join(OUT, "g7_dp.npz"), **out)


G9_SAMPLED = [   # (parameter, rows kept) -- full tensors for the small ones
    ("feature_extractor.model.conv1.weight", None), ("feature_extractor.model.bn1.weight", None), ("feature_extractor.model.bn1.bias", None),
    ("feature_extractor.model.layer1.0.conv1.weight", None), ("feature_extractor.model.layer1.0.downsample.0.weight", 16),
    ("feature_extractor.model.layer1.2.conv3.weight", 16), ("feature_extractor.model.layer2.0.conv2.weight", 4),
    ("feature_extractor.model.layer2.3.bn3.weight", None), ("feature_extractor.model.layer3.0.downsample.0.weight", 8),
    ("feature_extractor.model.layer3.5.conv2.weight", 2), ("feature_extractor.model.layer3.5.bn2.bias", None),
    ("feature_extractor.model.layer4.0.conv1.weight", 8), ("feature_extractor.model.layer4.2.conv3.weight", 8),
    ("feature_extractor.model.layer4.2.bn3.weight", None), ("embedding.0.weight", 8), ("embedding.0.bias", None),
    ("embedding.2.weight", 16), ("embedding.2.bias", None)]


g9_inputs = vo.g9_inputs


def g9_full(ref):
    arch, embed, B, K, hw, T = "ResNet50", 128, 256, 65536, 224, 0.2
    ref.loss_util.USE_FLOAT = None
    args = rh.make_args(backbone=arch, batch_size=B, vince_queue_size=K, vince_embedding_size=embed, num_frames=1,
                        vince_temperature=T, base_lr=0.03)
    model = ref.vince_model.VinceModel(args)
    load_seeded(model, arch, embed, seed=9)
    model.train()
    queue_model = ref.vince_model.VinceQueueModel(args, model)
    queue_model.train()
    vq = ref.storage_queue.StorageQueue(K, embed)
    g = torch.Generator().manual_seed(9 + 77)
    vq.vector_queue.copy_(torch.nn.functional.normalize(torch.randn(K, embed, generator=g), dim=-1))
    data, qdata = g9_inputs(B, hw)
    batch = {"data": data, "queue_data": qdata, "batch_types": ["images"], "batch_sizes": [B], "data_source": ["XX"],
             "num_frames": [1]}
    t0 = time.time()
    qb = queue_model(batch, shuffle=True)
    print("key forward %.1fs" % (time.time() - t0), flush=True)
    o = model.get_embeddings(batch, shuffle=True)[0]
    print("query forward %.1fs" % (time.time() - t0), flush=True)
    o.update(vq.dequeue())
    o.update(model.split_dict_by_type(batch["batch_types"], batch["batch_sizes"], batch)[0])
    o.update(qb[0])
    o.update(model(o))
    ld = model.loss(o)
    met = model.get_metrics(o)
    loss = sum(w * v for w, v in ld.values())
    model.zero_grad()
    loss.backward()
    print("backward %.1fs" % (time.time() - t0), flush=True)
    out = {"loss": np.array(float(loss))}
    out.update({"m_" + k: np.array(float(v)) for k, v in met.items()})
    out["embeddings"] = np_(o["embeddings"])
    out["queue_embeddings"] = np_(qb[0]["queue_embeddings"])
    out["prenorm"] = np_(o["prenorm_features"])
    out["extracted_checksum"] = np.array(vo.tensor_checksum(o["extracted_features"]))
    out["extracted_head"] = np_(o["extracted_features"][:4])
    named = dict(model.named_parameters())
    cs = {}
    for n, p in named.items():
        if p.grad is not None:
            cs[n] = vo.tensor_checksum(p.grad)
    out["grad_names"] = np.array(sorted(cs))
    out["grad_checksums"] = np.array([cs[n] for n in sorted(cs)])
    for n, rows in G9_SAMPLED:
        gr = named[n].grad
        out["grad_" + n] = np_(gr if rows is None else gr[:rows])
    sd = model.state_dict()
    for bn in ["feature_extractor.model.bn1", "feature_extractor.model.layer1.2.bn3", "feature_extractor.model.layer3.5.bn2",
               "feature_extractor.model.layer4.2.bn3"]:
        out["run_" + bn + ".running_mean"] = np_(sd[bn + ".running_mean"])
        out["run_" + bn + ".running_var"] = np_(sd[bn + ".running_var"])
    np.savez_compressed(os.path.join(OUT, "g9_full.npz"), **out)
    print("g9 loss %.6f  metrics %s" % (float(loss), {k: float(v) for k, v in met.items()}))


g2u_inputs = vo.g2u_inputs


def g2u_loss(ref):
    """The reference's USE_FLOAT branch (utils/loss_util.py:25-36,46-48): rows with 1, 2, 3 positives; then -- the decision being
    cached process-wide (App. D item 2) -- an EQUAL-count mask through the same float path."""
    out = {}
    sims, mask, eq = g2u_inputs()
    ref.loss_util.USE_FLOAT = None
    for tag, m in (("uneq", mask), ("eq_after", eq)):
        s = sims.clone().requires_grad_(True)
        r = ref.loss_util.similarity_cross_entropy(s, 0.2, 6, 1, m)
        r["dist"].backward()
        out[tag + "_dists"] = np_(r["dists"])
        out[tag + "_dist"] = np_(r["dist"])
        out[tag + "_softmax_weights"] = np_(r["softmax_weights"])
        out[tag + "_softmax_weight"] = np_(r["softmax_weight"])
        out[tag + "_dsims"] = np_(s.grad)
    assert bool(ref.loss_util.USE_FLOAT)
    ref.loss_util.USE_FLOAT = None
    np.savez_compressed(os.path.join(OUT, "g2u_loss_unequal.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    which = set(sys.argv[1:]) or {"g7", "g9"}
    ref = rh.load_reference()
    if "g2u" in which or not sys.argv[1:]:
        g2u_loss(ref)
        print("g2u done")
    if "g7" in which:
        g7_dp(ref)
        print("g7 done")
    if "g9" in which:
        g9_full(ref)
        print("g9 done")


if __name__ == "__main__":
    main()
