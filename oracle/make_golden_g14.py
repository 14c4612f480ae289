"""G14: BASELINE config 2 at its OWN size (BASELINE.json configs[1]: ResNet-18, 224x224, batch 256, K=4096, fp32; D=64, T=0.07 as in
vince/train_vince.sh:25,28) from the centred-head state (the name-seeded encoder with the head's output bias shifted by minus the batch
mean of the pre-norm features: G12's construction).  ONE full training iteration of the imported reference on CPU -- the BasicBlock trunk
(models/building_blocks/resnet.py:53-92,269) at N=256: loss, metrics, all embeddings / keys / pre-norm features, the checksum of every
gradient tensor, sampled gradient rows, BatchNorm running statistics, and the 64-float shift.

TEST INFRASTRUCTURE; runs only in the build container (needs /root/reference; a few minutes, ~15 GB of host memory):

    python -m oracle.make_golden_g14
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness as rh  # noqa: E402
from oracle import vince_oracle as vo  # noqa: E402
from oracle.make_golden import OUT, np_, load_seeded  # noqa: E402

G14_SAMPLED = [   # (parameter, rows kept) -- full tensors for the small ones
    ("feature_extractor.model.conv1.weight", None), ("feature_extractor.model.bn1.weight", None),
    ("feature_extractor.model.layer1.0.conv1.weight", 16), ("feature_extractor.model.layer1.1.conv2.weight", 16),
    ("feature_extractor.model.layer2.0.conv1.weight", 8), ("feature_extractor.model.layer2.0.downsample.0.weight", 16),
    ("feature_extractor.model.layer2.1.bn2.weight", None), ("feature_extractor.model.layer3.0.conv2.weight", 4),
    ("feature_extractor.model.layer3.1.bn1.bias", None), ("feature_extractor.model.layer4.0.downsample.0.weight", 8),
    ("feature_extractor.model.layer4.1.conv2.weight", 2), ("feature_extractor.model.layer4.1.bn2.weight", None),
    ("embedding.0.weight", 8), ("embedding.0.bias", None), ("embedding.2.weight", 16), ("embedding.2.bias", None)]
G14_RUNNING = ["feature_extractor.model.bn1", "feature_extractor.model.layer1.1.bn2", "feature_extractor.model.layer2.0.downsample.1",
               "feature_extractor.model.layer3.1.bn1", "feature_extractor.model.layer4.1.bn2"]


def main():
    torch.set_num_threads(8)
    ref = rh.load_reference()
    ref.loss_util.USE_FLOAT = None
    c = vo.G14
    args = rh.make_args(backbone=c["arch"], batch_size=c["B"], vince_queue_size=c["K"], vince_embedding_size=c["embed"], num_frames=1,
                        vince_temperature=c["T"], base_lr=c["lr"])
    model = ref.vince_model.VinceModel(args)
    load_seeded(model, c["arch"], c["embed"], c["seed"])
    model.train()
    data, qdata = vo.g14_inputs()
    batch = {"data": data, "queue_data": qdata, "batch_types": ["images"], "batch_sizes": [c["B"]], "data_source": ["XX"], "num_frames": [1]}
    t0 = time.time()
    with torch.no_grad():
        probe = model.get_embeddings(batch, shuffle=True)[0]["prenorm_features"]
        shift = -probe.mean(0)
        # the probe forward moved the BatchNorm running statistics: the state is "seeded + shift", nothing else
        fresh = vo.seeded_state(vo.model_spec(c["arch"], c["embed"]), c["seed"])
        fresh["embedding.2.bias"] = fresh["embedding.2.bias"] + shift
        model.load_state_dict(fresh, strict=True)
    print("probe forward %.1fs" % (time.time() - t0), flush=True)
    queue_model = ref.vince_model.VinceQueueModel(args, model)
    queue_model.train()
    vq = ref.storage_queue.StorageQueue(c["K"], c["embed"])
    vq.vector_queue.copy_(vo.g14_queue())
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
    e = o["embeddings"].detach()
    pair = float(((e @ e.t()).sum() - e.shape[0]) / (e.shape[0] * (e.shape[0] - 1)))
    out = {"shift": np_(shift), "loss": np.array(float(loss)), "pairwise_cosine": np.array(pair)}
    out.update({"m_" + k: np.array(float(v)) for k, v in met.items()})
    out["embeddings"] = np_(o["embeddings"])
    out["queue_embeddings"] = np_(qb[0]["queue_embeddings"])
    out["prenorm"] = np_(o["prenorm_features"])
    out["extracted_checksum"] = np.array(vo.tensor_checksum(o["extracted_features"]))
    out["extracted_head"] = np_(o["extracted_features"][:4])
    named = dict(model.named_parameters())
    cs = {n: vo.tensor_checksum(p.grad) for n, p in named.items() if p.grad is not None}
    out["grad_names"] = np.array(sorted(cs))
    out["grad_checksums"] = np.array([cs[n] for n in sorted(cs)])
    for n, rows in G14_SAMPLED:
        gr = named[n].grad
        out["grad_" + n] = np_(gr if rows is None else gr[:rows])
    sd = model.state_dict()
    for bn in G14_RUNNING:
        out["run_" + bn + ".running_mean"] = np_(sd[bn + ".running_mean"])
        out["run_" + bn + ".running_var"] = np_(sd[bn + ".running_var"])
    path = os.path.join(OUT, "g14_config2.npz")
    np.savez_compressed(path, **out)
    print("g14 loss %.6f  metrics %s  mean pairwise cosine %.4f  (%d bytes)" % (float(loss), {k: round(float(v), 5) for k, v in met.items()}, pair,
                                                                              os.path.getsize(path)))


if __name__ == "__main__":
    main()
