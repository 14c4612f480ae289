"""G13: BASELINE config 5's per-GPU work at ITS OWN size under parity: ResNet-50, 224 x 224 frames, 4 frames per
clip (8 clips = 32 frames), inter-batch + self-batch comparison, D=128, T=0.2, self-T 0.03 (vince/train_vince_large.sh:17,23-28,38-40),
K=65536, and the jigsaw head (vince/train_vince_jigsaw.sh:20): the jigsawed side is zero-padded to 225 and cut into 9 tiles of 75 x 75
(models/vince_model.py:144-155), every tile through the trunk, tiles shuffled per sample, Linear(9C, C)-ReLU-Linear(C, D).

ONE full iteration of the imported reference on CPU per coin outcome (solvers/vince_solver.py:397-403): "k" = the KEY side is jigsawed,
"q" = the QUERY side is (its backward then runs through the jigsaw head and the 9x batch).  The reference draws its random numbers
inside get_embeddings (the shuffle permutation, then one randperm(9) per sample); they are captured and stored IN ORIGINAL SAMPLE ORDER
(the reference applies them to the shuffled batch and un-shuffles the outputs), so that the other implementations get the same choices
as inputs.  TEST INFRASTRUCTURE; build container only (needs /root/reference):

    python -m oracle.make_golden_g13
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

G13_GRADS = ("embedding.2.weight", "jigsaw_embedding.2.weight", "jigsaw_embedding.0.weight", "jigsaw_linear.weight",
             "feature_extractor.model.layer4.2.conv3.weight", "feature_extractor.model.layer2.1.bn3.weight")


class RandpermTap:
    """Records what torch.randperm returns while active."""

    def __enter__(self):
        self.calls, self._orig = [], torch.randperm

        def tapped(*a, **k):
            r = self._orig(*a, **k)
            self.calls.append(r.clone())
            return r
        torch.randperm = tapped
        return self

    def __exit__(self, *exc):
        torch.randperm = self._orig


def sample_orders(calls, B):
    """calls of ONE jigsawed get_embeddings(shuffle=True): [shuffle permutation of B, then B tile orders for the SHUFFLED positions]
    -> tile order of every ORIGINAL sample (position i of the shuffled batch holds sample shuffle[i])."""
    shuffle, orders = calls[0], torch.stack(calls[1:1 + B])
    assert shuffle.numel() == B and orders.shape == (B, 9)
    out = torch.empty_like(orders)
    out[shuffle] = orders
    return out


def main():
    torch.set_num_threads(8)
    ref = rh.load_reference()
    c = vo.G13
    out = {}
    data, qdata = vo.g13_inputs()
    for coin in ("k", "q"):
        ref.loss_util.USE_FLOAT = None
        args = rh.make_args(backbone=c["arch"], batch_size=c["B"], vince_queue_size=c["K"], vince_embedding_size=c["embed"], num_frames=c["F"],
                            vince_temperature=c["T"], vince_self_temperature=c["self_T"], base_lr=c["lr"], jigsaw=True,
                            inter_batch_comparison=True, self_batch_comparison=True)
        model = ref.vince_model.VinceModel(args)
        load_seeded(model, c["arch"], c["embed"], c["seed"], jigsaw=True)
        model.train()
        queue_model = ref.vince_model.VinceQueueModel(args, model)
        queue_model.train()
        vq = ref.storage_queue.StorageQueue(c["K"], c["embed"])
        vq.vector_queue.copy_(vo.g13_queue())
        batch = {"data": data, "queue_data": qdata, "batch_types": ["images"], "batch_sizes": [c["B"]], "data_source": ["XX"],
                 "num_frames": [c["F"]]}
        t0 = time.time()
        torch.manual_seed(1300 + (coin == "q"))
        with RandpermTap() as tap_k:
            qb = queue_model(batch, jigsaw=(coin == "k"), shuffle=True)
        with RandpermTap() as tap_q:
            o = model.get_embeddings(batch, jigsaw=(coin == "q"), shuffle=True)[0]
        orders = sample_orders((tap_k if coin == "k" else tap_q).calls, c["B"])
        o.update(vq.dequeue())
        o.update(model.split_dict_by_type(batch["batch_types"], batch["batch_sizes"], batch)[0])
        o.update(qb[0])
        o.update(model(o))
        ld = model.loss(o)
        met = model.get_metrics(o)
        loss = sum(w * v for w, v in ld.values())
        model.zero_grad()
        loss.backward()
        p = coin + "_"
        out[p + "orders"] = np_(orders)
        out[p + "loss"] = np.array(float(loss))
        out[p + "loss_names"] = np.array(sorted(ld))
        out[p + "loss_terms"] = np.array([float(ld[k][0] * ld[k][1]) for k in sorted(ld)])
        out[p + "metric_names"] = np.array(sorted(met))
        out[p + "metrics"] = np.array([float(met[k]) for k in sorted(met)])
        for k in ("embeddings", "prenorm_features", "extracted_features"):
            out[p + k] = np_(o[k])
            out[p + "queue_" + k] = np_(qb[0]["queue_" + k])
        named = dict(model.named_parameters())
        cs = {n: vo.tensor_checksum(pp.grad) for n, pp in named.items() if pp.grad is not None}
        out[p + "grad_names"] = np.array(sorted(cs))
        out[p + "grad_checksums"] = np.array([cs[n] for n in sorted(cs)])
        for n in G13_GRADS:
            if named[n].grad is not None:
                out[p + "grad_" + n] = np_(named[n].grad[:8])
        print("coin %s: %.1fs  loss %.6f  terms %s  metrics %s" % (coin, time.time() - t0, float(loss), {k: round(float(ld[k][1]), 5) for k in ld},
                                                                {k: round(float(v), 4) for k, v in met.items()}), flush=True)
    path = os.path.join(OUT, "g13_config5.npz")
    np.savez_compressed(path, **out)
    print("g13 written: %d bytes" % os.path.getsize(path))


if __name__ == "__main__":
    main()
