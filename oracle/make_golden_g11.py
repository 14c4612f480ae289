"""G11: the REFERENCE (imported from /root/reference) trained for 20 SGD steps from the name-seeded ResNet-50, then ONE more
iteration recorded in full -- a state off the freshly initialised, nearly collapsed encoder every other trunk fixture starts from
(SURVEY.md 8(c) "input caveat"; VERDICT r2 missing #4).

TEST INFRASTRUCTURE; runs only in the build container (needs /root/reference):

    python -m oracle.make_golden_g11

ResNet-50, D=128, B=16, K=256, 64x64 frames, T=0.2, lr=0.03 (the MoCo-v2 recipe of vince/train_moco_v2.sh at toy size): 21 iterations
of solvers/vince_solver.py:405-499 replayed with the reference's own classes (oracle/make_golden.replay_steps).  The fixture keeps
the loss / metric trajectory of all 21 iterations and, for iteration 20, embeddings, keys, gradient checksums of every parameter
and three gradient tensors.  The state after 20 steps is NOT stored (100 MB): tests reach it by replaying the same 20 iterations
with oracle.vince_oracle.OracleTrainer, which tests/test_oracle_golden.py pins against this trajectory first."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness as rh  # noqa: E402
from oracle import vince_oracle as vo  # noqa: E402
from oracle.make_golden import OUT, np_, load_seeded  # noqa: E402

G11 = vo.G11
G11_GRADS = ("embedding.2.weight", "feature_extractor.model.layer4.2.conv3.weight", "feature_extractor.model.layer2.1.bn3.weight")


g11_inputs = vo.g11_inputs


def centred(ref, out):
    """G11c: the same seeded ResNet-50 with the head's output bias shifted by minus the batch mean of the pre-norm features, so the
    embeddings of the batch are spread over the sphere (mean pairwise cosine ~ -1/(B-1)) instead of sharing one direction: the
    regime in which trunk error is NOT hidden behind the L2 normalisation (SURVEY 8(c) input caveat).  One full iteration.  The
    shift (128 floats) is stored, so every implementation can enter the identical state."""
    c = G11
    ref.loss_util.USE_FLOAT = None
    args = rh.make_args(backbone=c["arch"], batch_size=c["B"], vince_queue_size=c["K"], vince_embedding_size=c["embed"], num_frames=1,
                        vince_temperature=c["T"], base_lr=c["lr"])
    model = ref.vince_model.VinceModel(args)
    load_seeded(model, c["arch"], c["embed"], c["seed"])
    model.train()
    data, qdata = g11_inputs(100)
    batch = {"data": data, "queue_data": qdata, "batch_types": ["images"], "batch_sizes": [c["B"]], "data_source": ["XX"], "num_frames": [1]}
    with torch.no_grad():
        probe = model.get_embeddings(batch, shuffle=True)[0]["prenorm_features"]
        shift = -probe.mean(0)
        sd = model.state_dict()
        sd["embedding.2.bias"] += shift
        # the probe forward moved the BatchNorm running statistics: put the seeded ones back so the state is "seeded + shift"
        fresh = vo.seeded_state(vo.model_spec(c["arch"], c["embed"]), c["seed"])
        fresh["embedding.2.bias"] = fresh["embedding.2.bias"] + shift
        model.load_state_dict(fresh, strict=True)
    queue_model = ref.vince_model.VinceQueueModel(args, model)
    queue_model.train()
    vq = ref.storage_queue.StorageQueue(c["K"], c["embed"])
    vq.vector_queue.copy_(vo.g11_queue(78))
    qb = queue_model(batch, shuffle=True)
    o = model.get_embeddings(batch, shuffle=True)[0]
    o.update(vq.dequeue())
    o.update(model.split_dict_by_type(batch["batch_types"], batch["batch_sizes"], batch)[0])
    o.update(qb[0])
    o.update(model(o))
    ld = model.loss(o)
    met = model.get_metrics(o)
    loss = sum(w * v for w, v in ld.values())
    model.zero_grad()
    loss.backward()
    e = o["embeddings"].detach()
    pair = float(((e @ e.t()).sum() - e.shape[0]) / (e.shape[0] * (e.shape[0] - 1)))
    print("centred: loss %.5f %s  mean pairwise cosine %.4f" % (float(loss), {k: round(float(v), 4) for k, v in met.items()}, pair), flush=True)
    out["c_shift"] = np_(shift)
    out["c_loss"] = np.array(float(loss))
    out["c_metrics"] = np.array([float(met[k]) for k in ("nce_accuracy_mean", "cosine_sim", "cosine_sim_neg_max")])
    out["c_pairwise_cosine"] = np.array(pair)
    out["c_embeddings"] = np_(o["embeddings"])
    out["c_queue_embeddings"] = np_(qb[0]["queue_embeddings"])
    out["c_prenorm"] = np_(o["prenorm_features"])
    named = dict(model.named_parameters())
    cs = {n: vo.tensor_checksum(p.grad) for n, p in named.items() if p.grad is not None}
    out["c_grad_names"] = np.array(sorted(cs))
    out["c_grad_checksums"] = np.array([cs[n] for n in sorted(cs)])
    for n in G11_GRADS:
        out["c_grad_" + n] = np_(named[n].grad[:8])


def main():
    torch.set_num_threads(8)
    ref = rh.load_reference()
    ref.loss_util.USE_FLOAT = None
    c = G11
    args = rh.make_args(backbone=c["arch"], batch_size=c["B"], vince_queue_size=c["K"], vince_embedding_size=c["embed"], num_frames=1,
                        vince_temperature=c["T"], base_lr=c["lr"])
    model = ref.vince_model.VinceModel(args)
    load_seeded(model, c["arch"], c["embed"], c["seed"])
    model.train()
    queue_model = ref.vince_model.VinceQueueModel(args, model)
    queue_model.train()
    vq = ref.storage_queue.StorageQueue(c["K"], c["embed"])
    vq.vector_queue.copy_(vo.g11_queue(77))
    opt = torch.optim.SGD(model.parameters(), lr=c["lr"], weight_decay=0.0001, momentum=0.9)
    out = {}
    traj = []
    for it in range(c["iters"]):
        data, qdata = g11_inputs(it)
        batch = {"data": data, "queue_data": qdata, "batch_types": ["images"], "batch_sizes": [c["B"]], "data_source": ["XX"],
                 "num_frames": [1], "queue_data_cpu": qdata}
        qb = queue_model(batch, shuffle=True)
        o = model.get_embeddings(batch, shuffle=True)[0]
        ib = model.split_dict_by_type(batch["batch_types"], batch["batch_sizes"], batch)[0]
        o.update(vq.dequeue())
        o.update(ib)
        o.update(qb[0])
        o.update(model(o))
        ld = model.loss(o)
        met = model.get_metrics(o)
        loss = sum(w * v for w, v in ld.values())
        opt.zero_grad()
        loss.backward()
        traj.append([float(loss)] + [float(met[k]) for k in ("nce_accuracy_mean", "cosine_sim", "cosine_sim_neg_max")])
        e = o["embeddings"].detach()
        pair = float(((e @ e.t()).sum() - e.shape[0]) / (e.shape[0] * (e.shape[0] - 1)))
        traj[-1].append(pair)
        print("it %2d loss %.5f acc %.3f cos+ %.4f cos-max %.4f  mean pairwise cosine of the batch's embeddings %.4f" % (it, *traj[-1]), flush=True)
        if it == c["iters"] - 1:
            out["embeddings"] = np_(o["embeddings"])
            out["queue_embeddings"] = np_(qb[0]["queue_embeddings"])
            out["prenorm"] = np_(o["prenorm_features"])
            named = dict(model.named_parameters())
            cs = {n: vo.tensor_checksum(p.grad) for n, p in named.items() if p.grad is not None}
            out["grad_names"] = np.array(sorted(cs))
            out["grad_checksums"] = np.array([cs[n] for n in sorted(cs)])
            for n in G11_GRADS:
                out["grad_" + n] = np_(named[n].grad[:8])       # sampled rows keep the fixture small
        opt.step()
        vq.enqueue(o["queue_embeddings"], ib["queue_data_cpu"], ib["data_source"])
        queue_model.vince_update(model)
    out["trajectory"] = np.array(traj)          # [iters][loss, nce_accuracy_mean, cosine_sim, cosine_sim_neg_max, pairwise cosine]
    out["tail"], out["full"] = np.array(vq.current_tail), np.array(bool(vq.full))
    centred(ref, out)
    np.savez_compressed(os.path.join(OUT, "g11_after20.npz"), **out)
    print("g11 written: %d bytes" % os.path.getsize(os.path.join(OUT, "g11_after20.npz")))


if __name__ == "__main__":
    main()
