"""Helper of tests/test_full_size_gpu.py: ONE grad-enabled forward + InfoNCE + backward of BASELINE config 3 (ResNet-50,
B=256, 224x224, K=65536, D=128, T=0.2) on the G9 inputs, dumped as an .npz.  Run as a child process so that the engine's
environment switches (VINCE_KNOBS="wgrad_stream=0,ds_stream=0,...", VINCE_OVERLAP_KEY -- read once per process) can differ between two runs.
VINCE_DUMP_FIXTURE=g12: the centred-head state of fixture G12 (seed 12, the stored head-bias shift, its own frames and queue).
VINCE_DUMP_FIXTURE=g14: BASELINE config 2 at its own size (ResNet-18, B=256, 224x224, K=4096, D=64, T=0.07) from fixture G14's centred-head state.
usage: full_size_grad_dump.py <out.npz> <bf16|fp32|x3> [gradient tensors to dump in full ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vince_oracle as vo   # noqa: E402  (test infrastructure: inputs + seeded weights + checksums only)


def main():
    out_path, dtype = sys.argv[1], sys.argv[2]
    from vince_amd.config import make_args
    from vince_amd.models.vince_model import VinceModel, VinceQueueModel
    dev = "cuda:0"
    fixture = os.environ.get("VINCE_DUMP_FIXTURE", "g9")
    g12, g14 = fixture == "g12", fixture == "g14"
    c = vo.G14 if g14 else vo.G12     # (G9 shares G12's sizes)
    B = c["B"]
    args = make_args(backbone=c["arch"], vince_embedding_size=c["embed"], compute_dtype=dtype, batch_size=B,
                     vince_queue_size=c["K"], vince_temperature=c["T"], base_lr=0.03, input_size=(c["hw"], c["hw"]))
    model = VinceModel(args)
    state = vo.seeded_state(vo.model_spec(c["arch"], c["embed"], False), c["seed"] if (g12 or g14) else 9)
    if g12 or g14:
        shift = np.load(os.path.join(ROOT, "tests", "golden", "g14_config2.npz" if g14 else "g12_full_centred.npz"))["shift"]
        state["embedding.2.bias"] = state["embedding.2.bias"] + torch.from_numpy(shift)
    model.load_state_dict(state)
    model.to(dev)
    model.train()
    qm = VinceQueueModel(args, model)
    qm.to(dev)
    qm.train()
    if g14:
        queue = vo.g14_queue().to(dev)
        data, qdata = vo.g14_inputs()
    elif g12:
        queue = vo.g12_queue().to(dev)
        data, qdata = vo.g12_inputs()
    else:
        queue = torch.nn.functional.normalize(torch.randn(65536, 128, generator=torch.Generator().manual_seed(9 + 77)), dim=-1).to(dev)
        data, qdata = vo.g9_inputs()
    batch = {"data": data.to(dev), "queue_data": qdata.to(dev), "batch_types": ["images"], "batch_sizes": [B],
             "data_source": ["XX"], "num_frames": [1]}
    if os.environ.get("VINCE_OVERLAP_KEY", "1") != "0":   # the solver's arrangement: key encoder on its own stream
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            qb = qm(batch, shuffle=True)
        o = model.get_embeddings(batch, shuffle=True)[0]
        torch.cuda.current_stream().wait_stream(side)
    else:
        qb = qm(batch, shuffle=True)
        o = model.get_embeddings(batch, shuffle=True)[0]
    o.update({"queue_vectors": queue, "queue_images": None, "queue_data_sources": None})
    o.update(model.split_dict_by_type(batch["batch_types"], batch["batch_sizes"], batch)[0])
    o.update(qb[0])
    o.update(model(o))
    ld = model.loss(o)
    met = model.get_metrics(o)
    loss = sum(w * v for w, v in ld.values())
    model.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    res = {"loss": np.array(float(loss))}
    res.update({"m_" + k: np.array(float(v)) for k, v in met.items()})
    res["embeddings"] = o["embeddings"].detach().float().cpu().numpy()
    res["queue_embeddings"] = qb[0]["queue_embeddings"].float().cpu().numpy()
    res["prenorm"] = o["prenorm_features"].detach().float().cpu().numpy()
    res["extracted_head"] = o["extracted_features"].detach().float().cpu().numpy()[:4]
    res["extracted_checksum"] = np.array(vo.tensor_checksum(o["extracted_features"].detach().float().cpu()))
    names, cs = [], []
    named = dict(model.named_parameters())
    for n in sorted(named):
        if named[n].grad is not None:
            names.append(n)
            cs.append(vo.tensor_checksum(named[n].grad.detach().float().cpu().contiguous()))
    res["grad_names"] = np.array(names)
    res["grad_checksums"] = np.array(cs)
    for n in sys.argv[3:]:
        res["grad_" + n] = named[n].grad.detach().float().cpu().numpy()
    sd = model.state_dict()
    for k in list(sd):
        if k.endswith("running_mean") or k.endswith("running_var"):
            if g14 or (any(t in k for t in (".bn1.", "layer1.2.bn3", "layer3.5.bn2", "layer4.2.bn3")) and k.count(".") <= 5):
                res["run_" + k] = sd[k].float().cpu().numpy()
    np.savez(out_path, **res)


if __name__ == "__main__":
    main()
