"""Diagnostic: gradients of the mixed mode (x3f) against x3's (fp32-grade) on one batch, per switch of the twin's backward: which tensors move
when the BatchNorm-backward algebra is on / off / with single bf16 matrices.  Usage: python tools/alg_probe.py [batch] [size] [knob-sets...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import vince_oracle as vo   # noqa: E402  (seeded weights and frames only)

DEV = torch.device("cuda:0")


def run(dtype, knobs, x):
    from vince_amd.config import make_args
    from vince_amd.models.vince_model import VinceModel
    if knobs:
        os.environ["VINCE_KNOBS"] = knobs
    else:
        os.environ.pop("VINCE_KNOBS", None)
    model = VinceModel(make_args(backbone="ResNet50", vince_embedding_size=128, compute_dtype=dtype))
    model.load_state_dict(vo.seeded_state(vo.model_spec("ResNet50", 128), 12))
    model.to(DEV).train()
    o = model.get_embeddings({"data": x})
    (o["embeddings"] * torch.linspace(-1, 1, 128, device=DEV)).sum().backward()
    torch.cuda.synchronize()
    return {n.replace("feature_extractor.model.", ""): p.grad.detach().double().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}


if __name__ == "__main__":
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    sets = sys.argv[3:] or ["", "alg_split=0", "x3f_alg=0"]
    x = vo.structured_frames(nb, size, size, seed=31).to(DEV)
    ref = run("x3", "", x)
    for k in sets:
        g = run("x3f", k, x)
        err = {n: float((g[n] - ref[n]).norm() / (ref[n].norm() + 1e-300)) for n in ref if float(ref[n].abs().max()) > 0}
        sm = {n: abs(float(g[n].abs().sum() / ref[n].abs().sum()) - 1) for n in err}
        fam = {}
        for n in err:
            key = ("alg-block " if n.startswith(("layer1.", "layer2.")) else "other ") + n.split(".")[-2] + "." + n.split(".")[-1] if "layer" in n else n
            fam.setdefault(key, []).append((err[n], sm[n]))
        print("== knobs [%s]" % k)
        for n in ("layer1.1.bn2.bias", "layer1.1.bn2.weight", "layer1.2.bn2.bias", "layer2.1.bn2.bias", "layer1.1.bn1.bias", "layer3.1.bn2.bias"):
            d, r = g[n] - ref[n], ref[n]
            print("   %-22s |d|/|ref| %.2e  cos(d, ref) %+.3f  sum d / sum|d| %+.3f  sum ref / sum|ref| %+.3f  |ref| %.2e   d[:6]/|ref|rms %s" % (
                n, float(d.norm() / r.norm()), float(d @ r / (d.norm() * r.norm())), float(d.sum() / d.abs().sum()), float(r.sum() / r.abs().sum()),
                float(r.norm()), " ".join("%+.3f" % float(v / (r.norm() / len(r) ** 0.5)) for v in d[:6])), flush=True)
        for key in sorted(fam):
            v = fam[key]
            print("   %-32s n=%2d  rel-norm err median %.2e worst %.2e | sum|g| err worst %.2e" %
                  (key, len(v), sorted(e for e, _ in v)[len(v) // 2], max(e for e, _ in v), max(s for _, s in v)), flush=True)
