"""Probe: eval-mode (model.eval()) grad-enabled forward + backward -- which gradient tensors are finite (debug aid)."""
import sys, torch
sys.path.insert(0, '.')
from tests.test_model_gpu import build, DEV
from oracle import vince_oracle as vo
x = vo.structured_frames(8, 96, 96, seed=31).to(DEV)
for dtype in ("fp32",):
    for arch, emb in (("ResNet50", 128),):
        _, model = build(arch, emb, dtype, 12)
        model.eval()
        model.zero_grad()
        o = model.get_embeddings({"data": x})
        print("forward finite:", bool(torch.isfinite(o["embeddings"]).all()), bool(torch.isfinite(o["spatial_features"]).all()))
        (o["embeddings"] * torch.linspace(-1, 1, emb, device=DEV)).sum().backward()
        torch.cuda.synchronize()
        for n, p in model.named_parameters():
            if p.grad is not None:
                f = bool(torch.isfinite(p.grad).all())
                if "layer3.5" in n or "layer4" in n or "embedding" in n or not f and "layer3.4" in n:
                    print(n, "finite" if f else "NON-FINITE", float(p.grad.float().abs().max()) if f else "")
