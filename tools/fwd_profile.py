"""Runs N no-grad train-mode forwards of ResNet-50 at B=256 bf16 (the forward + InfoNCE leg without the loss) -- the workload
of tools/fwd_kstats.sh (rocprofv3 kernel stats of the forward alone)."""
import sys
import torch
sys.path.insert(0, ".")
from vince_amd.config import make_args
from vince_amd.models.vince_model import VinceModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
args = make_args(backbone="ResNet50", vince_embedding_size=128, compute_dtype="bf16", batch_size=256, input_size=(224, 224))
model = VinceModel(args).to("cuda:0")
model.train()
model.clone_spatial = False
x = torch.randn(256, 3, 224, 224, device="cuda:0")
with torch.no_grad():
    for _ in range(3):
        model.get_embeddings({"data": x})
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        model.get_embeddings({"data": x})
    e1.record()
    torch.cuda.synchronize()
print("forward ms", e0.elapsed_time(e1) / n)
