"""eval-mode extract_features (BatchNorms folded) at BASELINE config 3's size, with the streaming join on / off in the folded forward.
Usage: python tools/infer_ab.py"""
import os
import subprocess
import sys
import torch
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ".")
    from vince_amd.config import make_args
    from vince_amd.models.vince_model import VinceModel
    args = make_args(backbone="ResNet50", vince_embedding_size=128, compute_dtype="bf16", batch_size=256, input_size=(224, 224))
    m = VinceModel(args).to("cuda:0")
    m.eval()
    x = torch.randn(256, 3, 224, 224, device="cuda:0")
    with torch.no_grad():
        for _ in range(3):
            m.extract_features(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            m.extract_features(x)
        e1.record()
        torch.cuda.synchronize()
    print("extract_features ms %.3f" % (e0.elapsed_time(e1) / 20))
else:
    for knobs in ("", "xjoin_folded=0", "", "xjoin_folded=0"):
        env = dict(os.environ, VINCE_KNOBS=knobs)
        out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        print("[VINCE_KNOBS=%s] %s" % (knobs, out))
