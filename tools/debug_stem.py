"""Debug aid: one teacher-forced C1 step (tests/test_model_gpu.py) with the stem formulation chosen by VINCE_STEM_PACKED;
dumps stem-adjacent gradients and forward features for an offline A/B."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import test_model_gpu as T
mode = sys.argv[1]
stack = T.gpu_stack(mode)
args, model, qm, queue, opt = stack
qi = torch.nn.functional.normalize(torch.randn(512, 64, generator=torch.Generator().manual_seed(5 + 77)), dim=-1)
queue.vector_queue.copy_(qi)
data, qdata = T.step_inputs(0)
output, ld, met, grads = T.gpu_step(*stack, data, qdata, mode)
out = {"conv1": grads["feature_extractor.model.conv1.weight"], "bn1w": grads["feature_extractor.model.bn1.weight"],
       "l1": grads["feature_extractor.model.layer1.0.conv1.weight"], "l4": grads["feature_extractor.model.layer4.1.conv2.weight"],
       "emb": output["embeddings"].detach(), "feat": output["extracted_features"].detach(),
       "spatial": output["spatial_features"].detach().float()}
np.savez(sys.argv[2], **{k: v.cpu().numpy() for k, v in out.items()})
print("saved")
