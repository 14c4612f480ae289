"""Diagnostic: the forced single-rank data-parallel step against the plain step, per compute dtype (ResNet-50, 16 x 64 x 64): parameters after
one and after two iterations, and the run-to-run noise floor of the plain step.  Usage: python tools/dp_x3f_probe.py [dtypes...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import vince_oracle as vo   # noqa: E402  (test infrastructure: seeded weights only)

DEV = torch.device("cuda:0")


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def run(force, dtype, backbone="ResNet50", steps=2, size=64):
    import torch.distributed as dist  # noqa: F401
    from vince_amd.config import make_args
    from vince_amd.data_source import SyntheticFrames
    from vince_amd.solvers.vince_solver import VinceSolver
    torch.manual_seed(0)
    if force:
        os.environ["VINCE_FORCE_DP"] = "1"
    else:
        os.environ.pop("VINCE_FORCE_DP", None)
    args = make_args(backbone=backbone, batch_size=16, vince_queue_size=64, input_size=(size, size), compute_dtype=dtype,
                     batch_source=SyntheticFrames(16, size, size, 1, device=DEV, seed=5))
    solver = VinceSolver(args)
    solver.model.load_state_dict(vo.seeded_state(vo.model_spec(backbone, 64), 2))
    solver.queue_model.queue_network.load_state_dict(vo.seeded_state(vo.model_spec(backbone, 64), 2))
    solver.vince_queue.vector_queue.copy_(torch.nn.functional.normalize(torch.randn(64, 64, generator=torch.Generator().manual_seed(1)), dim=1))
    solver.reset_epoch()
    out = []
    for _ in range(steps):
        loss = float(solver.run_train_iteration()[0]["nce_loss"].detach())
        torch.cuda.synchronize()
        out.append((loss, solver.model._flat.clone().cpu(), solver.model._flat_grad.clone().cpu()))
    return out


if __name__ == "__main__":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1)
    size = int(os.environ.get("PROBE_SIZE", "64"))
    for dtype in (sys.argv[1:] or ["fp32", "x3", "x3f", "bf16"]):
        a = run(False, dtype, size=size)
        b = run(False, dtype, size=size)
        c = run(True, dtype, size=size)
        for k in range(2):
            print("%-4s step %d: loss %.6f | plain again: loss %.2e params %.2e grads %.2e | forced DP: loss %.2e params %.2e grads %.2e"
                  % (dtype, k + 1, a[k][0], abs(b[k][0] / a[k][0] - 1), rel(b[k][1], a[k][1]), rel(b[k][2], a[k][2]),
                     abs(c[k][0] / a[k][0] - 1), rel(c[k][1], a[k][1]), rel(c[k][2], a[k][2])), flush=True)
    dist.destroy_process_group()
