"""the boundary is "import path : symbol".  With compat/ on sys.path the reference's module names
(solver_runner.py:6-9, arg_parser.py:6-11, solvers/vince_solver.py:22-24) resolve to the HIP classes."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r"""
import json, sys
import arg_parser, constants, solvers, solver_runner
from solvers.base_solver import BaseSolver
from solvers.vince_solver import VinceSolver
from models.vince_model import VinceModel, VinceQueueModel
from models.base_model import BaseModel
from models.building_blocks import backbone_models
from models.building_blocks.backbone_models import ResNet18, ResNet50
from utils.storage_queue import StorageQueue
from utils.loss_util import similarity_cross_entropy
from utils import transforms
import vince_amd.solvers.vince_solver as vs, vince_amd.models.vince_model as vm, vince_amd.utils.storage_queue as sq
import vince_amd.utils.loss_util as lu, vince_amd.solvers.base_solver as bs
assert VinceSolver is vs.VinceSolver and solvers.VinceSolver is vs.VinceSolver and BaseSolver is bs.BaseSolver
assert VinceModel is vm.VinceModel and VinceQueueModel is vm.VinceQueueModel and StorageQueue is sq.StorageQueue
assert similarity_cross_entropy is lu.similarity_cross_entropy
assert "VinceSolver" in solvers.__all__ and {"ResNet18", "ResNet50"} <= set(backbone_models.__all__)
sys.argv = ["solver_runner.py", "--title", "t", "--description", "d", "--solver", "VinceSolver", "--backbone", "ResNet50",
            "--vince-queue-size", "65536", "--vince-embedding-size", "128", "--vince-temperature", "0.2", "--batch-size", "256",
            "--base-lr", "0.03", "--transform", "MoCoV2ImagenetTransform", "--use-imagenet", "--num-workers", "40"]
a = arg_parser.parse_args()          # the reference's signature: no arguments, reads sys.argv (arg_parser.py:38)
assert a.solver is vs.VinceSolver and a.backbone is ResNet50 and a.input_size == (224, 224)
q = StorageQueue(8, 4)               # CPU construction works; compute on a CPU tensor must raise, not fall back
print(json.dumps({"ok": True, "has_main": callable(solver_runner.main), "use_imagenet": bool(a.use_imagenet),
                  "transforms": sorted(transforms.__all__)[:3]}))
"""


def _env(extra=()):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "compat")] + list(extra) + [ROOT])
    return env


def test_reference_import_names_resolve_to_the_hip_classes():
    r = subprocess.run([sys.executable, "-c", PROBE], env=_env(), capture_output=True, text=True, cwd="/tmp", timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"] and out["has_main"]
    assert out["use_imagenet"]   # train_moco_v2.sh:39 sets it; the parser must not drop it


def test_compat_modules_hold_no_logic():
    """Every module under compat/ (outside _standins/) is a re-export: imports and docstrings only."""
    import ast
    for d, _, names in os.walk(os.path.join(ROOT, "compat")):
        if "_standins" in d:
            continue
        for n in names:
            if not n.endswith(".py"):
                continue
            tree = ast.parse(open(os.path.join(d, n)).read())
            for node in tree.body:
                ok = isinstance(node, (ast.Import, ast.ImportFrom)) or \
                    (isinstance(node, ast.Expr) and isinstance(node.value, ast.Constant)) or \
                    (isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") == "__all__") or \
                    (isinstance(node, ast.If) and n == "solver_runner.py")
                assert ok, "%s holds more than re-exports: %s" % (os.path.join(d, n), ast.dump(node)[:80])


def test_jsonl_logger_records_what_the_path_logs(tmp_path):
    sys.path.insert(0, ROOT)
    from vince_amd.utils.jsonl_logger import Logger
    lg = Logger(str(tmp_path / "train"))
    lg.dict_log({"losses/x/nce_loss": 1.5}, 32)
    lg.scalar_summary("metrics/x/lr", 0.03, step=0, increment_counter=False)
    lines = [json.loads(l) for l in open(tmp_path / "train" / "events.jsonl")]
    assert lines[0] == {"step": 32, "scalars": {"losses/x/nce_loss": 1.5}} and lines[1]["scalars"]["metrics/x/lr"] == 0.03
