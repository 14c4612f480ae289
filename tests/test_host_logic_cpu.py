"""CPU-only tests of the host-side logic: ring-buffer index arithmetic vs the reference golden, LR schedule, argument
post-processing, batch splitting, state-dict layout / checkpoint key compatibility, gradient bucket plan."""
import math
import os
import types

import numpy as np
import pytest
import torch

from oracle import vince_oracle as vo
from vince_amd.utils.queue_index import enqueue_segments

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["k512", "k96"])
def test_queue_index_arithmetic_bit_exact_vs_reference_golden(name):
    g = np.load(os.path.join(GOLDEN, "g1_queue.npz"))
    K = int(g[name + "_K"])
    owner = np.full(K, -1, np.int64)
    tail, full, nid = 0, False, 0
    for step, n in enumerate(g[name + "_sizes"]):
        segs, tail, wrapped = enqueue_segments(tail, int(n), K)
        for dst, src, ln in segs:
            owner[dst:dst + ln] = nid + np.arange(src, src + ln)
        nid += int(n)
        full = full or wrapped
        assert tail == int(g[name + "_tails"][step]) and full == bool(g[name + "_fulls"][step])
        np.testing.assert_array_equal(owner, g[name + "_owners"][step])
    assert enqueue_segments(300, 300, 512) == ([(300, 0, 212), (0, 212, 88)], 88, True)   # SURVEY 8(a) a12 probe
    assert enqueue_segments(0, 0, 8) == ([], 0, False)
    assert enqueue_segments(5, 3, 8) == ([(5, 0, 3)], 8, False)       # exact fit does not wrap ...
    assert enqueue_segments(8, 1, 8) == ([(0, 0, 1)], 1, True)        # ... the next write does


def test_queue_index_closed_form_equals_the_oracles_loop():
    """Two independent formulations of storage_queue.py:31-49 -- the product's closed-form modular arithmetic and the oracle's restated
    recursion -- on every (tail, n, K) of a small exhaustive grid (tail AT maxsize, n = 0, exact fits, several laps) and a random
    sweep at the real queue size."""
    for K in (1, 2, 3, 7, 8):
        for tail in range(K + 1):
            for n in range(4 * K + 3):
                assert enqueue_segments(tail, n, K) == vo.enqueue_segments(tail, n, K), (tail, n, K)
    rng = np.random.default_rng(0)
    for _ in range(2000):
        K = 65536
        tail, n = int(rng.integers(0, K + 1)), int(rng.integers(0, 5 * K))
        assert enqueue_segments(tail, n, K) == vo.enqueue_segments(tail, n, K), (tail, n, K)


def test_state_dict_layout_matches_reference_and_loads_strictly():
    from vince_amd.config import make_args
    from vince_amd.models.vince_model import VinceModel
    for arch, embed, jig in [("ResNet18", 64, False), ("ResNet50", 128, False), ("ResNet18", 64, True)]:
        spec = vo.model_spec(arch, embed, jig)
        model = VinceModel(make_args(backbone=arch, vince_embedding_size=embed, jigsaw=jig))
        sd = model.state_dict()
        assert list(sd.keys()) == [n for n, _, _ in spec]
        assert all(tuple(sd[n].shape) == tuple(s) for n, s, _ in spec)
        seeded = vo.seeded_state(spec, 3)
        model.load_state_dict(seeded)   # strict
        for n in ["feature_extractor.model.layer2.0.conv1.weight", "embedding.2.bias", "feature_extractor.model.bn1.running_var"]:
            torch.testing.assert_close(model.state_dict()[n], seeded[n], rtol=0, atol=0)
        # conv weights are channels_last views of the flat buffer: memory order [Co][kh][kw][Ci]
        w = dict(model.named_parameters())["feature_extractor.model.layer1.0.conv1.weight"]
        assert w.stride()[1] == 1 and w.data_ptr() >= model._flat.data_ptr()
        # DataParallel-era prefix
        renamed = {k.replace("feature_extractor.", "feature_extractor.module."): v for k, v in seeded.items()}
        model.load_state_dict(renamed)
        # vince_parameters(): trunk (+fc) + embedding (+ jigsaw), BN buffers excluded (vince_model.py:96-104)
        assert sum(p.numel() for p in model.vince_parameters()) == sum(seeded[n].numel() for n in vo.param_names(spec))


def test_checkpoint_roundtrip(tmp_path):
    from vince_amd.config import make_args
    from vince_amd.models.vince_model import VinceModel
    args = make_args(save=True, restore=True, checkpoint_dir=str(tmp_path / "ck"), long_save_checkpoint_dir=str(tmp_path / "long"))
    m = VinceModel(args)
    m.load_state_dict(vo.seeded_state(vo.model_spec("ResNet18", 64), 9))
    for it in (100, 200, 300):
        m.save(it, num_to_keep=2)
    files = sorted(f for _, _, fs in os.walk(args.checkpoint_dir) for f in fs)
    assert files == ["000000200.pt", "000000300.pt"]
    m2 = VinceModel(args)
    assert m2.restore() == 300
    for k, v in m.state_dict().items():
        torch.testing.assert_close(m2.state_dict()[k], v, rtol=0, atol=0)


def test_lr_schedule_and_warmup_contract():
    from vince_amd.solvers.base_solver import BaseSolver

    class S(BaseSolver):
        def __init__(self, args):
            self.args, self.epoch, self.iteration, self.train_logger = args, 0, 0, None
            self.optimizer = types.SimpleNamespace(param_groups=[{"lr": args.base_lr, "initial_lr": args.base_lr}])
            self.model = None

    a = types.SimpleNamespace(base_lr=0.03, epochs=200, lr_decay_type="cos", lr_step_schedule=[120, 160])
    s = S(a)
    for ep in (0, 1, 100, 199):
        s.epoch = ep
        assert math.isclose(s.adjust_learning_rate(), 0.03 * 0.5 * (1 + math.cos(math.pi * ep / 200)), rel_tol=1e-12)
    a.lr_decay_type = "step"
    for ep, f in ((0, 1), (119, 1), (120, .1), (160, .01)):
        s.epoch = ep
        assert math.isclose(s.adjust_learning_rate(), 0.03 * f, rel_tol=1e-9)


def test_arg_parser_keeps_reference_flags():
    from vince_amd import arg_parser
    a = arg_parser.parse_args("--title t --description d --solver VinceSolver --backbone ResNet50 --batch-size 256 "
                              "--base-lr 0.03 --vince-embedding-size 128 --vince-queue-size 65536 --vince-momentum 0.999 "
                              "--vince-temperature 0.2 --epochs 200 --lr-decay-type cos --iterations-per-epoch 5005 "
                              "--input-width 224 --input-height 224 --num-frames 1 --num-workers 40 --pytorch-gpu-ids 0 "
                              "--feature-extractor-gpu-ids 0,1,2,3,4,5,6,7 --transform MoCoV2ImagenetTransform".split())
    assert a.backbone.__name__ == "ResNet50" and a.input_size == (224, 224) and a.vince_queue_size == 65536
    assert a.compute_dtype == "bf16" and a.use_warmup and a.save and a.restore
    with pytest.raises(AssertionError):
        arg_parser.parse_args(["--inter-batch-comparison", "--num-frames", "3"])
    with pytest.raises(AssertionError):
        arg_parser.parse_args(["--self-batch-comparison"])


def test_split_dict_by_type_and_stack():
    from vince_amd.models.vince_model import VinceModel
    from vince_amd.solvers.vince_solver import stack_dicts_in_list
    d = {"data": torch.arange(10).view(5, 2), "batch_types": ["a", "b"], "batch_sizes": [2, 3], "tags": ["x", "y"]}
    out = VinceModel.split_dict_by_type(d["batch_types"], d["batch_sizes"], d)
    assert [o["batch_type"] for o in out] == ["a", "b"] and out[1]["data"].shape[0] == 3 and out[0]["tags"] == "x"
    s = stack_dicts_in_list([{"l": torch.tensor(1.0), "n": "q"}, {"l": torch.tensor(3.0), "n": "r"}])
    assert float(s["l"].mean()) == 2.0 and s["n"] == ["q", "r"]


def test_gradient_bucket_plan_partitions_the_flat_buffer():
    from vince_amd import dp
    from vince_amd.config import make_args
    from vince_amd.models.vince_model import VinceModel
    for arch, layers in [("ResNet18", (2, 2, 2, 2)), ("ResNet50", (3, 4, 6, 3))]:
        m = VinceModel(make_args(backbone=arch))
        plan = dp.bucket_plan(m, layers)
        ranges = sorted((a, b) for _, a, b in plan)
        assert ranges[0][0] == 0 and ranges[-1][1] == m._n_train
        assert all(ranges[i][1] == ranges[i + 1][0] for i in range(len(ranges) - 1))
        # buckets come in the order backward finishes them: layer4 (+heads) first, stem/layer1 last
        assert [blk for blk, _, _ in plan] == [sum(layers[:3]), sum(layers[:2]), layers[0], None]
        assert plan[0][2] - plan[0][1] > plan[-1][2] - plan[-1][1]


def test_loader_output_to_batch_dict_layouts():
    """vince_solver.py:180-224: how loader output becomes the batch dict the step reads -- frames of one clip / views of one
    image on consecutive rows, labels repeated per frame (-1 for video)."""
    import torch
    from vince_amd.solvers.vince_solver import VinceSolver
    clips, F = 3, 4
    data = torch.arange(clips * F * 3 * 2 * 2, dtype=torch.float32).view(clips, F, 3, 2, 2)
    b = VinceSolver.process_video_data({"data": data, "queue_data": data + 1000}, F)
    assert b["data"].shape == (clips * F, 3, 2, 2) and b["batch_size"] == clips * F and b["batch_type"] == "video"
    assert torch.equal(b["data"][1 * F + 2], data[1, 2]) and torch.equal(b["queue_data"][2 * F + 3], data[2, 3] + 1000)
    assert b["data_source"] == "YT" and b["num_frames"] == F and (b["imagenet_labels"] == -1).all()
    B = 5
    views = [torch.full((B, 3, 2, 2), float(v)) + torch.arange(B).view(B, 1, 1, 1) * 100 for v in range(2 * F)]
    labels = torch.arange(B) * 7
    s = VinceSolver.process_imagenet_data((views, labels), F)
    assert s["data"].shape == (B * F, 3, 2, 2) and s["data_source"] == "IN" and s["batch_type"] == "images"
    for img in range(B):
        for f in range(F):
            assert float(s["data"][img * F + f, 0, 0, 0]) == f + img * 100
            assert float(s["queue_data"][img * F + f, 0, 0, 0]) == F + f + img * 100
    assert torch.equal(s["imagenet_labels"], labels.repeat_interleave(F))
    one = VinceSolver.process_imagenet_data(([views[0], views[1]], labels), 1)
    assert one["data"] is views[0] and one["queue_data"] is views[1] and one["batch_size"] == B


def test_gradient_buckets_cover_the_flat_buffer_for_resnet50():
    """the four all-reduce buckets (layer4 + heads first, then layer3, layer2, stem + layer1) tile
    [0, n_train) of the flat gradient buffer exactly -- no gap, no overlap -- and come in the order backward finishes them."""
    from vince_amd import dp
    from vince_amd.config import make_args
    from vince_amd.models.vince_model import VinceModel
    from vince_amd.solvers.vince_solver import ARCH_LAYERS
    for arch, embed in (("ResNet50", 128), ("ResNet18", 64)):
        model = VinceModel(make_args(backbone=arch, vince_embedding_size=embed))
        plan = dp.bucket_plan(model, ARCH_LAYERS[arch])
        spans = sorted((a, b) for _, a, b in plan)
        assert spans[0][0] == 0 and spans[-1][1] == model._n_train
        assert all(spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))
        assert [blk for blk, _, _ in plan][-1] is None and plan[0][2] == model._n_train      # heads ride with layer4, stem last
        blocks = [blk for blk, _, _ in plan[:-1]]
        assert blocks == sorted(blocks, reverse=True)                                          # deepest stage first
        if arch == "ResNet50":
            sizes_mb = [(b - a) * 4 / 1e6 for _, a, b in plan]
            # the RCCL message sizes DESIGN.md section 8 quotes (fp32 payload)
            assert abs(sum(sizes_mb) - model._n_train * 4 / 1e6) < 1e-6 and 100 < sum(sizes_mb) < 125


def test_bench_synthetic_batches_are_seeded_per_rank():
    """rank r of `world` draws batch i from Generator(1000 + i * world + r)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("vince_bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    world = 4
    for rank in (0, 3):
        pool = bench.PooledFrames(2, 8, 8, 1, "cpu", pool=3, rank=rank, world=world)
        for i, item in enumerate(pool.items):
            g = torch.Generator().manual_seed(1000 + i * world + rank)
            want = torch.randn(2, 3, 8, 8, generator=g)
            assert torch.equal(item["data"], want)
    o = type("O", (), dict(backbone="ResNet50", size=224, batch=256, queue=65536, embed=128, temperature=0.2, dtype="bf16", mode="moco"))
    assert bench.workload_label(o, 1).startswith("BASELINE config 3:") and bench.workload_label(o, 8).startswith("BASELINE config 4")
    o.backbone, o.queue, o.embed, o.temperature, o.dtype = "ResNet18", 4096, 64, 0.07, "fp32"
    assert bench.workload_label(o, 1).startswith("BASELINE config 2:")
    o.batch = 64
    assert bench.workload_label(o, 1).startswith("not a BASELINE configuration")
