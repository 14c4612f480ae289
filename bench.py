#!/usr/bin/env python
"""bench.py -- encoded frames/s of the VINCE encoder + contrastive training step on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 20 --warmup 5

One "step" = one full VinceSolver.run_train_iteration on one synthetic batch already resident in HBM: key-encoder
forward + query-encoder forward + fused similarity/InfoNCE + metrics + backward + SGD + queue enqueue + EMA
(reference solvers/vince_solver.py:386-518).  Every step encodes 2*B frames per GPU (B keys + B queries), so
value = 2 * B * n_gpus * steps / seconds ("encoded frames/sec/node", BASELINE.json).  Workload = BASELINE config 3:
ResNet-50, 224x224, B=256 per GPU, K=65536, D=128, T=0.2, bf16 trunk / fp32 head+loss, random-init weights,
synthetic N(0,1) frames.  Weak scaling: per-GPU batch fixed.

Extra legs (rank 0, N=1 only unless --no-extras):
  roofline      every kernel family's launches bracketed by hipEvent pairs over instrumented (stream-serialised) steps;
                `roofline.kernel` = the family with the most time per step whatever it is, held against the roof that binds
                it (dense bf16 MFMA 2.5 PFLOP/s, or HBM 8 TB/s for the streaming families), `kernels` lists them all;
                `traffic` / `step_hbm` from the committed PMC passes (profiles/pmc_conv_igemm.json, stamped with the
                kernel-source hash).
  fwd_infonce   forward + InfoNCE on B frames (the north-star quantity), no-grad and grad-enabled.
  x3_step / fp32_step   the same step with fp32 tensors: convolutions as split-half products (compute_dtype "x3": the mode
                that meets the reference's 1e-3 bar, priced against 2.5 PF / 3 -- three half-precision MFMAs per product)
                and as exact fp32 MFMAs (157 TF/s); each with its own forward + InfoNCE time (dtype-matched roofs).
  x3f_step      the x3 forward (the same bar on embeddings and loss) with a mixed-precision backward: the bf16 engine on bfloat16
                copies of the saved tensors (compute_dtype "x3f", round 6); priced against 2.5 PF / 2.
  c2_step / c2_x3_step / c2_x3f_step / c5_step     BASELINE config 2 and config 5's per-GPU work.
  blocks / steady_state   per-block ms per step of 5 x 20 steps (device events at the block boundaries): min / median / max.
  launches_per_step       kernel launches of the library per step, counted by the library over the timed loop.
  dp_forced_single_rank   the step through the data-parallel path on one GPU (single-rank RCCL group), engine side streams 1 / 2,
                and one traced step's gradient-bucket communication slots relative to the start of backward.
  cpu_baseline  the CPU oracle (oracle/vince_oracle.py, a torch-CPU restatement pinned to the reference) timed on the
                host cores at B=16 for a few steps.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# algorithmic work per unit (hook-counted on the reference's resnet.py)
STEP_GFLOP_PER_SAMPLE = {"ResNet50": 32.766, "ResNet18": 14.512}     # key fwd + query fwd + query bwd + similarity
FWD_GFLOP_PER_FRAME = {"ResNet50": 8.2000, "ResNet18": 3.6282}
TRUNK_GFLOP_PER_FRAME = {"ResNet50": 8.1743, "ResNet18": 3.6271}      # conv MACs x 2 per image
# MI355X_MICROARCH.md dense MFMA peaks; "x3" = fp32 tensors multiplied as hi*hi + hi*lo + lo*hi on the half-precision pipe:
# three MFMAs per algorithmic product, so the roof for ALGORITHMIC FLOPs is a third of the 16-bit peak
# "x3f" = the x3 forward with a bf16 backward (a bf16 twin engine on bfloat16 copies of the saved tensors): of the step's algorithmic
# FLOPs the two forwards (half) cost three MFMAs per product and the backward (half) one -> two on average.
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3, "x3": 2500.0 / 3.0, "x3f": 2500.0 / 2.0}
# One tag per kernel family, in the library's order (csrc/common.h VINCE_TAG_*).  bound "mfma": the profiler's `work` is algorithmic
# FLOPs and the roof is the dense MFMA peak of the dtype; bound "hbm": `work` is algorithmic BYTES and the roof is 8 TB/s.
KERNEL_TAGS = [("conv_igemm<%s,%s,%s>" % (t, shape, e), "mfma", t) for t in ("f32", "bf16")
               for shape in ("64ch x 128px", "64ch x 256px", "128ch x 128px", "128ch x 256px") for e in ("fwd", "bwd")] + \
              [("conv_wgrad<f32>", "mfma", "f32"), ("conv_wgrad<bf16>", "mfma", "bf16"),
               ("conv_m8<bf16,256ch x 256px,fwd>", "mfma", "bf16"), ("conv_m8<bf16,256ch x 256px,bwd>", "mfma", "bf16"),
               ("conv_xjoin<join>", "hbm", None), ("conv_xjoin<stats>", "hbm", None), ("conv_xjoin<dgrad>", "hbm", None),
               ("conv3x3_strip", "mfma", "bf16"), ("bn_apply", "hbm", None), ("bn_bwd_apply", "hbm", None),
               ("bn_bwd_reduce", "hbm", None), ("stem_pool_fwd", "hbm", None), ("stem_bwd", "hbm", None)]
HBM_PEAK_GBS = 8000.0


class PooledFrames:
    """Cycles through a small pool of pre-generated on-device batches (inputs resident in HBM before timing)."""

    def __init__(self, batch, h, w, frames, device, pool, rank, world):
        self.items = []
        g = torch.Generator(device=device)
        for i in range(pool):
            g.manual_seed(1000 + i * world + rank)
            data = torch.randn(batch, 3, h, w, generator=g, device=device)
            qdata = data + 0.25 * torch.randn(batch, 3, h, w, generator=g, device=device)
            self.items.append({"data": data, "queue_data": qdata, "batch_type": "images", "batch_size": batch,
                               "data_source": "SYN", "num_frames": frames})
        self.i = 0

    def __call__(self, loader_id=0):
        item = self.items[self.i % len(self.items)]
        self.i += 1
        return dict(item)


class stdout_to_stderr:
    """RCCL prints a version banner on the C-level stdout when a communicator is created: keep fd 1 clean for the one JSON line."""

    @staticmethod
    def _flush_c():
        try:
            ctypes.CDLL(None).fflush(None)      # RCCL printf()s into the C library's buffer: it must drain while fd 1 still points away
        except Exception:
            pass

    def __enter__(self):
        sys.stdout.flush()
        self._flush_c()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        self._flush_c()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    try:
        seen = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
                seen.add((phys, core))
        if seen:
            return len(seen)
    except Exception:
        pass
    return os.cpu_count() or 1


def cpu_baseline(arch, embed, K, T, hw, batch, steps, threads):
    """Times the CPU oracle's full training step on the host cores (baseline, not a target): `threads` torch threads (the physical
    cores), every step timed on its own, the MEDIAN step reported (the mean of 3 steps moved 6.86 -> 3.88 frames/s between two
    rounds on the same CPU model)."""
    from oracle import vince_oracle as vo
    torch.set_num_threads(threads)
    tr = vo.OracleTrainer(arch, embed, K, batch, T, 0.03, seed=0)
    data = vo.gaussian_frames(batch, hw, hw, 1000)
    qdata = data + 0.25 * vo.gaussian_frames(batch, hw, hw, 2000)
    tr.step(data, qdata)   # warm-up
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        tr.step(data, qdata)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2] if len(times) % 2 else 0.5 * (times[len(times) // 2 - 1] + times[len(times) // 2])
    return 2.0 * batch / med, [round(t, 3) for t in times]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    import platform
    return platform.processor() or "unknown"


def kernel_source_hash():
    """sha256 over the HIP sources (csrc/*.hip, *.cpp, common.h) + include/vince_hip.h: what a PMC pass was taken from."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.listdir(os.path.join(ROOT, "vince_amd", "csrc")))
    for f in files:
        if f.endswith((".hip", ".cpp", ".h")):
            h.update(open(os.path.join(ROOT, "vince_amd", "csrc", f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "vince_hip.h"), "rb").read())
    return h.hexdigest()[:16]


def conv_layer_table(arch, batch, hw):
    """(N, Hi, Ci, Co, k, stride, Ho) of every convolution of the trunk, in forward order: torchvision's ResNet-18 / -50 as the
    reference vendors it (models/building_blocks/resnet.py:140-250; stride 2 on the 3x3 of a bottleneck)."""
    rows = [(batch, hw, 3, 64, 7, 2, (hw + 1) // 2)]
    h = ((hw + 1) // 2 + 1) // 2          # after the 3x3 / s2 max-pool
    cin = 64
    if arch == "ResNet50":
        for li, (nb, w) in enumerate(zip((3, 4, 6, 3), (64, 128, 256, 512))):
            for bi in range(nb):
                s = 2 if (li > 0 and bi == 0) else 1
                ho = (h + 1) // 2 if s == 2 else h
                rows += [(batch, h, cin, w, 1, 1, h), (batch, h, w, w, 3, s, ho), (batch, ho, w, 4 * w, 1, 1, ho)]
                if bi == 0:
                    rows.append((batch, h, cin, 4 * w, 1, s, ho))
                cin, h = 4 * w, ho
    else:
        for li, w in enumerate((64, 128, 256, 512)):
            for bi in range(2):
                s = 2 if (li > 0 and bi == 0) else 1
                ho = (h + 1) // 2 if s == 2 else h
                rows += [(batch, h, cin, w, 3, s, ho), (batch, ho, w, w, 3, 1, ho)]
                if bi == 0 and (s == 2 or cin != w):
                    rows.append((batch, h, cin, w, 1, s, ho))
                cin, h = w, ho
    return rows


def wgrad_algorithmic_bytes(arch, batch, hw, esize):
    """Mean ALGORITHMIC bytes of a weight-gradient launch of the workload: x + dy once (esize bytes per element) + dw (fp32) --
    what `traffic` (the PMC bytes of the same family) is held against."""
    tab = conv_layer_table(arch, batch, hw)
    tot = 0.0
    for n, hi, ci, co, k, s, ho in tab:
        tot += (n * hi * hi * (4 if ci == 3 else ci) + n * ho * ho * co) * esize + co * k * k * ci * 4
    return tot / len(tab), len(tab)


def workload_label(opt, world):
    """Names the BASELINE.json configuration the ARGUMENTS describe (never a fixed string: a ResNet-18 fp32 run is config 2)."""
    shape = "%s %dx%d, B=%d per GPU, K=%d, D=%d, T=%g, %s trunk" % (opt.backbone, opt.size, opt.size, opt.batch, opt.queue,
                                                                   opt.embed, opt.temperature, opt.dtype)
    c3 = (opt.backbone, opt.size, opt.batch, opt.queue, opt.embed, opt.dtype) == ("ResNet50", 224, 256, 65536, 128, "bf16") \
        and abs(opt.temperature - 0.2) < 1e-9
    c2 = (opt.backbone, opt.size, opt.batch, opt.queue, opt.dtype) == ("ResNet18", 224, 256, 4096, "fp32")
    if opt.mode == "vince":
        name = "BASELINE config 5 per-GPU work (4 frames per clip, inter-batch + self-batch comparison, jigsaw side)" if \
            (opt.backbone, opt.size, opt.queue) == ("ResNet50", 224, 65536) else "multi-frame + jigsaw mode, not a BASELINE size"
    elif c3:
        name = "BASELINE config 3" if world == 1 else "BASELINE config 4 (config 3 per GPU x %d, data parallel)" % world
    elif c2 and opt.mode == "moco":
        name = "BASELINE config 2"
    else:
        name = "not a BASELINE configuration"
    return "%s: %s, %s, random init" % (name, shape, "MoCo-v2 mode" if opt.mode == "moco" else "VINCE mode")


def measure_copy_ceiling(L, device, nbytes=1 << 30):
    """TB/s (read + write) of the library's own copy kernel on this box: the rate an element-wise pass can be held against
    (`roofline.hbm_achievable`).  Best of the copy's shapes (grid-stride 16 B per lane / 32 B per lane / block-contiguous segments:
    the last one with nt loads and stores is the one that reaches the guide's 6.3 TB/s on a 256 MiB buffer, tools/ceilings.py), nt and
    plain, two grid sizes."""
    src = torch.empty(nbytes, dtype=torch.uint8, device=device).random_(0, 255)
    dst = torch.empty_like(src)
    st = torch.cuda.current_stream().cuda_stream
    best = (0.0, None)
    for nt in (5, 4, 1, 0, 2):          # bit 0 = nt; bits 1-2 = shape (vince_stream_copy)
        for blocks in (8192, 16384):
            for _ in range(2):
                L.vince_stream_copy(dst.data_ptr(), src.data_ptr(), nbytes, blocks, nt, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                L.vince_stream_copy(dst.data_ptr(), src.data_ptr(), nbytes, blocks, nt, st)
            e1.record()
            torch.cuda.synchronize()
            r = 2.0 * nbytes * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e12
            if r > best[0]:
                best = (r, "vince_stream_copy, 1 GiB, %d blocks, %s, %s" % (blocks, ("grid-stride", "32 B per lane", "block-contiguous")[nt >> 1],
                                                                            "nt" if nt & 1 else "plain"))
    del src, dst
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default 100: five blocks of 20, each block timed by device events -- `blocks` -- so that a 1 %% change "
                         "shows through the +-3 %% box-to-box spread; an explicit K is timed as exactly K steps, and a 5 x 20 `steady_state` "
                         "leg follows among the extras when K < 100)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--backbone", default="ResNet50")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--queue", type=int, default=65536)
    ap.add_argument("--embed", type=int, default=128)
    ap.add_argument("--temperature", type=float, default=0.2)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "x3", "x3f"])
    ap.add_argument("--no-extras", action="store_true", help="skip the roofline and cpu_baseline legs")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--fp32-steps", type=int, default=5,
                    help="extra leg: the same step with an fp32 trunk (the precision the reference's own config-3 script runs, "
                         "vince/train_moco_v2.sh:40 has --use-apex commented out); 0 = skip")
    ap.add_argument("--config-steps", type=int, default=5,
                    help="extra legs c2_step / c5_step (BASELINE configs 2 and 5 on this GPU), steps each; 0 = skip")
    ap.add_argument("--mode", default="moco", choices=["moco", "vince"],
                    help="moco: BASELINE config 3 (the headline); vince: config 5's per-GPU work -- 4 frames per clip, inter-batch + "
                         "self-batch comparison (self T 0.03) and the jigsaw head on one side per step")
    ap.add_argument("--input", default="float", choices=["float", "u8aug"],
                    help="float: normalised float frames resident in HBM (the headline metric); u8aug: raw uint8 256x320 frames "
                         "through the GPU input stage (MoCo-v2 recipe: resized crop, grayscale, colour jitter, flip, blur) "
                         "inside the timed step")
    opt = ap.parse_args()
    if opt.steps is None:
        opt.steps = 100

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # VINCE_BENCH_ONE_GPU=1 (debug aid): run all ranks on device 0 through gloo to exercise the multi-rank path of this
    # script on a single-GPU box (NCCL refuses two ranks per device)
    one_gpu = os.environ.get("VINCE_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    fd_guard = None
    if world > 1 or os.environ.get("VINCE_FORCE_DP"):
        fd_guard = stdout_to_stderr()
        fd_guard.__enter__()       # until rank 0 prints its line: communicators are created lazily, at the first collective
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo" if one_gpu else "nccl")
    elif os.environ.get("VINCE_FORCE_DP"):
        # debug aid: a single-rank RCCL group, so that the bucketed all-reduce / key all-gather machinery (its streams and
        # events) runs inside the measured step on a one-GPU box
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1)
    assert world == opt.gpus, "--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (opt.gpus, world)
    device = torch.device("cuda", local)

    from vince_amd import _lib as _vlib
    if not os.path.exists(_vlib.LIB_PATH):
        # a checkout that never ran __graft_entry__.build(): the harness builds (rank 0 first), the product itself never does
        from vince_amd import build as _vbuild
        if rank == 0:
            _vbuild.build(verbose=False)
        if world > 1:
            torch.distributed.barrier()
    from vince_amd.config import make_args
    from vince_amd.solvers.vince_solver import VinceSolver
    from vince_amd._lib import lib

    if opt.input == "u8aug":
        from vince_amd.data_source import AugmentedFrames
        from vince_amd.utils import transforms as T
        g = torch.Generator(device=device)
        g.manual_seed(1000 + rank)
        raw = torch.randint(0, 256, (4 * opt.batch, 256, 320, 3), dtype=torch.uint8, device=device, generator=g)
        pool = AugmentedFrames(raw, T.MoCoV2ImagenetTransform(opt.size, seed=rank), opt.batch)
    else:
        pool = PooledFrames(opt.batch, opt.size, opt.size, 4 if opt.mode == "vince" else 1, device, pool=4, rank=rank, world=world)
    args = make_args(backbone=opt.backbone, batch_size=opt.batch, vince_queue_size=opt.queue,
                     vince_embedding_size=opt.embed, vince_temperature=opt.temperature, compute_dtype=opt.dtype,
                     input_size=(opt.size, opt.size), base_lr=0.03, pytorch_gpu_ids=[local],
                     feature_extractor_gpu_ids=[local], batch_source=pool, log_frequency=10 ** 9,
                     iterations_per_epoch=10 ** 9,
                     **(dict(num_frames=4, inter_batch_comparison=True, self_batch_comparison=True, jigsaw=True,
                             vince_self_temperature=0.03) if opt.mode == "vince" else {}))
    import contextlib
    import io
    import random
    random.seed(1234 + rank)      # the per-step jigsaw side coin (vince_solver.py:397-403) is part of the workload: pin it
    with contextlib.redirect_stdout(io.StringIO() if rank != 0 else sys.stderr):
        solver = VinceSolver(args)
        solver.reset_epoch()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def run_blocks(s_, nblocks, per_block):
        """nblocks x per_block steps back to back; block boundaries are device events on the step's stream (no host synchronisation
        inside: the host keeps running ahead exactly as in a plain loop).  Returns (last step's output, per-block ms per step)."""
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(nblocks + 1)]
        last_ = None
        evs[0].record()
        for b_ in range(nblocks):
            for _ in range(per_block):
                last_ = s_.run_train_iteration()
            evs[b_ + 1].record()
        return last_, evs

    def block_stats(evs, per_block):
        ms_ = [evs[i].elapsed_time(evs[i + 1]) / per_block for i in range(len(evs) - 1)]
        return {"steps_per_block": per_block, "ms_per_step": [round(v, 3) for v in ms_], "min": round(min(ms_), 3),
                "median": round(float(np.median(ms_)), 3), "max": round(max(ms_), 3),
                "how": "device events at the block boundaries on the step's stream; the loop never synchronises the host"}

    for _ in range(opt.warmup):
        solver.run_train_iteration()
    nblocks = 5 if (opt.steps >= 25 and opt.steps % 5 == 0) else 1
    launches0 = lib().vince_launch_count()
    barrier()
    t0 = time.perf_counter()
    last, block_events = run_blocks(solver, nblocks, opt.steps // nblocks)
    t_enqueued = time.perf_counter() - t0      # the host has ENQUEUED every step (nothing in the loop synchronises); the GPU is still running
    barrier()
    dt = time.perf_counter() - t0
    launches_per_step = (lib().vince_launch_count() - launches0) / float(opt.steps)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    loss = float(last[0]["nce_loss"].detach())
    frames_per_s = 2.0 * opt.batch * world * opt.steps / dt
    ms_per_step = 1000.0 * dt / opt.steps
    step_tflop = STEP_GFLOP_PER_SAMPLE.get(opt.backbone, 0.0) * opt.batch / 1000.0
    whole_step_tflops = step_tflop / (ms_per_step / 1000.0)

    out = {
        "metric": "encoded frames/sec/node (full VINCE training step: key fwd + query fwd + InfoNCE + bwd + SGD + enqueue + EMA)",
        "value": round(frames_per_s, 2), "unit": "frames/s", "n_gpus": world, "steps": opt.steps, "warmup": opt.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": opt.dtype, "data": "synthetic",
        "config": {"workload": workload_label(opt, world),
                   "global_batch": opt.batch * world, "frames_per_step": 2 * opt.batch * world,
                   "parallelism": "dp%d" % world, "final_loss": round(loss, 5),
                   "input": ("float32 NCHW frames resident in HBM" if opt.input == "float" else
                             "uint8 256x320 frames resident in HBM, GPU input stage (MoCoV2ImagenetTransform) inside the step")},
        # host side of the step: wall time the Python / ctypes / HIP-runtime enqueue of one step takes, measured as the
        # time the un-synchronised loop needs to RETURN, before the closing barrier.  Close to ms_per_step = the host is the limiter
        # (or the runtime's launch queue is full and back-pressures the host); well below it = the GPU is
        "host_enqueue_ms": round(1000.0 * t_enqueued / opt.steps, 3),
        "host_enqueue_frac": round(t_enqueued / dt, 4),
        "step_mfma_frac": round(whole_step_tflops / PEAK_TFLOPS[opt.dtype], 4),
        "step_tflops_per_gpu": round(whole_step_tflops, 2),
        # launch budget: kernel launches of the HIP library per step, counted by the library itself over the timed loop
        # (csrc/common.h: every launch site); torch's own element-wise launches and runtime copies / fills come on top (~45 per step in
        # the kernel trace, profiles/r06_kernel_stats.txt)
        "launches_per_step": {"library": round(launches_per_step, 1), "counted_by": "vince_launch_count() over the timed loop"},
    }
    if nblocks > 1:
        out["blocks"] = block_stats(block_events, opt.steps // nblocks)

    if not opt.no_extras and world == 1:
        # ---- key-encoder overlap A/B, untraced (VERDICT r4 weak #6a): the same step with the key encoder inline on the main stream
        # and on its own stream, alternating, on this build
        def timed_steps(n):
            barrier()
            ta = time.perf_counter()
            for _ in range(n):
                solver.run_train_iteration()
            te = time.perf_counter() - ta
            barrier()
            return 1000.0 * (time.perf_counter() - ta) / n, 1000.0 * te / n

        saved = solver.overlap_key_encoder
        ab = {"inline_ms": [], "own_stream_ms": []}
        for _ in range(2):
            for flag, name in ((False, "inline_ms"), (True, "own_stream_ms")):
                solver.overlap_key_encoder = flag
                timed_steps(2)
                ab[name].append(round(timed_steps(8)[0], 3))
        solver.overlap_key_encoder = saved
        out["key_overlap_ab"] = dict(ab, what="ms per step, 8 steps per sample, alternating A/B/A/B in one process, no profiler attached; "
                                              "default = %s" % ("own stream" if saved else "inline"))
        # the host's own cost per step with the GPU never back-pressuring it: 3 steps enqueued from an idle device
        torch.cuda.synchronize()
        th = time.perf_counter()
        for _ in range(3):
            solver.run_train_iteration()
        out["host_enqueue_idle_ms"] = round(1000.0 * (time.perf_counter() - th) / 3, 3)
        torch.cuda.synchronize()
        if nblocks == 1:
            # an explicit short --steps (the driver's 20): the same loop once more as 5 blocks of 20, so that the line carries a
            # min / median over 100 steps whatever K was
            barrier()
            _, evs = run_blocks(solver, 5, 20)
            barrier()
            out["steady_state"] = dict(block_stats(evs, 20), steps=100)
            out["steady_state"]["frames_per_s_at_median"] = round(2.0 * opt.batch / (out["steady_state"]["median"] * 1e-3), 1)

    if not opt.no_extras:
        # ---- forward + InfoNCE only (the quantity the north-star roofline target is stated on): query-encoder forward in
        # train-mode BN + fused similarity / loss / metrics, no backward, no optimizer.  B frames per pass.
        batch_fwd = pool(0)
        batch_fwd = {"data": batch_fwd["data"], "batch_types": ["images"], "batch_sizes": [opt.batch]}
        kq = solver.vince_queue.dequeue()["queue_vectors"]
        keys = torch.nn.functional.normalize(torch.randn(opt.batch, opt.embed, device=device), dim=1)

        def fwd_once():
            with torch.no_grad():
                o = solver.model.get_embeddings(batch_fwd)[0]
                o.update(dict(queue_embeddings=keys, queue_vectors=kq, data_source="SYN", num_frames=1))
                o = solver.model(o)
                return solver.model.loss(o)["nce_loss"][1]

        for _ in range(3):
            fwd_once()
        barrier()
        tf0 = time.perf_counter()
        nf = max(5, opt.steps)
        for _ in range(nf):
            fwd_once()
        barrier()
        tfwd = (time.perf_counter() - tf0) / nf
        fwd_tflops = FWD_GFLOP_PER_FRAME.get(opt.backbone, 0.0) * opt.batch / 1000.0 / tfwd
        out["fwd_infonce"] = {"frames_per_s_per_gpu": round(opt.batch / tfwd, 1), "ms": round(tfwd * 1000, 3),
                              "tflops": round(fwd_tflops, 1), "mfma_frac": round(fwd_tflops / PEAK_TFLOPS[opt.dtype], 4),
                              "what": "torch.no_grad() forward (the key encoder's path: nothing saved for backward)"}

        # the same forward with autograd recording: the query encoder's path inside the training step (saves what backward reads)
        def fwd_grad_once():
            o = solver.model.get_embeddings(batch_fwd)[0]
            o.update(dict(queue_embeddings=keys, queue_vectors=kq, data_source="SYN", num_frames=1))
            o = solver.model(o)
            return solver.model.loss(o)["nce_loss"][1]

        for _ in range(2):
            fwd_grad_once()
        barrier()
        tg0 = time.perf_counter()
        for _ in range(nf):
            fwd_grad_once()
        barrier()
        tgrad = (time.perf_counter() - tg0) / nf
        g_tflops = FWD_GFLOP_PER_FRAME.get(opt.backbone, 0.0) * opt.batch / 1000.0 / tgrad
        out["fwd_infonce"]["grad_enabled"] = {"ms": round(tgrad * 1000, 3), "tflops": round(g_tflops, 1),
                                              "mfma_frac": round(g_tflops / PEAK_TFLOPS[opt.dtype], 4)}

        # ---- inference: eval-mode extract_features, BatchNorms folded into the convolutions -----------------
        solver.model.eval()

        def infer_once():
            with torch.no_grad():
                return solver.model.extract_features(batch_fwd["data"])["extracted_features"]

        for _ in range(3):
            infer_once()
        barrier()
        ti0 = time.perf_counter()
        for _ in range(nf):
            infer_once()
        barrier()
        tinf = (time.perf_counter() - ti0) / nf
        solver.model.train()
        inf_tflops = TRUNK_GFLOP_PER_FRAME.get(opt.backbone, 0.0) * opt.batch / 1000.0 / tinf
        out["inference_extract_features"] = {"frames_per_s_per_gpu": round(opt.batch / tinf, 1), "ms": round(tinf * 1000, 3),
                                             "tflops": round(inf_tflops, 1),
                                             "mfma_frac": round(inf_tflops / PEAK_TFLOPS[opt.dtype], 4)}

    if rank == 0 and world == 1 and not opt.no_extras:
        # ---- GPU input stage: one step's 2B augmented views (MoCo-v2 recipe, train_moco_v2.sh:18) from a pool
        # of uint8 256 x 320 frames; device time of crop+resize, colour chain and the blur/normalise/layout kernels, with the
        # host-side parameter draws timed separately (they can run a step ahead).  Not part of `value`.
        try:
            from vince_amd.utils import transforms as T
            tf = T.MoCoV2ImagenetTransform(opt.size, seed=0)
            gpool = torch.randint(0, 256, (opt.batch, 256, 320, 3), dtype=torch.uint8, device=device)
            src = np.tile(np.arange(opt.batch), 2).astype(np.int64)
            th0 = time.perf_counter()
            params = tf.draw(2 * opt.batch, (256, 320), src_index=src)
            t_draw = time.perf_counter() - th0
            for _ in range(2):
                views = tf.apply(gpool, params)
                views.float_tensor(torch.bfloat16 if opt.dtype == "bf16" else torch.float32)
            torch.cuda.synchronize()
            ta0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                views = tf.apply(gpool, params)
                views.float_tensor(torch.bfloat16 if opt.dtype == "bf16" else torch.float32)
            torch.cuda.synchronize()
            t_aug = (time.perf_counter() - ta0) / reps
            out["input_stage"] = {"recipe": "MoCoV2ImagenetTransform", "views": 2 * opt.batch, "source": "uint8 256x320",
                                  "ms": round(t_aug * 1000, 3), "views_per_s": round(2 * opt.batch / t_aug, 1),
                                  "host_draw_ms": round(t_draw * 1000, 3)}
            del gpool, views
        except Exception as e:
            out["input_stage"] = {"error": repr(e)}
        # ---- roofline leg: hipEvent pairs around every conv launch for a few extra steps -------------------------
        # (streams are serialised for these steps so that every launch's event pair times that kernel running alone)
        L = lib()
        L.vince_profile_enable(1)
        saved_overlap, solver.overlap_key_encoder = solver.overlap_key_encoder, False
        for _ in range(opt.profile_steps):
            solver.run_train_iteration()
        torch.cuda.synchronize()
        solver.overlap_key_encoder = saved_overlap
        L.vince_profile_enable(0)
        n = len(KERNEL_TAGS)
        ms, fl, cnt = (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_int64 * n)()
        L.vince_profile_collect(n, ms, fl, cnt)
        kernels = {}
        for i, (name, bound, kdt) in enumerate(KERNEL_TAGS):
            if cnt[i]:
                rate = fl[i] / (ms[i] * 1e-3)
                k = {"launches_per_step": cnt[i] // opt.profile_steps, "avg_us": round(1000.0 * ms[i] / cnt[i], 2),
                     "ms_per_step": round(ms[i] / opt.profile_steps, 3), "bound": bound}
                if bound == "mfma":
                    k["tflops"] = round(rate / 1e12, 1)
                    k["frac"] = round(rate / 1e12 / PEAK_TFLOPS["fp32" if kdt == "f32" else "bf16"], 4)
                else:
                    k["gbs"] = round(rate / 1e9, 1)
                    k["frac"] = round(rate / 1e9 / HBM_PEAK_GBS, 4)
                kernels[name] = k
        # the dominant kernel = the family with the most time per step, whatever it is, held against the roof that binds it
        dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"]) if kernels else None
        pm = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_conv_igemm.json")) as fh:
                pm = json.load(fh)
        except Exception:
            pass
        if dom:
            k = kernels[dom]
            # HBM bytes per launch of that kernel from rocprofv3 PMC passes (FETCH_SIZE doubled per the gfx950 note in
            # MI355X_MICROARCH.md, + WRITE_SIZE), recorded by tools/pmc_summary.py into profiles/ -- bench.py cannot run
            # the counter passes itself
            traffic = traffic_source = traffic_stamp = None
            if pm:
                traffic = pm.get("kernels", {}).get(dom, {}).get("bytes_per_launch")
                traffic_source = pm.get("source")
                # the PMC passes are a separate rocprofv3 run: the file says which kernel sources it was taken from, and a
                # mismatch with what is built now is reported instead of silently quoting a stale number
                traffic_stamp = {"kernel_source_hash": pm.get("kernel_source_hash"), "git_head": pm.get("git_head"),
                                 "stale": pm.get("kernel_source_hash") != kernel_source_hash()}
                if pm.get("step"):
                    # whole-step HBM view: the step as a whole is bandwidth-bound
                    out["step_hbm"] = {"bytes_per_step": pm["step"]["bytes"], "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                       "achieved": round(pm["step"]["bytes"] / (out["ms_per_step"] * 1e-3) / 1e9, 1),
                                       "frac": round(pm["step"]["bytes"] / (out["ms_per_step"] * 1e-3) / 8e12, 4),
                                       "source": pm["source"]}
            per_launch = fl[[t[0] for t in KERNEL_TAGS].index(dom)] / max(1, cnt[[t[0] for t in KERNEL_TAGS].index(dom)])
            if k["bound"] == "mfma":
                out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": k["tflops"], "peak": PEAK_TFLOPS[opt.dtype],
                                   "unit": "TFLOP/s", "frac": k["frac"], "flops_per_launch": round(per_launch)}
            else:
                out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": k["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": k["frac"], "algorithmic_bytes_per_launch": round(per_launch)}
            out["roofline"].update({"traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_source,
                                    "avg_us": k["avg_us"], "launches_per_step": k["launches_per_step"],
                                    "ms_per_step": k["ms_per_step"],
                                    "chosen_by": "largest ms/step over all instrumented kernel families (see `kernels`)"})
            # the largest matrix-core family beside it, when the dominant family is a streaming one (and vice versa)
            other = [kk for kk in kernels if kernels[kk]["bound"] != k["bound"]]
            if other:
                o = max(other, key=lambda kk: kernels[kk]["ms_per_step"])
                out["roofline"]["largest_%s_family" % kernels[o]["bound"]] = dict(kernel=o, **kernels[o])
            if dom.startswith("conv_wgrad"):
                ab, nl = wgrad_algorithmic_bytes(opt.backbone, opt.batch, opt.size, 4 if "f32" in dom else 2)
                out["roofline"]["algorithmic_bytes_per_launch"] = round(ab)
                out["roofline"]["algorithmic_bytes_note"] = ("mean over the %d weight-gradient launches of the layer table: x + dy once + dw in "
                                                             "fp32 (the Gram / algebra launches of the family excluded)" % nl)
                if traffic:
                    out["roofline"]["traffic_over_algorithmic"] = round(traffic / ab, 3)
            if traffic_stamp is not None:
                out["roofline"]["traffic_stale"] = bool(traffic_stamp["stale"])
                out["roofline"]["traffic_taken_from"] = {k2: traffic_stamp[k2] for k2 in ("git_head", "kernel_source_hash")}
                if "step_hbm" in out:
                    out["step_hbm"]["stale"] = bool(traffic_stamp["stale"])
            try:
                ceil, how = measure_copy_ceiling(L, device)
                out["roofline"]["hbm_achievable"] = {"value": round(ceil * 1000.0, 1), "unit": "GB/s", "how": how}
            except Exception as e:
                out["roofline"]["hbm_achievable"] = {"error": repr(e)}
            if traffic:
                # the same kernel's MEASURED bytes against the HBM roof ("HBM is the secondary bound and must be
                # reported alongside"): PMC bytes per launch / measured duration
                gbs = traffic / (k["avg_us"] * 1e-6) / 1e9
                out["roofline"]["hbm"] = {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
                ach = out["roofline"].get("hbm_achievable", {}).get("value")
                if ach:
                    out["roofline"]["hbm"]["frac_of_achievable"] = round(gbs / ach, 4)
        out["instrumented_ms_per_step"] = round(sum(kk["ms_per_step"] for kk in kernels.values()), 3)
        out["kernels"] = kernels
        # ---- extra step legs (a few steps each; every single-GPU BASELINE configuration gets a driver-observed number) --------
        #   fp32_step: config 3 with an fp32 trunk -- the precision the reference's own config-3 script runs
        #              (vince/train_moco_v2.sh:40 has --use-apex commented out) and the mode that meets the 1e-3 embedding bar
        #   c2_step:   BASELINE config 2 (ResNet-18, fp32, B=256, K=4096, D=64, T=0.07)
        #   c5_step:   BASELINE config 5's per-GPU work (4 frames per clip, inter-batch + self-batch comparison, jigsaw side)
        def step_leg(steps, frames=1, gflop_backbone=None, fwd_too=False, **over):
            nonlocal solver
            solver = None
            torch.cuda.empty_cache()
            base = {k_: getattr(args, k_) for k_ in vars(args)}
            bsz = over.get("batch_size", base["batch_size"])
            base["batch_source"] = PooledFrames(bsz, opt.size, opt.size, frames, device, pool=2, rank=rank, world=world)
            base.update(over)
            a2 = make_args(**base)
            with contextlib.redirect_stdout(sys.stderr):
                s2 = VinceSolver(a2)
                s2.reset_epoch()
            for _ in range(3):
                s2.run_train_iteration()
            barrier()
            t2 = time.perf_counter()
            for _ in range(steps):
                s2.run_train_iteration()
            barrier()
            t2 = (time.perf_counter() - t2) / steps
            dt_name = over.get("compute_dtype", base["compute_dtype"])
            leg = {"frames_per_s_per_gpu": round(2.0 * bsz / t2, 1), "ms_per_step": round(t2 * 1000, 3), "steps": steps,
                   "dtype": dt_name}
            gf = STEP_GFLOP_PER_SAMPLE.get(gflop_backbone) if gflop_backbone else None
            if gf:
                tfl = gf * bsz / 1000.0 / t2
                leg.update({"tflops": round(tfl, 1), "mfma_frac": round(tfl / PEAK_TFLOPS[dt_name], 4)})
            if fwd_too:   # the north-star quantity in THIS leg's dtype: no-grad forward + InfoNCE on B frames against its own roof
                fb = base["batch_source"](0) if callable(base["batch_source"]) else None
                fb = fb if fb is not None else pool(0)
                fb = {"data": fb["data"][:bsz], "batch_types": ["images"], "batch_sizes": [bsz]}
                kq2 = s2.vince_queue.dequeue()["queue_vectors"]
                keys2 = torch.nn.functional.normalize(torch.randn(bsz, a2.vince_embedding_size, device=device), dim=1)

                def f_once():
                    with torch.no_grad():
                        o = s2.model.get_embeddings(fb)[0]
                        o.update(dict(queue_embeddings=keys2, queue_vectors=kq2, data_source="SYN", num_frames=1))
                        return s2.model.loss(s2.model(o))["nce_loss"][1]

                for _ in range(2):
                    f_once()
                barrier()
                t3 = time.perf_counter()
                for _ in range(max(3, steps)):
                    f_once()
                barrier()
                t3 = (time.perf_counter() - t3) / max(3, steps)
                ftf = FWD_GFLOP_PER_FRAME.get(gflop_backbone, 0.0) * bsz / 1000.0 / t3
                froof = PEAK_TFLOPS["x3" if dt_name == "x3f" else dt_name]     # (x3f's forward IS x3's)
                leg["fwd_infonce"] = {"ms": round(t3 * 1000, 3), "tflops": round(ftf, 1), "mfma_frac": round(ftf / froof, 4),
                                      "roof_tflops": round(froof, 1)}
            del s2
            torch.cuda.empty_cache()
            return leg

        def dp_leg(side_streams, steps=10):
            """Data-parallel machinery without peers: the headline step with VINCE_FORCE_DP=1 -- a single-rank RCCL
            group, bucket events from the engine's backward, the all-reduce slots on the communication stream (a device copy of each
            bucket stands in for the collective RCCL elides at world size 1: VINCE_DP_TRACE_PROXY), key all-gather, the last bucket
            behind the stem event -- and one traced step's bucket slot offsets relative to backward."""
            nonlocal solver
            import torch.distributed as dist
            solver = None
            torch.cuda.empty_cache()
            own_group = not dist.is_initialized()
            env_keys = ("VINCE_FORCE_DP", "VINCE_DP_TRACE_PROXY", "VINCE_DP_SIDE_STREAMS")
            saved_env = {k_: os.environ.get(k_) for k_ in env_keys}
            os.environ.update({"VINCE_FORCE_DP": "1", "VINCE_DP_TRACE_PROXY": "1", "VINCE_DP_SIDE_STREAMS": str(side_streams)})
            guard = stdout_to_stderr()
            guard.__enter__()
            try:
                if own_group:
                    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                    os.environ.setdefault("MASTER_PORT", "29578")
                    dist.init_process_group("nccl", rank=0, world_size=1)
                base = {k_: getattr(args, k_) for k_ in vars(args)}
                base["batch_source"] = PooledFrames(opt.batch, opt.size, opt.size, 1, device, pool=2, rank=rank, world=world)
                with contextlib.redirect_stdout(sys.stderr):
                    s2 = VinceSolver(make_args(**base))
                    s2.reset_epoch()
                assert s2.reducer is not None
                for _ in range(3):
                    s2.run_train_iteration()
                barrier()
                t2 = time.perf_counter()
                for _ in range(steps):
                    s2.run_train_iteration()
                barrier()
                t2 = (time.perf_counter() - t2) / steps
                s2.reducer.enable_trace()
                s2.run_train_iteration()
                rep = s2.reducer.trace_report()
                s2.reducer.enable_trace(False)
                leg = {"ms_per_step": round(t2 * 1000, 3), "steps": steps, "engine_side_streams": side_streams, "trace": rep}
                del s2
                return leg
            finally:
                if own_group and dist.is_initialized():
                    dist.destroy_process_group()
                for k_, v_ in saved_env.items():
                    if v_ is None:
                        os.environ.pop(k_, None)
                    else:
                        os.environ[k_] = v_
                lib().vince_set_side_streams(2)      # the data-parallel solver lowered the engine's stream budget process-wide
                torch.cuda.empty_cache()
                guard.__exit__()

        is_c3 = opt.mode == "moco" and opt.input == "float"
        if is_c3 and opt.config_steps > 0:
            try:
                out["dp_forced_single_rank"] = {
                    "what": "the headline step through the data-parallel path on ONE GPU (single-rank RCCL group; a device copy of each "
                            "gradient bucket stands in for the elided collective): ms per step against `ms_per_step`, for the engine keeping "
                            "1 side stream (the data-parallel default: weight gradients only) and 2 (+ the downsample branch); `trace` = one "
                            "step's bucket communication slots relative to the start of backward",
                    "plain_ms_per_step": out["ms_per_step"],
                    "side_streams_1": dp_leg(1), "side_streams_2": dp_leg(2)}
            except Exception as e:
                out["dp_forced_single_rank"] = {"error": repr(e)}
        if opt.fp32_steps > 0 and opt.dtype not in ("x3", "x3f") and is_c3:
            try:
                out["x3_step"] = dict(step_leg(max(10, opt.fp32_steps), gflop_backbone=opt.backbone, fwd_too=True, compute_dtype="x3"),
                                      what="fp32 tensors, convolutions as split-half products (f16 hi/lo forward, bf16 hi/lo gradients): the "
                                           "parity mode -- embeddings and loss within 1e-3 of the fp32 reference (tests at the bar: G3, G9, "
                                           "G11c, G12); mfma_frac against 2.5 PF / 3")
            except Exception as e:
                out["x3_step"] = {"error": repr(e)}
        if opt.fp32_steps > 0 and opt.dtype not in ("x3", "x3f") and is_c3:
            try:
                out["x3f_step"] = dict(step_leg(max(10, opt.fp32_steps), gflop_backbone=opt.backbone, compute_dtype="x3f"),
                                       what="mixed precision: the x3 forward (embeddings, keys and loss at the reference's 1e-3 bar -- the same "
                                            "forward kernels, trunk features equal to rounding) also leaves bfloat16 copies of what backward reads "
                                            "(BatchNorm inputs centred) in the workspace of a bf16 twin engine, which runs the backward: the "
                                            "arithmetic of an AMP backward (the reference's --use-apex) behind an fp32-grade forward; gradients "
                                            "against the reference: cosine >= 0.998, sum|g| within 2.2e-2 (tests test_g9/g12/g13/g14[x3f]); "
                                            "mfma_frac against 2.5 PF / 2 (three MFMAs per product forward, one backward)")
            except Exception as e:
                out["x3f_step"] = {"error": repr(e)}
        if opt.fp32_steps > 0 and opt.dtype != "fp32":
            try:
                out["fp32_step"] = step_leg(opt.fp32_steps, gflop_backbone=opt.backbone, fwd_too=is_c3, compute_dtype="fp32")
            except Exception as e:
                out["fp32_step"] = {"error": repr(e)}
        if opt.config_steps > 0 and is_c3:
            try:
                out["c2_step"] = dict(step_leg(opt.config_steps, gflop_backbone="ResNet18", backbone="ResNet18", compute_dtype="fp32",
                                               vince_queue_size=4096, vince_embedding_size=64, vince_temperature=0.07),
                                      workload="BASELINE config 2: ResNet18 %dx%d, B=%d, K=4096, D=64, T=0.07, fp32 trunk"
                                               % (opt.size, opt.size, opt.batch))
            except Exception as e:
                out["c2_step"] = {"error": repr(e)}
            try:
                out["c2_x3_step"] = dict(step_leg(opt.config_steps, gflop_backbone="ResNet18", backbone="ResNet18", compute_dtype="x3",
                                                  vince_queue_size=4096, vince_embedding_size=64, vince_temperature=0.07),
                                         workload="BASELINE config 2 with the split-half convolutions: fp32 tensors and fp32-grade results "
                                                  "(embeddings within 1e-4 of the fp32 reference on G3 / G9 / G11c / G12 / G13) at a multiple of "
                                                  "the fp32 MFMA rate; mfma_frac against 2.5 PF / 3")
            except Exception as e:
                out["c2_x3_step"] = {"error": repr(e)}
            try:
                out["c2_x3f_step"] = dict(step_leg(opt.config_steps, gflop_backbone="ResNet18", backbone="ResNet18", compute_dtype="x3f",
                                                   vince_queue_size=4096, vince_embedding_size=64, vince_temperature=0.07),
                                          workload="BASELINE config 2 in the mixed mode: x3 forward (fp32-grade embeddings and loss: G14), bf16 twin backward")
            except Exception as e:
                out["c2_x3f_step"] = {"error": repr(e)}
            try:
                random.seed(1234 + rank)
                out["c5_step"] = dict(step_leg(opt.config_steps, frames=4, gflop_backbone=opt.backbone, num_frames=4, inter_batch_comparison=True,
                                               self_batch_comparison=True, jigsaw=True, vince_self_temperature=0.03),
                                      flops_note="tflops / mfma_frac count the plain step's trunk + head + similarity FLOPs per frame: the jigsawed "
                                                 "side's nine 75 x 75 tiles are 50 625 pixels against 50 176, its head GEMMs (2048 x 18432) add 0.3 %",
                                      workload="BASELINE config 5 per-GPU work: %s %dx%d, B=%d frames = %d clips x 4 frames, K=%d, inter-batch + "
                                               "self-batch comparison, jigsaw side by a seeded coin, %s trunk"
                                               % (opt.backbone, opt.size, opt.size, opt.batch, opt.batch // 4, opt.queue, opt.dtype))
            except Exception as e:
                out["c5_step"] = {"error": repr(e)}
        # ---- cpu_baseline leg ----------------------------------------------------------------------------------------
        try:
            if opt.cpu_steps <= 0:
                raise RuntimeError("skipped (--cpu-steps 0)")
            cores = physical_cores()
            cb, csteps = 16, opt.cpu_steps
            v, step_s = cpu_baseline(opt.backbone, opt.embed, opt.queue, opt.temperature, opt.size, cb, csteps, cores)
            out["cpu_baseline"] = {"value": round(v, 2), "unit": "frames/s", "cores": cores, "kind": "port",
                                   "cpu_model": cpu_model(), "os_cpu_count": os.cpu_count(),
                                   "torch_threads": torch.get_num_threads(), "step_seconds_sorted": step_s,
                                   "sample": "CPU oracle (torch-CPU restatement pinned to the reference), %s %dx%d, "
                                             "B=%d, K=%d, median of %d full training steps after 1 warm-up, "
                                             "torch.set_num_threads(%d) = the physical cores"
                                             % (opt.backbone, opt.size, opt.size, cb, opt.queue, csteps, cores)}
        except Exception as e:   # the baseline leg must never take the measurement down
            out["cpu_baseline"] = {"value": None, "error": repr(e)}

    if fd_guard is not None:
        fd_guard.__exit__()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        sys.stdout.flush()
        os.dup2(2, 1)              # (whatever the teardown prints)
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
