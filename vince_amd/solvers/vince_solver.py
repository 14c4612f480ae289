"""VinceSolver -- training / validation driver of the hot path (reference solvers/vince_solver.py:33-706).

The step order that parity depends on is the reference's (vince_solver.py:386-518):
key forward -> query forward -> dequeue -> similarities -> loss -> metrics -> backward -> SGD -> ENQUEUE -> EMA,
i.e. the queue is read before this step's keys are written, and ``iteration += batch_size``.

Differences that are deliberate:
  * batches come from ``args.batch_source`` (a callable returning the reference's batch dict) or, by default, from
    on-device synthetic frames -- the dataset / augmentation stack is upstream of the path and out of scope;
  * one process per GPU: with torch.distributed initialised, gradients are all-reduced in buckets overlapped with
    backward and keys are all-gathered into a replicated queue (vince_amd/dp.py);
  * no TensorBoard image dumps, no CIFAR kNN in run_val (eval / visualisation only).
"""
import copy
import os
import random
import time
from queue import Queue
from threading import Thread

import torch

from .. import constants, dp, ops
from ..data_source import SyntheticFrames
from ..models.vince_model import U8Frames, VinceModel, VinceQueueModel
from ..optim import FlatSGD
from ..utils.storage_queue import StorageQueue
from .base_solver import BaseSolver
from .meters import AverageMeter

ARCH_LAYERS = {"ResNet18": (2, 2, 2, 2), "ResNet50": (3, 4, 6, 3)}


def stack_dicts_in_list(dicts, concat=False):
    """pt_util.stack_dicts_in_list (App. B): dict of stacked / concatenated tensors; other values collected in lists."""
    out = {}
    for key in dicts[0]:
        vals = [d[key] for d in dicts]
        if isinstance(vals[0], torch.Tensor):
            out[key] = torch.cat(vals, 0) if concat else torch.stack(vals, 0)
        else:
            out[key] = vals
    return out


_KEY_STREAMS = {}


def _shared_key_stream(device):
    """The key encoder's side stream, ONE per device for the whole process: streams are dealt to the runtime's hardware queues in creation
    order, so a solver built later in the process must reuse the stream of the first instead of creating another (csrc/trunk.hip
    shared_stream has the measurement)."""
    key = (device.type, device.index)
    if key not in _KEY_STREAMS:
        with torch.cuda.device(device):
            _KEY_STREAMS[key] = torch.cuda.Stream()
    return _KEY_STREAMS[key]


class VinceSolver(BaseSolver):
    def __init__(self, args, train_logger=None, val_logger=None):
        # (attribute names are the reference's: end-task code and drivers reach into them)
        self.num_frames, self.batch_count = args.num_frames, 0
        self.train_batch_fns, self.val_batch_fns = [], []
        self.vince_queue, self.queue_model = None, None      # StorageQueue / VinceQueueModel, built by setup_model
        self.batch_queue, self.prefetch_thread, self.kill_thread = Queue(2), None, False
        self.drawn_this_epoch = False
        self.reducer = None
        self._dp_step = 0
        self._loss_latch = None      # int64[2] on the device: (non-finite losses seen, first offending iteration + 1)
        self._key_stream = None
        self._jigsaw_rng = None
        self.overlap_key_encoder = bool(int(os.environ.get("VINCE_OVERLAP_KEY", "1")))
        super().__init__(args, train_logger, val_logger)

    # ------------------------------------------------------------------------------------------ setup
    def _torch_device(self):
        dev = self.args.pytorch_gpu_ids[0]
        return torch.device(dev) if isinstance(dev, str) else torch.device("cuda:%d" % dev)   # vince_solver.py:269

    def setup_dataloader(self):
        src = getattr(self.args, "batch_source", None)
        if src is None:
            w, r = dp.world()
            h, wd = self.args.input_size
            src = SyntheticFrames(self.args.batch_size, h, wd, self.args.num_frames, device=self._torch_device(),
                                  rank=r, world=w)
        self.train_batch_fns = src if isinstance(src, (list, tuple)) else [src]
        self.val_batch_fns = list(getattr(self.args, "val_batch_source", None) or [])

    @property
    def iterations_per_epoch(self):
        return self.args.iterations_per_epoch

    @property
    def samples_per_step(self):
        """Samples one training step consumes over the whole job: args.batch_size is the PER-RANK batch here, the reference's is the
        batch nn.DataParallel splits over its GPUs (models/vince_model.py:35) -- so a data-parallel run and a single process at the same
        global batch agree on `iteration`, on the epoch a checkpoint resumes into and on the schedule position."""
        return int(self.args.batch_size) * dp.world()[0]

    def setup_other(self):
        pass   # CIFAR kNN data (vince_solver.py:236-250) is an eval-only extra

    def setup_optimizer(self):
        # vince_solver.py:252-265: SGD(lr=base_lr, weight_decay=1e-4, momentum=0.9) on model.parameters()
        self.optimizer = FlatSGD(self.model, lr=self.args.base_lr, momentum=0.9, weight_decay=0.0001)
        w, _ = dp.world()
        if w > 1 or (os.environ.get("VINCE_FORCE_DP") and torch.distributed.is_initialized()):
            # (VINCE_FORCE_DP exercises the bucketed all-reduce machinery with a single-rank process group)
            comm = None
            if self.model.device.type == "cuda":
                # stream budget (include/vince_hip.h vince_set_side_streams): main, key encoder (= gradient all-reduce during
                # backward), weight gradients, downsample branch
                from .._lib import lib
                # (round 6: decided by bench.py's `dp_forced_single_rank` leg -- single-rank RCCL group, one MI355X: 22.14 ms with the
                # downsample stream kept against 22.40 without, 21.94 for the plain step; RCCL enqueues its kernels on the stream it is
                # handed (the key-encoder stream here), so the process stays at four active streams.  1 = give the downsample stream up)
                lib().vince_set_side_streams(int(os.environ.get("VINCE_DP_SIDE_STREAMS", "2")))
                if self.overlap_key_encoder:
                    self._key_stream = _shared_key_stream(self.model.device)
                    comm = self._key_stream
            self.reducer = dp.GradientReducer(self.model, ARCH_LAYERS[self.model.feature_extractor.arch], comm_stream=comm,
                                              payload=getattr(self.args, "dp_grad_payload", None))
            self.optimizer.grad_scale = 1.0 / w
            if hasattr(self.model, "imagenet_decoders"):
                self.reducer.sync_extra_parameters(self.model.imagenet_decoders.parameters())
        # the optimiser and the key encoder's EMA run beside the stem's weight gradient, the last launch of backward (engine deferred stem
        # join); under data parallelism the last gradient bucket (stem + layer1) is what arrives late: dp.GradientReducer holds its
        # all-reduce behind the stem event and the optimiser steps that range last.  VINCE_DEFER_STEM=0: off (A/B measurements)
        self.defer_stem = self.model.device.type == "cuda" and os.environ.get("VINCE_DEFER_STEM", "1") != "0"
        self.model.defer_stem_join = self.defer_stem
        self.print_optimizer()

    def setup_model(self):
        device = self._torch_device()
        if device.type == "cuda":
            # the reference remaps devices through CUDA_VISIBLE_DEVICES (arg_parser.py:203-208) so its model always sits on
            # the current device; here the ordinal is used directly, so make it current -- stream handles
            # (ops.stream_ptr, the key-encoder stream) are taken from the CURRENT device
            torch.cuda.set_device(device)
        args = copy.copy(self.args)
        args.title = os.path.join(getattr(args, "title", "vince"), "VinceModel")
        if getattr(args, "checkpoint_dir", None):
            base = getattr(args, "base_logdir", constants.BASE_LOG_DIR)
            args.checkpoint_dir = os.path.join(base, args.title, *(args.checkpoint_dir.split(os.sep)[2:]))
            if getattr(args, "long_save_checkpoint_dir", None):
                args.long_save_checkpoint_dir = os.path.join(
                    base, args.title, *(args.long_save_checkpoint_dir.split(os.sep)[2:-1]), constants.TIME_STR)
        model = self.model = VinceModel(args)
        self.iteration = model.restore()      # the checkpoint's sample counter (0: fresh start)
        model.to(device)
        w, _ = dp.world()
        if w > 1:   # replicas must start identical (the reference re-broadcasts parameters every forward)
            torch.distributed.broadcast(self.model._flat, src=0)
            self.model._touch()
        key_model = self.queue_model = VinceQueueModel(args, model)
        # the training loop never reads spatial_features after the next forward: skip the per-forward copy (51 MB at R50/B=256)
        self.model.clone_spatial = False
        self.queue_model.queue_network.clone_spatial = False
        key_model.to(device)
        self.vince_queue = StorageQueue(args.vince_queue_size, args.vince_embedding_size, device=device,
                                        keep_images=getattr(args, "keep_queue_images", False))
        if w > 1:
            torch.distributed.broadcast(self.vince_queue.vector_queue, src=0)
        # `iteration` counts SAMPLES of the whole job (vince_solver.py:514 with the reference's batch_size = the batch of all GPUs
        # behind nn.DataParallel): under data parallelism every step consumes world x the per-rank batch
        samples_per_epoch = self.args.iterations_per_epoch * self.samples_per_step
        self.epoch = self.iteration // samples_per_epoch
        if self.iteration:
            print("Resuming epoch", self.epoch)
        if getattr(self.args, "prefetch_thread", False):
            self.start_prefetch()
        self.fill_queue_repeat()

    # ------------------------------------------------------------------------------------------ queue fill
    @torch.no_grad()
    def fill_queue_repeat(self):
        # vince_solver.py:315-333: hard-copy the parameters, encode ONE batch, enqueue it repeatedly up to K, then reset
        # tail = 0 / full = False so the first real enqueue overwrites row 0 (App. D item 6).
        self.queue_model.param_update(self.model, 0)
        queue = self.vince_queue
        queue.clear()
        concat, parts = self.get_batch()
        encoded = [(part["data_source"], dp.gather_keys(out["queue_embeddings"]))
                   for part, out in zip(parts, self.queue_model(concat))]
        written = 0
        while written < queue.maxsize:
            for source, keys in encoded:
                queue.enqueue(keys, None, source)
                written += keys.shape[0]
                if written >= queue.maxsize:
                    break
        queue.current_tail, queue.full = 0, False
        print("Queue filled with repeats")

    @torch.no_grad()
    def fill_queue(self):
        """vince_solver.py:293-313: the other way to warm the queue -- keys of as many DIFFERENT batches as it takes to write
        K rows (the last batch may wrap the ring, so the queue ends up full with the tail wherever the overshoot left it)."""
        queue, written = self.vince_queue, 0
        self.queue_model.param_update(self.model, 0)
        queue.clear()
        print("Filling queue")
        while written < queue.maxsize:
            concat, parts = self.get_batch()
            for part, out in zip(parts, self.queue_model(concat)):
                keys = dp.gather_keys(out["queue_embeddings"])
                queue.enqueue(keys, part.get("queue_data_cpu"), part["data_source"])
                written += keys.shape[0]
                if written >= queue.maxsize:
                    break
        print("Queue filled")

    # ------------------------------------------------------------------------------------------ loader output -> batch dict
    @staticmethod
    def process_video_data(batch, num_frames):
        """vince_solver.py:212-224: a video loader hands over [clips, frames, 3, H, W] for `data` and `queue_data`; the path
        wants the frames of a clip as consecutive rows (row c * F + f), no labels."""
        data, queue_data = (batch[k].flatten(0, 1) for k in ("data", "queue_data"))
        n = data.shape[0]
        return {"data": data, "queue_data": queue_data, "data_source": "YT", "batch_type": "video", "batch_size": n,
                "num_frames": num_frames, "imagenet_labels": torch.full((n,), -1, dtype=torch.int64)}

    @staticmethod
    def process_imagenet_data(sample, num_frames):
        """vince_solver.py:180-200: (list of 2F augmented views [B, 3, H, W] each, labels [B]) -> views 0..F-1 are the query
        frames, F..2F-1 the key frames, interleaved per image (row b * F + f); labels repeated per frame."""
        views, labels = sample
        q, k = views[:num_frames], views[num_frames:]
        if num_frames > 1:
            data, queue_data = torch.stack(q, dim=1).flatten(0, 1), torch.stack(k, dim=1).flatten(0, 1)
            labels = labels.repeat_interleave(num_frames)
        else:
            data, queue_data = q[0], k[0]
        return {"data": data, "queue_data": queue_data, "imagenet_labels": labels, "data_source": "IN",
                "num_frames": num_frames, "batch_type": "images", "batch_size": data.shape[0]}

    def reset_epoch(self):
        # (once per epoch: the loaders' epoch boundary synchronises anyway; every rank gets here together, so the verdict is shared)
        self.check_loss_latch(shared=True)
        super().reset_epoch()
        self.queue_model.train()   # the key encoder never leaves train mode (vince_solver.py:337)
        self.drawn_this_epoch = False

    # ------------------------------------------------------------------------------------------ batches
    def _next_batches(self):
        batches, n_loaders, device = [], len(self.train_batch_fns), self.model.device
        for _ in range(n_loaders):
            which = self.batch_count % n_loaders
            raw = self.train_batch_fns[which](which)
            if raw is None:
                return None
            self.batch_count = self.batch_count + 1
            moved = {k: (v.to(device) if isinstance(v, (torch.Tensor, U8Frames)) else v) for k, v in raw.items()}
            moved.setdefault("queue_data_cpu", None)
            batches.append(moved)
        if n_loaders == 1:
            concat = {k: v if isinstance(v, (torch.Tensor, U8Frames)) else [v] for k, v in batches[0].items()}
        else:
            concat = stack_dicts_in_list(batches, concat=True)
        concat["batch_types"] = concat.pop("batch_type")
        concat["batch_sizes"] = concat.pop("batch_size")
        return concat, batches

    def prefetch_batches(self):
        while not self.kill_thread:
            self.batch_queue.put(self._next_batches())

    def start_prefetch(self):
        self.prefetch_thread = Thread(target=self.prefetch_batches, daemon=True)
        self.prefetch_thread.start()

    def end(self):
        self.kill_thread = True

    def get_batch(self):
        # vince_solver.py:372-384: from the prefetch thread's queue when there is one; a None (the loaders' epoch ended) refills the
        # negative queue with repeats and draws again
        draw = self.batch_queue.get if self.prefetch_thread is not None else self._next_batches
        while True:
            drawn = draw()
            if drawn is not None:
                return drawn
            self.fill_queue_repeat()

    # ------------------------------------------------------------------------------------------ cross-rank shuffle-BN
    def _encode_keys(self, concat_batch, jig_key):
        """Key-encoder forward.  With `args.dp_shuffle_bn` (data parallel only) the key IMAGES are permuted across ranks
        first, so the batch each rank's key BatchNorms see is a random mix of the global batch -- MoCo's shuffle-BN, which
        the reference gets from permuting before the DataParallel scatter (vince_model.py:137-142).  Every rank then
        all-gathers the keys, undoes the permutation and keeps the rows of its own queries; the full natural-order block is
        reused for the enqueue.  Off by default: it costs an all_to_all of the key images per step."""
        w, r = dp.world()
        shuffle_dp = (getattr(self.args, "dp_shuffle_bn", False) and self.reducer is not None and not jig_key
                      and len(concat_batch["batch_sizes"]) == 1)
        if not shuffle_dp:
            return self.queue_model(concat_batch, jigsaw=jig_key, shuffle=True), None
        B = concat_batch["queue_data"].shape[0]
        perm = dp.global_permutation(w * B, step=self._dp_step, seed=17)
        self._dp_step += 1
        mixed = dict(concat_batch)
        mixed["queue_data"] = dp.exchange_rows(concat_batch["queue_data"], perm)
        out = self.queue_model(mixed, jigsaw=False, shuffle=True)
        natural = dp.unpermute_gathered(dp.gather_keys(out[0]["queue_embeddings"]), perm)
        mine = {"queue_embeddings": natural[r * B:(r + 1) * B].contiguous()}
        return [mine], natural

    def save(self, num_to_keep=-1, sync=False):
        """Replicas are bit-identical, so only rank 0 writes (concurrent ranks would race on one file and on the pruning of
        old checkpoints).  sync=True (only the periodic save inside run_train_iteration, which every rank reaches at the same
        iteration): the others wait so nobody runs ahead into a collective while rank 0 is still on the disk.  The default --
        what a driver's `finally: solver.save()` gets (solver_runner.py:47-54 of the reference), possibly on an exception path
        that only SOME ranks took -- is no collective at all: a rank that failed must not enter a barrier its peers answer with a
        gradient all-reduce."""
        w, r = dp.world()
        # The reference asserts a finite loss BEFORE every backward (vince_solver.py:446), so what it saves is always finite; here the
        # check is a device-side latch, read now: a model that has taken NaN steps is not written (and old checkpoints are not pruned).
        # Under sync every rank reaches this point, so the verdict is shared first -- one rank raising alone would strand its peers.
        self.check_loss_latch(shared=sync)
        if r == 0:
            self.model.save(self.iteration, num_to_keep)
        if w > 1 and sync:
            torch.distributed.barrier()

    def _jigsaw_coin(self):
        """Which side is jigsawed this step (vince_solver.py:397-403).  The reference is one process and makes one
        `random.random()` draw; with one process per GPU every rank must make the SAME choice -- otherwise the ranks touch
        different heads, the all-reduced gradient mixes both and each rank steps a different segment -- so data-parallel
        runs draw from a generator every rank seeds identically (`args.jigsaw_seed`, default 0) and advance in lockstep."""
        w, _ = dp.world()
        if w == 1 and getattr(self.args, "jigsaw_seed", None) is None:
            return random.random()
        if self._jigsaw_rng is None:
            self._jigsaw_rng = random.Random(0x5EED + int(getattr(self.args, "jigsaw_seed", None) or 0))
        return self._jigsaw_rng.random()

    # ------------------------------------------------------------------------------------------ the hot loop
    def _watch_loss(self, loss):
        if self._loss_latch is None:
            self._loss_latch = torch.zeros(2, dtype=torch.int64, device=loss.device)
        ops.nonfinite_latch(loss.detach().float().reshape(1), self.iteration, self._loss_latch)

    def check_loss_latch(self, context=None, shared=False):
        """Host side of the per-iteration finite-loss check: raises with the FIRST offending iteration (synchronises).
        shared=True -- only at points EVERY rank reaches at the same iteration (log iterations, the epoch boundary, the periodic
        save): the verdict is MAX-reduced over the ranks first, so that all of them raise together; a rank raising alone would leave
        its peers blocked in the next gradient all-reduce / key all-gather."""
        bad = None
        if self._loss_latch is not None:
            n, first = (int(v) for v in self._loss_latch.tolist())
            if n:
                bad = AssertionError("non-finite loss in %d iteration(s), first at iteration %d%s"
                                     % (n, first - 1, "" if context is None else " (now: %r)" % (context,)))
        if shared and dp.world()[0] > 1:
            flag = torch.tensor([1.0 if bad is not None else 0.0], device=self.model.device)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
            if bad is None and float(flag) > 0:
                bad = AssertionError("non-finite loss on another rank")
        if bad is not None:
            raise bad

    def run_train_iteration(self):
        began = time.time()
        clock = [began]

        def lap(meter):      # the reference's four host-side timers: time since the previous lap into self.time_meters[meter]
            now = time.time()
            self.time_meters[meter].update(now - clock[0])
            clock[0] = now

        concat_batch, batch_parts = self.get_batch()
        lap("data_cache_time")
        # measurement hook (tools/step_phases.py): a list here collects one CUDA event per phase boundary of this iteration
        marks = getattr(self, "phase_marks", None)

        def mark(name):
            if marks is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append((name, ev))
        mark("start")

        # key encoder (no grad) and query encoder (vince_solver.py:397-406).  The two forwards are independent (different
        # weights, different workspaces), so the key encoder runs on a side HIP stream and its kernels fill the launch
        # tails of the query encoder's; the main stream joins it before the similarity stage.
        if self.args.jigsaw:
            jig_key = self._jigsaw_coin() < 0.5
            jig_query = not jig_key
        else:
            jig_key = jig_query = False
        on_gpu = self.model.device.type == "cuda"
        if on_gpu and self.overlap_key_encoder:
            main = torch.cuda.current_stream()
            if self._key_stream is None:
                self._key_stream = _shared_key_stream(self.model.device)
            self._key_stream.wait_stream(main)
            with torch.cuda.stream(self._key_stream):
                queue_batches, gathered_keys = self._encode_keys(concat_batch, jig_key)
            outputs = self.model.get_embeddings(concat_batch, jigsaw=jig_query, shuffle=True)
            main.wait_stream(self._key_stream)
            for qb in queue_batches:      # produced on the side stream, consumed (and later freed) on the main one
                for v in qb.values():
                    if isinstance(v, torch.Tensor):
                        v.record_stream(main)
            if gathered_keys is not None:
                gathered_keys.record_stream(main)
        else:
            queue_batches, gathered_keys = self._encode_keys(concat_batch, jig_key)
            outputs = self.model.get_embeddings(concat_batch, jigsaw=jig_query, shuffle=True)

        lap("forward_time")
        mark("forwards")

        loss_list, metrics_list = [], []
        batch_parts = self.model.split_dict_by_type(concat_batch["batch_types"], concat_batch["batch_sizes"],
                                                      concat_batch)
        for image_batch, queue_batch, output in zip(batch_parts, queue_batches, outputs):
            # one dict per batch type, later entries win as in the reference's chain of updates: queue view, batch fields, keys
            output.update({**self.vince_queue.dequeue(), **image_batch, **queue_batch})
            output.update(self.model(output))
            weighted = {name: weight * value for name, (weight, value) in self.model.loss(output).items()}
            loss_list.append(weighted)
            metrics_list.append(self.model.get_metrics(output))

        if len(loss_list) == 1:
            # one batch type (the usual case): the mean over a stack of one is the value itself -- skip the ~10 tiny
            # stack / mean launches that would sit between the loss and the backward pass
            loss_dict, metrics = loss_list[0], metrics_list[0]
        else:
            loss_dict = {k: v.mean() for k, v in stack_dicts_in_list(loss_list).items()}
            metrics = {k: v.mean() for k, v in stack_dicts_in_list(metrics_list).items()}

        # what is optimised is the plain sum of the weighted terms; the meter names are the dict's keys (logging contract)
        updated_loss_meters = set(loss_dict)
        terms = list(loss_dict.values())
        loss = terms[0] if len(terms) == 1 else torch.stack([t.reshape(()) for t in terms]).sum()
        # every iteration, without a host sync: a NaN / inf loss latches the step index on the device (ops.nonfinite_latch); the
        # latch is read wherever the host synchronises anyway -- log iterations, every save() before it writes, the start of each
        # epoch -- and raises there
        self._watch_loss(loss)

        lap("metrics_time")
        mark("loss")
        self.optimizer.zero_grad()
        if self.reducer is not None:
            self.reducer.begin_step()
        loss.backward()
        if self.reducer is not None:
            self.reducer.reduce_after_backward()
        if getattr(self, "defer_stem", False):
            self.optimizer.step(defer_stem=True)     # (conv1.weight's step is finished inside vince_update below)
        else:
            self.optimizer.step()
        lap("backward_time")
        mark("backward+sgd")

        for image_batch, output in zip(batch_parts, outputs):
            # update queue (after the optimizer step, before the EMA, vince_solver.py:497-499); with several ranks
            # every rank enqueues the same world*B block in rank order
            keys = gathered_keys if gathered_keys is not None else dp.gather_keys(output["queue_embeddings"])
            self.vince_queue.enqueue(keys, image_batch.get("queue_data_cpu"), image_batch["data_source"])
        self.queue_model.vince_update(self.model)
        mark("enqueue+ema")

        step_no = self.logger_iteration
        if step_no % self.args.log_frequency == 0:
            # the only host synchronisation of the step: scalar read-back for the meters (the reference's
            # assert torch.isfinite(loss), vince_solver.py:446, synchronises every step)
            vals = {k: float(v.detach()) for k, v in loss_dict.items()}
            self.check_loss_latch(vals, shared=True)
            for name, v in vals.items():
                self.loss_meters[name].update(v)
            if "total_loss" in self.loss_meters:       # (a meter the reference keeps beside the named terms)
                self.loss_meters["total_loss"].update(sum(vals.values()))
                updated_loss_meters.add("total_loss")
            for name, value in metrics.items():
                self.metric_meters[name].update(float(value))
            if self.train_logger is not None:
                tag = self.full_name
                record = {"times/%s/%s" % (tag, k): m.val for k, m in self.time_meters.items()}
                record.update({"losses/%s/%s" % (tag, k): self.loss_meters[k].val for k in updated_loss_meters})
                record.update({"metrics/%s/%s" % (tag, k): self.metric_meters[k].val for k in metrics})
                self.train_logger.dict_log(record, self.iteration)

        if step_no % self.args.save_frequency == 0:
            self.save(5, sync=True)     # (checks the finite-loss latch first: a NaN model is never checkpointed)

        self.iteration += self.samples_per_step       # the reference counts SAMPLES (vince_solver.py:514); all ranks' samples under DP
        self.time_meters["total_time"].update(time.time() - began)
        self.logger_iteration += 1
        return loss_dict, metrics

    # ------------------------------------------------------------------------------------------ validation
    def knn_eval(self, data, labels, k=10):
        from .. import constants
        from ..utils.knn_eval import knn_accuracy
        device = self.model.device
        mean = torch.from_numpy(constants.IMAGENET_MEAN).to(device).view(1, -1, 1, 1)
        std = torch.from_numpy(constants.IMAGENET_STD).to(device).view(1, -1, 1, 1)
        feats = []
        with torch.no_grad():
            for i in range(0, data.shape[0], self.args.batch_size):
                x = (data[i:i + self.args.batch_size].to(device=device, dtype=torch.float32) - mean) / std
                feats.append(self.model.get_embeddings({"data": x})["embeddings"])
        return knn_accuracy(torch.cat(feats, 0), labels.to(device), k)[0]

    def run_val(self):
        """vince_solver.py:520-649 without the CIFAR kNN / image dumps: the QUERY encoder in eval mode (running BN
        statistics), the key encoder still in train mode, no backward, no enqueue, no EMA."""
        if not self.val_batch_fns:
            return {}
        self.model.eval()
        loss_meters = {k: AverageMeter() for k in self.model.loss(None).keys()}
        metric_meters = {k: AverageMeter() for k in self.model.get_metrics(None).keys()}
        with torch.no_grad():
            for fn in self.val_batch_fns:
                while True:
                    batch = fn(0)
                    if batch is None:
                        break
                    device = self.model.device
                    batch = {k: (v.to(device) if isinstance(v, (torch.Tensor, U8Frames)) else v) for k, v in batch.items()}
                    concat = {k: v if isinstance(v, (torch.Tensor, U8Frames)) else [v] for k, v in batch.items()}
                    concat["batch_types"] = concat.pop("batch_type")
                    concat["batch_sizes"] = concat.pop("batch_size")
                    queue_batches = self.queue_model(concat, shuffle=False)
                    outputs = self.model.get_embeddings(concat)
                    batch_parts = self.model.split_dict_by_type(concat["batch_types"], concat["batch_sizes"], concat)
                    for image_batch, queue_batch, output in zip(batch_parts, queue_batches, outputs):
                        output.update({**self.vince_queue.dequeue(), **image_batch, **queue_batch})
                        output.update(self.model(output))
                        for k, v in self.model.loss(output).items():
                            loss_meters[k].update(float(v[0] * v[1]), batch["batch_size"])
                        for k, v in self.model.get_metrics(output).items():
                            metric_meters[k].update(float(v), batch["batch_size"])
        out = {k: m.avg for k, m in loss_meters.items()}
        out.update({k: m.avg for k, m in metric_meters.items()})
        knn_set = getattr(self.args, "knn_dataset", None)
        if knn_set is not None:
            # vince_solver.py:651-679: embed a labelled image set (CIFAR in the reference; `data` in 0..255 NCHW like
            # cifar_dataset.data, `labels`) with the eval-mode encoder and score it with leave-one-out 10-NN
            out["epoch_knn_cifar"] = self.knn_eval(knn_set["data"], knn_set["labels"])
        self.model.train()
        if self.val_logger is not None:
            self.val_logger.dict_log({"losses/%s/%s" % (self.full_name, k): v for k, v in out.items()}, self.iteration)
        return out
