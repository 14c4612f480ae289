"""Data-parallel glue: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" in CPU tests).

The reference's only multi-GPU mechanism is nn.DataParallel around the trunk (models/vince_model.py:35): per-chunk BN
statistics, gradients reduced to device 0, heads / loss / queue on one GPU.  The MI355X-native equivalent
 keeps the same per-sample semantics with explicit collectives:

  * gradients: SUM all-reduce of the flat fp32 gradient buffer, in stage-sized buckets launched on a side stream as
    soon as the engine's bucket events fire (layer4 + heads first), 1/world folded into the SGD kernel;
  * keys: all-gather of each rank's B x D keys -> every rank enqueues the same world*B block in rank order at the same
    tail, so the replicated queues stay bit-identical;
  * shuffle-BN: optional cross-rank permutation of the key images before encoding (all_to_all) and inverse
    permutation of the gathered keys;
  * BN statistics stay per rank (the reference's per-chunk statistics); EMA and SGD are replicated.
"""
import os

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def is_distributed():
    return world()[0] > 1


# ------------------------------------------------------------------------------------------------ gradient buckets
def bucket_plan(model, arch_layers):
    """Contiguous flat-gradient ranges in the order their gradients become final during backward, with the index of
    the residual block after whose backward the range is complete.  Returns [(block_index or None, start, end)].
    The last bucket (stem + layer1, finished at the very end of backward) has block index None."""
    n_train = model._n_train
    so = model._stage_offsets
    first_block = {}
    acc = 0
    for li, n in enumerate(arch_layers):
        first_block["layer%d" % (li + 1)] = acc
        acc += n
    # heads sit after the trunk in the flat buffer; their gradients are final before the trunk backward starts
    plan = [(first_block["layer4"], so["layer4"], n_train),
            (first_block["layer3"], so["layer3"], so["layer4"]),
            (first_block["layer2"], so["layer2"], so["layer3"]),
            (None, 0, so["layer2"])]
    return plan


class GradientReducer:
    """Bucketed all-reduce of VinceModel's flat gradient buffer, overlapped with the rest of backward."""

    def __init__(self, model, arch_layers, comm_stream=None, payload=None):
        """comm_stream: the stream the bucket all-reduces are enqueued on (each waits for its bucket's event there).  The
        solver passes its key-encoder stream, which is idle during backward: one stream fewer competing for the runtime's
        four hardware queues.
        payload: "fp32" (default) or "bf16" (`args.dp_grad_payload` / VINCE_DP_GRAD_BF16=1): the buckets travel as bfloat16 --
        51 instead of 112 MB per step at ResNet-50 -- and are summed back into the fp32 gradient buffer; an opt-in for links
        where the all-reduce does not hide under backward (the sum of `world` bf16-rounded gradients, so not the reference's
        arithmetic: off by default)."""
        self.payload = payload or ("bf16" if os.environ.get("VINCE_DP_GRAD_BF16") == "1" else "fp32")
        self.model = model
        self.plan = bucket_plan(model, arch_layers)
        self.on_gpu = model._flat.is_cuda
        if self.on_gpu:
            self.comm_stream = comm_stream if comm_stream is not None else torch.cuda.Stream()
            self.events = [torch.cuda.Event() for _ in self.plan]
            for ev in self.events:
                ev.record()          # torch creates the underlying hipEvent lazily; the engine needs a live handle
            self.done = torch.cuda.Event()
            self.done_most = torch.cuda.Event()      # every bucket but the last (stem + layer1): the deferred stem join (see reduce_after_backward)
            model._bucket_events = [(blk, ev) for (blk, _, _), ev in zip(self.plan, self.events) if blk is not None]
            # every bucket but the last is launched from INSIDE the engine's backward call, the moment its event has been recorded
            # (engine.Trunk.set_bucket_callback): the all-reduce of layer4 + heads is queued behind its event while the host is
            # still enqueueing layer3's backward, not after backward() has returned
            model._bucket_hook = self._launch_bucket
            self._launched = set()
            self._hook_error = None
            # VINCE_DP_TRACE_PROXY=1 (single-rank traces): a device copy of each bucket stands in for the collective, which RCCL
            # elides at world size 1 -- so a kernel trace shows WHEN each bucket's communication slot runs relative to backward
            self._proxy = (torch.empty_like(model._flat_grad) if os.environ.get("VINCE_DP_TRACE_PROXY") == "1" and world()[0] == 1
                           else None)
        self._trace = None       # enable_trace(): timing events around every bucket's communication slot (bench.py `dp_forced_single_rank`)

    def enable_trace(self, on=True):
        """Measurement aid: from the next begin_step() on, every bucket's communication slot is bracketed by timing events on the
        communication stream and backward by events on the compute stream; trace_report() reads them (after a synchronise)."""
        self._trace = {} if (on and self.on_gpu) else None

    def _slot(self, e, a, b, inside):
        if self._trace is None or "t0" not in self._trace:
            return self._reduce_bucket(a, b)
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record(self.comm_stream)
        self._reduce_bucket(a, b)
        s1.record(self.comm_stream)
        self._trace["buckets"][e] = (s0, s1, (b - a) * 4, inside)

    def trace_report(self):
        """[{bucket, first block, MB, start_ms, ms, launched}] relative to begin_step() + where backward ended, from the last traced step."""
        t = self._trace
        if not t or "t0" not in t or "tail" not in t:
            return None
        torch.cuda.synchronize()
        rows = []
        for e in sorted(t["buckets"]):
            s0, s1, nbytes, inside = t["buckets"][e]
            rows.append({"bucket": e, "first_block": self.plan[e][0], "MB": round(nbytes / 1e6, 1),
                         "start_ms": round(t["t0"].elapsed_time(s0), 3), "ms": round(s0.elapsed_time(s1), 3),
                         "launched": "inside backward (engine callback)" if inside else "after backward returned"})
        return {"backward_ms": round(t["t0"].elapsed_time(t["tail"]), 3), "all_reduced_ms": round(t["t0"].elapsed_time(t["done"]), 3),
                "buckets": rows}

    def _reduce_bucket(self, a, b):
        grad = self.model._flat_grad
        if self.payload == "bf16":
            buf = grad[a:b].to(torch.bfloat16)
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            grad[a:b].copy_(buf)
        else:
            dist.all_reduce(grad[a:b], op=dist.ReduceOp.SUM)
        if self.on_gpu and self._proxy is not None:
            self._proxy[a:b].copy_(grad[a:b])

    def _launch_bucket(self, e):
        """Engine callback (bucket index e in plan order): wait for the bucket's event on the communication stream, then reduce."""
        try:
            blk, a, b = self.plan[e]
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(self.events[e])
                self._slot(e, a, b, True)
            self._launched.add(e)
        except BaseException as exc:        # a ctypes callback cannot propagate: reduce_after_backward re-raises
            self._hook_error = exc

    def sync_extra_parameters(self, params):
        """Parameters outside the flat buffer (the optional ImageNet side decoders, vince_model.py:79-90 of the reference): their
        per-rank nn.Linear initialisation is replaced by rank 0's, and reduce_after_backward averages their gradients -- otherwise
        the replicas drift apart and rank 0's checkpoint is not the model the other ranks trained."""
        self.extra = [p for p in params]
        if world()[0] > 1:
            for p in self.extra:
                dist.broadcast(p.data, src=0)

    def _reduce_extra(self):
        w = world()[0]
        if w == 1:
            return
        extra = getattr(self, "extra", ())
        if not extra:
            return
        # ONE collective per step over every extra parameter, whether or not this rank's batch gave it a gradient (a rank whose batch
        # carried no labelled data contributes zeros): the collective sequence can then never differ between ranks
        # One "touched" flag per parameter rides at the end of the same buffer: a parameter NO rank produced a gradient for keeps
        # grad = None, so that the optimiser skips it exactly as a single-process run does (torch SGD applies weight decay and
        # momentum to a zero gradient, but not to a missing one -- ADVICE r4)
        dev = extra[0].device
        touched = torch.tensor([0.0 if p.grad is None else 1.0 for p in extra], device=dev)
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in extra] + [touched])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        any_touched = (flat[-len(extra):] > 0).tolist()      # (one host read per step, only when side decoders exist)
        flat.div_(w)
        off = 0
        for p, hit in zip(extra, any_touched):
            n = p.numel()
            if hit:
                g = flat[off:off + n].view_as(p).to(p.dtype)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
            off += n

    def reduce_after_backward(self):
        """Call right after loss.backward(): launches whatever the engine hook has not (the stem + layer1 bucket, final only when
        backward ends; every bucket on CPU / when the trunk was not part of this backward) and makes the compute stream wait."""
        self._reduce_extra()
        if not self.on_gpu:
            for _, a, b in self.plan:
                self._reduce_bucket(a, b)
            return
        if self._hook_error is not None:
            err, self._hook_error = self._hook_error, None
            self._launched.clear()      # (a failed step must not leave buckets marked as reduced for the next one)
            raise err
        cur = torch.cuda.current_stream()
        tail_event = torch.cuda.Event()
        tail_event.record(cur)
        if self._trace is not None and "t0" in self._trace:
            self._trace["tail"] = torch.cuda.Event(enable_timing=True)
            self._trace["tail"].record(cur)
        # Deferred stem join (round 5; models/vince_model.py defer_stem_join): backward has returned with conv1's weight gradient still
        # in flight behind model._stem_event.  Only the LAST bucket (flat range [0, layer2): stem + layer1) contains it: that bucket's
        # all-reduce waits for the event, and the compute stream is made to wait for every OTHER bucket only (done_most) -- the
        # optimiser steps [layer2, end) beside the stem's weight gradient and the last all-reduce, then waits for `done` (model._late).
        deferred = bool(getattr(self.model, "_stem_pending", False))
        last = len(self.plan) - 1
        try:
            with torch.cuda.stream(self.comm_stream):
                for e, (blk, a, b) in enumerate(self.plan):
                    if e in self._launched:
                        continue
                    if deferred and e == last:
                        self.done_most.record(self.comm_stream)
                    self.comm_stream.wait_event(tail_event)
                    if deferred and a == 0:
                        self.comm_stream.wait_event(self.model._stem_event)
                    self._slot(e, a, b, False)
                self.done.record(self.comm_stream)
                if self._trace is not None and "t0" in self._trace:
                    self._trace["done"] = torch.cuda.Event(enable_timing=True)
                    self._trace["done"].record(self.comm_stream)
        finally:
            launched_last = last in self._launched
            self._launched.clear()
        if deferred and not launched_last and self.plan[last][1] == 0:
            cur.wait_event(self.done_most)
            done = self.done
            self.model._late = (self.plan[last][2], lambda: torch.cuda.current_stream().wait_event(done))
        else:
            cur.wait_event(self.done)
            if deferred:
                self.model.finish_stem_grad()

    def begin_step(self):
        """Call before loss.backward(): forgets bucket launches of a step whose backward raised before reduce_after_backward ran."""
        if self.on_gpu:
            self._launched.clear()
            self._hook_error = None
            if self._trace is not None:
                t0 = torch.cuda.Event(enable_timing=True)
                t0.record()
                self._trace = {"t0": t0, "buckets": {}}


# ------------------------------------------------------------------------------------------------ keys
def gather_keys(keys):
    """[B, D] per rank -> [world*B, D], rank order (rank r's rows at [r*B, (r+1)*B))."""
    w, _ = world()
    if w == 1:
        return keys
    out = torch.empty(w * keys.shape[0], keys.shape[1], dtype=keys.dtype, device=keys.device)
    dist.all_gather_into_tensor(out, keys.contiguous())
    return out


def global_permutation(n_global, step, seed=0):
    """The same permutation on every rank (seeded by step) -- which rank encodes which key image (shuffle-BN)."""
    g = torch.Generator().manual_seed(seed * 1000003 + step)
    return torch.randperm(n_global, generator=g)


def exchange_rows(local, perm):
    """Cross-rank gather: returns rows perm[r*B:(r+1)*B] of the GLOBAL tensor whose rank-s slice is `local` on rank s.
    all_to_all_single with uneven splits; row order inside each received block follows the sender's index order, so a
    local re-ordering restores the requested order."""
    w, r = world()
    B = local.shape[0]
    if w == 1:
        return local[perm.to(local.device)]
    want = perm[r * B:(r + 1) * B]                       # global indices this rank must end up with
    # what every rank d wants from me: indices in perm[d*B:(d+1)*B] that fall in my slice
    send_idx, send_counts = [], []
    for d in range(w):
        wd = perm[d * B:(d + 1) * B]
        mine = wd[(wd >= r * B) & (wd < (r + 1) * B)] - r * B
        send_idx.append(mine)
        send_counts.append(int(mine.numel()))
    recv_counts = [int(((want >= s * B) & (want < (s + 1) * B)).sum()) for s in range(w)]
    send = local[torch.cat(send_idx).to(local.device)].contiguous()
    recv = torch.empty((sum(recv_counts),) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=send_counts)
    # recv holds, source rank by source rank, the wanted rows in the order they appear in `want`
    order = torch.cat([torch.nonzero((want >= s * B) & (want < (s + 1) * B)).flatten() for s in range(w)])
    out = torch.empty_like(recv)
    out[order.to(local.device)] = recv
    return out


def unpermute_gathered(gathered, perm):
    """gathered[j] is the key of global sample perm[j]; returns keys in natural global order."""
    out = torch.empty_like(gathered)
    out[perm.to(gathered.device)] = gathered
    return out


def all_reduce_mean_scalars(t):
    w, _ = world()
    if w == 1:
        return t
    t = t.clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t / w
