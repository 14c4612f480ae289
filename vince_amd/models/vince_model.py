"""VinceModel / VinceQueueModel -- the reference's class API (models/vince_model.py:19-613) on the HIP kernels.

What is different underneath
  * All trainable tensors of the path (trunk, projection MLP, jigsaw head) live in ONE flat fp32 device buffer with
    a parallel flat gradient buffer; the ``nn.Parameter`` objects the reference's callers see are views into it
    (conv weights as channels_last views, whose memory is the kernels' [Co][kh][kw][Ci] layout).  SGD, the momentum
    (EMA) update and the data-parallel gradient all-reduce are single passes over that buffer.
  * The trunk runs in the C++ engine (csrc/trunk.hip) in ``args.compute_dtype`` (bf16 or fp32, NHWC); the head
    (avg-pool output, MLP, L2 normalise) and the similarity / InfoNCE stage are always exact fp32 on the f32 MFMA.
  * ``forward`` never materialises the B x (K+1) logits: ``vince_similarities`` is a lazy handle, ``loss`` and
    ``get_metrics`` read the fused kernel's per-row results.
  * Autograd: two coarse ``torch.autograd.Function`` nodes (encoder, InfoNCE) so ``loss.backward()`` works as in
    vince_solver.py:463-468; parameter gradients are written straight into the flat gradient buffer.
There is no CPU path: tensors must be on the GPU.
"""
import ctypes
import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .. import constants, ops
from ..engine import Trunk, nograd_workspace, pointer_table
from ..utils import loss_util
from .base_model import BaseModel

class U8Frames:
    """GPU input stage: raw uint8 HWC frames plus the deterministic augmentation parameters, accepted wherever
    the reference API takes the float NCHW `data` tensor.  `frames`: uint8 [N, Hs, Ws, 3] on the GPU; `size`: (H, W) of the
    network input; `crop_yx`: int32 [N, 2] top-left corner of each frame's H x W window (None: (0, 0)); `flip`: uint8 / bool
    [N] horizontal flip of the window.  The crop, the flip and ToTensor + Normalize ((u8 - 255*mean) / (255*std),
    constants.py:28-29) happen inside the layout kernel that feeds the stem -- no float copy of the frames ever exists."""

    def __init__(self, frames, size, crop_yx=None, flip=None, blur=None):
        self.frames, self.size = frames, (int(size[0]), int(size[1]))
        self.crop_yx = None if crop_yx is None else crop_yx.to(torch.int32).contiguous()
        self.flip = None if flip is None else flip.to(torch.uint8).contiguous()
        # blur: (taps float32 [N, ks], do_blur uint8 [N]) -- RandomGaussianBlur on the normalised tensor (utils/transforms.py
        # SimCLR / Jigsaw / MoCo-v2 recipes), applied inside the layout kernels; frames must already be H x W (no crop window)
        self.blur = None if blur is None else (blur[0].to(torch.float32).contiguous(), blur[1].to(torch.uint8).contiguous())
        if self.blur is not None and (self.crop_yx is not None or tuple(frames.shape[1:3]) != self.size):
            raise ValueError("U8Frames: blur needs frames that are already at the network input size")

    @property
    def shape(self):   # what the float tensor would look like: batch-size logic upstream keeps working
        return (self.frames.shape[0], 3) + self.size

    @property
    def device(self):
        return self.frames.device

    def __len__(self):
        return self.frames.shape[0]

    def __getitem__(self, rows):
        """Batch slicing (split_dict_by_type slices every per-sample entry of a batch dict)."""
        if not isinstance(rows, slice):
            raise TypeError("U8Frames supports slice indexing only")
        pick = lambda t: None if t is None else t[rows]   # noqa: E731
        return U8Frames(self.frames[rows], self.size, pick(self.crop_yx), pick(self.flip),
                        None if self.blur is None else (self.blur[0][rows], self.blur[1][rows]))

    def to(self, device):
        return U8Frames(self.frames.to(device), self.size, None if self.crop_yx is None else self.crop_yx.to(device),
                        None if self.flip is None else self.flip.to(device),
                        None if self.blur is None else (self.blur[0].to(device), self.blur[1].to(device)))

    def float_tensor(self, dtype=torch.float32):
        """The float NCHW tensor these frames stand for, produced by the SAME layout kernels that feed the stem (the
        per-sample contract of the reference's transforms returns this tensor)."""
        from .. import ops
        mean, std = constants.IMAGENET_MEAN, constants.IMAGENET_STD
        if self.blur is not None:
            rows = ops.aug_blur_to_rows(self.frames.contiguous(), dtype, mean, std, self.flip, self.blur[0], self.blur[1])
        else:
            rows = ops.input_u8hwc_to_rows(self.frames.contiguous(), dtype, self.size, mean, std, self.crop_yx, self.flip)
        w = self.size[1]
        return rows[:, :, ops.STEM_LEFT:ops.STEM_LEFT + w, :3].permute(0, 3, 1, 2).contiguous()

    def float_reference(self):
        """The float NCHW tensor this stands for (tests / debugging): same arithmetic, torch ops."""
        n = self.frames.shape[0]
        h, w = self.size
        out = torch.empty(n, 3, h, w, dtype=torch.float32, device=self.frames.device)
        mean = torch.from_numpy(constants.IMAGENET_MEAN).to(self.frames.device).view(3, 1, 1)
        std = torch.from_numpy(constants.IMAGENET_STD).to(self.frames.device).view(3, 1, 1)
        for i in range(n):
            cy, cx = (0, 0) if self.crop_yx is None else [int(v) for v in self.crop_yx[i]]
            win = self.frames[i, cy:cy + h, cx:cx + w].permute(2, 0, 1).float()
            if self.flip is not None and bool(self.flip[i]):
                win = win.flip(-1)
            out[i] = (win - mean) / std
            if self.blur is not None and bool(self.blur[1][i]):   # utils/util_functions.py:119-129
                k = self.blur[0][i]
                ks = k.numel()
                x = out[i][None]
                x = torch.nn.functional.conv2d(x, k.view(1, 1, ks, 1).expand(3, 1, ks, 1), padding=(ks // 2, 0), groups=3)
                x = torch.nn.functional.conv2d(x, k.view(1, 1, 1, ks).expand(3, 1, 1, ks), padding=(0, ks // 2), groups=3)
                out[i] = x[0]
        return out


# VINCE_FOLD_BN=0: eval-mode forwards keep the separate BatchNorm passes (cross-check aid)
FOLD_BN = os.environ.get("VINCE_FOLD_BN", "1") != "0"

_ALIGN = 64  # floats; every tensor in the flat buffers starts on a 256-byte boundary


X3_NAMES = ("x3", "fp32x3", "f32x3")
# "x3f": the x3 forward (embeddings and loss at the reference's 1e-3 bar) with every GRADIENT convolution as single bfloat16 products --
# the arithmetic of a mixed-precision backward (what the reference's --use-apex would run) behind an fp32-grade forward
X3F_NAMES = ("x3f",)


def _compute_dtype(args):
    """args.compute_dtype -> the trunk's tensor dtype.  "x3" (= "fp32x3"): float32 tensors whose convolutions run as split-half
    products on the half-precision matrix pipe (csrc/common.h x3_split): the mode that meets the reference's fp32 results to 1e-3
    (vince/train_moco_v2.sh:40 runs fp32) at a multiple of the fp32 MFMA rate."""
    name = str(getattr(args, "compute_dtype", "fp32")).lower()
    if name in ("bf16", "bfloat16"):
        return torch.bfloat16
    if name in ("fp32", "float32", "f32") + X3_NAMES + X3F_NAMES:
        return torch.float32
    raise ValueError("compute_dtype must be bf16, fp32, x3 or x3f, got %r" % name)


class _TransposedHeads(dict):
    """W^T of the head Linears for their input gradients, made on first use after every parameter update: a model that never
    runs backward (the key encoder) or a head the step does not touch (the jigsaw head in MoCo mode, 151 MB) costs nothing."""

    def __init__(self, params):
        super().__init__()
        self._params = {id(p): p for p in params if p.dim() == 2}

    def __missing__(self, key):
        wt = ops.transpose_f32(self._params[key].data)
        self[key] = wt
        return wt


class _HeadCopies(dict):
    """Split-half compute copies of the head Linears (round 5, bf16 trunk only: the projection MLP as hi*hi + hi*lo + lo*hi of BFLOAT16
    halves -- three 16-bit MFMAs per product, 2^-16 per product, fp32's exponent range -- instead of exact fp32 MFMAs at a sixteenth of
    that rate; the fp32 and x3 trunks, the modes held to the reference's 1e-3 bar, keep the exact fp32 head).  bfloat16 halves, not the
    IEEE-half halves of the trunk's forward: nothing normalises the head's operands, and a half-range overflow (|x| >= 4095) turns
    into NaN by design (csrc/common.h) -- a toy run that diverges for a few steps must not die of its head.
    ops.prepare_weight(x3="b") gives the forward copy and, where a backward will follow, the transposed one in ONE launch, which also
    replaces the W^T launches of _TransposedHeads.  Made on first use after every parameter update."""

    def __init__(self, params):
        super().__init__()
        self._params = {id(p): p for p in params if p.dim() == 2}

    def copies(self, key, need_t):
        have = self.get(key)
        if have is None or (need_t and have[1] is None):
            p = self._params[key]
            co, ci = p.shape
            have = ops.prepare_weight(p.data.view(co, 1, ci), torch.float32, want_transposed=need_t, x3="b")
            self[key] = have
        return have


def _head_x3_shape(w):
    """Which head Linears take the split-half route: both dimensions multiples of 16 (the split weight layout) and at most 2^23
    elements -- the copies are rebuilt after every parameter update (read once, written twice), which a 2048 x 2048 weight repays in its
    three GEMMs and the jigsaw head's 2048 x 18432 (151 MB, 64 rows per step) does not: 0.53 ms of preparation per step for GEMMs that
    read the weight at the HBM rate either way (`profiles/r05_c5_kernel_stats_before_head_limit.txt`)."""
    return w.shape[0] % 16 == 0 and w.shape[1] % 16 == 0 and w.numel() <= (1 << 23)


def head_x3():
    """VINCE_HEAD_X3=0: the head stays on exact fp32 MFMAs (cross-check / A-B switch)."""
    return os.environ.get("VINCE_HEAD_X3", "1") != "0"


class LazySimilarities:
    """Stand-in for the reference's ``vince_similarities`` tensor: shape is known, values are produced on demand."""

    def __init__(self, q, inb, queue, moco_mode, include_inb=True):
        self._q, self._inb, self._queue, self._moco, self._include_inb = q, inb, queue, moco_mode, include_inb
        ncols = ((1 if moco_mode else inb.shape[0]) if include_inb else 0) + (0 if queue is None else queue.shape[0])
        self.shape = torch.Size((q.shape[0], ncols))
        self._value = None

    def materialize(self):
        if self._value is None:
            q = self._q.detach().contiguous()
            parts = []
            if not self._include_inb:
                pass
            elif self._moco:
                parts.append((q * self._inb).sum(1, keepdim=True))          # vince_model.py:227
            else:
                parts.append(ops.linear_fwd(q, self._inb.contiguous(), None))  # q @ k.T
            if self._queue is not None:
                parts.append(ops.linear_fwd(q, self._queue.contiguous(), None))
            self._value = torch.cat(parts, dim=1)
        return self._value

    def __array__(self, dtype=None):
        return self.materialize().cpu().numpy()


class _InfoNCEFn(torch.autograd.Function):
    """Fused (B x D) . (D x (Bk+K)) similarity + InfoNCE.  Gradient flows to the query operand only (keys and the
    queue are detached, vince_model.py:610, storage_queue.py:53); for the self-similarity term both operands are q."""

    @staticmethod
    def forward(ctx, q, inb, queue, temperature, frames, offdiag_neg, self_sim):
        ctx.set_materialize_grads(False)
        qc = q.detach().contiguous()
        inbc = qc if self_sim else inb.detach().contiguous()
        # (a forward that will be differentiated keeps its logits for backward: 67 MB at B = 256, K = 65536)
        r = ops.infonce_fwd(qc, inbc, queue, temperature, frames=frames, offdiag_neg=offdiag_neg,
                            save_logits=bool(ctx.needs_input_grad[0]))
        ctx.r, ctx.qc, ctx.inbc, ctx.queue, ctx.self_sim = r, qc, inbc, queue, self_sim
        ctx.mark_non_differentiable(r.scalars, r.dists, r.softmax_weights, r.pos)
        return r.scalars[0].clone(), r.scalars, r.dists, r.softmax_weights, r.pos

    @staticmethod
    def backward(ctx, g_loss, *unused):
        r = ctx.r
        if g_loss is None:
            return None, None, None, None, None, None, None
        dq = torch.zeros_like(ctx.qc)
        gs = g_loss.detach().reshape(1).contiguous().float()
        wmat = ops.infonce_bwd(r, ctx.qc, ctx.inbc, ctx.queue, gs, dq, want_wmat=ctx.self_sim)
        if ctx.self_sim:   # column side of q.q^T: dq_j += sum_i w_ij q_i
            B, D = ctx.qc.shape
            ops.conv_wgrad(ops.linear_desc(B, D, B), ctx.qc, wmat, dq)
        return dq, None, None, None, None, None, None


class _EncodeFn(torch.autograd.Function):
    """trunk -> avg-pool -> head -> L2 normalise as one autograd node.  ``anchor`` is a dummy requires-grad scalar that
    makes autograd call backward; parameter gradients are accumulated into the model's flat gradient buffer."""

    @staticmethod
    def forward(ctx, anchor, data, model, jigsaw, orders, with_head):
        ctx.set_materialize_grads(False)
        out = model._encode(data, jigsaw, orders, with_head, save=True)
        ctx.model, ctx.generation = model, model._fwd_generation
        ctx.mark_non_differentiable(out[0])
        return out

    @staticmethod
    def backward(ctx, d_spatial, d_pooled, d_pre, d_emb):
        model = ctx.model
        if ctx.generation != model._fwd_generation:
            raise RuntimeError("VinceModel: backward of a stale forward -- the saved activations were overwritten by a "
                               "later grad-enabled forward of the same model")
        model._encode_backward(d_pooled, d_pre, d_emb)
        return torch.zeros_like(model._anchor), None, None, None, None, None


class VinceModel(BaseModel):
    def __init__(self, args):
        super().__init__(args)
        self.args, self.num_frames = args, args.num_frames
        self.compute_dtype = _compute_dtype(args)
        self.conv_x3 = str(getattr(args, "compute_dtype", "fp32")).lower() in X3_NAMES + X3F_NAMES
        self.conv_x3f = str(getattr(args, "compute_dtype", "fp32")).lower() in X3F_NAMES

        # Modules (vince_model.py:25-49).  Attribute names and nesting ARE the state-dict layout of the reference's checkpoints:
        # feature_extractor.*, embedding.{0,2}.*, jigsaw_linear.*, jigsaw_embedding.{0,2}.*, imagenet_decoders.{0,1.0,1.2}.*
        trunk = self.feature_extractor = args.backbone(args, -2)       # everything up to (not including) avgpool / fc
        trunk.bind_owner(self)
        width = self.output_channels = trunk.output_channels

        def two_layer_head(n_in, n_out):
            return nn.Sequential(nn.Linear(n_in, width), constants.NONLINEARITY(), nn.Linear(width, n_out))

        if getattr(self.args, "use_attention", False):
            raise NotImplementedError("--use-attention (dg_util AttentionPool2D) is not part of the HIP path")
        self.embedding = two_layer_head(width, args.vince_embedding_size)
        if args.jigsaw:                                 # nine tile features -> one embedding (vince_model.py:44-49)
            self.jigsaw_linear = nn.Linear(width, width)
            self.jigsaw_embedding = two_layer_head(9 * width, args.vince_embedding_size)
        if getattr(args, "use_imagenet", False):        # vince_model.py:79-90: plain torch side heads on detached features
            self.imagenet_decoders = nn.ModuleList([nn.Linear(width, 1000), two_layer_head(width, 1000)])
            self.num_imagenet_decoders = 2

        self._anchor = torch.zeros((), requires_grad=True)
        self._trunks = {}
        self._saved_ws = None
        self._wcache = None
        # compute_dtype "x3f", train mode: the backward runs on a bf16 TWIN engine over bf16 copies the x3 forward leaves in the twin's
        # workspace (engine.Trunk.set_shadow); VINCE_X3F_HYBRID=0: the fp32-tensor backward with single bfloat16 products instead
        self._twins, self._saved_ws_bf, self._wcache_bf, self._wcache_bf_version = {}, None, None, 0
        self.x3f_hybrid = self.conv_x3f and os.environ.get("VINCE_X3F_HYBRID", "1") != "0"
        self._saved = None
        self._fwd_generation = 0
        self._param_version, self._wcache_version = 1, 0
        self._bn_version, self._fold_version, self._wcache_folded = 0, None, None
        self._grad_zero_pending = True
        self._flat = self._flat_grad = None
        # Deferred stem join (engine.Trunk.set_stem_event; opt-in, VinceSolver sets it for single-process runs): backward returns
        # before conv1's weight gradient -- the last launch of the step -- has landed; FlatSGD.step(defer_stem=True) and
        # VinceQueueModel.param_update step / average every OTHER parameter beside it and conv1.weight behind the event.
        self.defer_stem_join = False
        self._stem_event, self._stem_pending, self._deferred_step = None, False, None
        # data parallel: (flat offset below which gradients arrive late, waiter) set by the reducer -- the last gradient bucket
        # (stem + layer1) instead of conv1.weight alone; _deferred_split: where FlatSGD actually split the step
        self._late, self._deferred_split = None, None
        self._build_flat()

    # ------------------------------------------------------------------------------------------ flat storage
    def _flat_entries(self):
        """(parameter, is_conv) in flat order: trunk (engine order), heads, then the EMA-only tail (fc)."""
        res = self.feature_extractor.model
        entries = [(p, p.dim() == 4) for p in res.trunk_params]
        heads = [self.embedding[0].weight, self.embedding[0].bias, self.embedding[2].weight, self.embedding[2].bias]
        if self.args.jigsaw:
            heads += [self.jigsaw_linear.weight, self.jigsaw_linear.bias, self.jigsaw_embedding[0].weight,
                      self.jigsaw_embedding[0].bias, self.jigsaw_embedding[2].weight, self.jigsaw_embedding[2].bias]
        entries += [(p, False) for p in heads]
        tail = [(res.fc.weight, False), (res.fc.bias, False)]
        return entries, tail

    def _build_flat(self):
        entries, tail = self._flat_entries()
        device = entries[0][0].device
        offs, total = [], 0
        for p, _ in entries + tail:
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        n_train = offs[len(entries)] if tail else total
        flat = torch.zeros(total, dtype=torch.float32, device=device)
        grad = torch.zeros(n_train, dtype=torch.float32, device=device)
        with torch.no_grad():
            for i, (p, is_conv) in enumerate(entries + tail):
                n, off = p.numel(), offs[i]

                def view_of(buf):
                    if is_conv:
                        co, ci, kh, kw = p.shape
                        return buf[off:off + n].view(co, kh, kw, ci).permute(0, 3, 1, 2)
                    return buf[off:off + n].view(p.shape)

                v = view_of(flat)
                v.copy_(p.data)
                p.data = v
                p.grad = view_of(grad) if (i < len(entries) and p.requires_grad) else None
        self._flat, self._flat_grad = flat, grad
        self._n_train, self._n_ema = n_train, total
        self._offs = offs
        self._anchor = torch.zeros((), requires_grad=True, device=device)
        res = self.feature_extractor.model
        ntrunk = len(res.trunk_params)
        self._head_params = [p for p, _ in entries[ntrunk:]]
        # contiguous flat ranges by role: SGD skips a head whose parameters received no gradient this step, exactly as
        # torch.optim.SGD skips parameters whose .grad is None (the standard head idles when the query side is jigsawed)
        emb_end = offs[ntrunk + 4] if len(entries) > ntrunk + 4 else n_train
        self._segments = {"trunk": (0, offs[ntrunk]), "embedding": (offs[ntrunk], emb_end), "jigsaw": (emb_end, n_train)}
        self._touched = {"trunk": False, "embedding": False, "jigsaw": False}
        self._bucket_events = None
        self._bucket_hook = None     # callable(e): data-parallel reducer's per-bucket launch (vince_amd/dp.py)
        # first trunk-parameter offset of every ResNet stage (gradient buckets for data parallelism)
        self._stage_offsets = {}
        for i, (name, _, _, _) in enumerate(res.plan.params):
            stage = name.split(".")[0]
            if stage.startswith("layer") and stage not in self._stage_offsets:
                self._stage_offsets[stage] = offs[i]
        if device.type == "cuda":
            self._param_ptrs = pointer_table([p.data for p in res.trunk_params])
            self._grad_ptrs = pointer_table([flat_grad_view(grad, offs[i], res.trunk_params[i]) for i in range(ntrunk)])
            self._head_grads = [flat_grad_view(grad, offs[ntrunk + j], hp) for j, hp in enumerate(self._head_params)]
            run, nbt = [], []
            for node in res.bn_nodes():
                run += [node.running_mean, node.running_var]
                nbt.append(node.num_batches_tracked)
            self._bn_running_ptrs, self._bn_nbt_ptrs = pointer_table(run), pointer_table(nbt)
        self._trunks, self._saved_ws, self._wcache, self._saved = {}, None, None, None
        self._twins, self._saved_ws_bf, self._wcache_bf, self._wcache_bf_version = {}, None, None, 0
        self._touch()

    def _apply(self, fn, *a, **k):
        super(VinceModel, self)._apply(fn, *a, **k)
        if getattr(self, "_flat", None) is not None:
            self._build_flat()
        return self

    def to(self, device):
        super().to(device)   # the reference's VinceModel.to returns None (vince_model.py:92-94); we return self
        return self

    def load_state_dict(self, state_dict, strict=True, **kw):
        # accept DataParallel-era keys (feature_extractor.module.model.*)
        fixed = {k.replace("feature_extractor.module.", "feature_extractor."): v for k, v in state_dict.items()}
        out = super(VinceModel, self).load_state_dict(fixed, strict=strict, **kw)
        self._touch()
        return out

    def _touch(self):
        """Parameters changed: the compute-dtype weight copies must be rebuilt before the next forward."""
        self._param_version += 1

    def zero_grad(self, set_to_none=True):
        self.finish_deferred_step()
        self._grad_zero_pending = True
        self._touched = {"trunk": False, "embedding": False, "jigsaw": False}
        if hasattr(self, "imagenet_decoders"):
            # torch 1.4 (the reference's pin) zeroes in place: a decoder that has received a gradient once keeps a zero .grad and is
            # stepped (weight decay + momentum) on every later iteration, also on batches without labelled data -- the same rule
            # FlatSGD applies to the heads inside the flat buffer (optim.py `_ever_touched`)
            for p in self.imagenet_decoders.parameters():
                if p.grad is not None:
                    p.grad.detach_()
                    p.grad.zero_()

    def flat_parameters(self):
        """(flat params, flat grads, trainable length, ema length) for the fused optimiser / EMA / all-reduce."""
        return self._flat, self._flat_grad, self._n_train, self._n_ema

    def vince_parameters(self):
        # vince_model.py:96-104 (average_layers has no parameters without attention)
        params = list(self.feature_extractor.parameters()) + list(self.embedding.parameters())
        if self.args.jigsaw:
            params += list(self.jigsaw_linear.parameters()) + list(self.jigsaw_embedding.parameters())
        return params

    # ------------------------------------------------------------------------------------------ engine plumbing
    def _require_gpu(self):
        if self._flat.device.type != "cuda":
            raise RuntimeError("VinceModel: parameters are on %s -- the HIP path needs the GPU (model.to('cuda:0')); "
                               "there is no CPU fallback" % self._flat.device)

    def _trunk(self, n, h, w):
        key = (n, h, w)
        t = self._trunks.get(key)
        if t is None:
            t = Trunk(self.feature_extractor.arch, n, h, w, self.compute_dtype, x3=("f" if self.conv_x3f else self.conv_x3))
            self._trunks[key] = t
        return t

    def _ensure_weights(self, trunk):
        self.finish_deferred_step()
        if self._wcache is None:
            self._wcache = torch.empty(trunk.wc_bytes, dtype=torch.uint8, device=self._flat.device)
        if self._wcache_version != self._param_version:
            trunk.prepare_weights(self._param_ptrs, self._wcache)
            self._head_t = _TransposedHeads(self._head_params)
            self._head_c = _HeadCopies(self._head_params)
            self._wcache_version = self._param_version

    def prepare_weights_early(self, part):
        """The compute-dtype weight copies rebuilt AHEAD of the next forward (engine.Trunk.prepare_weights parts 1 / 2), for the callers
        that step conv1.weight behind the stem event: part 1 right after every other parameter has been stepped -- the launch then runs
        beside the stem's weight gradient instead of at the head of the next forward -- part 2 once conv1.weight is final.
        `weights_current()` declares the cache up to date afterwards.  False: nothing to rebuild yet (no forward has run)."""
        if self._wcache is None or not self._trunks or os.environ.get("VINCE_EARLY_PREP", "1") == "0":   # (=0: A/B measurements)
            return False
        next(iter(self._trunks.values())).prepare_weights(self._param_ptrs, self._wcache, part=part)
        if self._twins and self._wcache_bf is not None:      # x3f: the bf16 twin's cache (what its backward multiplies with) in the same two parts
            next(iter(self._twins.values())).prepare_weights(self._param_ptrs, self._wcache_bf, part=part)
        return True

    def weights_current(self):
        self._head_t = _TransposedHeads(self._head_params)
        self._head_c = _HeadCopies(self._head_params)
        self._wcache_version = self._param_version
        if self._twins and self._wcache_bf is not None:
            self._wcache_bf_version = self._param_version

    def _ensure_folded_weights(self, trunk):
        """Inference cache: BatchNorms folded into the conv weights (eval mode only).  Rebuilt when parameters change
        (_touch) or a train-mode forward has moved the running statistics."""
        if self._wcache_folded is None:
            self._wcache_folded = torch.empty(trunk.wc_bytes, dtype=torch.uint8, device=self._flat.device)
        if self._fold_version != (self._param_version, self._bn_version):
            if self.conv_x3:
                self._check_folded_half_range()
            trunk.prepare_weights_folded(self._param_ptrs, self._bn_running_ptrs, self._wcache_folded)
            self._fold_version = (self._param_version, self._bn_version)

    def _check_folded_half_range(self):
        """x3 inference cache: a folded weight w * gamma / sqrt(running_var + eps) is split into IEEE-half halves after a scale of 2^8
        (csrc/common.h X3_WSHIFT); a near-zero running variance can push it past the half range (|w'| >= 255.9), where the split turns
        into inf / NaN by design.  Say WHICH layer before that happens (one host read per rebuild of the inference cache)."""
        res = self.feature_extractor.model
        bns, params, worst, names = res.bn_nodes(), res.plan.params, [], []
        for i, (name, kind, _, _) in enumerate(params):
            if kind != 0 or i + 1 >= len(params) or params[i + 1][1] != 1:
                continue
            node = bns[params[i + 1][3]]
            scale = (res.trunk_params[i + 1].detach() / torch.sqrt(node.running_var + 1e-5)).abs()
            worst.append((res.trunk_params[i].detach().abs().flatten(1).amax(1) * scale).max())
            names.append(name)
        vals = torch.stack(worst).tolist()
        bad = [(n, v) for n, v in zip(names, vals) if not v * 256.0 < 65504.0]
        if bad:
            raise RuntimeError("VinceModel (compute_dtype x3): folded inference weights leave the IEEE-half range of the split-half "
                               "products (|w * gamma / sqrt(var + eps)| must stay below 255.9): %s -- run extract_features with an "
                               "fp32 or bf16 trunk, or VINCE_FOLD_BN=0" % ", ".join("%s %.1f" % b for b in bad[:4]))

    def _encode(self, data, jigsaw, orders, with_head, save):
        """Returns (spatial, pooled, prenorm, embeddings).  data: float32 NCHW on the GPU."""
        self._require_gpu()
        u8 = data if isinstance(data, U8Frames) else None
        if u8 is not None and jigsaw:
            # the 3 x 3 tiling kernel reads float NCHW: materialise the tensor the handle stands for (same layout kernels)
            data, u8 = u8.to(self._flat.device).float_tensor(), None
        if u8 is not None:
            u8 = u8.to(self._flat.device)
            n, (h, w) = u8.frames.shape[0], u8.size
        else:
            if data.dtype != torch.float32 or data.dim() != 4 or data.shape[1] != 3:
                raise ValueError("VinceModel: expected float32 N x 3 x H x W frames, got %s %s" % (data.dtype, tuple(data.shape)))
            data = data.to(self._flat.device).contiguous()
            n, _, h, w = data.shape
        if jigsaw:
            hp, wp = (h + 3 - h % 3, w + 3 - w % 3) if (h % 3 or w % 3) else (h, w)   # vince_model.py:145-146
            trunk = self._trunk(n * 9, hp // 3, wp // 3)
        else:
            trunk = self._trunk(n, h, w)
        self._ensure_weights(trunk)
        twin = None
        if save:
            if self._saved_ws is None or self._saved_ws.numel() < trunk.ws_bytes:
                self._saved_ws = None
                self._saved_ws = torch.empty(trunk.ws_bytes, dtype=torch.uint8, device=self._flat.device)
            ws = self._saved_ws
            self._fwd_generation += 1
            if self.x3f_hybrid and self.training:
                # x3f: this forward also writes bf16 copies of what backward reads into the workspace of a bf16 twin engine
                key = (trunk.N, trunk.H, trunk.W)
                twin = self._twins.get(key)
                if twin is None:
                    twin = self._twins[key] = Trunk(self.feature_extractor.arch, trunk.N, trunk.H, trunk.W, torch.bfloat16)
                if self._saved_ws_bf is None or self._saved_ws_bf.numel() < twin.ws_bytes:
                    self._saved_ws_bf = None
                    self._saved_ws_bf = torch.empty(twin.ws_bytes, dtype=torch.uint8, device=self._flat.device)
                if self._wcache_bf is None:
                    self._wcache_bf = torch.empty(twin.wc_bytes, dtype=torch.uint8, device=self._flat.device)
                if self._wcache_bf_version != self._param_version:     # the bf16 weight copies its input / weight gradients multiply with
                    twin.prepare_weights(self._param_ptrs, self._wcache_bf)
                    self._wcache_bf_version = self._param_version
                trunk.set_shadow(twin, self._saved_ws_bf)
        else:
            ws = nograd_workspace(self._flat.device, trunk.ws_bytes)
        nt = trunk.N
        pooled = torch.empty(nt, self.output_channels, device=self._flat.device, dtype=torch.float32)
        if u8 is not None:   # GPU input stage: the stem layout is written straight from the uint8 frames
            if u8.blur is not None:
                trunk.stage_blur(ws, u8.frames.contiguous(), u8.flip, u8.blur[0], u8.blur[1], constants.IMAGENET_MEAN,
                                 constants.IMAGENET_STD)
            else:
                trunk.stage_u8(ws, u8.frames.contiguous(), u8.crop_yx, u8.flip, None, constants.IMAGENET_MEAN,
                               constants.IMAGENET_STD)
            data = None
        if not self.training and not save and FOLD_BN:
            # inference (extract_features for the end tasks, validation): BatchNorms folded into the convolutions
            self._ensure_folded_weights(trunk)
            trunk.forward_folded(self._wcache_folded, data, ws, pooled, jigsaw_src=(h, w) if jigsaw else None)
        else:
            try:
                trunk.forward(self._param_ptrs, self._wcache, self._bn_running_ptrs, self._bn_nbt_ptrs, data, ws, pooled,
                              train_bn=self.training, jigsaw_src=(h, w) if jigsaw else None, save=bool(save))
            finally:
                if twin is not None:
                    trunk.set_shadow(None, None)
            if self.training:
                self._bn_version += 1   # running statistics moved
        # `spatial_features` is a copy by default: the workspace it lives in is rewritten by the next forward.  A caller that
        # never reads it past that point (the training loop) sets clone_spatial = False and gets the zero-copy view.
        spatial = trunk.spatial_view(ws)
        if getattr(self, "clone_spatial", True):
            # (the view is channels-last over one dense block of the workspace: copied with the library's streaming copy -- torch's
            # clone() of it becomes the runtime's blit, 25 pieces at 0.8 TB/s = 0.13 ms per 256 frames of ResNet-50)
            sn, sc, sh_, sw = spatial.shape
            dense = spatial.permute(0, 2, 3, 1)
            copy = torch.empty((sn, sh_, sw, sc), dtype=spatial.dtype, device=spatial.device)
            if dense.is_contiguous() and (copy.numel() * copy.element_size()) % 16 == 0:
                spatial = ops.stream_copy(copy, dense).permute(0, 3, 1, 2)
            else:
                spatial = spatial.clone()
        pre = emb = None
        # (detached aliases: `pooled` and `pre` are also RETURNED through _EncodeFn, whose autograd node holds the model -- saving the
        # returned objects themselves would close a cycle model -> _saved -> tensor -> grad_fn -> ctx.model that Python's collector
        # cannot see through, and every discarded model would keep its workspaces: tens of GB per solver at the benchmark size)
        saved = dict(trunk=trunk, twin=twin, pooled=pooled.detach(), jigsaw=jigsaw, train_bn=bool(self.training))
        if with_head:
            hx3 = head_x3() and self.compute_dtype == torch.bfloat16

            def lin(x, layer, relu=False):   # one head Linear: split-half products on its prepared copy, or exact fp32 MFMAs
                if hx3 and _head_x3_shape(layer.weight):
                    return ops.linear_fwd(x, self._head_c.copies(id(layer.weight), bool(save))[0], layer.bias.data, relu=relu, x3="b")
                return ops.linear_fwd(x, layer.weight.data, layer.bias.data, relu=relu)
            if jigsaw:   # vince_model.py:161-171
                f = lin(pooled, self.jigsaw_linear)
                c = f.shape[1]
                idx = orders.to(f.device).unsqueeze(-1).expand(n, 9, c)
                g = torch.gather(f.view(n, 9, c), 1, idx).reshape(n, 9 * c).contiguous()
                hid = lin(g, self.jigsaw_embedding[0], relu=True)
                pre = lin(hid, self.jigsaw_embedding[2])
                saved.update(g=g, hid=hid, orders=orders.to(f.device))
            else:        # vince_model.py:175-177
                hid = lin(pooled, self.embedding[0], relu=True)
                pre = lin(hid, self.embedding[2])
                saved.update(hid=hid)
            emb, norms = ops.l2norm_fwd(pre)   # F.normalize(dim=1), vince_model.py:180
            saved.update(pre=pre.detach(), norms=norms)
        if save:
            self._saved = saved
        return spatial, pooled, pre, emb

    def _encode_backward(self, d_pooled, d_pre, d_emb):
        s = self._saved
        if self._grad_zero_pending:
            self._flat_grad.zero_()
            self._grad_zero_pending = False
        hg = {id(p): g for p, g in zip(self._head_params, self._head_grads)}
        ht = self._head_t
        dpool_total = d_pooled.contiguous().float() if d_pooled is not None else None
        if "pre" in s and (d_pre is not None or d_emb is not None):
            dpre = None
            if d_emb is not None:
                dpre = ops.l2norm_bwd(s["pre"], s["norms"], d_emb.contiguous().float())
            if d_pre is not None:
                dpre = d_pre.contiguous().float() if dpre is None else dpre + d_pre
            self._touched["jigsaw" if s["jigsaw"] else "embedding"] = True
            hx3 = head_x3() and self.compute_dtype == torch.bfloat16

            def lbwd(x, layer, dy):          # gradients of one head Linear, on the route its forward took
                w = layer.weight
                if hx3 and _head_x3_shape(w):
                    return ops.linear_bwd(x, self._head_c.copies(id(w), True)[1], dy, hg[id(w)], hg[id(layer.bias)], x3=True)
                return ops.linear_bwd(x, ht[id(w)], dy, hg[id(w)], hg[id(layer.bias)])
            if s["jigsaw"]:
                l0, l2, lj = self.jigsaw_embedding[0], self.jigsaw_embedding[2], self.jigsaw_linear
                dh = lbwd(s["hid"], l2, dpre)
                dh = ops.relu_bwd(dh, s["hid"])
                dg = lbwd(s["g"], l0, dh)
                n, c = dg.shape[0], dg.shape[1] // 9
                df = torch.zeros(n, 9, c, device=dg.device)
                df.scatter_(1, s["orders"].unsqueeze(-1).expand(n, 9, c), dg.view(n, 9, c))
                df = df.view(n * 9, c).contiguous()
                dp = lbwd(s["pooled"], lj, df)
            else:
                l0, l2 = self.embedding[0], self.embedding[2]
                dh = lbwd(s["hid"], l2, dpre)
                dh = ops.relu_bwd(dh, s["hid"])
                dp = lbwd(s["pooled"], l0, dh)
            dpool_total = dp if dpool_total is None else dpool_total + dp
        if dpool_total is None:
            return
        if not s.get("train_bn", True):
            # The engine's BatchNorm backward is the TRAIN-mode one (batch statistics: dy = s (g - mean g - xhat mean(g xhat))); through an
            # eval-mode forward (running statistics) autograd's is dy = s g.  Found in round 6: the train-mode formula on unnormalised
            # inputs overflows within a few layers.  Outside the hot path (the reference trains in train mode, solvers/vince_solver.py:386):
            # refused, not approximated.
            raise RuntimeError("VinceModel: backward through an eval-mode trunk forward is not supported (BatchNorm backward with running "
                               "statistics); call model.train() before the forward, or detach the features")
        self._touched["trunk"] = True
        # data parallel: the reducer's hook runs inside the engine call, right after each bucket's event is recorded
        # x3f: the bf16 twin runs the backward on the bf16 copies the forward left in ITS workspace, with its own bf16 weight cache
        bt = s["twin"] if s.get("twin") is not None else s["trunk"]
        bwc, bws = (self._wcache_bf, self._saved_ws_bf) if s.get("twin") is not None else (self._wcache, self._saved_ws)
        bt.set_bucket_callback(self._bucket_hook if self._bucket_events else None)
        defer = self.defer_stem_join      # (with gradient buckets too: dp.GradientReducer.reduce_after_backward holds the last bucket back)
        if defer and self._stem_event is None:
            self._stem_event = torch.cuda.Event()
            self._stem_event.record()            # (torch creates the hipEvent lazily; the engine needs a live handle)
        bt.set_stem_event(self._stem_event if defer else None)
        bt.backward(self._param_ptrs, bwc, bws, dpool_total.contiguous(), self._grad_ptrs, bucket_events=self._bucket_events)
        self._stem_pending = defer

    def finish_stem_grad(self):
        """Makes the current stream wait for conv1's weight gradient of the last backward (no-op unless defer_stem_join is on)."""
        if self._stem_pending:
            torch.cuda.current_stream().wait_event(self._stem_event)
            self._stem_pending = False

    def finish_deferred_step(self):
        """Runs what FlatSGD.step(defer_stem=True) left for later: the optimiser step of conv1.weight behind the stem event."""
        fn, self._deferred_step = self._deferred_step, None
        if fn is not None:
            fn()
        self.finish_stem_grad()

    def _run_encoder(self, data, jigsaw=False, orders=None, with_head=True):
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.feature_extractor.model.trunk_params)
        if needs_grad:
            return _EncodeFn.apply(self._anchor, data, self, jigsaw, orders, with_head)
        return self._encode(data, jigsaw, orders, with_head, save=False)

    # ------------------------------------------------------------------------------------------ reference API
    @staticmethod
    def split_dict_by_type(batch_types, batch_sizes, dict_to_split):
        # vince_model.py:106-121
        # One dict per batch type.  An entry with one element per type (lists like data_source, num_frames -- the reference tells
        # them apart by LENGTH, a quirk kept: a tensor whose first dimension happens to equal the number of types is indexed too)
        # is indexed by type; anything else is a row-wise concatenation and is sliced by the running row offset.
        assert "queue_vectors" not in dict_to_split
        n_types = len(batch_types)
        starts = [sum(batch_sizes[:i]) for i in range(n_types)]
        pieces = []
        for i, (kind, rows, lo) in enumerate(zip(batch_types, batch_sizes, starts)):
            piece = {name: (v[i] if len(v) == n_types else v[lo: lo + rows]) for name, v in dict_to_split.items()}
            piece.pop("batch_types", None)
            piece["batch_type"] = kind
            pieces.append(piece)
        return pieces

    def extract_features(self, inputs, run_average_layer=True):
        # vince_model.py:123-133
        spatial, pooled, _, _ = self._run_encoder(inputs, with_head=False)
        return_val = {"spatial_features": spatial}
        if run_average_layer:
            return_val["extracted_features"] = pooled
        return return_val

    def get_embeddings(self, inputs, jigsaw=False, shuffle=False):
        # vince_model.py:135-196
        data = inputs["data"]
        if shuffle:
            # The reference permutes the batch before the DataParallel scatter and un-permutes every output
            # (vince_model.py:137-142,184-192).  On one device that is the identity on every returned tensor, so only
            # the RNG draw is kept; across ranks the key shuffle is done by the data-parallel glue (vince_amd/dp.py).
            torch.randperm(data.shape[0], device=data.device)
        orders = None
        if jigsaw:
            orders = inputs.get("jigsaw_orders")
            if orders is None:   # vince_model.py:166: an independent random tile order per sample
                orders = torch.rand(data.shape[0], 9, device=data.device).argsort(dim=1)
        spatial, pooled, pre, emb = self._run_encoder(data, jigsaw, orders)
        # vince_model.py:171-172 overwrites extracted_features with the nine-tile feature in jigsaw mode
        feats = dict(spatial_features=spatial, extracted_features=pre if jigsaw else pooled, prenorm_features=pre, embeddings=emb)
        if "batch_types" not in inputs:
            return feats
        return self.split_dict_by_type(inputs["batch_types"], inputs["batch_sizes"], feats)

    def forward(self, inputs: Dict[str, torch.Tensor]):
        # vince_model.py:198-250
        result = dict(inputs)             # shallow: the caller's dict is left as it was
        from_imagenet = inputs["data_source"] == "IN"
        if from_imagenet:                 # side heads see the features, the trunk sees none of their gradient
            side_in = result["extracted_features"].detach().clone()

        output = result["embeddings"]
        if "queue_embeddings" in inputs and "vince_similarities" not in inputs:
            q_emb, q_vec = inputs["queue_embeddings"], inputs["queue_vectors"]
            B = output.shape[0]
            if self.args.inter_batch_comparison:
                if B != self.args.batch_size:
                    raise ValueError("inter-batch comparison needs full batches (%d != %d): the reference's mask slice "
                                     "would misalign the [B | K] blocks (vince_model.py:240)" % (B, self.args.batch_size))
                frames = inputs["num_frames"] if inputs["num_frames"] > 1 else 1
                if self.args.self_batch_comparison:
                    loss_s, scal_s, dists_s, sw_s, _ = _InfoNCEFn.apply(
                        output, None, None, self.args.vince_self_temperature, frames, True, True)
                    result.update(dict(
                        vince_self_similarities=LazySimilarities(output, output.detach(), None, False),
                        vince_self_similarities_mask=("block_diag", frames, B, 0),
                        _vince_nce_self=dict(loss=loss_s, scalars=scal_s, dists=dists_s, softmax_weights=sw_s),
                    ))
                loss, scal, dists, sw, pos = _InfoNCEFn.apply(output, q_emb, q_vec,
                                                              self.args.vince_temperature, frames, True, False)
                sims = LazySimilarities(output, q_emb.detach(), q_vec, False)
                mask = ("block_diag", frames, B, q_vec.shape[0])
                result["vince_l_neg"] = sims
            else:
                loss, scal, dists, sw, pos = _InfoNCEFn.apply(output, q_emb, q_vec,
                                                              self.args.vince_temperature, 1, False, False)
                sims = LazySimilarities(output, q_emb.detach(), q_vec, True)
                mask = ("first_column", 1, B, q_vec.shape[0])
                result["vince_l_pos"] = pos
                result["vince_l_neg"] = LazySimilarities(output, q_emb.detach(), q_vec, False, include_inb=False)
            result.update(dict(
                vince_similarities=sims,
                vince_similarities_mask=mask,
                _vince_nce=dict(loss=loss, scalars=scal, dists=dists, softmax_weights=sw),
            ))

        if from_imagenet:                 # vince_model.py:245-249: only the labelled rows (the first of each clip) are decoded
            side_in = side_in[: inputs["imagenet_labels"].shape[0]]
            result.update(("imagenet_decoder_%d" % i, head(side_in)) for i, head in enumerate(self.imagenet_decoders))
        return result

    def loss(self, network_outputs: Optional[Dict]) -> Dict[str, Optional[Tuple[float, torch.Tensor]]]:
        # vince_model.py:252-290
        if network_outputs is None:       # the solver asks for the NAMES first (meters): vince_model.py:253-261
            names = ["nce_loss"] + (["nce_loss_self"] if self.args.self_batch_comparison else [])
            names += ["imagenet_loss_%d" % i for i in range(getattr(self, "num_imagenet_decoders", 0))]
            return dict.fromkeys(names)

        losses = {}
        for key, lkey in (("", "nce_loss"), ("self_", "nce_loss_self")):
            fused = network_outputs.get("_vince_nce" if key == "" else "_vince_nce_self")
            if fused is not None:
                b = fused["dists"].shape[0]
                res = dict(dists=fused["dists"].view(b, 1, -1), dist=fused["loss"],
                           softmax_weights=fused["softmax_weights"].view(b, 1, -1), softmax_weight=fused["scalars"][1])
            elif ("vince_" + key + "similarities") in network_outputs and (key == "" or self.args.self_batch_comparison):
                # a caller handed us materialised similarities + a boolean mask: the row-wise kernel of loss_util
                sims = network_outputs["vince_" + key + "similarities"]
                mask = network_outputs["vince_" + key + "similarities_mask"]
                if isinstance(sims, LazySimilarities):
                    sims = sims.materialize()
                temperature = self.args.vince_temperature if key == "" else self.args.vince_self_temperature
                res = loss_util.similarity_cross_entropy(sims, temperature, sims.shape[0], 1, mask)
            else:
                continue
            network_outputs.update({"vince_loss_" + key + k: v for k, v in res.items()})
            losses[lkey] = (1.0, res["dist"])

        if network_outputs["data_source"] == "IN":
            for ii in range(self.num_imagenet_decoders):
                losses["imagenet_loss_%d" % ii] = (1.0, F.cross_entropy(network_outputs["imagenet_decoder_%d" % ii],
                                                                        network_outputs["imagenet_labels"]))
        return losses

    def get_metrics(self, network_outputs: Optional[Dict]) -> Dict[str, Optional[float]]:
        # vince_model.py:292-349
        with torch.no_grad():
            metrics = {}
            if network_outputs is None:   # names only (vince_model.py:294-311)
                names = ["nce_accuracy_mean", "nce_softmax_weight_mean", "cosine_sim", "cosine_sim_neg_max"]
                if self.args.self_batch_comparison:
                    names += ["nce_accuracy_self_mean", "nce_softmax_weight_self_mean", "cosine_self_sim"]
                names += ["imagenet_accuracy_%d" % i for i in range(getattr(self, "num_imagenet_decoders", 0))]
                return dict.fromkeys(names)

            for key in ["", "self_"]:
                fused = network_outputs.get("_vince_nce" if key == "" else "_vince_nce_self")
                if fused is None:
                    continue
                sc = fused["scalars"]
                metrics.update({
                    "nce_accuracy_" + key + "mean": sc[2],
                    "nce_softmax_weight_" + key + "mean": sc[1],
                    "cosine_" + key + "sim": sc[3],
                })
                if key == "":
                    metrics["cosine_sim_neg_max"] = sc[4]

            if network_outputs["data_source"] == "IN":
                for ii in range(self.num_imagenet_decoders):
                    predictions = torch.argmax(network_outputs["imagenet_decoder_%d" % ii], dim=1)
                    metrics["imagenet_accuracy_%d" % ii] = (predictions == network_outputs["imagenet_labels"]).float().mean()
            return metrics

    def get_image_output(self, network_outputs):
        # vince_model.py:351-570 draws TensorBoard sheets with dg_util.drawing / cv2: off the timed path, out of scope.
        return {}


def flat_grad_view(grad, off, p):
    n = p.numel()
    if p.dim() == 4:
        co, ci, kh, kw = p.shape
        return grad[off:off + n].view(co, kh, kw, ci)
    return grad[off:off + n].view(p.shape)


class VinceQueueModel(BaseModel):
    """Momentum (key) encoder, vince_model.py:573-613."""

    def __init__(self, args, encoder: VinceModel):
        super().__init__(args)
        # the reference deep-copies the encoder (:576); here a fresh model is built and the state copied, which is the
        # same thing without cloning engine handles / workspaces
        self.queue_network = VinceModel(args)
        self.queue_network.load_state_dict(encoder.state_dict())
        if encoder._flat.device.type == "cuda":
            self.queue_network.to(encoder._flat.device)
        self.vince_momentum = self.args.vince_momentum
        for param in self.queue_network.parameters():
            param.requires_grad = False
        self.queue_network._build_flat()   # drop the gradient views of the frozen copy

    def to(self, device):
        super().to(device)
        self._device = device
        return self

    def param_update(self, encoder_model: VinceModel, momentum: float):
        # vince_model.py:587-592: theta_k <- theta_k * m + (1 - m) * theta_q over vince_parameters() -- one pass over the
        # flat buffer (trunk + heads + the unused fc); BatchNorm buffers are NOT touched.
        with torch.no_grad():
            kflat, _, _, n_ema = self.queue_network.flat_parameters()
            qflat, _, _, _ = encoder_model.flat_parameters()
            if getattr(encoder_model, "_deferred_step", None) is not None:
                # the optimiser left conv1.weight (flat range [0, n1); under data parallelism the whole last gradient bucket) for behind
                # the stem event: average everything else now, beside the stem's weight gradient, then finish that step and average
                # the first range
                n1 = encoder_model._deferred_split or encoder_model._offs[1]
                ops.ema_flat(kflat[n1:n_ema], qflat[n1:n_ema], float(momentum))
                # (the early rebuild of the weight copies splits at conv1 only: single-process runs)
                early = n1 == encoder_model._offs[1] and self.queue_network.prepare_weights_early(1)
                encoder_model.finish_deferred_step()
                ops.ema_flat(kflat[:n1], qflat[:n1], float(momentum))
                self.queue_network._touch()
                if early and self.queue_network.prepare_weights_early(2):
                    self.queue_network.weights_current()
                return
            ops.ema_flat(kflat[:n_ema], qflat[:n_ema], float(momentum))
        self.queue_network._touch()

    def vince_update(self, encoder_model):
        self.param_update(encoder_model, self.vince_momentum)

    def forward(self, inputs, jigsaw=False, shuffle=True):
        # vince_model.py:597-613
        with torch.no_grad():
            queue_data = inputs["queue_data"]
            sub = {"data": queue_data, "batch_types": inputs["batch_types"], "batch_sizes": inputs["batch_sizes"]}
            if "queue_jigsaw_orders" in inputs:
                sub["jigsaw_orders"] = inputs["queue_jigsaw_orders"]
            output_mini_batches = self.queue_network.get_embeddings(sub, jigsaw=jigsaw, shuffle=shuffle)
            # every output of the key encoder re-keyed "queue_<name>", tensors cut from any graph
            return [{"queue_" + name: (v.detach() if isinstance(v, torch.Tensor) else v) for name, v in part.items()}
                    for part in output_mini_batches]
