"""GPU input stage: the reference's train / val image transforms (utils/transforms.py:62-235) as BATCH
operators on uint8 frames that already sit in HBM.

The reference builds one torchvision ``Compose`` per class and runs it per sample on PIL images inside 40 loader
processes.  Here each class of the same name is a *recipe* (which random draws, in which order) over three HIP launches
(csrc/augment.hip): the host draws the per-sample parameters with torchvision 0.5's distributions, the device does all pixel
work -- crop + Pillow-exact BILINEAR resize, the ColorJitter / RandomGrayscale chain, and flip + ToTensor/Normalize
[+ Gaussian blur] fused into the kernel that writes the stem's input layout.  The result is a ``U8Frames`` handle that
``VinceModel.get_embeddings`` accepts in place of the float ``data`` tensor, or (``as_tensor=True`` / single-image call, the
reference's per-sample contract) the float CHW tensor itself.
"""
import math
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch

from .. import constants

OP_NONE, OP_BRIGHTNESS, OP_CONTRAST, OP_SATURATION, OP_HUE, OP_GRAY = -1, 0, 1, 2, 3, 4
MAX_OPS = 5   # four jitter components + grayscale


@dataclass(frozen=True)
class Recipe:
    """What one reference transform class draws (utils/transforms.py line of its Compose in the comment of RECIPES)."""
    crop_scale: Tuple[float, float]
    crop_ratio: Tuple[float, float] = (3.0 / 4.0, 4.0 / 3.0)            # torchvision RandomResizedCrop default
    jitter: Optional[Tuple[float, float, float, float]] = None          # brightness, contrast, saturation, hue
    gray_p: float = 0.0
    gray_first: bool = False                                            # RandomGrayscale before ColorJitter (MoCo v2)
    flip_p: float = 0.5
    blur_p: float = 0.0                                                 # RandomApply([RandomGaussianBlur(size // 10)])


RECIPES = {
    "BasicImagenetTransform": Recipe((0.2, 1.0), (0.7, 1.4), (0.4, 0.4, 0.4, 0.2), 0.2),                     # :64-76
    "StandardVideoTransform": Recipe((0.2, 1.0), jitter=(0.4, 0.4, 0.4, 0.2), gray_p=0.2),                   # :91-103
    "SimCLRTransform": Recipe((0.2, 1.0), jitter=(0.8, 0.8, 0.8, 0.2), gray_p=0.2, blur_p=0.5),              # :106-119
    "JigsawTransform": Recipe((0.7, 1.0), jitter=(0.8, 0.8, 0.8, 0.2), gray_p=0.2, blur_p=0.5),              # :122-135
    "SunSceneTransform": Recipe((0.7, 1.0), jitter=(0.4, 0.4, 0.4, 0.2), gray_p=0.2),                        # :138-150
    "Kinetics400Transform": Recipe((0.5, 1.0), jitter=(0.4, 0.4, 0.4, 0.2), gray_p=0.2),                     # :153-165
    "GOT10KTransform": Recipe((0.2, 1.0)),                                                                   # :168-178
    "MoCoV1ImagenetTransform": Recipe((0.08, 1.0), jitter=(0.4, 0.4, 0.4, 0.2), gray_p=0.2),                 # :211-224
    "MoCoV2ImagenetTransform": Recipe((0.2, 1.0), jitter=(0.4, 0.4, 0.4, 0.4), gray_p=0.2, gray_first=True,
                                      blur_p=0.5),                                                           # :227-239
}


@dataclass
class AugmentParams:
    """Host-side draws for a batch of N outputs."""
    box: np.ndarray                       # int32 [N, 4]  top, left, height, width of the crop window
    op: np.ndarray                        # int32 [N, MAX_OPS]
    factor: np.ndarray                    # float32 [N, MAX_OPS]  (hue slot: the uint8 H-plane shift)
    flip: np.ndarray                      # uint8 [N]
    sigma: np.ndarray                     # float32 [N]  Gaussian-blur sigma, 0 = not blurred
    src_index: Optional[np.ndarray] = None   # int64 [N]  source frame of each output (None: identity)


def blur_kernel_size(size):
    ks = int(size) // 10                  # RandomGaussianBlur(self.size[0] // 10), utils/transforms.py:117
    return ks + 1 if ks % 2 == 0 else ks  # utils/util_functions.py:107-108


def blur_taps(sigma, ks):
    """utils/util_functions.py:110,115-117 for a vector of sigmas: float32 [N, ks] (rows with sigma == 0 stay zero).  One
    broadcast expression instead of a per-sample loop; equal to the per-sample evaluation up to the summation order of the
    normalising sum (1 ulp)."""
    sigma = np.asarray(sigma, np.float64)
    d2 = ((ks - 1) * 0.5 - np.arange(ks, dtype=np.float32)) ** 2                  # float32, like the reference's buffer
    on = sigma > 0
    coef = (-0.5 / np.where(on, sigma, 1.0) ** 2).astype(np.float32)              # the Python scalar of :115
    k = np.exp(coef[:, None] * d2[None, :])                                       # (numpy: torch.exp on a small CPU tensor
    k = k / np.maximum(k.sum(axis=1, keepdims=True, dtype=np.float32), np.float32(1e-10))   # costs ~60 ms of thread start-up)
    return torch.from_numpy(np.where(on[:, None], k, np.float32(0)).astype(np.float32))


def hue_shift_u8(hue_factor):
    """torchvision F.adjust_hue: np.uint8(hue_factor * 255) added to the H plane with wrap-around."""
    return float(int(hue_factor * 255) & 0xFF)


class BatchTransform:
    """Base of the recipe classes.  ``size``: (H, W) or int; ``data_subset``: "train" | "val" (utils/transforms.py:38-59)."""

    recipe: Recipe = None

    def __init__(self, size, data_subset="train", seed=None):
        self.size = (int(size), int(size)) if isinstance(size, (int, float)) else (int(size[0]), int(size[1]))
        self.data_subset = data_subset
        self.rng = np.random.default_rng(seed)
        if data_subset not in ("train", "val"):
            raise NotImplementedError("No transform for data_subset %s" % data_subset)

    # ------------------------------------------------------------------------------------------ draws (host)
    def _draw_box(self, hs, ws):
        """torchvision 0.5 RandomResizedCrop.get_params: ten tries for an area / log-uniform aspect draw that fits, then the
        central crop with the aspect clamped into range."""
        r, area = self.recipe, hs * ws
        lo, hi = math.log(r.crop_ratio[0]), math.log(r.crop_ratio[1])
        for _ in range(10):
            target = self.rng.uniform(*r.crop_scale) * area
            aspect = math.exp(self.rng.uniform(lo, hi))
            w = int(round(math.sqrt(target * aspect)))
            h = int(round(math.sqrt(target / aspect)))
            if 0 < w <= ws and 0 < h <= hs:
                return int(self.rng.integers(0, hs - h + 1)), int(self.rng.integers(0, ws - w + 1)), h, w
        return self._fallback_box(hs, ws)

    def _draw_color(self):
        r = self.recipe
        steps = []
        if r.jitter is not None:
            b, c, s, h = r.jitter
            jit = []
            if b > 0:
                jit.append((OP_BRIGHTNESS, self.rng.uniform(max(0.0, 1 - b), 1 + b)))
            if c > 0:
                jit.append((OP_CONTRAST, self.rng.uniform(max(0.0, 1 - c), 1 + c)))
            if s > 0:
                jit.append((OP_SATURATION, self.rng.uniform(max(0.0, 1 - s), 1 + s)))
            if h > 0:
                jit.append((OP_HUE, hue_shift_u8(self.rng.uniform(-h, h))))
            order = self.rng.permutation(len(jit))            # ColorJitter.get_params shuffles the component order
            steps = [jit[i] for i in order]
        if r.gray_p > 0 and self.rng.random() < r.gray_p:
            steps = [(OP_GRAY, 0.0)] + steps if r.gray_first else steps + [(OP_GRAY, 0.0)]
        return steps

    def draw(self, n, src_hw, src_index=None):
        """All draws of a batch at once (numpy, no per-sample Python): same distributions as _draw_box / _draw_color."""
        hs, ws = src_hw
        r, g = self.recipe, self.rng
        p = AugmentParams(np.zeros((n, 4), np.int32), np.full((n, MAX_OPS), OP_NONE, np.int32), np.zeros((n, MAX_OPS), np.float32),
                          np.zeros(n, np.uint8), np.zeros(n, np.float32),
                          None if src_index is None else np.asarray(src_index, np.int64))
        # ---- crop windows: ten candidate (area, aspect) draws per sample, the first that fits wins
        target = g.uniform(r.crop_scale[0], r.crop_scale[1], (n, 10)) * (hs * ws)
        aspect = np.exp(g.uniform(math.log(r.crop_ratio[0]), math.log(r.crop_ratio[1]), (n, 10)))
        w = np.rint(np.sqrt(target * aspect)).astype(np.int64)
        h = np.rint(np.sqrt(target / aspect)).astype(np.int64)
        ok = (w > 0) & (w <= ws) & (h > 0) & (h <= hs)
        first = np.argmax(ok, axis=1)
        rows = np.arange(n)
        bw, bh = w[rows, first], h[rows, first]
        top = np.floor(g.random(n) * (hs - bh + 1)).astype(np.int64)
        left = np.floor(g.random(n) * (ws - bw + 1)).astype(np.int64)
        none = ~ok.any(axis=1)
        if none.any():
            ft, fl, fh, fw = self._fallback_box(hs, ws)
            top[none], left[none], bh[none], bw[none] = ft, fl, fh, fw
        p.box[:] = np.stack([top, left, bh, bw], 1)
        # ---- colour chain
        jit = []
        if r.jitter is not None:
            for code, lim in zip((OP_BRIGHTNESS, OP_CONTRAST, OP_SATURATION), r.jitter[:3]):
                if lim > 0:
                    jit.append((code, g.uniform(max(0.0, 1 - lim), 1 + lim, n)))
            if r.jitter[3] > 0:
                hue = g.uniform(-r.jitter[3], r.jitter[3], n)
                jit.append((OP_HUE, ((hue * 255).astype(np.int64) & 0xFF).astype(np.float64)))   # trunc, then uint8 wrap
        gray = g.random(n) < r.gray_p if r.gray_p > 0 else np.zeros(n, bool)
        nj = len(jit)
        if nj:
            order = g.permuted(np.tile(np.arange(nj), (n, 1)), axis=1)            # ColorJitter shuffles its components
            codes = np.array([c for c, _ in jit], np.int32)[order]
            facs = np.stack([f for _, f in jit], 1)[rows[:, None], order]
            shift = (gray & r.gray_first).astype(np.int64)                        # grayscale first: jitter moves one slot right
            for j in range(nj):
                p.op[rows, j + shift] = codes[:, j]
                p.factor[rows, j + shift] = facs[:, j]
        gpos = np.where(r.gray_first, 0, nj)
        p.op[gray, gpos] = OP_GRAY
        p.flip[:] = g.random(n) < r.flip_p
        if r.blur_p > 0:
            blur = g.random(n) < r.blur_p
            p.sigma[:] = np.where(blur, g.random(n) * (2.0 - 0.1) + 0.1, 0.0)      # utils/util_functions.py:105,114
        return p

    def _fallback_box(self, hs, ws):
        """RandomResizedCrop's fallback: the central crop with the aspect clamped into the allowed range."""
        r = self.recipe
        in_ratio = ws / hs
        if in_ratio < r.crop_ratio[0]:
            w, h = ws, int(round(ws / r.crop_ratio[0]))
        elif in_ratio > r.crop_ratio[1]:
            h, w = hs, int(round(hs * r.crop_ratio[1]))
        else:
            w, h = ws, hs
        return (hs - h) // 2, (ws - w) // 2, h, w

    # ------------------------------------------------------------------------------------------ device work
    def apply(self, frames, params):
        """frames: uint8 [Ns, Hs, Ws, 3] on the GPU; params: AugmentParams -> U8Frames (uint8 [N, H, W, 3] + flip + blur).
        All per-sample parameters cross PCIe as ONE pinned, asynchronous copy (a pageable copy would make the host wait for
        the stream, i.e. for the previous training step)."""
        from .. import ops
        from ..models.vince_model import U8Frames
        dev = frames.device
        n = params.box.shape[0]
        has_color = bool((params.op >= 0).any())
        has_flip = bool(params.flip.any())
        has_blur = bool((params.sigma > 0).any())
        ks = blur_kernel_size(self.size[0])
        parts = [("src", (params.src_index if params.src_index is not None else np.arange(n)).astype(np.int64)),
                 ("box", params.box.astype(np.int32)), ("op", params.op.astype(np.int32)),
                 ("factor", params.factor.astype(np.float32)),
                 ("taps", blur_taps(params.sigma, ks).numpy() if has_blur else np.zeros((0, ks), np.float32)),
                 ("flip", params.flip.astype(np.uint8)), ("do_blur", (params.sigma > 0).astype(np.uint8))]
        sizes = [a.nbytes for _, a in parts]
        offs = np.concatenate([[0], np.cumsum([(b + 7) // 8 * 8 for b in sizes])])
        host = self._staging(int(offs[-1]))
        hview = host.numpy()
        for (name, a), o, b in zip(parts, offs, sizes):
            hview[o:o + b] = a.reshape(-1).view(np.uint8)
        blob = host.to(dev, non_blocking=True)
        self._ring[self._ring_pos][1].record()          # the staging buffer is free again once this copy has run
        d = {}
        for (name, a), o, b in zip(parts, offs, sizes):
            d[name] = blob[int(o):int(o) + b].view(torch.from_numpy(a[:0]).dtype).view(a.shape)
        img = ops.aug_resized_crop_u8(frames, d["box"], self.size, None if params.src_index is None else d["src"])
        if has_color:
            ops.aug_color_u8(img, d["op"], d["factor"])
        return U8Frames(img, self.size, None, d["flip"] if has_flip else None,
                        blur=(d["taps"], d["do_blur"]) if has_blur else None)

    def _staging(self, nbytes):
        """A pinned host buffer from a small ring (allocated once: pinning memory synchronises the device), guarded by the
        event of the copy that last read it."""
        if getattr(self, "_ring", None) is None or self._ring[0][0].numel() < nbytes:
            self._ring = [(torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8).pin_memory(), torch.cuda.Event())
                          for _ in range(4)]
            self._ring_pos = -1
            self._ring_used = [False] * 4
        self._ring_pos = (self._ring_pos + 1) % 4
        buf, ev = self._ring[self._ring_pos]
        if self._ring_used[self._ring_pos]:
            ev.synchronize()
        self._ring_used[self._ring_pos] = True
        return buf[:nbytes]

    def apply_val(self, frames):
        """Resize((H / 0.875, W / 0.875), BILINEAR) + CenterCrop(size) (utils/transforms.py:78-88); the crop is taken by the
        layout kernel."""
        from .. import ops
        from ..models.vince_model import U8Frames
        n, hs, ws, _ = frames.shape
        rh, rw = int(self.size[0] / 0.875), int(self.size[1] / 0.875)
        box = torch.tensor([[0, 0, hs, ws]] * n, dtype=torch.int32, device=frames.device)
        img = ops.aug_resized_crop_u8(frames, box, (rh, rw))
        # torchvision center_crop: int(round((h - th) / 2.))
        cy, cx = int(round((rh - self.size[0]) / 2.0)), int(round((rw - self.size[1]) / 2.0))
        crop = torch.tensor([[cy, cx]] * n, dtype=torch.int32, device=frames.device)
        return U8Frames(img, self.size, crop, None)

    def __call__(self, frames, repeats=1, as_tensor=False):
        """Batch call: uint8 [N, Hs, Ws, 3] GPU tensor -> U8Frames of N * repeats views (view v of frame i at v * N + i).
        Per-sample call (the reference contract, utils/transforms.py:52-59): one HWC uint8 image (numpy or tensor) -> the
        float CHW tensor."""
        single = getattr(frames, "ndim", 0) == 3
        if single:
            frames = torch.as_tensor(np.asarray(frames) if not torch.is_tensor(frames) else frames)[None]
        if not frames.is_cuda:
            if not torch.cuda.is_available():
                raise RuntimeError("vince_amd transforms run on the GPU (csrc/augment.hip); no CPU fallback")
            frames = frames.cuda()
        frames = frames.contiguous()
        n, hs, ws, _ = frames.shape
        if self.data_subset == "val":
            out = self.apply_val(frames)
        else:
            src = None if repeats == 1 else np.tile(np.arange(n), repeats)
            out = self.apply(frames, self.draw(n * repeats, (hs, ws), src))
        if single:
            return out.float_tensor()[0]
        return out.float_tensor() if as_tensor else out


def _make(name, recipe):
    return type(name, (BatchTransform,), {"recipe": recipe, "__doc__": "utils/transforms.py `%s` as a batch recipe: %r" % (name, recipe)})


for _name, _recipe in RECIPES.items():
    globals()[_name] = _make(_name, _recipe)

__all__ = list(RECIPES) + ["BatchTransform", "AugmentParams", "Recipe"]
