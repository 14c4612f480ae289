"""CPU oracle for the GPU input stage: the train-time image pipeline of utils/transforms.py:62-235.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  numpy restatement of the arithmetic the reference gets from two
third-party packages that are not under /root/reference:

  * Pillow (requirements.txt:12 pins 7.0.0): BILINEAR resize (``Resample.c``: antialiased triangle filter, 22-bit fixed-point
    coefficients, horizontal pass -> uint8 -> vertical pass), ``Image.blend`` (``Blend.c``), ``convert("L")`` (ITU-R 601-2
    luma in 16-bit fixed point), ``convert("HSV")`` / back (``Convert.c``, colorsys in float);
  * torchvision 0.5.0 (requirements.txt:18) ``transforms.functional``: ``resized_crop`` = crop then resize,
    ``adjust_brightness / contrast / saturation`` = ``ImageEnhance`` blends against a degenerate image, ``adjust_hue`` = uint8
    wrap-around shift of the H plane, ``to_grayscale(num_output_channels=3)``, ``hflip``;

plus the reference's own Gaussian blur (utils/util_functions.py:104-132) and ToTensor + Normalize (utils/transforms.py:72-73).

Pinning: torchvision is absent from this image, Pillow 12.2.0 is present.  ``tests/test_augment_oracle_cpu.py`` checks every
function here against Pillow itself (the HSV conversions over all 2^24 colours, resize over a sweep of boxes / sizes, the
blends over every (value, factor) pair class) -- so the oracle is pinned to the Pillow installed in this image, and the
torchvision layer (a few lines of glue per op, restated from its published source) is NOT pinned by execution.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2   # Resample.c


# ----------------------------------------------------------------------------------------------- resize (Pillow Resample.c)
def bilinear_coeffs(in_size, in0, in1, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter: returns (bounds [out,2] int, kk [out,ksize] int32).
    Scalar double arithmetic in the order of the C source."""
    scale = float(np.float32(in1) - np.float32(in0)) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int64)
    kk = np.zeros((out_size, ksize), np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = float(in0) + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(xmax, np.float64)
        ww = 0.0
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            if t < 0.0:
                t = -t
            w[x] = 1.0 - t if t < 1.0 else 0.0
            ww += w[x]
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis0(img, bounds, kk):
    """One 8-bit pass along axis 0 of img [L, ..., C] uint8 (ImagingResampleVertical_8bpc / Horizontal on the transpose)."""
    out = np.empty((bounds.shape[0],) + img.shape[1:], np.uint8)
    src = img.astype(np.int64)
    for xx in range(bounds.shape[0]):
        xmin, xmax = int(bounds[xx, 0]), int(bounds[xx, 1])
        ss = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(xmax):
            ss = ss + src[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(ss >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_bilinear_u8(img, out_h, out_w):
    """``Image.fromarray(img).resize((out_w, out_h), Image.BILINEAR)`` for an HWC uint8 array: horizontal pass first, its
    uint8 result feeds the vertical pass; a pass whose size does not change is skipped (ImagingResample)."""
    h, w = img.shape[:2]
    cur = img
    if w != out_w:
        b, k = bilinear_coeffs(w, 0, w, out_w)
        cur = np.ascontiguousarray(_resample_axis0(np.ascontiguousarray(cur.transpose(1, 0, 2)), b, k).transpose(1, 0, 2))
    if h != out_h:
        b, k = bilinear_coeffs(h, 0, h, out_h)
        cur = _resample_axis0(cur, b, k)
    return cur


def resized_crop_u8(img, top, left, ch, cw, out_h, out_w):
    """torchvision.transforms.functional.resized_crop: crop (a new image, so the filter never sees pixels outside the
    window), then BILINEAR resize."""
    return resize_bilinear_u8(np.ascontiguousarray(img[top:top + ch, left:left + cw]), out_h, out_w)


# ----------------------------------------------------------------------------------------------- colour (Pillow Convert.c, Blend.c)
def luma_u8(img):
    """convert("L"): (R*19595 + G*38470 + B*7471 + 0x8000) >> 16."""
    x = img.astype(np.int64)
    return ((x[..., 0] * 19595 + x[..., 1] * 38470 + x[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend_u8(deg, img, alpha):
    """Image.blend(deg, img, alpha) (Blend.c): float32 arithmetic, truncation; clipping only on the extrapolating branch."""
    a = np.float32(alpha)
    d = deg.astype(np.int32)
    t = d.astype(np.float32) + a * (img.astype(np.int32) - d).astype(np.float32)
    if 0.0 <= float(a) <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    out = np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32)))
    return out.astype(np.uint8)


def adjust_brightness(img, f):
    return blend_u8(np.zeros_like(img), img, f)


def adjust_contrast(img, f):
    """ImageEnhance.Contrast: degenerate = the rounded mean of the luma plane everywhere."""
    L = luma_u8(img)
    mean = int(float(L.astype(np.int64).sum()) / L.size + 0.5)
    return blend_u8(np.full_like(img, mean), img, f)


def adjust_saturation(img, f):
    return blend_u8(np.repeat(luma_u8(img)[..., None], 3, axis=-1), img, f)


def rgb_to_hsv_u8(img):
    """Convert.c rgb2hsv_row (float32 / double mix of the C source)."""
    r, g, b = [img[..., i].astype(np.int32) for i in range(3)]
    maxc = np.maximum(r, np.maximum(g, b))
    minc = np.minimum(r, np.minimum(g, b))
    flat = maxc == minc
    cr = np.where(flat, 1, maxc - minc).astype(np.float32)
    s = cr / np.where(flat, 1, maxc).astype(np.float32)
    rc = (maxc - r).astype(np.float32) / cr
    gc = (maxc - g).astype(np.float32) / cr
    bc = (maxc - b).astype(np.float32) / cr
    h = np.where(r == maxc, (bc - gc).astype(np.float64),
                 np.where(g == maxc, 2.0 + rc.astype(np.float64) - bc.astype(np.float64),
                          4.0 + gc.astype(np.float64) - rc.astype(np.float64)))
    h = h.astype(np.float32)
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    uh = np.where(flat, 0, uh)
    us = np.where(flat, 0, us)
    return np.stack([uh, us, maxc], -1).astype(np.uint8)


def hsv_to_rgb_u8(hsv):
    """Convert.c hsv2rgb."""
    h, s, v = [hsv[..., i].astype(np.int32) for i in range(3)]
    hf = h.astype(np.float32).astype(np.float64) * 6.0 / 255.0
    i = np.floor(hf).astype(np.int32)
    f = (hf - i.astype(np.float32).astype(np.float64)).astype(np.float32)
    fs = (s.astype(np.float32).astype(np.float64) / 255.0).astype(np.float32)
    vf = v.astype(np.float32).astype(np.float64)
    fs64, f64 = fs.astype(np.float64), f.astype(np.float64)

    def rnd(x):   # C round(): half away from zero
        return np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5)).astype(np.int64)
    p = np.clip(rnd(vf * (1.0 - fs64)), 0, 255)
    q = np.clip(rnd(vf * (1.0 - fs64 * f64)), 0, 255)
    t = np.clip(rnd(vf * (1.0 - fs64 * (1.0 - f64))), 0, 255)
    k = i % 6
    r = np.select([k == 0, k == 1, k == 2, k == 3, k == 4, k == 5], [v, q, p, p, t, v])
    g = np.select([k == 0, k == 1, k == 2, k == 3, k == 4, k == 5], [t, v, v, q, p, p])
    b = np.select([k == 0, k == 1, k == 2, k == 3, k == 4, k == 5], [p, p, t, v, v, q])
    grey = s == 0
    r, g, b = np.where(grey, v, r), np.where(grey, v, g), np.where(grey, v, b)
    return np.stack([r, g, b], -1).astype(np.uint8)


def adjust_hue(img, hue_factor):
    """torchvision F.adjust_hue: H plane (uint8) += uint8(hue_factor * 255) with wrap-around, S and V untouched."""
    hsv = rgb_to_hsv_u8(img)
    shift = int(np.uint8(np.int64(hue_factor * 255) & 0xFF))
    hsv[..., 0] = ((hsv[..., 0].astype(np.int32) + shift) & 0xFF).astype(np.uint8)
    return hsv_to_rgb_u8(hsv)


def to_grayscale3(img):
    return np.repeat(luma_u8(img)[..., None], 3, axis=-1)


OPS = {0: adjust_brightness, 1: adjust_contrast, 2: adjust_saturation, 3: adjust_hue}
OP_GRAY = 4


def color_chain(img, ops):
    """ops: sequence of (code, factor): 0 brightness, 1 contrast, 2 saturation, 3 hue, 4 grayscale (factor unused)."""
    for code, f in ops:
        if code == OP_GRAY:
            img = to_grayscale3(img)
        elif code >= 0:
            img = OPS[int(code)](img, float(f))
    return img


# ----------------------------------------------------------------------------------------------- tensor side
IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], np.float32)   # utils/transforms.py:73
IMAGENET_STD = np.array([0.229, 0.224, 0.225], np.float32)


def to_tensor_normalize(img):
    """pt_util.ToTensor(scale=255) + Normalize(mean, std) (utils/transforms.py:72-73): CHW float32."""
    x = img.astype(np.float32).transpose(2, 0, 1) / np.float32(255)
    return (x - IMAGENET_MEAN[:, None, None]) / IMAGENET_STD[:, None, None]


def gaussian_kernel(kernel_size, sigma):
    """utils/util_functions.py:105-117: odd kernel size, exp(-0.5/sigma^2 * d^2) normalised (float32 torch arithmetic)."""
    import torch
    if kernel_size % 2 == 0:
        kernel_size += 1
    rng = (kernel_size - 1) * 0.5 - torch.arange(kernel_size)
    k = torch.exp(-0.5 / (sigma ** 2) * (rng ** 2))
    k /= max(1e-10, k.sum())
    return k


def gaussian_blur_chw(x, kernel):
    """utils/util_functions.py:119-132: depthwise conv along H then along W, zero padding, on the normalised tensor."""
    import torch
    import torch.nn.functional as F
    x = torch.as_tensor(x)[None]
    c, ks = x.shape[1], kernel.numel()
    k = kernel[None, :].expand(c, ks)
    x = F.conv2d(x, k[:, None, :, None], padding=(ks // 2, 0), groups=c)
    x = F.conv2d(x, k[:, None, None, :], padding=(0, ks // 2), groups=c)
    return x[0].numpy()


def full_pipeline(img, box, out_hw, ops, flip, blur_kernel=None):
    """One training sample through crop+resize -> colour chain -> flip -> ToTensor/Normalize -> optional blur."""
    top, left, ch, cw = box
    x = resized_crop_u8(img, top, left, ch, cw, out_hw[0], out_hw[1])
    x = color_chain(x, ops)
    if flip:
        x = np.ascontiguousarray(x[:, ::-1])
    t = to_tensor_normalize(x)
    if blur_kernel is not None:
        t = gaussian_blur_chw(t, blur_kernel)
    return x, t
