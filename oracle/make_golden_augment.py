"""Generate tests/golden/g8_augment_pillow.npz by running PILLOW (the third-party package under the reference's transforms,
utils/transforms.py:62-235) on a seeded image: the uint8 results of crop + BILINEAR resize, every ImageEnhance op, the
torchvision-0.5 hue shift, grayscale and one full chain.  TEST INFRASTRUCTURE; needs Pillow:

    python -m oracle.make_golden_augment

The fixture holds the input image and Pillow's outputs only; it pins the oracle (and through it the HIP kernels) to the
Pillow version that wrote it (recorded in the file) wherever the tests later run.
"""
import os

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g8_augment_pillow.npz")


def seeded_image(h=60, w=80, seed=8):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    img[:, : w // 2] = (127 + 120 * np.sin(yy / 9.0)[..., None] * np.cos(xx[..., None] / 7.0 + np.arange(3))).astype(np.uint8)[:, : w // 2]
    return img


def tv_adjust_hue(pil, shift_u8):
    from PIL import Image
    h, s, v = pil.convert("HSV").split()
    np_h = (np.array(h, dtype=np.uint8).astype(np.int32) + shift_u8) & 0xFF
    return Image.merge("HSV", (Image.fromarray(np_h.astype(np.uint8), "L"), s, v)).convert("RGB")


BOXES = [(0, 0, 60, 80), (5, 7, 40, 33), (59, 79, 1, 1), (10, 0, 50, 80)]        # top, left, height, width
SIZES = [(32, 32), (48, 40), (75, 75), (60, 80)]
FACTORS = [0.0, 0.37, 1.0, 1.42, 1.8]
SHIFTS = [0, 13, 128, 240]
CHAIN = [(4, 0.0), (2, 1.3), (3, 222.0), (0, 0.7), (1, 1.25)]


def main():
    import PIL
    from PIL import Image, ImageEnhance
    img = seeded_image()
    pil = Image.fromarray(img)
    out = {"image": img, "pillow_version": np.array(PIL.__version__)}
    for i, ((top, left, ch, cw), (oh, ow)) in enumerate(zip(BOXES, SIZES)):
        out["resized_crop_%d" % i] = np.asarray(pil.crop((left, top, left + cw, top + ch)).resize((ow, oh), Image.BILINEAR))
    for f in FACTORS:
        out["brightness_%g" % f] = np.asarray(ImageEnhance.Brightness(pil).enhance(f))
        out["contrast_%g" % f] = np.asarray(ImageEnhance.Contrast(pil).enhance(f))
        out["saturation_%g" % f] = np.asarray(ImageEnhance.Color(pil).enhance(f))
    for s in SHIFTS:
        out["hue_%d" % s] = np.asarray(tv_adjust_hue(pil, s))
    out["gray"] = np.asarray(pil.convert("L").convert("RGB"))
    cur = pil
    for code, f in CHAIN:
        cur = {0: lambda im: ImageEnhance.Brightness(im).enhance(f), 1: lambda im: ImageEnhance.Contrast(im).enhance(f),
               2: lambda im: ImageEnhance.Color(im).enhance(f), 3: lambda im: tv_adjust_hue(im, int(f)),
               4: lambda im: im.convert("L").convert("RGB")}[code](cur)
    out["chain"] = np.asarray(cur)
    np.savez_compressed(OUT, **out)
    print(OUT, os.path.getsize(OUT), "bytes, Pillow", PIL.__version__)


if __name__ == "__main__":
    main()
