"""Pins oracle/augment_oracle.py (the CPU restatement behind the GPU input stage's parity tests) against Pillow itself --
the third-party package whose arithmetic the reference's transforms run on (utils/transforms.py:62-235; requirements.txt:12
pins Pillow 7.0.0, this image has a newer one; torchvision is absent, so its few lines of glue per op are restated here with
Pillow calls from its published 0.5.0 source).  Also covers the host-side parameter draws of vince_amd/utils/transforms.py."""
import numpy as np
import pytest

from oracle import augment_oracle as ao

PIL = pytest.importorskip("PIL")
from PIL import Image, ImageEnhance  # noqa: E402


def _img(h, w, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    # low-frequency structure on half of the image so resampling sees smooth gradients as well as noise
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = (127 + 120 * np.sin(yy / 9.0)[..., None] * np.cos(xx[..., None] / 7.0 + np.arange(3))).astype(np.uint8)
    base[:, : w // 2] = smooth[:, : w // 2]
    return base


@pytest.mark.parametrize("h,w,oh,ow", [(97, 131, 64, 64), (300, 400, 224, 224), (50, 60, 224, 224), (224, 224, 224, 224),
                                       (223, 500, 224, 224), (480, 37, 75, 75), (1, 9, 5, 5), (225, 225, 75, 75)])
def test_resize_is_pillow_bilinear(h, w, oh, ow):
    img = _img(h, w, h * 1000 + w)
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
    assert np.array_equal(ao.resize_bilinear_u8(img, oh, ow), ref)


def test_resized_crop_is_crop_then_resize():
    img = _img(240, 320, 5)
    for top, left, ch, cw in [(0, 0, 240, 320), (10, 20, 100, 200), (239, 319, 1, 1), (17, 3, 223, 61)]:
        ref = np.asarray(Image.fromarray(img).crop((left, top, left + cw, top + ch)).resize((64, 48), Image.BILINEAR))
        assert np.array_equal(ao.resized_crop_u8(img, top, left, ch, cw, 48, 64), ref)


def test_luma_and_enhance_blends():
    img = _img(64, 80, 1)
    pil = Image.fromarray(img)
    assert np.array_equal(ao.luma_u8(img), np.asarray(pil.convert("L")))
    assert np.array_equal(ao.to_grayscale3(img), np.asarray(pil.convert("L").convert("RGB")))
    for f in [0.0, 0.2, 0.4999, 0.6, 1.0, 1.0001, 1.2, 1.4, 1.8]:
        assert np.array_equal(ao.adjust_brightness(img, f), np.asarray(ImageEnhance.Brightness(pil).enhance(f))), f
        assert np.array_equal(ao.adjust_contrast(img, f), np.asarray(ImageEnhance.Contrast(pil).enhance(f))), f
        assert np.array_equal(ao.adjust_saturation(img, f), np.asarray(ImageEnhance.Color(pil).enhance(f))), f


def test_hsv_round_trip_over_every_colour():
    v = np.arange(1 << 24, dtype=np.uint32)
    cube = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], -1).astype(np.uint8).reshape(4096, 4096, 3)
    assert np.array_equal(ao.rgb_to_hsv_u8(cube), np.asarray(Image.fromarray(cube, "RGB").convert("HSV")))
    assert np.array_equal(ao.hsv_to_rgb_u8(cube), np.asarray(Image.fromarray(cube, "HSV").convert("RGB")))


def _tv_adjust_hue(pil, hue_factor):
    """torchvision 0.5 functional.adjust_hue on a PIL image."""
    h, s, v = pil.convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    with np.errstate(over="ignore"):
        np_h += np.uint8(int(hue_factor * 255) & 0xFF)
    return Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB")


def test_hue_and_full_chain():
    img = _img(48, 56, 2)
    pil = Image.fromarray(img)
    for hf in [-0.4, -0.2, -0.01, 0.0, 0.1, 0.4]:
        assert np.array_equal(ao.adjust_hue(img, hf), np.asarray(_tv_adjust_hue(pil, hf))), hf
    # a ColorJitter order + grayscale, the way MoCoV2ImagenetTransform / StandardVideoTransform compose them
    cur = pil
    ops = [(2, 1.3), (3, -0.13), (0, 0.7), (1, 1.25)]
    for code, f in ops:
        cur = {0: lambda im: ImageEnhance.Brightness(im).enhance(f), 1: lambda im: ImageEnhance.Contrast(im).enhance(f),
               2: lambda im: ImageEnhance.Color(im).enhance(f), 3: lambda im: _tv_adjust_hue(im, f)}[code](cur)
    assert np.array_equal(ao.color_chain(img, ops), np.asarray(cur))
    assert np.array_equal(ao.color_chain(img, ops + [(4, 0)]), np.asarray(cur.convert("L").convert("RGB")))


def test_blur_taps_and_separable_blur():
    import torch
    from vince_amd.utils import transforms as T
    ks = T.blur_kernel_size(224)
    assert ks == 23 and T.blur_kernel_size(64) == 7 and T.blur_kernel_size(100) == 11
    taps = T.blur_taps(np.array([0.0, 0.1, 1.3, 2.0], np.float32), ks)
    assert float(taps[0].abs().sum()) == 0.0
    for i, s in [(1, 0.1), (2, 1.3), (3, 2.0)]:
        assert torch.allclose(taps[i], ao.gaussian_kernel(224 // 10, float(np.float32(s))), rtol=1e-6, atol=1e-9)
        assert abs(float(taps[i].sum()) - 1.0) < 1e-6
    x = torch.randn(3, 20, 24, generator=torch.Generator().manual_seed(0)).numpy()
    y = ao.gaussian_blur_chw(x, ao.gaussian_kernel(6, 1.0))
    # against a dense 2-D convolution with the outer-product kernel, zero padding
    k = ao.gaussian_kernel(6, 1.0).double().numpy()
    ref = np.zeros_like(x, dtype=np.float64)
    pad = np.pad(x.astype(np.float64), ((0, 0), (3, 3), (3, 3)))
    for a in range(7):
        for b in range(7):
            ref += k[a] * k[b] * pad[:, a:a + 20, b:b + 24]
    assert np.abs(y - ref).max() < 1e-5


def test_recipes_match_the_reference_class_list_and_draws_are_valid():
    import os
    from vince_amd.utils import transforms as T
    ref = "/root/reference/utils/transforms.py"
    if os.path.exists(ref):
        src = open(ref).read()
        for name in T.RECIPES:
            assert "class %s(" % name in src, name
    for name, recipe in T.RECIPES.items():
        t = getattr(T, name)(224, seed=3)
        p = t.draw(64, (240, 320))
        top, left, h, w = p.box.T
        assert (h > 0).all() and (w > 0).all() and (top >= 0).all() and (left >= 0).all()
        assert (top + h <= 240).all() and (left + w <= 320).all()
        area = h * w / (240.0 * 320.0)
        assert area.min() >= recipe.crop_scale[0] * 0.9 and area.max() <= 1.0
        n_jit = 0 if recipe.jitter is None else sum(1 for v in recipe.jitter if v > 0)
        for i in range(64):
            codes = [c for c in p.op[i] if c >= 0]
            jit = [c for c in codes if c != T.OP_GRAY]
            assert sorted(jit) == list(range(4))[:n_jit] or len(jit) == n_jit
            if T.OP_GRAY in codes:
                assert codes.index(T.OP_GRAY) == (0 if recipe.gray_first else len(codes) - 1)
            for c, f in zip(p.op[i], p.factor[i]):
                if c in (0, 1, 2):
                    lim = recipe.jitter[c]
                    assert max(0.0, 1 - lim) - 1e-6 <= f <= 1 + lim + 1e-6
                if c == 3:
                    assert f == int(f) and 0 <= f <= 255
        assert ((p.sigma == 0) | ((p.sigma >= 0.1) & (p.sigma <= 2.0))).all()
        assert (p.sigma > 0).any() == (recipe.blur_p > 0)
    # the aspect-ratio fallback (a frame no draw fits): central crop with the clamped aspect
    t = T.StandardVideoTransform(224, seed=0)
    t.recipe = T.Recipe((4.0, 5.0))
    assert t._draw_box(100, 400) == (0, 133, 100, 133) or t._draw_box(100, 400)[2:] == (100, 133)


def test_g10_recipes_equal_the_reference_compose_lists():
    """Golden set G10 (oracle/make_golden_recipes.py): op order and literal parameters of every Compose list in the
    reference's utils/transforms.py, read from its syntax tree.  RECIPES must restate exactly those."""
    import json
    import os
    from vince_amd import constants
    from vince_amd.utils import transforms as T
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g10_recipes.json")))
    ref_classes = {k for k, v in g.items() if "train" in v}
    assert ref_classes == set(T.RECIPES)                                  # every class with a train recipe, and no other
    default_ratio = (3.0 / 4.0, 4.0 / 3.0)                               # torchvision 0.5 RandomResizedCrop default
    for name, recipe in T.RECIPES.items():
        ops = [o for o in g[name]["train"] if o[0] != "ToPILImage"]
        names = [o[0] for o in ops]
        by = {o[0]: o for o in ops}
        # ---- order: crop, colour ops, flip, ToTensor, Normalize, [blur]
        want = ["RandomResizedCrop"]
        if recipe.jitter is not None or recipe.gray_p > 0:
            want += ["RandomGrayscale", "ColorJitter"] if recipe.gray_first else ["ColorJitter", "RandomGrayscale"]
        want += ["RandomHorizontalFlip", "ToTensor", "Normalize"]
        if recipe.blur_p > 0:
            want += ["RandomApply"]
        assert names == want, (name, names, want)
        # ---- crop
        _, args, kw = by["RandomResizedCrop"]
        assert args == ["SIZE"] and tuple(kw["scale"]) == recipe.crop_scale
        assert tuple(kw.get("ratio", default_ratio)) == recipe.crop_ratio
        assert kw.get("interpolation", "BILINEAR") == "BILINEAR"             # BILINEAR is also torchvision's default
        # ---- colour
        if recipe.jitter is not None:
            assert tuple(by["ColorJitter"][1]) == recipe.jitter and by["ColorJitter"][2] == {}
            assert by["RandomGrayscale"][2] == {"p": recipe.gray_p}
        else:
            assert "ColorJitter" not in by and "RandomGrayscale" not in by and recipe.gray_p == 0
        # ---- flip (torchvision default p = 0.5), ToTensor(scale=255), Normalize literals
        assert by["RandomHorizontalFlip"][1:] == [[], {}] and recipe.flip_p == 0.5
        assert by["ToTensor"][2] == {"scale": 255}
        mean, std = by["Normalize"][2]["mean"], by["Normalize"][2]["std"]
        # ToTensor(scale=255) keeps 0..255 floats / 255 -> the constants the layout kernel uses are mean * 255, std * 255
        assert np.allclose(np.asarray(mean, np.float32) * 255, constants.IMAGENET_MEAN, rtol=0, atol=0)
        assert np.allclose(np.asarray(std, np.float32) * 255, constants.IMAGENET_STD, rtol=0, atol=0)
        # ---- blur: RandomApply([RandomGaussianBlur(size[0] // 10)], p) AFTER Normalize
        if recipe.blur_p > 0:
            _, (inner,), kw = by["RandomApply"]
            assert kw == {"p": recipe.blur_p} and inner == [["RandomGaussianBlur", ["SIZE0//10"], {}]]
    assert g["RandomGaussianBlur"]["defaults"] == {"sigma_range": [0.1, 2.0]}          # the draw in BatchTransform.draw
    for size in (64, 100, 224):
        ks = size // 10
        assert T.blur_kernel_size(size) == (ks + 1 if ks % 2 == 0 else ks)
    # ---- val: every class inherits BasicImagenetTransform's Resize(size / 0.875) + CenterCrop(size)
    assert [k for k, v in g.items() if "val" in v] == ["BasicImagenetTransform"]
    val = [o for o in g["BasicImagenetTransform"]["val"] if o[0] != "ToPILImage"]
    assert [o[0] for o in val] == ["Resize", "CenterCrop", "ToTensor", "Normalize"]
    assert val[0][1] == ["SIZE/0.875"] and val[0][2] == {"interpolation": "BILINEAR"} and val[1][1] == ["SIZE"]
    for name in T.RECIPES:                                                # all reach BasicImagenetTransform through bases
        k = name
        while k != "BasicImagenetTransform":
            (k,) = g[k]["bases"]
    assert g["constants"]["IMAGENET_MEAN"] == {"values": [0.485, 0.456, 0.406], "times": 255.0}
    assert np.array_equal(np.asarray(g["constants"]["IMAGENET_MEAN"]["values"], np.float32) * 255, constants.IMAGENET_MEAN)
    assert np.array_equal(np.asarray(g["constants"]["IMAGENET_STD"]["values"], np.float32) * 255, constants.IMAGENET_STD)


def _g8():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g8_augment_pillow.npz"))


def _hue_by_shift(img, shift):
    hsv = ao.rgb_to_hsv_u8(img)
    hsv[..., 0] = ((hsv[..., 0].astype(np.int32) + int(shift)) & 0xFF).astype(np.uint8)
    return ao.hsv_to_rgb_u8(hsv)


def test_oracle_against_the_committed_pillow_fixture():
    """tests/golden/g8_augment_pillow.npz (oracle/make_golden_augment.py) holds Pillow's own outputs: this check does not
    need Pillow where it runs."""
    from oracle import make_golden_augment as mg
    g = _g8()
    img = g["image"]
    assert np.array_equal(img, mg.seeded_image())
    for i, ((top, left, ch, cw), (oh, ow)) in enumerate(zip(mg.BOXES, mg.SIZES)):
        assert np.array_equal(ao.resized_crop_u8(img, top, left, ch, cw, oh, ow), g["resized_crop_%d" % i]), i
    for f in mg.FACTORS:
        assert np.array_equal(ao.adjust_brightness(img, f), g["brightness_%g" % f])
        assert np.array_equal(ao.adjust_contrast(img, f), g["contrast_%g" % f])
        assert np.array_equal(ao.adjust_saturation(img, f), g["saturation_%g" % f])
    for s in mg.SHIFTS:
        assert np.array_equal(_hue_by_shift(img, s), g["hue_%d" % s])
    assert np.array_equal(ao.to_grayscale3(img), g["gray"])
    cur = img
    for code, f in mg.CHAIN:
        cur = _hue_by_shift(cur, f) if code == 3 else ao.color_chain(cur, [(code, f)])
    assert np.array_equal(cur, g["chain"])
