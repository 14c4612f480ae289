"""GPU input stage (csrc/augment.hip) against the Pillow-pinned CPU oracle (oracle/augment_oracle.py, itself
checked against Pillow in tests/test_augment_oracle_cpu.py): uint8 stages bit-exact, the float tail to fp32 rounding."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import augment_oracle as ao  # noqa: E402

DEV = "cuda:0"


def _frames(n, h, w, seed):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = (127 + 120 * np.sin(yy / 9.0)[..., None] * np.cos(xx[..., None] / 7.0 + np.arange(3))).astype(np.uint8)
    x[::2, :, : w // 2] = smooth[:, : w // 2]
    return x


def _ops():
    from vince_amd import ops
    return ops


@pytest.mark.parametrize("hs,ws,h,w", [(120, 160, 64, 64), (240, 320, 224, 224), (60, 50, 75, 75), (64, 64, 64, 64)])
def test_resized_crop_is_pillow_exact(hs, ws, h, w):
    frames = _frames(6, hs, ws, hs + ws)
    rng = np.random.default_rng(1)
    boxes = [(0, 0, hs, ws), (hs - 1, ws - 1, 1, 1), (3, 5, hs // 2, ws // 3), (0, 0, min(hs, h), min(ws, w))]
    while len(boxes) < 8:
        ch, cw = int(rng.integers(1, hs + 1)), int(rng.integers(1, ws + 1))
        boxes.append((int(rng.integers(0, hs - ch + 1)), int(rng.integers(0, ws - cw + 1)), ch, cw))
    src = np.array([0, 1, 2, 3, 4, 5, 2, 2], np.int64)
    out = _ops().aug_resized_crop_u8(torch.from_numpy(frames).to(DEV), torch.tensor(boxes, dtype=torch.int32, device=DEV), (h, w),
                                     torch.from_numpy(src).to(DEV)).cpu().numpy()
    for i, (top, left, ch, cw) in enumerate(boxes):
        ref = ao.resized_crop_u8(frames[src[i]], top, left, ch, cw, h, w)
        assert np.array_equal(out[i], ref), (i, boxes[i], int((out[i] != ref).sum()))


def test_resized_crop_matches_pillow_directly():
    Image = pytest.importorskip("PIL.Image")
    frames = _frames(2, 200, 300, 9)
    boxes = [(10, 20, 150, 250), (0, 100, 200, 60)]
    out = _ops().aug_resized_crop_u8(torch.from_numpy(frames).to(DEV), torch.tensor(boxes, dtype=torch.int32, device=DEV),
                                     (224, 224)).cpu().numpy()
    for i, (top, left, ch, cw) in enumerate(boxes):
        ref = np.asarray(Image.fromarray(frames[i]).crop((left, top, left + cw, top + ch)).resize((224, 224), Image.BILINEAR))
        assert np.array_equal(out[i], ref)


def test_colour_ops_each_and_chained_are_pillow_exact():
    img = _frames(12, 48, 56, 3)
    chains = [[(0, 0.0)], [(0, 0.63)], [(0, 1.37)], [(1, 0.61)], [(1, 1.4)], [(2, 0.2)], [(2, 1.79)], [(3, 200.0)], [(4, 0.0)],
              [(2, 1.3), (3, 222.0), (0, 0.7), (1, 1.25)], [(4, 0.0), (1, 1.1), (3, 17.0), (0, 1.2), (2, 0.8)],
              [(3, 0.0), (1, 0.75), (4, 0.0)]]
    op = np.full((12, 5), -1, np.int32)
    fac = np.zeros((12, 5), np.float32)
    for i, c in enumerate(chains):
        for j, (code, f) in enumerate(c):
            op[i, j], fac[i, j] = code, f
    out = _ops().aug_color_u8(torch.from_numpy(img.copy()).to(DEV), torch.from_numpy(op).to(DEV),
                              torch.from_numpy(fac).to(DEV)).cpu().numpy()
    for i, c in enumerate(chains):
        # the oracle's hue op takes the hue factor; the kernel takes the uint8 shift it turns into
        cur = img[i]
        for code, f in c:
            if code == 3:
                hsv = ao.rgb_to_hsv_u8(cur)
                hsv[..., 0] = ((hsv[..., 0].astype(np.int32) + int(f)) & 0xFF).astype(np.uint8)
                cur = ao.hsv_to_rgb_u8(hsv)
            else:
                cur = ao.color_chain(cur, [(code, float(np.float32(f)))])
        assert np.array_equal(out[i], cur), (i, c, int((out[i] != cur).sum()))


def test_hsv_path_over_a_million_colours_and_every_shift():
    rng = np.random.default_rng(5)
    v = rng.integers(0, 1 << 24, 1 << 20, dtype=np.uint32)
    # plus the corners where the float formulas sit on a rounding edge: greys, saturated primaries, max == 255 / min == 0
    cols = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], -1).astype(np.uint8)
    cols[:256] = np.arange(256, dtype=np.uint8)[:, None]
    cols[256:512, 0], cols[256:512, 1], cols[256:512, 2] = 255, np.arange(256), 0
    img = np.ascontiguousarray(np.broadcast_to(cols.reshape(1, 1024, 1024, 3), (8, 1024, 1024, 3)))
    shifts = [0, 1, 43, 85, 128, 170, 213, 255]
    op = np.full((8, 1), 3, np.int32)
    fac = np.array(shifts, np.float32).reshape(8, 1)
    out = _ops().aug_color_u8(torch.from_numpy(img.copy()).to(DEV), torch.from_numpy(op).to(DEV),
                              torch.from_numpy(fac).to(DEV)).cpu().numpy()
    hsv0 = ao.rgb_to_hsv_u8(img[0])
    for i, s in enumerate(shifts):
        hsv = hsv0.copy()
        hsv[..., 0] = ((hsv[..., 0].astype(np.int32) + s) & 0xFF).astype(np.uint8)
        ref = ao.hsv_to_rgb_u8(hsv)
        assert np.array_equal(out[i], ref), (s, int((out[i] != ref).any(-1).sum()))


def test_contrast_mean_on_a_full_size_image():
    """The luma mean is a whole-image reduction (one workgroup per image): check at 224 x 224 and at a size that is not a
    multiple of the workgroup."""
    for h, w in [(224, 224), (225, 225), (37, 1023)]:
        img = _frames(3, h, w, h)
        op = np.full((3, 2), 1, np.int32)
        fac = np.array([[0.6, 1.3], [1.4, 0.9], [1.0, 0.5]], np.float32)
        out = _ops().aug_color_u8(torch.from_numpy(img.copy()).to(DEV), torch.from_numpy(op).to(DEV),
                                  torch.from_numpy(fac).to(DEV)).cpu().numpy()
        for i in range(3):
            ref = ao.color_chain(img[i], [(1, float(fac[i, 0])), (1, float(fac[i, 1]))])
            assert np.array_equal(out[i], ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_flip_normalize_blur_into_the_stem_layout(dtype):
    from vince_amd import constants
    from vince_amd.utils import transforms as T
    ops = _ops()
    n, h, w = 5, 64, 72
    img = _frames(n, h, w, 11)
    flip = np.array([0, 1, 0, 1, 1], np.uint8)
    sigma = np.array([0.0, 0.1, 1.3, 2.0, 0.0], np.float32)
    ks = T.blur_kernel_size(h)
    taps = T.blur_taps(sigma, ks)
    rows = ops.aug_blur_to_rows(torch.from_numpy(img).to(DEV), dtype, constants.IMAGENET_MEAN, constants.IMAGENET_STD,
                                torch.from_numpy(flip).to(DEV), taps.to(DEV), torch.from_numpy((sigma > 0).astype(np.uint8)).to(DEV))
    wp = ops.stem_row_width(w)
    assert rows.shape == (n, h, wp, 4)
    rows = rows.float().cpu()
    # margins and the 4th channel are zero: the stem's packed row taps read them
    assert float(rows[:, :, :ops.STEM_LEFT].abs().max()) == 0 and float(rows[:, :, ops.STEM_LEFT + w:].abs().max()) == 0
    assert float(rows[..., 3].abs().max()) == 0
    got = rows[:, :, ops.STEM_LEFT:ops.STEM_LEFT + w, :3].permute(0, 3, 1, 2).numpy()
    for i in range(n):
        x = img[i][:, ::-1] if flip[i] else img[i]
        ref = ao.to_tensor_normalize(np.ascontiguousarray(x))
        if sigma[i] > 0:
            ref = ao.gaussian_blur_chw(ref, ao.gaussian_kernel(h // 10, float(sigma[i])))
        tol = 2e-5 if dtype == torch.float32 else 2e-2
        assert np.abs(got[i] - ref).max() < tol, (i, np.abs(got[i] - ref).max())
    # an image that is not blurred goes through untouched arithmetic: identical to the plain uint8 layout kernel
    plain = ops.input_u8hwc_to_rows(torch.from_numpy(img).to(DEV), dtype, (h, w), constants.IMAGENET_MEAN, constants.IMAGENET_STD,
                                    None, torch.from_numpy(flip).to(DEV)).float().cpu()
    for i in (0, 4):
        assert torch.equal(plain[i], rows[i])


@pytest.mark.parametrize("name", ["MoCoV2ImagenetTransform", "SimCLRTransform", "StandardVideoTransform", "GOT10KTransform"])
def test_recipe_end_to_end_equals_the_oracle_pipeline(name):
    """A whole reference transform class as one batch call: same draws through the CPU oracle, sample by sample."""
    from vince_amd.utils import transforms as T
    t = getattr(T, name)(64, seed=7)
    frames = _frames(6, 90, 120, 21)
    params = t.draw(12, (90, 120), src_index=np.tile(np.arange(6), 2))
    out = t.apply(torch.from_numpy(frames).to(DEV), params)
    u8 = out.frames.cpu().numpy()
    ten = out.float_tensor().cpu().numpy()
    ks = T.blur_kernel_size(64)
    for i in range(12):
        chain = []
        for code, f in zip(params.op[i], params.factor[i]):
            chain.append((int(code), float(f)))
        # hue: the draw already holds the uint8 shift; replay it on the H plane
        cur = ao.resized_crop_u8(frames[params.src_index[i]], *[int(v) for v in params.box[i]], 64, 64)
        for code, f in chain:
            if code == 3:
                hsv = ao.rgb_to_hsv_u8(cur)
                hsv[..., 0] = ((hsv[..., 0].astype(np.int32) + int(f)) & 0xFF).astype(np.uint8)
                cur = ao.hsv_to_rgb_u8(hsv)
            elif code >= 0:
                cur = ao.color_chain(cur, [(code, f)])
        assert np.array_equal(u8[i], cur), (name, i)
        x = np.ascontiguousarray(cur[:, ::-1]) if params.flip[i] else cur
        ref = ao.to_tensor_normalize(x)
        if params.sigma[i] > 0:
            ref = ao.gaussian_blur_chw(ref, ao.gaussian_kernel(64 // 10, float(params.sigma[i])))
        assert np.abs(ten[i] - ref).max() < 2e-5, (name, i, np.abs(ten[i] - ref).max())
    # the per-sample contract of the reference (one HWC image in, one CHW float tensor out) and the val branch
    one = t(frames[0])
    assert tuple(one.shape) == (3, 64, 64) and one.dtype == torch.float32
    tv = getattr(T, name)(56, data_subset="val")
    val = tv(torch.from_numpy(frames).to(DEV), as_tensor=True).cpu().numpy()
    for i in range(6):
        r = ao.resize_bilinear_u8(frames[i], 64, 64)[4:60, 4:60]
        assert np.abs(val[i] - ao.to_tensor_normalize(np.ascontiguousarray(r))).max() < 2e-5


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_model_takes_augmented_frames(dtype):
    """VinceModel.get_embeddings on the handle a transform returns == on the float tensor it stands for."""
    from test_model_gpu import build, rel
    from vince_amd.utils import transforms as T
    _, model = build("ResNet18", 64, dtype, 31)
    t = T.MoCoV2ImagenetTransform(64, seed=5)
    frames = torch.from_numpy(_frames(8, 80, 96, 2)).to(DEV)
    params = t.draw(8, (80, 96))
    params.sigma[:] = [0, 1.1, 0, 0.4, 2.0, 0, 0.7, 0]
    u8 = t.apply(frames, params)
    assert u8.blur is not None
    model.eval()
    with torch.no_grad():
        a = model.get_embeddings({"data": u8})
        b = model.get_embeddings({"data": u8.float_tensor()})
        c = model.get_embeddings({"data": u8.float_reference()})
    tol = 1e-5 if dtype == "fp32" else 2e-2
    assert rel(a["extracted_features"].cpu(), b["extracted_features"].cpu()) < tol
    assert rel(a["extracted_features"].cpu(), c["extracted_features"].cpu()) < (1e-4 if dtype == "fp32" else 3e-2)


def test_solver_trains_on_augmented_uint8_frames():
    """solver -> data source -> transform -> U8Frames -> both encoders: three MoCo steps on a uint8 frame pool with the
    MoCo-v2 recipe; finite loss, queue advanced, and the first step's loss equals a solver fed the float tensors those
    handles stand for."""
    from oracle import vince_oracle as vo
    from vince_amd.config import make_args
    from vince_amd.data_source import AugmentedFrames
    from vince_amd.solvers.vince_solver import VinceSolver
    from vince_amd.utils import transforms as T

    pool = torch.from_numpy(_frames(32, 80, 96, 4)).to(DEV)

    class FloatTwin:
        """Same draws, handed over as float NCHW tensors."""
        def __init__(self):
            self.src = AugmentedFrames(pool, T.MoCoV2ImagenetTransform(64, seed=11), 16)

        def __call__(self, loader_id=0):
            b = self.src(loader_id)
            b["data"], b["queue_data"] = b["data"].float_tensor(), b["queue_data"].float_tensor()
            return b

    def run(source):
        torch.manual_seed(0)
        args = make_args(backbone="ResNet18", batch_size=16, vince_queue_size=64, input_size=(64, 64), compute_dtype="fp32",
                         batch_source=source)
        solver = VinceSolver(args)
        solver.model.load_state_dict(vo.seeded_state(vo.model_spec("ResNet18", 64), 2))
        solver.queue_model.queue_network.load_state_dict(vo.seeded_state(vo.model_spec("ResNet18", 64), 2))
        solver.vince_queue.vector_queue.copy_(torch.nn.functional.normalize(
            torch.randn(64, 64, generator=torch.Generator().manual_seed(1)), dim=1))
        solver.reset_epoch()
        losses = [float(solver.run_train_iteration()[0]["nce_loss"].detach()) for _ in range(3)]
        return losses, solver.vince_queue.current_tail

    la, tail_a = run(AugmentedFrames(pool, T.MoCoV2ImagenetTransform(64, seed=11), 16))
    lb, tail_b = run(FloatTwin())
    assert all(np.isfinite(la)) and tail_a == tail_b == 48
    assert abs(la[0] - lb[0]) < 1e-4 * max(1.0, abs(lb[0])), (la, lb)


def test_jigsaw_accepts_augmented_frames():
    """The jigsaw side (vince_model.py:144-171) fed with a transform's handle == fed with the float tensor it stands for."""
    from test_model_gpu import build, rel
    from vince_amd.utils import transforms as T
    _, model = build("ResNet18", 64, "fp32", 7, jigsaw=True)
    t = T.JigsawTransform(66, seed=1)
    u8 = t.apply(torch.from_numpy(_frames(4, 80, 96, 6)).to(DEV), t.draw(4, (80, 96)))
    model.eval()
    with torch.no_grad():
        torch.manual_seed(3)
        a = model.get_embeddings({"data": u8}, jigsaw=True)
        torch.manual_seed(3)
        b = model.get_embeddings({"data": u8.float_tensor()}, jigsaw=True)
    assert rel(a["embeddings"].cpu(), b["embeddings"].cpu()) < 1e-6


def test_kernels_against_the_committed_pillow_fixture():
    """tests/golden/g8_augment_pillow.npz: Pillow's own outputs (written by oracle/make_golden_augment.py) for crop + resize,
    every enhance op, hue shifts, grayscale and a five-step chain -- the kernels must reproduce every byte."""
    import os
    from oracle import make_golden_augment as mg
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g8_augment_pillow.npz"))
    ops = _ops()
    img = torch.from_numpy(g["image"]).to(DEV)[None].contiguous()
    for i, (box, (oh, ow)) in enumerate(zip(mg.BOXES, mg.SIZES)):
        out = ops.aug_resized_crop_u8(img, torch.tensor([box], dtype=torch.int32, device=DEV), (oh, ow))[0].cpu().numpy()
        assert np.array_equal(out, g["resized_crop_%d" % i]), i

    def run(chain):
        op = np.full((1, 5), -1, np.int32)
        fac = np.zeros((1, 5), np.float32)
        for j, (code, f) in enumerate(chain):
            op[0, j], fac[0, j] = code, f
        return ops.aug_color_u8(img.clone(), torch.from_numpy(op).to(DEV), torch.from_numpy(fac).to(DEV))[0].cpu().numpy()

    for f in mg.FACTORS:
        assert np.array_equal(run([(0, f)]), g["brightness_%g" % f]), f
        assert np.array_equal(run([(1, f)]), g["contrast_%g" % f]), f
        assert np.array_equal(run([(2, f)]), g["saturation_%g" % f]), f
    for s in mg.SHIFTS:
        assert np.array_equal(run([(3, float(s))]), g["hue_%d" % s]), s
    assert np.array_equal(run([(4, 0.0)]), g["gray"])
    assert np.array_equal(run(mg.CHAIN), g["chain"])
