"""Python handle on the C++ trunk engine (csrc/trunk.hip): owns the engine handle and the ctypes pointer tables."""
import ctypes

import torch

from . import ops
from ._lib import TrunkCfg, VINCE_BF16, VINCE_F32, VINCE_F32X3, VINCE_F32X3F, check, lib

ARCH_CODE = {"ResNet18": 18, "ResNet50": 50}

# scratch workspace shared by every no-grad forward on a device (key encoder, validation): nothing in it outlives
# the call, so one buffer of the largest size seen is enough.
_NOGRAD_WS = {}


def nograd_workspace(device, nbytes):
    cur = _NOGRAD_WS.get(device)
    if cur is None or cur.numel() < nbytes:
        cur = None
        _NOGRAD_WS[device] = None
        cur = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _NOGRAD_WS[device] = cur
    return cur


class TrunkPlan:
    """Static description of the trunk's parameters / BN layers, queried from the engine (reference state-dict order)."""

    def __init__(self, arch):
        L = lib()
        h = ctypes.c_void_p()
        check(L.vince_trunk_create(ctypes.byref(TrunkCfg(arch=ARCH_CODE[arch], N=1, H=64, W=64, dtype=VINCE_F32)),
                                   ctypes.byref(h)))
        self.params = []   # (name, kind, shape, bn_index)
        for i in range(L.vince_trunk_num_params(h)):
            buf = ctypes.create_string_buffer(128)
            kind, shape, bn = ctypes.c_int32(), (ctypes.c_int32 * 4)(), ctypes.c_int32()
            check(L.vince_trunk_param_info(h, i, buf, 128, ctypes.byref(kind), ctypes.byref(shape), ctypes.byref(bn)))
            shp = tuple(shape) if kind.value == 0 else (shape[0],)
            self.params.append((buf.value.decode(), kind.value, shp, bn.value))
        self.bns = []      # (name, channels)
        for i in range(L.vince_trunk_num_bn(h)):
            buf = ctypes.create_string_buffer(128)
            c = ctypes.c_int32()
            check(L.vince_trunk_bn_info(h, i, buf, 128, ctypes.byref(c)))
            self.bns.append((buf.value.decode(), c.value))
        self.out_channels = L.vince_trunk_out_channels(h)
        L.vince_trunk_destroy(h)


class Trunk:
    """One engine instance for a fixed (arch, N, H, W, dtype).  x3: float32 tensors with every convolution as split-half products
    (VINCE_F32X3: three half-precision MFMAs of hi / lo halves instead of fp32 MFMAs)."""

    def __init__(self, arch, N, H, W, dtype, x3=False):
        L = lib()
        self.arch, self.N, self.H, self.W, self.dtype, self.x3 = arch, N, H, W, dtype, bool(x3)
        if self.x3 and dtype != torch.float32:
            raise ValueError("Trunk: split-half products (x3) keep float32 tensors")
        self._h = ctypes.c_void_p()
        # x3 == "f": the mixed mode VINCE_F32X3F (split-half forward, single bfloat16 products in every gradient convolution)
        code = ((VINCE_F32X3F if x3 == "f" else VINCE_F32X3) if self.x3 else VINCE_F32) if dtype == torch.float32 else VINCE_BF16
        check(L.vince_trunk_create(ctypes.byref(TrunkCfg(arch=ARCH_CODE[arch], N=N, H=H, W=W, dtype=code)),
                                   ctypes.byref(self._h)))
        self.ws_bytes = L.vince_trunk_workspace_bytes(self._h)
        self.wc_bytes = L.vince_trunk_weight_cache_bytes(self._h)
        self.out_channels = L.vince_trunk_out_channels(self._h)
        oh, ow = ctypes.c_int32(), ctypes.c_int32()
        L.vince_trunk_out_hw(self._h, ctypes.byref(oh), ctypes.byref(ow))
        self.out_h, self.out_w = oh.value, ow.value

    def __del__(self):
        try:
            if self._h:
                lib().vince_trunk_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def prepare_weights(self, param_ptrs, wcache, part=0):
        """part: 0 every layer, 1 every layer but the stem, 2 the stem alone (include/vince_hip.h vince_trunk_prepare_weights_part)."""
        if part == 0:
            check(lib().vince_trunk_prepare_weights(self._h, param_ptrs, ctypes.c_void_p(wcache.data_ptr()), ops.stream_ptr()))
        else:
            check(lib().vince_trunk_prepare_weights_part(self._h, param_ptrs, ctypes.c_void_p(wcache.data_ptr()), int(part),
                                                         ops.stream_ptr()))

    def forward(self, param_ptrs, wcache, bn_running_ptrs, bn_nbt_ptrs, data, workspace, pooled, train_bn, perm=None,
                jigsaw_src=None, save=True):
        """save=False: no backward follows -- the engine may skip the bottleneck-internal activation tensors."""
        jh, jw = (0, 0) if jigsaw_src is None else jigsaw_src
        check(lib().vince_trunk_forward(
            self._h, param_ptrs, ctypes.c_void_p(wcache.data_ptr()), bn_running_ptrs, bn_nbt_ptrs,
            None if data is None else ctypes.c_void_p(data.data_ptr()),      # None: the input was staged (stage_u8)
            None if perm is None else ctypes.c_void_p(perm.data_ptr()), jh, jw,
            ctypes.c_void_p(workspace.data_ptr()), ctypes.c_void_p(pooled.data_ptr()), int(train_bn), int(save),
            ops.stream_ptr()))

    def stage_u8(self, workspace, frames_u8, crop_yx=None, flip=None, perm=None, mean255=None, std255=None):
        """GPU input stage: uint8 HWC frames [N][Hs][Ws][3] -> the stem layout inside `workspace` (crop window, flip, batch
        gather and (u8 - mean) / std in one pass).  Follow with forward(..., data=None)."""
        ops.require_gpu(frames_u8, crop_yx, flip, perm)
        n, hs, ws, c = frames_u8.shape
        if frames_u8.dtype != torch.uint8 or c != 3 or n != self.N or not frames_u8.is_contiguous():
            raise ValueError("stage_u8: expected contiguous uint8 [%d, Hs, Ws, 3] frames, got %s %s"
                             % (self.N, frames_u8.dtype, tuple(frames_u8.shape)))
        wp, left = ctypes.c_int32(), ctypes.c_int32()
        check(lib().vince_trunk_stem_join(self._h, ops.stream_ptr()))     # a deferred stem weight gradient may still be reading x0
        x0 = lib().vince_trunk_input_ptr(self._h, ctypes.c_void_p(workspace.data_ptr()), ctypes.byref(wp), ctypes.byref(left))
        mean = (ctypes.c_float * 3)(*[float(v) for v in mean255])
        std = (ctypes.c_float * 3)(*[float(v) for v in std255])
        code = VINCE_F32 if self.dtype == torch.float32 else VINCE_BF16
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())   # noqa: E731
        check(lib().vince_input_u8hwc_to_rows(code, p(frames_u8), p(perm), p(crop_yx), p(flip), mean, std, ctypes.c_void_p(x0),
                                              self.N, hs, ws, self.H, self.W, wp.value, left.value, ops.stream_ptr()))

    def stage_blur(self, workspace, img_u8, flip, taps, do_blur, mean255, std255):
        """GPU input stage with RandomGaussianBlur: uint8 [N][H][W][3] (already at the input size) -> the stem layout inside
        `workspace`: flip, (u8 - mean) / std and the per-image separable blur.  Follow with forward(..., data=None)."""
        n, h, w, c = img_u8.shape
        if img_u8.dtype != torch.uint8 or c != 3 or (n, h, w) != (self.N, self.H, self.W) or not img_u8.is_contiguous():
            raise ValueError("stage_blur: expected contiguous uint8 [%d, %d, %d, 3] frames, got %s %s"
                             % (self.N, self.H, self.W, img_u8.dtype, tuple(img_u8.shape)))
        wp, left = ctypes.c_int32(), ctypes.c_int32()
        check(lib().vince_trunk_stem_join(self._h, ops.stream_ptr()))
        x0 = lib().vince_trunk_input_ptr(self._h, ctypes.c_void_p(workspace.data_ptr()), ctypes.byref(wp), ctypes.byref(left))
        if wp.value != ops.stem_row_width(w) or left.value != ops.STEM_LEFT:
            raise RuntimeError("stage_blur: stem layout mismatch")
        ops.aug_blur_to_rows(img_u8, self.dtype, mean255, std255, flip, taps, do_blur, out=int(x0))

    def prepare_weights_folded(self, param_ptrs, bn_running_ptrs, wcache):
        """BatchNorm-folded inference weights + biases into `wcache` (a cache separate from the training one)."""
        check(lib().vince_trunk_prepare_weights_folded(self._h, param_ptrs, bn_running_ptrs,
                                                       ctypes.c_void_p(wcache.data_ptr()), ops.stream_ptr()))

    def forward_folded(self, wcache, data, workspace, pooled, perm=None, jigsaw_src=None):
        """Eval-mode forward with folded BatchNorms: convolutions only, nothing kept for backward."""
        jh, jw = (0, 0) if jigsaw_src is None else jigsaw_src
        check(lib().vince_trunk_forward_folded(
            self._h, ctypes.c_void_p(wcache.data_ptr()), None if data is None else ctypes.c_void_p(data.data_ptr()),
            None if perm is None else ctypes.c_void_p(perm.data_ptr()), jh, jw, ctypes.c_void_p(workspace.data_ptr()),
            ctypes.c_void_p(pooled.data_ptr()), ops.stream_ptr()))

    def backward(self, param_ptrs, wcache, workspace, dpooled, grad_ptrs, bucket_events=None):
        """bucket_events: optional [(block_index, torch.cuda.Event)] recorded as soon as that block's (and every later
        block's) parameter gradients are final."""
        n = 0 if not bucket_events else len(bucket_events)
        blocks = (ctypes.c_int32 * max(n, 1))()
        events = (ctypes.c_void_p * max(n, 1))()
        for i in range(n):
            blocks[i] = bucket_events[i][0]
            events[i] = bucket_events[i][1].cuda_event
        check(lib().vince_trunk_backward(self._h, param_ptrs, ctypes.c_void_p(wcache.data_ptr()),
                                         ctypes.c_void_p(workspace.data_ptr()), ctypes.c_void_p(dpooled.data_ptr()),
                                         grad_ptrs, blocks, events, n, ops.stream_ptr()))

    def set_shadow(self, twin, workspace):
        """The mixed mode "x3f": grad-enabled train-mode forwards of this fp32-tensor trunk also fill `workspace` (twin.ws_bytes) of the
        bf16 `twin` (same arch / N / H / W) with everything a backward reads; twin.backward(...) then runs the bf16 kernels
        (include/vince_hip.h vince_trunk_set_shadow).  twin None: off."""
        if twin is None:
            check(lib().vince_trunk_set_shadow(self._h, None, None))
        else:
            check(lib().vince_trunk_set_shadow(self._h, twin._h, ctypes.c_void_p(workspace.data_ptr())))

    def set_stem_event(self, event):
        """torch.cuda.Event (already recorded once, so its handle exists) or None: see include/vince_hip.h vince_trunk_set_stem_event."""
        check(lib().vince_trunk_set_stem_event(self._h, None if event is None else ctypes.c_void_p(event.cuda_event)))

    def set_bucket_callback(self, fn):
        """fn(e) is called from inside backward() right after bucket event e has been recorded (None clears it)."""
        if fn is None:
            self._bucket_cb = None
            check(lib().vince_trunk_set_bucket_callback(self._h, None, None))
            return
        self._bucket_cb = ctypes.CFUNCTYPE(None, ctypes.c_int32, ctypes.c_void_p)(lambda e, _user: fn(int(e)))   # kept alive here
        check(lib().vince_trunk_set_bucket_callback(self._h, ctypes.cast(self._bucket_cb, ctypes.c_void_p), None))

    def spatial_view(self, workspace):
        """The trunk output inside `workspace` as an [N, C, h, w] tensor with channels_last strides (zero copy)."""
        ptr = lib().vince_trunk_spatial_ptr(self._h, ctypes.c_void_p(workspace.data_ptr()))
        off = ptr - workspace.data_ptr()
        esize = 4 if self.dtype == torch.float32 else 2
        n = self.N * self.out_h * self.out_w * self.out_channels
        flat = workspace[off: off + n * esize].view(self.dtype)
        return flat.view(self.N, self.out_h, self.out_w, self.out_channels).permute(0, 3, 1, 2)


def pointer_table(tensors):
    """ctypes array of device pointers (None -> NULL)."""
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr
