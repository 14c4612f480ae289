"""Tensor-level wrappers over the C ABI (include/vince_hip.h).

Every function takes CUDA (ROCm) torch tensors, checks device / dtype / contiguity, and launches on torch's current
stream.  Nothing here computes on the CPU: a CPU tensor raises.
"""
import ctypes

import torch

from . import _lib
from ._lib import BnReduce, BnTrain, ConvDesc, ConvEpi, InfoNCEDesc, VINCE_BF16, VINCE_F32, VINCE_F32X1B, VINCE_F32X3B, VINCE_F32X3H, check, lib

EPI_ACCUMULATE, EPI_RELU, EPI_IN_HALF_PAIRS = _lib.EPI_ACCUMULATE, _lib.EPI_RELU, _lib.EPI_IN_HALF_PAIRS
STATS_REPLICAS = 16   # VINCE_STATS_REPLICAS in include/vince_hip.h


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dtype_code(t):
    if t.dtype == torch.float32:
        return VINCE_F32
    if t.dtype == torch.bfloat16:
        return VINCE_BF16
    raise TypeError("vince_amd: unsupported dtype %s (float32 or bfloat16)" % t.dtype)


def torch_dtype(code):
    return torch.float32 if code == VINCE_F32 else torch.bfloat16


def require_gpu(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("vince_amd: tensor on %s -- the HIP path needs GPU tensors (no CPU fallback)" % t.device)
        if not t.is_contiguous():
            raise RuntimeError("vince_amd: tensor must be contiguous")


# ------------------------------------------------------------------------------------------------ conv descriptors
def conv_desc(N, Hi, Wi, Ci, Co, k, stride, pad):
    """Forward descriptor of a k x k conv (also the wgrad descriptor)."""
    Ho = (Hi + 2 * pad - k) // stride + 1
    Wo = (Wi + 2 * pad - k) // stride + 1
    return ConvDesc(N=N, Hi=Hi, Wi=Wi, Ci=Ci, Ho=Ho, Wo=Wo, Co=Co, sh=stride, sw=stride, TA=k, TB=k, dh0=-pad, dhs=1,
                    dw0=-pad, dws=1, wt0=0, wta=k, wtb=1, WT=k * k, OH=Ho, OW=Wo, osh=1, osw=1, oh0=0, ow0=0)


STEM_CS, STEM_K, STEM_LEFT = 4, 32, 3   # csrc/trunk.hip stem_desc


def stem_row_width(W):
    """Padded row length of the packed-row-tap stem input for image width W (csrc/trunk.hip: sWp)."""
    Wo = (W + 6 - 7) // 2 + 1
    return (2 * Wo + 6 + 1) & ~1


def stem_desc(N, H, W, Co=64):
    """The 7x7 / stride 2 / pad 3 stem as 7 packed row taps over the [N][H][Wp][4] input layout (vince_conv_desc.Cs)."""
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    return ConvDesc(N=N, Hi=H, Wi=stem_row_width(W), Ci=STEM_K, Ho=Ho, Wo=Wo, Co=Co, sh=2, sw=2, TA=7, TB=1, dh0=-3, dhs=1,
                    dw0=0, dws=0, wt0=0, wta=1, wtb=0, WT=7, OH=Ho, OW=Wo, osh=1, osw=1, oh0=0, ow0=0, Cs=STEM_CS, Kw=7)


def input_nchw_to_rows(x, dtype, perm=None):
    """float NCHW frames -> the packed stem layout [N][H][Wp][4] (zero margins, 3 columns on the left)."""
    require_gpu(x, perm)
    N, C, H, W = x.shape
    Wp = stem_row_width(W)
    out = torch.empty(N, H, Wp, STEM_CS, device=x.device, dtype=dtype)
    check(lib().vince_input_nchw_to_rows(dtype_code(out), _ptr(x), _ptr(perm), _ptr(out), N, C, H, W, Wp, STEM_LEFT,
                                         stream_ptr()))
    return out


def dgrad_descs(N, Hi, Wi, Ci, Co, k, stride, pad):
    """Input-gradient descriptors of the same conv: one per output-pixel parity class (csrc/trunk.hip dgrad_descs)."""
    Ho = (Hi + 2 * pad - k) // stride + 1
    Wo = (Wi + 2 * pad - k) // stride + 1
    s = stride
    out = []
    for ph in range(s):
        for pw in range(s):
            r0, s0 = (ph + pad) % s, (pw + pad) % s
            TA = (k - r0 + s - 1) // s if r0 < k else 0
            TB = (k - s0 + s - 1) // s if s0 < k else 0
            gh = (Hi - ph + s - 1) // s if Hi > ph else 0
            gw = (Wi - pw + s - 1) // s if Wi > pw else 0
            if not (TA and TB and gh and gw):
                continue
            out.append(ConvDesc(N=N, Hi=Ho, Wi=Wo, Ci=Co, Ho=gh, Wo=gw, Co=Ci, sh=1, sw=1, TA=TA, TB=TB,
                                dh0=(ph + pad - r0) // s, dhs=-1, dw0=(pw + pad - s0) // s, dws=-1,
                                wt0=r0 * k + s0, wta=s * k, wtb=s, WT=k * k, OH=Hi, OW=Wi, osh=s, osw=s, oh0=ph, ow0=pw))
    return out


def linear_desc(rows, cin, cout):
    return conv_desc(rows, 1, 1, cin, cout, 1, 1, 0)


def bn_reduce_arg(y, mean, invstd, sums, mask_bits=None, mask_scale=None, mask_shift=None):
    """vince_bn_reduce for conv_igemm(bnred=...): fuse a BatchNorm-backward reduction into a dgrad epilogue."""
    require_gpu(y, mean, invstd, sums, mask_bits, mask_scale, mask_shift)
    r = BnReduce(y.data_ptr(), None if mask_bits is None else mask_bits.data_ptr(),
                 None if mask_scale is None else mask_scale.data_ptr(), None if mask_shift is None else mask_shift.data_ptr(),
                 mean.data_ptr(), invstd.data_ptr(), sums.data_ptr())
    r._keep = (y, mean, invstd, sums, mask_bits, mask_scale, mask_shift)
    return r


X3_CODE = {None: None, "h": VINCE_F32X3H, "b": VINCE_F32X3B, "1": VINCE_F32X1B}    # "1": single bfloat16 products (gradient launches of x3f)


def _conv_dtype(x, x3):
    """x3: None = the tensors' own dtype; "h" / "b" = float32 tensors multiplied as split-half products (VINCE_F32X3H / VINCE_F32X3B)."""
    if x3 is None:
        return dtype_code(x)
    if x.dtype != torch.float32:
        raise TypeError("vince_amd: split-half products take float32 tensors")
    return X3_CODE[x3]


def conv_igemm(desc, x, w, out, bias=None, stats=None, flags=0, acc_mask=None, bnred=None, replicas=0, out_scale=None,
               id_scale=None, id_shift=None, out_mask=None, in2=None, x3=None, in2_repeat=0):
    """vince_conv_igemm with the epilogue options of vince_conv_epi."""
    require_gpu(x, w, out, bias, stats, acc_mask, out_scale, id_scale, id_shift, out_mask, in2)
    e = ConvEpi()
    e.flags = flags
    e.bias = None if bias is None else bias.data_ptr()
    e.stats = None if stats is None else stats.data_ptr()
    e.acc_mask = None if acc_mask is None else acc_mask.data_ptr()
    if bnred is not None:
        e.bnred = bnred
    e.replicas = replicas
    e.out_scale = None if out_scale is None else out_scale.data_ptr()   # residual join with known BatchNorm constants
    e.id_scale = None if id_scale is None else id_scale.data_ptr()
    e.id_shift = None if id_shift is None else id_shift.data_ptr()
    e.out_mask = None if out_mask is None else out_mask.data_ptr()
    if in2 is not None:     # the last tap reads this tensor (vince_conv_epi.in2)
        e.in2, e.in2_channels, e.in2_repeat = in2.data_ptr(), in2.shape[-1], in2_repeat
    check(lib().vince_conv_igemm(ctypes.byref(desc), _conv_dtype(x, x3), _ptr(x), _ptr(w), _ptr(out), ctypes.byref(e),
                                 stream_ptr()))
    return out


def conv_expand_join(x, w, out_scale, out_shift, identity, out=None, id_scale=None, id_shift=None, relu=True, y_raw=None,
                     mask_out=None):
    """relu(out_scale * (x @ w.T) + out_shift + identity') in one streaming launch (vince_conv_expand_join); x [rows, K],
    w [Co, K], identity [rows, Co]; out defaults to in place on identity.  y_raw / mask_out: the training forward's extra
    outputs (the raw convolution output and the ReLU mask bytes)."""
    require_gpu(x, w, out_scale, out_shift, identity, out, id_scale, id_shift, y_raw, mask_out)
    out = identity if out is None else out
    rows, K = x.numel() // x.shape[-1], x.shape[-1]
    check(lib().vince_conv_expand_join(dtype_code(x), _ptr(x), _ptr(w), rows, K, w.shape[0], _ptr(out_scale), _ptr(out_shift),
                                       _ptr(identity), _ptr(id_scale), _ptr(id_shift), _ptr(out), _ptr(y_raw), _ptr(mask_out),
                                       int(relu), stream_ptr()))
    return out


def conv_expand_join_next(x, w, out_scale, out_shift, identity, w_next, y_next, out=None, id_scale=None, id_shift=None, relu=True,
                          y_raw=None, mask_out=None, stats_next=None, replicas_next=0, bias_next=None, relu_next=False):
    """conv_expand_join plus the next bottleneck's conv1 on the block output while it is on chip (vince_conv_expand_join_next):
    y_next [rows, 64] = out @ w_next.T as vince_conv_igemm stores it, its BatchNorm statistics into stats_next (float64
    [replicas, 64, 2], zeroed by the caller); bias_next / relu_next: the folded-inference epilogue instead.  K = 64, Co = 256."""
    require_gpu(x, w, out_scale, out_shift, identity, out, id_scale, id_shift, y_raw, mask_out, w_next, y_next, stats_next, bias_next)
    out = identity if out is None else out
    rows, K = x.numel() // x.shape[-1], x.shape[-1]
    check(lib().vince_conv_expand_join_next(dtype_code(x), _ptr(x), _ptr(w), rows, K, w.shape[0], _ptr(out_scale), _ptr(out_shift),
                                            _ptr(identity), _ptr(id_scale), _ptr(id_shift), _ptr(out), _ptr(y_raw), _ptr(mask_out),
                                            int(relu), _ptr(w_next), w_next.shape[0], _ptr(y_next), _ptr(stats_next), int(replicas_next),
                                            _ptr(bias_next), int(relu_next), stream_ptr()))
    return out, y_next


def conv_expand_stats(x, w, out, stats=None, replicas=0):
    """out = x @ w.T (bf16, K = 64 / 128, Co multiple of 256) through the streaming kernel, BatchNorm statistics of the stored
    values into stats (double[R][Co][2], zeroed by the caller)."""
    require_gpu(x, w, out, stats)
    rows, K = x.numel() // x.shape[-1], x.shape[-1]
    check(lib().vince_conv_expand_stats(dtype_code(x), _ptr(x), _ptr(w), rows, K, w.shape[0], _ptr(out), _ptr(stats), replicas,
                                        stream_ptr()))
    return out


def conv3x3_strip_dgrad(dy, wt, dx, bnred=None, replicas=0):
    """dx = the input gradient of layer1's 3x3 (dy, dx [N, H, 56, 64] bf16, wt [64, 9, 64] = the prepared [Ci][tap][Co] copy) through the
    image-strip kernel, with the fused BatchNorm-backward reduction of vince_conv_igemm's gradient epilogue (bnred: bn_reduce_arg(...,
    mask_scale=, mask_shift=))."""
    require_gpu(dy, wt, dx)
    N, H, W, C = dy.shape
    check(lib().vince_conv3x3_strip_dgrad(dtype_code(dy), _ptr(dy), _ptr(wt), N, H, W, C, _ptr(dx), None if bnred is None else ctypes.byref(bnred),
                                          replicas, stream_ptr()))
    return dx


def conv3x3_strip(x, w, out, stats=None, replicas=0, tap_map=None):
    """out = conv3x3(x, w) for layer1's shape (x, out [N, H, 56, 64] bf16, w [64, 9, 64]) through the image-strip kernel, BatchNorm
    statistics of the stored values into stats (double[R][64][2], zeroed by the caller)."""
    require_gpu(x, w, out, stats)
    N, H, W, Ci = x.shape
    tm = None
    if tap_map is not None:
        tm = (ctypes.c_int32 * 9)(*[int(v) for v in tap_map])
    check(lib().vince_conv3x3_strip(dtype_code(x), _ptr(x), _ptr(w), N, H, W, Ci, w.shape[0], tm, _ptr(out), _ptr(stats), replicas,
                                    stream_ptr()))
    return out


def conv3x3_strip_bias(x, w, out, bias=None, relu=True):
    """out = [relu](conv3x3(x, w) + bias) for layer1's shape through the image-strip kernel: the epilogue of the BatchNorm-folded
    inference forward (vince_conv3x3_strip_bias)."""
    require_gpu(x, w, out, bias)
    N, H, W, Ci = x.shape
    check(lib().vince_conv3x3_strip_bias(dtype_code(x), _ptr(x), _ptr(w), N, H, W, Ci, w.shape[0], _ptr(bias), int(relu), _ptr(out),
                                         stream_ptr()))
    return out


def conv_expand_dgrad(dy, wt, out, accumulate=False, acc_mask=None, bnred=None, replicas=0):
    """out (+)= dy @ wt.T through the streaming kernel with the gradient epilogues (vince_conv_expand_dgrad); dy [rows, K],
    wt [Co, K], out [rows, Co] in place."""
    require_gpu(dy, wt, out, acc_mask)
    rows, K = dy.numel() // dy.shape[-1], dy.shape[-1]
    check(lib().vince_conv_expand_dgrad(dtype_code(dy), _ptr(dy), _ptr(wt), rows, K, wt.shape[0], _ptr(out), int(accumulate),
                                        _ptr(acc_mask), ctypes.byref(bnred) if bnred is not None else None, replicas, stream_ptr()))
    return out


def conv_expand_dgrad_masked(dy, wt, out, out_mask, gsums, accumulate=False, acc_mask=None, replicas=0):
    """vince_conv_expand_dgrad_masked: out = (dy @ wt.T + old [gated by acc_mask]) gated by out_mask; gsums += per-channel (sum, sum sq)."""
    require_gpu(dy, wt, out, acc_mask, out_mask, gsums)
    rows, K = dy.numel() // dy.shape[-1], dy.shape[-1]
    check(lib().vince_conv_expand_dgrad_masked(dtype_code(dy), _ptr(dy), _ptr(wt), rows, K, wt.shape[0], _ptr(out), int(accumulate),
                                               _ptr(acc_mask), _ptr(out_mask), _ptr(gsums), replicas, stream_ptr()))
    return out


def bn3_bwd_prepare(R, w, gsums, mean, invstd, gamma, count, dgamma, dbeta, colsum=None, split=False):
    """vince_bn3_bwd_prepare (csrc/bn_algebra.hip): R float[Co][K] = g^T a, w bf16 [Co][K], gsums double[replicas][Co][2].
    Returns coef float[5][Co] (row 4: the mean the formulas used -- what bn3_bwd_finish_dw takes), w2 bf16 [K][2][Co] -- tap 0 = wd =
    W^T diag(s), tap 1 = nq = -W^T diag(t) W in its first K entries: the weights of the one-launch input gradient da = wd g + nq a + nr
    (conv_igemm(..., in2=a)) -- and nr float[K]; dgamma / dbeta are accumulated in place.  colsum (double[replicas][K], column sums of
    a): the self-consistent form (implied mean, nr from the rounded matrices) the engine uses.  w: bf16, or the fp32 master weights.
    split=True: w2 is [K][3][Co] -- taps wd_hi, wd_lo, and [nq_hi | nq_lo] in the first 2K entries of the third (conv_igemm(..., in2=a,
    in2_repeat=2)); split="nq": [K][2][Co] with nq alone in two parts, [nq_hi | nq_lo] in the second tap -- the mixed mode's form."""
    require_gpu(R, w, gsums, mean, invstd, gamma, dgamma, dbeta, colsum)
    Co, K = w.shape[0], w.shape[-1]
    dev = w.device
    taps = 3 if split is True else 2
    coef = torch.empty(5, Co, device=dev, dtype=torch.float32)
    w2 = torch.zeros(K, taps, Co, device=dev, dtype=torch.bfloat16)
    nr = torch.empty(K, device=dev, dtype=torch.float32)
    base, ld = w2.data_ptr(), taps * Co
    check(lib().vince_bn3_bwd_prepare(_ptr(R), _ptr(w), _ptr(gsums), gsums.shape[0], _ptr(mean), _ptr(invstd), _ptr(gamma), int(count),
                                      Co, K, _ptr(coef), base, ld, base + 2 * (taps - 1) * Co, ld, _ptr(nr), _ptr(dgamma),
                                      _ptr(dbeta), _ptr(colsum), 0 if colsum is None else colsum.shape[0],
                                      VINCE_F32 if w.dtype == torch.float32 else VINCE_BF16,
                                      base + 2 * Co if split is True else None, base + 2 * ((taps - 1) * Co + K) if split else None, stream_ptr()))
    return coef, w2, nr


def bn3_bwd_finish_dw(RdW, w, gram, colsum, coef, mean, invstd, dw_accum=None):
    """vince_bn3_bwd_finish_dw: the raw weight gradient R becomes the weight gradient -- in place, or (dw_accum given) ADDED into
    dw_accum with R left untouched."""
    require_gpu(RdW, w, gram, colsum, coef, mean, invstd, dw_accum)
    Co, K = w.shape[0], w.shape[-1]
    check(lib().vince_bn3_bwd_finish_dw(_ptr(RdW), _ptr(dw_accum), _ptr(w), _ptr(gram), _ptr(colsum), colsum.shape[0], _ptr(coef), _ptr(mean),
                                        _ptr(invstd), Co, K, VINCE_F32 if w.dtype == torch.float32 else VINCE_BF16,
                                        stream_ptr()))
    return RdW if dw_accum is None else dw_accum


def conv_wgrad(desc, x, dy, dw, ci_dw=None, variant=0, x3=None):
    require_gpu(x, dy, dw)
    if dw.dtype != torch.float32:
        raise TypeError("vince_amd: weight gradients are float32")
    check(lib().vince_conv_wgrad(ctypes.byref(desc), _conv_dtype(x, x3), _ptr(x), _ptr(dy), _ptr(dw),
                                 desc.Ci if ci_dw is None else ci_dw, variant, stream_ptr()))
    return dw


def conv_wgrad_det(desc, x, dy, dw, ci_dw=None, scratch=None, x3=None):
    """vince_conv_wgrad_det: the reproducible weight gradient (per-split slabs + a fixed-order reduction instead of fp32 atomics)."""
    require_gpu(x, dy, dw, scratch)
    ci = desc.Ci if ci_dw is None else ci_dw
    code = _conv_dtype(x, x3)
    if scratch is None:
        need = lib().vince_conv_wgrad_scratch_bytes(ctypes.byref(desc), code, ci)
        scratch = torch.empty(max(need, 16), dtype=torch.uint8, device=x.device)
    check(lib().vince_conv_wgrad_det(ctypes.byref(desc), code, _ptr(x), _ptr(dy), _ptr(dw), ci, _ptr(scratch), scratch.numel(),
                                     stream_ptr()))
    return dw


# ------------------------------------------------------------------------------------------------ linear layer helpers (fp32)
def linear_fwd(x, weight, bias, relu=False, x3=False):
    """y = [relu](x @ weight.T + bias) (vince_model.py:38-42) on the fp32 MFMA path, or -- x3 = True / "h" / "b": `weight` is the
    split-half copy prepare_weight(..., x3=...) returned (IEEE half pairs / bfloat16 pairs for "b") -- as split-half products."""
    rows, cin = x.shape
    cout = weight.shape[0]
    out = torch.empty(rows, cout, device=x.device, dtype=torch.float32)
    conv_igemm(linear_desc(rows, cin, cout), x, weight, out, bias=bias, flags=EPI_RELU if relu else 0,
               x3=(x3 if x3 in ("h", "b") else "h") if x3 else None)
    return out


def nonfinite_latch(value, step, latch):
    """Device-side stand-in for the reference's per-iteration `assert torch.isfinite(total_loss)` (solvers/vince_solver.py:446-452):
    latch (int64[2], zeroed once) counts non-finite values and keeps the first offending step + 1; no host synchronisation."""
    require_gpu(value, latch)
    assert value.dtype == torch.float32 and value.numel() == 1 and latch.dtype == torch.int64 and latch.numel() == 2
    check(lib().vince_nonfinite_latch(_ptr(value), int(step), _ptr(latch), stream_ptr()))


def linear_bwd(x, weight_t, dy, dweight, dbias, need_dx=True, x3=False):
    """dweight += dy.T @ x ; dbias += dy.sum(0) ; returns dx = dy @ weight (weight_t = weight.T contiguous; x3: the transposed
    split-half copy of prepare_weight(..., x3=True) -- bfloat16 pairs -- and both gradient GEMMs as split-half products)."""
    rows, cin = x.shape
    cout = dy.shape[1]
    conv_wgrad(linear_desc(rows, cin, cout), x, dy, dweight, x3="b" if x3 else None)
    require_gpu(dbias)
    check(lib().vince_colsum(_ptr(dy), _ptr(dbias), rows, cout, stream_ptr()))
    if not need_dx:
        return None
    dx = torch.empty(rows, cin, device=x.device, dtype=torch.float32)
    conv_igemm(linear_desc(rows, cout, cin), dy, weight_t, dx, x3="b" if x3 else None)
    return dx


def relu_bwd(dout, act):
    require_gpu(dout, act)
    dx = torch.empty_like(dout)
    check(lib().vince_relu_bwd(_ptr(dout), _ptr(act), _ptr(dx), dout.numel(), stream_ptr()))
    return dx


def l2norm_fwd(x, eps=1e-12):
    require_gpu(x)
    out = torch.empty_like(x)
    norms = torch.empty(x.shape[0], device=x.device, dtype=torch.float32)
    check(lib().vince_l2norm_fwd(_ptr(x), _ptr(out), _ptr(norms), x.shape[0], x.shape[1], eps, stream_ptr()))
    return out, norms


def l2norm_bwd(x, norms, dout, eps=1e-12):
    require_gpu(x, norms, dout)
    dx = torch.empty_like(x)
    check(lib().vince_l2norm_bwd(_ptr(x), _ptr(norms), _ptr(dout), _ptr(dx), x.shape[0], x.shape[1], eps, stream_ptr()))
    return dx


# ------------------------------------------------------------------------------------------------ batch norm / pooling
def bn_finalize(stats, count, gamma, beta, running_mean, running_var, nbt, train, momentum=0.1, eps=1e-5):
    C = gamma.numel()
    consts = torch.empty(4, C, device=gamma.device, dtype=torch.float32)
    require_gpu(stats, gamma, beta, running_mean, running_var, nbt)
    check(lib().vince_bn_finalize(_ptr(stats), count, C, _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                  _ptr(nbt), momentum, eps, int(train), _ptr(consts[0]), _ptr(consts[1]), _ptr(consts[2]),
                                  _ptr(consts[3]), stream_ptr()))
    return consts  # scale, shift, mean, invstd


def bn_apply(y, scale, shift, identity=None, id_scale=None, id_shift=None, relu=True, want_mask=False):
    require_gpu(y, scale, shift, identity, id_scale, id_shift)
    out = torch.empty_like(y)
    C = y.shape[-1]
    ch = 4 if y.dtype == torch.float32 else 8
    mask = torch.empty(y.numel() // ch, device=y.device, dtype=torch.uint8) if want_mask else None
    check(lib().vince_bn_apply(dtype_code(y), _ptr(y), _ptr(scale), _ptr(shift), _ptr(identity), _ptr(id_scale),
                               _ptr(id_shift), _ptr(out), _ptr(mask), y.numel() // C, C, int(relu), stream_ptr()))
    return (out, mask) if want_mask else out


def bn_train_apply(y, stats, count, gamma, beta, running_mean=None, running_var=None, nbt=None, identity=None, id_scale=None,
                   id_shift=None, relu=True, want_mask=False, replicas=0, momentum=0.1, eps=1e-5, out_sum=None, half_pairs=False):
    """Train-mode finalize + apply in one launch.  Returns (out, mask or None, scale, shift, mean, invstd).
    out_sum: optional zeroed double[R][C] -- per-channel sums of the stored output are accumulated into it."""
    require_gpu(y, stats, gamma, beta, running_mean, running_var, nbt, identity, id_scale, id_shift, out_sum)
    C = y.shape[-1]
    ch = 4 if y.dtype == torch.float32 else 8
    out = torch.empty_like(y)
    mask = torch.empty(y.numel() // ch, device=y.device, dtype=torch.uint8) if want_mask else None
    scale, shift, mean, invstd = (torch.empty(C, device=y.device) for _ in range(4))
    bt = BnTrain()
    bt.stats, bt.replicas, bt.count = stats.data_ptr(), replicas, count
    bt.gamma, bt.beta = gamma.data_ptr(), beta.data_ptr()
    bt.running_mean = None if running_mean is None else running_mean.data_ptr()
    bt.running_var = None if running_var is None else running_var.data_ptr()
    bt.num_batches_tracked = None if nbt is None else nbt.data_ptr()
    bt.momentum, bt.eps = momentum, eps
    bt.scale, bt.shift, bt.save_mean, bt.save_invstd = scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr()
    if out_sum is not None:
        bt.out_sum, bt.out_sum_replicas = out_sum.data_ptr(), out_sum.shape[0]
    bt.out_half_pairs = int(bool(half_pairs))     # `out` as stored IEEE-half pairs (conv_igemm(..., flags=EPI_IN_HALF_PAIRS, x3="h") reads them)
    check(lib().vince_bn_train_apply(dtype_code(y), _ptr(y), ctypes.byref(bt), _ptr(identity), _ptr(id_scale), _ptr(id_shift),
                                     _ptr(out), _ptr(mask), y.numel() // C, C, int(relu), stream_ptr()))
    return out, mask, scale, shift, mean, invstd


def bn_train_apply_gram(y, stats, count, gamma, beta, gram, out_sum, running_mean=None, running_var=None, nbt=None, replicas=0,
                        momentum=0.1, eps=1e-5):
    """Train-mode BatchNorm + ReLU of a bf16 [rows][C] tensor (C = 64 / 128) that also accumulates the Gram matrix of what it stores
    into `gram` (zeroed float[C][C]) and its column sums into `out_sum` (zeroed double[R][C]): vince_bn_train_apply_gram.
    Returns (out, scale, shift, mean, invstd)."""
    require_gpu(y, stats, gamma, beta, gram, out_sum, running_mean, running_var, nbt)
    C = y.shape[-1]
    rows = y.numel() // C
    out = torch.empty_like(y)
    scale, shift, mean, invstd = (torch.empty(C, device=y.device) for _ in range(4))
    bt = BnTrain()
    bt.stats, bt.replicas, bt.count = stats.data_ptr(), replicas, count
    bt.gamma, bt.beta = gamma.data_ptr(), beta.data_ptr()
    bt.running_mean = None if running_mean is None else running_mean.data_ptr()
    bt.running_var = None if running_var is None else running_var.data_ptr()
    bt.num_batches_tracked = None if nbt is None else nbt.data_ptr()
    bt.momentum, bt.eps = momentum, eps
    bt.scale, bt.shift, bt.save_mean, bt.save_invstd = scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr()
    bt.out_sum, bt.out_sum_replicas = out_sum.data_ptr(), out_sum.shape[0]
    need = lib().vince_bn_train_apply_gram_scratch_bytes(rows, C)
    scratch = torch.empty(max(int(need), 16), dtype=torch.uint8, device=y.device)
    check(lib().vince_bn_train_apply_gram(dtype_code(y), _ptr(y), ctypes.byref(bt), _ptr(out), rows, C, _ptr(gram), _ptr(scratch),
                                          scratch.numel(), stream_ptr()))
    return out, scale, shift, mean, invstd


def bn_gram_finalize(gram, colsum, count, w, gamma, beta, running_mean=None, running_var=None, nbt=None, momentum=0.1, eps=1e-5):
    """BatchNorm constants of the 1x1 conv y = w a from the Gram matrix of a (vince_bn_gram_finalize).
    gram float[K][K], colsum double[R][K], w [Co][K] in the compute dtype.  Returns consts [4][Co]: scale, shift, mean, invstd."""
    require_gpu(gram, colsum, w, gamma, beta, running_mean, running_var, nbt)
    Co, K = w.shape[0], w.shape[-1]
    consts = torch.empty(4, Co, device=w.device, dtype=torch.float32)
    check(lib().vince_bn_gram_finalize(dtype_code(w), _ptr(gram), _ptr(colsum), colsum.shape[0], count, _ptr(w), K, Co,
                                       _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), _ptr(nbt), momentum, eps,
                                       _ptr(consts[0]), _ptr(consts[1]), _ptr(consts[2]), _ptr(consts[3]), stream_ptr()))
    return consts


def bn_bwd_reduce(dz, y, mean, invstd, mask_src=None, mask_bits=None, mask_scale=None, mask_shift=None, replicas=0):
    """(sum g, sum g*xhat) per channel, replicas folded: the stand-alone pass the fused dgrad epilogue replaces."""
    require_gpu(dz, y, mean, invstd, mask_src, mask_bits, mask_scale, mask_shift)
    C = y.shape[-1]
    rows = y.numel() // C
    sums = torch.zeros(STATS_REPLICAS, C, 2, device=y.device, dtype=torch.float64)
    check(lib().vince_bn_bwd_reduce(dtype_code(y), _ptr(dz), _ptr(mask_src), _ptr(mask_bits), _ptr(mask_scale),
                                    _ptr(mask_shift), _ptr(y), _ptr(mean), _ptr(invstd), _ptr(sums), rows, C, replicas,
                                    stream_ptr()))
    return sums.sum(0)


def bn_bwd(dz, mask_src, y, mean, invstd, gamma, dgamma, dbeta, want_g=False, mask_bits=None, mask_scale=None,
           mask_shift=None, replicas=0, second=None):
    """second: optional ctypes pointer to a vince_bn_reduce2 (another BatchNorm reduced over the same masked gradient)."""
    require_gpu(dz, mask_src, y, mean, invstd, gamma, dgamma, dbeta, mask_bits, mask_scale, mask_shift)
    C = y.shape[-1]
    rows = y.numel() // C
    sums = torch.zeros(STATS_REPLICAS, C, 2, device=y.device, dtype=torch.float64)
    check(lib().vince_bn_bwd_reduce(dtype_code(y), _ptr(dz), _ptr(mask_src), _ptr(mask_bits), _ptr(mask_scale),
                                    _ptr(mask_shift), _ptr(y), _ptr(mean), _ptr(invstd), _ptr(sums), rows, C, replicas,
                                    stream_ptr()))
    dy = torch.empty_like(y)
    g = torch.empty_like(y) if want_g else None
    check(lib().vince_bn_bwd_apply(dtype_code(y), _ptr(dz), _ptr(mask_src), _ptr(mask_bits), _ptr(mask_scale),
                                   _ptr(mask_shift), _ptr(y), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(sums), rows,
                                   _ptr(dy), _ptr(g), _ptr(dgamma), _ptr(dbeta), rows, C, replicas, second, stream_ptr()))
    return dy, g


def stem_pool_fwd(y, scale, shift):
    require_gpu(y, scale, shift)
    N, H, W, C = y.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    out = torch.empty(N, Ho, Wo, C, device=y.device, dtype=y.dtype)
    amax = torch.empty(N, Ho, Wo, C, device=y.device, dtype=torch.uint8)
    check(lib().vince_stem_pool_fwd(dtype_code(y), _ptr(y), _ptr(scale), _ptr(shift), _ptr(out), _ptr(amax), N, H, W, C,
                                    stream_ptr()))
    return out, amax


def stem_bwd(dpool, amax, y, mean, invstd, gamma, dgamma, dbeta):
    """Fused stem backward (max-pool gather + bn1 backward): returns dy wrt the stem conv output y [N][H][W][C]."""
    require_gpu(dpool, amax, y, mean, invstd, gamma, dgamma, dbeta)
    N, H, W, C = y.shape
    sums = torch.zeros(STATS_REPLICAS, C, 2, device=y.device, dtype=torch.float64)
    dy = torch.empty_like(y)
    check(lib().vince_stem_bwd_reduce(dtype_code(y), _ptr(dpool), _ptr(amax), _ptr(y), _ptr(mean), _ptr(invstd), _ptr(sums),
                                      N, H, W, C, stream_ptr()))
    check(lib().vince_stem_bwd_apply(dtype_code(y), _ptr(dpool), _ptr(amax), _ptr(y), _ptr(mean), _ptr(invstd), _ptr(gamma),
                                     _ptr(sums), _ptr(dy), _ptr(dgamma), _ptr(dbeta), N, H, W, C, stream_ptr()))
    return dy


def stem_pool_bwd(dpool, amax, H, W):
    require_gpu(dpool, amax)
    N, _, _, C = dpool.shape
    g = torch.empty(N, H, W, C, device=dpool.device, dtype=dpool.dtype)
    check(lib().vince_stem_pool_bwd(dtype_code(dpool), _ptr(dpool), _ptr(amax), _ptr(g), N, H, W, C, stream_ptr()))
    return g


def avgpool_fwd(x):
    require_gpu(x)
    N, H, W, C = x.shape
    out = torch.empty(N, C, device=x.device, dtype=torch.float32)
    check(lib().vince_avgpool_fwd(dtype_code(x), _ptr(x), _ptr(out), N, H * W, C, stream_ptr()))
    return out


def avgpool_bwd(dout, H, W, dtype):
    require_gpu(dout)
    N, C = dout.shape
    dx = torch.empty(N, H, W, C, device=dout.device, dtype=dtype)
    check(lib().vince_avgpool_bwd(dtype_code(dx), _ptr(dout), _ptr(dx), N, H * W, C, stream_ptr()))
    return dx


# ------------------------------------------------------------------------------------------------ layouts
def input_nchw_to_nhwc(x, dtype, perm=None):
    require_gpu(x, perm)
    N, C, H, W = x.shape
    Cp = 4 if dtype == torch.float32 else 8
    out = torch.empty(N, H, W, Cp, device=x.device, dtype=dtype)
    check(lib().vince_input_nchw_to_nhwc(dtype_code(out), _ptr(x), _ptr(perm), _ptr(out), N, C, H, W, Cp, stream_ptr()))
    return out


def jigsaw_nchw_to_nhwc(x, dtype):
    require_gpu(x)
    N, C, H, W = x.shape
    # vince_model.py:145-146 pads BOTH axes by 3 - (size % 3) whenever either needs it
    if H % 3 != 0 or W % 3 != 0:
        Hp, Wp = H + 3 - H % 3, W + 3 - W % 3
    else:
        Hp, Wp = H, W
    th, tw = Hp // 3, Wp // 3
    Cp = 4 if dtype == torch.float32 else 8
    out = torch.empty(N * 9, th, tw, Cp, device=x.device, dtype=dtype)
    check(lib().vince_jigsaw_nchw_to_nhwc(dtype_code(out), _ptr(x), _ptr(out), N, C, H, W, th, tw, Cp, stream_ptr()))
    return out


def jigsaw_nchw_to_rows(x, dtype):
    """Jigsaw tiling (vince_model.py:144-155) straight into the packed stem layout [9N][th][Wp][4]."""
    require_gpu(x)
    N, C, H, W = x.shape
    if H % 3 != 0 or W % 3 != 0:
        Hp, Wq = H + 3 - H % 3, W + 3 - W % 3
    else:
        Hp, Wq = H, W
    th, tw = Hp // 3, Wq // 3
    Wp = stem_row_width(tw)
    out = torch.empty(N * 9, th, Wp, STEM_CS, device=x.device, dtype=dtype)
    check(lib().vince_jigsaw_nchw_to_rows(dtype_code(out), _ptr(x), _ptr(out), N, C, H, W, th, tw, Wp, STEM_LEFT, stream_ptr()))
    return out


def prepare_weight(w_master, dtype, cip=None, want_transposed=True, x3=False):
    """w_master: float32 [Co][T][Ci] contiguous.  Returns (wk [Co][T][Cip], wt [Ci][T][Co] or None).
    x3: the split-half weight layout of VINCE_F32X3 (float32-sized tensors holding hi / lo half pairs: wk IEEE half pairs for forward
    launches, wt bfloat16 pairs for gradient launches) -- what conv_igemm(..., x3="h" / "b") takes as its weight operand."""
    require_gpu(w_master)
    Co, T, Ci = w_master.shape
    cip = Ci if cip is None else cip
    if x3 and dtype != torch.float32:
        raise TypeError("vince_amd: the split-half weight layout lives in float32-sized tensors")
    wk = torch.empty(Co, T, cip, device=w_master.device, dtype=dtype)
    wt = torch.empty(Ci, T, Co, device=w_master.device, dtype=dtype) if want_transposed else None
    # x3 = "b": the forward copy as bfloat16 pairs too (conv_igemm(..., x3="b") on it: fp32's exponent range, 2^-16 per product)
    code = (VINCE_F32X3B if x3 == "b" else VINCE_F32X3H) if x3 else dtype_code(wk)
    check(lib().vince_prepare_weight(code, _ptr(w_master), _ptr(wk), _ptr(wt), Co, T, Ci, cip, stream_ptr()))
    return wk, wt


def stream_copy(dst, src):
    """dst <- src (same byte count, a multiple of 16; both dense) through the library's block-contiguous non-temporal copy kernel
    (vince_stream_copy: ~6 TB/s where the runtime's device-to-device blit, which torch's clone() of a dense tensor turns into, moves
    0.8 TB/s in 2 MB pieces)."""
    require_gpu(dst, src)
    nbytes = src.numel() * src.element_size()
    if nbytes != dst.numel() * dst.element_size() or nbytes % 16:
        raise ValueError("vince_amd: stream_copy needs equal byte counts, a multiple of 16")
    check(lib().vince_stream_copy(_ptr(dst), _ptr(src), nbytes, 0, 5, stream_ptr()))
    return dst


def transpose_f32(x):
    """fp32 [R, C] -> [C, R] (LDS-tiled)."""
    require_gpu(x)
    r, c = x.shape
    out = torch.empty(c, r, device=x.device, dtype=torch.float32)
    check(lib().vince_transpose_f32(_ptr(x.contiguous()), _ptr(out), r, c, stream_ptr()))
    return out


def nhwc_to_nchw_f32(x):
    require_gpu(x)
    N, H, W, C = x.shape
    out = torch.empty(N, C, H, W, device=x.device, dtype=torch.float32)
    check(lib().vince_nhwc_to_nchw_f32(dtype_code(x), _ptr(x), _ptr(out), N, C, H, W, stream_ptr()))
    return out


# ------------------------------------------------------------------------------------------------ InfoNCE
class InfoNCEResult:
    __slots__ = ("desc", "pos", "row_max", "neg_sum", "dists", "softmax_weights", "scalars", "logits")


def infonce_fwd(q, inb, queue, temperature, frames=1, offdiag_neg=False, save_logits=False):
    """Fused similarity + loss + metrics.  q:[B,D] inb:[B,D] queue:[K,D] or None (all float32).
    PRECONDITION: the rows are UNIT vectors (the reference normalises both sides, vince_model.py:180).  The logits are products of
    IEEE-half hi / lo halves scaled by 2^8: an entry with |x| >= 255.9 makes its row NaN by design (include/vince_hip.h).
    save_logits: also store the B x (Bk + K) raw cosines (r.logits) for infonce_bwd to read back instead of recomputing them."""
    require_gpu(q, inb, queue)
    B, D = q.shape
    K = 0 if queue is None else queue.shape[0]
    d = InfoNCEDesc(B=B, D=D, Bk=inb.shape[0], K=K, frames=frames, offdiag_neg=int(offdiag_neg),
                    inv_temperature=1.0 / temperature)
    nbytes = lib().vince_infonce_workspace_bytes(ctypes.byref(d))
    if nbytes == 0:
        raise RuntimeError("libvince_hip: %s" % lib().vince_last_error().decode())
    ws = torch.empty(nbytes, device=q.device, dtype=torch.uint8)
    r = InfoNCEResult()
    r.desc = d
    r.pos = torch.empty(B, frames, device=q.device)
    r.row_max = torch.empty(B, device=q.device)
    r.neg_sum = torch.empty(B, device=q.device)
    r.dists = torch.empty(B, frames, device=q.device)
    r.softmax_weights = torch.empty(B, frames, device=q.device)
    r.scalars = torch.empty(8, device=q.device)
    r.logits = torch.empty(B, inb.shape[0] + K, device=q.device) if save_logits else None
    check(lib().vince_infonce_fwd(ctypes.byref(d), _ptr(q), _ptr(inb), _ptr(queue), _ptr(r.pos), _ptr(r.row_max),
                                  _ptr(r.neg_sum), _ptr(r.dists), _ptr(r.softmax_weights), _ptr(r.scalars), _ptr(r.logits),
                                  _ptr(ws), stream_ptr()))
    return r


def infonce_bwd(r, q, inb, queue, grad_scale, dq, want_wmat=False):
    require_gpu(q, inb, queue, grad_scale, dq)
    wmat = torch.zeros(r.desc.B, r.desc.Bk, device=q.device) if want_wmat else None
    check(lib().vince_infonce_bwd(ctypes.byref(r.desc), _ptr(q), _ptr(inb), _ptr(queue), _ptr(r.pos), _ptr(r.row_max),
                                  _ptr(r.neg_sum), _ptr(grad_scale), _ptr(getattr(r, "logits", None)), _ptr(dq), _ptr(wmat),
                                  stream_ptr()))
    return wmat


# ------------------------------------------------------------------------------------------------ queue / EMA / SGD
def queue_enqueue(queue, items, tail, full):
    """Returns (new_tail, new_full).  Integer arithmetic is done in C (utils/storage_queue.py:31-49)."""
    require_gpu(queue, items)
    if items.dtype != queue.dtype or queue.dtype != torch.float32:
        raise TypeError("vince_amd: queue and items must be float32")
    t = ctypes.c_int64(tail)
    f = ctypes.c_int32(int(full))
    check(lib().vince_queue_enqueue(_ptr(queue), queue.shape[0], queue.shape[1], _ptr(items), items.shape[0],
                                    ctypes.byref(t), ctypes.byref(f), stream_ptr()))
    return t.value, bool(f.value)


def ema_flat(key, query, momentum):
    require_gpu(key, query)
    check(lib().vince_ema_flat(_ptr(key), _ptr(query), key.numel(), momentum, stream_ptr()))


def sgd_flat(param, grad, buf, lr, momentum=0.9, weight_decay=1e-4, grad_scale=1.0):
    require_gpu(param, grad, buf)
    check(lib().vince_sgd_flat(_ptr(param), _ptr(grad), _ptr(buf), param.numel(), lr, momentum, weight_decay, grad_scale,
                               stream_ptr()))


# ------------------------------------------------------------------------------------------------ GPU input stage (csrc/augment.hip)
def aug_resized_crop_u8(frames, box, size, src_index=None):
    """uint8 [Ns, Hs, Ws, 3] frames + int32 [N, 4] windows (top, left, height, width) -> uint8 [N, H, W, 3]: crop, then
    Pillow's BILINEAR resize, bit for bit."""
    require_gpu(frames, box, src_index)
    ns, hs, ws, c = frames.shape
    n = box.shape[0]
    if frames.dtype != torch.uint8 or c != 3 or not frames.is_contiguous() or box.dtype != torch.int32 or box.shape[1] != 4:
        raise ValueError("aug_resized_crop_u8: expected contiguous uint8 [Ns, Hs, Ws, 3] frames and int32 [N, 4] boxes")
    if src_index is None and n != ns:
        raise ValueError("aug_resized_crop_u8: %d boxes for %d frames (pass src_index)" % (n, ns))
    h, w = int(size[0]), int(size[1])
    tmp = torch.empty(n, hs, w, 3, dtype=torch.uint8, device=frames.device)
    out = torch.empty(n, h, w, 3, dtype=torch.uint8, device=frames.device)
    table = torch.empty(int(lib().vince_aug_resample_table_ints(n, hs, ws, h, w)), dtype=torch.int32, device=frames.device)
    check(lib().vince_aug_resized_crop_u8(_ptr(frames), _ptr(src_index), _ptr(box.contiguous()), _ptr(table), _ptr(tmp), _ptr(out),
                                          n, hs, ws, h, w, stream_ptr()))
    return out


def aug_color_u8(img, op, factor):
    """In-place ColorJitter / RandomGrayscale chain on uint8 [N, H, W, 3]; op int32 [N, M], factor float32 [N, M]."""
    require_gpu(img, op, factor)
    n, h, w, c = img.shape
    if img.dtype != torch.uint8 or c != 3 or not img.is_contiguous() or op.dtype != torch.int32 or factor.dtype != torch.float32 \
            or op.shape != factor.shape or op.shape[0] != n:
        raise ValueError("aug_color_u8: expected contiguous uint8 [N, H, W, 3], int32 [N, M] ops and float32 [N, M] factors")
    check(lib().vince_aug_color_u8(_ptr(img), _ptr(op.contiguous()), _ptr(factor.contiguous()), op.shape[1], n, h, w,
                                   stream_ptr()))
    return img


def aug_blur_to_rows(img, dtype, mean255, std255, flip=None, kernels=None, do_blur=None, out=None):
    """uint8 [N, H, W, 3] -> the packed stem layout [N][H][Wp][4] with flip, (u8 - mean) / std and the optional per-image
    separable Gaussian blur (kernels float32 [N, ks], do_blur uint8 [N])."""
    require_gpu(img, flip, kernels, do_blur)
    n, h, w, _ = img.shape
    wp = stem_row_width(w)
    if out is None:
        out = torch.empty(n, h, wp, STEM_CS, device=img.device, dtype=dtype)
    tmp = torch.empty(n, h, w, 4, device=img.device, dtype=torch.float32)
    mean = (ctypes.c_float * 3)(*[float(v) for v in mean255])
    std = (ctypes.c_float * 3)(*[float(v) for v in std255])
    ks = 0 if kernels is None else int(kernels.shape[1])
    code = VINCE_F32 if dtype == torch.float32 else VINCE_BF16
    out_ptr = ctypes.c_void_p(out) if isinstance(out, int) else _ptr(out)
    check(lib().vince_aug_blur_to_rows(code, _ptr(img), _ptr(flip), _ptr(kernels), _ptr(do_blur), ks, mean, std, _ptr(tmp),
                                       out_ptr, n, h, w, wp, STEM_LEFT, stream_ptr()))
    return out


def input_u8hwc_to_rows(frames, dtype, size, mean255, std255, crop_yx=None, flip=None, perm=None):
    """uint8 [N, Hs, Ws, 3] -> packed stem layout [N][H][Wp][4]: crop window, flip, (u8 - mean) / std (csrc/misc.hip)."""
    require_gpu(frames, crop_yx, flip, perm)
    n, hs, ws, _ = frames.shape
    h, w = int(size[0]), int(size[1])
    wp = stem_row_width(w)
    out = torch.empty(n, h, wp, STEM_CS, device=frames.device, dtype=dtype)
    mean = (ctypes.c_float * 3)(*[float(v) for v in mean255])
    std = (ctypes.c_float * 3)(*[float(v) for v in std255])
    code = VINCE_F32 if dtype == torch.float32 else VINCE_BF16
    check(lib().vince_input_u8hwc_to_rows(code, _ptr(frames), _ptr(perm), _ptr(crop_yx), _ptr(flip), mean, std, _ptr(out), n, hs,
                                          ws, h, w, wp, STEM_LEFT, stream_ptr()))
    return out
