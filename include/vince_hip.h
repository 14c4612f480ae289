/* vince_hip.h -- flat C ABI of libvince_hip.so: the MI355X (gfx950) kernels behind the VINCE
 * encoder + contrastive hot path.
 *
 * The reference (danielgordon10/vince) has no FFI of its own: every "kernel" is a torch op call
 * site inside models/vince_model.py, models/building_blocks/resnet.py, utils/loss_util.py and
 * utils/storage_queue.py (SURVEY.md section 2.3 lists them as K1..K21).  Each entry point below
 * names the reference call site(s) it replaces.  The Python host (vince_amd/) binds these with
 * ctypes and keeps the reference's class API on top.
 *
 * Conventions
 *  - plain pointers + sizes only; all pointers are DEVICE pointers unless marked host.
 *  - nothing here allocates or frees caller-visible memory; scratch is passed in (workspace).
 *  - every function returns 0 (VINCE_OK) or a negative VINCE_E_* code and never throws;
 *    vince_last_error() returns a thread-local message.  No kernel silently clamps a shape.
 *  - every launch goes to the hipStream_t passed as `stream` (void*; NULL = default stream);
 *    no host synchronisation inside any export.
 *  - activations are NHWC; `dtype` is VINCE_F32 or VINCE_BF16 (raw bfloat16 bits, uint16).
 */
#ifndef VINCE_HIP_H
#define VINCE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VINCE_OK 0
#define VINCE_E_SHAPE (-1)
#define VINCE_E_DTYPE (-2)
#define VINCE_E_ALIGN (-3)
#define VINCE_E_HIP (-4)
#define VINCE_E_ARG (-5)
#define VINCE_E_UNSUPPORTED (-6)

#define VINCE_F32 0
#define VINCE_BF16 1
/* Split-half products: the tensors are fp32 exactly as with VINCE_F32; the matrix kernels split every operand element into
 * hi + lo half-precision halves in registers and run hi*hi + hi*lo + lo*hi on the half-precision matrix pipe with fp32 accumulation
 * (3 MFMAs per block instead of fp32's 8 at 1/16 the rate).  H: IEEE half halves, products good to ~2^-22 -- the forward launches
 * (operands of order one; the reference's config-3 script runs fp32, vince/train_moco_v2.sh:40, and this is the mode that meets its
 * 1e-3 bar at bf16-class speed).  B: bfloat16 halves, ~2^-16 with fp32's exponent range -- the gradient launches.  Accepted by
 * vince_conv_igemm, vince_conv_wgrad(_det) and, as VINCE_F32X3 (= H forward / B backward), by vince_trunk_cfg.dtype; everywhere else
 * such tensors are passed as VINCE_F32. */
#define VINCE_F32X3H 2
#define VINCE_F32X3B 3
#define VINCE_F32X3 VINCE_F32X3H
/* fp32 tensors, SINGLE bfloat16 products (the hi halves only: one MFMA per block, 2^-9 per operand) -- gradient launches of the mixed mode
 * below; accepted by vince_conv_igemm and vince_conv_wgrad(_det) with the VINCE_F32X3B weight layout (the lo halves are not read).
 * VINCE_F32X3F (vince_trunk_cfg.dtype only): forward as VINCE_F32X3 -- split-half products, embeddings and loss at the fp32 reference's
 * 1e-3 bar -- and every GRADIENT convolution (input and weight gradients) as VINCE_F32X1B: the arithmetic of a bf16 mixed-precision
 * backward (what the reference's --use-apex would run) behind an fp32-grade forward.  Gram matrices (they feed forward statistics) stay
 * on VINCE_F32X3B.  (ABI 11) */
#define VINCE_F32X1B 4
#define VINCE_F32X3F 5

const char* vince_last_error(void);
/* Bumped whenever an exported signature, a struct layout or a dtype code changes.  vince_abi_version() returns the value the library
 * was BUILT with; a binding compares it at load (vince_amd/_lib.py: a stale .so behind VINCE_HIP_LIB would otherwise be called with
 * shifted arguments). */
#define VINCE_ABI_VERSION 12
int vince_abi_version(void);

/* Measurement aid (bench.py): while enabled, every conv_igemm / conv_wgrad launch is bracketed by a hipEvent pair on
 * its stream.  Tags: 0..3 = conv_igemm {f32 CT64, f32 CT128, bf16 CT64, bf16 CT128}, 4 = conv_wgrad f32,
 * 5 = conv_wgrad bf16.  collect() synchronises, returns per-tag total milliseconds / algorithmic FLOPs / launches and
 * clears the log.  Not for production runs (one event pair per launch). */
int vince_profile_enable(int on);
/* How many streams of its own the trunk engine may use beside the caller's: 2 (default) = weight gradients + the backward of
 * the downsample branch, 1 = weight gradients only, 0 = everything on the caller's stream.  This runtime serves at most
 * GPU_MAX_HW_QUEUES (4) hardware queues; streams beyond that share queues in creation order and serialise against each
 * other, so a host that brings streams of its own (a key-encoder stream, an RCCL communicator) lowers the engine's share:
 * the data-parallel solver sets 1.  Process-wide; call before the first backward. */
int vince_set_side_streams(int32_t n);
int vince_profile_collect(int32_t ntags, double* ms, double* flops, int64_t* count);
/* Kernel launches this library has enqueued since it was loaded (every launch site counts itself; memsets / copies issued through the
 * HIP runtime and the caller's own kernels are not included).  bench.py differences it over a step: `launches_per_step`.  (ABI 11) */
int64_t vince_launch_count(void);
/* Calibration aid (bench.py `roofline.hbm_achievable`, tools/ceilings.py): copies `bytes` (multiple of 16) with a plain
 * 16-byte-per-lane grid-stride kernel of `blocks` workgroups (<= 0: 2048) -- the HBM streaming rate this box gives an
 * element-wise pass, measured with the library's own code instead of a framework copy.  nontemporal: bit 0 = `nt` loads and stores;
 * bits 1-2 = the copy's shape (0: grid-stride, 16 bytes per lane; 1: 32 bytes per lane; 2: block-contiguous segments) -- the shapes
 * tried against the guide's 6.29 TB/s, tools/ceilings.py. */
int vince_stream_copy(void* dst, const void* src, size_t bytes, int32_t blocks, int32_t nontemporal, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Generalised tap convolution as implicit GEMM on MFMA (K1-K3, K8 forward; dgrad of the same).
 *
 *   out[n, ho*osh+oh0, wo*osw+ow0, co] (+)= sum_{a<TA, b<TB, ci<Ci}
 *        w[co, wt0 + a*wta + b*wtb, ci] * in[n, ho*sh + dh0 + a*dhs, wo*sw + dw0 + b*dws, ci]
 *
 * with out-of-range input pixels reading as zero.  One descriptor covers
 *  - forward conv2d (models/building_blocks/resnet.py:34-50,170): dh0=-pad, dhs=dil, sh=stride;
 *  - dgrad of a stride-1 conv: input = dY, weights = W^T ([Ci][T][Co]), dh0=+pad, dhs=-1;
 *  - dgrad of a stride-2 conv: one launch per output-pixel parity class (osh=osw=2, oh0/ow0 = parity);
 *  - nn.Linear forward / input-gradient (models/vince_model.py:38-42): N=batch, H=W=1, one tap.
 * Requirements: Ci multiple of 16 bytes worth of elements (8 bf16 / 4 f32), Co multiple of 8,
 * Ci/chunk a power of two unless TA*TB == 1.
 */
typedef struct vince_conv_desc {
    int32_t N, Hi, Wi, Ci;      /* input tensor [N][Hi][Wi][Ci] */
    int32_t Ho, Wo, Co;         /* output grid and channels */
    int32_t sh, sw;             /* grid -> input stride */
    int32_t TA, TB;             /* tap grid */
    int32_t dh0, dhs, dw0, dws; /* input offset of tap (a,b) */
    int32_t wt0, wta, wtb, WT;  /* weight tap index; weights are [Co][WT][Ci] */
    int32_t OH, OW;             /* output tensor [N][OH][OW][Co] */
    int32_t osh, osw, oh0, ow0; /* grid -> output pixel */
    int32_t Cs, Kw;             /* packed row taps, 0 / 0 = off.  Cs (< Ci) = element stride between input pixels and
                                 * Kw = kernel width: tap a reads the Ci contiguous elements (Ci / Cs pixels of one
                                 * input row) that start at pixel (ho*sh + dh0 + a*dhs, wo*sw + dw0); TB must be 1.
                                 * Weight element k of a tap is (kw = k / Cs, c = k % Cs) and must be zero where
                                 * kw >= Kw or c is a padding channel.  This is how the 7x7 stem (resnet.py:170) runs
                                 * with K = 7 x 32 instead of 49 taps x 8 padded channels: the input is stored
                                 * [N][H][Wp][4] with zero margins (vince_input_nchw_to_rows). */
} vince_conv_desc;

/* Per-channel BatchNorm reductions are accumulated with fp64 atomics into VINCE_STATS_REPLICAS interleaved copies
 * (workgroup b adds into copy b % R) so that tens of thousands of workgroups do not serialise on one L2 line per
 * channel; the consumer (vince_bn_finalize / vince_bn_bwd_apply) sums the copies.  Layout: double[R][C][2]. */
#define VINCE_STATS_REPLICAS 16

/* epilogue flags */
#define VINCE_EPI_ACCUMULATE 1 /* out = result + out (residual-gradient add); with acc_mask: out = result + out*mask */
#define VINCE_EPI_RELU 2       /* max(.,0) last */
/* VINCE_F32X3H launches only: `in` holds its elements as stored IEEE-half PAIRS (vince_bn_train.out_half_pairs writes that format) --
 * per 16 consecutive channels (64 bytes) four 16-byte chunks [hi e0-3,e8-11][hi e4-7,e12-15][lo e0-3,e8-11][lo e4-7,e12-15] of
 * x * 2^4, the very halves the kernel would split the fp32 element into -- so the main loop multiplies them as they are: no split
 * instructions (15-17 % of a 3x3 layer), bit-identical results.  Ci multiple of 16; direct-to-LDS kernels only.  (ABI 11) */
#define VINCE_EPI_IN_HALF_PAIRS 8

/* Optional fused BatchNorm-backward reduction: when the tensor a dgrad launch writes is the gradient dz that a
 * BatchNorm(+ReLU) backward consumes next, the epilogue also accumulates that BatchNorm's (sum g, sum g*xhat) over the
 * values it stores -- g = stored dz * relu-mask, xhat = (y - mean) * invstd -- into sums[R][C][2], i.e. exactly what
 * vince_bn_bwd_reduce would compute in a separate pass over dz and y (autograd of resnet.py:69-72,110-135).
 * Mask: mask_bits (1 byte per 16-byte chunk, as written by vince_bn_apply), or mask_scale/mask_shift
 * (sign of y*scale+shift), or neither (no ReLU). */
typedef struct vince_bn_reduce {
    const void* y;            /* BatchNorm input (conv output), same shape and dtype as this launch's `out`; NULL = off */
    const uint8_t* mask_bits;
    const float* mask_scale;
    const float* mask_shift;
    const float* mean;        /* saved batch mean / inverse std (vince_bn_finalize) */
    const float* invstd;
    double* sums;             /* double[R][C][2], zeroed by the caller, R = VINCE_STATS_REPLICAS */
} vince_bn_reduce;

typedef struct vince_conv_epi {
    int32_t flags;            /* VINCE_EPI_* */
    const float* bias;        /* optional float[Co] */
    double* stats;            /* optional double[R][Co][2]: per-channel (sum, sum of squares) of the conv output as
                               * stored in dtype, atomically accumulated (feeds vince_bn_finalize; BatchNorm2d train
                               * mode, resnet.py:69) */
    const uint8_t* acc_mask;  /* optional, only with VINCE_EPI_ACCUMULATE: one byte per 16-byte chunk of `out` (the ReLU
                               * bits written by vince_bn_apply); bit e gates element e, so the residual join
                               * out = dgrad + out * (z > 0) (autograd of resnet.py:132-133) happens in place */
    vince_bn_reduce bnred;    /* excludes stats */
    int32_t replicas;         /* how many of the R replicas of `stats` / `bnred.sums` this launch spreads its atomics over
                               * (0 = all VINCE_STATS_REPLICAS); few workgroups need few replicas, and a consumer that
                               * folds them itself (vince_bn_train_apply, vince_bn_bwd_apply) then reads less */
    /* Residual join with KNOWN BatchNorm constants (vince_bn_gram_finalize), only with VINCE_EPI_ACCUMULATE, in place:
     *   out = [relu]( conv * out_scale[co] + bias[co] + (id_scale ? out_old * id_scale[co] + id_shift[co] : out_old) )
     * i.e. the block's last BatchNorm (resnet.py:125-126), the identity / downsample-BatchNorm branch (:128-131) and the
     * ReLU (:133) all in the epilogue of conv3 -- its raw output is never written.  NULL = 1 / identity. */
    const float* out_scale;
    const float* id_scale;
    const float* id_shift;
    /* Gradient epilogues only (VINCE_EPI_ACCUMULATE or stats): one byte per 16-byte chunk of `out` (the ReLU bits of the block
     * BELOW, as vince_bn_apply / vince_conv_expand_join write them) applied to the value being stored:
     *   out = (dgrad + old) gated by out_mask
     * i.e. the gradient leaves this launch already masked by the ReLU it flows into next (autograd of resnet.py:132-133), and
     * `stats` then holds its per-channel sums -- what the BatchNorm-backward algebra of the block below needs
     * (vince_bn3_bwd_prepare) instead of a pass over that block's conv output. */
    const uint8_t* out_mask;
    /* A reduction split over TWO input tensors: the LAST tap (index TA*TB - 1; needs TA*TB >= 2) reads in2 -- same N x Hi x Wi, with
     * in2_channels <= Ci channels per pixel (its own row stride) -- instead of `in`; the weights stay [Co][WT][Ci] with only the
     * first in2_channels entries of that tap's block used.  With all tap displacements zero this is  out = W1 in + W2 in2  in one
     * launch (the input gradient of the BatchNorm-backward algebra: da = wd g + nq a, csrc/bn_algebra.hip).  Direct-to-LDS kernels only.
     * in2_repeat > 1 (ABI 12): the in2 row is read in2_repeat times over -- the tap's reduction is in2_repeat * in2_channels <= Ci long,
     * with consecutive weight blocks of in2_channels entries each: out = W1 in + (W2a + W2b + ...) in2, the way a weight matrix split
     * into bfloat16 hi + lo parts multiplies the same tensor (in2_channels / 8 a power of two).  0 / 1: once. */
    const void* in2;
    int32_t in2_channels;
    int32_t in2_repeat;
    /* Forward epilogue of fp32-store launches only (VINCE_F32 / VINCE_F32X3H, no split reduction): a bfloat16 SHADOW of `out` -- the
     * same NHWC element offsets, every stored value also written rounded to bfloat16 (8 bytes per 16-byte chunk).  What the mixed mode
     * "x3f" saves for its bf16 backward (vince_trunk_set_shadow).  NULL = none.  (ABI 11) */
    void* out2;
    /* Forward residual join (out_scale / id_scale epilogue) of an fp32-store launch, for the same purpose: raw2 = bfloat16 of the RAW
     * convolution output minus raw2_mean[c] (the centred input of the BatchNorm this epilogue applies: what a backward over the shadow
     * reads, see vince_bn_train.y_centred_bf16), and mask2 = the ReLU bits of the stored value in the bf16 tensors' format (one byte per
     * 8 channels).  out2 is allowed there too (the bf16 copy of the joined output).  raw2 and raw2_mean come together.  (ABI 11) */
    void* raw2;
    const float* raw2_mean;
    uint8_t* mask2;
} vince_conv_epi;

/* in/w/out have element type `dtype`.  epi may be NULL (plain store). */
int vince_conv_igemm(const vince_conv_desc* d, int dtype, const void* in, const void* w, void* out,
                     const vince_conv_epi* epi, void* stream);

/* The bottleneck's last 1x1 convolution (resnet.py:123) with its BatchNorm, the residual join and the ReLU (resnet.py:125-133) in
 * registers, for BatchNorm constants known beforehand (vince_bn_gram_finalize):
 *   out[p][co] = [relu]( out_scale[co] * sum_k w[co][k] x[p][k] + out_shift[co] + (id_scale ? identity * id_scale + id_shift : identity) )
 * x [rows][K], w [Co][K], identity / out [rows][Co] (out may alias identity), bf16, K = 64 or 128, Co multiple of 256.
 * A persistent streaming kernel (csrc/conv_xjoin.hip): weights resident in LDS, input tiles by a loader wavefront, outputs
 * straight from the accumulators -- the HBM-bound replacement of vince_conv_igemm's join epilogue for layer1 / layer2.
 * Training forwards also pass y_raw (the convolution's own output rounded to bf16, [rows][Co] -- what BatchNorm backward
 * reads) and mask_out (one byte per 16-byte chunk of out, bit e = pre-ReLU value e > 0, as vince_bn_apply writes it); with
 * either, out must not alias identity.  mask_out alone (y_raw NULL) is the forward of the BatchNorm-backward algebra, which never
 * reads the convolution's output again (vince_bn3_bwd_prepare). */
int vince_conv_expand_join(int dtype, const void* x, const void* w, int64_t rows, int32_t K, int32_t Co,
                           const float* out_scale, const float* out_shift, const void* identity, const float* id_scale,
                           const float* id_shift, void* out, void* y_raw, uint8_t* mask_out, int relu, void* stream);

/* vince_conv_expand_join plus the FIRST convolution of the bottleneck that follows (resnet.py:117 of the next block: 1x1, stride 1,
 * Co -> Co_next) on the block output while it is still on chip:
 *   y_next[p][c] = sum_k w_next[c][k] out[p][k]      (bf16, what vince_conv_igemm stores for that layer, bit for bit)
 * with the BatchNorm statistics of the stored y_next in stats_next (double[replicas_next][Co_next][2], zeroed by the caller; optional);
 * or, for a folded inference trunk (bias_next non-NULL), y_next = [relu_next]( y_next + bias_next[c] ) with vince_conv_igemm's roundings.
 * The block output is 4x wider than anything else in a bottleneck; this launch saves its re-read (411 MB per layer1 block boundary at
 * 256 frames).  K = 64, Co = 256; Co_next = 64 (layer1's identity blocks) or, with a plain identity (id_scale NULL), 128 (the first
 * block of layer2 behind layer1's last); everything else as vince_conv_expand_join. */
int vince_conv_expand_join_next(int dtype, const void* x, const void* w, int64_t rows, int32_t K, int32_t Co,
                                const float* out_scale, const float* out_shift, const void* identity, const float* id_scale,
                                const float* id_shift, void* out, void* y_raw, uint8_t* mask_out, int relu, const void* w_next,
                                int32_t Co_next, void* y_next, double* stats_next, int32_t replicas_next, const float* bias_next,
                                int relu_next, void* stream);

/* The same streaming structure for an expand convolution on its own: out[p][co] = sum_k w[co][k] x[p][k] (bf16, K = 64 / 128,
 * Co multiple of 256, stride 1) with the BatchNorm statistics of the STORED values accumulated into stats
 * (double[replicas][Co][2], zeroed by the caller; optional) -- what vince_conv_igemm(stats) computes for the same layer, as a
 * persistent HBM stream whose statistics live in registers for the whole launch. */
int vince_conv_expand_stats(int dtype, const void* x, const void* w, int64_t rows, int32_t K, int32_t Co, void* out,
                            double* stats, int32_t replicas, void* stream);

/* Layer1's 3x3 convolution (reference models/building_blocks/resnet.py:119-121 conv2 of the 64-wide bottlenecks: 64 -> 64 channels,
 * 56 pixels wide, stride 1, pad 1, bf16) as an image-strip kernel: input rows resident in an LDS ring, every input element crosses
 * the L2 -> LDS path once, the nine taps are shifted fragment reads.  x [N][H][56][64], w [64][9][64] (the prepared [Co][tap][Ci]
 * copy), out [N][H][56][64]; per-channel (sum, sum of squares) of the STORED output into stats (double[replicas][64][2], zeroed by
 * the caller; optional) exactly as vince_conv_igemm(stats) computes them.  tap_map: the weight tap read for kernel position
 * (dh + 1) * 3 + (dw + 1) (null: identity = the forward convolution).  H must be a multiple of 4. */
int vince_conv3x3_strip(int dtype, const void* x, const void* w, int32_t N, int32_t H, int32_t W, int32_t Ci, int32_t Co,
                        const int32_t* tap_map, void* out, double* stats, int32_t replicas, void* stream);
/* The same layer with the epilogue of the BatchNorm-folded inference forward (vince_trunk_forward_folded): out = [relu](conv + bias[co]),
 * bias / ReLU applied to the bf16-rounded convolution output as vince_conv_igemm's epilogue does (bit-identical to it); no statistics. */
int vince_conv3x3_strip_bias(int dtype, const void* x, const void* w, int32_t N, int32_t H, int32_t W, int32_t Ci, int32_t Co,
                             const float* bias, int32_t relu, void* out, void* stream);
/* The INPUT GRADIENT of that layer through the same kernel (autograd of resnet.py:119-121 under loss.backward()): dx [N][H][56][64] =
 * conv3x3(dy, wt with the taps flipped), wt = the prepared [Ci][tap][Co] copy, with vince_conv_igemm's fused BatchNorm-backward reduction
 * of the BatchNorm + ReLU below (bnred with mask_scale / mask_shift, no mask bits; may be NULL): bnred->sums += (sum g, sum g * xhat) of
 * the STORED gradient, g gated by the sign of y * mask_scale + mask_shift. */
int vince_conv3x3_strip_dgrad(int dtype, const void* dy, const void* wt, int32_t N, int32_t H, int32_t W, int32_t C, void* dx,
                              const vince_bn_reduce* bnred, int32_t replicas, void* stream);

/* And for the block-input gradient of a bottleneck (the input gradient of its conv1 = an expand-shaped 1x1 again: dx [rows][Co]
 * from dy [rows][K] and W^T [Co][K], bf16, K = 64 / 128, Co multiple of 256) with vince_conv_igemm's gradient epilogues:
 *   out = dgrad + (accumulate ? (acc_mask ? out_old gated by the mask bits : out_old) : 0)        in place, and
 *   bnred->sums += (sum g', sum g' * xhat) of the stored out, g' = out gated by bnred->mask_bits, xhat = (bnred->y - mean) * invstd
 * (mask_scale / mask_shift are not supported here). */
int vince_conv_expand_dgrad(int dtype, const void* dy, const void* wt, int64_t rows, int32_t K, int32_t Co, void* out,
                            int accumulate, const uint8_t* acc_mask, const vince_bn_reduce* bnred, int32_t replicas, void* stream);
/* The same with the epilogue of the BatchNorm-backward algebra (vince_conv_epi.out_mask): out = (dgrad + old [gated by acc_mask])
 * gated by out_mask, and gsums (double[replicas][Co][2], zeroed by the caller) += per-channel (sum, sum of squares) of the
 * stored values.  No conv output of the block below is read. */
int vince_conv_expand_dgrad_masked(int dtype, const void* dy, const void* wt, int64_t rows, int32_t K, int32_t Co, void* out,
                                   int accumulate, const uint8_t* acc_mask, const uint8_t* out_mask, double* gsums,
                                   int32_t replicas, void* stream);

/* Weight gradient (wgrad) of the same generalised conv, reduction over output pixels:
 *   dw[co, wt(a,b), ci] += sum_{n,ho,wo} dy[n,ho,wo,co] * in[n, ho*sh+dh0+a*dhs, wo*sw+dw0+b*dws, ci]
 * dw is float[Co][WT][Ci_dw] accumulated with fp32 atomics (zero it first); only ci < Ci_dw is written
 * (the stem pads Ci 3 -> 4/8).  dy is [N][Ho][Wo][Co] dense.  Also nn.Linear weight gradient (one tap).
 * Packed row taps (d.Cs > 0): dw is float[Co][TA][Kw][Ci_dw] -- element k of tap a lands at (kw = k / Cs, c = k % Cs),
 * the padding positions (kw >= Kw, c >= Ci_dw) are dropped -- i.e. the stem's ordinary [Co][7][7][3] gradient.
 * `variant`: 0 = default operand fetch (ds_read_b64_tr_b16 for bf16), 1 = scalar-gather fallback. */
int vince_conv_wgrad(const vince_conv_desc* d, int dtype, const void* in, const void* dy, float* dw,
                     int32_t Ci_dw, int variant, void* stream);

/* The same weight gradient, REPRODUCIBLE: every pixel-range split stores its partial result into `scratch` (plain stores, dw layout per
 * split) and one reduction launch adds the splits into dw in a fixed order -- no fp32 atomics, so two runs give identical bits (and the
 * Gram matrices / BatchNorm constants derived from them likewise).  scratch: vince_conv_wgrad_scratch_bytes() bytes (16-byte aligned;
 * a shorter buffer lowers the split count); NULL = the atomic path of vince_conv_wgrad.  Needs the plain forward tap order. */
int vince_conv_wgrad_det(const vince_conv_desc* d, int dtype, const void* in, const void* dy, float* dw, int32_t Ci_dw,
                         void* scratch, size_t scratch_bytes, void* stream);
size_t vince_conv_wgrad_scratch_bytes(const vince_conv_desc* d, int dtype, int32_t Ci_dw);

/* ---------------------------------------------------------------------------------------------
 * BatchNorm2d (K4/K5; resnet.py:69,72,110,112,171; eps 1e-5, momentum 0.1)
 */
/* train != 0: (sum,sumsq) -> batch mean / biased var -> scale = gamma*invstd, shift = beta - mean*scale;
 * running stats updated with the UNBIASED variance, num_batches_tracked += 1.
 * train == 0: scale/shift from the running stats (VinceSolver.run_val, vince_solver.py:522). */
int vince_bn_finalize(const double* stats, int64_t count, int32_t C, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                      float eps, int train, float* scale, float* shift, float* save_mean, float* save_invstd,
                      void* stream);

/* vince_bn_finalize(train) + vince_bn_apply in ONE launch: every workgroup folds the statistic replicas of its own channels,
 * row-block 0 publishes scale / shift / mean / invstd (read by the backward pass) and updates the running statistics. */
typedef struct vince_bn_train {
    const double* stats;          /* double[R][C][2], only the first `replicas` copies need to hold data (rest zero or unused) */
    int32_t replicas;             /* copies to fold; 0 = VINCE_STATS_REPLICAS */
    int64_t count;                /* elements per channel */
    const float* gamma;
    const float* beta;
    float* running_mean;          /* optional pair */
    float* running_var;
    int64_t* num_batches_tracked; /* optional */
    float momentum, eps;
    float* scale;                 /* outputs, float[C] */
    float* shift;
    float* save_mean;             /* optional outputs */
    float* save_invstd;
    double* out_sum;              /* optional double[out_sum_replicas][C], zeroed by the caller: per-channel sums of the values
                                   * this launch STORES (after ReLU, rounded to dtype), atomically accumulated -- the column sums
                                   * vince_bn_gram_finalize needs beside the Gram matrix of the same tensor */
    int32_t out_sum_replicas;
    /* fp32 launches only: a bfloat16 shadow of `out` (same element offsets) and, instead of mask_out's fp32 format (one byte per 4
     * channels), a mask in the BF16 format -- one byte per 8 channels, bit e = channel e of the chunk -- at mask_bf16.  Either may be
     * NULL.  (vince_trunk_set_shadow; ABI 11) */
    void* out_bf16;
    uint8_t* mask_bf16;
    /* ... and a bfloat16 shadow of the INPUT y, CENTRED: y - mean[c], rounded to bfloat16 (same element offsets), with the constants a
     * backward over that shadow needs in shadow_consts = float[4][C]: scale, beta (= shift + mean * scale), 0 (the mean of the centred
     * tensor), invstd.  A bf16 copy of the raw y would lose (y - mean) to cancellation wherever |mean| >> std; the centred one keeps
     * 8 significand bits of the deviation itself.  Both NULL or both set. */
    void* y_centred_bf16;
    float* shadow_consts;
    /* fp32 launches, C multiple of 16: `out` is written as stored IEEE-half pairs (the layout of VINCE_EPI_IN_HALF_PAIRS) instead of
     * fp32 -- for a tensor whose ONLY reader is a split-half convolution (a bottleneck's bn1 output in front of its 3x3).  The shadows
     * and out_sum are still made from the fp32 values. */
    int32_t out_half_pairs;
} vince_bn_train;
int vince_bn_train_apply(int dtype, const void* y, const vince_bn_train* bt, const void* identity, const float* id_scale,
                         const float* id_shift, void* out, uint8_t* mask_out, int64_t rows, int32_t C, int relu,
                         void* stream);

/* vince_bn_train_apply (relu, no identity) for a narrow bf16 tensor -- a = relu(bn2(y)), resnet.py:119-121 -- that also returns what
 * vince_bn_gram_finalize needs of the tensor it stores: gram[C][C] += sum over rows of a a^T (float, zeroed by the caller) beside
 * bt->out_sum.  C = 64 or 128, bf16 only (VINCE_E_SHAPE / VINCE_E_DTYPE otherwise: the caller then runs vince_bn_train_apply and
 * vince_conv_wgrad(in = dy = a)).  Each workgroup keeps the 64-row slices it writes in LDS and multiplies them on the matrix
 * pipe; partial matrices go to per-workgroup slabs in `scratch` (vince_bn_train_apply_gram_scratch_bytes) and are added in a fixed
 * order: the result is run-to-run identical.  Saves the extra read of `a` and two launches per bottleneck. */
size_t vince_bn_train_apply_gram_scratch_bytes(int64_t rows, int32_t C);
int vince_bn_train_apply_gram(int dtype, const void* y, const vince_bn_train* bt, void* out, int64_t rows, int32_t C, float* gram,
                              void* scratch, size_t scratch_bytes, void* stream);

/* Train-mode BatchNorm constants of a 1x1 convolution's output WITHOUT running the convolution first (resnet.py:125-126:
 * bn3(conv3(a))).  For y = W a per pixel, mean_y = W mean_a and var_y[c] = w_c^T Cov(a) w_c, so the batch statistics of y follow
 * from the Gram matrix of the conv INPUT:  gram = sum_pix a a^T  (float[K][K]; vince_conv_wgrad with in = dy = a) and its column
 * sums (colsum double[R][K]; vince_bn_train.out_sum of the pass that wrote a).  Products are exact in fp32 for bf16 operands and
 * the covariance is formed and contracted in fp64, so the result is the statistic of the UNROUNDED conv output -- closer to
 * the fp32 reference than the statistics of a bf16-stored output.  K <= 512, K and Co multiples of the 16-byte chunk.
 * Outputs as vince_bn_finalize(train): scale = gamma * invstd, shift = beta - mean * scale, save_mean / save_invstd, running
 * statistics with the unbiased variance, num_batches_tracked += 1.  What this buys: conv3's epilogue can then apply BatchNorm +
 * identity + ReLU (vince_conv_epi.out_scale) and the [rows][Co] raw output is neither written nor re-read. */
int vince_bn_gram_finalize(int dtype, const float* gram, const double* colsum, int32_t colsum_replicas, int64_t count,
                           const void* w, int32_t K, int32_t Co, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, int64_t* num_batches_tracked, float momentum, float eps, float* scale,
                           float* shift, float* save_mean, float* save_invstd, void* stream);

/* BatchNorm backward THROUGH a bottleneck's last 1x1 convolution, without that convolution's output (csrc/bn_algebra.hip; autograd of
 * resnet.py:123-133 with BatchNorm2d in train mode).  y = W a is linear and pointwise, so with g = the gradient of the block output
 * gated by its ReLU, R = g^T a (vince_conv_wgrad with dy := g) and the per-channel sums of g:
 *   coef[5][Co] = s = gamma*invstd, c1 = mean g, c2 = mean g*xhat = invstd (<W[c,:], R[c,:]> - mean sum g) / count, t = s*c2*invstd,
 *                 and the mean those formulas used (row 4: pass it to vince_bn3_bwd_finish_dw)
 *   dgamma += count*c2, dbeta += count*c1
 *   wd [K][Co] bf16 = W^T diag(s)  and  nq [K][K] bf16 = -W^T diag(t) W  (row strides wd_ld / nq_ld in elements): the weights of
 *                     da = wd g + nq a + nr -- ONE launch when interleaved as the two taps [K][2][Co] of vince_conv_epi.in2
 *   nr [K]     f32  = -sum_c W[c][k] (s c1 - t mean)
 * w is the bf16 [Co][K] copy the forward multiplied with; gsums double[replicas][Co][2] (first of each pair = sum of g). K <= 128.
 * colsum (double[colsum_replicas][K], the column sums of a; may be NULL): the algebra is then self-consistent on the IMPLIED y = w a --
 * the mean is sum_k w[c][k] colsum[k] / count instead of `mean` (they differ when the forward multiplied with other weights than w: the
 * mixed mode's fp32 masters) and nr is formed from the ROUNDED wd / nq, nr = -wd_r c1 - nq_r colsum / count, so that the pixel sums of da
 * vanish to fp32 rounding as they do behind a stored dy (ABI 12).
 * w_dtype (ABI 12): VINCE_BF16, or VINCE_F32 -- `w_bf16` then points at the fp32 master weights [Co][K] (the mixed mode: the forward
 * multiplied with those).  wd_lo / nq_lo (both or neither; same row strides as wd / nq): the bfloat16 remainders wd - wd_hi, nq - nq_hi
 * -- 16 mantissa bits for the input gradient's weights, multiplied as two more reduction blocks (taps [wd | wd_lo] on g,
 * vince_conv_epi.in2_repeat = 2 over [nq | nq_lo] on a): what keeps the masked pixel sums the BatchNorm below reduces at the
 * separate passes' accuracy (csrc/bn_algebra.hip). */
int vince_bn3_bwd_prepare(const float* R, const void* w_bf16, const double* gsums, int32_t replicas, const float* mean,
                          const float* invstd, const float* gamma, int64_t count, int32_t Co, int32_t K, float* coef, void* wd,
                          int32_t wd_ld, void* nq, int32_t nq_ld, float* nr, float* dgamma, float* dbeta, const double* colsum,
                          int32_t colsum_replicas, int32_t w_dtype, void* wd_lo, void* nq_lo, void* stream);
/* ... and the weight gradient, in place of R:  dW = diag(s) (R - c1 A^T - diag(c2 invstd) (W G - mean A^T)),  G = a^T a (float[K][K], the
 * Gram matrix the forward's statistics came from, vince_bn_gram_finalize), A = column sums of a (double[colsum_replicas][K]). */
int vince_bn3_bwd_finish_dw(float* RdW, float* dw_accum, const void* w_bf16, const float* gram, const double* colsum, int32_t colsum_replicas,
                            const float* coef, const float* mean, const float* invstd, int32_t Co, int32_t K, int32_t w_dtype, void* stream);
/* dw_accum NULL: in place (R becomes dW).  dw_accum non-NULL: R is left untouched and the finished gradient is ADDED into dw_accum -- R then
 * lives in scratch and the gradient buffer keeps the accumulate-into contract of vince_conv_wgrad (gradient accumulation over several
 * backward passes without zero_grad). */

/* out = [relu]( y*scale + shift + (identity ? (id_scale ? identity*id_scale + id_shift : identity) : 0) ).
 * mask_out (optional): one byte per 16-byte chunk of `out` (8 bf16 / 4 f32 channels), bit e = pre-ReLU value e > 0 --
 * the ReLU mask backward needs, at 1/16 of the bytes of re-reading the activation. */
int vince_bn_apply(int dtype, const void* y, const float* scale, const float* shift, const void* identity,
                   const float* id_scale, const float* id_shift, void* out, uint8_t* mask_out, int64_t rows, int32_t C,
                   int relu, void* stream);

/* Backward pass 1: g = dz * relu_mask;  sums[c] += (sum g, sum g*xhat), xhat = (y - mean)*invstd.
 * relu_mask comes from (first non-NULL wins) mask_bits (bytes written by vince_bn_apply), mask_scale/mask_shift
 * (sign of y*scale+shift, recomputed from y: plain BN+ReLU), mask_src (sign of a materialised activation), else 1.
 * sums is double[R][C][2], zeroed by the caller; atomics are spread over the first `replicas` copies (0 = all). */
int vince_bn_bwd_reduce(int dtype, const void* dz, const void* mask_src, const uint8_t* mask_bits, const float* mask_scale,
                        const float* mask_shift, const void* y, const float* mean, const float* invstd, double* sums,
                        int64_t rows, int32_t C, int32_t replicas, void* stream);

/* Optional second reduction of vince_bn_bwd_apply: another BatchNorm that consumes the SAME masked gradient g (the
 * downsample branch next to a block's last BatchNorm, resnet.py:128-133) gets its (sum g, sum g*xhat2) accumulated in the
 * same pass, xhat2 = (y - mean)*invstd with that BatchNorm's own input and statistics. */
typedef struct vince_bn_reduce2 {
    const void* y;        /* NULL = off */
    const float* mean;
    const float* invstd;
    double* sums;         /* double[R][C][2], zeroed by the caller */
    int32_t replicas;     /* copies to spread the atomics over; 0 = VINCE_STATS_REPLICAS */
} vince_bn_reduce2;

/* Backward pass 2: dy = gamma*invstd*(g - sum_g/count - xhat*sum_gx/count); optional g_out = g (the
 * residual branch's gradient); dgamma += sum_gx, dbeta += sum_g (float[C]).  The first `replicas` copies of `sums`
 * (0 = all) are folded by every workgroup for its own channels; no separate fold launch. */
int vince_bn_bwd_apply(int dtype, const void* dz, const void* mask_src, const uint8_t* mask_bits, const float* mask_scale,
                       const float* mask_shift, const void* y, const float* mean, const float* invstd, const float* gamma,
                       const double* sums, int64_t count, void* dy, void* g_out, float* dgamma, float* dbeta, int64_t rows,
                       int32_t C, int32_t replicas, const vince_bn_reduce2* second, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pooling (K6 MaxPool2d 3x3/s2/p1 resnet.py:173, fused with the stem's BN-apply + ReLU; K7 AdaptiveAvgPool2d
 * vince_model.py:33)
 */
int vince_stem_pool_fwd(int dtype, const void* y, const float* scale, const float* shift, void* out,
                        uint8_t* argmax, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
/* g[n,h,w,c] = sum of dpool over the windows whose argmax is (h,w)  (zero where relu(bn(y)) == 0) */
int vince_stem_pool_bwd(int dtype, const void* dpool, const uint8_t* argmax, void* g, int32_t N, int32_t H,
                        int32_t W, int32_t C, void* stream);
/* Stem backward without the materialised pre-pool gradient g: the max-pool gather above fused into the BatchNorm
 * backward of bn1 (resnet.py:171-173 under autograd).  _reduce accumulates (sum g, sum g*xhat) into sums[R][C][2]
 * (zeroed by the caller); _apply folds the replicas, adds dgamma / dbeta and writes
 * dy = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)).  H, W: pre-pool size.  Results equal
 * vince_stem_pool_bwd -> vince_bn_bwd_reduce / vince_bn_bwd_apply on the stored g. */
int vince_stem_bwd_reduce(int dtype, const void* dpool, const uint8_t* argmax, const void* y, const float* mean,
                          const float* invstd, double* sums, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int vince_stem_bwd_apply(int dtype, const void* dpool, const uint8_t* argmax, const void* y, const float* mean,
                         const float* invstd, const float* gamma, double* sums, void* dy, float* dgamma, float* dbeta,
                         int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int vince_avgpool_fwd(int dtype, const void* x, float* out, int32_t N, int32_t HW, int32_t C, void* stream);
int vince_avgpool_bwd(int dtype, const float* dout, void* dx, int32_t N, int32_t HW, int32_t C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Layout transforms
 */
/* float NCHW [N][3][H][W] -> dtype NHWC [N][H][W][Cp] with channels 3..Cp-1 zero (Cp = 4 f32 / 8 bf16).
 * `perm` (optional int64[N]) gathers source images: out[i] = in[perm[i]] (batch shuffle, vince_model.py:137-142). */
int vince_input_nchw_to_nhwc(int dtype, const float* in, const int64_t* perm, void* out, int32_t N, int32_t C,
                             int32_t H, int32_t W, int32_t Cp, void* stream);
/* jigsaw tiling (vince_model.py:144-155): float NCHW [N][C][H][W] -> dtype NHWC [9N][th][tw][Cp], zero pad */
int vince_jigsaw_nchw_to_nhwc(int dtype, const float* in, void* out, int32_t N, int32_t C, int32_t H, int32_t W,
                              int32_t th, int32_t tw, int32_t Cp, void* stream);
/* The packed-row-tap input layout of the stem (vince_conv_desc.Cs): dtype [N][H][Wp][4], image column w at index
 * w + left, channel 3 and the columns outside [left, left + W) zero.  Same `perm` / jigsaw semantics as above. */
int vince_input_nchw_to_rows(int dtype, const float* in, const int64_t* perm, void* out, int32_t N, int32_t C,
                             int32_t H, int32_t W, int32_t Wp, int32_t left, void* stream);
int vince_jigsaw_nchw_to_rows(int dtype, const float* in, void* out, int32_t N, int32_t C, int32_t H, int32_t W,
                              int32_t th, int32_t tw, int32_t Wp, int32_t left, void* stream);
/* GPU input stage (the deterministic part of utils/transforms.py:62-235): uint8 HWC frames
 * [N][Hs][Ws][3] -> the packed stem layout in ONE pass: per-image crop window (crop_yx int32[N][2] = top-left corner of the
 * H x W window, NULL = (0,0)), horizontal flip (flip uint8[N], NULL = none), batch gather (perm, as above) and the
 * ToTensor + Normalize arithmetic ((u8 - mean255[c]) / std255[c], constants.py:28-29; mean255 / std255 are HOST float[3])
 * -- a quarter of the bytes of float frames across PCIe and no separate normalisation pass.  crop windows must lie
 * inside the source frame (the caller's responsibility, not checked on the device). */
int vince_input_u8hwc_to_rows(int dtype, const uint8_t* in, const int64_t* perm, const int32_t* crop_yx, const uint8_t* flip,
                              const float* mean255, const float* std255, void* out, int32_t N, int32_t Hs, int32_t Ws,
                              int32_t H, int32_t W, int32_t Wp, int32_t left, void* stream);
/* GPU input stage, random part: the per-sample PIL pipeline of utils/transforms.py:62-235
 * (RandomResizedCrop -> ColorJitter / RandomGrayscale -> RandomHorizontalFlip -> ToTensor + Normalize -> RandomGaussianBlur)
 * on uint8 frames resident in HBM.  The random DRAWS stay on the host (the caller passes boxes, op orders, factors, flips,
 * blur kernels); the kernels reproduce the arithmetic of Pillow (requirements.txt:12) and torchvision 0.5 functional ops
 * bit for bit on the uint8 stages.
 *
 * _resized_crop_u8: output n = BILINEAR resize to H x W of the window box[n] = {top, left, height, width} (int32[N][4]) of
 *   frame src_index[n] (int64[N], NULL: n) of frames uint8 [..][Hs][Ws][3]  (torchvision F.resized_crop on a PIL image:
 *   crop, then Image.resize -- Resample.c: antialiased triangle filter, 22-bit fixed-point coefficients, horizontal pass into
 *   a uint8 intermediate, then the vertical pass).  tmp: uint8 [N][Hs][W][3] scratch; table: int32 scratch of
 *   vince_aug_resample_table_ints() entries; out: uint8 [N][H][W][3].  Windows must lie inside the frame.
 * _color_u8: in place on uint8 [N][H][W][3]; per image up to max_ops steps in order, op int32[N][max_ops], factor
 *   float[N][max_ops]: -1 skip; 0 brightness, 1 contrast, 2 saturation (ImageEnhance: Image.blend(degenerate, image, factor),
 *   Blend.c float32 arithmetic); 3 hue (factor = the uint8 shift of the H plane, i.e. uint8(hue_factor * 255), as a float;
 *   Convert.c RGB<->HSV); 4 grayscale with 3 equal channels (RandomGrayscale; Convert.c luma).
 * _blur_to_rows: uint8 [N][H][W][3] -> the packed stem layout (see vince_input_u8hwc_to_rows): horizontal flip (flip uint8[N]
 *   or NULL), (u8 - mean255) / std255, and for images with do_blur[n] != 0 (uint8[N] or NULL = none) the separable Gaussian
 *   blur of utils/util_functions.py:104-132 on the NORMALISED tensor (H direction, then W direction, zero padding) with the
 *   per-image taps kernels float[N][ks] (ks odd; the caller evaluates exp(-d^2 / (2 sigma^2)) / sum exactly as the reference
 *   does).  tmp: float [N][H][W][4] scratch, used only when the kernel is too long for the fused LDS path (ks > ~40). */
int vince_aug_resized_crop_u8(const uint8_t* frames, const int64_t* src_index, const int32_t* box, int32_t* table, uint8_t* tmp,
                              uint8_t* out, int32_t N, int32_t Hs, int32_t Ws, int32_t H, int32_t W, void* stream);
/* ints of the `table` scratch above (the per-image filter coefficient tables, computed once per call by a small kernel) and
 * the longest filter a window inside an Hs x Ws frame can need (Resample.c ksize). */
int64_t vince_aug_resample_table_ints(int32_t N, int32_t Hs, int32_t Ws, int32_t H, int32_t W);
int vince_aug_resample_kmax(int32_t Hs, int32_t Ws, int32_t H, int32_t W);
int vince_aug_color_u8(uint8_t* img, const int32_t* op, const float* factor, int32_t max_ops, int32_t N, int32_t H, int32_t W,
                       void* stream);
int vince_aug_blur_to_rows(int dtype, const uint8_t* img, const uint8_t* flip, const float* kernels, const uint8_t* do_blur,
                           int32_t ks, const float* mean255, const float* std255, float* tmp, void* out, int32_t N, int32_t H,
                           int32_t W, int32_t Wp, int32_t left, void* stream);
/* fp32 master weights [Co][T][Ci] -> compute copy [Co][T][Cip] (dtype) and, if wt != NULL, the dgrad copy
 * [Ci][T][Co] (dtype). */
int vince_prepare_weight(int dtype, const float* w, void* wk, void* wt, int32_t Co, int32_t T, int32_t Ci,
                         int32_t Cip, void* stream);
/* out[cols][rows] = transpose of in[rows][cols] (fp32; the input-gradient copy W^T of an nn.Linear weight, vince_model.py:38-49:
 * made on demand by the host, only for heads whose backward actually runs). */
int vince_transpose_f32(const float* in, float* out, int32_t rows, int32_t cols, void* stream);
/* The same for many layers in ONE launch; `table_dev` is a device array of n entries. */
typedef struct vince_prep_entry {
    const void* w; /* float [Co][T][Ci] */
    void* wk;      /* dtype [Co][T][Cip] */
    void* wt;      /* dtype [Ci][T][Co] or NULL */
    const float* scale; /* optional float[Co]: the row of output channel co is multiplied by scale[co] (BatchNorm folding) */
    int32_t Co, T, Ci, Cip;
    int32_t Cs, Kw; /* packed row taps (0 / 0 = off): w is float [Co][T][Kw][Ci], wk element k of tap t is
                     * (kw = k / Cs, c = k % Cs) -> w[co][t][kw][c], zero where kw >= Kw or c >= Ci */
} vince_prep_entry;
int vince_prepare_weights_batched(int dtype, const vince_prep_entry* table_dev, int32_t n, void* stream);

/* dtype NHWC -> float NCHW (spatial_features for callers that want the reference layout) */
int vince_nhwc_to_nchw_f32(int dtype, const void* in, float* out, int32_t N, int32_t C, int32_t H, int32_t W,
                           void* stream);

/* ---------------------------------------------------------------------------------------------
 * Head pieces (K8 bias/ReLU glue, K9 F.normalize vince_model.py:180)
 */
int vince_l2norm_fwd(const float* x, float* out, float* norms, int32_t rows, int32_t D, float eps, void* stream);
int vince_l2norm_bwd(const float* x, const float* norms, const float* dout, float* dx, int32_t rows, int32_t D,
                     float eps, void* stream);
int vince_relu_bwd(const float* dout, const float* act, float* dx, int64_t n, void* stream);
int vince_colsum(const float* x, float* out, int32_t rows, int32_t cols, void* stream); /* out[c] += sum_r x[r][c] */
/* The reference asserts torch.isfinite(total_loss) on EVERY iteration (solvers/vince_solver.py:446-452), a host sync per step.
 * Here the check stays on the device: latch[0] += 1 and latch[1] = step + 1 (first offender only) when *value is NaN or +-inf.
 * The caller zeroes latch (int64[2]) once, enqueues this every step and reads the latch whenever it synchronises anyway. */
int vince_nonfinite_latch(const float* value, int64_t step, int64_t* latch, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused similarity + InfoNCE + metrics (K10-K14; vince_model.py:198-349, utils/loss_util.py:7-62)
 *
 * Column space of row i: [ in-batch columns 0..Bk-1 from `inb` | queue columns 0..K-1 ].
 * Positives of row i: in-batch columns j with j / frames == i / frames (block diagonal; frames = 1 -> diagonal).
 * offdiag_neg != 0 (inter-batch mode, vince_model.py:207-225): other in-batch columns are negatives.
 * offdiag_neg == 0 (MoCo mode, :227-233): they are ignored, i.e. logits = [q_i.k_i | q_i.queue].
 * K == 0 with inb == q gives the self-similarity term (:213-222).
 * All arithmetic fp32 on the f32 MFMA; logits are never materialised.
 */
typedef struct vince_infonce_desc {
    int32_t B, D, Bk, K;
    int32_t frames;
    int32_t offdiag_neg;
    float inv_temperature;
} vince_infonce_desc;

size_t vince_infonce_workspace_bytes(const vince_infonce_desc* d);
/* outputs: pos[B][P] raw cosines of the positives (P = frames), row_max[B], neg_sum[B] (sum of exp(s - row_max)
 * over negatives), dists[B][P], softmax_weights[B][P], scalars[8] = {loss mean, softmax_weight mean,
 * accuracy mean, mean positive cosine, mean row-max negative cosine, 0,0,0}. */
/* PRECONDITION: q, inb and queue hold UNIT vectors (the reference L2-normalises both sides, vince_model.py:180): the logits are formed
 * from IEEE-half hi / lo halves of the operands scaled by 2^8, so an entry with |x| >= 255.9 turns its row into NaN by design (loud, not
 * saturated) -- this is not a general-purpose similarity kernel.
 * logits: optional float[B][Bk + K] -- the raw cosines, in-batch columns first (what the reference materialises as
 * `vince_similarities`, vince_model.py:207-242); a training forward stores them so that vince_infonce_bwd reads them back instead of
 * recomputing them.  NULL: not written. */
int vince_infonce_fwd(const vince_infonce_desc* d, const float* q, const float* inb, const float* queue,
                      float* pos, float* row_max, float* neg_sum, float* dists, float* softmax_weights,
                      float* scalars, float* logits, void* workspace, void* stream);
/* dq[B][D] += dloss/dq (atomic fp32; zero it first).  grad_scale: device pointer to the upstream scalar gradient.
 * logits: the matrix vince_infonce_fwd stored for the same operands, or NULL (recomputed with exact fp32 products: they differ from the
 * forward's split-half logits by fp32 rounding, ~2e-7 per logit, while row_max / neg_sum are the forward's -- the training path always
 * passes the stored matrix, so its gradients are those of the logits the loss was computed from).
 * wmat: optional float[B][Bk] receiving dloss/dlogit for the in-batch columns (self-similarity column-side grad). */
int vince_infonce_bwd(const vince_infonce_desc* d, const float* q, const float* inb, const float* queue,
                      const float* pos, const float* row_max, const float* neg_sum, const float* grad_scale,
                      const float* logits, float* dq, float* wmat, void* stream);

/* similarity_cross_entropy on MATERIALISED similarities [B][cols] with an arbitrary boolean mask (uint8, P positives in
 * every row) -- utils/loss_util.py:7-62 as called by VinceModel.loss with a caller-provided tensor.  dists /
 * softmax_weights are [B][P] in column order of the positives. */
int vince_sce_rows_fwd(const float* sims, const uint8_t* mask, int32_t B, int32_t cols, int32_t P,
                       float inv_temperature, float* dists, float* softmax_weights, float* row_max, float* neg_sum,
                       void* stream);
int vince_sce_rows_bwd(const float* sims, const uint8_t* mask, int32_t B, int32_t cols, int32_t P,
                       float inv_temperature, const float* row_max, const float* neg_sum, const float* grad_dists,
                       float* dsims, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Queue, momentum encoder, optimiser (K18-K21)
 */
/* StorageQueue.enqueue (utils/storage_queue.py:31-49): writes n rows at *tail with split-on-wrap (and repeated
 * laps when n > K); tail/full are HOST ints updated in place.  Index arithmetic is exact integer. */
int vince_queue_enqueue(float* queue, int64_t K, int64_t D, const float* items, int64_t n, int64_t* tail,
                        int32_t* full, void* stream);
/* theta_k = theta_k*m + (1-m)*theta_q  (VinceQueueModel.param_update, vince_model.py:587-592) over a flat range */
int vince_ema_flat(float* key, const float* query, int64_t n, float momentum, void* stream);
/* torch.optim.SGD(momentum, weight_decay), dampening 0, no nesterov (vince_solver.py:256,469) over a flat range:
 * d = g + wd*p; buf = buf*mom + d; p -= lr*buf.  grad_scale multiplies g first (1/world for DP mean). */
int vince_sgd_flat(float* param, const float* grad, float* buf, int64_t n, float lr, float momentum,
                   float weight_decay, float grad_scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Trunk engine: the ResNet-18/50 trunk (conv1..layer4, backbone_models.py:39-54 with final_layer=-2) + average
 * pool as one object that sequences the kernels above on a stream.  One engine per (arch, N, H, W, dtype).
 */
typedef struct vince_trunk* vince_trunk_t;

typedef struct vince_trunk_cfg {
    int32_t arch;   /* 18 or 50 */
    int32_t N, H, W;
    int32_t dtype;  /* VINCE_F32 / VINCE_BF16 / VINCE_F32X3 (fp32 tensors, convolutions as split-half products) */
} vince_trunk_cfg;

int vince_trunk_create(const vince_trunk_cfg* cfg, vince_trunk_t* out);
void vince_trunk_destroy(vince_trunk_t t);
/* parameter table: conv weights and BN (gamma, beta) in reference state-dict order, fc excluded */
int32_t vince_trunk_num_params(vince_trunk_t t);
int32_t vince_trunk_num_bn(vince_trunk_t t);
/* kind: 0 conv weight (shape = Co,Ci,kh,kw; memory [Co][kh][kw][Ci] = torch channels_last), 1 bn gamma, 2 bn beta */
int vince_trunk_param_info(vince_trunk_t t, int32_t idx, char* name, int32_t name_cap, int32_t* kind,
                           int32_t shape[4], int32_t* bn_index);
int vince_trunk_bn_info(vince_trunk_t t, int32_t bn_index, char* name, int32_t name_cap, int32_t* channels);
int32_t vince_trunk_out_channels(vince_trunk_t t);
int32_t vince_trunk_out_hw(vince_trunk_t t, int32_t* h, int32_t* w);
size_t vince_trunk_workspace_bytes(vince_trunk_t t);       /* activations + saved tensors + gradient scratch */
size_t vince_trunk_weight_cache_bytes(vince_trunk_t t);    /* compute-dtype weight copies (per encoder) + descriptor table */

/* fp32 master weights -> compute copies in `wcache`.  Call after every parameter update. */
int vince_trunk_prepare_weights(vince_trunk_t t, const float* const* params, void* wcache, void* stream);

/* input: float NCHW (perm optional, see vince_input_nchw_to_nhwc) or, when input_is_tiles != 0, jigsaw source
 * (N/9 images [3][srcH][srcW]).  bn_buffers: per BN {running_mean, running_var} float pointers; nbt: int64 ptrs.
 * train_bn: batch statistics + running-stat update.  save != 0: everything vince_trunk_backward reads stays in the
 * workspace; save == 0 (no backward follows) lets the engine fuse what backward would have needed apart (the Gram-statistics
 * residual join of the bottlenecks, in place on the block input).
 * Outputs: pooled float[N][C]; spatial (the trunk output, dtype NHWC) stays in the workspace:
 * vince_trunk_spatial_ptr(). */
int vince_trunk_forward(vince_trunk_t t, const float* const* params, const void* wcache, float* const* bn_running,
                        int64_t* const* bn_nbt, const float* input, const int64_t* perm, int32_t jigsaw_src_h,
                        int32_t jigsaw_src_w, void* workspace, float* pooled, int32_t train_bn, int32_t save,
                        void* stream);
const void* vince_trunk_spatial_ptr(vince_trunk_t t, const void* workspace);
/* Where the stem input lives inside `workspace` and its row layout ([N][H][row_width][4], image column w at w + left).
 * A caller may stage it itself -- e.g. straight from uint8 frames with vince_input_u8hwc_to_rows -- and then call
 * vince_trunk_forward / _forward_folded with input == NULL. */
void* vince_trunk_input_ptr(vince_trunk_t t, void* workspace, int32_t* row_width, int32_t* left);

/* vince_trunk_prepare_weights in parts: 0 = every layer (the same launch), 1 = every layer BUT conv1 (resnet.py:170), 2 = conv1 alone.
 * With the deferred stem join (vince_trunk_set_stem_event) conv1.weight is stepped last, behind its own event: the caller rebuilds the
 * compute copies of everything else beside the stem's weight gradient (part 1) and conv1's once it has been stepped (part 2). */
int vince_trunk_prepare_weights_part(vince_trunk_t t, const float* const* params, void* wcache, int32_t part, void* stream);
/* Inference with the BatchNorms FOLDED into the convolutions (eval mode, running statistics; the end-task feature
 * extraction of end_task_base_solver.py:199-212 / vince_model.py:97-117 `extract_features`): no BatchNorm pass at all.
 * _prepare_weights_folded writes w * gamma/sqrt(var+eps) (compute dtype) and the per-channel bias beta - mean*scale into
 * `wcache` (vince_trunk_weight_cache_bytes; a cache separate from the training one); call it whenever parameters or
 * running statistics change.  _forward_folded then runs conv(+bias)(+ReLU) only, the residual join in the conv3
 * epilogue; nothing is kept for a backward pass.  Outputs as vince_trunk_forward. */
int vince_trunk_prepare_weights_folded(vince_trunk_t t, const float* const* params, float* const* bn_running, void* wcache,
                                       void* stream);
int vince_trunk_forward_folded(vince_trunk_t t, const void* wcache, const float* input, const int64_t* perm,
                               int32_t jigsaw_src_h, int32_t jigsaw_src_w, void* workspace, float* pooled, void* stream);
/* Host callback of vince_trunk_backward: cb(e, user) is called on the calling thread right after bucket event e (see below) has been
 * recorded -- the data-parallel layer enqueues that bucket's gradient all-reduce behind the event at that moment, while the host is
 * still enqueueing the rest of backward (vince_amd/dp.py).  NULL clears it. */
int vince_trunk_set_bucket_callback(vince_trunk_t t, void (*cb)(int32_t, void*), void* user);
/* Deferred stem join.  With a hipEvent_t set here, vince_trunk_backward returns with `stream` waiting for every weight gradient
 * EXCEPT conv1's (resnet.py:170; the last launch of backward, ~200 us alone on the machine at ResNet-50 / B = 256) and records the
 * event behind that launch instead: the gradient of params[0] is final only after the event; everything else as before.  The caller
 * steps every other parameter while the stem's weight gradient runs (optim.FlatSGD.step(defer_stem=True)).  NULL (default): the
 * stream waits for all of them.
 * Workspace lifetime: that launch keeps READING the stem input, a gradient ring slot and the scratch area of `workspace` until the event
 * has passed.  The next vince_trunk_forward / _forward_folded / _backward on this handle makes its own stream wait for the event before it
 * touches the workspace; a caller that writes the staged input itself (vince_trunk_input_ptr) or frees / reuses the workspace for anything
 * else must wait for the event first.  Replacing or clearing the event while such a launch is pending drains it on the host. */
int vince_trunk_set_stem_event(vince_trunk_t t, void* event);
/* The mixed mode "x3f": an fp32-tensor trunk (VINCE_F32 / VINCE_F32X3 / VINCE_F32X3F) whose grad-enabled forwards ALSO leave everything a
 * backward reads -- stem input, every convolution's raw output, every activation, block outputs, ReLU masks, pool argmax bytes,
 * BatchNorm constants -- as bfloat16 in the workspace of a TWIN handle created with VINCE_BF16 for the same architecture and input
 * shape (shadow_workspace: vince_trunk_workspace_bytes(shadow) bytes).  vince_trunk_backward is then called on the TWIN with its own
 * bf16 weight cache: the forward keeps fp32-grade embeddings and loss, the backward runs the bf16 kernels on half the bytes -- the
 * arithmetic of a mixed-precision (AMP) backward.  Costs the forward one extra 2-byte write per saved element.  NULL clears.  (ABI 11) */
int vince_trunk_set_shadow(vince_trunk_t t, vince_trunk_t shadow, void* shadow_workspace);
/* n floats -> n bfloat16 (round to nearest even); n multiple of 8, both pointers 16-byte aligned.  (ABI 11) */
int vince_cast_f32_to_bf16(const float* src, void* dst, size_t n, void* stream);
/* Makes `stream` wait for a deferred stem weight gradient still in flight on this handle (no-op otherwise): for callers that write the
 * staged input themselves.  (ABI 11) */
int vince_trunk_stem_join(vince_trunk_t t, void* stream);
/* grads: float pointers parallel to params (accumulated into; zero them first).  dpooled: float[N][C].
 * Gradient buckets for data parallelism: after the backward of residual block event_blocks[e] (blocks are numbered in
 * forward order; backward visits them last to first) has been enqueued, hipEvent_t events[e] is recorded on `stream`;
 * every parameter gradient of that block and of all later blocks is final at that event, so the caller can launch
 * the RCCL all-reduce of that contiguous parameter range on a side stream while the rest of backward runs. */
int vince_trunk_backward(vince_trunk_t t, const float* const* params, const void* wcache, void* workspace,
                         const float* dpooled, float* const* grads, const int32_t* event_blocks, void* const* events,
                         int32_t n_events, void* stream);
int32_t vince_trunk_num_blocks(vince_trunk_t t);

#ifdef __cplusplus
}
#endif
#endif /* VINCE_HIP_H */
