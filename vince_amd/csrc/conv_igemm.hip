// Generalised tap convolution as an implicit GEMM on the gfx950 matrix cores.
//
// Replaces the reference's conv2d / nn.Linear call sites (models/building_blocks/resnet.py:34-50,170;
// models/vince_model.py:38-42) and their input-gradient halves.  See include/vince_hip.h for the op definition.
//
// Mapping to the hardware
//   GEMM:  D[co][pix] = sum_k  W[co][k] * X[pix][k],   k = (tap, ci) with ci fastest -> both operands are
//          K-contiguous in HBM (weights [Co][T][Ci], activations NHWC), so every global access is a 16-byte
//          chunk (8 bf16 / 4 f32) and a wavefront reads whole 64-byte row segments.
//   Tile:  128 pixels x CT (64|128) output channels per 256-thread workgroup (4 waves as 2x2), K step = 64 bytes.
//          Each wave owns 64 pixels x CT/2 channels as 32x32 MFMA tiles: v_mfma_f32_32x32x16_bf16 (bf16) or
//          4 x v_mfma_f32_32x32x2_f32 per 16-byte fragment (exact fp32).  A and B fragments are the same 16 bytes
//          per lane at row (lane&31), k-offset (lane>>5)*16 B, so the dtype only changes the MFMA, not the loads.
//   LDS:   register-staged double buffer (global -> VGPR -> ds_write_b128), rows padded 64 -> 80 B so the
//          ds_read_b128 fragment reads of 16 consecutive rows land on 16 distinct 16-B slots (conflict free).
//   Store: accumulators hold one pixel per lane and 4 consecutive channels per register group; the tile is
//          transposed through LDS and written as full 16-byte chunks along the channel axis (coalesced NHWC rows).
//          The optional BatchNorm statistics (per-channel sum / sum of squares of the values just stored) are
//          reduced in registers -> shuffles -> LDS -> one fp64 atomic per channel per workgroup.
//   Grid:  1-D, remapped so that each XCD (private L2) owns a contiguous range of tiles; channel tiles of the
//          same pixel tile are adjacent and re-read the activation tile from that L2.
#include "conv_igemm_impl.h"


extern "C" int vince_conv_igemm(const vince_conv_desc* dd, int dtype, const void* in, const void* w, void* out,
                                const vince_conv_epi* epi, void* stream) {
    vince_conv_epi e;
    if (epi) e = *epi; else memset(&e, 0, sizeof(e));
    VINCE_CHECK_ARG(dd && in && w && out, VINCE_E_ARG, "vince_conv_igemm: null pointer");
    if (e.replicas <= 0 || e.replicas > VINCE_STATS_REPLICAS) e.replicas = VINCE_STATS_REPLICAS;
    VINCE_CHECK_ARG(!e.acc_mask || (e.flags & VINCE_EPI_ACCUMULATE), VINCE_E_ARG, "vince_conv_igemm: acc_mask needs VINCE_EPI_ACCUMULATE");
    VINCE_CHECK_ARG(!e.bnred.y || (e.bnred.mean && e.bnred.invstd && e.bnred.sums && !e.stats), VINCE_E_ARG,
                    "vince_conv_igemm: bnred needs y, mean, invstd and sums, and excludes stats");
    VINCE_CHECK_ARG(!e.out_mask || (!e.out_scale && !e.id_scale && !e.bnred.y), VINCE_E_ARG,
                    "vince_conv_igemm: out_mask belongs to the gradient epilogue and excludes bnred / the forward join");
    VINCE_CHECK_ARG(!e.id_scale == !e.id_shift, VINCE_E_ARG, "vince_conv_igemm: id_scale and id_shift come together");
    {
        const bool fwd_join = e.out_scale || e.id_scale;
        VINCE_CHECK_ARG(!(e.out2 || e.raw2 || e.mask2) ||
                        ((dtype == VINCE_F32 || dtype == VINCE_F32X3H) && !e.bnred.y && !e.out_mask && !e.acc_mask && ((uintptr_t)e.out2 & 15) == 0 &&
                         ((uintptr_t)e.raw2 & 15) == 0 && (fwd_join || !(e.flags & VINCE_EPI_ACCUMULATE)) && dd->Co % 8 == 0), VINCE_E_ARG,
                        "vince_conv_igemm: the bf16 shadows (out2 / raw2 / mask2) belong to the forward epilogues of fp32-store launches");
        VINCE_CHECK_ARG((!e.raw2 && !e.mask2) || fwd_join, VINCE_E_ARG, "vince_conv_igemm: raw2 / mask2 belong to the forward residual join");
        VINCE_CHECK_ARG(!e.raw2 == !e.raw2_mean, VINCE_E_ARG, "vince_conv_igemm: raw2 and raw2_mean come together");
    }
    VINCE_CHECK_ARG((!e.out_scale && !e.id_scale) || ((e.flags & VINCE_EPI_ACCUMULATE) && !e.acc_mask && !e.bnred.y && !e.stats), VINCE_E_ARG,
                    "vince_conv_igemm: out_scale / id_scale need VINCE_EPI_ACCUMULATE and exclude acc_mask, bnred and stats");
    VINCE_CHECK_ARG(!e.bnred.mask_scale == !e.bnred.mask_shift, VINCE_E_ARG,
                    "vince_conv_igemm: bnred mask_scale and mask_shift come together");
    VINCE_CHECK_ARG(dtype == VINCE_F32 || dtype == VINCE_BF16 || dtype == VINCE_F32X3H || dtype == VINCE_F32X3B || dtype == VINCE_F32X1B, VINCE_E_DTYPE,
                    "vince_conv_igemm: bad dtype %d", dtype);
    const bool f32_store = dtype != VINCE_BF16;    // the split-half types are fp32 tensors
    VINCE_CHECK_ARG(!e.in2 || (dd->TA * dd->TB >= 2 && dd->Cs == 0 && e.in2_channels > 0 && e.in2_channels <= dd->Ci &&
                               e.in2_channels % (f32_store ? 4 : 8) == 0 && ((uintptr_t)e.in2 & 15) == 0), VINCE_E_ARG,
                    "vince_conv_igemm: in2 is the input of the last of at least two taps, in2_channels <= Ci, 16-byte aligned");
    const int in2_rep = (e.in2 && e.in2_repeat > 1) ? e.in2_repeat : 1;
    VINCE_CHECK_ARG(in2_rep == 1 || (!f32_store && e.in2_channels * in2_rep <= dd->Ci && e.in2_channels % 32 == 0 &&
                                     ((e.in2_channels / 8) & (e.in2_channels / 8 - 1)) == 0), VINCE_E_ARG,
                    "vince_conv_igemm: in2_repeat needs bf16, in2_repeat * in2_channels <= Ci and in2_channels / 8 a power of two >= 4");
    const vince_conv_desc& d = *dd;
    const int CH = f32_store ? 4 : 8;
    VINCE_CHECK_ARG(d.N > 0 && d.Hi > 0 && d.Wi > 0 && d.Ho > 0 && d.Wo > 0 && d.Co > 0 && d.Ci > 0, VINCE_E_SHAPE,
                    "vince_conv_igemm: non-positive dimension");
    VINCE_CHECK_ARG(d.Ci % CH == 0, VINCE_E_SHAPE, "vince_conv_igemm: Ci=%d not a multiple of %d", d.Ci, CH);
    if (d.Cs > 0) {   // packed row taps: every tap start must stay 16-byte aligned and inside its input row
        const int eb = f32_store ? 4 : 2;
        VINCE_CHECK_ARG(d.TB == 1 && d.Cs < d.Ci && d.Ci % d.Cs == 0 && d.Kw > 0 && d.Kw <= d.Ci / d.Cs, VINCE_E_SHAPE,
                        "vince_conv_igemm: packed row taps need TB=1, Cs | Ci, 0 < Kw <= Ci/Cs");
        VINCE_CHECK_ARG((d.Cs * eb) % 8 == 0 && (d.sw * d.Cs * eb) % 16 == 0 && (d.dw0 * d.Cs * eb) % 16 == 0 &&
                        (d.Wi * d.Cs * eb) % 16 == 0, VINCE_E_ALIGN, "vince_conv_igemm: packed row taps must start 16-byte aligned");
        VINCE_CHECK_ARG(d.dw0 >= 0 && (d.Wo - 1) * d.sw + d.dw0 + d.Ci / d.Cs <= d.Wi, VINCE_E_SHAPE,
                        "vince_conv_igemm: packed row taps must stay inside the (padded) input row");
    }
    VINCE_CHECK_ARG(d.Co % CH == 0, VINCE_E_SHAPE, "vince_conv_igemm: Co=%d not a multiple of %d", d.Co, CH);
    VINCE_CHECK_ARG(d.TA >= 1 && d.TB >= 1 && d.TB <= 8 && d.TA * d.TB <= 64, VINCE_E_SHAPE,
                    "vince_conv_igemm: tap grid %dx%d unsupported", d.TA, d.TB);
    VINCE_CHECK_ARG((long long)d.N * d.Ho * d.Wo < (1ll << 31), VINCE_E_SHAPE, "vince_conv_igemm: too many output pixels");
    VINCE_CHECK_ARG((((uintptr_t)in | (uintptr_t)w | (uintptr_t)out) & 15) == 0, VINCE_E_ALIGN,
                    "vince_conv_igemm: pointers must be 16-byte aligned");
    ConvParams p;
    p.d = d;
    const int T = d.TA * d.TB, cpt = d.Ci / CH;
    if (T == 1) {
        p.log2_cpt = 31;
        p.cpt_mask = 0x7fffffff;
    } else {
        VINCE_CHECK_ARG((cpt & (cpt - 1)) == 0, VINCE_E_SHAPE,
                        "vince_conv_igemm: Ci/%d = %d must be a power of two for multi-tap convs", CH, cpt);
        int l = 0;
        while ((1 << l) < cpt) ++l;
        p.log2_cpt = l;
        p.cpt_mask = cpt - 1;
    }
    p.total_chunks = T * cpt;
    if (e.in2) p.total_chunks = (T - 1) * cpt + e.in2_channels * in2_rep / CH;   // the last tap reads the (narrower) second tensor
    p.log2_cpt2 = 31;
    p.cpt2_mask = 0x7fffffff;
    p.log2_tapid = p.log2_cpt;
    if (in2_rep > 1) {
        int l = 0;
        while ((1 << l) < e.in2_channels / CH) ++l;
        p.log2_cpt2 = l;
        p.cpt2_mask = e.in2_channels / CH - 1;
        p.log2_tapid = l;
    }
    p.nkt = (p.total_chunks + 3) / 4;
    p.M = d.N * d.Ho * d.Wo;
    p.tb_mul = (65536 + d.TB - 1) / d.TB;
    p.div_howo = make_fastdiv((uint32_t)(d.Ho * d.Wo));
    p.div_wo = make_fastdiv((uint32_t)d.Wo);
    p.in = in; p.w = w; p.out = out; p.e = e;
    p.variant = 0;
    p.cs = d.Cs > 0 ? d.Cs : d.Ci;
    p.in2 = e.in2;
    p.cs2 = e.in2_channels;
    p.in2_bytes = 0;
    p.kt_per_split = 0;
    p.presplit = (e.flags & VINCE_EPI_IN_HALF_PAIRS) ? 1 : 0;
    VINCE_CHECK_ARG(!p.presplit || (dtype == VINCE_F32X3H && d.Cs == 0 && d.Ci % 16 == 0), VINCE_E_ARG,
                    "vince_conv_igemm: VINCE_EPI_IN_HALF_PAIRS needs dtype VINCE_F32X3H, plain taps and Ci a multiple of 16");
    p.ablate = 0;
#ifdef VINCE_MEASURE
    static int ablate = VINCE_MEASURE_KNOB("conv_ablate", 0);
    p.ablate = ablate;
#endif
    {
        const unsigned long long esz = f32_store ? 4 : 2;
        const unsigned long long ib = (unsigned long long)d.N * d.Hi * d.Wi * p.cs * esz, wb = (unsigned long long)d.Co * d.WT * d.Ci * esz;
        p.in_bytes = ib < 0x7ff00000ull ? (uint32_t)ib : 0;   // the direct-to-LDS path addresses with 31-bit offsets
        if (e.in2) {
            const unsigned long long ib2 = (unsigned long long)d.N * d.Hi * d.Wi * e.in2_channels * esz;
            p.in2_bytes = ib2 < 0x7ff00000ull ? (uint32_t)ib2 : 0;
            if (!p.in2_bytes) p.in_bytes = 0;
        }
        p.w_bytes = wb < 0x7ff00000ull ? (uint32_t)wb : 0;
    }
    p.ptiles = (p.M + PT - 1) / PT;
    // 64-channel tiles for short reductions: such layers are HBM-bound and the smaller accumulator footprint buys
    // occupancy (5 waves/SIMD vs 3), which is what a streaming kernel needs
    static int ct64_max_k = VINCE_MEASURE_KNOB("ct64_max_k", 0);
    bool narrow = d.Co <= 64 || (d.TA * d.TB * d.Ci <= ct64_max_k);
    // tiny-M GEMMs (the projection MLP: 256 rows -> 2 pixel tiles): 64-channel tiles double the workgroup count
    if (!narrow && (long)p.ptiles * ((d.Co + 127) / 128) < 128) narrow = true;
    const int CT = narrow ? 64 : 128;
    p.ctiles = (d.Co + CT - 1) / CT;
    hipStream_t s = (hipStream_t)stream;
    void* tok = nullptr;
    if (vince_profile_enabled()) {
        // algorithmic FLOPs: the stem's input channels are padded 3 -> CH; count the 3 real ones
        // (the stem: 3 real channels behind the padding, and Kw real pixels per packed row tap)
        const double ci_alg = d.Cs > 0 ? 3.0 * d.Kw : ((d.Ci == CH && T > 1) ? 3.0 : (double)d.Ci);
        vince_profile_begin_launch(0, 2.0 * p.M * d.Co * T * ci_alg, stream, &tok);
        vince_profile_set_dims(tok, p.M, d.Co, T * d.Ci, T, d.sh * 10 + d.osh, e.flags);
    }
    int rc;
    // gradient epilogue instantiation (in2: the BatchNorm-backward algebra's input gradient -- its bias goes onto the fp32 accumulators there)
    const bool bwd = (e.flags & VINCE_EPI_ACCUMULATE) || e.bnred.y || e.out_mask || e.in2;
    const bool join = e.out_scale || e.id_scale;   // forward residual join with known BatchNorm constants
    // The 8-wavefront 256 x 256 core (conv_m8.hip) takes the long bf16 reductions with 256-channel output tiles: the 3x3 and wide
    // 1x1 convolutions of layer3 / layer4 and their input gradients.  VINCE_M8=0 keeps everything on this file's tiles (the
    // cross-check of the parity tests), VINCE_M8_MIN_K moves the threshold.
    static const int m8_on = vince_knob("m8", 1);
    static const int m8_min_k = vince_knob("m8_min_k", 1024);
    // (96 since late round 5: layer4's 3x3 and 2048 -> 512 -- 98 tiles of 256 x 256 -- moved here from the 128-pixel tiles; with
    // big_min_k 512 / s3_min_k 1024 in conv_igemm_impl.h -0.2 ms per step in two same-box sweeps, profiles/r05_knob_sweep.txt)
    static const int m8_min_tiles = vince_knob("m8_min_tiles", 96);
    rc = VINCE_M8_NOT_ELIGIBLE;
    if (m8_on && !e.in2 && dtype == VINCE_BF16 && d.Co % 256 == 0 && T * d.Ci >= m8_min_k &&
        (long)((p.M + 255) / 256) * (d.Co / 256) >= m8_min_tiles)
        rc = vince_conv_m8_launch(p, join ? 2 : (bwd ? 1 : 0), s);
    if (rc != VINCE_M8_NOT_ELIGIBLE) {
        if (tok) {
            vince_profile_set_tag(tok, VINCE_TAG_M8_FWD + (bwd ? 1 : 0));
            vince_profile_end_launch(tok, stream);
        }
        return rc;
    }
    if (dtype == VINCE_F32X3H || dtype == VINCE_F32X3B || dtype == VINCE_F32X1B) {
        rc = vince_conv_igemm_x3_launch(p, dtype, join ? 2 : (bwd ? 1 : 0), narrow, s);
    } else if (dtype == VINCE_F32) {
        if (join) rc = narrow ? launch<float, 64, 2>(p, s) : launch<float, 128, 2>(p, s);
        else if (bwd) rc = narrow ? launch<float, 64, 1>(p, s) : launch<float, 128, 1>(p, s);
        else rc = narrow ? launch<float, 64, 0>(p, s) : launch<float, 128, 0>(p, s);
    } else {
        if (join) rc = narrow ? launch<bf16_t, 64, 2>(p, s) : launch<bf16_t, 128, 2>(p, s);
        else if (bwd) rc = narrow ? launch<bf16_t, 64, 1>(p, s) : launch<bf16_t, 128, 1>(p, s);
        else rc = narrow ? launch<bf16_t, 64, 0>(p, s) : launch<bf16_t, 128, 0>(p, s);
    }
    if (tok) {
        // one tag per kernel symbol: [dtype][64ch | 128ch] x [128px | 256px][fwd | bwd epilogue]; the register-staged
        // fallback (never taken at the benchmark sizes) is counted with the 128-pixel tile of its shape
        const int shape = (narrow ? 0 : 2) + (p.variant == 1 ? 1 : 0);   // 64ch x 128px, 64ch x 256px, 128ch x 128px, 128ch x 256px
        vince_profile_set_tag(tok, (f32_store ? 0 : 8) + shape * 2 + (bwd ? 1 : 0));
        vince_profile_end_launch(tok, stream);
    }
    return rc;
}
