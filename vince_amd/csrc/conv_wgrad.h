// Shared between the weight-gradient translation units (conv_wgrad.hip: dispatch, the fp32 / fallback kernels; conv_wgrad_tr.hip:
// the bf16 kernel that carries the step).
#pragma once
#include "common.h"

namespace vince_wgrad {

struct WgradParams {
    vince_conv_desc d;
    int log2_cpt, cpt_mask, log2_ci, total_nchunks, M, nkt_total, kt_per_split, ctiles, ntiles, Ci_dw, variant;
    uint32_t tb_mul;
    FastDiv div_howo, div_wo;
    const void* in;
    const void* dy;
    float* dw;
    uint32_t in_bytes, dy_bytes;   // descriptor ranges for the direct-to-LDS variants (0 = tensor too large)
    int linear_x;                  // 1x1 / stride 1 / no padding: the input pixel IS the output pixel
    int xcd_group;                 // remap workgroups so that the tiles of one pixel range share an XCD
    int cs;                        // element stride between input pixels (= Ci unless the descriptor packs row taps)
    float* slab;                   // deterministic mode: split `by` STORES its partial tile into slab + by * slab_stride (dw layout) and
    size_t slab_stride;            // wgrad_reduce_kernel adds the slabs into dw in split order; null = fp32 atomics straight into dw
    int ablate;                    // measurement build only (VINCE_WGRAD_ABLATE): 1 no atomics, 2 no main loop, 4 no DMA, 8 no LDS reads, 16 no MFMA
};

constexpr int TR_SLICE = 32;       // pixels per slice of conv_wgrad_tr_kernel

// Tile the bf16 transpose-read kernel would take for this layer (0 x 0: not eligible -- the caller keeps its own kernels).
void wgrad_tr_tile(const WgradParams& p, int* ct, int* nt);
// Launch with p.ctiles / p.ntiles / p.nkt_total (slices of TR_SLICE pixels) / p.kt_per_split set for that tile.
int wgrad_tr_launch(const WgradParams& p, int ct, int nt, int splits, hipStream_t stream);

// conv_wgrad_x3.hip: the split-half (VINCE_F32X3B) weight gradient with conv_wgrad_tr's addressing; same contract as the pair above
void wgrad_x3_tile(const WgradParams& p, int* ct, int* nt);
int wgrad_x3_launch(const WgradParams& p, int ct, int nt, int splits, bool single, hipStream_t stream);   // single: VINCE_F32X1B (hi halves only)

// dst[i] += slab[0][i] + slab[1][i] + ... (n4 float4 columns, `stride` floats between slabs) in a fixed order: run-to-run identical
int slab_reduce(const float* slab, int splits, size_t stride, float* dst, size_t n4, hipStream_t stream);

}  // namespace vince_wgrad
