// The implicit-GEMM convolution for the split-half element types (VINCE_F32X3H / VINCE_F32X3B): fp32 tensors in HBM and LDS, every
// product as three half-precision MFMAs of hi / lo halves (common.h: x3_split / x3_mma).  Same kernels, tiles and epilogues as
// conv_igemm.hip -- this translation unit only instantiates them for x3h_t (IEEE half halves: the forward launches of the reference's
// conv2d call sites, models/building_blocks/resnet.py:34-50,170) and x3b_t (bfloat16 halves: their input gradients).
#include "conv_igemm_impl.h"

int vince_conv_igemm_x3_launch(vince_conv::ConvParams& p, int dtype, int mode, bool narrow, hipStream_t s) {
    if (dtype == VINCE_F32X3H) {
        if (mode == 2) return narrow ? launch_x3<x3h_t, 64, 2>(p, s) : launch_x3<x3h_t, 128, 2>(p, s);
        if (mode == 1) return narrow ? launch_x3<x3h_t, 64, 1>(p, s) : launch_x3<x3h_t, 128, 1>(p, s);
        return narrow ? launch_x3<x3h_t, 64, 0>(p, s) : launch_x3<x3h_t, 128, 0>(p, s);
    }
    if (dtype == VINCE_F32X1B) {   // the bfloat16 hi halves only (the gradient launches of VINCE_F32X3F): one MFMA per block
        if (mode == 2) return narrow ? launch_x3<x1b_t, 64, 2>(p, s) : launch_x3<x1b_t, 128, 2>(p, s);
        if (mode == 1) return narrow ? launch_x3<x1b_t, 64, 1>(p, s) : launch_x3<x1b_t, 128, 1>(p, s);
        return narrow ? launch_x3<x1b_t, 64, 0>(p, s) : launch_x3<x1b_t, 128, 0>(p, s);
    }
    if (mode == 2) return narrow ? launch_x3<x3b_t, 64, 2>(p, s) : launch_x3<x3b_t, 128, 2>(p, s);
    if (mode == 1) return narrow ? launch_x3<x3b_t, 64, 1>(p, s) : launch_x3<x3b_t, 128, 1>(p, s);
    return narrow ? launch_x3<x3b_t, 64, 0>(p, s) : launch_x3<x3b_t, 128, 0>(p, s);
}
