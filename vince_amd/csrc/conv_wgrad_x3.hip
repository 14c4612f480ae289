// Weight gradient as split-half products (VINCE_F32X3B): fp32 tensors, every product as three bfloat16 MFMAs of hi / lo halves.
//   dw[co][tap][ci] += sum_pix dy[pix][co] * x[pix @ tap][ci]
// Same reference call sites as conv_wgrad.hip (the weight-gradient half of conv2d's autograd, models/building_blocks/resnet.py:34-50,170
// under loss.backward(), solvers/vince_solver.py:463-468).
//
// conv_wgrad_dlds_kernel<x3b_t> (conv_wgrad.hip) runs the same arithmetic but decodes a pixel per LDS-DMA instruction; with the
// matrix work cut to 3/16 of exact fp32 that address arithmetic is what bounds it (3x3 layers 520-590 us against 84 us in bf16).
// This kernel takes conv_wgrad_tr's addressing: a tile is stored as SUB-TILES of 32 pixel rows x 32 channels (128-byte fp32 rows),
// one LDS-DMA instruction per wavefront fills 8 rows of one sub-tile, so a lane keeps ONE pixel row for the whole launch -- the
// pixel -> (image, row, column) decode runs once per slice and lane, the tap of a sub-tile is wave-uniform, sub-tiles of one tap
// differ by an instruction offset, dy and 1x1 addresses are "offset += step", and the pixel range of a split is cut by dy's buffer
// descriptor.  Fragments: MFMA lane (column lane & 31, K group lane >> 5) reads its column's 8 consecutive pixels as 8 dwords one row
// apart (conflict-free: a half-wave covers one 128-byte row), splits them (common.h x3_split) and feeds three v_mfma_f32_32x32x16_bf16.
#include "conv_wgrad.h"

namespace {

using vince_wgrad::WgradParams;
constexpr int SL = vince_wgrad::TR_SLICE;    // 32 pixels per slice
constexpr int SUB = SL * 128;                // bytes of one sub-tile: 32 rows x 32 floats

static __device__ __forceinline__ void dma16_imm(uint32_t lds_addr_uniform, uint32_t voff, v4i_t rsrc, int imm) {
    // imm in {0, 128, 256, 384}: sub-tiles of the same pixel row and tap are 32 channels = 128 bytes apart; the instruction offset is
    // added to the LDS address as well as to the memory address, so M0 carries the destination minus it
    lds_addr_uniform -= (uint32_t)imm;
    if (imm == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr_uniform), "v"(voff), "s"(rsrc) : "memory");
    else if (imm == 128) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:128 lds" ::"s"(lds_addr_uniform), "v"(voff), "s"(rsrc) : "memory");
    else if (imm == 256) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:256 lds" ::"s"(lds_addr_uniform), "v"(voff), "s"(rsrc) : "memory");
    else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:384 lds" ::"s"(lds_addr_uniform), "v"(voff), "s"(rsrc) : "memory");
}

static __device__ __forceinline__ uint32_t fdiv(uint32_t n, uint32_t mul, uint32_t shift_m1) {   // d >= 2
    const uint32_t t = __umulhi(n, mul);
    return (t + ((n - t) >> 1)) >> shift_m1;
}

// 4 wavefronts as 2 (dy channel halves) x 2 (x column halves); wavefront tile CT/2 x NT/2 = (CT/64) x (NT/64) MFMA blocks.
// RUN = 0: 1x1, stride 1, no padding -- the input pixel is the output pixel.  RUN > 0: consecutive x sub-tiles that share a tap
// (min(NT, Ci) / 32).
// XT: x3b_t (three MFMAs per block) or x1b_t (VINCE_F32X1B: the hi halves only, one MFMA per block)
template <typename XT, int CT, int NT, int STAGES, int RUN>
__global__ __launch_bounds__(256, 2) void conv_wgrad_x3_kernel(const WgradParams p) {
    constexpr bool LINEAR = RUN == 0;
    constexpr int YS = CT / 32, XS = NT / 32;                 // sub-tiles per operand
    constexpr int STAGE = (YS + XS) * SUB;
    constexpr int CJ = CT / 64, NJ = NT / 64;
    constexpr int PER = YS + XS;                              // DMA instructions per thread per slice
    static_assert(STAGES >= 2 && YS <= 4 && XS <= 4, "tile");
    __shared__ __attribute__((aligned(16))) unsigned char smem[STAGES * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 1, wn = wave >> 1;
    uint32_t bx = blockIdx.x, by = blockIdx.y;
    if (p.xcd_group) {   // all (channel, tap) tiles of one pixel range on ONE XCD: its L2 serves their shared dy / x rows
        const uint32_t lid = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
        bx = lid % gridDim.x;
        by = lid / gridDim.x;
    }
    const int ctile = bx % p.ctiles, ntile = bx / p.ctiles;
    const int c0 = ctile * CT, n0 = ntile * NT;
    const vince_conv_desc& d = p.d;
    const int kt_begin = by * p.kt_per_split;
    const int kt_end = min(kt_begin + p.kt_per_split, p.nkt_total);
    if (kt_begin >= kt_end) return;
    const int nkt = kt_end - kt_begin;
    const uint32_t pix_begin = (uint32_t)kt_begin * SL;
    const uint32_t pix_end = min((uint32_t)kt_end * SL, (uint32_t)p.M);

    // dy rows past the split (and past M) read as zero: the descriptor ends there; a zero dy row silences whatever x holds
    const v4i_t rsrc_y = make_rsrc(p.dy, pix_end * (uint32_t)d.Co * 4u);
    const v4i_t rsrc_x = make_rsrc(p.in, LINEAR ? pix_end * (uint32_t)d.Ci * 4u : p.in_bytes);
    const uint32_t smem_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    constexpr uint32_t OOB = 0x80000000u;

    // ---- DMA role of this lane: pixel row (tid >> 3) of every slice, 16-byte slot (tid & 7) of the 128-byte sub-tile row
    const int drow = tid >> 3, dslot = tid & 7;
    const uint32_t dcolb = (uint32_t)dslot * 16u;
    uint32_t yoff = (pix_begin + drow) * (uint32_t)d.Co * 4u + (uint32_t)c0 * 4u + dcolb;
    const uint32_t ystep = (uint32_t)SL * d.Co * 4u;
    uint32_t xoff = 0, xpix = pix_begin + drow;
    const uint32_t xstep = (uint32_t)SL * d.Ci * 4u;
    constexpr int NRUN = LINEAR ? 1 : XS / RUN;
    int sdh[NRUN], sdw[NRUN];
    uint32_t scol[NRUN];
    if constexpr (LINEAR) {
        xoff = (pix_begin + drow) * (uint32_t)d.Ci * 4u + (uint32_t)n0 * 4u + dcolb;
    } else {
#pragma unroll
        for (int e = 0; e < NRUN; ++e) {
            const int n = n0 + e * RUN * 32;
            const int tap = (d.TA * d.TB == 1) ? 0 : (n >> p.log2_ci);
            const int ci0 = n - (tap << ((d.TA * d.TB == 1) ? 0 : p.log2_ci));
            const int ta = (int)(((uint32_t)tap * p.tb_mul) >> 16), tb = tap - ta * d.TB;
            sdh[e] = d.dh0 + ta * d.dhs;
            sdw[e] = d.dw0 + tb * d.dws;
            scol[e] = (uint32_t)ci0 * 4u;
        }
    }
    const uint32_t cs4 = (uint32_t)p.cs * 4u;
    const uint32_t hw_mul = p.div_howo.mul, hw_sh = p.div_howo.shift - 1, hw_d = p.div_howo.d;
    const uint32_t w_mul = p.div_wo.mul, w_sh = p.div_wo.shift - 1, w_d = p.div_wo.d;

    auto issue_slice = [&](const int stage) {
        const uint32_t ys = __builtin_amdgcn_readfirstlane(smem_base + stage * STAGE + wave * 1024);
#pragma unroll
        for (int e = 0; e < YS; ++e) dma16_imm(ys + e * SUB, yoff, rsrc_y, e * 128);
        yoff += ystep;
        const uint32_t xs = ys + YS * SUB;
        if constexpr (LINEAR) {
#pragma unroll
            for (int e = 0; e < XS; ++e) dma16_imm(xs + e * SUB, xoff, rsrc_x, e * 128);
            xoff += xstep;
        } else {
            const uint32_t n = fdiv(xpix, hw_mul, hw_sh);
            const uint32_t rem = xpix - n * hw_d;
            const uint32_t ho = fdiv(rem, w_mul, w_sh);
            const uint32_t wo = rem - ho * w_d;
            const int bh = (int)(ho * d.sh), bw = (int)(wo * d.sw);
            const uint32_t nimg = n * (uint32_t)(d.Hi * d.Wi);
            xpix += SL;
#pragma unroll
            for (int q = 0; q < NRUN; ++q) {
                const int hi = bh + sdh[q], wi = bw + sdw[q];
                const bool ok = (unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi;
                const uint32_t off = ok ? (nimg + (uint32_t)(hi * d.Wi + wi)) * cs4 + scol[q] + dcolb : OOB;
#pragma unroll
                for (int e = 0; e < RUN; ++e) dma16_imm(xs + (q * RUN + e) * SUB, off, rsrc_x, e * 128);
            }
        }
    };

    // ---- fragment addresses: column (lane & 31) of an MFMA block = float (lane & 31) of a sub-tile row; K group (lane >> 5) = rows 8..15
    const uint32_t fbase = (uint32_t)((lane >> 5) * 8 * 128 + (lane & 31) * 4);
    uint32_t ya[CJ], xa[NJ];
#pragma unroll
    for (int j = 0; j < CJ; ++j) ya[j] = (uint32_t)((wc * (CT / 64) + j) * SUB) + fbase;
#pragma unroll
    for (int i = 0; i < NJ; ++i) xa[i] = (uint32_t)((YS + wn * (NT / 64) + i) * SUB) + fbase;
    // 8 consecutive pixels of this lane's column (K step ks = rows 16 ks ..), split into hi / lo bfloat16 halves
    auto frag = [&](const unsigned char* base, uint4& hi, uint4& lo) {
        const uint4 f0 = make_uint4(*(const uint32_t*)base, *(const uint32_t*)(base + 128), *(const uint32_t*)(base + 256), *(const uint32_t*)(base + 384));
        const uint4 f1 = make_uint4(*(const uint32_t*)(base + 512), *(const uint32_t*)(base + 640), *(const uint32_t*)(base + 768), *(const uint32_t*)(base + 896));
        x3_split<XT, false>(f0, f1, hi, lo);
    };

    f32x16_t acc[CJ][NJ];
#pragma unroll
    for (int j = 0; j < CJ; ++j)
#pragma unroll
        for (int i = 0; i < NJ; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

#pragma unroll
    for (int st = 0; st < STAGES - 1; ++st) issue_slice(st);
    wait_vmcnt<(STAGES - 2) * PER>();
    __builtin_amdgcn_s_barrier();
    for (int it0 = 0; it0 < nkt; it0 += STAGES) {
#pragma unroll
        for (int s = 0; s < STAGES; ++s) {
            if (it0 + s >= nkt) break;
            issue_slice((s + STAGES - 1) % STAGES);               // the stage whose slice was multiplied in the previous iteration
            const unsigned char* sb = smem + s * STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 ah[CJ], al[CJ], bh[NJ], bl[NJ];
#pragma unroll
                for (int j = 0; j < CJ; ++j) frag(sb + ya[j] + ks * 2048, ah[j], al[j]);
#pragma unroll
                for (int i = 0; i < NJ; ++i) frag(sb + xa[i] + ks * 2048, bh[i], bl[i]);
#pragma unroll
                for (int j = 0; j < CJ; ++j)
#pragma unroll
                    for (int i = 0; i < NJ; ++i) x3_mma<XT>(ah[j], al[j], bh[i], bl[i], acc[j][i]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wait_vmcnt<(STAGES - 2) * PER>();                     // this thread's share of the NEXT slice has landed
            __builtin_amdgcn_s_barrier();                         // ... everyone's has; and everyone is done reading this one
        }
    }
    wait_vmcnt<0>();

    // ---- the partial tile into dw: fp32 atomics (32 consecutive floats per row per instruction), or plain stores into this split's slab
    const int T_ = d.TA * d.TB;
    float* const dst = p.slab ? p.slab + (size_t)by * p.slab_stride : p.dw;
    const uint32_t row_stride = (uint32_t)d.WT * (uint32_t)p.Ci_dw;
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const int n = n0 + (wn * (NT / 64) + i) * 32 + (lane & 31);
        const int tp = (T_ == 1) ? 0 : (n >> p.log2_ci);
        const int ci = (T_ == 1) ? n : (n & ((1 << p.log2_ci) - 1));
        const int a = (int)(((uint32_t)tp * p.tb_mul) >> 16), b = tp - a * d.TB;
        const uint32_t col = (uint32_t)(d.wt0 + a * d.wta + b * d.wtb) * (uint32_t)p.Ci_dw + (uint32_t)ci;
        if (ci >= p.Ci_dw) continue;
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const uint32_t co0 = (uint32_t)(c0 + (wc * (CT / 64) + j) * 32 + 4 * (lane >> 5));
            float* const base = dst + (size_t)co0 * row_stride + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* const q = base + (uint32_t)((r & 3) + 8 * (r >> 2)) * row_stride;
                if (p.slab) *q = acc[j][i][r];
                else unsafeAtomicAdd(q, acc[j][i][r]);
            }
        }
    }
}

template <typename XT, int CT, int NT, int STAGES>
int launch_x3(const WgradParams& p, int splits, hipStream_t stream) {
    const dim3 grid(p.ctiles * p.ntiles, splits), block(256);
    constexpr int XS = NT / 32;
    const int run = p.linear_x ? 0 : (p.d.Ci / 32 < XS ? p.d.Ci / 32 : XS);
    if (run == 0) hipLaunchKernelGGL((conv_wgrad_x3_kernel<XT, CT, NT, STAGES, 0>), grid, block, 0, stream, p);
    else if (run == 2) hipLaunchKernelGGL((conv_wgrad_x3_kernel<XT, CT, NT, STAGES, 2>), grid, block, 0, stream, p);
    else if constexpr (XS >= 4) {
        if (run == 4) hipLaunchKernelGGL((conv_wgrad_x3_kernel<XT, CT, NT, STAGES, 4>), grid, block, 0, stream, p);
        else return VINCE_E_SHAPE;
    } else return VINCE_E_SHAPE;
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

template <typename XT>
int launch_x3_tile(const WgradParams& p, int ct, int nt, int splits, hipStream_t stream) {
    // 128 x 128: 32 KB per slice, two stages (64 KB: two workgroups per CU); the narrower tiles take three
    if (ct == 128 && nt == 128) return launch_x3<XT, 128, 128, 2>(p, splits, stream);
    if (ct == 128 && nt == 64) return launch_x3<XT, 128, 64, 3>(p, splits, stream);
    if (ct == 64 && nt == 128) return launch_x3<XT, 64, 128, 3>(p, splits, stream);
    if (ct == 64 && nt == 64) return launch_x3<XT, 64, 64, 3>(p, splits, stream);
    vince_set_error("wgrad_x3_launch: no %d x %d tile", ct, nt);
    return VINCE_E_SHAPE;
}

}  // namespace

namespace vince_wgrad {

// Eligible: both tensors within a 2 GB descriptor, plain (unpacked) taps, Co and T * Ci multiples of 64 with Ci itself a multiple of
// 64 (a sub-tile never straddles a tap), dw as wide as the descriptor's Ci, images of at least one slice.
void wgrad_x3_tile(const WgradParams& p, int* ct, int* nt) {
    const vince_conv_desc& d = p.d;
    *ct = *nt = 0;
    const int ntot = d.TA * d.TB * d.Ci;
    if (!(p.in_bytes && p.dy_bytes) || p.variant != 0 || d.Cs != 0 || d.Co % 64 || d.Ci % 64 || p.Ci_dw != d.Ci) return;
    if (d.Ho * d.Wo < 2 || d.Wo < 2) return;                       // fdiv wants divisors >= 2
    if ((unsigned long long)p.M * d.Co * 4 + (1u << 20) >= 0x7ff00000ull) return;
    static const long forced = vince_knob("x3_wgrad_tile", 0);   // cross-check / measurement switch: ct * 1000 + nt (64 / 128 each)
    if (forced) {
        *ct = (int)(forced / 1000);
        *nt = (int)(forced % 1000);
        if ((*ct == 64 || *ct == 128) && (*nt == 64 || *nt == 128) && d.Co % *ct == 0 && ntot % *nt == 0) return;
    }
    *ct = d.Co % 128 ? 64 : 128;
    *nt = ntot % 128 ? 64 : 128;
}

int wgrad_x3_launch(const WgradParams& p, int ct, int nt, int splits, bool single, hipStream_t stream) {
    return single ? launch_x3_tile<x1b_t>(p, ct, nt, splits, stream) : launch_x3_tile<x3b_t>(p, ct, nt, splits, stream);
}

}  // namespace vince_wgrad
