// bf16 weight gradient, the kernel that carries the step:  dw[co][tap][ci] += sum_pix dy[pix][co] * x[pix @ tap][ci]
//
// Same reference call sites as conv_wgrad.hip (the weight-gradient half of conv2d's autograd, resnet.py:34-50,170 under
// loss.backward(), solvers/vince_solver.py:463-468).  A TN GEMM whose reduction axis is the output pixel: both operands sit in HBM
// with the NON-reduced axis (channels) contiguous, so the tiles are staged [pixel][channel] by LDS-DMA and the MFMA fragments are
// fetched with the LDS transpose read ds_read_b64_tr_b16.
//
// What this kernel does differently from conv_wgrad_dlds_kernel (kept for fp32 and for shapes outside the conditions below):
//   * a tile is stored as SUB-TILES of 32 pixel rows x 64 channels (128-byte rows, 4 KB).  One LDS-DMA instruction per wavefront
//     fills 8 rows of one sub-tile, so every lane keeps ONE pixel row for the whole kernel: the pixel -> (image, row, column)
//     decode runs once per slice and lane, the tap of a sub-tile is wave-uniform, sub-tiles of the same tap differ by an
//     immediate offset, and the dy / 1x1 addresses are "offset += step" -- no divergent branch, no per-instruction M0 save/restore;
//   * the pixel range of a split is cut with the BUFFER DESCRIPTOR of dy (rows past the split read as zero, which also silences
//     whatever x holds there), not with compares;
//   * register allocation held to the workgroups per CU the LDS ring allows (three for the 128 x 128 tile; the old kernel sat 3
//     registers above that and ran two).
// Rows of a sub-tile are unpadded (LDS-DMA writes lane-linear); the two 64-byte halves of a row are swapped on odd row PAIRS
// (source-side swizzle), which puts the four rows a transposing half-wave touches on four disjoint bank groups.
#include "conv_wgrad.h"

namespace {

using vince_wgrad::WgradParams;
constexpr int SL = vince_wgrad::TR_SLICE;
constexpr int SUB = SL * 128;            // bytes of one sub-tile

static __device__ __forceinline__ void lds_dma16_imm(uint32_t lds_addr_uniform, uint32_t voff, v4i_t rsrc, int imm) {
    // imm in {0, 128, 256, 384}: sub-tiles of the same pixel row and tap are 64 channels apart (the immediate takes part in the range
    // check).  The instruction offset is added to the LDS address as well as to the memory address (LDS_ADDR = M0 + inst_offset +
    // lane * 16), so M0 carries the destination minus it.
    lds_addr_uniform -= (uint32_t)imm;
    if (imm == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr_uniform), "v"(voff), "s"(rsrc) : "memory");
    else if (imm == 128) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:128 lds" ::"s"(lds_addr_uniform), "v"(voff), "s"(rsrc) : "memory");
    else if (imm == 256) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:256 lds" ::"s"(lds_addr_uniform), "v"(voff), "s"(rsrc) : "memory");
    else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:384 lds" ::"s"(lds_addr_uniform), "v"(voff), "s"(rsrc) : "memory");
}

// n / d for the divisors of this kernel (d >= 2, host-made multiplier): fastdiv() without its d == 1 branch
static __device__ __forceinline__ uint32_t fdiv(uint32_t n, uint32_t mul, uint32_t shift_m1) {
    const uint32_t t = __umulhi(n, mul);
    return (t + ((n - t) >> 1)) >> shift_m1;
}

template <int CT, int NT, int STAGES>
constexpr int tr_min_blocks() {
    const int by_lds = 163840 / (STAGES * (CT + NT) / 64 * SUB);
    const int by_regs = CT * NT >= 256 * 128 ? 2 : CT * NT >= 128 * 128 ? 3 : 4;     // 128 / 64 / fewer accumulator registers per lane
    return by_lds < by_regs ? (by_lds < 1 ? 1 : by_lds) : by_regs;
}

// 4 wavefronts as 2 (channel halves) x 2 (reduction-side column halves); wavefront tile CT/2 x NT/2.
// RUN = 0: 1x1, stride 1, no padding -- the input pixel is the output pixel.  RUN > 0: consecutive x sub-tiles that share a tap
// (min(NT, Ci) / 64: tiles start at multiples of NT and taps at multiples of Ci, both powers of two times 64).
template <int CT, int NT, int STAGES, int RUN>
__global__ __launch_bounds__(256, (tr_min_blocks<CT, NT, STAGES>())) void conv_wgrad_tr_kernel(const WgradParams p) {
    constexpr bool LINEAR = RUN == 0;
    constexpr int YS = CT / 64, XS = NT / 64;                 // sub-tiles per operand
    constexpr int STAGE = (YS + XS) * SUB;
    constexpr int CJ = CT / 64, NJ = NT / 64;                 // 32 x 32 MFMA tiles per wavefront: CJ x NJ
    constexpr int PER = YS + XS;                              // DMA instructions per thread per slice
    static_assert(CJ * NJ <= 8 && STAGES >= 3, "wavefront tile");
    __shared__ __attribute__((aligned(16))) unsigned char smem[STAGES * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 1, wn = wave >> 1;
    // all (channel, tap) tiles of one pixel range on ONE XCD and adjacent in launch order: its L2 serves their shared dy / x rows
    uint32_t bx = blockIdx.x, by = blockIdx.y;
    if (p.xcd_group) {
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
#if defined(VINCE_MEASURE) || defined(VINCE_STEP_ABLATE)
    const int nkt = (p.ablate & 2) ? 0 : kt_end - kt_begin;
#else
    const int nkt = kt_end - kt_begin;
#endif
    const uint32_t pix_begin = (uint32_t)kt_begin * SL;
    const uint32_t pix_end = min((uint32_t)kt_end * SL, (uint32_t)p.M);

    // dy rows past the split (and past M) read as zero: the descriptor ends there.  x then needs no test of its own -- a zero
    // dy row silences it -- beyond the image bounds of its tap.
    const v4i_t rsrc_y = make_rsrc(p.dy, pix_end * (uint32_t)d.Co * 2u);
    const v4i_t rsrc_x = make_rsrc(p.in, LINEAR ? pix_end * (uint32_t)d.Ci * 2u : p.in_bytes);
    const uint32_t smem_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    constexpr uint32_t OOB = 0x80000000u;

    // ---- DMA role of this lane: pixel row (tid >> 3) of every slice, 16-byte slot (tid & 7) of the 128-byte sub-tile row
    const int drow = tid >> 3, dslot = tid & 7;
    const uint32_t dcolb = (uint32_t)(((((dslot >> 2) ^ ((drow >> 1) & 1)) << 2) | (dslot & 3)) * 16);   // logical byte column of the slot
    uint32_t yoff = (pix_begin + drow) * (uint32_t)d.Co * 2u + (uint32_t)c0 * 2u + dcolb;
    const uint32_t ystep = (uint32_t)SL * d.Co * 2u;
    uint32_t xoff = 0, xpix = pix_begin + drow;                  // LINEAR: byte offset; else the lane's pixel of the next slice to issue
    const uint32_t xstep = (uint32_t)SL * d.Ci * 2u;
    // general case: tap of sub-tile e (wave-uniform) -> displacement and byte column; sub-tiles of one tap share one address
    constexpr int NRUN = LINEAR ? 1 : XS / RUN;
    int sdh[NRUN], sdw[NRUN];
    uint32_t scol[NRUN];
    if constexpr (LINEAR) {
        xoff = (pix_begin + drow) * (uint32_t)d.Ci * 2u + (uint32_t)n0 * 2u + dcolb;
    } else {
#pragma unroll
        for (int e = 0; e < NRUN; ++e) {
            const int n = n0 + e * RUN * 64;
            const int tap = (d.TA * d.TB == 1) ? 0 : (n >> p.log2_ci);
            const int ci0 = n - (tap << ((d.TA * d.TB == 1) ? 0 : p.log2_ci));
            const int ta = (int)(((uint32_t)tap * p.tb_mul) >> 16), tb = tap - ta * d.TB;
            sdh[e] = d.dh0 + ta * d.dhs;
            sdw[e] = d.dw0 + tb * d.dws;
            scol[e] = (uint32_t)ci0 * 2u;
        }
    }
    const uint32_t cs2 = (uint32_t)p.cs * 2u;
    const uint32_t hw_mul = p.div_howo.mul, hw_sh = p.div_howo.shift - 1, hw_d = p.div_howo.d;
    const uint32_t w_mul = p.div_wo.mul, w_sh = p.div_wo.shift - 1, w_d = p.div_wo.d;

    auto issue_slice = [&](const int stage) {
        const uint32_t ys = __builtin_amdgcn_readfirstlane(smem_base + stage * STAGE + wave * 1024);
#pragma unroll
        for (int e = 0; e < YS; ++e) lds_dma16_imm(ys + e * SUB, yoff, rsrc_y, e * 128);
        yoff += ystep;
        const uint32_t xs = ys + YS * SUB;
        if constexpr (LINEAR) {
#pragma unroll
            for (int e = 0; e < XS; ++e) lds_dma16_imm(xs + e * SUB, xoff, rsrc_x, e * 128);
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
                const uint32_t off = ok ? (nimg + (uint32_t)(hi * d.Wi + wi)) * cs2 + scol[q] + dcolb : OOB;
#pragma unroll
                for (int e = 0; e < RUN; ++e) lds_dma16_imm(xs + (q * RUN + e) * SUB, off, rsrc_x, e * 128);
            }
        }
    };

    // ---- fragment read addresses (bytes inside a sub-tile, K step 0): lane -> pixel row, 64-byte half `h` of the sub-tile row.
    // A transposing read returns, to MFMA lane (column lane & 31, K group lane >> 5), 4 consecutive pixels of its column; the second
    // read (+4 rows) completes the 8.  K step 1 is +16 rows = +2048 bytes; the swizzle bit is the same for all four.
    const int g = lane >> 4, t = lane & 15;
    const int frow = (g >> 1) * 8 + (t >> 2);
    const int fsw = (frow >> 1) & 1;
    const uint32_t fcb = (uint32_t)((16 * (g & 1) + (t & 3) * 4) * 2);
    uint32_t ya[CJ], xa[NJ];                                     // per MFMA tile of this wavefront: sub-tile base + row + half
#pragma unroll
    for (int j = 0; j < CJ; ++j) {
        const int col = wc * (CT / 2) + j * 32;
        ya[j] = (uint32_t)((col >> 6) * SUB + frow * 128 + ((((col >> 5) & 1) ^ fsw) << 6)) + fcb;
    }
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const int col = wn * (NT / 2) + i * 32;
        xa[i] = (uint32_t)((YS + (col >> 6)) * SUB + frow * 128 + ((((col >> 5) & 1) ^ fsw) << 6)) + fcb;
    }
    auto frag = [&](const uint32_t addr) -> bf16x8_t {
        typedef __attribute__((address_space(3))) s16x4_t* lp;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)addr);
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)(addr + 512));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    f32x16_t acc[CJ][NJ];
#pragma unroll
    for (int j = 0; j < CJ; ++j)
#pragma unroll
        for (int i = 0; i < NJ; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

    // ---- ring of STAGES slices: slice s lives in stage s % STAGES; STAGES - 1 slices are in flight ahead of the one being multiplied
#pragma unroll
    for (int st = 0; st < STAGES - 1; ++st) issue_slice(st);
    wait_vmcnt<(STAGES - 2) * PER>();
    __builtin_amdgcn_s_barrier();
    for (int it0 = 0; it0 < nkt; it0 += STAGES) {
#pragma unroll
        for (int s = 0; s < STAGES; ++s) {
            if (it0 + s >= nkt) break;
#ifdef VINCE_MEASURE
            if (!(p.ablate & 4))
#endif
            issue_slice((s + STAGES - 1) % STAGES);               // the stage whose slice was multiplied in the previous iteration
            const uint32_t sb = smem_base + s * STAGE;
            bf16x8_t af[2][CJ], bfr[2][NJ];
#ifdef VINCE_MEASURE
            if (p.ablate & 8) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int j = 0; j < CJ; ++j) af[ks][j] = (bf16x8_t)(short)(lane + j);
#pragma unroll
                    for (int i = 0; i < NJ; ++i) bfr[ks][i] = (bf16x8_t)(short)(lane + i);
                }
            } else
#endif
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int j = 0; j < CJ; ++j) af[ks][j] = frag(sb + ya[j] + ks * 2048);
#pragma unroll
                for (int i = 0; i < NJ; ++i) bfr[ks][i] = frag(sb + xa[i] + ks * 2048);
            }
#ifdef VINCE_MEASURE
            if (!(p.ablate & 16))
#endif
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < CJ; ++j)
#pragma unroll
                    for (int i = 0; i < NJ; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][j], bfr[ks][i], acc[j][i], 0, 0, 0);
#ifdef VINCE_MEASURE
            if (p.ablate & 16) {
#pragma unroll
                for (int j = 0; j < CJ; ++j)
#pragma unroll
                    for (int i = 0; i < NJ; ++i) acc[j][i][0] += (float)(af[0][j][0] + bfr[0][i][0] + af[1][j][7] + bfr[1][i][7]);
            }
#endif
            // The compiler is free to sink the barrier between the MFMAs (they touch no memory), which overlaps the wait with the matrix
            // pipe; what it must not do is cross it with a fragment read still in flight -- the next iteration's DMA overwrites this stage.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wait_vmcnt<(STAGES - 2) * PER>();                     // this thread's share of the NEXT slice has landed
            __builtin_amdgcn_s_barrier();                         // ... everyone's has; and everyone is done reading this one
        }
    }
    wait_vmcnt<0>();                                              // the zero-filled slices issued past the end
#if defined(VINCE_MEASURE) || defined(VINCE_STEP_ABLATE)
    if (p.ablate & 1) {
        float t_ = 0.f;
        for (int j = 0; j < CJ; ++j) for (int i = 0; i < NJ; ++i) t_ += acc[j][i][0];
        if (t_ == 1.2345f) p.dw[0] = t_;
        return;
    }
#endif

    // ---- the partial tile into dw: fp32 atomics (32 consecutive floats per row per instruction), or plain stores into this
    // split's slab (reproducible mode)
    const int T_ = d.TA * d.TB;
    float* const dst = p.slab ? p.slab + (size_t)by * p.slab_stride : p.dw;
    const uint32_t row_stride = (uint32_t)d.WT * (uint32_t)p.Ci_dw;
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const int n = n0 + wn * (NT / 2) + i * 32 + (lane & 31);
        const int tp = (T_ == 1) ? 0 : (n >> p.log2_ci);
        const int ci = (T_ == 1) ? n : (n & ((1 << p.log2_ci) - 1));
        const int a = (int)(((uint32_t)tp * p.tb_mul) >> 16), b = tp - a * d.TB;
        const uint32_t col = (uint32_t)(d.wt0 + a * d.wta + b * d.wtb) * (uint32_t)p.Ci_dw + (uint32_t)ci;
        if (ci >= p.Ci_dw) continue;
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const uint32_t co0 = (uint32_t)(c0 + wc * (CT / 2) + j * 32 + 4 * (lane >> 5));
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

template <int CT, int NT, int STAGES>
int launch_tr(const WgradParams& p, int splits, hipStream_t stream) {
    const dim3 grid(p.ctiles * p.ntiles, splits), block(256);
    constexpr int XS = NT / 64;
    const int run = p.linear_x ? 0 : (p.d.Ci / 64 < XS ? p.d.Ci / 64 : XS);
    if (run == 0) hipLaunchKernelGGL((conv_wgrad_tr_kernel<CT, NT, STAGES, 0>), grid, block, 0, stream, p);
    else if (run == 1) hipLaunchKernelGGL((conv_wgrad_tr_kernel<CT, NT, STAGES, 1>), grid, block, 0, stream, p);
    else if constexpr (XS >= 2) {
        if (run == 2) hipLaunchKernelGGL((conv_wgrad_tr_kernel<CT, NT, STAGES, 2>), grid, block, 0, stream, p);
        else if constexpr (XS >= 4) hipLaunchKernelGGL((conv_wgrad_tr_kernel<CT, NT, STAGES, 4>), grid, block, 0, stream, p);
    }
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

}  // namespace

namespace vince_wgrad {

// Eligible: bf16, both tensors within a 2 GB descriptor, plain (unpacked) taps, Co and T * Ci multiples of 64 with Ci itself a
// multiple of 64 (a sub-tile never straddles a tap), dw as wide as the descriptor's Ci, images of at least one slice.
void wgrad_tr_tile(const WgradParams& p, int* ct, int* nt) {
    const vince_conv_desc& d = p.d;
    *ct = *nt = 0;
    const int ntot = d.TA * d.TB * d.Ci;
    if (!(p.in_bytes && p.dy_bytes) || p.variant != 0 || d.Cs != 0 || d.Co % 64 || d.Ci % 64 || p.Ci_dw != d.Ci) return;
    if (d.Ho * d.Wo < 2 || d.Wo < 2) return;                       // fdiv wants divisors >= 2
    if ((unsigned long long)p.M * d.Co * 2 + (1u << 20) >= 0x7ff00000ull) return;
    static const long forced = vince_knob("wgrad_tile", 0);   // cross-check switch: ct * 1000 + nt (64 / 128 each) for every layer it divides
    if (forced) {
        *ct = (int)(forced / 1000);
        *nt = (int)(forced % 1000);
        if ((*ct == 64 || *ct == 128) && (*nt == 64 || *nt == 128) && d.Co % *ct == 0 && ntot % *nt == 0) return;
    }
    *ct = d.Co % 128 ? 64 : 128;
    *nt = ntot % 128 ? 64 : 128;
}

int wgrad_tr_launch(const WgradParams& p, int ct, int nt, int splits, hipStream_t stream) {
    // Ring depth.  1x1 / stride 1 layers (one address step per slice, nothing else between the MFMAs): FOUR stages -- three slices in flight,
    // two workgroups per CU instead of three -- 28x28 512<-128 66.1 -> 57.3 us, 128<-512 62.1 -> 56.3, the 14x14 and 7x7 pairs -4.5 / -5.5 %,
    // layer1's HBM-bound pair -2 % (tools/wgrad_micro.py, round 4).  The multi-tap layers keep three: with four the 3x3 layers lose 13-16 %
    // (85 -> 99 us on layer3's).  `wgrad_stages4_linear=0`: three everywhere (cross-check switch).  (With four stages the 1x1 layers timed
    // ALONE prefer 256 workgroups per launch to 512 -- 50 -> 42 us at 14x14, 58 -> 49 at 28x28 -- but inside the overlapped step that split
    // is 0.05-0.09 ms SLOWER in two paired runs: one long-lived workgroup per CU holds its 64 KB of LDS against the chain's kernels.  512 stays.)
    static const int stages_knob = (int)VINCE_MEASURE_KNOB("wgrad_stages", 0);
    static const bool linear4 = vince_knob("wgrad_stages4_linear", 1) != 0;
    const int stages4 = stages_knob ? stages_knob == 4 : (linear4 && p.linear_x);
#define TR_CASE(C, N)                                                                                   \
    if (ct == C && nt == N) {                                                                           \
        if constexpr ((C + N) / 64 * SUB * 4 <= 163840) {                                               \
            if (stages4) return launch_tr<C, N, 4>(p, splits, stream);                                  \
        }                                                                                               \
        return launch_tr<C, N, 3>(p, splits, stream);                                                   \
    }
    // (256-wide members -- 256 x 128, 128 x 256, 256 x 64, 64 x 256: the template takes them -- were measured at the benchmark batch and not
    // kept: two workgroups per CU hide less latency than three, layer3's 3x3 115 us against 86, profiles/r03_wgrad_tr.txt)
    TR_CASE(64, 64) TR_CASE(64, 128) TR_CASE(128, 64) TR_CASE(128, 128)
#undef TR_CASE
    vince_set_error("wgrad_tr_launch: no %d x %d tile", ct, nt);
    return VINCE_E_SHAPE;
}

}  // namespace vince_wgrad
