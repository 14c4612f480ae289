// Weight gradient of the generalised tap convolution: a "TN" GEMM whose reduction axis is the output pixel.
//
//   dw[co][tap][ci] += sum_pix dy[pix][co] * x[pix @ tap][ci]
//
// Replaces the weight-gradient half of the reference's conv2d / nn.Linear autograd (resnet.py:34-50,170;
// vince_model.py:38-42 under loss.backward(), solvers/vince_solver.py:463-468).
//
// Both operands are contiguous along the NON-reduced axis (channels), the opposite of what an MFMA fragment
// wants (8 consecutive k per lane).  The tiles are staged [pixel][channel] in LDS exactly as they sit in HBM
// (16-byte coalesced chunks) and the fragments are fetched with the gfx950 LDS transpose read
// ds_read_b64_tr_b16 (bf16) or with conflict-free ds_read_b32 (f32: one k per lane per v_mfma_f32_32x32x2_f32).
// Work split: (Co tile) x (tap*Ci tile) x (split of the pixel range); partial tiles are accumulated with
// fp32 atomics straight into the gradient buffer (rows of 32 consecutive floats per wave instruction).
#include <stdlib.h>

#include <type_traits>

#include "conv_wgrad.h"

namespace {

using vince_wgrad::WgradParams;

constexpr int KP = 64;  // pixels per K tile

template <typename T, int CT, int NT>
struct WSmem {
    // (+64 B padding, which puts the 4 rows of a ds_read_b64_tr_b16 half-wave on disjoint bank groups, measured no
    // faster than +16 B on MI355X and costs a workgroup per CU at KP=64)
    static constexpr int YRS = CT * (int)sizeof(T) + 16;
    static constexpr int XRS = NT * (int)sizeof(T) + 16;
    static constexpr int BUF = KP * (YRS + XRS);
    static constexpr int BYTES = 2 * BUF;
};

// 32(k=16.. see below) fragment fetch: returns the 16 bytes an MFMA lane needs for column `col0 + (lane&31)`
// and k-group (lane>>5) of k-step ks, from an LDS tile stored [k][column] with row stride rs bytes.
template <typename T> struct FragT;
template <> struct FragT<bf16_t> {
    static constexpr int KSTEPS = KP / 16;
    __device__ static inline uint4 load(const unsigned char* tile, int rs, int col0, int ks, int lane, int variant) {
        uint4 out;
        if (variant == 0) {
            const int g = lane >> 4, t = lane & 15;
            const int col = col0 + 16 * (g & 1) + (t & 3) * 4;
            const int row = ks * 16 + (g >> 1) * 8 + (t >> 2);
            const unsigned char* a0 = tile + row * rs + col * 2;
            s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0));
            s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0 + 4 * rs));
            __builtin_memcpy(&out.x, &lo, 8);
            __builtin_memcpy(&out.z, &hi, 8);
        } else {
            const unsigned char* a0 = tile + (ks * 16 + (lane >> 5) * 8) * rs + (col0 + (lane & 31)) * 2;
            uint32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *(const uint16_t*)(a0 + k * rs);
            out = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
        }
        return out;
    }
    __device__ static inline void mma(const uint4& a, const uint4& b, f32x16_t& c) {
        bf16x8_t av, bv;
        __builtin_memcpy(&av, &a, 16);
        __builtin_memcpy(&bv, &b, 16);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
    }
};
template <> struct FragT<float> {
    static constexpr int KSTEPS = KP / 8;   // 4 x (32x32x2) per fetch of 4 k
    __device__ static inline uint4 load(const unsigned char* tile, int rs, int col0, int ks, int lane, int) {
        // lane reads k = ks*8 + (lane>>5)*4 + {0..3} for column col0 + (lane&31): 4 conflict-free ds_read_b32
        const unsigned char* a0 = tile + (ks * 8 + (lane >> 5) * 4) * rs + (col0 + (lane & 31)) * 4;
        return make_uint4(*(const uint32_t*)a0, *(const uint32_t*)(a0 + rs), *(const uint32_t*)(a0 + 2 * rs),
                          *(const uint32_t*)(a0 + 3 * rs));
    }
    __device__ static inline void mma(const uint4& a, const uint4& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};

// split-half products (bfloat16 halves: gradients need fp32's exponent range): the fp32 tiles and fetches, two fetches (8 pixels per
// lane and column) split into hi / lo halves in registers feed three v_mfma_f32_32x32x16_bf16 per block
template <> struct FragT<x3b_t> : FragT<float> {};
template <> struct FragT<x1b_t> : FragT<float> {};

template <typename T, int CT, int NT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
    constexpr int CH = Elem<T>::CH;
    using S = WSmem<T, CT, NT>;
    constexpr int CPRY = CT / CH, CPRX = NT / CH;          // 16-byte chunks per tile row
    constexpr int NY = (KP * CPRY) / 256 > 0 ? (KP * CPRY) / 256 : 1;
    constexpr int NX = (KP * CPRX) / 256 > 0 ? (KP * CPRX) / 256 : 1;
    constexpr int RPY = 256 / CPRY, RPX = 256 / CPRX;       // rows covered per pass
    constexpr int CJ = CT / 64, NJ = NT / 64;
    __shared__ __attribute__((aligned(16))) unsigned char smem[S::BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave & 1, wn = wave >> 1;
    // All (channel, tap) tiles of one pixel range read the same dY / X rows: keep them on ONE XCD (one L2) and adjacent
    // in launch order.  The hardware deals workgroups to the 8 XCDs round-robin by linear id, so the linear id is
    // remapped to "XCD x gets a contiguous run of (split, tile) pairs" (VINCE_WGRAD_XCD=0: plain order, measurement aid).
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

    const T* __restrict__ in = (const T*)p.in;
    const T* __restrict__ dy = (const T*)p.dy;

    // dy staging: chunk column ycol, rows yrow + e*RPY
    const int ycol = tid % CPRY, yrow = tid / CPRY;
    const bool yvalid_c = (c0 + ycol * CH) < d.Co && yrow < KP;
    // x staging: chunk column xcol -> fixed (tap, ci chunk)
    const int xcol = tid % CPRX, xrow = tid / CPRX;
    const int qn = n0 / CH + xcol;
    const int tap = qn >> p.log2_cpt, cc = qn & p.cpt_mask;
    const int ta = (int)(((uint32_t)tap * p.tb_mul) >> 16), tb = tap - ta * d.TB;
    const int dh = d.dh0 + ta * d.dhs, dw_ = d.dw0 + tb * d.dws;
    const bool xvalid_c = qn < p.total_nchunks && xrow < KP;

    uint4 yr[NY], xr[NX];
    auto load_tile = [&](int kt) {
        const uint32_t pix0 = (uint32_t)kt * KP;
#pragma unroll
        for (int e = 0; e < NY; ++e) {
            const uint32_t pix = pix0 + yrow + e * RPY;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (yvalid_c && pix < (uint32_t)p.M) v = *(const uint4*)(dy + (size_t)pix * d.Co + c0 + ycol * CH);
            yr[e] = v;
        }
#pragma unroll
        for (int e = 0; e < NX; ++e) {
            const uint32_t pix = pix0 + xrow + e * RPX;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (xvalid_c && pix < (uint32_t)p.M) {
                const uint32_t n = fastdiv(pix, p.div_howo);
                const uint32_t rem = pix - n * p.div_howo.d;
                const uint32_t ho = fastdiv(rem, p.div_wo);
                const uint32_t wo = rem - ho * p.div_wo.d;
                const int hi = (int)(ho * d.sh) + dh, wi = (int)(wo * d.sw) + dw_;
                if ((unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi)
                    v = *(const uint4*)(in + (((size_t)n * d.Hi + hi) * d.Wi + wi) * p.cs + (size_t)cc * CH);
            }
            xr[e] = v;
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* ys = smem + buf * S::BUF;
        unsigned char* xs = ys + KP * S::YRS;
#pragma unroll
        for (int e = 0; e < NY; ++e)
            if (yrow + e * RPY < KP) *(uint4*)(ys + (yrow + e * RPY) * S::YRS + ycol * 16) = yr[e];
#pragma unroll
        for (int e = 0; e < NX; ++e)
            if (xrow + e * RPX < KP) *(uint4*)(xs + (xrow + e * RPX) * S::XRS + xcol * 16) = xr[e];
    };

    f32x16_t acc[CJ][NJ];
#pragma unroll
    for (int j = 0; j < CJ; ++j)
#pragma unroll
        for (int i = 0; i < NJ; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

    load_tile(kt_begin);
    store_tile(0);
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int buf = (kt - kt_begin) & 1;
        if (kt + 1 < kt_end) load_tile(kt + 1);
        const unsigned char* ys = smem + buf * S::BUF;
        const unsigned char* xs = ys + KP * S::YRS;
        if constexpr (X3<T>::on) {
#pragma unroll
            for (int ks = 0; ks < FragT<T>::KSTEPS; ks += 2) {
                uint4 ah[CJ], al[CJ], bh[NJ], bl[NJ];
#pragma unroll
                for (int j = 0; j < CJ; ++j)
                    x3_split<T, false>(FragT<T>::load(ys, S::YRS, wc * (CT / 2) + j * 32, ks, lane, p.variant),
                                       FragT<T>::load(ys, S::YRS, wc * (CT / 2) + j * 32, ks + 1, lane, p.variant), ah[j], al[j]);
#pragma unroll
                for (int i = 0; i < NJ; ++i)
                    x3_split<T, false>(FragT<T>::load(xs, S::XRS, wn * (NT / 2) + i * 32, ks, lane, p.variant),
                                       FragT<T>::load(xs, S::XRS, wn * (NT / 2) + i * 32, ks + 1, lane, p.variant), bh[i], bl[i]);
#pragma unroll
                for (int j = 0; j < CJ; ++j)
#pragma unroll
                    for (int i = 0; i < NJ; ++i) x3_mma<T>(ah[j], al[j], bh[i], bl[i], acc[j][i]);
            }
        } else
#pragma unroll
        for (int ks = 0; ks < FragT<T>::KSTEPS; ++ks) {
            uint4 af[CJ], bf[NJ];
#pragma unroll
            for (int j = 0; j < CJ; ++j) af[j] = FragT<T>::load(ys, S::YRS, wc * (CT / 2) + j * 32, ks, lane, p.variant);
#pragma unroll
            for (int i = 0; i < NJ; ++i) bf[i] = FragT<T>::load(xs, S::XRS, wn * (NT / 2) + i * 32, ks, lane, p.variant);
#pragma unroll
            for (int j = 0; j < CJ; ++j)
#pragma unroll
                for (int i = 0; i < NJ; ++i) FragT<T>::mma(af[j], bf[i], acc[j][i]);
        }
        if (kt + 1 < kt_end) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- accumulate the partial tile into dw (fp32 atomics; 32 consecutive floats per row per instruction) ----
    const int T_ = d.TA * d.TB;
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const int n = n0 + wn * (NT / 2) + i * 32 + (lane & 31);
        const int tp = (T_ == 1) ? 0 : (n >> p.log2_ci);
        const int ci = (T_ == 1) ? n : (n & ((1 << p.log2_ci) - 1));
        if (tp >= T_ || (T_ == 1 && n >= d.Ci)) continue;
        const int a = (int)(((uint32_t)tp * p.tb_mul) >> 16), b = tp - a * d.TB;
        int widx = d.wt0 + a * d.wta + b * d.wtb, wtn = d.WT, cdst = ci;
        if (d.Cs > 0) {   // packed row taps: element ci of tap a is (kw, c); dw is [Co][TA][Kw][Ci_dw]
            const int kw = ci / d.Cs;
            cdst = ci - kw * d.Cs;
            if (kw >= d.Kw) continue;
            widx = widx * d.Kw + kw;
            wtn = d.WT * d.Kw;
        }
        if (cdst >= p.Ci_dw) continue;
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = c0 + wc * (CT / 2) + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < d.Co) {
                    const size_t idx = ((size_t)co * wtn + widx) * p.Ci_dw + cdst;
                    if (p.slab) p.slab[(size_t)by * p.slab_stride + idx] = acc[j][i][r];
                    else unsafeAtomicAdd(p.dw + idx, acc[j][i][r]);
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Direct-to-LDS variant: both tiles are filled by LDS-DMA into a STAGES-deep ring of 32-pixel slices (counted vmcnt,
// one barrier per slice, no staging VGPRs, no ds_write pass).  Rows are unpadded (a DMA instruction writes
// wave-base + lane*16), so the four pixel rows a ds_read_b64_tr_b16 half-wave touches are de-conflicted by XOR-ing
// the 64-byte block index of a row with a function of the row, applied to the per-lane SOURCE address.
constexpr int KPD = 32;

template <typename T, int CT, int NT, int STAGES>
struct WSmemD {
    static constexpr int YRB = CT * (int)sizeof(T), XRB = NT * (int)sizeof(T);   // row bytes
    static constexpr int YB = KPD * YRB, XB = KPD * XRB, STAGE = YB + XB;
    static constexpr int BYTES = STAGES * STAGE;
};

// block swizzle of a row (bf16 only; the f32 path reads with conflict-free ds_read_b32)
template <typename T, int ROWBYTES>
__device__ __forceinline__ int row_swz(int row) {
    if constexpr (sizeof(T) == 4) return 0;
    else if constexpr (ROWBYTES == 128) return (row >> 1) & 1;
    else return row & 3;
}

template <typename T, int ROWBYTES> struct FragD;
template <int ROWBYTES> struct FragD<bf16_t, ROWBYTES> {
    static constexpr int KSTEPS = KPD / 16;
    __device__ static inline uint4 load(const unsigned char* tile, int col0, int ks, int lane) {
        const int g = lane >> 4, t = lane & 15;
        const int colb = (col0 + 16 * (g & 1) + (t & 3) * 4) * 2;
        const int row = ks * 16 + (g >> 1) * 8 + (t >> 2);
        const int sw = row_swz<bf16_t, ROWBYTES>(row);          // identical for row and row + 4
        const unsigned char* a0 = tile + row * ROWBYTES + (((colb >> 6) ^ sw) << 6) + (colb & 63);
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0 + 4 * ROWBYTES));
        uint4 out;
        __builtin_memcpy(&out.x, &lo, 8);
        __builtin_memcpy(&out.z, &hi, 8);
        return out;
    }
};
template <int ROWBYTES> struct FragD<float, ROWBYTES> {
    static constexpr int KSTEPS = KPD / 8;
    __device__ static inline uint4 load(const unsigned char* tile, int col0, int ks, int lane) {
        const unsigned char* a0 = tile + (ks * 8 + (lane >> 5) * 4) * ROWBYTES + (col0 + (lane & 31)) * 4;
        return make_uint4(*(const uint32_t*)a0, *(const uint32_t*)(a0 + ROWBYTES), *(const uint32_t*)(a0 + 2 * ROWBYTES),
                          *(const uint32_t*)(a0 + 3 * ROWBYTES));
    }
};

template <int ROWBYTES> struct FragD<x3b_t, ROWBYTES> : FragD<float, ROWBYTES> {};
template <int ROWBYTES> struct FragD<x1b_t, ROWBYTES> : FragD<float, ROWBYTES> {};

// NW wavefronts as 2 (channel halves) x NW/2 (reduction-side column groups): 4 = the 128 x 128 tile of 2 x 2 MFMA tiles per wavefront,
// 8 = the 256 x 256 tile of the long multi-tap layers (4 x 2 MFMA tiles per wavefront: half the operand bytes per FLOP through L2 -> LDS)
// workgroups per CU the register allocation is held to (the LDS ring allows as many): the 128 x 128 bf16 tile sat 3 registers above three
template <typename T, int CT, int NT, int STAGES, int NW>
constexpr int wgrad_min_blocks() {
    const int by_lds = 163840 / (STAGES * KPD * (CT + NT) * (int)sizeof(T));
    const int want0 = (CT * NT / (NW * 64)) >= 128 ? 2 : (CT * NT / (NW * 64)) >= 64 ? 3 : 4;   // accumulator registers per lane: 128 / 64 / fewer
    const int want = (X3<T>::on && want0 > 2) ? want0 - 1 : want0;   // the split-half types also hold raw + split fragments
    return by_lds < want ? (by_lds < 1 ? 1 : by_lds) : want;
}

template <typename T, int CT, int NT, int STAGES, int NW = 4>
__global__ __launch_bounds__(NW * 64, (wgrad_min_blocks<T, CT, NT, STAGES, NW>())) void conv_wgrad_dlds_kernel(const WgradParams p) {
    constexpr int CH = Elem<T>::CH;
    using S = WSmemD<T, CT, NT, STAGES>;
    constexpr int SPRY = S::YRB / 16, SPRX = S::XRB / 16;          // 16-byte slots per row
    constexpr int NTHR = NW * 64, WNN = NW / 2;
    constexpr int RPPY = NTHR / SPRY, RPPX = NTHR / SPRX;         // rows per pass of the NW waves
    constexpr int NY = KPD / RPPY, NX = KPD / RPPX;                // DMA instructions per thread per stage
    constexpr int PER_STAGE = NY + NX;
    constexpr int CJ = CT / 64, NJ = NT / (32 * WNN);
    constexpr uint32_t OOB = 0x80000000u;
    __shared__ __attribute__((aligned(16))) unsigned char smem[S::BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 1, wn = wave >> 1;    // wn in [0, WNN)
    // All (channel, tap) tiles of one pixel range read the same dY / X rows: keep them on ONE XCD (one L2) and adjacent
    // in launch order.  The hardware deals workgroups to the 8 XCDs round-robin by linear id, so the linear id is
    // remapped to "XCD x gets a contiguous run of (split, tile) pairs" (VINCE_WGRAD_XCD=0: plain order, measurement aid).
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
#ifdef VINCE_MEASURE
    const int nkt = (p.ablate & 2) ? 0 : kt_end - kt_begin;
#else
    const int nkt = kt_end - kt_begin;
#endif

    const v4i_t rsrc_y = make_rsrc(p.dy, p.dy_bytes);
    const v4i_t rsrc_x = make_rsrc(p.in, p.in_bytes);
    const uint32_t smem_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;

    // dy: slot -> logical 16-byte chunk of the channel tile (64-byte blocks XOR-swizzled by row)
    const int yrow = tid / SPRY, yslot = tid % SPRY;
    const int ychunk = (((yslot >> 2) ^ row_swz<T, S::YRB>(yrow)) << 2) | (yslot & 3);
    const bool yvalid_c = (c0 + ychunk * CH) < d.Co;
    const uint32_t ycol_off = (uint32_t)(c0 + ychunk * CH) * (uint32_t)sizeof(T);
    // x: slot -> (tap, ci chunk), fixed for the whole kernel
    const int xrow = tid / SPRX, xslot = tid % SPRX;
    const int xchunk = (((xslot >> 2) ^ row_swz<T, S::XRB>(xrow)) << 2) | (xslot & 3);
    const int qn = n0 / CH + xchunk;
    const int tap = qn >> p.log2_cpt, cc = qn & p.cpt_mask;
    const int ta = (int)(((uint32_t)tap * p.tb_mul) >> 16), tb = tap - ta * d.TB;
    const int dh = d.dh0 + ta * d.dhs, dw_ = d.dw0 + tb * d.dws;
    const bool xvalid_c = qn < p.total_nchunks;
    const uint32_t xcol_off = (uint32_t)cc * CH * (uint32_t)sizeof(T);

    auto issue_slice = [&](int kt, int buf) {      // kt is the absolute slice index (32 pixels)
        const uint32_t pix0 = (uint32_t)kt * KPD;
        const bool live = kt < kt_end;
        const uint32_t ys = __builtin_amdgcn_readfirstlane(smem_base + buf * S::STAGE + wave * 1024);
        const uint32_t xs = __builtin_amdgcn_readfirstlane(smem_base + buf * S::STAGE + S::YB + wave * 1024);
#pragma unroll
        for (int e = 0; e < NY; ++e) {
            const uint32_t pix = pix0 + yrow + e * RPPY;
            const uint32_t off = (live && yvalid_c && pix < (uint32_t)p.M) ? pix * (uint32_t)d.Co * (uint32_t)sizeof(T) + ycol_off : OOB;
            lds_dma16(ys + e * (NW * 1024), off, rsrc_y);
        }
#pragma unroll
        for (int e = 0; e < NX; ++e) {
            const uint32_t pix = pix0 + xrow + e * RPPX;
            uint32_t off = OOB;
            if (live && xvalid_c && pix < (uint32_t)p.M) {
                if (p.linear_x) {
                    off = pix * (uint32_t)d.Ci * (uint32_t)sizeof(T) + xcol_off;
                } else {
                    const uint32_t n = fastdiv(pix, p.div_howo);
                    const uint32_t rem = pix - n * p.div_howo.d;
                    const uint32_t ho = fastdiv(rem, p.div_wo);
                    const uint32_t wo = rem - ho * p.div_wo.d;
                    const int hi = (int)(ho * d.sh) + dh, wi = (int)(wo * d.sw) + dw_;
                    if ((unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi)
                        off = ((n * (uint32_t)d.Hi + hi) * (uint32_t)d.Wi + wi) * (uint32_t)p.cs * (uint32_t)sizeof(T) + xcol_off;
                }
            }
            lds_dma16(xs + e * (NW * 1024), off, rsrc_x);
        }
    };

    f32x16_t acc[CJ][NJ];
#pragma unroll
    for (int j = 0; j < CJ; ++j)
#pragma unroll
        for (int i = 0; i < NJ; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

#pragma unroll
    for (int st = 0; st < STAGES - 1; ++st) issue_slice(kt_begin + st, st);
    wait_vmcnt<(STAGES - 2) * PER_STAGE>();
    __syncthreads();
    int buf = 0, nbuf = STAGES - 1;
    for (int it = 0; it < nkt; ++it) {
#ifdef VINCE_MEASURE
        if (!(p.ablate & 4))      // 4: no DMA in the loop (stale tiles)
#endif
        issue_slice(kt_begin + it + STAGES - 1, nbuf);
        const unsigned char* ys = smem + buf * S::STAGE;
        const unsigned char* xs = ys + S::YB;
#pragma unroll
        for (int ks = 0; ks < FragD<T, S::YRB>::KSTEPS; ks += (X3<T>::on ? 2 : 1)) {
            uint4 af[CJ], bf[NJ];
#ifdef VINCE_MEASURE
            if (p.ablate & 8) {   // 8: no LDS fragment reads
#pragma unroll
                for (int j = 0; j < CJ; ++j) af[j] = make_uint4(it + j, lane, ks, 1);
#pragma unroll
                for (int i = 0; i < NJ; ++i) bf[i] = make_uint4(it, lane + i, ks, 2);
            } else
#endif
            {
#pragma unroll
            for (int j = 0; j < CJ; ++j) af[j] = FragD<T, S::YRB>::load(ys, wc * (CT / 2) + j * 32, ks, lane);
#pragma unroll
            for (int i = 0; i < NJ; ++i) bf[i] = FragD<T, S::XRB>::load(xs, wn * (NT / WNN) + i * 32, ks, lane);
            }
#ifdef VINCE_MEASURE
            if (p.ablate & 16) {  // 16: no MFMA (fragments consumed by one add)
#pragma unroll
                for (int j = 0; j < CJ; ++j)
#pragma unroll
                    for (int i = 0; i < NJ; ++i) acc[j][i][0] += __uint_as_float(af[j].x ^ bf[i].y ^ af[j].z ^ bf[i].w ^ af[j].y ^ bf[i].x ^ af[j].w ^ bf[i].z);
                continue;
            }
#endif
            if constexpr (X3<T>::on) {   // two fetches -> one split-half block (ks advances by 2 below)
                uint4 ah[CJ], al[CJ], bh[NJ], bl[NJ];
#pragma unroll
                for (int j = 0; j < CJ; ++j) x3_split<T, false>(af[j], FragD<T, S::YRB>::load(ys, wc * (CT / 2) + j * 32, ks + 1, lane), ah[j], al[j]);
#pragma unroll
                for (int i = 0; i < NJ; ++i) x3_split<T, false>(bf[i], FragD<T, S::XRB>::load(xs, wn * (NT / WNN) + i * 32, ks + 1, lane), bh[i], bl[i]);
#pragma unroll
                for (int j = 0; j < CJ; ++j)
#pragma unroll
                    for (int i = 0; i < NJ; ++i) x3_mma<T>(ah[j], al[j], bh[i], bl[i], acc[j][i]);
            } else {
#pragma unroll
            for (int j = 0; j < CJ; ++j)
#pragma unroll
                for (int i = 0; i < NJ; ++i) FragT<T>::mma(af[j], bf[i], acc[j][i]);
            }
        }
        wait_vmcnt<(STAGES - 2) * PER_STAGE>();
        __syncthreads();
        buf = buf + 1 == STAGES ? 0 : buf + 1;
        nbuf = nbuf + 1 == STAGES ? 0 : nbuf + 1;
    }
    wait_vmcnt<0>();
#ifdef VINCE_MEASURE
    if (p.ablate & 1) {   // measurement build: no atomics (one dummy store keeps the accumulators alive)
        float t_ = 0.f;
        for (int j = 0; j < CJ; ++j) for (int i = 0; i < NJ; ++i) t_ += acc[j][i][0];
        if (t_ == 1.2345f) p.dw[0] = t_;
        return;
    }
#endif

    const int T_ = d.TA * d.TB;
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const int n = n0 + wn * (NT / WNN) + i * 32 + (lane & 31);
        const int tp = (T_ == 1) ? 0 : (n >> p.log2_ci);
        const int ci = (T_ == 1) ? n : (n & ((1 << p.log2_ci) - 1));
        if (tp >= T_ || (T_ == 1 && n >= d.Ci)) continue;
        const int a = (int)(((uint32_t)tp * p.tb_mul) >> 16), b = tp - a * d.TB;
        int widx = d.wt0 + a * d.wta + b * d.wtb, wtn = d.WT, cdst = ci;
        if (d.Cs > 0) {   // packed row taps: element ci of tap a is (kw, c); dw is [Co][TA][Kw][Ci_dw]
            const int kw = ci / d.Cs;
            cdst = ci - kw * d.Cs;
            if (kw >= d.Kw) continue;
            widx = widx * d.Kw + kw;
            wtn = d.WT * d.Kw;
        }
        if (cdst >= p.Ci_dw) continue;
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = c0 + wc * (CT / 2) + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < d.Co) {
                    const size_t idx = ((size_t)co * wtn + widx) * p.Ci_dw + cdst;
                    if (p.slab) p.slab[(size_t)by * p.slab_stride + idx] = acc[j][i][r];
                    else unsafeAtomicAdd(p.dw + idx, acc[j][i][r]);
                }
            }
        }
    }
}

// dw[i] += slab[0][i] + slab[1][i] + ... in a fixed order (-> run-to-run identical).  Thread = (float4 column, split lane): the 256
// threads of a workgroup cover 16 float4 columns x 16 split lanes; lane l sums splits [l*q, (l+1)*q) in order, the 16 partial sums are
// then added in lane order.  (16 lanes per column, not 4: a Gram matrix of layer1 is 1024 float4 columns under up to 512 slabs -- 16
// workgroups walking 128 slabs each took 40 us on the forward's critical path, twice the GEMM that produced them.)
constexpr int RED_COLS = 16, RED_LANES = 16;
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int splits, size_t stride, float* __restrict__ dw, size_t n4) {
    __shared__ float4 red[RED_LANES][RED_COLS];
    const int col = threadIdx.x % RED_COLS, sl = threadIdx.x / RED_COLS;
    const size_t i4 = (size_t)blockIdx.x * RED_COLS + col;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i4 < n4) {
        const int q = (splits + RED_LANES - 1) / RED_LANES, s0 = sl * q, s1 = min(splits, s0 + q);
        const float* src = slab + (size_t)s0 * stride + i4 * 4;
        int s_ = s0;
        for (; s_ + 4 <= s1; s_ += 4, src += 4 * stride) {     // four loads in flight, added in split order
            const float4 v0 = *(const float4*)src, v1 = *(const float4*)(src + stride), v2 = *(const float4*)(src + 2 * stride),
                         v3 = *(const float4*)(src + 3 * stride);
            acc.x = (((acc.x + v0.x) + v1.x) + v2.x) + v3.x; acc.y = (((acc.y + v0.y) + v1.y) + v2.y) + v3.y;
            acc.z = (((acc.z + v0.z) + v1.z) + v2.z) + v3.z; acc.w = (((acc.w + v0.w) + v1.w) + v2.w) + v3.w;
        }
        for (; s_ < s1; ++s_, src += stride) {
            const float4 v = *(const float4*)src;
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    red[sl][col] = acc;
    __syncthreads();
    if (sl == 0 && i4 < n4) {
        float4 o = *(float4*)(dw + i4 * 4);
#pragma unroll
        for (int k = 0; k < RED_LANES; ++k) { o.x += red[k][col].x; o.y += red[k][col].y; o.z += red[k][col].z; o.w += red[k][col].w; }
        *(float4*)(dw + i4 * 4) = o;
    }
}

template <typename T, int CT, int NT>
int launch(const WgradParams& p, int splits, hipStream_t stream) {
    static int use_dlds = vince_knob("wgrad_dlds", 1);
    if (use_dlds && p.in_bytes && p.dy_bytes && p.variant == 0) {
        if (use_dlds == 2)
            hipLaunchKernelGGL((conv_wgrad_dlds_kernel<T, CT, NT, 4>), dim3(p.ctiles * p.ntiles, splits), dim3(256), 0, stream, p);
        else
            hipLaunchKernelGGL((conv_wgrad_dlds_kernel<T, CT, NT, 3>), dim3(p.ctiles * p.ntiles, splits), dim3(256), 0, stream, p);
        VINCE_CHECK_LAUNCH();
        return VINCE_OK;
    }
    hipLaunchKernelGGL((conv_wgrad_kernel<T, CT, NT>), dim3(p.ctiles * p.ntiles, splits), dim3(256), 0, stream, p);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

// scratch / scratch_bytes: the slab buffer of the deterministic mode (null: atomics); need_out: when non-null only the bytes the
// default split count wants are computed, nothing is launched
template <typename T>
int dispatch(WgradParams& p, hipStream_t stream, void* scratch = nullptr, size_t scratch_bytes = 0, size_t* need_out = nullptr) {
    const vince_conv_desc& d = p.d;
    const int ntot = d.TA * d.TB * d.Ci;
    int CT = d.Co <= 64 ? 64 : 128, NT = ntot <= 64 ? 64 : 128;
    // bf16: the transposing-read kernel of conv_wgrad_tr.hip with its own tile choice, whenever the layer qualifies (`wgrad_tr=0`: the
    // LDS-DMA kernel of this file, cross-check switch)
    const int use_tr = (int)vince_knob_live("wgrad_tr", 1);
    int tr_ct = 0, tr_nt = 0;
    if (use_tr && std::is_same<T, bf16_t>::value) vince_wgrad::wgrad_tr_tile(p, &tr_ct, &tr_nt);
    // split-half products: the kernel of conv_wgrad_x3.hip (the addressing of conv_wgrad_tr on fp32 tiles) whenever the layer qualifies
    // (`wgrad_x3=0`: the LDS-DMA kernel of this file instantiated for x3b_t, cross-check switch)
    bool x3k = false;
    if (X3<T>::on && vince_knob_live("wgrad_x3", 1)) {
        vince_wgrad::wgrad_x3_tile(p, &tr_ct, &tr_nt);
        x3k = tr_ct > 0;
    }
    const bool tr = tr_ct > 0;
    if (tr) { CT = tr_ct; NT = tr_nt; }
    // (Measured and not kept, round 3: the LDS-DMA kernel as 8 wavefronts on a 256 x 256 tile for layer3 / layer4's 3x3 -- half the operand
    // bytes per FLOP, one workgroup per CU -- 175 against 122 us: without a staggered schedule the lone lock-step workgroup loses more
    // than the traffic gives back.  The template keeps its NW parameter.)
    p.ctiles = (d.Co + CT - 1) / CT;
    p.ntiles = (ntot + NT - 1) / NT;
    static int use_dlds_d = vince_knob("wgrad_dlds", 1);
    const bool dlds = tr || (use_dlds_d && p.in_bytes && p.dy_bytes && p.variant == 0);
    const int kp = dlds ? KPD : KP;                 // pixels per K slice of the kernel that will run
    static_assert(KPD == vince_wgrad::TR_SLICE, "slice");
    p.nkt_total = (p.M + kp - 1) / kp;
    // Split the pixel range so that the grid is about one resident wave of workgroups (256 CUs x 2): every extra split
    // costs Co*T*Ci fp32 atomics in the epilogue, and the L2 atomic rate -- not MFMA -- bounds this kernel when the grid
    // is cut 3x finer.  At least 8 K-tiles per split.
    static int target_blocks = VINCE_MEASURE_KNOB("wgrad_blocks", 512);
    const int tiles = p.ctiles * p.ntiles;
    // a multi-tap layer with a small output (layer1's 3x3: 64 x 576 floats in 5 tiles) is bound by its loop, not by atomics: twice
    // the workgroups hide twice the latency (192 -> 134 us timed alone; every other shape is best at 512)
    int target = target_blocks;
    if (d.TA * d.TB > 1 && (long)d.Co * ntot <= 65536) target *= 2;
    if (x3k) {
        // the split-half kernel is bound by its in-register splits, not by atomics: multi-tap layers take twice the workgroups
        // (measured alone at N = 256: 3x3 layers 431 / 306 / 292 / 267 us at 512 -> 398 / 248 / 267 / 237 at 1024; the 1x1 layers are
        // best at 512: 105-122 against 115-134)
        static const int x3_blocks = (int)vince_knob("x3_wgrad_blocks", 0);
        const int base = x3_blocks > 0 ? x3_blocks : (d.TA * d.TB > 1 ? 1024 : 512);
        target = base * ((d.TA * d.TB > 1 && (long)d.Co * ntot <= 65536) ? 2 : 1);
    }
    int splits = (target + tiles - 1) / tiles;
    const int max_splits = (p.nkt_total * kp / 64 + 7) / 8;   // at least 512 pixels per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    p.kt_per_split = (p.nkt_total + splits - 1) / splits;
    splits = (p.nkt_total + p.kt_per_split - 1) / p.kt_per_split;
    // XCD grouping pays most when many tiles share a pixel range (kernels timed alone: -15 % at 16-36 tiles, +12 % at 5
    // tiles; in the overlapped step always-on measured best, so the threshold stays a measurement knob)
    static const int xcd_min_tiles = VINCE_MEASURE_KNOB("wgrad_xcd_min_tiles", 1);
    p.xcd_group = p.xcd_group && tiles >= xcd_min_tiles;
    p.slab = nullptr;
    p.slab_stride = 0;
    // deterministic mode: plain forward tap order (every [Co][WT][Ci_dw] position is produced by exactly one tile), float4-sized
    const size_t dw_floats = (size_t)d.Co * d.WT * (d.Cs > 0 ? d.Kw : 1) * p.Ci_dw;
    const bool plain_taps = d.wt0 == 0 && d.wtb == 1 && d.wta == d.TB && d.WT == d.TA * d.TB;
    const bool det_ok = plain_taps && dw_floats % 4 == 0 && ((uintptr_t)p.dw & 15) == 0;
    if (need_out) {
        *need_out = det_ok ? (size_t)splits * dw_floats * sizeof(float) : 0;
        return VINCE_OK;
    }
    if (scratch && det_ok) {
        while (splits > 1 && (size_t)splits * dw_floats * sizeof(float) > scratch_bytes) {   // a short scratch buffer: fewer, longer splits
            --splits;
            p.kt_per_split = (p.nkt_total + splits - 1) / splits;
            splits = (p.nkt_total + p.kt_per_split - 1) / p.kt_per_split;
        }
        if ((size_t)splits * dw_floats * sizeof(float) <= scratch_bytes) {
            p.slab = (float*)scratch;
            p.slab_stride = dw_floats;
        }
    }
    int rc;
    if (x3k) rc = vince_wgrad::wgrad_x3_launch(p, CT, NT, splits, X3<T>::single, stream);
    else if (tr) rc = vince_wgrad::wgrad_tr_launch(p, CT, NT, splits, stream);
    else     if (CT == 64 && NT == 64) rc = launch<T, 64, 64>(p, splits, stream);
    else if (CT == 64) rc = launch<T, 64, 128>(p, splits, stream);
    else if (NT == 64) rc = launch<T, 128, 64>(p, splits, stream);
    else rc = launch<T, 128, 128>(p, splits, stream);
    if (rc == VINCE_OK && p.slab) rc = vince_wgrad::slab_reduce(p.slab, splits, p.slab_stride, p.dw, dw_floats / 4, stream);
    return rc;
}

}  // namespace

int vince_wgrad::slab_reduce(const float* slab, int splits, size_t stride, float* dst, size_t n4, hipStream_t stream) {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n4 + RED_COLS - 1) / RED_COLS)), dim3(256), 0, stream, slab, splits, stride, dst, n4);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

static int wgrad_common(const vince_conv_desc* dd, int dtype, const void* in, const void* dy, float* dw, int32_t Ci_dw, int variant,
                        void* scratch, size_t scratch_bytes, size_t* need_out, void* stream);

extern "C" int vince_conv_wgrad(const vince_conv_desc* dd, int dtype, const void* in, const void* dy, float* dw,
                                int32_t Ci_dw, int variant, void* stream) {
    return wgrad_common(dd, dtype, in, dy, dw, Ci_dw, variant, nullptr, 0, nullptr, stream);
}

extern "C" int vince_conv_wgrad_det(const vince_conv_desc* dd, int dtype, const void* in, const void* dy, float* dw, int32_t Ci_dw,
                                    void* scratch, size_t scratch_bytes, void* stream) {
    VINCE_CHECK_ARG(!scratch || ((uintptr_t)scratch & 15) == 0, VINCE_E_ALIGN, "vince_conv_wgrad_det: scratch must be 16-byte aligned");
    return wgrad_common(dd, dtype, in, dy, dw, Ci_dw, 0, scratch, scratch_bytes, nullptr, stream);
}

extern "C" size_t vince_conv_wgrad_scratch_bytes(const vince_conv_desc* dd, int dtype, int32_t Ci_dw) {
    size_t need = 0;
    static float dummy_dw[4] __attribute__((aligned(16)));
    if (wgrad_common(dd, dtype, dummy_dw, dummy_dw, dummy_dw, Ci_dw, 0, nullptr, 0, &need, nullptr) != VINCE_OK) return 0;
    return need;
}

static int wgrad_common(const vince_conv_desc* dd, int dtype, const void* in, const void* dy, float* dw, int32_t Ci_dw, int variant,
                        void* scratch, size_t scratch_bytes, size_t* need_out, void* stream) {
    VINCE_CHECK_ARG(dd && in && dy && dw, VINCE_E_ARG, "vince_conv_wgrad: null pointer");
#if defined(VINCE_MEASURE) || defined(VINCE_STEP_ABLATE)
    {   // measurement builds (-DVINCE_STEP_ABLATE: the product kernels + this switch): 1 = no weight-gradient launch at all, 2 = only
        // the Gram matrices (in == dy) run -- what the family costs inside the overlapped step
        static const long skip = getenv("VINCE_WGRAD_SKIP") ? atol(getenv("VINCE_WGRAD_SKIP")) : 0;
        if (!need_out && (skip == 1 || (skip == 2 && in != dy))) return VINCE_OK;
    }
#endif
    if (dtype == VINCE_F32X3H) dtype = VINCE_F32X3B;   // weight gradients always split into bfloat16 halves (the operand dy spans many decades)
    VINCE_CHECK_ARG(dtype == VINCE_F32 || dtype == VINCE_BF16 || dtype == VINCE_F32X3B || dtype == VINCE_F32X1B, VINCE_E_DTYPE, "vince_conv_wgrad: bad dtype %d", dtype);
    const vince_conv_desc& d = *dd;
    const bool f32_store = dtype != VINCE_BF16;
    const int CH = f32_store ? 4 : 8;
    VINCE_CHECK_ARG(d.N > 0 && d.Hi > 0 && d.Wi > 0 && d.Ho > 0 && d.Wo > 0 && d.Co > 0 && d.Ci > 0, VINCE_E_SHAPE,
                    "vince_conv_wgrad: non-positive dimension");
    VINCE_CHECK_ARG(d.Ci % CH == 0 && d.Co % CH == 0, VINCE_E_SHAPE, "vince_conv_wgrad: Ci=%d / Co=%d not multiples of %d",
                    d.Ci, d.Co, CH);
    VINCE_CHECK_ARG(Ci_dw >= 1 && Ci_dw <= (d.Cs > 0 ? d.Cs : d.Ci), VINCE_E_SHAPE, "vince_conv_wgrad: Ci_dw=%d out of range", Ci_dw);
    if (d.Cs > 0) {
        const int eb = f32_store ? 4 : 2;
        VINCE_CHECK_ARG(d.TB == 1 && d.Cs < d.Ci && d.Ci % d.Cs == 0 && d.Kw > 0 && d.Kw <= d.Ci / d.Cs, VINCE_E_SHAPE,
                        "vince_conv_wgrad: packed row taps need TB=1, Cs | Ci, 0 < Kw <= Ci/Cs");
        VINCE_CHECK_ARG((d.Cs * eb) % 8 == 0 && (d.sw * d.Cs * eb) % 16 == 0 && (d.dw0 * d.Cs * eb) % 16 == 0 &&
                        (d.Wi * d.Cs * eb) % 16 == 0, VINCE_E_ALIGN, "vince_conv_wgrad: packed row taps must start 16-byte aligned");
        VINCE_CHECK_ARG(d.dw0 >= 0 && (d.Wo - 1) * d.sw + d.dw0 + d.Ci / d.Cs <= d.Wi, VINCE_E_SHAPE,
                        "vince_conv_wgrad: packed row taps must stay inside the (padded) input row");
    }
    VINCE_CHECK_ARG(d.TA >= 1 && d.TB >= 1 && d.TB <= 8 && d.TA * d.TB <= 64, VINCE_E_SHAPE,
                    "vince_conv_wgrad: tap grid %dx%d unsupported", d.TA, d.TB);
    VINCE_CHECK_ARG((long long)d.N * d.Ho * d.Wo < (1ll << 31), VINCE_E_SHAPE, "vince_conv_wgrad: too many pixels");
    VINCE_CHECK_ARG((((uintptr_t)in | (uintptr_t)dy) & 15) == 0, VINCE_E_ALIGN, "vince_conv_wgrad: pointers must be 16-byte aligned");
    WgradParams p;
    p.d = d;
    const int T = d.TA * d.TB, cpt = d.Ci / CH;
    if (T == 1) {
        p.log2_cpt = 31;
        p.cpt_mask = 0x7fffffff;
        p.log2_ci = 30;
    } else {
        VINCE_CHECK_ARG((cpt & (cpt - 1)) == 0, VINCE_E_SHAPE, "vince_conv_wgrad: Ci/%d = %d must be a power of two", CH, cpt);
        int l = 0;
        while ((1 << l) < cpt) ++l;
        p.log2_cpt = l;
        p.cpt_mask = cpt - 1;
        p.log2_ci = l + (f32_store ? 2 : 3);
    }
    p.total_nchunks = T * cpt;
    p.M = d.N * d.Ho * d.Wo;
    p.tb_mul = (65536 + d.TB - 1) / d.TB;
    p.div_howo = make_fastdiv((uint32_t)(d.Ho * d.Wo));
    p.div_wo = make_fastdiv((uint32_t)d.Wo);
    p.in = in; p.dy = dy; p.dw = dw; p.Ci_dw = Ci_dw; p.variant = variant;
    p.cs = d.Cs > 0 ? d.Cs : d.Ci;
    p.ablate = (int)VINCE_MEASURE_KNOB("wgrad_ablate", 0);
#ifdef VINCE_STEP_ABLATE
    {
        static const int abl = getenv("VINCE_WGRAD_ABLATE") ? atoi(getenv("VINCE_WGRAD_ABLATE")) : 0;
        p.ablate = in != dy ? abl : 0;     // the Gram matrices stay exact: the forward's statistics depend on them
    }
#endif
    static const int xcd_group = (VINCE_MEASURE_KNOB("wgrad_xcd", 1) != 0);
    p.xcd_group = xcd_group;
    {
        const unsigned long long esz = f32_store ? 4 : 2;
        const unsigned long long ib = (unsigned long long)d.N * d.Hi * d.Wi * p.cs * esz, yb = (unsigned long long)p.M * d.Co * esz;
        p.in_bytes = ib < 0x7ff00000ull ? (uint32_t)ib : 0;
        p.dy_bytes = yb < 0x7ff00000ull ? (uint32_t)yb : 0;
        p.linear_x = (T == 1 && d.sh == 1 && d.sw == 1 && d.dh0 == 0 && d.dw0 == 0 && d.Hi == d.Ho && d.Wi == d.Wo) ? 1 : 0;
    }
    hipStream_t s = (hipStream_t)stream;
    void* tok = nullptr;
    if (vince_profile_enabled() && !need_out)
    {
        vince_profile_begin_launch(f32_store ? 16 : 17, 2.0 * p.M * d.Co * T * (double)Ci_dw * (d.Cs > 0 ? d.Kw : 1), stream, &tok);
        vince_profile_set_dims(tok, p.M, d.Co, T * d.Ci, T, d.sh, 0);
    }
    const int rc = dtype == VINCE_F32 ? dispatch<float>(p, s, scratch, scratch_bytes, need_out)
                 : dtype == VINCE_F32X3B ? dispatch<x3b_t>(p, s, scratch, scratch_bytes, need_out)
                 : dtype == VINCE_F32X1B ? dispatch<x1b_t>(p, s, scratch, scratch_bytes, need_out)
                                         : dispatch<bf16_t>(p, s, scratch, scratch_bytes, need_out);
    if (tok) vince_profile_end_launch(tok, stream);
    return rc;
}
