// Shared pieces of the implicit-GEMM convolution kernels (conv_igemm.hip: 4-wavefront tiles; conv_m8.hip: the 8-wavefront
// 256 x 256 tile): launch parameters, the MFMA wrapper and the epilogue (accumulators -> LDS -> coalesced NHWC stores with the
// fused bias / ReLU / BatchNorm-statistics / residual-join / BatchNorm-backward-reduction options of vince_conv_epi).
#pragma once
#include <type_traits>

#include "common.h"

namespace vince_conv {

struct ConvParams {
    vince_conv_desc d;
    int log2_cpt, cpt_mask, total_chunks, nkt, M, ptiles, ctiles, uniform_taps, ablate;
    int cs;             // element stride between input pixels (= Ci unless the descriptor packs row taps)
    int kt_per_split;   // > 0: split-K (grid.y splits, fp32 atomics into a zeroed output; f32 only)
    int kt_per_tap;     // conv_m8: K tiles of 64 elements per tap (Ci / 64)
    int ktpt_mask, log2_ktpt;
    int presplit;  // split-half forward launches: the activation operand is stored IEEE-half pairs (VINCE_EPI_IN_HALF_PAIRS): no split in the loop
    int variant;   // host side: which kernel the launcher picked (0 = 128-pixel tile, 1 = 256-pixel tile, 2 = register-staged)
    uint32_t tb_mul;
    FastDiv div_howo, div_wo;
    const void* in;
    const void* w;
    void* out;
    uint32_t in_bytes, w_bytes;   // buffer-descriptor ranges for the direct-to-LDS variant (0 = tensor too large)
    const void* in2;    // second input tensor of the last tap (vince_conv_epi.in2), its byte range and channel stride
    uint32_t in2_bytes;
    int cs2;
    int log2_cpt2, cpt2_mask;   // in2 read several times over (vince_conv_epi.in2_repeat): 16-byte chunks per in2 row (31 / 0x7fffffff: once)
    int log2_tapid;             // K chunks per stretch of linearly advancing offsets: log2_cpt, or log2_cpt2 with in2_repeat
    vince_conv_epi e;   // epilogue options (bias, statistics, residual join, fused BatchNorm forward / backward-reduce)
};

}  // namespace vince_conv

// conv_m8.hip: the 8-wavefront 256 x 256 core (bf16).  VINCE_M8_NOT_ELIGIBLE (positive: outside the ABI's error codes, which are <= 0)
// = the shape does not qualify, the caller keeps its own tiles.
#define VINCE_M8_NOT_ELIGIBLE 1
int vince_conv_m8_launch(vince_conv::ConvParams& p, int mode, hipStream_t stream);
// conv_igemm_x3.hip: the conv_igemm kernels instantiated for the split-half element types (dtype VINCE_F32X3H / VINCE_F32X3B)
int vince_conv_igemm_x3_launch(vince_conv::ConvParams& p, int dtype, int mode, bool narrow, hipStream_t stream);

namespace {
using vince_conv::ConvParams;

#ifdef VINCE_MEASURE   // measurement build: VINCE_CONV_ABLATE bits 32 (no output stores) and 8 (no statistics atomics)
#define CORE_ABL(bit) (p.ablate & (bit))
#else
#define CORE_ABL(bit) 0
#endif

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    __device__ static inline void run(const uint4& a, const uint4& b, f32x16_t& c) {
        bf16x8_t av, bv;
        __builtin_memcpy(&av, &a, 16);
        __builtin_memcpy(&bv, &b, 16);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    // the 4 floats of a fragment are 4 different k; lanes 0-31 / 32-63 carry k and k+4 -- any pairing of k between
    // the two halves is fine as long as A and B use the same one.
    __device__ static inline void run(const uint4& a, const uint4& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};

// x3h_t scales its operands by 2^X3_WSHIFT and 2^X3_XSHIFT before splitting them (common.h): the accumulators go back by the product
template <typename T, int A, int B> __device__ __forceinline__ void x3_unscale(f32x16_t (&acc)[A][B]) {
    if constexpr (X3<T>::on && X3<T>::half) {
#pragma unroll
        for (int j = 0; j < A; ++j)
#pragma unroll
            for (int i = 0; i < B; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][i][e] *= 1.f / (float)(1 << (X3_WSHIFT + X3_XSHIFT));
    }
}

// MODE: 0 = forward (bias / ReLU / statistics), 1 = gradient epilogues (residual-gradient join through acc_mask, fused BatchNorm-
// backward reduction), 2 = forward residual join with known BatchNorm constants (out_scale / bias / id_scale / id_shift / ReLU,
// in place).  Separate instantiations keep each one's per-thread constant arrays -- and so its registers -- to what it uses.
// NTHR threads as (NTHR / 64 / WN) x WN wavefronts (pixel groups x channel groups); wavefront (wp, wc) holds PTL / WP pixels x
// CT / WN channels as 32 x 32 MFMA tiles acc[channel tile][pixel tile].
template <typename T, int CT, int CRS, int MODE, int PTL = 128, int UBM = 4, int NTHR = 256, int WN = 2>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, unsigned char* smem,
                                              f32x16_t (&acc)[CT / (32 * WN)][PTL / (32 * (NTHR / 64 / WN))],
                                              uint32_t tile, int p0, int c0, int tid, int lane, int wave, int wp, int wc) {
    constexpr int CH = Elem<T>::CH;
    constexpr int NW = NTHR / 64, WP = NW / WN;
    constexpr int CJ = CT / (32 * WN), PI = PTL / (32 * WP);
    constexpr bool BWD = MODE == 1, JOIN = MODE == 2;
    const vince_conv_desc& d = p.d;
    // ---- epilogue: accumulators -> LDS [pixel][channel] as T -> coalesced 16-byte stores --------------------
    // The bias of a 16-bit GRADIENT launch goes onto the fp32 ACCUMULATORS, before the one rounding to T.  Added to the staged 16-bit value
    // instead (rounds 3-5), a bias that cancels most of the accumulator -- the constant of the BatchNorm-backward algebra's input gradient,
    // csrc/bn_algebra.hip, the one gradient launch with a bias -- leaves the sum on the coarse grid of the LARGER number, and where the
    // spread of the result is a few of those steps the rounding error stops averaging out over pixels: the per-channel sums the BatchNorm
    // below reduces were 1e-1 off (round 6, tools/alg_op_probe.py: 1.6e-1 -> 2e-3 with this and the hi + lo matrices).  Forward launches
    // (the folded inference path's bias + ReLU, bit-compatible with the streaming kernels) and fp32 tensors (no second rounding) keep theirs.
    constexpr bool BIAS_FIRST = BWD && sizeof(T) == 2;
    const bool bias_first = BIAS_FIRST && p.e.bias != nullptr && blockIdx.y == 0;
    auto stage = [&](auto with_bias) {
#pragma unroll
        for (int j = 0; j < CJ; ++j)
#pragma unroll
            for (int i = 0; i < PI; ++i) {
                const int pix = wp * (PTL / WP) + i * 32 + (lane & 31);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = wc * (CT / WN) + j * 32 + 8 * g + 4 * (lane >> 5);
                    unsigned char* dst = smem + pix * CRS + ch * (int)sizeof(T);
                    float v0 = acc[j][i][4 * g], v1 = acc[j][i][4 * g + 1], v2 = acc[j][i][4 * g + 2], v3 = acc[j][i][4 * g + 3];
                    if constexpr (decltype(with_bias)::value) {
                        const int cg = c0 + ch;
                        v0 += cg < d.Co ? p.e.bias[cg] : 0.f;
                        v1 += cg + 1 < d.Co ? p.e.bias[cg + 1] : 0.f;
                        v2 += cg + 2 < d.Co ? p.e.bias[cg + 2] : 0.f;
                        v3 += cg + 3 < d.Co ? p.e.bias[cg + 3] : 0.f;
                    }
                    if constexpr (sizeof(T) == 4) *(float4*)dst = make_float4(v0, v1, v2, v3);
                    else *(uint2*)dst = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
                }
            }
    };
    if constexpr (BIAS_FIRST) {
        if (bias_first) stage(std::true_type{});       // (uniform; the launches without a bias keep the plain loop)
        else stage(std::false_type{});
    } else {
        stage(std::false_type{});
    }
    __syncthreads();

    constexpr int CPR = CT * (int)sizeof(T) / 16;   // 16-byte chunks per tile row
    constexpr int RPP = NTHR / CPR;                 // rows per pass
    const int chunk = tid % CPR, row0 = tid / CPR;
    const int cbase = c0 + chunk * CH;
    const bool cvalid = cbase < d.Co;
    float bias_v[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) bias_v[e] = (!BIAS_FIRST && p.e.bias && cvalid && blockIdx.y == 0) ? p.e.bias[cbase + e] : 0.f;
    // residual join with known BatchNorm constants (MODE 2): conv * osc + bias + (old * isc + ish)
    float osc_v[CH], isc_v[CH], ish_v[CH];
    const bool id_affine = JOIN && p.e.id_scale != nullptr;
    if constexpr (JOIN) {
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            osc_v[e] = (p.e.out_scale && cvalid) ? p.e.out_scale[cbase + e] : 1.f;
            isc_v[e] = (id_affine && cvalid) ? p.e.id_scale[cbase + e] : 1.f;
            ish_v[e] = (id_affine && cvalid) ? p.e.id_shift[cbase + e] : 0.f;
        }
    }
    float rmu_v[CH];
    if constexpr (sizeof(T) == 4 && JOIN) {
#pragma unroll
        for (int e = 0; e < CH; ++e) rmu_v[e] = (p.e.raw2 && cvalid) ? p.e.raw2_mean[cbase + e] : 0.f;
    }
    float ssum[CH], ssq[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) ssum[e] = ssq[e] = 0.f;
    T* __restrict__ out = (T*)p.out;
    const bool identity_map = (d.osh == 1 && d.osw == 1 && d.oh0 == 0 && d.ow0 == 0 && d.OH == d.Ho && d.OW == d.Wo);
    const int flags = p.e.flags;
    // BWD (compile time): the gradient epilogues -- residual join (ACCUMULATE, acc_mask) and the fused BatchNorm-backward
    // reduction (bnred).  Forward launches take the lean instantiation.
    const bool accum = (BWD || JOIN) && (flags & VINCE_EPI_ACCUMULATE) != 0;
    const bool touch = (!BIAS_FIRST && p.e.bias) || accum || (flags & VINCE_EPI_RELU) || (BWD && p.e.out_mask);
    const T* __restrict__ br_y = BWD ? (const T*)p.e.bnred.y : nullptr;
    float br_mu[CH], br_is[CH], br_sc[CH], br_sh[CH];
    if constexpr (BWD) {
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            const bool on = br_y && cvalid;
            br_mu[e] = on ? p.e.bnred.mean[cbase + e] : 0.f;
            br_is[e] = on ? p.e.bnred.invstd[cbase + e] : 0.f;
            br_sc[e] = (on && p.e.bnred.mask_scale) ? p.e.bnred.mask_scale[cbase + e] : 0.f;
            br_sh[e] = (on && p.e.bnred.mask_scale) ? p.e.bnred.mask_shift[cbase + e] : 0.f;
        }
    }
    constexpr int NR = PTL / RPP;            // rows this thread stores
    constexpr int UB = NR < UBM ? NR : UBM;  // rows per batch: every global load of a batch is issued before its arithmetic
    for (int rb = 0; rb < NR; rb += UB) {
        size_t off[UB];
        bool ok[UB];
        uint4 oldv[UB], yv[UB];
        uint32_t ab[UB], bb[UB], om[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int row = row0 + (rb + u) * RPP;
            const uint32_t m = p0 + row;
            ok[u] = cvalid && m < (uint32_t)p.M && (NR % UB == 0 || rb + u < NR);     // (a batch size that does not divide the rows: ragged last batch)
            size_t opix = m;
            if (!identity_map) {
                uint32_t n = fastdiv(m, p.div_howo);
                uint32_t rem = m - n * p.div_howo.d;
                uint32_t ho = fastdiv(rem, p.div_wo);
                uint32_t wo = rem - ho * p.div_wo.d;
                opix = ((size_t)n * d.OH + (ho * d.osh + d.oh0)) * d.OW + (wo * d.osw + d.ow0);
            }
            off[u] = opix * d.Co + cbase;
            if constexpr (JOIN) {
                if (ok[u] && accum) oldv[u] = *(const uint4*)(out + off[u]);
            }
            if constexpr (BWD) {
                ab[u] = bb[u] = 0xffu;
                om[u] = 0xffu;
                if (ok[u] && p.e.out_mask) om[u] = p.e.out_mask[off[u] / CH];
                if (ok[u]) {
                    if (accum) {
                        oldv[u] = *(const uint4*)(out + off[u]);
                        if (p.e.acc_mask) ab[u] = p.e.acc_mask[off[u] / CH];
                    }
                    if (br_y) {
                        yv[u] = *(const uint4*)(br_y + off[u]);
                        if (p.e.bnred.mask_bits) bb[u] = p.e.bnred.mask_bits[off[u] / CH];
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (!ok[u]) continue;
            const int row = row0 + (rb + u) * RPP;
            uint4 v = *(const uint4*)(smem + row * CRS + chunk * 16);
            if constexpr (sizeof(T) == 4 && JOIN) {
                if (p.e.raw2) {   // (uniform) the raw convolution output, centred, as bfloat16: the backward's view of the BatchNorm applied below
                    *(uint2*)((bf16_t*)p.e.raw2 + off[u]) =
                        make_uint2(pack_bf16x2(__uint_as_float(v.x) - rmu_v[0], __uint_as_float(v.y) - rmu_v[1]),
                                   pack_bf16x2(__uint_as_float(v.z) - rmu_v[2], __uint_as_float(v.w) - rmu_v[3]));
                }
            }
            if (touch) {
                float f[CH];
                Chunk<T>::unpack(v, f);
                if constexpr (JOIN) {
#pragma unroll
                    for (int e = 0; e < CH; ++e) f[e] = f[e] * osc_v[e] + bias_v[e];
                    if (accum) {
                        float o[CH];
                        Chunk<T>::unpack(oldv[u], o);
#pragma unroll
                        for (int e = 0; e < CH; ++e) f[e] += o[e] * isc_v[e] + ish_v[e];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < CH; ++e) f[e] += bias_v[e];
                }
                if constexpr (BWD) {
                    if (accum) {
                        float o[CH];
                        Chunk<T>::unpack(oldv[u], o);
                        // residual join: the old value passes through the ReLU of the block output (acc_mask bits)
#pragma unroll
                        for (int e = 0; e < CH; ++e) f[e] += ((ab[u] >> e) & 1u) ? o[e] : 0.f;
                    }
                }
                if constexpr (BWD) {
                    if (p.e.out_mask) {   // (uniform) the gradient is stored already gated by the ReLU it flows into next
#pragma unroll
                        for (int e = 0; e < CH; ++e) f[e] = ((om[u] >> e) & 1u) ? f[e] : 0.f;
                    }
                }
                if constexpr (sizeof(T) == 4 && JOIN) {
                    if (p.e.mask2) {   // (uniform) ReLU bits in the bf16 tensors' format: one byte per 8 channels = two neighbouring chunks
                        uint32_t b = 0;
#pragma unroll
                        for (int e = 0; e < CH; ++e) b |= (f[e] > 0.f ? 1u : 0u) << e;
                        const uint32_t hi = __shfl_xor(b, 1, 64);      // (chunks c, c ^ 1 of one row sit in neighbouring lanes: CPR is even)
                        if ((chunk & 1) == 0) p.e.mask2[off[u] >> 3] = (uint8_t)(b | (hi << 4));
                    }
                }
                if (flags & VINCE_EPI_RELU) {
#pragma unroll
                    for (int e = 0; e < CH; ++e) f[e] = fmaxf(f[e], 0.f);
                }
                v = Chunk<T>::pack(f);
            }
            if constexpr (sizeof(T) == 4 && (MODE == 0 || JOIN)) {
                if (p.e.out2) {   // (uniform) bfloat16 shadow of the stored value: what the mixed mode's bf16 backward reads
                    *(uint2*)((bf16_t*)p.e.out2 + off[u]) =
                        make_uint2(pack_bf16x2(__uint_as_float(v.x), __uint_as_float(v.y)), pack_bf16x2(__uint_as_float(v.z), __uint_as_float(v.w)));
                }
            }
            if constexpr (sizeof(T) == 4) {
                if (p.kt_per_split > 0) {   // split-K partial: accumulate into the zeroed output
                    float f[CH];
                    Chunk<T>::unpack(v, f);
#pragma unroll
                    for (int e = 0; e < CH; ++e) unsafeAtomicAdd((float*)out + off[u] + e, f[e]);
                } else {
                    *(uint4*)(out + off[u]) = v;
                }
            } else {
                if (!CORE_ABL(32)) *(uint4*)(out + off[u]) = v;
            }
            if (p.e.stats) {
                float f[CH];
                Chunk<T>::unpack(v, f);
#pragma unroll
                for (int e = 0; e < CH; ++e) { ssum[e] += f[e]; ssq[e] += f[e] * f[e]; }
            }
            if constexpr (BWD) {
                if (br_y) {
                    // (sum g, sum g*xhat) of the STORED gradient g = v * relu-mask, exactly what vince_bn_bwd_reduce computes
                    float g[CH], yy[CH];
                    Chunk<T>::unpack(v, g);
                    Chunk<T>::unpack(yv[u], yy);
#pragma unroll
                    for (int e = 0; e < CH; ++e) {
                        bool keep = ((bb[u] >> e) & 1u) != 0;
                        if (p.e.bnred.mask_scale) keep = (yy[e] * br_sc[e] + br_sh[e]) > 0.f;
                        const float ge = keep ? g[e] : 0.f;
                        ssum[e] += ge;
                        ssq[e] += ge * (yy[e] - br_mu[e]) * br_is[e];
                    }
                }
            }
        }
    }
    double* const red_out = p.e.stats ? p.e.stats : (BWD ? p.e.bnred.sums : nullptr);
    if (red_out) {   // uniform branch
        float* red = (float*)(smem + PTL * CRS);      // [NW waves][CPR][CH][2]
#pragma unroll
        for (int e = 0; e < CH; ++e) {
#pragma unroll
            for (int o = CPR; o < 64; o <<= 1) {
                ssum[e] += __shfl_xor(ssum[e], o, 64);
                ssq[e] += __shfl_xor(ssq[e], o, 64);
            }
        }
        if (lane < CPR) {
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                red[((wave * CPR + lane) * CH + e) * 2 + 0] = ssum[e];
                red[((wave * CPR + lane) * CH + e) * 2 + 1] = ssq[e];
            }
        }
        __syncthreads();
        // CPR <= 32: chunk column ck sits in lane ck of every wave; thread t finalises (channel, which) = (t>>1, t&1)
        static_assert(CPR <= 32, "statistics reduction assumes at most 32 chunks per tile row");
        for (int t = tid; t < CT * 2; t += NTHR) {
            const int ch = t >> 1, which = t & 1;
            const int ck = ch / CH, e = ch % CH;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += red[((w * CPR + ck) * CH + e) * 2 + which];
            if (c0 + ch < d.Co && !CORE_ABL(8))
                unsafeAtomicAdd(red_out + ((size_t)(tile % (uint32_t)p.e.replicas) * d.Co + (c0 + ch)) * 2 + which, (double)s);
        }
    }
}

}  // namespace
