// BatchNorm2d (train / eval), fused BN+ReLU+residual apply, BN backward, stem max-pool, global average pool.
//
// Reference call sites: nn.BatchNorm2d resnet.py:69,72,110,112,171; ReLU / residual add resnet.py:81,89-90,122,126,
// 134-135; MaxPool2d(3,2,1) resnet.py:173; AdaptiveAvgPool2d vince_model.py:33.
//
// All of these are HBM-bound NHWC streams.  Every thread owns a fixed 16-byte channel chunk (8 bf16 / 4 f32) and
// walks pixel rows, so a wavefront always touches whole contiguous row segments and the per-channel constants
// (scale/shift/mean/invstd) live in registers for the whole walk.  Reductions go registers -> LDS -> one fp64
// atomic per channel per workgroup.
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "common.h"
#ifndef VINCE_BN_U
#define VINCE_BN_U 4      // rows in flight per thread of the forward BatchNorm pass and the backward reduction (A/B: -DVINCE_BN_U=8)
#endif

namespace {

// 2-D decomposition shared by the row-walking kernels.
struct RowWalk {
    int cpr;             // chunks per row = C / CH
    int tpc;             // threads across chunk columns (<= 256, divides 256)
    int rpp;             // rows per pass = 256 / tpc
    int colgroups;       // grid.x
    int rowblocks;       // grid.y
    int64_t rows_per_block;
    int nt;              // bit 0: non-temporal loads of the streamed inputs, bit 1: non-temporal stores (bn_nt_policy)
};

// 16-byte accesses with a COMPILE-TIME `nt` hint.  (Rounds 2-4 carried the hint as a run-time argument -- `nt ? nontemporal_load(p) : *p`
// -- which the compiler folds into one plain load: the ISA held no `nt` instruction and the "+-0.3 %" of round 2 measured nothing.)
// What the hint buys (tools/micro/mall_micro, profiles/r05_mall_micro.txt): a read-only pass over a 392 MB tensor the previous launch
// has just written runs at 4.0 TB/s with plain loads and 6.5 TB/s with nt loads; a copy 5.4 -> 5.9 TB/s.  In the step (same-box A/B,
// profiles/r05_nt_ab.txt): nt LOADS in the apply passes -0.18 ... -0.22 ms on every box and for every tensor size (bn_nt_min_mb = 0 beats
// 64, 128 loses the gain); nt stores -0.05 on one box, +0.25 on another (off); nt loads in the backward reduction neutral (off: the apply
// pass reads the same two tensors again).  Measured and NOT kept, same A/B: the hint on the stem passes' and bn_apply_gram's loads
// (neutral), on conv_xjoin's input LDS-DMA (neutral) and identity loads (+0.5 ms: the identity is the previous launch's output and
// still cached), on conv_igemm's single-tap input LDS-DMA (+0.45 ms).
template <bool NT>
__device__ __forceinline__ uint4 ld16(const void* ptr) {
    if constexpr (NT) {
        const f32x4_t v = __builtin_nontemporal_load((const f32x4_t*)ptr);
        uint4 r;
        __builtin_memcpy(&r, &v, 16);
        return r;
    } else {
        return *(const uint4*)ptr;
    }
}
template <bool NT>
__device__ __forceinline__ void st16(void* ptr, const uint4& v) {
    if constexpr (NT) {
        f32x4_t t;
        __builtin_memcpy(&t, &v, 16);
        __builtin_nontemporal_store(t, (f32x4_t*)ptr);
    } else {
        *(uint4*)ptr = v;
    }
}
// bit 0: nt loads of the streamed inputs, bit 1: nt stores.  VINCE_KNOBS: bn_nt (policy bits for tensors of at least bn_nt_min_mb MB)
inline int bn_nt_policy(int64_t rows, int C, int esize) {
    static const int mask = (int)vince_knob("bn_nt", 1);
    static const long long min_bytes = (long long)vince_knob("bn_nt_min_mb", 0) << 20;
    return (long long)rows * C * esize >= min_bytes ? (mask & 3) : 0;
}

inline RowWalk make_rowwalk(int64_t rows, int C, int CH, int target_blocks = 2048) {
    RowWalk w;
    w.cpr = C / CH;
    int tpc = 1;
    while (tpc < w.cpr && tpc < 256) tpc <<= 1;
    w.tpc = tpc;
    w.rpp = 256 / tpc;
    w.colgroups = (w.cpr + tpc - 1) / tpc;
    int64_t want = target_blocks / w.colgroups;
    if (want < 1) want = 1;
    int64_t rpb = (rows + want - 1) / want;
    int64_t minrows = (int64_t)w.rpp * 8;
    if (rpb < minrows) rpb = minrows;
    rpb = (rpb + w.rpp - 1) / w.rpp * w.rpp;
    w.rows_per_block = rpb;
    w.rowblocks = (int)((rows + rpb - 1) / rpb);
    w.nt = bn_nt_policy(rows, C, CH == 4 ? 4 : 2);
    return w;
}


// How the backward kernels obtain the ReLU mask of the activation that followed this BatchNorm:
//   bits  : 1 byte per 16-byte chunk written by bn_apply (bit e = output e > 0)  -- residual outputs
//   from y: sign of y*scale + shift recomputed from the conv output that is read anyway -- plain BN+ReLU
//   src   : sign of a materialised activation tensor (2 bytes / element; legacy, tests)
struct MaskArgs {
    const void* src;
    const uint8_t* bits;
    const float* scale;
    const float* shift;
};

template <typename T, int CH>
__device__ __forceinline__ void apply_mask(const MaskArgs& ma, size_t off, size_t chunk_index, const float (&yy)[CH],
                                           const float (&msc)[CH], const float (&msh)[CH], float (&g)[CH]) {
    if (ma.bits) {
        const uint32_t b = ma.bits[chunk_index];
#pragma unroll
        for (int e = 0; e < CH; ++e) g[e] = ((b >> e) & 1u) ? g[e] : 0.f;
    } else if (ma.scale) {
#pragma unroll
        for (int e = 0; e < CH; ++e) g[e] = (yy[e] * msc[e] + msh[e]) > 0.f ? g[e] : 0.f;
    } else if (ma.src) {
        float m[CH];
        Chunk<T>::unpack(*(const uint4*)((const T*)ma.src + off), m);
#pragma unroll
        for (int e = 0; e < CH; ++e) g[e] = m[e] > 0.f ? g[e] : 0.f;
    }
}

__global__ void bn_finalize_kernel(const double* __restrict__ stats, double count, int C, const float* gamma,
                                   const float* beta, float* rmean, float* rvar, int64_t* nbt, float momentum,
                                   float eps, int train, float* scale, float* shift, float* save_mean,
                                   float* save_invstd) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && train && nbt) *nbt += 1;
    if (c >= C) return;
    float mean, invstd;
    if (train) {
        double s1 = 0, s2 = 0;
        for (int r = 0; r < VINCE_STATS_REPLICAS; ++r) {
            s1 += stats[((size_t)r * C + c) * 2];
            s2 += stats[((size_t)r * C + c) * 2 + 1];
        }
        double m = s1 / count;
        double var = s2 / count - m * m;
        if (var < 0) var = 0;
        mean = (float)m;
        invstd = (float)(1.0 / sqrt(var + (double)eps));
        if (rmean) {
            double unbiased = count > 1 ? var * count / (count - 1) : var;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
        }
    } else {
        mean = rmean[c];
        invstd = 1.f / sqrtf(rvar[c] + eps);
    }
    float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
    if (save_mean) save_mean[c] = mean;
    if (save_invstd) save_invstd[c] = invstd;
}

template <typename T, int NT>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const T* __restrict__ idn,
                                                       const float* __restrict__ ids, const float* __restrict__ idt,
                                                       T* __restrict__ out, uint8_t* __restrict__ mask_out, int64_t rows,
                                                       int C, int relu, RowWalk w, const vince_bn_train fin) {
    constexpr int CH = Elem<T>::CH;
    __shared__ float cst[sizeof(T) == 4 ? 3 : 2][256 * CH];   // fused finalize: (scale, shift[, mean: fp32 launches, for the centred shadow]) of this block's channels
    const int col = blockIdx.x * w.tpc + (threadIdx.x % w.tpc);
    if (fin.stats) {
        // train-mode finalize fused in (vince_bn_train_apply): every workgroup folds the statistic replicas of ITS channels
        // (a few KB from L2) instead of a separate launch per BatchNorm; row-block 0 also publishes the constants the
        // backward pass needs and updates the running statistics, exactly once.
        const int nch = w.tpc * CH, c_base = blockIdx.x * nch;
        for (int cl = threadIdx.x; cl < nch; cl += 256) {
            const int c = c_base + cl;
            if (c >= C) continue;
            double s1, s2;
            fold_replicas(fin.stats, fin.replicas, C, c, s1, s2);
            const double cnt = (double)fin.count;
            const double m = s1 / cnt;
            double var = s2 / cnt - m * m;
            if (var < 0) var = 0;
            const float mean = (float)m;
            const float invstd = (float)(1.0 / sqrt(var + (double)fin.eps));
            const float scv = fin.gamma[c] * invstd;
            const float shv = fin.beta[c] - mean * scv;
            cst[0][cl] = scv;
            cst[1][cl] = shv;
            if constexpr (sizeof(T) == 4) cst[2][cl] = mean;
            if (blockIdx.y == 0) {
                if constexpr (sizeof(T) == 4) {
                    if (fin.shadow_consts) {   // what a backward over the CENTRED bf16 shadow of y needs (include/vince_hip.h)
                        fin.shadow_consts[c] = scv;
                        fin.shadow_consts[C + c] = fin.beta[c];
                        fin.shadow_consts[2 * C + c] = 0.f;
                        fin.shadow_consts[3 * C + c] = invstd;
                    }
                }
                fin.scale[c] = scv;
                fin.shift[c] = shv;
                if (fin.save_mean) fin.save_mean[c] = mean;
                if (fin.save_invstd) fin.save_invstd[c] = invstd;
                if (fin.running_mean) {
                    const double unbiased = cnt > 1 ? var * cnt / (cnt - 1) : var;
                    fin.running_mean[c] = (1.f - fin.momentum) * fin.running_mean[c] + fin.momentum * mean;
                    fin.running_var[c] = (1.f - fin.momentum) * fin.running_var[c] + fin.momentum * (float)unbiased;
                }
                if (c == 0 && fin.num_batches_tracked) *fin.num_batches_tracked += 1;
            }
        }
        __syncthreads();
    }
    const bool want_sum = fin.out_sum != nullptr;   // uniform: per-channel sums of the stored values (vince_bn_gram_finalize)
    const bool active = col < w.cpr;
    if (!active && !want_sum) return;
    float sc[CH], sh[CH], isc[CH], ish[CH], osum[CH], mu[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) {
        const int cl = (threadIdx.x % w.tpc) * CH + e;
        sc[e] = !active ? 0.f : (fin.stats ? cst[0][cl] : scale[col * CH + e]);
        sh[e] = !active ? 0.f : (fin.stats ? cst[1][cl] : shift[col * CH + e]);
        if constexpr (sizeof(T) == 4) mu[e] = (active && fin.stats) ? cst[2][cl] : 0.f; else mu[e] = 0.f;
        isc[e] = (ids && active) ? ids[col * CH + e] : 1.f;
        ish[e] = (ids && active) ? idt[col * CH + e] : 0.f;
        osum[e] = 0.f;
    }
    const int64_t r0 = (int64_t)blockIdx.y * w.rows_per_block + threadIdx.x / w.tpc;
    const int64_t r1 = min(rows, (int64_t)(blockIdx.y + 1) * w.rows_per_block);
    // 4 rows per trip with all loads issued first: more bytes in flight per wave (the kernel is a pure HBM stream)
    constexpr int U = VINCE_BN_U;
    for (int64_t rb = r0; active && rb < r1; rb += (int64_t)U * w.rpp) {
        uint4 yv[U], iv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = rb + (int64_t)u * w.rpp;
            if (r < r1) {
                const size_t off = (size_t)r * C + (size_t)col * CH;
                yv[u] = ld16<(NT & 1) != 0>(y + off);
                if (idn) iv[u] = ld16<(NT & 1) != 0>(idn + off);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = rb + (int64_t)u * w.rpp;
            if (r >= r1) break;
            const size_t off = (size_t)r * C + (size_t)col * CH;
            float f[CH];
            Chunk<T>::unpack(yv[u], f);
            if constexpr (sizeof(T) == 4) {
                if (fin.y_centred_bf16)    // (uniform) y - mean, rounded to bfloat16: the backward's view of this BatchNorm's input
                    *(uint2*)((bf16_t*)fin.y_centred_bf16 + off) = make_uint2(pack_bf16x2(f[0] - mu[0], f[1] - mu[1]), pack_bf16x2(f[2] - mu[2], f[3] - mu[3]));
            }
#pragma unroll
            for (int e = 0; e < CH; ++e) f[e] = f[e] * sc[e] + sh[e];
            if (idn) {
                float g[CH];
                Chunk<T>::unpack(iv[u], g);
#pragma unroll
                for (int e = 0; e < CH; ++e) f[e] += g[e] * isc[e] + ish[e];
            }
            if (mask_out) {
                uint32_t b = 0;
#pragma unroll
                for (int e = 0; e < CH; ++e) b |= (f[e] > 0.f ? 1u : 0u) << e;
                mask_out[(size_t)r * w.cpr + col] = (uint8_t)b;
            }
            if constexpr (sizeof(T) == 4) {
                if (fin.mask_bf16) {   // (uniform) the same bits in the bf16 tensors' format: one byte per EIGHT channels -- two neighbouring chunks
                    uint32_t b = 0;
#pragma unroll
                    for (int e = 0; e < CH; ++e) b |= (f[e] > 0.f ? 1u : 0u) << e;
                    const uint32_t hi = __shfl_xor(b, 1, 64);        // (lanes col, col ^ 1 walk the same row: tpc is even, C % 8 == 0)
                    if ((col & 1) == 0) fin.mask_bf16[(size_t)r * (w.cpr >> 1) + (col >> 1)] = (uint8_t)(b | (hi << 4));
                }
            }
            if (relu) {
#pragma unroll
                for (int e = 0; e < CH; ++e) f[e] = fmaxf(f[e], 0.f);
            }
            uint4 pv = Chunk<T>::pack(f);
            if constexpr (sizeof(T) == 4) {
                if (fin.out_half_pairs) {
                    // (uniform) stored IEEE-half pairs: this lane's four floats become (hi, lo) dwords; the four lanes of a 16-channel group
                    // then trade them so that lane q stores chunk q of [hi e0-3,e8-11][hi e4-7,e12-15][lo e0-3,e8-11][lo e4-7,e12-15]
                    uint32_t h0, h1, l0, l1;
                    x3_split_pair<x3h_t, false>(f[0], f[1], h0, l0);
                    x3_split_pair<x3h_t, false>(f[2], f[3], h1, l1);
                    // quad_perm [0,1,0,1] = 0x44: lane q reads lane (q & 1); [2,3,2,3] = 0xEE: lane (q & 1) + 2
                    const uint32_t ha0 = __builtin_amdgcn_update_dpp(0, h0, 0x44, 0xf, 0xf, false), ha1 = __builtin_amdgcn_update_dpp(0, h1, 0x44, 0xf, 0xf, false);
                    const uint32_t hb0 = __builtin_amdgcn_update_dpp(0, h0, 0xEE, 0xf, 0xf, false), hb1 = __builtin_amdgcn_update_dpp(0, h1, 0xEE, 0xf, 0xf, false);
                    const uint32_t la0 = __builtin_amdgcn_update_dpp(0, l0, 0x44, 0xf, 0xf, false), la1 = __builtin_amdgcn_update_dpp(0, l1, 0x44, 0xf, 0xf, false);
                    const uint32_t lb0 = __builtin_amdgcn_update_dpp(0, l0, 0xEE, 0xf, 0xf, false), lb1 = __builtin_amdgcn_update_dpp(0, l1, 0xEE, 0xf, 0xf, false);
                    const bool lo_chunk = (col & 2) != 0;
                    pv = lo_chunk ? make_uint4(la0, la1, lb0, lb1) : make_uint4(ha0, ha1, hb0, hb1);
                }
            }
            st16<(NT & 2) != 0>(out + off, pv);
            if constexpr (sizeof(T) == 4) {
                if (fin.out_bf16)      // (uniform) bfloat16 shadow of what was just stored
                    *(uint2*)((bf16_t*)fin.out_bf16 + off) = make_uint2(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]));
            }
            if (want_sum) {   // sum what a reader of `out` will see (the rounded values)
                float q[CH];
                Chunk<T>::unpack(Chunk<T>::pack(f), q);
#pragma unroll
                for (int e = 0; e < CH; ++e) osum[e] += q[e];
            }
        }
    }
    if (want_sum) {
        // threads that share a chunk column (same threadIdx.x % tpc) fold through LDS; one fp64 atomic per channel per workgroup
        __syncthreads();                                  // (the constants in cst[] were copied to registers above)
        float* red = &cst[0][0];                          // [256][CH] floats fit: cst is [2][256 * CH]
#pragma unroll
        for (int e = 0; e < CH; ++e) red[threadIdx.x * CH + e] = osum[e];
        __syncthreads();
        const int nch = w.tpc * CH;
        for (int cl = threadIdx.x; cl < nch; cl += 256) {
            const int tc = cl / CH, e = cl % CH, c = (blockIdx.x * w.tpc + tc) * CH + e;
            if (c >= C) continue;
            float sacc = 0.f;
            for (int r = 0; r < w.rpp; ++r) sacc += red[(r * w.tpc + tc) * CH + e];
            const int rep = fin.out_sum_replicas > 0 ? (int)(blockIdx.y % (unsigned)fin.out_sum_replicas) : 0;
            unsafeAtomicAdd(fin.out_sum + (size_t)rep * C + c, (double)sacc);
        }
    }
}

// BatchNorm constants of y = W a from the Gram matrix of a (include/vince_hip.h: vince_bn_gram_finalize):
//   mean_y[c] = w_c . mean_a,   E[y^2][c] = w_c^T (G / count) w_c,   var = E[y^2] - mean_y^2   (fp64 from the fp32 Gram sums).
// One wavefront per output channel, GF_NC = 4 channels per workgroup (Co / 4 workgroups: enough of them to matter on 256 CUs).
// The workgroup first pulls G into LDS with every load in flight at once (K <= GF_LDS_K: 64 KB; longer reductions read G from
// L2), then lane l owns the columns k2 = l, l + 64, ..., walks the rows k and keeps p[k2] = sum_k w[k] G[k][k2]; the
// contraction with w[k2] and the fold over lanes happen once at the end.
constexpr int GF_NC = 4, GF_LDS_K = 128;
template <typename T>
__global__ __launch_bounds__(256) void bn_gram_finalize_kernel(const float* __restrict__ gram, const double* __restrict__ colsum,
                                                               int R, double count, const T* __restrict__ W, int K, int Co,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float* running_mean, float* running_var, int64_t* nbt,
                                                               float momentum, float eps, float* scale, float* shift,
                                                               float* save_mean, float* save_invstd) {
    __shared__ double mean_a[512];
    __shared__ float wl[GF_NC][512];
    __shared__ __attribute__((aligned(16))) float gs[GF_LDS_K * GF_LDS_K];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool staged = K <= GF_LDS_K;
    if (staged) {
        const int n4 = K * K / 4;                                  // K is a multiple of 4
        for (int i = tid; i < n4; i += 256) ((float4*)gs)[i] = ((const float4*)gram)[i];
    }
    for (int k = tid; k < K; k += 256) {
        double sk = 0;
        for (int r = 0; r < R; ++r) sk += colsum[(size_t)r * K + k];
        mean_a[k] = sk / count;
    }
    for (int i = tid; i < GF_NC * K; i += 256) {
        const int cc = blockIdx.x * GF_NC + i / K, k = i % K;
        float v = 0.f;
        if (cc < Co) {
            if constexpr (sizeof(T) == 4) v = ((const float*)W)[(size_t)cc * K + k];
            else v = bf16_to_f32(((const bf16_t*)W)[(size_t)cc * K + k]);
        }
        wl[i / K][k] = v;
    }
    __syncthreads();
    const float* __restrict__ G = staged ? gs : gram;
    const float* wrow = wl[wave];
    double q = 0, mu = 0;
    for (int k2 = lane; k2 < K; k2 += 64) {
        double p0 = 0, p1 = 0;                                      // two chains: the fp64 adds are dependent
        for (int k = 0; k < K; k += 4) {
            p0 += (double)wrow[k] * (double)G[(size_t)k * K + k2];
            p1 += (double)wrow[k + 1] * (double)G[(size_t)(k + 1) * K + k2];
            p0 += (double)wrow[k + 2] * (double)G[(size_t)(k + 2) * K + k2];
            p1 += (double)wrow[k + 3] * (double)G[(size_t)(k + 3) * K + k2];
        }
        const double w2 = (double)wrow[k2];
        q += (p0 + p1) * w2;
        mu += w2 * mean_a[k2];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        q += __shfl_xor(q, o, 64);
        mu += __shfl_xor(mu, o, 64);
    }
    const int c = blockIdx.x * GF_NC + wave;
    if (lane == 0 && c < Co) {
        double var = q / count - mu * mu;
        if (var < 0) var = 0;
        const float mean = (float)mu;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float scv = gamma[c] * invstd;
        scale[c] = scv;
        shift[c] = beta[c] - mean * scv;
        if (save_mean) save_mean[c] = mean;
        if (save_invstd) save_invstd[c] = invstd;
        if (running_mean) {
            const double unbiased = count > 1 ? var * count / (count - 1) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
        if (c == 0 && nbt) *nbt += 1;
    }
}

// The same for K = 64 / 128 / 256 (round 5; the bottleneck shapes), arranged so that G crosses L2 -> CU once per WORKGROUP instead of
// once per wavefront and nothing is staged: thread t owns column k2 = t % K of G and the K / 4 rows of its part (t / K), walks them
// with coalesced row reads and keeps p[ch] = sum_k w[ch][k] G[k][k2] for the workgroup's GF2_NC channels in fp64; the contraction with
// w[ch][k2] and the fold over the workgroup (wavefront shuffles, then four partials through LDS) happen once at the end.  At K = 256
// (layer3: Co = 1024) the kernel above read the 256 KB matrix from L2 once per output channel and took 93 us; this one 256 times.
constexpr int GF2_NC = 4;
constexpr int gf2_threads(int K) { return K == 256 ? 1024 : K == 128 ? 512 : 256; }   // 4 row parts per column: a chain of K / 4 dependent-latency loads per thread
template <typename T, int K>
__global__ __launch_bounds__(gf2_threads(K)) void bn_gram_finalize2_kernel(const float* __restrict__ gram, const double* __restrict__ colsum,
                                                                int R, double count, const T* __restrict__ W, int Co,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* running_mean, float* running_var, int64_t* nbt,
                                                                float momentum, float eps, float* scale, float* shift,
                                                                float* save_mean, float* save_invstd) {
    constexpr int NT = gf2_threads(K), PARTS = NT / K, ROWS = K / PARTS, NWV = NT / 64;
    __shared__ double mean_a[K];
    __shared__ float wl[GF2_NC][K];
    __shared__ double part[NWV][GF2_NC][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k2 = tid % K, k0 = (tid / K) * ROWS;
    for (int k = tid; k < K; k += NT) {
        double sk = 0;
        for (int r = 0; r < R; ++r) sk += colsum[(size_t)r * K + k];
        mean_a[k] = sk / count;
    }
    for (int i = tid; i < GF2_NC * K; i += NT) {
        const int cc = blockIdx.x * GF2_NC + i / K, k = i % K;
        float v = 0.f;
        if (cc < Co) {
            if constexpr (sizeof(T) == 4) v = ((const float*)W)[(size_t)cc * K + k];
            else v = bf16_to_f32(((const bf16_t*)W)[(size_t)cc * K + k]);
        }
        wl[i / K][k] = v;
    }
    __syncthreads();
    double p[GF2_NC];
#pragma unroll
    for (int c = 0; c < GF2_NC; ++c) p[c] = 0;
    const float* __restrict__ gcol = gram + (size_t)k0 * K + k2;
#pragma unroll 16
    for (int k = 0; k < ROWS; ++k) {
        const double g = (double)gcol[(size_t)k * K];
#pragma unroll
        for (int c = 0; c < GF2_NC; ++c) p[c] += (double)wl[c][k0 + k] * g;
    }
    double q[GF2_NC], mu[GF2_NC];
#pragma unroll
    for (int c = 0; c < GF2_NC; ++c) {
        const double w2 = (double)wl[c][k2];
        q[c] = p[c] * w2;
        mu[c] = k0 == 0 ? w2 * mean_a[k2] : 0.0;      // (one part contributes the mean's dot product)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            q[c] += __shfl_xor(q[c], o, 64);
            mu[c] += __shfl_xor(mu[c], o, 64);
        }
        if (lane == 0) { part[wave][c][0] = q[c]; part[wave][c][1] = mu[c]; }
    }
    __syncthreads();
    if (tid < GF2_NC) {
        const int c = blockIdx.x * GF2_NC + tid;
        if (c < Co) {
            double qq = 0, mm = 0;
            for (int w = 0; w < NWV; ++w) { qq += part[w][tid][0]; mm += part[w][tid][1]; }
            double var = qq / count - mm * mm;
            if (var < 0) var = 0;
            const float mean = (float)mm;
            const float invstd = (float)(1.0 / sqrt(var + (double)eps));
            const float scv = gamma[c] * invstd;
            scale[c] = scv;
            shift[c] = beta[c] - mean * scv;
            if (save_mean) save_mean[c] = mean;
            if (save_invstd) save_invstd[c] = invstd;
            if (running_mean) {
                const double unbiased = count > 1 ? var * count / (count - 1) : var;
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
            }
            if (c == 0 && nbt) *nbt += 1;
        }
    }
}

template <typename T, int NT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dz, const MaskArgs msk,
                                                            const T* __restrict__ y, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, double* sums,
                                                            int64_t rows, int C, RowWalk w, int replicas) {
    constexpr int CH = Elem<T>::CH;
    __shared__ float red[256 * 2 * CH];
    const int tc = threadIdx.x % w.tpc, tr = threadIdx.x / w.tpc;
    const int col = blockIdx.x * w.tpc + tc;
    const bool cv = col < w.cpr;
    float mu[CH], is[CH], sg[CH], sgx[CH], msc[CH], msh[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) {
        mu[e] = cv ? mean[col * CH + e] : 0.f;
        is[e] = cv ? invstd[col * CH + e] : 0.f;
        msc[e] = (cv && msk.scale) ? msk.scale[col * CH + e] : 0.f;
        msh[e] = (cv && msk.scale) ? msk.shift[col * CH + e] : 0.f;
        sg[e] = sgx[e] = 0.f;
    }
    const int64_t r0 = (int64_t)blockIdx.y * w.rows_per_block + tr;
    const int64_t r1 = min(rows, (int64_t)(blockIdx.y + 1) * w.rows_per_block);
    if (cv) {
        constexpr int U = VINCE_BN_U;   // rows per trip, all loads issued before the arithmetic (more bytes in flight)
        for (int64_t rb = r0; rb < r1; rb += (int64_t)U * w.rpp) {
            uint4 dv[U], yv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t r = rb + (int64_t)u * w.rpp;
                if (r < r1) {
                    const size_t off = (size_t)r * C + (size_t)col * CH;
                    dv[u] = ld16<NT != 0>(dz + off);
                    yv[u] = ld16<NT != 0>(y + off);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t r = rb + (int64_t)u * w.rpp;
                if (r >= r1) break;
                const size_t off = (size_t)r * C + (size_t)col * CH;
                float g[CH], yy[CH];
                Chunk<T>::unpack(dv[u], g);
                Chunk<T>::unpack(yv[u], yy);
                apply_mask<T, CH>(msk, off, (size_t)r * w.cpr + col, yy, msc, msh, g);
#pragma unroll
                for (int e = 0; e < CH; ++e) {
                    sg[e] += g[e];
                    sgx[e] += g[e] * (yy[e] - mu[e]) * is[e];
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < CH; ++e) {
        red[threadIdx.x * 2 * CH + e] = sg[e];
        red[threadIdx.x * 2 * CH + CH + e] = sgx[e];
    }
    __syncthreads();
    // thread (tc, which-slot) sums over the rpp row-threads of its column
    for (int t = threadIdx.x; t < w.tpc * 2 * CH; t += 256) {
        const int c_ = t / (2 * CH), slot = t % (2 * CH);
        const int gcol = blockIdx.x * w.tpc + c_;
        if (gcol >= w.cpr) continue;
        float s = 0.f;
        for (int rr = 0; rr < w.rpp; ++rr) s += red[(rr * w.tpc + c_) * 2 * CH + slot];
        const int ch = gcol * CH + (slot % CH), which = slot / CH;
        unsafeAtomicAdd(sums + ((size_t)(blockIdx.y % replicas) * C + ch) * 2 + which, (double)s);
    }
}

// Folds the R replica sums into replica 0 and emits dgamma / dbeta (one thread per channel).
__global__ void bn_bwd_fold_kernel(double* sums, int C, float* dgamma, float* dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double sg = 0, sgx = 0;
    for (int r = 0; r < VINCE_STATS_REPLICAS; ++r) {
        sg += sums[((size_t)r * C + c) * 2];
        sgx += sums[((size_t)r * C + c) * 2 + 1];
    }
    sums[2 * c] = sg;
    sums[2 * c + 1] = sgx;
    if (dgamma) dgamma[c] += (float)sgx;
    if (dbeta) dbeta[c] += (float)sg;
}

#ifndef BWD_APPLY_ROWS
#define BWD_APPLY_ROWS 2
#endif
template <typename T, bool R2, int NT>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dz, const MaskArgs msk,
                                                           const T* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const double* __restrict__ sums, double inv_count,
                                                           T* __restrict__ dy, T* __restrict__ gout, float* dgamma,
                                                           float* dbeta, int64_t rows, int C, RowWalk w, int replicas,
                                                           const vince_bn_reduce2 r2) {
    constexpr int CH = Elem<T>::CH;
    // The replica fold of (sum g, sum g*xhat) is done here, per workgroup for ITS channels, instead of a separate launch
    // per BatchNorm; row-block 0 adds dgamma / dbeta.
    __shared__ float fold[2][256 * CH];
    {
        const int nch = w.tpc * CH, c_base = blockIdx.x * nch;
        for (int cl = threadIdx.x; cl < nch; cl += 256) {
            const int c = c_base + cl;
            if (c >= C) continue;
            double sg, sgx;
            fold_replicas(sums, replicas, C, c, sg, sgx);
            fold[0][cl] = (float)(sg * inv_count);
            fold[1][cl] = (float)(sgx * inv_count);
            if (blockIdx.y == 0) {
                if (dgamma) dgamma[c] += (float)sgx;
                if (dbeta) dbeta[c] += (float)sg;
            }
        }
        __syncthreads();
    }
    const int col = blockIdx.x * w.tpc + (threadIdx.x % w.tpc);
    const bool cv = col < w.cpr;
    const T* __restrict__ y2 = R2 ? (const T*)r2.y : nullptr;
    if (!cv && !y2) return;
    // dx = k1 * (g - ma - (y - mu) * is * mb) as ca * g + cb * y + cc: three constants per channel in registers instead of
    // five (register count decides how many workgroups stay resident)
    float ca[CH], cb[CH], cc[CH], msc[CH], msh[CH];
    // second reduction (r2): the downsample BatchNorm of the same block consumes the SAME masked gradient g -- its
    // (sum g, sum g*xhat) is accumulated here, on the g this pass has in registers, instead of a separate pass over dz
    float mu2[CH], is2[CH], sg2[CH], sgx2[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) {
        const int c = col * CH + e, cl = (threadIdx.x % w.tpc) * CH + e;
        msc[e] = (cv && msk.scale) ? msk.scale[c] : 0.f;
        msh[e] = (cv && msk.scale) ? msk.shift[c] : 0.f;
        {
            const float mu = cv ? mean[c] : 0.f, is = cv ? invstd[c] : 0.f;
            const float k1 = cv ? gamma[c] * is : 0.f;
            ca[e] = k1;
            cb[e] = -k1 * is * fold[1][cl];
            cc[e] = -k1 * fold[0][cl] - cb[e] * mu;
        }
        mu2[e] = (cv && y2) ? r2.mean[c] : 0.f;
        is2[e] = (cv && y2) ? r2.invstd[c] : 0.f;
        sg2[e] = sgx2[e] = 0.f;
    }
    const int64_t r0 = (int64_t)blockIdx.y * w.rows_per_block + threadIdx.x / w.tpc;
    const int64_t r1 = min(rows, (int64_t)(blockIdx.y + 1) * w.rows_per_block);
    constexpr int U = BWD_APPLY_ROWS;
    if (cv)
        for (int64_t rb = r0; rb < r1; rb += (int64_t)U * w.rpp) {
            uint4 dv[U], yv[U], y2v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t r = rb + (int64_t)u * w.rpp;
                if (r < r1) {
                    const size_t off = (size_t)r * C + (size_t)col * CH;
                    dv[u] = ld16<(NT & 1) != 0>(dz + off);
                    yv[u] = ld16<(NT & 1) != 0>(y + off);
                    if constexpr (R2) { if (y2) y2v[u] = *(const uint4*)(y2 + off); }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t r = rb + (int64_t)u * w.rpp;
                if (r >= r1) break;
                const size_t off = (size_t)r * C + (size_t)col * CH;
                float g[CH], yy[CH];
                Chunk<T>::unpack(dv[u], g);
                Chunk<T>::unpack(yv[u], yy);
                apply_mask<T, CH>(msk, off, (size_t)r * w.cpr + col, yy, msc, msh, g);
                if (gout) *(uint4*)(gout + off) = Chunk<T>::pack(g);
                float o[CH];
#pragma unroll
                for (int e = 0; e < CH; ++e) o[e] = ca[e] * g[e] + (cb[e] * yy[e] + cc[e]);
                st16<(NT & 2) != 0>(dy + off, Chunk<T>::pack(o));
                if constexpr (R2) if (y2) {
                    float y2f[CH];
                    Chunk<T>::unpack(y2v[u], y2f);
#pragma unroll
                    for (int e = 0; e < CH; ++e) {
                        sg2[e] += g[e];
                        sgx2[e] += g[e] * (y2f[e] - mu2[e]) * is2[e];
                    }
                }
            }
        }
    if constexpr (R2) if (y2) {   // uniform: registers -> LDS -> one fp64 atomic per channel per workgroup (as bn_bwd_reduce_kernel)
        __shared__ float red2[256 * 2 * CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            red2[threadIdx.x * 2 * CH + e] = sg2[e];
            red2[threadIdx.x * 2 * CH + CH + e] = sgx2[e];
        }
        __syncthreads();
        for (int t = threadIdx.x; t < w.tpc * 2 * CH; t += 256) {
            const int c_ = t / (2 * CH), slot = t % (2 * CH);
            const int gcol = blockIdx.x * w.tpc + c_;
            if (gcol >= w.cpr) continue;
            float s_ = 0.f;
            for (int rr = 0; rr < w.rpp; ++rr) s_ += red2[(rr * w.tpc + c_) * 2 * CH + slot];
            const int ch = gcol * CH + (slot % CH), which = slot / CH;
            unsafeAtomicAdd(r2.sums + ((size_t)(blockIdx.y % r2.replicas) * C + ch) * 2 + which, (double)s_);
        }
    }
}

// ---- stem max-pool 3x3/s2/p1 over relu(bn(y)) -------------------------------------------------------------
// One thread = one 16-byte channel chunk of TWO horizontally adjacent outputs (wo = 2 j, 2 j + 1): their 3 x 3 windows share a column, so
// the pair reads 3 rows x 5 pixels = 15 chunks instead of 18 (round 5: 163 -> 151 us at 256 x 112 x 112 x 64 bf16; the kernel is bound by
// its load instructions -- the one-output-per-thread form requests every input pixel 2.25 times, this one 1.9 times; a 2 x 2 block per
// thread, 1.56 times, holds four running maxima + argmax sets in registers and measured 165 us).  Ties keep the first maximum in
// (kh, kw) order per window, as before.
template <typename T>
__global__ __launch_bounds__(256) void stem_pool_fwd_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, T* __restrict__ out,
                                                            uint8_t* __restrict__ amax, int N, int H, int W, int C,
                                                            int Ho, int Wo) {
    constexpr int CH = Elem<T>::CH;
    const int cpr = C / CH, Wp = (Wo + 1) / 2;
    const int64_t total = (int64_t)N * Ho * Wp * cpr;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int col = (int)(idx % cpr);
        int64_t pix = idx / cpr;
        const int wp = (int)(pix % Wp);
        pix /= Wp;
        const int ho = (int)(pix % Ho);
        const int n = (int)(pix / Ho);
        const int wo0 = 2 * wp;
        const bool two = wo0 + 1 < Wo;
        float sc[CH], sh[CH], best[2][CH];
        int bi[2][CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            sc[e] = scale[col * CH + e];
            sh[e] = shift[col * CH + e];
            best[0][e] = best[1][e] = -INFINITY;
            bi[0][e] = bi[1][e] = 255;
        }
        uint4 v[3][5];
        bool ok[3][5];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {                  // all fifteen requests before any arithmetic
            const int h = ho * 2 - 1 + kh;
            const bool hin = (unsigned)h < (unsigned)H;
            const T* __restrict__ row = y + ((size_t)n * H + (hin ? h : 0)) * W * C + (size_t)col * CH;
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const int w_ = wo0 * 2 - 1 + c;
                ok[kh][c] = hin && (unsigned)w_ < (unsigned)W && (c < 3 || two);
                if (ok[kh][c]) v[kh][c] = *(const uint4*)(row + (size_t)w_ * C);
            }
        }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                if (!ok[kh][c]) continue;
                float f[CH];
                Chunk<T>::unpack(v[kh][c], f);
#pragma unroll
                for (int e = 0; e < CH; ++e) {
                    const float val = fmaxf(f[e] * sc[e] + sh[e], 0.f);
                    if (c < 3 && val > best[0][e]) { best[0][e] = val; bi[0][e] = kh * 3 + c; }
                    if (c >= 2 && val > best[1][e]) { best[1][e] = val; bi[1][e] = kh * 3 + c - 2; }
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            if (o == 1 && !two) break;
            const size_t ooff = (((size_t)n * Ho + ho) * Wo + (wo0 + o)) * C + (size_t)col * CH;
            *(uint4*)(out + ooff) = Chunk<T>::pack(best[o]);
            // a window whose max is 0 passes no gradient (ReLU backward kills it): mark with 255
            uint32_t packed[CH / 4];
#pragma unroll
            for (int q = 0; q < CH / 4; ++q) {
                packed[q] = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    packed[q] |= (uint32_t)(best[o][4 * q + e] > 0.f ? bi[o][4 * q + e] : 255) << (8 * e);
            }
            if constexpr (CH == 8) *(uint2*)(amax + ooff) = make_uint2(packed[0], packed[1]);
            else *(uint32_t*)(amax + ooff) = packed[0];
        }
    }
}

// Gradient of the stem max-pool for the 2x2 block of pre-pool pixels (2a..2a+1, 2b..2b+1), channel chunk `col`: each
// pixel receives dpool from the windows whose argmax it is.  The block touches exactly the 4 windows (a..a+1, b..b+1)
// -- one window load per output pixel instead of the 2.25 a per-pixel gather needs.  acc[2*dh + dw] = pixel (2a+dh, 2b+dw).
template <typename T>
__device__ __forceinline__ void stem_pool_gather2x2(const T* __restrict__ dpool, const uint8_t* __restrict__ amax, int n,
                                                    int a, int b, int col, int C, int Ho, int Wo,
                                                    float (&acc)[4][Elem<T>::CH]) {
    constexpr int CH = Elem<T>::CH;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int e = 0; e < CH; ++e) acc[p][e] = 0.f;
    // window (a+i, b+j) holds pixel (2a+dh, 2b+dw) at (kh, kw) = (dh + 1 - 2i, dw + 1 - 2j), valid when both are in 0..2
    // (round 5: the four windows' gradient chunks and argmax codes are all requested before any of them is used -- with the loads
    // inside the window loop each waited for the previous window's compares)
    uint4 dv[4];
    uint2 mv[4];
    bool live[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = q >> 1, j = q & 1;
        live[q] = a + i < Ho && b + j < Wo;
        if (live[q]) {
            const size_t ooff = (((size_t)n * Ho + (a + i)) * Wo + (b + j)) * C + (size_t)col * CH;
            dv[q] = *(const uint4*)(dpool + ooff);
            if constexpr (CH == 8) mv[q] = *(const uint2*)(amax + ooff);
            else mv[q] = make_uint2(*(const uint32_t*)(amax + ooff), 0u);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (!live[q]) continue;
        const int i = q >> 1, j = q & 1;
        float d[CH];
        Chunk<T>::unpack(dv[q], d);
        uint32_t codes[CH];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            codes[e] = (mv[q].x >> (8 * e)) & 0xff;
            if constexpr (CH == 8) codes[4 + e] = (mv[q].y >> (8 * e)) & 0xff;
        }
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) {
            const int kh = dh + 1 - 2 * i;
            if (kh < 0 || kh > 2) continue;
#pragma unroll
            for (int dw = 0; dw < 2; ++dw) {
                const int kw = dw + 1 - 2 * j;
                if (kw < 0 || kw > 2) continue;
                const uint32_t code = (uint32_t)(kh * 3 + kw);
#pragma unroll
                for (int e = 0; e < CH; ++e)
                    if (codes[e] == code) acc[dh * 2 + dw][e] += d[e];
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void stem_pool_bwd_kernel(const T* __restrict__ dpool,
                                                            const uint8_t* __restrict__ amax, T* __restrict__ g,
                                                            int N, int H, int W, int C, int Ho, int Wo) {
    constexpr int CH = Elem<T>::CH;
    const int cpr = C / CH, H2 = (H + 1) / 2, W2 = (W + 1) / 2;
    const int64_t total = (int64_t)N * H2 * W2 * cpr;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int col = (int)(idx % cpr);
        int64_t blk = idx / cpr;
        const int b = (int)(blk % W2);
        blk /= W2;
        const int a = (int)(blk % H2);
        const int n = (int)(blk / H2);
        float acc[4][CH];
        stem_pool_gather2x2<T>(dpool, amax, n, a, b, col, C, Ho, Wo, acc);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int h = 2 * a + (p >> 1), w_ = 2 * b + (p & 1);
            if (h < H && w_ < W) *(uint4*)(g + (((size_t)n * H + h) * W + w_) * C + (size_t)col * CH) = Chunk<T>::pack(acc[p]);
        }
    }
}

// Stem backward without the materialised pre-pool gradient: max-pool backward (gather) fused into the BatchNorm backward.
// APPLY = false: (sum g, sum g*xhat) into sums[R][C][2]; APPLY = true: dy = gamma*invstd*(g - mean_g - xhat*mean_gx).
// g is rounded to T first, so the results equal stem_pool_bwd -> bn_bwd_reduce / bn_bwd_apply on the stored tensor.
// Requires 256 % (C / CH) == 0 (a thread keeps its channel chunk across the grid-stride loop).
template <typename T, bool APPLY>
__global__ __launch_bounds__(256) void stem_bwd_kernel(const T* __restrict__ dpool, const uint8_t* __restrict__ amax,
                                                       const T* __restrict__ y, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       double* __restrict__ sums, double inv_count, T* __restrict__ dy, int N,
                                                       int H, int W, int C, int Ho, int Wo) {
    constexpr int CH = Elem<T>::CH;
    __shared__ float red[256 * 2 * CH];
    const int cpr = C / CH, H2 = (H + 1) / 2, W2 = (W + 1) / 2;
    const int col = threadIdx.x % cpr;
    float mu[CH], is[CH], k1[CH], ma[CH], mb[CH], sg[CH], sgx[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) {
        const int c = col * CH + e;
        mu[e] = mean[c];
        is[e] = invstd[c];
        sg[e] = sgx[e] = 0.f;
        if constexpr (APPLY) {
            k1[e] = gamma[c] * is[e];
            ma[e] = (float)(sums[2 * c] * inv_count);       // replica 0 holds the folded totals (bn_bwd_fold_kernel)
            mb[e] = (float)(sums[2 * c + 1] * inv_count);
        }
    }
    const int64_t total = (int64_t)N * H2 * W2 * cpr;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int64_t blk = idx / cpr;
        const int b = (int)(blk % W2);
        blk /= W2;
        const int a = (int)(blk % H2);
        const int n = (int)(blk / H2);
        size_t off[4];
        bool ok[4];
        uint4 yv[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int h = 2 * a + (p >> 1), w_ = 2 * b + (p & 1);
            ok[p] = h < H && w_ < W;
            off[p] = (((size_t)n * H + h) * W + w_) * C + (size_t)col * CH;
            if (ok[p]) yv[p] = *(const uint4*)(y + off[p]);
        }
        float acc[4][CH];
        stem_pool_gather2x2<T>(dpool, amax, n, a, b, col, C, Ho, Wo, acc);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (!ok[p]) continue;
            float g[CH], yy[CH];
            Chunk<T>::unpack(Chunk<T>::pack(acc[p]), g);   // the rounding of the tensor the unfused path stores
            Chunk<T>::unpack(yv[p], yy);
            if constexpr (APPLY) {
                float o[CH];
#pragma unroll
                for (int e = 0; e < CH; ++e) o[e] = k1[e] * (g[e] - ma[e] - (yy[e] - mu[e]) * is[e] * mb[e]);
                *(uint4*)(dy + off[p]) = Chunk<T>::pack(o);
            } else {
#pragma unroll
                for (int e = 0; e < CH; ++e) {
                    sg[e] += g[e];
                    sgx[e] += g[e] * (yy[e] - mu[e]) * is[e];
                }
            }
        }
    }
    if constexpr (!APPLY) {
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            red[threadIdx.x * 2 * CH + e] = sg[e];
            red[threadIdx.x * 2 * CH + CH + e] = sgx[e];
        }
        __syncthreads();
        const int rpp = 256 / cpr;
        for (int t = threadIdx.x; t < cpr * 2 * CH; t += 256) {
            const int c_ = t / (2 * CH), slot = t % (2 * CH);
            float s = 0.f;
            for (int rr = 0; rr < rpp; ++rr) s += red[(rr * cpr + c_) * 2 * CH + slot];
            const int ch = c_ * CH + (slot % CH), which = slot / CH;
            unsafeAtomicAdd(sums + ((size_t)(blockIdx.x % VINCE_STATS_REPLICAS) * C + ch) * 2 + which, (double)s);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ x, float* __restrict__ out, int N,
                                                          int HW, int C) {
    constexpr int CH = Elem<T>::CH;
    const int cpr = C / CH;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * cpr) return;
    const int n = idx / cpr, col = idx % cpr;
    float s[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) s[e] = 0.f;
    const T* __restrict__ xp = x + (size_t)n * HW * C + (size_t)col * CH;
    int p = 0;
    for (; p + 7 <= HW; p += 7) {                  // seven pixels' chunks requested before they are added (HW = 49 at 224 x 224)
        uint4 v[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) v[u] = *(const uint4*)(xp + (size_t)(p + u) * C);
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            float f[CH];
            Chunk<T>::unpack(v[u], f);
#pragma unroll
            for (int e = 0; e < CH; ++e) s[e] += f[e];
        }
    }
    for (; p < HW; ++p) {
        float f[CH];
        Chunk<T>::unpack(*(const uint4*)(xp + (size_t)p * C), f);
#pragma unroll
        for (int e = 0; e < CH; ++e) s[e] += f[e];
    }
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int e = 0; e < CH; ++e) out[(size_t)n * C + col * CH + e] = s[e] * inv;
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dout, T* __restrict__ dx, int N,
                                                          int HW, int C) {
    constexpr int CH = Elem<T>::CH;
    const int cpr = C / CH;
    const int64_t total = (int64_t)N * HW * cpr;
    const float inv = 1.f / (float)HW;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int col = (int)(idx % cpr);
        const int64_t pix = idx / cpr;
        const int n = (int)(pix / HW);
        float f[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) f[e] = dout[(size_t)n * C + col * CH + e] * inv;
        *(uint4*)(dx + (size_t)pix * C + (size_t)col * CH) = Chunk<T>::pack(f);
    }
}

// workgroups per element-wise BatchNorm launch (VINCE_BN_BLOCKS: measurement knob)
inline int bn_bwd_target_blocks() {
    // bn_bwd_apply opens with the replica fold + seven per-channel constant vectors: ~3 us of dependent L2 latency per
    // workgroup before the first row moves.  Half as many, twice as long workgroups than the forward apply: -0.4 ms/step
    // (swept 512..3072; the forward apply stays best at 2048).
    static const int n = VINCE_MEASURE_KNOB("bn_bwd_blocks", 1024);
    return n;
}

inline int bn_target_blocks() {
    static const int n = VINCE_MEASURE_KNOB("bn_blocks", 2048);   // swept 1024..16384: 1536-2048 best
    return n;
}

inline int grid_for(int64_t total_threads) {
    int64_t b = (total_threads + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

#define DTYPE_OK(fn) VINCE_CHECK_ARG(dtype == VINCE_F32 || dtype == VINCE_BF16, VINCE_E_DTYPE, fn ": bad dtype %d", dtype)
#define CH_OF(dtype) ((dtype) == VINCE_F32 ? 4 : 8)
#define ESZ_OF(dtype) ((dtype) == VINCE_F32 ? 4 : 2)

extern "C" int vince_bn_finalize(const double* stats, int64_t count, int32_t C, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps,
                                 int train, float* scale, float* shift, float* save_mean, float* save_invstd,
                                 void* stream) {
    VINCE_CHECK_ARG(C > 0 && gamma && beta && scale && shift, VINCE_E_ARG, "vince_bn_finalize: bad arguments");
    VINCE_CHECK_ARG(train ? (stats != nullptr && count > 0) : (running_mean && running_var), VINCE_E_ARG,
                    "vince_bn_finalize: missing statistics");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats,
                       (double)count, C, gamma, beta, running_mean, running_var, nbt, momentum, eps, train, scale, shift,
                       save_mean, save_invstd);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

// the BatchNorm apply pass in the instantiation the row walk's nt policy asks for (plain / nt loads / nt loads + stores)
static void launch_bn_apply(int dtype, dim3 grid, hipStream_t stream, const void* y, const float* scale, const float* shift, const void* identity,
                            const float* id_scale, const float* id_shift, void* out, uint8_t* mask_out, int64_t rows, int32_t C, int relu,
                            const RowWalk& w, const vince_bn_train& fin) {
#define VINCE_BN_APPLY(TT, NN)                                                                                                         \
    hipLaunchKernelGGL((bn_apply_kernel<TT, NN>), grid, dim3(256), 0, stream, (const TT*)y, scale, shift, (const TT*)identity, id_scale, \
                       id_shift, (TT*)out, mask_out, rows, C, relu, w, fin)
    if (dtype == VINCE_F32) { if (w.nt == 3) VINCE_BN_APPLY(float, 3); else if (w.nt & 1) VINCE_BN_APPLY(float, 1); else VINCE_BN_APPLY(float, 0); }
    else { if (w.nt == 3) VINCE_BN_APPLY(bf16_t, 3); else if (w.nt & 1) VINCE_BN_APPLY(bf16_t, 1); else VINCE_BN_APPLY(bf16_t, 0); }
#undef VINCE_BN_APPLY
}

extern "C" int vince_bn_apply(int dtype, const void* y, const float* scale, const float* shift, const void* identity,
                              const float* id_scale, const float* id_shift, void* out, uint8_t* mask_out, int64_t rows,
                              int32_t C, int relu, void* stream) {
    DTYPE_OK("vince_bn_apply");
    VINCE_CHECK_ARG(y && scale && shift && out && rows > 0, VINCE_E_ARG, "vince_bn_apply: bad arguments");
    VINCE_CHECK_ARG(C % CH_OF(dtype) == 0, VINCE_E_SHAPE, "vince_bn_apply: C=%d not a multiple of %d", C, CH_OF(dtype));
    RowWalk w = make_rowwalk(rows, C, CH_OF(dtype), bn_target_blocks());
    dim3 grid(w.colgroups, w.rowblocks);
    VinceProfScope prof(VINCE_TAG_BN_APPLY, (double)rows * C * ESZ_OF(dtype) * (identity ? 3 : 2) + (mask_out ? (double)rows * C / CH_OF(dtype) : 0), stream);
    vince_bn_train fin;
    memset(&fin, 0, sizeof(fin));
    launch_bn_apply(dtype, grid, (hipStream_t)stream, y, scale, shift, identity, id_scale, id_shift, out, mask_out, rows, C, relu, w, fin);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_bn_train_apply(int dtype, const void* y, const vince_bn_train* bt, const void* identity,
                                    const float* id_scale, const float* id_shift, void* out, uint8_t* mask_out, int64_t rows,
                                    int32_t C, int relu, void* stream) {
    DTYPE_OK("vince_bn_train_apply");
    VINCE_CHECK_ARG(y && bt && out && rows > 0, VINCE_E_ARG, "vince_bn_train_apply: bad arguments");
    VINCE_CHECK_ARG(bt->stats && bt->count > 0 && bt->gamma && bt->beta && bt->scale && bt->shift, VINCE_E_ARG,
                    "vince_bn_train_apply: stats, count, gamma, beta, scale and shift are required");
    VINCE_CHECK_ARG(!bt->running_mean == !bt->running_var, VINCE_E_ARG, "vince_bn_train_apply: running_mean and running_var come together");
    VINCE_CHECK_ARG(C % CH_OF(dtype) == 0, VINCE_E_SHAPE, "vince_bn_train_apply: C=%d not a multiple of %d", C, CH_OF(dtype));
    VINCE_CHECK_ARG((!bt->out_bf16 && !bt->mask_bf16 && !bt->y_centred_bf16 && !bt->shadow_consts) ||
                    (dtype == VINCE_F32 && C % 8 == 0 && ((uintptr_t)bt->out_bf16 & 15) == 0 && ((uintptr_t)bt->y_centred_bf16 & 15) == 0), VINCE_E_ARG,
                    "vince_bn_train_apply: the bf16 shadows belong to an fp32 launch with C a multiple of 8");
    VINCE_CHECK_ARG(!bt->y_centred_bf16 == !bt->shadow_consts, VINCE_E_ARG, "vince_bn_train_apply: y_centred_bf16 and shadow_consts come together");
    VINCE_CHECK_ARG(!bt->out_half_pairs || (dtype == VINCE_F32 && C % 16 == 0 && !identity && !bt->out_sum), VINCE_E_ARG,
                    "vince_bn_train_apply: out_half_pairs is for a plain fp32 BatchNorm + ReLU pass with C a multiple of 16");
    vince_bn_train fin = *bt;
    if (fin.replicas <= 0 || fin.replicas > VINCE_STATS_REPLICAS) fin.replicas = VINCE_STATS_REPLICAS;
    RowWalk w = make_rowwalk(rows, C, CH_OF(dtype), bn_target_blocks());
    dim3 grid(w.colgroups, w.rowblocks);
    VinceProfScope prof(VINCE_TAG_BN_APPLY, (double)rows * C * ESZ_OF(dtype) * (identity ? 3 : 2) + (mask_out ? (double)rows * C / CH_OF(dtype) : 0), stream);
    launch_bn_apply(dtype, grid, (hipStream_t)stream, y, nullptr, nullptr, identity, id_scale, id_shift, out, mask_out, rows, C, relu, w, fin);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_bn_gram_finalize(int dtype, const float* gram, const double* colsum, int32_t colsum_replicas, int64_t count,
                                      const void* w, int32_t K, int32_t Co, const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                                      float eps, float* scale, float* shift, float* save_mean, float* save_invstd, void* stream) {
    DTYPE_OK("vince_bn_gram_finalize");
    VINCE_CHECK_ARG(gram && colsum && w && gamma && beta && scale && shift && count > 0 && colsum_replicas > 0, VINCE_E_ARG,
                    "vince_bn_gram_finalize: bad arguments");
    VINCE_CHECK_ARG(K > 0 && K <= 512 && Co > 0 && K % CH_OF(dtype) == 0, VINCE_E_SHAPE,
                    "vince_bn_gram_finalize: K=%d (1..512, multiple of %d), Co=%d", K, CH_OF(dtype), Co);
    VINCE_CHECK_ARG(!running_mean == !running_var, VINCE_E_ARG, "vince_bn_gram_finalize: running_mean and running_var come together");
    if (K == 64 || K == 128 || K == 256) {
        const dim3 grid2((Co + GF2_NC - 1) / GF2_NC);
#define GF2_LAUNCH(TT, KK)                                                                                                             \
        hipLaunchKernelGGL((bn_gram_finalize2_kernel<TT, KK>), grid2, dim3(gf2_threads(KK)), 0, (hipStream_t)stream, gram, colsum, colsum_replicas, \
                           (double)count, (const TT*)w, Co, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, \
                           scale, shift, save_mean, save_invstd)
        if (dtype == VINCE_F32) { if (K == 64) GF2_LAUNCH(float, 64); else if (K == 128) GF2_LAUNCH(float, 128); else GF2_LAUNCH(float, 256); }
        else { if (K == 64) GF2_LAUNCH(bf16_t, 64); else if (K == 128) GF2_LAUNCH(bf16_t, 128); else GF2_LAUNCH(bf16_t, 256); }
#undef GF2_LAUNCH
        VINCE_CHECK_LAUNCH();
        return VINCE_OK;
    }
    const dim3 grid((Co + GF_NC - 1) / GF_NC);
    if (dtype == VINCE_F32)
        hipLaunchKernelGGL(bn_gram_finalize_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, gram, colsum, colsum_replicas,
                           (double)count, (const float*)w, K, Co, gamma, beta, running_mean, running_var, num_batches_tracked,
                           momentum, eps, scale, shift, save_mean, save_invstd);
    else
        hipLaunchKernelGGL(bn_gram_finalize_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, gram, colsum, colsum_replicas,
                           (double)count, (const bf16_t*)w, K, Co, gamma, beta, running_mean, running_var, num_batches_tracked,
                           momentum, eps, scale, shift, save_mean, save_invstd);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_bn_bwd_reduce(int dtype, const void* dz, const void* mask_src, const uint8_t* mask_bits,
                                   const float* mask_scale, const float* mask_shift, const void* y, const float* mean,
                                   const float* invstd, double* sums, int64_t rows, int32_t C, int32_t replicas,
                                   void* stream) {
    const MaskArgs msk{mask_src, mask_bits, mask_scale, mask_shift};
    DTYPE_OK("vince_bn_bwd_reduce");
    if (replicas <= 0 || replicas > VINCE_STATS_REPLICAS) replicas = VINCE_STATS_REPLICAS;
    VINCE_CHECK_ARG(dz && y && mean && invstd && sums && rows > 0, VINCE_E_ARG, "vince_bn_bwd_reduce: bad arguments");
    VINCE_CHECK_ARG(C % CH_OF(dtype) == 0, VINCE_E_SHAPE, "vince_bn_bwd_reduce: C=%d not a multiple of %d", C, CH_OF(dtype));
    // every workgroup ends with 2 C fp64 atomics: on a wide, short tensor (layer4's last block: 12544 x 2048) 1568 workgroups of 8 rows
    // spent their time there (6.4 M atomics, 62 us for 103 MB); row blocks x C is held near 2^19
    int target = (1 << 19) / (C > 0 ? C : 1);
    target = target < 128 ? 128 : target > 2048 ? 2048 : target;
    RowWalk w = make_rowwalk(rows, C, CH_OF(dtype), target);
    dim3 grid(w.colgroups, w.rowblocks);
    VinceProfScope prof(VINCE_TAG_BN_BWD_REDUCE, (double)rows * C * ESZ_OF(dtype) * 2, stream);
    // nt loads here are their own policy bit (4): the apply pass that follows reads dz and y again
    static const bool reduce_nt = (vince_knob("bn_nt", 1) & 4) != 0;
#define VINCE_BWD_REDUCE(TT, NN)                                                                                   \
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<TT, NN>), grid, dim3(256), 0, (hipStream_t)stream, (const TT*)dz, msk, \
                       (const TT*)y, mean, invstd, sums, rows, C, w, replicas)
    static const long long reduce_nt_min = (long long)vince_knob("bn_nt_min_mb", 0) << 20;
    const bool rnt = reduce_nt && (long long)rows * C * ESZ_OF(dtype) >= reduce_nt_min;
    if (dtype == VINCE_F32) { if (rnt) VINCE_BWD_REDUCE(float, 1); else VINCE_BWD_REDUCE(float, 0); }
    else { if (rnt) VINCE_BWD_REDUCE(bf16_t, 1); else VINCE_BWD_REDUCE(bf16_t, 0); }
#undef VINCE_BWD_REDUCE
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_bn_bwd_apply(int dtype, const void* dz, const void* mask_src, const uint8_t* mask_bits,
                                  const float* mask_scale, const float* mask_shift, const void* y, const float* mean,
                                  const float* invstd, const float* gamma, const double* sums, int64_t count, void* dy,
                                  void* g_out, float* dgamma, float* dbeta, int64_t rows, int32_t C, int32_t replicas,
                                  const vince_bn_reduce2* second, void* stream) {
    const MaskArgs msk{mask_src, mask_bits, mask_scale, mask_shift};
    DTYPE_OK("vince_bn_bwd_apply");
    vince_bn_reduce2 r2;
    memset(&r2, 0, sizeof(r2));
    if (second) {
        VINCE_CHECK_ARG(second->y && second->mean && second->invstd && second->sums, VINCE_E_ARG,
                        "vince_bn_bwd_apply: second reduction needs y, mean, invstd and sums");
        r2 = *second;
        if (r2.replicas <= 0 || r2.replicas > VINCE_STATS_REPLICAS) r2.replicas = VINCE_STATS_REPLICAS;
    }
    if (replicas <= 0 || replicas > VINCE_STATS_REPLICAS) replicas = VINCE_STATS_REPLICAS;
    VINCE_CHECK_ARG(dz && y && mean && invstd && gamma && sums && dy && rows > 0 && count > 0, VINCE_E_ARG,
                    "vince_bn_bwd_apply: bad arguments");
    VINCE_CHECK_ARG(C % CH_OF(dtype) == 0, VINCE_E_SHAPE, "vince_bn_bwd_apply: C=%d not a multiple of %d", C, CH_OF(dtype));
    RowWalk w = make_rowwalk(rows, C, CH_OF(dtype), bn_bwd_target_blocks());
    dim3 grid(w.colgroups, w.rowblocks);
    VinceProfScope prof(VINCE_TAG_BN_BWD_APPLY, (double)rows * C * ESZ_OF(dtype) * (3 + (g_out ? 1 : 0) + (r2.y ? 1 : 0)) +
                        (mask_bits ? (double)rows * C / CH_OF(dtype) : 0), stream);
    const double inv_count = 1.0 / (double)count;
#define VINCE_BWD_APPLY_N(TT, RR, NN)                                                                                    \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<TT, RR, NN>), grid, dim3(256), 0, (hipStream_t)stream, (const TT*)dz, msk,       \
                       (const TT*)y, mean, invstd, gamma, sums, inv_count, (TT*)dy, (TT*)g_out, dgamma, dbeta, rows, C, w, \
                       replicas, r2)
#define VINCE_BWD_APPLY(TT, RR)                                                                                          \
    do { if (w.nt == 3) VINCE_BWD_APPLY_N(TT, RR, 3); else if (w.nt & 1) VINCE_BWD_APPLY_N(TT, RR, 1); else VINCE_BWD_APPLY_N(TT, RR, 0); } while (0)
    const bool has_r2 = r2.y != nullptr;
    if (dtype == VINCE_F32) { if (has_r2) VINCE_BWD_APPLY(float, true); else VINCE_BWD_APPLY(float, false); }
    else { if (has_r2) VINCE_BWD_APPLY(bf16_t, true); else VINCE_BWD_APPLY(bf16_t, false); }
#undef VINCE_BWD_APPLY
#undef VINCE_BWD_APPLY_N
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_stem_pool_fwd(int dtype, const void* y, const float* scale, const float* shift, void* out,
                                   uint8_t* argmax, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    DTYPE_OK("vince_stem_pool_fwd");
    VINCE_CHECK_ARG(y && scale && shift && out && argmax && N > 0 && H > 0 && W > 0, VINCE_E_ARG, "vince_stem_pool_fwd: bad arguments");
    VINCE_CHECK_ARG(C % CH_OF(dtype) == 0, VINCE_E_SHAPE, "vince_stem_pool_fwd: C=%d not a multiple of %d", C, CH_OF(dtype));
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)N * Ho * ((Wo + 1) / 2) * (C / CH_OF(dtype));     // one thread per chunk of an output PAIR
    VinceProfScope prof(VINCE_TAG_STEM_POOL, ((double)N * H * W + (double)N * Ho * Wo) * C * ESZ_OF(dtype) + (double)N * Ho * Wo * (C / CH_OF(dtype)), stream);
    if (dtype == VINCE_F32)
        hipLaunchKernelGGL(stem_pool_fwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)y, scale, shift, (float*)out, argmax, N, H, W, C, Ho, Wo);
    else
        hipLaunchKernelGGL(stem_pool_fwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)y, scale, shift, (bf16_t*)out, argmax, N, H, W, C, Ho, Wo);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_stem_pool_bwd(int dtype, const void* dpool, const uint8_t* argmax, void* g, int32_t N, int32_t H,
                                   int32_t W, int32_t C, void* stream) {
    DTYPE_OK("vince_stem_pool_bwd");
    VINCE_CHECK_ARG(dpool && argmax && g && N > 0 && H > 0 && W > 0, VINCE_E_ARG, "vince_stem_pool_bwd: bad arguments");
    VINCE_CHECK_ARG(C % CH_OF(dtype) == 0, VINCE_E_SHAPE, "vince_stem_pool_bwd: C=%d not a multiple of %d", C, CH_OF(dtype));
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / CH_OF(dtype));   // one thread per 2x2 pixel block
    VinceProfScope prof(VINCE_TAG_STEM_BWD, ((double)N * H * W + (double)N * Ho * Wo) * C * ESZ_OF(dtype), stream);
    if (dtype == VINCE_F32)
        hipLaunchKernelGGL(stem_pool_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)dpool, argmax, (float*)g, N, H, W, C, Ho, Wo);
    else
        hipLaunchKernelGGL(stem_pool_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)dpool, argmax, (bf16_t*)g, N, H, W, C, Ho, Wo);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_stem_bwd_reduce(int dtype, const void* dpool, const uint8_t* argmax, const void* y, const float* mean,
                                     const float* invstd, double* sums, int32_t N, int32_t H, int32_t W, int32_t C,
                                     void* stream) {
    DTYPE_OK("vince_stem_bwd_reduce");
    VINCE_CHECK_ARG(dpool && argmax && y && mean && invstd && sums && N > 0 && H > 0 && W > 0, VINCE_E_ARG,
                    "vince_stem_bwd_reduce: bad arguments");
    const int CH = CH_OF(dtype);
    VINCE_CHECK_ARG(C % CH == 0 && 256 % (C / CH) == 0, VINCE_E_SHAPE, "vince_stem_bwd_reduce: C=%d unsupported", C);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / CH);
    VinceProfScope prof(VINCE_TAG_STEM_BWD, ((double)N * H * W + 0.25 * N * H * W) * C * ESZ_OF(dtype), stream);   // reads y and the pooled gradient
    const int grid = (int)std::min<int64_t>((total + 255) / 256, 4096);
    if (dtype == VINCE_F32)
        hipLaunchKernelGGL((stem_bwd_kernel<float, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)dpool,
                           argmax, (const float*)y, mean, invstd, nullptr, sums, 0.0, (float*)nullptr, N, H, W, C, Ho, Wo);
    else
        hipLaunchKernelGGL((stem_bwd_kernel<bf16_t, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dpool,
                           argmax, (const bf16_t*)y, mean, invstd, nullptr, sums, 0.0, (bf16_t*)nullptr, N, H, W, C, Ho, Wo);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_stem_bwd_apply(int dtype, const void* dpool, const uint8_t* argmax, const void* y, const float* mean,
                                    const float* invstd, const float* gamma, double* sums, void* dy, float* dgamma,
                                    float* dbeta, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    DTYPE_OK("vince_stem_bwd_apply");
    VINCE_CHECK_ARG(dpool && argmax && y && mean && invstd && gamma && sums && dy && N > 0 && H > 0 && W > 0, VINCE_E_ARG,
                    "vince_stem_bwd_apply: bad arguments");
    const int CH = CH_OF(dtype);
    VINCE_CHECK_ARG(C % CH == 0 && 256 % (C / CH) == 0, VINCE_E_SHAPE, "vince_stem_bwd_apply: C=%d unsupported", C);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / CH);
    VinceProfScope prof(VINCE_TAG_STEM_BWD, (2.0 * N * H * W + 0.25 * N * H * W) * C * ESZ_OF(dtype), stream);   // + writes dy
    const double inv_count = 1.0 / ((double)N * H * W);
    hipLaunchKernelGGL(bn_bwd_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, C, dgamma, dbeta);
    if (dtype == VINCE_F32)
        hipLaunchKernelGGL((stem_bwd_kernel<float, true>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)dpool, argmax, (const float*)y, mean, invstd, gamma, sums, inv_count, (float*)dy, N, H, W,
                           C, Ho, Wo);
    else
        hipLaunchKernelGGL((stem_bwd_kernel<bf16_t, true>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)dpool, argmax, (const bf16_t*)y, mean, invstd, gamma, sums, inv_count, (bf16_t*)dy, N, H,
                           W, C, Ho, Wo);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_avgpool_fwd(int dtype, const void* x, float* out, int32_t N, int32_t HW, int32_t C, void* stream) {
    DTYPE_OK("vince_avgpool_fwd");
    VINCE_CHECK_ARG(x && out && N > 0 && HW > 0, VINCE_E_ARG, "vince_avgpool_fwd: bad arguments");
    VINCE_CHECK_ARG(C % CH_OF(dtype) == 0, VINCE_E_SHAPE, "vince_avgpool_fwd: C=%d not a multiple of %d", C, CH_OF(dtype));
    const int total = N * (C / CH_OF(dtype));
    if (dtype == VINCE_F32)
        hipLaunchKernelGGL(avgpool_fwd_kernel<float>, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, out, N, HW, C);
    else
        hipLaunchKernelGGL(avgpool_fwd_kernel<bf16_t>, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)x, out, N, HW, C);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_avgpool_bwd(int dtype, const float* dout, void* dx, int32_t N, int32_t HW, int32_t C, void* stream) {
    DTYPE_OK("vince_avgpool_bwd");
    VINCE_CHECK_ARG(dout && dx && N > 0 && HW > 0, VINCE_E_ARG, "vince_avgpool_bwd: bad arguments");
    VINCE_CHECK_ARG(C % CH_OF(dtype) == 0, VINCE_E_SHAPE, "vince_avgpool_bwd: C=%d not a multiple of %d", C, CH_OF(dtype));
    const int64_t total = (int64_t)N * HW * (C / CH_OF(dtype));
    if (dtype == VINCE_F32)
        hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dout,
                           (float*)dx, N, HW, C);
    else
        hipLaunchKernelGGL(avgpool_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dout,
                           (bf16_t*)dx, N, HW, C);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}
