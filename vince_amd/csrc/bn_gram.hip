// Train-mode BatchNorm + ReLU of a narrow tensor (C = 64 / 128, bf16) that ALSO returns the Gram matrix of what it stores.
//
// Reference semantics: a = relu(bn2(y)) of a bottleneck (models/building_blocks/resnet.py:116-121, BatchNorm2d in train mode);
// the Gram matrix G = a^T a (sum over pixels) together with the column sums is what vince_bn_gram_finalize turns into the
// statistics of bn3(conv3(a)) BEFORE conv3 runs (the fused join, conv_xjoin.hip) and what the BatchNorm-backward algebra
// (bn_algebra.hip) contracts the weight gradient with.
//
// Until round 3 the Gram matrix came from a weight-gradient launch with in = dy = a: one more full read of `a` (103 MB in layer1)
// and two more launches (GEMM + slab reduction) on the forward's critical path, per block and encoder.  Here the pass that WRITES
// a keeps each 64-row slice of it in LDS (the [pixel][channel] sub-tile layout of conv_wgrad_tr.hip: 32 rows x 64 channels, halves
// of a row swapped on odd row pairs) and feeds it to the matrix pipe through transposing reads; the kernel stays an HBM stream
// (16 MFMAs per wavefront per 64 rows at C = 128).  Every workgroup owns a contiguous row range and stores its partial matrix into
// a slab; vince_slab_reduce adds the slabs in a fixed order -- no atomics, run-to-run identical like the path it replaces.
#include <string.h>

#include "conv_wgrad.h"

namespace {

constexpr int GS = 64;   // rows per slice

template <int K>
__global__ __launch_bounds__(256) void bn_apply_gram_kernel(const bf16_t* __restrict__ y, bf16_t* __restrict__ out, int64_t rows,
                                                            int64_t rows_per_wg, float* __restrict__ slabs, const vince_bn_train fin) {
    constexpr int CPR = K / 8, RPP = 256 / CPR, NP = GS / RPP;      // 16-byte chunks per row, rows per pass, passes per slice
    constexpr int NSUB = K / 64, BUF = (GS / 32) * NSUB * 4096;     // sub-tiles of 32 rows x 64 channels
    constexpr int NT = K == 64 ? 1 : 4;                             // 32 x 32 tiles of the matrix per wavefront
    __shared__ float cst[2][K];
    __shared__ __attribute__((aligned(16))) unsigned char tile[2][BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // train-mode finalize, as in bn_apply_kernel: every workgroup folds the statistic replicas of the K channels, workgroup 0 publishes
    for (int c = tid; c < K; c += 256) {
        double s1, s2;
        fold_replicas(fin.stats, fin.replicas, K, c, s1, s2);
        const double cnt = (double)fin.count;
        const double m = s1 / cnt;
        double var = s2 / cnt - m * m;
        if (var < 0) var = 0;
        const float mean = (float)m;
        const float invstd = (float)(1.0 / sqrt(var + (double)fin.eps));
        const float scv = fin.gamma[c] * invstd;
        const float shv = fin.beta[c] - mean * scv;
        cst[0][c] = scv;
        cst[1][c] = shv;
        if (blockIdx.x == 0) {
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

    const int chunk = tid % CPR, prow = tid / CPR;
    float sc[8], sh[8], osum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = cst[0][chunk * 8 + e];
        sh[e] = cst[1][chunk * 8 + e];
        osum[e] = 0.f;
    }
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_wg, r_end = min(rows, r_begin + rows_per_wg);
    const int nslices = (int)((r_end - r_begin + GS - 1) / GS);

    // LDS position of this thread's chunk in pass p: row p*RPP + prow of the slice
    uint32_t wr[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int row = p * RPP + prow, r32 = row & 31, c8 = chunk & 7;
        wr[p] = (uint32_t)(((row >> 5) * NSUB + (chunk >> 3)) * 4096 + r32 * 128 + ((((c8 >> 2) ^ ((r32 >> 1) & 1)) << 6) | ((c8 & 3) << 4)));
    }
    // fragment read addresses (conv_wgrad_tr.hip): channel tile t (32 channels) -> sub-tile t >> 1, half t & 1
    const int g = lane >> 4, t16 = lane & 15;
    const int frow = (g >> 1) * 8 + (t16 >> 2);
    const int fsw = (frow >> 1) & 1;
    const uint32_t fcb = (uint32_t)((16 * (g & 1) + (t16 & 3) * 4) * 2);
    const int ti = K == 64 ? (wave >> 1) : wave;                     // this wavefront's row tile of the matrix
    auto faddr = [&](const int t) -> uint32_t { return (uint32_t)((t >> 1) * 4096 + frow * 128 + ((((t & 1) ^ fsw)) << 6)) + fcb; };
    const uint32_t fa_i = faddr(ti);
    uint32_t fa_j[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) fa_j[j] = faddr(K == 64 ? (wave & 1) : j);
    const uint32_t tile_base = (uint32_t)(uintptr_t)(lds_ptr_t)&tile[0][0];
    auto frag = [&](const uint32_t addr) -> bf16x8_t {
        typedef __attribute__((address_space(3))) s16x4_t* lp;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)addr);
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)(addr + 512));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    f32x16_t acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    uint4 cur[NP], nxt[NP];
    auto load_slice = [&](const int s, uint4 (&v)[NP]) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int64_t r = r_begin + (int64_t)s * GS + p * RPP + prow;
            v[p] = r < r_end ? *(const uint4*)(y + (size_t)r * K + chunk * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    if (nslices > 0) load_slice(0, cur);
    for (int s = 0; s < nslices; ++s) {
        if (s + 1 < nslices) load_slice(s + 1, nxt);
        unsigned char* buf = &tile[s & 1][0];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int64_t r = r_begin + (int64_t)s * GS + p * RPP + prow;
            uint4 pv = make_uint4(0, 0, 0, 0);
            if (r < r_end) {
                float f[8];
                Chunk<bf16_t>::unpack(cur[p], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e] * sc[e] + sh[e], 0.f);
                pv = Chunk<bf16_t>::pack(f);
                *(uint4*)(out + (size_t)r * K + chunk * 8) = pv;
                float q[8];
                Chunk<bf16_t>::unpack(pv, q);                       // the sums are those of the STORED (rounded) values
#pragma unroll
                for (int e = 0; e < 8; ++e) osum[e] += q[e];
            }
            *(uint4*)(buf + wr[p]) = pv;                            // rows past the range enter the matrix as zeros
        }
        __syncthreads();     // one barrier per slice: the next slice goes to the other buffer, and this one is overwritten only after
                             // every wavefront has passed the NEXT barrier, i.e. has finished the reads below
        const uint32_t b = tile_base + (uint32_t)(s & 1) * BUF;
#pragma unroll
        for (int h = 0; h < GS / 32; ++h)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint32_t off = b + (uint32_t)(h * NSUB * 4096 + ks * 2048);
                const bf16x8_t a = frag(off + fa_i);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag(off + fa_j[j]), acc[j], 0, 0, 0);
            }
#pragma unroll
        for (int p = 0; p < NP; ++p) cur[p] = nxt[p];
    }

    // the partial matrix into this workgroup's slab ([K][K] floats, row = first channel index)
    float* const slab = slabs + (size_t)blockIdx.x * K * K;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int tj = K == 64 ? (wave & 1) : j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            slab[(size_t)row * K + tj * 32 + (lane & 31)] = acc[j][r];
        }
    }
    // column sums: threads that share a chunk fold through LDS, one fp64 atomic per channel per workgroup (as bn_apply_kernel)
    if (fin.out_sum) {
        __syncthreads();
        float* red = (float*)&tile[0][0];      // 256 x 8 floats = 8 KB
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid * 8 + e] = osum[e];
        __syncthreads();
        for (int c = tid; c < K; c += 256) {
            const int ck = c >> 3, e = c & 7;
            float sacc = 0.f;
            for (int r = 0; r < RPP; ++r) sacc += red[(r * CPR + ck) * 8 + e];
            const int rep = fin.out_sum_replicas > 0 ? (int)(blockIdx.x % (unsigned)fin.out_sum_replicas) : 0;
            unsafeAtomicAdd(fin.out_sum + (size_t)rep * K + c, (double)sacc);
        }
    }
}

}  // namespace

// most workgroups (= slabs): enough of them to keep an HBM stream going through one barrier per slice, few enough that the slabs
// (C * C floats each, written and read once) stay small beside the tensor.  Layer1 / layer2 at the benchmark batch, kernel + slab
// reduction: 512 workgroups 64.0 + 6.3 / 32.8 + 6.3 us, 768 / 640 54.0 + 8.0 / 31.7 + 8.0, 2048 / 1024 56.8 + 11.5 / 40.3 + 11.5
// (the plain pass + the weight-gradient launch + its reduction: 49 + 25 + 6 / 24.5 + 17 + 6).
static int64_t gram_max_wgs(int32_t C) { return C == 64 ? 768 : 640; }

extern "C" size_t vince_bn_train_apply_gram_scratch_bytes(int64_t rows, int32_t C) {
    if ((C != 64 && C != 128) || rows <= 0) return 0;
    int64_t g = (rows + GS - 1) / GS;
    if (g > gram_max_wgs(C)) g = gram_max_wgs(C);
    return (size_t)g * C * C * sizeof(float);
}

extern "C" int vince_bn_train_apply_gram(int dtype, const void* y, const vince_bn_train* bt, void* out, int64_t rows, int32_t C,
                                         float* gram, void* scratch, size_t scratch_bytes, void* stream) {
    VINCE_CHECK_ARG(dtype == VINCE_BF16, VINCE_E_DTYPE, "vince_bn_train_apply_gram: bf16 only (dtype %d)", dtype);
    VINCE_CHECK_ARG(y && bt && out && gram && scratch && rows > 0, VINCE_E_ARG, "vince_bn_train_apply_gram: bad arguments");
    VINCE_CHECK_ARG(C == 64 || C == 128, VINCE_E_SHAPE, "vince_bn_train_apply_gram: C=%d (64 or 128)", C);
    VINCE_CHECK_ARG(bt->stats && bt->count > 0 && bt->gamma && bt->beta && bt->scale && bt->shift, VINCE_E_ARG,
                    "vince_bn_train_apply_gram: stats, count, gamma, beta, scale and shift are required");
    VINCE_CHECK_ARG(!bt->running_mean == !bt->running_var, VINCE_E_ARG, "vince_bn_train_apply_gram: running_mean and running_var come together");
    VINCE_CHECK_ARG((((uintptr_t)y | (uintptr_t)out | (uintptr_t)gram | (uintptr_t)scratch) & 15) == 0, VINCE_E_ALIGN,
                    "vince_bn_train_apply_gram: pointers must be 16-byte aligned");
    vince_bn_train fin = *bt;
    if (fin.replicas <= 0 || fin.replicas > VINCE_STATS_REPLICAS) fin.replicas = VINCE_STATS_REPLICAS;
    int64_t wgs = (rows + GS - 1) / GS;
    if (wgs > gram_max_wgs(C)) wgs = gram_max_wgs(C);
    const int64_t fit = (int64_t)(scratch_bytes / ((size_t)C * C * sizeof(float)));
    VINCE_CHECK_ARG(fit >= 1, VINCE_E_ARG, "vince_bn_train_apply_gram: scratch holds no slab (%zu bytes)", scratch_bytes);
    if (wgs > fit) wgs = fit;
    int64_t rpw = (rows + wgs - 1) / wgs;
    rpw = (rpw + GS - 1) / GS * GS;
    wgs = (rows + rpw - 1) / rpw;
    VinceProfScope prof(VINCE_TAG_BN_APPLY, (double)rows * C * 2 * 2, stream);
    if (C == 64)
        hipLaunchKernelGGL(bn_apply_gram_kernel<64>, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y, (bf16_t*)out, rows,
                           rpw, (float*)scratch, fin);
    else
        hipLaunchKernelGGL(bn_apply_gram_kernel<128>, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y, (bf16_t*)out, rows,
                           rpw, (float*)scratch, fin);
    VINCE_CHECK_LAUNCH();
    return vince_wgrad::slab_reduce((const float*)scratch, (int)wgs, (size_t)C * C, gram, (size_t)C * C / 4, (hipStream_t)stream);
}
