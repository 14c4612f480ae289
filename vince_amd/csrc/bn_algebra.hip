// Backward of a bottleneck's last BatchNorm THROUGH its 1x1 convolution, without the convolution's output.
//
// Reference semantics: autograd of  y = conv3(a) ; u = bn3(y) ; z = relu(u + identity)  (models/building_blocks/resnet.py:123-133,
// BatchNorm2d in train mode, resnet.py:69).  With g = dL/dz gated by z > 0 (per pixel p, output channel c), s = gamma * invstd,
// c1 = mean_p g, c2 = mean_p (g * xhat), xhat = (y - mean) * invstd, BatchNorm backward is
//     dy[p][c] = s[c] * (g[p][c] - c1[c] - xhat[p][c] * c2[c])
// and conv3's gradients are  da = W^T dy ,  dW = dy^T a.  The separate-pass implementation reads g and y (4w wide each) and
// writes dy (4w wide), which dgrad and wgrad read again.  Because conv3 is LINEAR AND POINTWISE, y[p][c] = sum_j W[c][j] a[p][j],
// everything that involves y collapses onto small matrices:
//     R      = g^T a                      (the "raw" weight gradient: one wgrad launch with dy := g)
//     sum_p g[p][c] y[p][c]  = <W[c,:], R[c,:]>                     ->  c2[c] = invstd[c] (<W[c,:], R[c,:]> - mean[c] sum_p g[p][c]) / n
//     da[p][:] = (W^T diag(s)) g[p][:]  -  Q a[p][:]  -  r          with t = s * c2 * invstd,
//                Q = W^T diag(t) W   (w x w),    r[k] = sum_c W[c][k] (s[c] c1[c] - t[c] mean[c])
//     dW[c][k] = s[c] (R[c][k] - c1[c] A[k] - c2[c] invstd[c] (sum_j W[c][j] G[j][k] - mean[c] A[k]))
//                with G = a^T a (the Gram matrix the forward's statistics came from) and A[k] = sum_p a[p][k]
// so the 4w-wide tensors are read ONCE each (g by the raw wgrad and by the dgrad) and y is never stored.  W here is the bf16 copy the
// forward multiplied with, so the implied y is exactly the forward's (unrounded) convolution output.
//
// Round 6 (the mixed mode's twin runs this behind a forward that multiplied with the fp32 master weights): the algebra is kept
// SELF-CONSISTENT on the implied y = W_bf16 a.  (1) With the column sums A given, the mean every formula uses is the implied one,
// mean'[c] = sum_k W[c][k] A[k] / n (coef row 4), not the forward's.  (2) The separate passes hand conv3's input gradient a dy whose
// pixel sums vanish per channel, so sum_p da = 0 whatever rounding the weights carry; here wd and nq are rounded to bf16 SEPARATELY from
// the constant r, and the residue (2^-9 of each term, the same sign at every pixel) lands in the pixel sums the BatchNorm below
// reduces -- bn2's bias gradient was 1e-1 off where the separate passes give 1e-2.  So nr is formed FROM THE ROUNDED matrices:
//     nr[k] = - sum_c wd_r[k][c] c1[c] - sum_j nq_r[k][j] A[j] / n            (= -r[k] in exact arithmetic)
// which makes sum_p da[p][k] = 0 to fp32 rounding again.  (3) That alone is not enough (tools/alg_emu.py restates the arithmetic on the
// CPU): what the BatchNorm below reduces are MASKED sums (its ReLU), the rounding residue of nq multiplies a -- zero exactly where the
// mask is -- and the residue of wd multiplies the part of g that correlates with xhat, hence with a.  Both matrices therefore come in
// bfloat16 hi + lo parts (wd_lo / nq_lo non-NULL: 16 mantissa bits) and the input gradient reduces over [wd_hi | wd_lo] g and
// [nq_hi | nq_lo] a (vince_conv_epi.in2_repeat); and W may be the fp32 master (w_dtype VINCE_F32) -- the implied y is then the forward's
// to the rounding of the stored a.  With all three the bias gradients of bn2 sit where the separate passes put them (G9: 1.3e-2 worst
// sum |g|; 1.25e-1 with single bf16 matrices).  The bf16 mode keeps single matrices from its own bf16 weights (its whole step is at that grade).
//
// Three small launches per block (w <= 256, 4w <= 1024: everything fits simple one-thread-per-output loops):
//   bn3_prepare_kernel   per output channel: the dot product, c1, c2, s, t, dgamma / dbeta
//   bn3_derive_kernel    wd = bf16(W^T diag(s)) [w][4w],  nq = bf16(-Q) [w][w] (row strides given: the engine interleaves them as the
//                        two taps of ONE input-gradient launch, vince_conv_epi.in2),  nr = -r [w]
//   bn3_finish_dw_kernel dW from R -- only the optimiser waits for it: the engine runs it on its weight-gradient stream (round 5)
// Round 5 also tried the first two as ONE launch, every workgroup deriving all coefficients itself (a launch gap saved on backward's
// critical path): one thread per channel (64 different rows per load instruction) 14-27 us, rows read coalesced by lane groups with
// a shuffle fold (a chain of 64 dependent passes) 25-70 us -- against 11 + 6 us for the pair below.  Kept as two.
#include <string.h>

#include "common.h"

namespace {

static __device__ __forceinline__ float wload(const bf16_t* W, size_t i) { return bf16_to_f32(W[i]); }
static __device__ __forceinline__ float wload(const float* W, size_t i) { return W[i]; }
// v -> bf16 hi (+ bf16 lo of the remainder when lo != nullptr); returns what the pair represents
static __device__ __forceinline__ float store_split(bf16_t* hi, bf16_t* lo, size_t i, float v) {
    const bf16_t h = f32_to_bf16(v);
    hi[i] = h;
    float r = bf16_to_f32(h);
    if (lo) {
        const bf16_t l = f32_to_bf16(v - r);
        lo[i] = l;
        r += bf16_to_f32(l);
    }
    return r;
}

// coef layout: float[5][Co] = s, c1, c2, t, the mean the formulas use
constexpr int ALG_MAX_CO = 1024, ALG_MAX_K = 256;
template <typename WT>
__global__ __launch_bounds__(256) void bn3_prepare_kernel(const float* __restrict__ R, const WT* __restrict__ W, const double* __restrict__ gsums,
                                                          int replicas, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, double inv_n, int Co, int K, float* __restrict__ coef,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, const double* __restrict__ colsum,
                                                          int colsum_replicas, float* __restrict__ nr) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);   // one wavefront per channel
    if (c >= Co) return;
    if (c < K && lane == 0) nr[c] = 0.f;          // (bn3_derive_kernel adds its two parts: K <= Co)
    float dot = 0.f;
    double mimp = 0;
    for (int k = lane; k < K; k += 64) {
        const float w = wload(W, (size_t)c * K + k);
        dot += w * R[(size_t)c * K + k];
        if (colsum) {
            double A = 0;
            for (int r = 0; r < colsum_replicas; ++r) A += colsum[(size_t)r * K + k];
            mimp += (double)w * A;
        }
    }
    dot = wave_sum(dot);
    if (colsum) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mimp += __shfl_xor(mimp, o);
    }
    if (lane == 0) {
        double sg = 0;
        for (int r = 0; r < replicas; ++r) sg += gsums[((size_t)r * Co + c) * 2];
        const double is = invstd[c], mu = colsum ? mimp * inv_n : (double)mean[c];
        coef[4 * Co + c] = (float)mu;
        const double sgx = is * ((double)dot - mu * sg);
        const double s = (double)gamma[c] * is, c1 = sg * inv_n, c2 = sgx * inv_n;
        coef[c] = (float)s;
        coef[Co + c] = (float)c1;
        coef[2 * Co + c] = (float)c2;
        coef[3 * Co + c] = (float)(s * c2 * is);
        dgamma[c] += (float)sgx;
        dbeta[c] += (float)sg;
    }
}

// grid: blocks 0 .. K-1 produce row k of wd (and nr[k]); blocks K .. 2K-1 produce row k of nq
// (colsum == nullptr: nr from the unrounded coefficients, -sum_c W[c][k] (s c1 - t mean) -- the form of rounds 3-5, kept for comparison)
template <typename WT>
__global__ __launch_bounds__(256) void bn3_derive_kernel(const WT* __restrict__ W, const float* __restrict__ coef,
                                                         int Co, int K, bf16_t* __restrict__ wd, int wd_ld, bf16_t* __restrict__ nq, int nq_ld, float* __restrict__ nr,
                                                         const double* __restrict__ colsum, int colsum_replicas, double inv_n,
                                                         bf16_t* __restrict__ wd_lo, bf16_t* __restrict__ nq_lo) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const float* const mean = coef + 4 * (size_t)Co;
    if ((int)blockIdx.x < K) {
        const int k = blockIdx.x;
        float acc = 0.f;
        for (int c = tid; c < Co; c += 256) {
            const float w = wload(W, (size_t)c * K + k);
            const float s = coef[c], c1 = coef[Co + c], t = coef[3 * Co + c];
            const float r = store_split(wd, wd_lo, (size_t)k * wd_ld + c, s * w);
            acc += colsum ? r * c1 : w * (s * c1 - t * mean[c]);
        }
        red[tid] = acc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        if (tid == 0) atomicAdd(&nr[k], -red[0]);      // (two addends per entry, onto the zero bn3_prepare_kernel left: order-independent)
    } else {
        // nq[k][j] = -sum_c W[c][k] t[c] W[c][j].  Column k of W scaled by t is staged in LDS once; thread = (j, part): the Co
        // reduction is split over the 256 / K parts of a column (row c of W is read coalesced across j), folded through LDS.
        __shared__ float tk[ALG_MAX_CO];
        const int k = blockIdx.x - K;
        for (int c = tid; c < Co; c += 256) tk[c] = coef[3 * Co + c] * wload(W, (size_t)c * K + k);
        __syncthreads();
        const int parts = 256 / K, j = tid % K, part = tid / K;
        float acc = 0.f;
        if (part < parts) {
            const int c0 = part * (Co / parts), c1 = part + 1 == parts ? Co : c0 + Co / parts;
#pragma unroll 8
            for (int c = c0; c < c1; ++c) acc += tk[c] * wload(W, (size_t)c * K + j);
        }
        red[tid] = acc;
        __syncthreads();
        float part_nr = 0.f;
        if (tid < K) {
            float sum = 0.f;
            for (int q = 0; q < parts; ++q) sum += red[q * K + tid];
            const float r = store_split(nq, nq_lo, (size_t)k * nq_ld + tid, -sum);
            if (colsum) {
                double A = 0;
                for (int q = 0; q < colsum_replicas; ++q) A += colsum[(size_t)q * K + tid];
                part_nr = r * (float)(A * inv_n);
            }
        }
        if (colsum) {
            __syncthreads();
            red[tid] = part_nr;
            __syncthreads();
            for (int o = 128; o > 0; o >>= 1) {
                if (tid < o) red[tid] += red[tid + o];
                __syncthreads();
            }
            if (tid == 0) atomicAdd(&nr[k], -red[0]);
        }
    }
}

// dW[c][k] = s (R - c1 A[k] - c2 invstd (sum_j W[c][j] G[j][k] - mean A[k])), one workgroup per channel c, thread k
// dw_accum != nullptr: R is read-only scratch and the finished gradient is ADDED into dw_accum (the accumulate-into contract of every other
// weight gradient); nullptr: in place, R becomes dW
template <typename WT>
__global__ __launch_bounds__(256) void bn3_finish_dw_kernel(float* __restrict__ RdW, float* __restrict__ dw_accum, const WT* __restrict__ W, const float* __restrict__ gram,
                                                            const double* __restrict__ colsum, int colsum_replicas, const float* __restrict__ coef,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd, int Co, int K) {
    __shared__ float wrow[ALG_MAX_K];
    const int c = blockIdx.x, k = threadIdx.x;
    if (k < K) wrow[k] = wload(W, (size_t)c * K + k);
    __syncthreads();
    if (k >= K) return;
    double A = 0;
    for (int r = 0; r < colsum_replicas; ++r) A += colsum[(size_t)r * K + k];
    // centred cross term in fp64: sum_j W[c][j] G[j][k] and mean * A[k] are both ~ n * |y| * |a| and nearly cancel
    double wg = 0;
    for (int j = 0; j < K; ++j) wg += (double)wrow[j] * (double)gram[(size_t)j * K + k];
    const double s = coef[c], c1 = coef[Co + c], c2 = coef[2 * Co + c];
    const double v = s * ((double)RdW[(size_t)c * K + k] - c1 * A - c2 * (double)invstd[c] * (wg - (double)mean[c] * A));
    if (dw_accum) dw_accum[(size_t)c * K + k] += (float)v;
    else RdW[(size_t)c * K + k] = (float)v;
}

}  // namespace

template <typename WT>
static void launch_prepare(const float* R, const void* w, const double* gsums, int replicas, const float* mean, const float* invstd, const float* gamma,
                           int64_t count, int Co, int K, float* coef, void* wd, int wd_ld, void* nq, int nq_ld, float* nr, float* dgamma, float* dbeta,
                           const double* colsum, int colsum_replicas, void* wd_lo, void* nq_lo, hipStream_t stream) {
    hipLaunchKernelGGL(bn3_prepare_kernel<WT>, dim3((Co + 3) / 4), dim3(256), 0, stream, R, (const WT*)w, gsums, replicas,
                       mean, invstd, gamma, 1.0 / (double)count, Co, K, coef, dgamma, dbeta, colsum, colsum_replicas, nr);
    hipLaunchKernelGGL(bn3_derive_kernel<WT>, dim3(2 * K), dim3(256), 0, stream, (const WT*)w, (const float*)coef, Co, K,
                       (bf16_t*)wd, wd_ld, (bf16_t*)nq, nq_ld, nr, colsum, colsum_replicas, 1.0 / (double)count, (bf16_t*)wd_lo, (bf16_t*)nq_lo);
}

extern "C" int vince_bn3_bwd_prepare(const float* R, const void* w_bf16, const double* gsums, int32_t replicas, const float* mean,
                                     const float* invstd, const float* gamma, int64_t count, int32_t Co, int32_t K, float* coef,
                                     void* wd, int32_t wd_ld, void* nq, int32_t nq_ld, float* nr, float* dgamma, float* dbeta,
                                     const double* colsum, int32_t colsum_replicas, int32_t w_dtype, void* wd_lo, void* nq_lo, void* stream) {
    VINCE_CHECK_ARG(R && w_bf16 && gsums && mean && invstd && gamma && coef && wd && nq && nr && dgamma && dbeta, VINCE_E_ARG,
                    "vince_bn3_bwd_prepare: null pointer");
    VINCE_CHECK_ARG((w_dtype == VINCE_BF16 || w_dtype == VINCE_F32) && (!wd_lo || nq_lo), VINCE_E_ARG,
                    "vince_bn3_bwd_prepare: w_dtype %d (bf16 or fp32); wd_lo needs nq_lo", w_dtype);
    VINCE_CHECK_ARG(count > 0 && Co > 0 && Co <= ALG_MAX_CO && (K == 64 || K == 128 || K == 256), VINCE_E_SHAPE,
                    "vince_bn3_bwd_prepare: K=%d (64, 128 or 256), Co=%d (at most %d)", K, Co, ALG_MAX_CO);
    VINCE_CHECK_ARG(K <= Co && (!colsum || colsum_replicas > 0), VINCE_E_SHAPE, "vince_bn3_bwd_prepare: K=%d > Co=%d, or colsum without replicas", K, Co);
    if (replicas <= 0 || replicas > VINCE_STATS_REPLICAS) replicas = VINCE_STATS_REPLICAS;
    if (w_dtype == VINCE_F32)
        launch_prepare<float>(R, w_bf16, gsums, replicas, mean, invstd, gamma, count, Co, K, coef, wd, wd_ld, nq, nq_ld, nr, dgamma, dbeta, colsum,
                              colsum_replicas, wd_lo, nq_lo, (hipStream_t)stream);
    else
        launch_prepare<bf16_t>(R, w_bf16, gsums, replicas, mean, invstd, gamma, count, Co, K, coef, wd, wd_ld, nq, nq_ld, nr, dgamma, dbeta, colsum,
                               colsum_replicas, wd_lo, nq_lo, (hipStream_t)stream);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_bn3_bwd_finish_dw(float* RdW, float* dw_accum, const void* w_bf16, const float* gram, const double* colsum, int32_t colsum_replicas,
                                       const float* coef, const float* mean, const float* invstd, int32_t Co, int32_t K, int32_t w_dtype, void* stream) {
    VINCE_CHECK_ARG(RdW && w_bf16 && gram && colsum && coef && mean && invstd, VINCE_E_ARG, "vince_bn3_bwd_finish_dw: null pointer");
    VINCE_CHECK_ARG(w_dtype == VINCE_BF16 || w_dtype == VINCE_F32, VINCE_E_ARG, "vince_bn3_bwd_finish_dw: w_dtype %d (bf16 or fp32)", w_dtype);
    VINCE_CHECK_ARG(Co > 0 && K > 0 && K <= ALG_MAX_K && colsum_replicas > 0, VINCE_E_SHAPE, "vince_bn3_bwd_finish_dw: K=%d (at most %d)", K, ALG_MAX_K);
    const dim3 block(K <= 64 ? 64 : K <= 128 ? 128 : 256);
    if (w_dtype == VINCE_F32)
        hipLaunchKernelGGL(bn3_finish_dw_kernel<float>, dim3(Co), block, 0, (hipStream_t)stream, RdW, dw_accum, (const float*)w_bf16, gram, colsum,
                           colsum_replicas, coef, mean, invstd, Co, K);
    else
        hipLaunchKernelGGL(bn3_finish_dw_kernel<bf16_t>, dim3(Co), block, 0, (hipStream_t)stream, RdW, dw_accum, (const bf16_t*)w_bf16, gram, colsum,
                           colsum_replicas, coef, mean, invstd, Co, K);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}
