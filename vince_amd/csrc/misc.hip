// Layout transforms, head glue (L2 normalise, ReLU backward, bias gradient), FIFO queue write, momentum (EMA) update
// and SGD -- the bandwidth-bound odds and ends of the path.
//
// Reference call sites: batch shuffle gather vince_model.py:137-142; jigsaw tiling :144-155; F.normalize :180;
// StorageQueue.enqueue utils/storage_queue.py:31-49; VinceQueueModel.param_update vince_model.py:587-592;
// optim.SGD vince_solver.py:256,469.
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void input_to_nhwc_kernel(const float* __restrict__ in, const int64_t* __restrict__ perm,
                                                            T* __restrict__ out, int N, int C, int H, int W, int Cp) {
    const int64_t total = (int64_t)N * H * W;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t hw = idx % ((int64_t)H * W);
        const int n = (int)(idx / ((int64_t)H * W));
        const int64_t src_n = perm ? perm[n] : n;
        float f[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) f[c] = (c < C) ? in[((size_t)src_n * C + c) * H * W + hw] : 0.f;
        if constexpr (sizeof(T) == 4) {
            *(float4*)((float*)out + (size_t)idx * Cp) = make_float4(f[0], f[1], f[2], f[3]);
        } else {
            *(uint4*)((bf16_t*)out + (size_t)idx * Cp) = Chunk<bf16_t>::pack(f);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void jigsaw_to_nhwc_kernel(const float* __restrict__ in, T* __restrict__ out, int N,
                                                             int C, int H, int W, int th, int tw, int Cp) {
    // out image index = n*9 + ty*3 + tx (vince_model.py:151-155: permute(0,2,4,1,3,5) then merge dims 0,1,2)
    const int64_t total = (int64_t)N * 9 * th * tw;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int x = (int)(idx % tw);
        int64_t r = idx / tw;
        const int y = (int)(r % th);
        r /= th;
        const int tile = (int)(r % 9);
        const int n = (int)(r / 9);
        const int sy = (tile / 3) * th + y, sx = (tile % 3) * tw + x;
        float f[8];
        const bool inside = sy < H && sx < W;   // zero padding on the bottom/right (F.pad, vince_model.py:146)
#pragma unroll
        for (int c = 0; c < 8; ++c) f[c] = (c < C && inside) ? in[(((size_t)n * C + c) * H + sy) * W + sx] : 0.f;
        if constexpr (sizeof(T) == 4) {
            *(float4*)((float*)out + (size_t)idx * Cp) = make_float4(f[0], f[1], f[2], f[3]);
        } else {
            *(uint4*)((bf16_t*)out + (size_t)idx * Cp) = Chunk<bf16_t>::pack(f);
        }
    }
}

// Packed-row-tap layouts (4 channels per pixel, zero margins): one thread per output pixel incl. the margins.
template <typename T>
__device__ __forceinline__ void store_px4(T* out, size_t pix, const float (&f)[4]) {
    if constexpr (sizeof(T) == 4) {
        *(float4*)((float*)out + pix * 4) = make_float4(f[0], f[1], f[2], f[3]);
    } else {
        *(uint2*)((bf16_t*)out + pix * 4) = make_uint2(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void input_to_rows_kernel(const float* __restrict__ in, const int64_t* __restrict__ perm,
                                                            T* __restrict__ out, int N, int C, int H, int W, int Wp, int left) {
    const int64_t total = (int64_t)N * H * Wp;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int wp = (int)(idx % Wp);
        const int64_t r = idx / Wp;
        const int h = (int)(r % H);
        const int n = (int)(r / H);
        const int w = wp - left;
        const bool inside = w >= 0 && w < W;
        const int64_t src_n = perm ? perm[n] : n;
        float f[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) f[c] = (c < C && inside) ? in[(((size_t)src_n * C + c) * H + h) * W + w] : 0.f;
        store_px4<T>(out, (size_t)idx, f);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void jigsaw_to_rows_kernel(const float* __restrict__ in, T* __restrict__ out, int N, int C,
                                                             int H, int W, int th, int tw, int Wp, int left) {
    const int64_t total = (int64_t)N * 9 * th * Wp;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int wp = (int)(idx % Wp);
        int64_t r = idx / Wp;
        const int y = (int)(r % th);
        r /= th;
        const int tile = (int)(r % 9);
        const int n = (int)(r / 9);
        const int x = wp - left;
        const int sy = (tile / 3) * th + y, sx = (tile % 3) * tw + x;
        const bool inside = x >= 0 && x < tw && sy < H && sx < W;   // F.pad zero padding (vince_model.py:146) + margins
        float f[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) f[c] = (c < C && inside) ? in[(((size_t)n * C + c) * H + sy) * W + sx] : 0.f;
        store_px4<T>(out, (size_t)idx, f);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void input_u8_to_rows_kernel(const uint8_t* __restrict__ in, const int64_t* __restrict__ perm,
                                                               const int32_t* __restrict__ crop, const uint8_t* __restrict__ flip,
                                                               float m0, float m1, float m2, float s0, float s1, float s2,
                                                               T* __restrict__ out, int N, int Hs, int Ws, int H, int W,
                                                               int Wp, int left) {
    const int64_t total = (int64_t)N * H * Wp;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int wp = (int)(idx % Wp);
        const int64_t r = idx / Wp;
        const int h = (int)(r % H);
        const int n = (int)(r / H);
        const int w = wp - left;
        float f[4] = {0.f, 0.f, 0.f, 0.f};
        if (w >= 0 && w < W) {
            const int64_t sn = perm ? perm[n] : n;
            const int cy = crop ? crop[2 * sn] : 0, cx = crop ? crop[2 * sn + 1] : 0;
            const int sw_ = (flip && flip[sn]) ? (W - 1 - w) : w;    // RandomHorizontalFlip acts on the cropped window
            const uint8_t* px = in + (((size_t)sn * Hs + (cy + h)) * Ws + (cx + sw_)) * 3;
            f[0] = __fdiv_rn((float)px[0] - m0, s0);   // correctly rounded, like a CPU (x - mean) / std
            f[1] = __fdiv_rn((float)px[1] - m1, s1);
            f[2] = __fdiv_rn((float)px[2] - m2, s2);
        }
        store_px4<T>(out, (size_t)idx, f);
    }
}

template <typename T> __device__ inline T cvt_from_f32(float f);
template <> __device__ inline float cvt_from_f32<float>(float f) { return f; }
template <> __device__ inline bf16_t cvt_from_f32<bf16_t>(float f) { return f32_to_bf16(f); }
template <typename T> __device__ inline float cvt_to_f32(T v);
template <> __device__ inline float cvt_to_f32<float>(float v) { return v; }
template <> __device__ inline float cvt_to_f32<bf16_t>(bf16_t v) { return bf16_to_f32(v); }

template <typename T>
__global__ __launch_bounds__(256) void prepare_weight_kernel(const float* __restrict__ w, T* __restrict__ wk,
                                                             T* __restrict__ wt, int Co, int Tt, int Ci, int Cip) {
    const int64_t total = (int64_t)Co * Tt * Cip;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int ci = (int)(idx % Cip);
        const int64_t r = idx / Cip;
        const int t = (int)(r % Tt);
        const int co = (int)(r / Tt);
        const float v = ci < Ci ? w[((size_t)co * Tt + t) * Ci + ci] : 0.f;
        const T o = cvt_from_f32<T>(v);
        wk[idx] = o;
        if (wt && ci < Ci) wt[((size_t)ci * Tt + t) * Co + co] = o;
    }
}

// All conv layers of a trunk in one launch: blockIdx.y = layer (descriptor table in device memory), blockIdx.x strides
// over that layer's elements.
template <typename T>
__global__ __launch_bounds__(256) void prepare_weights_batched_kernel(const vince_prep_entry* __restrict__ table, int tiled) {
    const vince_prep_entry e = table[blockIdx.y];
    const float* __restrict__ w = (const float*)e.w;
    const float* __restrict__ sc = e.scale;   // optional per-output-channel multiplier (BatchNorm folding)
    T* __restrict__ wk = (T*)e.wk;
    T* __restrict__ wt = (T*)e.wt;
    if (e.Cs > 0) {   // packed row taps (the stem): wk[co][t][k] = w[co][t][k / Cs][k % Cs]
        const int64_t total = (int64_t)e.Co * e.T * e.Cip;
        for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
            const int k = (int)(idx % e.Cip);
            const int64_t r = idx / e.Cip;   // co * T + t
            const int kw = k / e.Cs, c = k - kw * e.Cs;
            const float m = sc ? sc[r / e.T] : 1.f;
            wk[idx] = cvt_from_f32<T>((kw < e.Kw && c < e.Ci) ? w[((size_t)r * e.Kw + kw) * e.Ci + c] * m : 0.f);
        }
        return;
    }
    if (tiled && wt && e.Ci % 64 == 0 && e.Co % 64 == 0 && e.Cip == e.Ci) {
        // 64 x 64 (co x ci) tiles through LDS: the [Co][T][Ci] copy and its [Ci][T][Co] transpose are both written in
        // whole rows (element-wise transposed stores cost ~5x their bytes in partial-line HBM writes)
        __shared__ float tile[64][65];
        const int tci = e.Ci / 64, tco = e.Co / 64;
        const int ntiles = tco * e.T * tci;
        const int col = threadIdx.x & 63, rq = threadIdx.x >> 6;
        for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
            const int ci0 = (tl % tci) * 64;
            const int t = (tl / tci) % e.T;
            const int co0 = (tl / (tci * e.T)) * 64;
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const int row = i * 4 + rq;
                const size_t off = ((size_t)(co0 + row) * e.T + t) * e.Ci + ci0 + col;
                const float v = w[off] * (sc ? sc[co0 + row] : 1.f);
                wk[off] = cvt_from_f32<T>(v);
                tile[row][col] = v;
            }
            __syncthreads();
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const int row = i * 4 + rq;   // ci within the tile
                wt[((size_t)(ci0 + row) * e.T + t) * e.Co + co0 + col] = cvt_from_f32<T>(tile[col][row]);
            }
            __syncthreads();
        }
        return;
    }
    const int64_t total = (int64_t)e.Co * e.T * e.Cip;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int ci = (int)(idx % e.Cip);
        const int64_t r = idx / e.Cip;
        const int t = (int)(r % e.T);
        const int co = (int)(r / e.T);
        const float v = ci < e.Ci ? w[((size_t)co * e.T + t) * e.Ci + ci] * (sc ? sc[co] : 1.f) : 0.f;
        const T o = cvt_from_f32<T>(v);
        wk[idx] = o;
        if (wt && ci < e.Ci) wt[((size_t)ci * e.T + t) * e.Co + co] = o;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Split-half weight layout (dtype VINCE_F32X3; csrc/common.h x3_split).  The convolution kernels split every ACTIVATION fragment into
// hi / lo halves in registers; the weights are split HERE, once per parameter update, so that a weight fragment is two plain 16-byte
// LDS reads.  A K-contiguous row (K = Cip for the forward copy, Co for the transposed one; multiples of 16) is stored in groups of
// 16 elements = 64 bytes -- the footprint of 16 floats -- as four 16-byte chunks
//      [hi of e0-3, e8-11] [hi of e4-7, e12-15] [lo of e0-3, e8-11] [lo of e4-7, e12-15]
// which is the order in which MFMA lane half `h` finds the 8 activations of its K slots (16-byte chunks h and 2 + h of the fp32 row).
// wk: IEEE half pairs scaled by 2^X3_WSHIFT (forward launches, VINCE_F32X3H); wt: bfloat16 pairs (gradient launches, VINCE_F32X3B).
template <typename T> __device__ __forceinline__ void x3_store_group(void* dst, const float* v) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint4 hi, lo;
        x3_split<T, true>(make_uint4(__float_as_uint(v[4 * h]), __float_as_uint(v[4 * h + 1]), __float_as_uint(v[4 * h + 2]), __float_as_uint(v[4 * h + 3])),
                          make_uint4(__float_as_uint(v[8 + 4 * h]), __float_as_uint(v[9 + 4 * h]), __float_as_uint(v[10 + 4 * h]), __float_as_uint(v[11 + 4 * h])),
                          hi, lo);
        *(uint4*)((unsigned char*)dst + 16 * h) = hi;
        *(uint4*)((unsigned char*)dst + 32 + 16 * h) = lo;
    }
}

template <typename WKT>
__device__ void prep_x3_entry(const vince_prep_entry& e, unsigned char* tile_raw) {
    const float* __restrict__ w = (const float*)e.w;
    const float* __restrict__ sc = e.scale;
    unsigned char* __restrict__ wk = (unsigned char*)e.wk;
    unsigned char* __restrict__ wt = (unsigned char*)e.wt;
    if (e.Cs > 0) {   // packed row taps (the stem): element k of tap t is (kw = k / Cs, c = k % Cs); forward copy only
        const int64_t groups = (int64_t)e.Co * e.T * (e.Cip / 16);
        for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (int64_t)gridDim.x * 256) {
            const int k0 = (int)(g % (e.Cip / 16)) * 16;
            const int64_t r = g / (e.Cip / 16);   // co * T + t
            const float m = sc ? sc[r / e.T] : 1.f;
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k = k0 + i, kw = k / e.Cs, c = k - kw * e.Cs;
                v[i] = (kw < e.Kw && c < e.Ci) ? w[((size_t)r * e.Kw + kw) * e.Ci + c] * m : 0.f;
            }
            x3_store_group<WKT>(wk + ((size_t)r * e.Cip + k0) * 4, v);
        }
        return;
    }
    if (wt && e.Ci % 64 == 0 && e.Co % 64 == 0 && e.Cip == e.Ci) {
        // 64 x 64 (co x ci) tiles through LDS, both copies written as whole 64-byte groups
        float (*tile)[65] = (float (*)[65])tile_raw;
        const int tci = e.Ci / 64, tco = e.Co / 64;
        const int ntiles = tco * e.T * tci;
        const int row = threadIdx.x >> 2, grp = threadIdx.x & 3;
        for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
            const int ci0 = (tl % tci) * 64;
            const int t = (tl / tci) % e.T;
            const int co0 = (tl / (tci * e.T)) * 64;
            {
                const size_t off = ((size_t)(co0 + row) * e.T + t) * e.Ci + ci0 + grp * 16;
                const float m = sc ? sc[co0 + row] : 1.f;
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; i += 4) {
                    const float4 q = *(const float4*)(w + off + i);
                    v[i] = q.x * m; v[i + 1] = q.y * m; v[i + 2] = q.z * m; v[i + 3] = q.w * m;
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) tile[row][grp * 16 + i] = v[i];
                x3_store_group<WKT>(wk + off * 4, v);
            }
            __syncthreads();
            {
                float v[16];   // row = ci within the tile, the group runs along co
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = tile[grp * 16 + i][row];
                x3_store_group<x3b_t>(wt + (((size_t)(ci0 + row) * e.T + t) * e.Co + co0 + grp * 16) * 4, v);
            }
            __syncthreads();
        }
        return;
    }
    {   // any other shape (Cip and, for the transposed copy, Co multiples of 16): one group per thread, gathered element-wise
        const int64_t groups = (int64_t)e.Co * e.T * (e.Cip / 16);
        for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (int64_t)gridDim.x * 256) {
            const int k0 = (int)(g % (e.Cip / 16)) * 16;
            const int64_t r = g / (e.Cip / 16);
            const int t = (int)(r % e.T), co = (int)(r / e.T);
            const float m = sc ? sc[co] : 1.f;
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = (k0 + i) < e.Ci ? w[((size_t)co * e.T + t) * e.Ci + k0 + i] * m : 0.f;
            x3_store_group<WKT>(wk + ((size_t)r * e.Cip + k0) * 4, v);
        }
        if (wt) {
            const int64_t tg = (int64_t)e.Ci * e.T * (e.Co / 16);
            for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < tg; g += (int64_t)gridDim.x * 256) {
                const int co0 = (int)(g % (e.Co / 16)) * 16;
                const int64_t r = g / (e.Co / 16);   // ci * T + t
                const int t = (int)(r % e.T), ci = (int)(r / e.T);
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = w[((size_t)(co0 + i) * e.T + t) * e.Ci + ci] * (sc ? sc[co0 + i] : 1.f);
                x3_store_group<x3b_t>(wt + ((size_t)r * e.Co + co0) * 4, v);
            }
        }
    }
}

__global__ __launch_bounds__(256) void prepare_weights_batched_x3_kernel(const vince_prep_entry* __restrict__ table) {
    __shared__ __attribute__((aligned(16))) unsigned char tile[64 * 65 * 4];
    prep_x3_entry<x3h_t>(table[blockIdx.y], tile);
}
// WKT: the element type of the forward copy -- x3h_t (IEEE half pairs: the trunk's forward launches) or x3b_t (bfloat16 pairs: fp32's
// exponent range at 2^-16 per product, for operands no BatchNorm keeps inside the half range: the projection head)
template <typename WKT>
__global__ __launch_bounds__(256) void prepare_weight_x3_kernel(const vince_prep_entry e) {
    __shared__ __attribute__((aligned(16))) unsigned char tile[64 * 65 * 4];
    prep_x3_entry<WKT>(e, tile);
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ in, float* __restrict__ out, int N,
                                                           int C, int H, int W) {
    const int64_t total = (int64_t)N * C * H * W;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t hw = idx % ((int64_t)H * W);
        const int64_t r = idx / ((int64_t)H * W);
        const int c = (int)(r % C);
        const int n = (int)(r / C);
        out[idx] = cvt_to_f32<T>(in[((size_t)n * H * W + hw) * C + c]);
    }
}

// one wavefront per row
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                         float* __restrict__ norms, int rows, int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) { const float v = x[(size_t)row * D + i]; s += v * v; }
    s = wave_sum(s);
    const float nrm = sqrtf(s), den = fmaxf(nrm, eps);   // F.normalize: x / max(||x||, eps)
    for (int i = lane; i < D; i += 64) out[(size_t)row * D + i] = x[(size_t)row * D + i] / den;
    if (lane == 0 && norms) norms[row] = nrm;
}

__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ norms,
                                                         const float* __restrict__ dout, float* __restrict__ dx,
                                                         int rows, int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float nrm = norms[row];
    if (nrm > eps) {
        float dot = 0.f;
        for (int i = lane; i < D; i += 64) dot += dout[(size_t)row * D + i] * x[(size_t)row * D + i];
        dot = wave_sum(dot);
        const float inv = 1.f / nrm, k = dot * inv * inv * inv;
        for (int i = lane; i < D; i += 64) dx[(size_t)row * D + i] = dout[(size_t)row * D + i] * inv - x[(size_t)row * D + i] * k;
    } else {
        for (int i = lane; i < D; i += 64) dx[(size_t)row * D + i] = dout[(size_t)row * D + i] / eps;
    }
}

__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ act,
                                                       float* __restrict__ dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dx[i] = act[i] > 0.f ? dout[i] : 0.f;
}

// out[c] += sum over rows of x[r][c] (bias gradients of the projection MLP: 256 x 2048).  64 columns x 4 row lanes per
// workgroup, grid.y row chunks; partial sums meet in `out` through fp32 atomics.
constexpr int COLSUM_CHUNKS = 8;
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int rows,
                                                     int cols) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const int per = (rows + COLSUM_CHUNKS - 1) / COLSUM_CHUNKS;
    const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
    float s = 0.f;
    if (c < cols)
        for (int r = r0 + ty; r < r1; r += 4) s += x[(size_t)r * cols + c];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < cols) unsafeAtomicAdd(out + c, red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]);
}

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ k, const float* __restrict__ q, int64_t n, float m,
                                                  float one_minus_m) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 a = ((float4*)k)[i];
        const float4 b = ((const float4*)q)[i];
        // mul_(m) then add_(alpha=1-m): round the product first, then a fused multiply-add like torch's CPU kernel
        a.x = fmaf(one_minus_m, b.x, a.x * m); a.y = fmaf(one_minus_m, b.y, a.y * m);
        a.z = fmaf(one_minus_m, b.z, a.z * m); a.w = fmaf(one_minus_m, b.w, a.w * m);
        ((float4*)k)[i] = a;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        k[i] = fmaf(one_minus_m, q[i], k[i] * m);
    }
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  int64_t n, float lr, float mom, float wd, float gs) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 pv = ((float4*)p)[i];
        const float4 gv = ((const float4*)g)[i];
        float4 bv = ((float4*)buf)[i];
        float d;
        d = fmaf(wd, pv.x, gv.x * gs); bv.x = fmaf(bv.x, mom, d); pv.x = fmaf(-lr, bv.x, pv.x);
        d = fmaf(wd, pv.y, gv.y * gs); bv.y = fmaf(bv.y, mom, d); pv.y = fmaf(-lr, bv.y, pv.y);
        d = fmaf(wd, pv.z, gv.z * gs); bv.z = fmaf(bv.z, mom, d); pv.z = fmaf(-lr, bv.z, pv.z);
        d = fmaf(wd, pv.w, gv.w * gs); bv.w = fmaf(bv.w, mom, d); pv.w = fmaf(-lr, bv.w, pv.w);
        ((float4*)p)[i] = pv;
        ((float4*)buf)[i] = bv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        const float d = fmaf(wd, p[i], g[i] * gs);
        const float b = fmaf(buf[i], mom, d);
        buf[i] = b;
        p[i] = fmaf(-lr, b, p[i]);
    }
}

inline int grid_for(int64_t total_threads) {
    int64_t b = (total_threads + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

#define DTYPE_OK(fn) VINCE_CHECK_ARG(dtype == VINCE_F32 || dtype == VINCE_BF16, VINCE_E_DTYPE, fn ": bad dtype %d", dtype)

extern "C" int vince_input_nchw_to_nhwc(int dtype, const float* in, const int64_t* perm, void* out, int32_t N, int32_t C,
                                        int32_t H, int32_t W, int32_t Cp, void* stream) {
    DTYPE_OK("vince_input_nchw_to_nhwc");
    VINCE_CHECK_ARG(in && out && N > 0 && H > 0 && W > 0, VINCE_E_ARG, "vince_input_nchw_to_nhwc: bad arguments");
    VINCE_CHECK_ARG(C >= 1 && C <= Cp && Cp == (dtype == VINCE_F32 ? 4 : 8), VINCE_E_SHAPE,
                    "vince_input_nchw_to_nhwc: C=%d Cp=%d unsupported", C, Cp);
    const int64_t total = (int64_t)N * H * W;
    if (dtype == VINCE_F32)
        hipLaunchKernelGGL(input_to_nhwc_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, perm,
                           (float*)out, N, C, H, W, Cp);
    else
        hipLaunchKernelGGL(input_to_nhwc_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, perm,
                           (bf16_t*)out, N, C, H, W, Cp);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_jigsaw_nchw_to_nhwc(int dtype, const float* in, void* out, int32_t N, int32_t C, int32_t H, int32_t W,
                                         int32_t th, int32_t tw, int32_t Cp, void* stream) {
    DTYPE_OK("vince_jigsaw_nchw_to_nhwc");
    VINCE_CHECK_ARG(in && out && N > 0 && H > 0 && W > 0 && th > 0 && tw > 0, VINCE_E_ARG, "vince_jigsaw_nchw_to_nhwc: bad arguments");
    VINCE_CHECK_ARG(C >= 1 && C <= Cp && Cp == (dtype == VINCE_F32 ? 4 : 8), VINCE_E_SHAPE,
                    "vince_jigsaw_nchw_to_nhwc: C=%d Cp=%d unsupported", C, Cp);
    VINCE_CHECK_ARG(3 * th >= H && 3 * tw >= W, VINCE_E_SHAPE, "vince_jigsaw_nchw_to_nhwc: tiles %dx%d do not cover %dx%d", th, tw, H, W);
    const int64_t total = (int64_t)N * 9 * th * tw;
    if (dtype == VINCE_F32)
        hipLaunchKernelGGL(jigsaw_to_nhwc_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in,
                           (float*)out, N, C, H, W, th, tw, Cp);
    else
        hipLaunchKernelGGL(jigsaw_to_nhwc_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in,
                           (bf16_t*)out, N, C, H, W, th, tw, Cp);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_prepare_weight(int dtype, const float* w, void* wk, void* wt, int32_t Co, int32_t T, int32_t Ci,
                                    int32_t Cip, void* stream) {
    VINCE_CHECK_ARG(w && wk && Co > 0 && T > 0 && Ci > 0 && Cip >= Ci, VINCE_E_ARG, "vince_prepare_weight: bad arguments");
    const int64_t total = (int64_t)Co * T * Cip;
    if (dtype == VINCE_F32X3 || dtype == VINCE_F32X3B) {   // split-half layout: wk = IEEE half pairs (VINCE_F32X3B: bfloat16 pairs), wt = bfloat16 pairs (gradients)
        VINCE_CHECK_ARG(Cip % 16 == 0 && (!wt || Co % 16 == 0), VINCE_E_SHAPE,
                        "vince_prepare_weight: the split-half layout needs Cip (and Co for the transposed copy) to be multiples of 16");
        vince_prep_entry e;
        e.w = w; e.wk = wk; e.wt = wt; e.scale = nullptr;
        e.Co = Co; e.T = T; e.Ci = Ci; e.Cip = Cip; e.Cs = e.Kw = 0;
        if (dtype == VINCE_F32X3B) hipLaunchKernelGGL(prepare_weight_x3_kernel<x3b_t>, dim3(512), dim3(256), 0, (hipStream_t)stream, e);
        else hipLaunchKernelGGL(prepare_weight_x3_kernel<x3h_t>, dim3(512), dim3(256), 0, (hipStream_t)stream, e);
        VINCE_CHECK_LAUNCH();
        return VINCE_OK;
    }
    DTYPE_OK("vince_prepare_weight");
    if (dtype == VINCE_F32)
        hipLaunchKernelGGL(prepare_weight_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w,
                           (float*)wk, (float*)wt, Co, T, Ci, Cip);
    else
        hipLaunchKernelGGL(prepare_weight_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w,
                           (bf16_t*)wk, (bf16_t*)wt, Co, T, Ci, Cip);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

namespace {
// out[c][r] = in[r][c] through a padded 64 x 64 LDS tile: both sides move 256-byte row segments
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty + 4 * i, c = c0 + tx;
        tile[ty + 4 * i][tx] = (r < rows && c < cols) ? in[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i, r = r0 + tx;
        if (c < cols && r < rows) out[(size_t)c * rows + r] = tile[tx][ty + 4 * i];
    }
}
}  // namespace

extern "C" int vince_transpose_f32(const float* in, float* out, int32_t rows, int32_t cols, void* stream) {
    VINCE_CHECK_ARG(in && out && rows > 0 && cols > 0, VINCE_E_ARG, "vince_transpose_f32: bad arguments");
    hipLaunchKernelGGL(transpose_f32_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream, in, out,
                       rows, cols);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_prepare_weights_batched(int dtype, const vince_prep_entry* table_dev, int32_t n, void* stream) {
    VINCE_CHECK_ARG(table_dev && n > 0, VINCE_E_ARG, "vince_prepare_weights_batched: bad arguments");
    static const int tiled = (VINCE_MEASURE_KNOB("prep_tiled", 1) != 0);   // measurement aid
    const dim3 grid(512, n);   // blocks beyond a small layer's element count fall through the grid-stride loop at once
    if (dtype == VINCE_F32X3) {   // (every entry: Cip and, with a transposed copy, Co multiples of 16 -- the caller's contract)
        hipLaunchKernelGGL(prepare_weights_batched_x3_kernel, grid, dim3(256), 0, (hipStream_t)stream, table_dev);
        VINCE_CHECK_LAUNCH();
        return VINCE_OK;
    }
    DTYPE_OK("vince_prepare_weights_batched");
    if (dtype == VINCE_F32)
        hipLaunchKernelGGL(prepare_weights_batched_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, table_dev, tiled);
    else
        hipLaunchKernelGGL(prepare_weights_batched_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, table_dev, tiled);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_nhwc_to_nchw_f32(int dtype, const void* in, float* out, int32_t N, int32_t C, int32_t H, int32_t W,
                                      void* stream) {
    DTYPE_OK("vince_nhwc_to_nchw_f32");
    VINCE_CHECK_ARG(in && out && N > 0 && C > 0 && H > 0 && W > 0, VINCE_E_ARG, "vince_nhwc_to_nchw_f32: bad arguments");
    const int64_t total = (int64_t)N * C * H * W;
    if (dtype == VINCE_F32)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)in, out, N, C, H, W);
    else
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)in, out, N, C, H, W);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_l2norm_fwd(const float* x, float* out, float* norms, int32_t rows, int32_t D, float eps, void* stream) {
    VINCE_CHECK_ARG(x && out && rows > 0 && D > 0, VINCE_E_ARG, "vince_l2norm_fwd: bad arguments");
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, out, norms, rows, D, eps);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_l2norm_bwd(const float* x, const float* norms, const float* dout, float* dx, int32_t rows, int32_t D,
                                float eps, void* stream) {
    VINCE_CHECK_ARG(x && norms && dout && dx && rows > 0 && D > 0, VINCE_E_ARG, "vince_l2norm_bwd: bad arguments");
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, norms, dout, dx, rows, D, eps);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_relu_bwd(const float* dout, const float* act, float* dx, int64_t n, void* stream) {
    VINCE_CHECK_ARG(dout && act && dx && n > 0, VINCE_E_ARG, "vince_relu_bwd: bad arguments");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dout, act, dx, n);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

__global__ void nonfinite_latch_kernel(const float* value, long long step, long long* latch) {
    const float v = *value;
    if (!(fabsf(v) <= 3.402823466e38f)) {   // NaN compares false, +-inf exceeds FLT_MAX
        latch[0] += 1;
        if (latch[1] == 0) latch[1] = step + 1;
    }
}

extern "C" int vince_nonfinite_latch(const float* value, int64_t step, int64_t* latch, void* stream) {
    VINCE_CHECK_ARG(value && latch, VINCE_E_ARG, "vince_nonfinite_latch: null pointer");
    hipLaunchKernelGGL(nonfinite_latch_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, value, (long long)step, (long long*)latch);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_colsum(const float* x, float* out, int32_t rows, int32_t cols, void* stream) {
    VINCE_CHECK_ARG(x && out && rows > 0 && cols > 0, VINCE_E_ARG, "vince_colsum: bad arguments");
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64, COLSUM_CHUNKS), dim3(256), 0, (hipStream_t)stream, x, out, rows, cols);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_queue_enqueue(float* queue, int64_t K, int64_t D, const float* items, int64_t n, int64_t* tail,
                                   int32_t* full, void* stream) {
    VINCE_CHECK_ARG(queue && items && tail && full && K > 0 && D > 0 && n >= 0, VINCE_E_ARG, "vince_queue_enqueue: bad arguments");
    VINCE_CHECK_ARG(*tail >= 0 && *tail <= K, VINCE_E_ARG, "vince_queue_enqueue: tail %lld outside [0, %lld]", (long long)*tail, (long long)K);
    // utils/storage_queue.py:31-49 -- integer index arithmetic restated exactly: a write that would cross the end fills
    // [tail, K), resets tail to 0, marks the queue full and recurses on the remainder (possibly several laps).
    int64_t t = *tail, src = 0;
    while (true) {
        if (t + n > K) {
            const int64_t num_start = K - t;
            if (num_start > 0)
                VINCE_CHECK_HIP(hipMemcpyAsync(queue + t * D, items + src * D, (size_t)num_start * D * sizeof(float),
                                               hipMemcpyDeviceToDevice, (hipStream_t)stream));
            t = 0;
            *full = 1;
            src += num_start;
            n -= num_start;
        } else {
            if (n > 0)
                VINCE_CHECK_HIP(hipMemcpyAsync(queue + t * D, items + src * D, (size_t)n * D * sizeof(float),
                                               hipMemcpyDeviceToDevice, (hipStream_t)stream));
            t += n;
            break;
        }
    }
    *tail = t;
    return VINCE_OK;
}

extern "C" int vince_ema_flat(float* key, const float* query, int64_t n, float momentum, void* stream) {
    VINCE_CHECK_ARG(key && query && n > 0, VINCE_E_ARG, "vince_ema_flat: bad arguments");
    VINCE_CHECK_ARG((((uintptr_t)key | (uintptr_t)query) & 15) == 0, VINCE_E_ALIGN, "vince_ema_flat: pointers must be 16-byte aligned");
    const float omm = (float)(1.0 - (double)momentum);
    hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, key, query, n, momentum, omm);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_sgd_flat(float* param, const float* grad, float* buf, int64_t n, float lr, float momentum,
                              float weight_decay, float grad_scale, void* stream) {
    VINCE_CHECK_ARG(param && grad && buf && n > 0, VINCE_E_ARG, "vince_sgd_flat: bad arguments");
    VINCE_CHECK_ARG((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)buf) & 15) == 0, VINCE_E_ALIGN,
                    "vince_sgd_flat: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, param, grad, buf, n, lr,
                       momentum, weight_decay, grad_scale);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

namespace {
__global__ __launch_bounds__(256) void zero_kernel(uint4* p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4(0, 0, 0, 0);
}
}  // namespace

namespace {
// Calibration kernel (bench.py `roofline.hbm_achievable`): a plain 16-byte-per-lane grid-stride copy, UNROLL loads in flight per
// thread -- the streaming rate an element-wise pass of this library can be held against on the box it runs on.
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void stream_copy_kernel(const f32x4_t* __restrict__ src, f32x4_t* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        f32x4_t v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
// Two other shapes of the same copy, tried once against the guide's 6.29 TB/s (MI355X_MICROARCH.md:35):
// SHAPE 1: 32 bytes per lane -- a lane moves two ADJACENT 16-byte pieces, a wavefront instruction pair covers 2 KB contiguous;
// SHAPE 2: block-contiguous -- a workgroup owns one contiguous segment of the buffer and walks it in 4 KB steps (no grid stride).
template <int SHAPE, bool NT>
__global__ __launch_bounds__(256) void stream_copy_shape_kernel(const f32x4_t* __restrict__ src, f32x4_t* __restrict__ dst, size_t n16) {
    auto ld = [&](size_t k) { return NT ? __builtin_nontemporal_load(src + k) : src[k]; };
    auto st = [&](size_t k, f32x4_t v) { if (NT) __builtin_nontemporal_store(v, dst + k); else dst[k] = v; };
    if constexpr (SHAPE == 1) {
        const size_t stride = (size_t)gridDim.x * 512;     // 16-byte pieces per grid pass
        for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2; i < n16; i += 2 * stride) {
            if (i + 1 >= n16) { st(i, ld(i)); break; }     // an odd piece count: the last 16 bytes travel alone
            const size_t j = i + stride;
            const bool two = j + 1 < n16;
            const f32x4_t a0 = ld(i), a1 = ld(i + 1);
            f32x4_t b0, b1;
            if (two) { b0 = ld(j); b1 = ld(j + 1); }
            st(i, a0); st(i + 1, a1);
            if (two) { st(j, b0); st(j + 1, b1); }
            else if (j < n16) st(j, ld(j));
        }
    } else {
        const size_t per = ((n16 + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
        const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n16 ? b0 + per : n16;
        for (size_t i = b0 + threadIdx.x; i < b1; i += 1024) {
            f32x4_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i + u * 256 < b1) v[u] = ld(i + u * 256);
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i + u * 256 < b1) st(i + u * 256, v[u]);
        }
    }
}
}  // namespace

namespace {
// 8 floats -> 8 bfloat16 per thread and trip (two 16-byte loads, one 16-byte store), block-contiguous like the streaming copy
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n8) {
    const size_t per = (n8 + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n8 ? lo + per : n8;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
        const uint4 a = *(const uint4*)(src + i * 8), b = *(const uint4*)(src + i * 8 + 4);
        *(uint4*)(dst + i * 8) = make_uint4(pack_bf16x2(__uint_as_float(a.x), __uint_as_float(a.y)), pack_bf16x2(__uint_as_float(a.z), __uint_as_float(a.w)),
                                            pack_bf16x2(__uint_as_float(b.x), __uint_as_float(b.y)), pack_bf16x2(__uint_as_float(b.z), __uint_as_float(b.w)));
    }
}
}  // namespace

// The bf16 shadow of an fp32 tensor nobody's epilogue can write on the side (the staged stem input, the stem pool's output): the mixed
// mode "x3f" (vince_trunk_set_shadow).
extern "C" int vince_cast_f32_to_bf16(const float* src, void* dst, size_t n, void* stream) {
    VINCE_CHECK_ARG(src && dst && n % 8 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0, VINCE_E_ARG,
                    "vince_cast_f32_to_bf16: n multiple of 8, 16-byte aligned pointers");
    if (!n) return VINCE_OK;
    const size_t n8 = n / 8;
    const unsigned blocks = (unsigned)(n8 < 2048 * 256 ? (n8 + 255) / 256 : 2048);
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n8);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_stream_copy(void* dst, const void* src, size_t bytes, int32_t blocks, int32_t nontemporal, void* stream) {
    VINCE_CHECK_ARG(dst && src && (((uintptr_t)dst | (uintptr_t)src) & 15) == 0 && (bytes & 15) == 0, VINCE_E_ALIGN,
                    "vince_stream_copy: 16-byte granularity");
    if (bytes == 0) return VINCE_OK;
    const size_t n16 = bytes / 16;
    if (blocks <= 0) blocks = 2048;
    // nontemporal: bit 0 = nt loads / stores; bits 1-2 = copy shape (0 the grid-stride copy above, 1 = 32 bytes per lane, 2 = block-contiguous)
    const int shape = (nontemporal >> 1) & 3;
    nontemporal &= 1;
    if (shape == 1 || shape == 2) {
        const dim3 g((unsigned)blocks), b(256);
        hipStream_t s = (hipStream_t)stream;
        if (shape == 1 && nontemporal) hipLaunchKernelGGL((stream_copy_shape_kernel<1, true>), g, b, 0, s, (const f32x4_t*)src, (f32x4_t*)dst, n16);
        else if (shape == 1) hipLaunchKernelGGL((stream_copy_shape_kernel<1, false>), g, b, 0, s, (const f32x4_t*)src, (f32x4_t*)dst, n16);
        else if (nontemporal) hipLaunchKernelGGL((stream_copy_shape_kernel<2, true>), g, b, 0, s, (const f32x4_t*)src, (f32x4_t*)dst, n16);
        else hipLaunchKernelGGL((stream_copy_shape_kernel<2, false>), g, b, 0, s, (const f32x4_t*)src, (f32x4_t*)dst, n16);
        VINCE_CHECK_LAUNCH();
        return VINCE_OK;
    }
    if (nontemporal)
        hipLaunchKernelGGL((stream_copy_kernel<4, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           (const f32x4_t*)src, (f32x4_t*)dst, n16);
    else
        hipLaunchKernelGGL((stream_copy_kernel<4, false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           (const f32x4_t*)src, (f32x4_t*)dst, n16);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

int vince_zero_async(void* ptr, size_t bytes, void* stream) {
    VINCE_CHECK_ARG(ptr && ((uintptr_t)ptr & 15) == 0 && (bytes & 15) == 0, VINCE_E_ALIGN, "vince_zero_async: 16-byte granularity");
    if (bytes == 0) return VINCE_OK;
    const size_t n16 = bytes / 16;
    const unsigned grid = (unsigned)((n16 + 255) / 256 < 2048 ? (n16 + 255) / 256 : 2048);
    hipLaunchKernelGGL(zero_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (uint4*)ptr, n16);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_input_nchw_to_rows(int dtype, const float* in, const int64_t* perm, void* out, int32_t N, int32_t C,
                                        int32_t H, int32_t W, int32_t Wp, int32_t left, void* stream) {
    DTYPE_OK("vince_input_nchw_to_rows");
    VINCE_CHECK_ARG(in && out && N > 0 && C >= 1 && C <= 4 && H > 0 && W > 0 && left >= 0 && Wp >= W + left, VINCE_E_ARG,
                    "vince_input_nchw_to_rows: bad arguments");
    const int64_t total = (int64_t)N * H * Wp;
    if (dtype == VINCE_F32)
        hipLaunchKernelGGL(input_to_rows_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, perm,
                           (float*)out, N, C, H, W, Wp, left);
    else
        hipLaunchKernelGGL(input_to_rows_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, perm,
                           (bf16_t*)out, N, C, H, W, Wp, left);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_jigsaw_nchw_to_rows(int dtype, const float* in, void* out, int32_t N, int32_t C, int32_t H, int32_t W,
                                         int32_t th, int32_t tw, int32_t Wp, int32_t left, void* stream) {
    DTYPE_OK("vince_jigsaw_nchw_to_rows");
    VINCE_CHECK_ARG(in && out && N > 0 && C >= 1 && C <= 4 && H > 0 && W > 0 && th > 0 && tw > 0 && left >= 0 &&
                    Wp >= tw + left, VINCE_E_ARG, "vince_jigsaw_nchw_to_rows: bad arguments");
    VINCE_CHECK_ARG(3 * th >= H && 3 * tw >= W, VINCE_E_SHAPE, "vince_jigsaw_nchw_to_rows: 3x3 tiles do not cover the image");
    const int64_t total = (int64_t)N * 9 * th * Wp;
    if (dtype == VINCE_F32)
        hipLaunchKernelGGL(jigsaw_to_rows_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in,
                           (float*)out, N, C, H, W, th, tw, Wp, left);
    else
        hipLaunchKernelGGL(jigsaw_to_rows_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in,
                           (bf16_t*)out, N, C, H, W, th, tw, Wp, left);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

namespace {
__global__ void fill_f32_kernel(float* p, int n, float v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
}  // namespace

int vince_fill_f32_async(float* ptr, int n, float value, void* stream) {
    hipLaunchKernelGGL(fill_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ptr, n, value);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_input_u8hwc_to_rows(int dtype, const uint8_t* in, const int64_t* perm, const int32_t* crop_yx,
                                         const uint8_t* flip, const float* mean255, const float* std255, void* out, int32_t N,
                                         int32_t Hs, int32_t Ws, int32_t H, int32_t W, int32_t Wp, int32_t left, void* stream) {
    DTYPE_OK("vince_input_u8hwc_to_rows");
    VINCE_CHECK_ARG(in && out && mean255 && std255 && N > 0 && H > 0 && W > 0 && Hs >= H && Ws >= W && left >= 0 &&
                    Wp >= W + left, VINCE_E_ARG, "vince_input_u8hwc_to_rows: bad arguments");
    // (mean255 / std255 are HOST pointers: three floats each, passed by value to the kernel)
    const int64_t total = (int64_t)N * H * Wp;
    if (dtype == VINCE_F32)
        hipLaunchKernelGGL(input_u8_to_rows_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, perm,
                           crop_yx, flip, mean255[0], mean255[1], mean255[2], std255[0], std255[1], std255[2], (float*)out, N,
                           Hs, Ws, H, W, Wp, left);
    else
        hipLaunchKernelGGL(input_u8_to_rows_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, perm,
                           crop_yx, flip, mean255[0], mean255[1], mean255[2], std255[0], std255[1], std255[2], (bf16_t*)out, N,
                           Hs, Ws, H, W, Wp, left);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}
