// GPU input stage, random part: the train-time image pipeline of utils/transforms.py:62-235 --
// RandomResizedCrop, ColorJitter, RandomGrayscale, RandomHorizontalFlip, ToTensor + Normalize, RandomGaussianBlur -- on
// uint8 frames that are already in HBM.  The reference runs these per sample on 40 PIL worker processes; here a batch is
// three small launches (resize H, resize V, colour chain) plus the layout kernel that feeds the stem.
//
// The uint8 stages reproduce Pillow's arithmetic bit for bit (Resample.c 22-bit fixed-point triangle filter with a uint8
// intermediate; Blend.c float32 blend with truncation; Convert.c luma / HSV), so this file is compiled with floating-point
// contraction OFF: an fma where the C library rounds twice changes the last bit of a coefficient.
// This is HBM/L2-bound byte work (77 MB of frames per 512-image step); no MFMA, no LDS staging needed.
#include "common.h"

#include <math.h>
#include <stdlib.h>

#pragma clang fp contract(off)

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;   // Resample.c

// One output coordinate of Pillow's precompute_coeffs for the BILINEAR filter (support 1, stretched by the scale when
// shrinking): window [xmin, xmin + xmax) of the input and the normalisation sum; weight() returns the 22-bit coefficient.
struct ResampleWindow {
    int xmin, xmax;
    double center, ss, ww;

    __device__ ResampleWindow(int in_size, int out_size, int xx) {
        const double scale = (double)in_size / (double)out_size;
        const double filterscale = scale < 1.0 ? 1.0 : scale;
        const double support = 1.0 * filterscale;
        center = 0.0 + ((double)xx + 0.5) * scale;
        ss = 1.0 / filterscale;
        xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        ww = 0.0;
        for (int x = 0; x < xmax; ++x) ww += tri(x);
    }
    __device__ double tri(int x) const {
        double t = ((double)(x + xmin) - center + 0.5) * ss;
        if (t < 0.0) t = -t;
        return t < 1.0 ? 1.0 - t : 0.0;
    }
    __device__ int weight(int x) const {
        double w = tri(x);
        if (ww != 0.0) w /= ww;
        return w < 0 ? (int)(-0.5 + w * (double)(1 << PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << PRECISION_BITS));
    }
};

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Coefficient tables: one entry per (image, axis, output coordinate) = {xmin, xmax, k[0..kmax)} ints, computed ONCE (the
// double-precision window arithmetic is ~100x the cost of the integer filter itself).  axis 0 = horizontal (window width ->
// W), axis 1 = vertical (window height -> H).
__global__ __launch_bounds__(256) void resample_table_kernel(const int32_t* __restrict__ box, int32_t* __restrict__ table, int N,
                                                             int H, int W, int kmax) {
    const int per_image = W + H;
    const int64_t total = (int64_t)N * per_image;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int n = (int)(idx / per_image), e = (int)(idx % per_image);
        const bool vertical = e >= W;
        const int in_size = vertical ? box[4 * n + 2] : box[4 * n + 3];
        const ResampleWindow win(in_size, vertical ? H : W, vertical ? e - W : e);
        int32_t* t = table + (size_t)idx * (kmax + 2);
        const int taps = win.xmax < kmax ? win.xmax : kmax;      // kmax bounds every window of a box inside the frame
        t[0] = win.xmin;
        t[1] = taps;
        for (int x = 0; x < kmax; ++x) t[2 + x] = x < taps ? win.weight(x) : 0;
    }
}

// horizontal pass: crop window of the source frame -> tmp[n][r][xx] for the window's rows r
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ frames, const int64_t* __restrict__ src_index,
                                                         const int32_t* __restrict__ box, const int32_t* __restrict__ table,
                                                         uint8_t* __restrict__ tmp, int N, int Hs, int Ws, int H, int W, int kmax) {
    const int64_t total = (int64_t)N * Hs * W;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int xx = (int)(idx % W);
        const int64_t q = idx / W;
        const int r = (int)(q % Hs);
        const int n = (int)(q / Hs);
        const int top = box[4 * n], left = box[4 * n + 1], ch = box[4 * n + 2];
        if (r >= ch) continue;
        const int64_t sn = src_index ? src_index[n] : n;
        const uint8_t* row = frames + ((size_t)sn * Hs + clampi(top + r, 0, Hs - 1)) * Ws * 3;
        const int32_t* t = table + ((size_t)n * (W + H) + xx) * (kmax + 2);
        const int xmin = t[0], taps = t[1];
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int x = 0; x < taps; ++x) {
            const int k = t[2 + x];
            const uint8_t* px = row + (size_t)clampi(left + xmin + x, 0, Ws - 1) * 3;
            s0 += (int)px[0] * k;
            s1 += (int)px[1] * k;
            s2 += (int)px[2] * k;
        }
        uint8_t* o = tmp + (((size_t)n * Hs + r) * W + xx) * 3;
        o[0] = clip8(s0);
        o[1] = clip8(s1);
        o[2] = clip8(s2);
    }
}

// vertical pass over the uint8 result of the horizontal one
__global__ __launch_bounds__(256) void resample_v_kernel(const uint8_t* __restrict__ tmp, const int32_t* __restrict__ table,
                                                         uint8_t* __restrict__ out, int N, int Hs, int H, int W, int kmax) {
    const int64_t total = (int64_t)N * H * W;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int xx = (int)(idx % W);
        const int64_t q = idx / W;
        const int yy = (int)(q % H);
        const int n = (int)(q / H);
        const int32_t* t = table + ((size_t)n * (W + H) + W + yy) * (kmax + 2);
        const int ymin = t[0], taps = t[1];
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int y = 0; y < taps; ++y) {
            const int k = t[2 + y];
            const uint8_t* px = tmp + (((size_t)n * Hs + clampi(ymin + y, 0, Hs - 1)) * W + xx) * 3;
            s0 += (int)px[0] * k;
            s1 += (int)px[1] * k;
            s2 += (int)px[2] * k;
        }
        uint8_t* o = out + (size_t)idx * 3;
        o[0] = clip8(s0);
        o[1] = clip8(s1);
        o[2] = clip8(s2);
    }
}

// ------------------------------------------------------------------------------------------------ colour chain
struct Rgb {
    int r, g, b;
};

__device__ __forceinline__ int luma(const Rgb& p) {   // Convert.c rgb2l
    return (p.r * 19595 + p.g * 38470 + p.b * 7471 + 0x8000) >> 16;
}

// Blend.c: float32 in1 + alpha * (in2 - in1); the interpolating branch truncates, the extrapolating one clips first
__device__ __forceinline__ int blend1(int deg, int v, float alpha, bool inside) {
    const float t = (float)deg + alpha * (float)(v - deg);
    if (inside) return (int)t & 0xFF;
    return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}

__device__ __forceinline__ Rgb blend(const Rgb& deg, const Rgb& p, float alpha) {
    const bool inside = alpha >= 0.f && alpha <= 1.f;
    return Rgb{blend1(deg.r, p.r, alpha, inside), blend1(deg.g, p.g, alpha, inside), blend1(deg.b, p.b, alpha, inside)};
}

// Convert.c rgb2hsv_row / hsv2rgb (colorsys in a float / double mix) around the uint8 wrap-around shift of the H plane
__device__ Rgb hue_shift(const Rgb& p, int shift) {
    const int maxc = max(p.r, max(p.g, p.b)), minc = min(p.r, min(p.g, p.b));
    int uh = 0, us = 0;
    const int uv = maxc;
    if (minc != maxc) {
        const float cr = (float)(maxc - minc);
        const float s = cr / (float)maxc;
        const float rc = (float)(maxc - p.r) / cr, gc = (float)(maxc - p.g) / cr, bc = (float)(maxc - p.b) / cr;
        float h;
        if (p.r == maxc) h = bc - gc;
        else if (p.g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
        else h = (float)(4.0 + (double)gc - (double)rc);
        const double hh = (double)h / 6.0 + 1.0;
        h = (float)(hh - floor(hh));                                   // fmod(x, 1.0), x > 0
        uh = clampi((int)((double)h * 255.0), 0, 255);
        us = clampi((int)((double)s * 255.0), 0, 255);
    }
    uh = (uh + shift) & 0xFF;
    if (us == 0) return Rgb{uv, uv, uv};
    const double hf = (double)(float)uh * 6.0 / 255.0;
    const int i = (int)floor(hf);
    const float f = (float)(hf - (double)(float)i);
    const float fs = (float)((double)(float)us / 255.0);
    const double v = (double)(float)uv;
    const int pp = clampi((int)round(v * (1.0 - (double)fs)), 0, 255);
    const int qq = clampi((int)round(v * (1.0 - (double)fs * (double)f)), 0, 255);
    const int tt = clampi((int)round(v * (1.0 - (double)fs * (1.0 - (double)f))), 0, 255);
    switch (i % 6) {
        case 0: return Rgb{uv, tt, pp};
        case 1: return Rgb{qq, uv, pp};
        case 2: return Rgb{pp, uv, tt};
        case 3: return Rgb{pp, qq, uv};
        case 4: return Rgb{tt, pp, uv};
        default: return Rgb{uv, pp, qq};
    }
}

constexpr int COLOR_THREADS = 1024;

// One workgroup per image; every thread owns the pixels tid, tid + 1024, ... through the whole chain, so the only
// cross-thread step is the luma mean of ImageEnhance.Contrast.
__global__ __launch_bounds__(COLOR_THREADS) void color_chain_kernel(uint8_t* __restrict__ img, const int32_t* __restrict__ op,
                                                                   const float* __restrict__ factor, int max_ops, int HW) {
    __shared__ unsigned long long wave_sum[COLOR_THREADS / 64];
    __shared__ int mean_s;
    const int n = blockIdx.x;
    uint8_t* base = img + (size_t)n * HW * 3;
    for (int o = 0; o < max_ops; ++o) {
        const int code = op[n * max_ops + o];
        const float f = factor[n * max_ops + o];
        if (code < 0) continue;                       // uniform per workgroup
        int mean = 0;
        if (code == 1) {
            unsigned long long s = 0;
            for (int p = threadIdx.x; p < HW; p += COLOR_THREADS) {
                const uint8_t* px = base + (size_t)p * 3;
                s += (unsigned long long)luma(Rgb{px[0], px[1], px[2]});
            }
            for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
            __syncthreads();                          // the previous step's readers of mean_s / wave_sum are done
            if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = s;
            __syncthreads();
            if (threadIdx.x == 0) {
                unsigned long long t = 0;
                for (int w = 0; w < COLOR_THREADS / 64; ++w) t += wave_sum[w];
                mean_s = (int)((double)t / (double)HW + 0.5);       // int(ImageStat.mean + 0.5)
            }
            __syncthreads();
            mean = mean_s;
        }
        for (int p = threadIdx.x; p < HW; p += COLOR_THREADS) {
            uint8_t* px = base + (size_t)p * 3;
            const Rgb v{px[0], px[1], px[2]};
            Rgb r;
            if (code == 0) r = blend(Rgb{0, 0, 0}, v, f);
            else if (code == 1) r = blend(Rgb{mean, mean, mean}, v, f);
            else if (code == 2) { const int l = luma(v); r = blend(Rgb{l, l, l}, v, f); }
            else if (code == 3) r = hue_shift(v, (int)f & 0xFF);
            else if (code == 4) { const int l = luma(v); r = Rgb{l, l, l}; }
            else r = v;                                   // unknown step code: leave the pixel alone
            px[0] = (uint8_t)r.r;
            px[1] = (uint8_t)r.g;
            px[2] = (uint8_t)r.b;
        }
    }
}

// ------------------------------------------------------------------------------------------------ tensor side
template <typename T>
__device__ __forceinline__ void store_px4(T* out, size_t pix, const float (&f)[4]) {
    if constexpr (sizeof(T) == 4) {
        *(float4*)((float*)out + pix * 4) = make_float4(f[0], f[1], f[2], f[3]);
    } else {
        *(uint2*)((bf16_t*)out + pix * 4) = make_uint2(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]));
    }
}

struct Norm {
    float m0, m1, m2, s0, s1, s2;
};

// flip + ToTensor/Normalize + the H-direction half of RandomGaussianBlur (util_functions.py:124-126, zero padding of the
// NORMALISED tensor) -> float tmp [N][H][W][4]
__global__ __launch_bounds__(256) void blur_v_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ flip,
                                                     const float* __restrict__ kernels, const uint8_t* __restrict__ do_blur,
                                                     int ks, Norm nm, float* __restrict__ tmp, int N, int H, int W) {
    const int64_t total = (int64_t)N * H * W;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int x = (int)(idx % W);
        const int64_t q = idx / W;
        const int y = (int)(q % H);
        const int n = (int)(q / H);
        const int xs = (flip && flip[n]) ? (W - 1 - x) : x;
        const uint8_t* col = img + ((size_t)n * H * W + xs) * 3;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        if (do_blur && do_blur[n]) {
            const float* k = kernels + (size_t)n * ks;
            for (int j = 0; j < ks; ++j) {
                const int yy = y + j - ks / 2;
                if (yy < 0 || yy >= H) continue;
                const uint8_t* px = col + (size_t)yy * W * 3;
                a0 += k[j] * __fdiv_rn((float)px[0] - nm.m0, nm.s0);
                a1 += k[j] * __fdiv_rn((float)px[1] - nm.m1, nm.s1);
                a2 += k[j] * __fdiv_rn((float)px[2] - nm.m2, nm.s2);
            }
        } else {
            const uint8_t* px = col + (size_t)y * W * 3;
            a0 = __fdiv_rn((float)px[0] - nm.m0, nm.s0);
            a1 = __fdiv_rn((float)px[1] - nm.m1, nm.s1);
            a2 = __fdiv_rn((float)px[2] - nm.m2, nm.s2);
        }
        *(float4*)(tmp + (size_t)idx * 4) = make_float4(a0, a1, a2, 0.f);
    }
}

// W-direction half (util_functions.py:127-129) + the packed stem layout (zero margins, zero 4th channel)
template <typename T>
__global__ __launch_bounds__(256) void blur_h_rows_kernel(const float* __restrict__ tmp, const float* __restrict__ kernels,
                                                          const uint8_t* __restrict__ do_blur, int ks, T* __restrict__ out,
                                                          int N, int H, int W, int Wp, int left) {
    const int64_t total = (int64_t)N * H * Wp;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int wp = (int)(idx % Wp);
        const int64_t r = idx / Wp;            // n * H + y
        const int n = (int)(r / H);
        const int w = wp - left;
        float f[4] = {0.f, 0.f, 0.f, 0.f};
        if (w >= 0 && w < W) {
            const float* row = tmp + (size_t)r * W * 4;
            if (do_blur && do_blur[n]) {
                const float* k = kernels + (size_t)n * ks;
                for (int j = 0; j < ks; ++j) {
                    const int xx = w + j - ks / 2;
                    if (xx < 0 || xx >= W) continue;
                    const float4 v = *(const float4*)(row + (size_t)xx * 4);
                    f[0] += k[j] * v.x;
                    f[1] += k[j] * v.y;
                    f[2] += k[j] * v.z;
                }
            } else {
                const float4 v = *(const float4*)(row + (size_t)w * 4);
                f[0] = v.x; f[1] = v.y; f[2] = v.z;
            }
        }
        store_px4<T>(out, (size_t)idx, f);
    }
}

// Fused version of the two passes above for kernels that fit the LDS: one workgroup owns a BLUR_TH x BLUR_TW tile of the
// output rows, stages the normalised pixels of the tile + halo in LDS (A), runs the H-direction pass into a second LDS
// plane (B) and the W-direction pass straight into the stem layout -- no float intermediate in HBM.  Both passes are
// register-blocked 4 outputs per thread along the filter direction: every staged value is read once per 4 outputs and meets
// 12 independent fma chains; the taps sit in LDS with 3 zeros on either side so that the blocked loops carry no predicates
// (a zero tap adds exactly 0).  VALU-bound: 2 x 23 taps x 3 channels per output.
constexpr int BLUR_TH = 16, BLUR_TW = 64, BLUR_THREADS = 256, BLUR_PAD = 3;

template <typename T>
__global__ __launch_bounds__(BLUR_THREADS) void blur_fused_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ flip,
                                                                  const float* __restrict__ kernels,
                                                                  const uint8_t* __restrict__ do_blur, int ks, Norm nm,
                                                                  T* __restrict__ out, int H, int W, int Wp, int left) {
    extern __shared__ float4 smem[];
    const int n = blockIdx.z, y0 = blockIdx.y * BLUR_TH, p0 = blockIdx.x * BLUR_TW;   // p: padded column index
    const int half = ks / 2, aw = BLUR_TW + ks - 1, ah = BLUR_TH + ks - 1;
    const bool blurred = do_blur && do_blur[n];
    const bool flipped = flip && flip[n];
    const uint8_t* base = img + (size_t)n * H * W * 3;
    if (!blurred) {
        for (int i = threadIdx.x; i < BLUR_TH * BLUR_TW; i += BLUR_THREADS) {
            const int y = y0 + i / BLUR_TW, p = p0 + i % BLUR_TW, w = p - left;
            if (y >= H || p >= Wp) continue;
            float f[4] = {0.f, 0.f, 0.f, 0.f};
            if (w >= 0 && w < W) {
                const uint8_t* px = base + ((size_t)y * W + (flipped ? W - 1 - w : w)) * 3;
                f[0] = __fdiv_rn((float)px[0] - nm.m0, nm.s0);
                f[1] = __fdiv_rn((float)px[1] - nm.m1, nm.s1);
                f[2] = __fdiv_rn((float)px[2] - nm.m2, nm.s2);
            }
            store_px4<T>(out, ((size_t)n * H + y) * Wp + p, f);
        }
        return;
    }
    float4* A = smem;                        // [ah][aw]  normalised input, rows y0 - half .., columns (p0 - left) - half ..
    float4* B = smem + (size_t)ah * aw;      // [BLUR_TH][aw + BLUR_PAD]  after the H-direction pass (+ zero columns)
    float* kp = (float*)(B + (size_t)BLUR_TH * (aw + BLUR_PAD));   // [BLUR_PAD + ks + BLUR_PAD + 1] zero-padded taps
    const int bw = aw + BLUR_PAD;
    for (int i = threadIdx.x; i < ks + 2 * BLUR_PAD + 1; i += BLUR_THREADS) {
        const int t = i - BLUR_PAD;
        kp[i] = (t >= 0 && t < ks) ? kernels[(size_t)n * ks + t] : 0.f;
    }
    // staging: 8 pixels per thread per round, all 24 byte loads issued (clamped addresses, no branches) before the first
    // conversion -- with 4 waves per workgroup the load latency is otherwise paid once per pixel
    for (int i0 = threadIdx.x; i0 < ah * aw; i0 += BLUR_THREADS * 8) {
        uint32_t raw[8][3];
        bool ok[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * BLUR_THREADS;
            const int y = y0 - half + i / aw, w = p0 - left - half + i % aw;
            ok[u] = i < ah * aw && y >= 0 && y < H && w >= 0 && w < W;
            const int yc = clampi(y, 0, H - 1), wc = clampi(w, 0, W - 1);
            const uint8_t* px = base + ((size_t)yc * W + (flipped ? W - 1 - wc : wc)) * 3;
            raw[u][0] = px[0];
            raw[u][1] = px[1];
            raw[u][2] = px[2];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * BLUR_THREADS;
            if (i >= ah * aw) break;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok[u]) {
                v.x = __fdiv_rn((float)raw[u][0] - nm.m0, nm.s0);
                v.y = __fdiv_rn((float)raw[u][1] - nm.m1, nm.s1);
                v.z = __fdiv_rn((float)raw[u][2] - nm.m2, nm.s2);
            }
            A[i] = v;
        }
    }
    for (int i = threadIdx.x; i < BLUR_TH * BLUR_PAD; i += BLUR_THREADS)     // the W pass's blocked loop reads 3 columns past aw
        B[(size_t)(i / BLUR_PAD) * bw + aw + i % BLUR_PAD] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    // H-direction pass (util_functions.py:124-126): B[y][x] = sum_j k[j] * A[y + j][x], outputs y = yg .. yg + 3 per thread.
    // Tap of output o at step j is k[j - o] = kp[BLUR_PAD + j - o].
    for (int i = threadIdx.x; i < (BLUR_TH / 4) * aw; i += BLUR_THREADS) {
        const int yg = (i / aw) * 4, x = i % aw;
        float acc[4][3] = {};
        float k0 = kp[BLUR_PAD], k1 = kp[BLUR_PAD - 1], k2 = kp[BLUR_PAD - 2], k3 = kp[BLUR_PAD - 3];
        const float4* col = A + (size_t)yg * aw + x;
        for (int j = 0; j < ks + 3; ++j) {
            const float4 v = col[(size_t)j * aw];
            acc[0][0] = fmaf(k0, v.x, acc[0][0]); acc[0][1] = fmaf(k0, v.y, acc[0][1]); acc[0][2] = fmaf(k0, v.z, acc[0][2]);
            acc[1][0] = fmaf(k1, v.x, acc[1][0]); acc[1][1] = fmaf(k1, v.y, acc[1][1]); acc[1][2] = fmaf(k1, v.z, acc[1][2]);
            acc[2][0] = fmaf(k2, v.x, acc[2][0]); acc[2][1] = fmaf(k2, v.y, acc[2][1]); acc[2][2] = fmaf(k2, v.z, acc[2][2]);
            acc[3][0] = fmaf(k3, v.x, acc[3][0]); acc[3][1] = fmaf(k3, v.y, acc[3][1]); acc[3][2] = fmaf(k3, v.z, acc[3][2]);
            k3 = k2; k2 = k1; k1 = k0; k0 = kp[BLUR_PAD + j + 1];
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) B[(size_t)(yg + o) * bw + x] = make_float4(acc[o][0], acc[o][1], acc[o][2], 0.f);
    }
    __syncthreads();
    // W-direction pass (util_functions.py:127-129) + layout, outputs tx = xg .. xg + 3 per thread
    for (int i = threadIdx.x; i < BLUR_TH * (BLUR_TW / 4); i += BLUR_THREADS) {
        const int ty = i / (BLUR_TW / 4), xg = (i % (BLUR_TW / 4)) * 4;
        const int y = y0 + ty;
        if (y >= H) continue;
        float acc[4][3] = {};
        float k0 = kp[BLUR_PAD], k1 = kp[BLUR_PAD - 1], k2 = kp[BLUR_PAD - 2], k3 = kp[BLUR_PAD - 3];
        const float4* row = B + (size_t)ty * bw + xg;
        for (int j = 0; j < ks + 3; ++j) {
            const float4 v = row[j];
            acc[0][0] = fmaf(k0, v.x, acc[0][0]); acc[0][1] = fmaf(k0, v.y, acc[0][1]); acc[0][2] = fmaf(k0, v.z, acc[0][2]);
            acc[1][0] = fmaf(k1, v.x, acc[1][0]); acc[1][1] = fmaf(k1, v.y, acc[1][1]); acc[1][2] = fmaf(k1, v.z, acc[1][2]);
            acc[2][0] = fmaf(k2, v.x, acc[2][0]); acc[2][1] = fmaf(k2, v.y, acc[2][1]); acc[2][2] = fmaf(k2, v.z, acc[2][2]);
            acc[3][0] = fmaf(k3, v.x, acc[3][0]); acc[3][1] = fmaf(k3, v.y, acc[3][1]); acc[3][2] = fmaf(k3, v.z, acc[3][2]);
            k3 = k2; k2 = k1; k1 = k0; k0 = kp[BLUR_PAD + j + 1];
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int p = p0 + xg + o, w = p - left;
            if (p >= Wp) continue;
            float f[4] = {0.f, 0.f, 0.f, 0.f};
            if (w >= 0 && w < W) { f[0] = acc[o][0]; f[1] = acc[o][1]; f[2] = acc[o][2]; }
            store_px4<T>(out, ((size_t)n * H + y) * Wp + p, f);
        }
    }
}

inline size_t blur_fused_lds(int ks) {
    const size_t aw = BLUR_TW + ks - 1, ah = BLUR_TH + ks - 1;
    return (ah * aw + (size_t)BLUR_TH * (aw + BLUR_PAD)) * sizeof(float4) + (size_t)(ks + 2 * BLUR_PAD + 1) * sizeof(float);
}

inline int grid_for(int64_t total_threads) {
    int64_t b = (total_threads + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int vince_aug_resample_kmax(int32_t Hs, int32_t Ws, int32_t H, int32_t W) {
    // Resample.c: ksize = ceil(support) * 2 + 1, support = max(scale, 1); a window inside the frame has scale <= Hs/H, Ws/W
    const double sh = (double)Hs / (double)(H > 0 ? H : 1), sw = (double)Ws / (double)(W > 0 ? W : 1);
    double sc = sh > sw ? sh : sw;
    if (sc < 1.0) sc = 1.0;
    return (int)ceil(sc) * 2 + 1;
}

extern "C" int64_t vince_aug_resample_table_ints(int32_t N, int32_t Hs, int32_t Ws, int32_t H, int32_t W) {
    return (int64_t)N * (W + H) * (vince_aug_resample_kmax(Hs, Ws, H, W) + 2);
}

extern "C" int vince_aug_resized_crop_u8(const uint8_t* frames, const int64_t* src_index, const int32_t* box, int32_t* table,
                                         uint8_t* tmp, uint8_t* out, int32_t N, int32_t Hs, int32_t Ws, int32_t H, int32_t W,
                                         void* stream) {
    VINCE_CHECK_ARG(frames && box && table && tmp && out, VINCE_E_ARG, "aug_resized_crop_u8: null pointer");
    VINCE_CHECK_ARG(N > 0 && Hs > 0 && Ws > 0 && H > 0 && W > 0, VINCE_E_SHAPE, "aug_resized_crop_u8: bad shape N=%d %dx%d -> %dx%d",
                    N, Hs, Ws, H, W);
    const int kmax = vince_aug_resample_kmax(Hs, Ws, H, W);
    hipLaunchKernelGGL(resample_table_kernel, dim3(grid_for((int64_t)N * (W + H))), dim3(256), 0, (hipStream_t)stream, box, table, N,
                       H, W, kmax);
    VINCE_CHECK_LAUNCH();
    hipLaunchKernelGGL(resample_h_kernel, dim3(grid_for((int64_t)N * Hs * W)), dim3(256), 0, (hipStream_t)stream, frames,
                       src_index, box, table, tmp, N, Hs, Ws, H, W, kmax);
    VINCE_CHECK_LAUNCH();
    hipLaunchKernelGGL(resample_v_kernel, dim3(grid_for((int64_t)N * H * W)), dim3(256), 0, (hipStream_t)stream, tmp, table, out, N,
                       Hs, H, W, kmax);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_aug_color_u8(uint8_t* img, const int32_t* op, const float* factor, int32_t max_ops, int32_t N, int32_t H,
                                  int32_t W, void* stream) {
    VINCE_CHECK_ARG(img && op && factor, VINCE_E_ARG, "aug_color_u8: null pointer");
    VINCE_CHECK_ARG(N > 0 && H > 0 && W > 0 && max_ops > 0 && (int64_t)H * W < (1ll << 31), VINCE_E_SHAPE,
                    "aug_color_u8: bad shape N=%d %dx%d ops=%d", N, H, W, max_ops);
    hipLaunchKernelGGL(color_chain_kernel, dim3(N), dim3(COLOR_THREADS), 0, (hipStream_t)stream, img, op, factor, max_ops, H * W);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

extern "C" int vince_aug_blur_to_rows(int dtype, const uint8_t* img, const uint8_t* flip, const float* kernels,
                                      const uint8_t* do_blur, int32_t ks, const float* mean255, const float* std255, float* tmp,
                                      void* out, int32_t N, int32_t H, int32_t W, int32_t Wp, int32_t left, void* stream) {
    VINCE_CHECK_ARG(dtype == VINCE_F32 || dtype == VINCE_BF16, VINCE_E_DTYPE, "aug_blur_to_rows: bad dtype %d", dtype);
    VINCE_CHECK_ARG(img && mean255 && std255 && tmp && out, VINCE_E_ARG, "aug_blur_to_rows: null pointer");
    VINCE_CHECK_ARG((do_blur == nullptr) || (kernels && ks > 0 && (ks & 1)), VINCE_E_ARG,
                    "aug_blur_to_rows: blur needs an odd kernel size and a [N][ks] kernel table (ks=%d)", ks);
    VINCE_CHECK_ARG(N > 0 && H > 0 && W > 0 && left >= 0 && Wp >= left + W, VINCE_E_SHAPE,
                    "aug_blur_to_rows: bad shape N=%d %dx%d Wp=%d left=%d", N, H, W, Wp, left);
    VINCE_CHECK_ARG((((uintptr_t)out | (uintptr_t)tmp) & 15) == 0, VINCE_E_ALIGN, "aug_blur_to_rows: tmp / out must be 16-byte aligned");
    const Norm nm{mean255[0], mean255[1], mean255[2], std255[0], std255[1], std255[2]};
    static const bool fused_env = (VINCE_MEASURE_KNOB("blur_fused", 1) != 0);
    const size_t lds = blur_fused_lds(do_blur ? ks : 1);
    if (fused_env && lds <= 160 * 1024) {
        const dim3 grid((Wp + BLUR_TW - 1) / BLUR_TW, (H + BLUR_TH - 1) / BLUR_TH, N);
        if (dtype == VINCE_F32) {
            VINCE_CHECK_HIP(hipFuncSetAttribute((const void*)blur_fused_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(blur_fused_kernel<float>, grid, dim3(BLUR_THREADS), lds, (hipStream_t)stream, img, flip, kernels, do_blur,
                               do_blur ? ks : 1, nm, (float*)out, H, W, Wp, left);
        } else {
            VINCE_CHECK_HIP(hipFuncSetAttribute((const void*)blur_fused_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(blur_fused_kernel<bf16_t>, grid, dim3(BLUR_THREADS), lds, (hipStream_t)stream, img, flip, kernels,
                               do_blur, do_blur ? ks : 1, nm, (bf16_t*)out, H, W, Wp, left);
        }
        VINCE_CHECK_LAUNCH();
        return VINCE_OK;
    }
    hipLaunchKernelGGL(blur_v_kernel, dim3(grid_for((int64_t)N * H * W)), dim3(256), 0, (hipStream_t)stream, img, flip, kernels,
                       do_blur, ks, nm, tmp, N, H, W);
    VINCE_CHECK_LAUNCH();
    const int64_t total = (int64_t)N * H * Wp;
    if (dtype == VINCE_F32)
        hipLaunchKernelGGL(blur_h_rows_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, tmp, kernels, do_blur,
                           ks, (float*)out, N, H, W, Wp, left);
    else
        hipLaunchKernelGGL(blur_h_rows_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, tmp, kernels,
                           do_blur, ks, (bf16_t*)out, N, H, W, Wp, left);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}
