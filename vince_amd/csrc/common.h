// Shared device/host helpers for the VINCE MI355X (gfx950) kernels.
// wave = 64 lanes, 256 CUs in 8 XCDs, MFMA 32x32 tiles, 16-byte vector memory ops everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/vince_hip.h"

// Every kernel launch of the library is counted (one relaxed atomic increment on the host): vince_launch_count() is how bench.py states
// `launches_per_step` without a profiler attached (a launch budget).  hipLaunchKernelGGL is the only launch form used.
void vince_note_launch();
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                  \
    do {                                                                                                   \
        vince_note_launch();                                                                               \
        kernelName<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__);                 \
    } while (0)

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define VINCE_NUM_XCD 8

// ---------------------------------------------------------------------------------------------
// error plumbing: every export returns int, never throws; message kept thread-local
// ---------------------------------------------------------------------------------------------
void vince_set_error(const char* fmt, ...);
// Cross-check switches of the product library, all inside ONE environment variable:  VINCE_KNOBS="name=value,name=value".
// vince_knob: parsed at the call (callers cache it in a static); vince_knob_live: for switches a test flips inside one process.
// Documented names: DESIGN.md section 9.  Everything else that used to be an environment knob is a constant of the product build
// and an environment variable VINCE_<NAME> only in the measurement build (python -m vince_amd.build --measure).
long vince_knob(const char* name, long dflt);
inline long vince_knob_live(const char* name, long dflt) { return vince_knob(name, dflt); }
#ifdef VINCE_MEASURE
long vince_measure_knob(const char* name, long dflt);
#define VINCE_MEASURE_KNOB(name, dflt) vince_measure_knob(name, dflt)
#else
#define VINCE_MEASURE_KNOB(name, dflt) (dflt)
#endif
bool vince_profile_enabled();
int vince_side_stream_budget();   // vince_set_side_streams(): 2 = wgrad + downsample streams, 1 = wgrad only, 0 = none
void vince_profile_begin_launch(int tag, double work, void* stream, void** token);
void vince_profile_end_launch(void* token, void* stream);
void vince_profile_set_tag(void* token, int tag);
// Zero `bytes` (multiple of 16, 16-byte aligned pointer) on `stream` with a full-width grid; the runtime's fill kernel runs
// a few-MB clear at a fraction of HBM speed.  Defined in misc.hip.
int vince_zero_async(void* ptr, size_t bytes, void* stream);
int vince_fill_f32_async(float* ptr, int n, float value, void* stream);   // small constant vectors (misc.hip)
void vince_profile_set_dims(void* token, int a, int b, int c, int d, int e, int f);

// Profile tags (vince_profile_enable: one hipEvent pair per launch, streams serialised by the caller).  `work` is algorithmic FLOPs for
// the matrix kernels and algorithmic BYTES for the streaming ones; bench.py holds the same table with the roof of each family.
enum {
    VINCE_TAG_CONV_IGEMM0 = 0,      // 0..15: conv_igemm [dtype][tile shape][fwd|bwd epilogue]
    VINCE_TAG_WGRAD_F32 = 16, VINCE_TAG_WGRAD_BF16 = 17,
    VINCE_TAG_M8_FWD = 18, VINCE_TAG_M8_BWD = 19,
    VINCE_TAG_XJOIN = 20,           // conv_xjoin: expand + BatchNorm + join (bytes)
    VINCE_TAG_XSTATS = 21,          // conv_xjoin: expand + statistics (bytes)
    VINCE_TAG_XDGRAD = 22,          // conv_xjoin: block-input gradient (bytes)
    VINCE_TAG_STRIP = 23,           // conv3x3_strip (FLOPs)
    VINCE_TAG_BN_APPLY = 24,        // bn_apply / bn_train_apply (bytes)
    VINCE_TAG_BN_BWD_APPLY = 25,    // bn_bwd_apply (bytes)
    VINCE_TAG_BN_BWD_REDUCE = 26,   // bn_bwd_reduce (bytes)
    VINCE_TAG_STEM_POOL = 27,       // stem_pool_fwd (bytes)
    VINCE_TAG_STEM_BWD = 28,        // stem_bwd_reduce / stem_bwd_apply / stem_pool_bwd (bytes)
    VINCE_TAG_COUNT = 29
};
struct VinceProfScope {   // brackets the launches an entry point enqueues between construction and return
    void* tok = nullptr;
    void* stream;
    VinceProfScope(int tag, double work, void* s) : stream(s) {
        if (vince_profile_enabled()) vince_profile_begin_launch(tag, work, s, &tok);
    }
    ~VinceProfScope() {
        if (tok) vince_profile_end_launch(tok, stream);
    }
};

#define VINCE_CHECK_ARG(cond, code, ...)   \
    do {                                   \
        if (!(cond)) {                     \
            vince_set_error(__VA_ARGS__);  \
            return (code);                 \
        }                                  \
    } while (0)

#define VINCE_CHECK_HIP(expr)                                                                   \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            vince_set_error("HIP error %s at %s:%d", hipGetErrorString(_e), __FILE__, __LINE__); \
            return VINCE_E_HIP;                                                                 \
        }                                                                                       \
    } while (0)

#define VINCE_CHECK_LAUNCH() VINCE_CHECK_HIP(hipGetLastError())

// ---------------------------------------------------------------------------------------------
// bf16 <-> f32 (round-to-nearest-even), element traits
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline float bf16_to_f32(bf16_t h) {
    union { uint32_t u; float f; } v;
    v.u = ((uint32_t)h) << 16;
    return v.f;
}
__host__ __device__ inline bf16_t f32_to_bf16(float f) {
    union { uint32_t u; float f; } v;
    v.f = f;
    uint32_t u = v.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
// two floats -> two bf16 in one dword: v_cvt_pk_bf16_f32 (round to nearest even, the same bits as f32_to_bf16 for every finite input
// and infinity; a NaN stays a quiet NaN).  One instruction where the integer sequence takes ~10: the store side of every bf16
// epilogue (64 values per lane and 128 x 128 tile) is VALU work that the matrix pipe does not hide.
__device__ inline uint32_t pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float pk_f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 pk_bf16x2_t;
    const pk_f32x2_t v = {lo, hi};
    const pk_bf16x2_t h = __builtin_convertvector(v, pk_bf16x2_t);
    uint32_t u;
    __builtin_memcpy(&u, &h, 4);
    return u;
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int CH = 4;  // elements per 16-byte chunk
    static constexpr int LOG2_CH = 2;
};
template <> struct Elem<bf16_t> {
    static constexpr int CH = 8;
    static constexpr int LOG2_CH = 3;
};

// A 16-byte chunk viewed as CH elements of T, converted to/from float.
template <typename T> struct Chunk;
template <> struct Chunk<float> {
    __device__ static inline void unpack(const uint4& v, float* f) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
        f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    __device__ static inline uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};
template <> struct Chunk<bf16_t> {
    __device__ static inline void unpack(const uint4& v, float* f) {
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
        f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
        f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
    }
    __device__ static inline uint4 pack(const float* f) {
        return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
    }
};

// ---------------------------------------------------------------------------------------------
// Split-half products (VINCE_F32X3H / VINCE_F32X3B): tensors stay fp32 in HBM and LDS; inside the MFMA loops every operand
// element x is split into hi = half(x) and lo = half(x - hi) and the product runs as hi*hi + hi*lo + lo*hi on the half-precision
// matrix pipe (3 x v_mfma_f32_32x32x16 = 96 cycles per 32x32x16 block against 512 for 8 x v_mfma_f32_32x32x2_f32), fp32 accumulate.
//   x3h_t: IEEE half halves (11 + 11 significand bits: products to ~2^-22, the forward's precision; both operands are scaled by a
//          power of two before the split -- weights by 2^X3_WSHIFT, activations by 2^X3_XSHIFT -- so that their lo halves stay
//          normal half numbers over the ranges a BatchNorm network produces; the accumulators are scaled back once);
//   x3b_t: bfloat16 halves (8 + 8 bits, fp32's exponent range: the gradient launches, whose operands span many decades).
// Both are tags over float: same chunk / element traits, same layouts, same epilogues.
// ---------------------------------------------------------------------------------------------
//   x1b_t: the bfloat16 HI halves only (VINCE_F32X1B: one MFMA per block, 2^-9 per operand -- the gradient launches of the mixed mode
//          VINCE_F32X3F).  Same kernels: the lo halves of x3_split / the lo chunks of the weight cache are simply never used.
struct x3h_t { float v; };
struct x3b_t { float v; };
struct x1b_t { float v; };
template <> struct Elem<x3h_t> : Elem<float> {};
template <> struct Elem<x3b_t> : Elem<float> {};
template <> struct Elem<x1b_t> : Elem<float> {};
template <> struct Chunk<x3h_t> : Chunk<float> {};
template <> struct Chunk<x3b_t> : Chunk<float> {};
template <> struct Chunk<x1b_t> : Chunk<float> {};
template <typename T> struct X3 { static constexpr bool on = false, half = false, single = false; };
template <> struct X3<x3h_t> { static constexpr bool on = true, half = true, single = false; };
template <> struct X3<x3b_t> { static constexpr bool on = true, half = false, single = false; };
template <> struct X3<x1b_t> { static constexpr bool on = true, half = false, single = true; };
constexpr int X3_WSHIFT = 8;   // weights:     |w| < 2^(16 - 8) keeps hi finite; lo is a normal half number down to |w| ~ 5e-4
constexpr int X3_XSHIFT = 4;   // activations: |x| < 2^(16 - 4); lo normal down to |x| ~ 8e-3, below that absolute steps of 4e-9

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) __fp16 fp16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2v_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16v8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16v2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

// Two floats -> one dword of hi halves + one dword of lo halves.
template <typename T, bool WEIGHT> __device__ __forceinline__ void x3_split_pair(float x0, float x1, uint32_t& h, uint32_t& l) {
    if constexpr (X3<T>::half) {
        constexpr float K = (float)(1 << (WEIGHT ? X3_WSHIFT : X3_XSHIFT));
        // Both halves by v_cvt_pk_f16_f32 (round to nearest even, OVERFLOW -> INF): a - hi is exact in fp32 either way, and a scaled
        // operand past the half range (|x| >= 65520 / 2^shift) becomes inf - inf = NaN in the products -- loud, where the
        // round-toward-zero conversion saturated both halves at 65504 and produced a finite, silently wrong result.
        const f32x2_t ab = {x0 * K, x1 * K};
        const f16x2v_t hp = __builtin_convertvector(ab, f16x2v_t);
        const f32x2_t res = {ab[0] - (float)hp[0], ab[1] - (float)hp[1]};
        const f16x2v_t lp = __builtin_convertvector(res, f16x2v_t);
        __builtin_memcpy(&h, &hp, 4);
        __builtin_memcpy(&l, &lp, 4);
    } else {
        const f32x2_t v = {x0, x1};
        const bf16v2_t hp = __builtin_convertvector(v, bf16v2_t);       // v_cvt_pk_bf16_f32, round to nearest even
        uint32_t hb;
        __builtin_memcpy(&hb, &hp, 4);
        h = hb;
        if constexpr (X3<T>::single) {
            l = 0;        // (never multiplied)
        } else {
            const f32x2_t r = {x0 - __uint_as_float(hb << 16), x1 - __uint_as_float(hb & 0xffff0000u)};
            const bf16v2_t lp = __builtin_convertvector(r, bf16v2_t);
            __builtin_memcpy(&l, &lp, 4);
        }
    }
}
// 8 floats (two 16-byte fragments) -> 8 hi halves + 8 lo halves, each one MFMA operand register quad.
template <typename T, bool WEIGHT> __device__ __forceinline__ void x3_split(const uint4& f0, const uint4& f1, uint4& hi, uint4& lo) {
    x3_split_pair<T, WEIGHT>(__uint_as_float(f0.x), __uint_as_float(f0.y), hi.x, lo.x);
    x3_split_pair<T, WEIGHT>(__uint_as_float(f0.z), __uint_as_float(f0.w), hi.y, lo.y);
    x3_split_pair<T, WEIGHT>(__uint_as_float(f1.x), __uint_as_float(f1.y), hi.z, lo.z);
    x3_split_pair<T, WEIGHT>(__uint_as_float(f1.z), __uint_as_float(f1.w), hi.w, lo.w);
}
// c += a*b with a = ah + al, b = bh + bl, the lo*lo term dropped; small terms first
template <typename T> __device__ __forceinline__ void x3_mma(const uint4& ah, const uint4& al, const uint4& bh, const uint4& bl, f32x16_t& c) {
    if constexpr (X3<T>::half) {
        f16x8_t a0, a1, b0, b1;
        __builtin_memcpy(&a0, &ah, 16); __builtin_memcpy(&a1, &al, 16); __builtin_memcpy(&b0, &bh, 16); __builtin_memcpy(&b1, &bl, 16);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c, 0, 0, 0);
    } else if constexpr (X3<T>::single) {
        bf16v8_t a0, b0;
        __builtin_memcpy(&a0, &ah, 16); __builtin_memcpy(&b0, &bh, 16);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c, 0, 0, 0);
    } else {
        bf16v8_t a0, a1, b0, b1;
        __builtin_memcpy(&a0, &ah, 16); __builtin_memcpy(&a1, &al, 16); __builtin_memcpy(&b0, &bh, 16); __builtin_memcpy(&b1, &bl, 16);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c, 0, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------
// division by a runtime constant (host precomputes); exact for n < 2^31, d < 2^31
// ---------------------------------------------------------------------------------------------
struct FastDiv {
    uint32_t mul, shift, d;
};
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    if (d == 1) { f.mul = 0; f.shift = 0; return f; }
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;                       // l = ceil(log2 d)
    uint64_t m = ((1ull << 32) * ((1ull << l) - d)) / d + 1;
    f.mul = (uint32_t)m;
    f.shift = l;
    return f;
}
__host__ __device__ inline uint32_t fastdiv(uint32_t n, const FastDiv& f) {
    if (f.d == 1) return n;
#ifdef __HIP_DEVICE_COMPILE__
    uint32_t t = __umulhi(n, f.mul);
#else
    uint32_t t = (uint32_t)(((uint64_t)n * f.mul) >> 32);
#endif
    // (t + ((n - t) >> 1)) >> (shift - 1)   -- Granlund-Montgomery round-up method
    return (t + ((n - t) >> 1)) >> (f.shift - 1);
}

// ---------------------------------------------------------------------------------------------
// XCD-aware block remap: blocks that the dispatcher round-robins over the 8 XCDs are remapped so every XCD
// works on one contiguous range of tiles (its private 4 MiB L2 then sees the operand reuse).  Bijective for any
// grid size (cdna_hip_programming.md T1).
// ---------------------------------------------------------------------------------------------
__device__ inline uint32_t xcd_remap(uint32_t bid, uint32_t nblk) {
    uint32_t xcd = bid % VINCE_NUM_XCD, s = bid / VINCE_NUM_XCD;
    uint32_t q = nblk / VINCE_NUM_XCD, r = nblk % VINCE_NUM_XCD;
    uint32_t base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + s;
}

// (sum, sum of squares) of channel c folded over the R replicas of a double[R][C][2] statistics block: one 16-byte load per replica,
// four replicas requested before they are added (the loop with one 8-byte load per iteration waited for a round trip to L2 per
// replica and value: 32 in a row at R = 16, in the prologue of every BatchNorm pass over a large tensor).  Same summation order.
__device__ __forceinline__ void fold_replicas(const double* __restrict__ stats, int R, int C, int c, double& s1, double& s2) {
    typedef __attribute__((ext_vector_type(2))) double d2_t;
    s1 = 0;
    s2 = 0;
    int r = 0;
    for (; r + 4 <= R; r += 4) {
        d2_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const d2_t*)(stats + ((size_t)(r + u) * C + c) * 2);
#pragma unroll
        for (int u = 0; u < 4; ++u) { s1 += v[u][0]; s2 += v[u][1]; }
    }
    for (; r < R; ++r) {
        const d2_t v = *(const d2_t*)(stats + ((size_t)r * C + c) * 2);
        s1 += v[0];
        s2 += v[1];
    }
}

// wave-level helpers (64 lanes)
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA (buffer_load ... lds) helpers shared by the direct-to-LDS kernels
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(4))) int v4i_t;

// One LDS-DMA instruction: 64 lanes x 16 bytes from (descriptor, per-lane byte offset) to LDS[m0 + lane*16].
// Issued through inline asm so that hipcc neither tracks it (it would wait vmcnt(0) before every later ds_read of
// the same __shared__ array, serialising load and compute) nor reuses M0 across it; the caller owns the waits:
// s_waitcnt vmcnt(N) + barrier before any wave reads the destination.
static __device__ __forceinline__ void lds_dma16(uint32_t lds_addr_uniform, uint32_t voff, v4i_t rsrc) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_addr_uniform), "v"(voff), "s"(rsrc)
        : "memory");
}

// The same without preserving M0: only for kernels whose only M0 users are these instructions (M0 is a reserved register the
// compiler touches for indirect register indexing, LDS-direct and message sends -- none of which such a kernel may contain).
static __device__ __forceinline__ void lds_dma16_m0(uint32_t lds_addr_uniform, uint32_t voff, v4i_t rsrc) {
    asm volatile(
        "s_mov_b32 m0, %0\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, 0 offen lds"
        :
        : "s"(lds_addr_uniform), "v"(voff), "s"(rsrc)
        : "memory");
}

static __device__ __forceinline__ v4i_t make_rsrc(const void* ptr, uint32_t bytes) {
    const uint64_t a = (uint64_t)ptr;
    v4i_t r;
    r[0] = (int)(uint32_t)a;
    r[1] = (int)(uint32_t)(a >> 32);      // stride 0
    r[2] = (int)bytes;
    r[3] = 0x00020000;
    return r;
}

template <int N> static __device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else static_assert(N == 0, "add the vmcnt literal");
}
