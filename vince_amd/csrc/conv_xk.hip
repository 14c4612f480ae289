// The expand convolution of the 256-wide bottlenecks (layer3's conv3: 256 -> 1024 at 14 x 14, resnet.py:123) as an HBM stream, with
// the BatchNorm + ReLU that produces its input (bn2, resnet.py:119-121) optionally folded into the operand path.
//
//     out[p][co] = sum_k W[co][k] x'[p][k],    x' = x   or   relu(x * scale[k] + shift[k])  (train-mode constants from bn_in->stats)
//     stats[co] += (sum, sum of squares) of the STORED out                                      (feeds bn3's finalize)
//
// 2 * 256 FLOP per output element against 2 bytes written and 0.5 read: the layer is a write stream (103 MB out, 26 MB in at the
// benchmark batch) that the implicit-GEMM kernel runs at 2.1 TB/s -- its 128 x 128 tiles fetch the input tile once per channel tile
// (8 x), pay a transposition through LDS, a shuffle + LDS reduction and 256 atomics per tile for the statistics, and all their
// epilogues fall into the same phase.  Here:
//   * persistent workgroups of eight wavefronts, one per CU, each owning 128 output channels.  A wavefront owns 32 pixels x 64
//     channels (2 MFMA tiles) and keeps its WEIGHT fragments -- 64 channels x 256 reduction elements = 128 registers per lane -- in
//     registers for the whole launch: no weight traffic, no weight reads from LDS, and all of LDS left for the input;
//   * the input streams through a ring of seven 16 KB units (128 pixels x 64 reduction elements), SIX units = 96 KB in flight per
//     CU: the first version of this kernel kept the weights in LDS and two units in flight and ran at the LDS-DMA round trip
//     (~3 us per unit under load, 72 us per launch); every wavefront issues its two 1 KB pieces of the unit six ahead right after
//     the barrier that publishes the current one, one barrier per unit;
//   * v_permlane32_swap turns the accumulators into whole 16-byte channel chunks, a private 4 KB LDS buffer turns "lane = pixel"
//     into "8 lanes = one 128-byte line", and the statistics are plain register accumulators (a lane owns the same 8 channels for
//     the whole launch) that leave as one fp64 atomic per channel and wavefront at the end;
//   * the channel groups of one pixel-tile lane share an XCD, so the input crosses HBM once and comes from that L2 seven times.
// vmcnt bookkeeping: the wavefronts that issue the LDS-DMA also issue the output stores (buffer stores, always issued: rows past the
// end carry an out-of-range offset), and vector memory operations retire in order, so "my pieces of unit u have landed" is
// vmcnt(10 + 4 * store groups issued since) -- 10 = the two pieces of each of the five younger units, a store group = the four
// stores of one pixel tile -- which is 10 / 14 / 18 depending on the position in the tile cycle.
// The MFMA pipe is ~15 % busy in such a kernel, which is what makes the bn_in transform (24 VALU instructions per activation
// fragment, scale / shift from a 2 KB LDS table) cheap here, where it costs the implicit-GEMM kernel more than the pass it removes
// (tools/bnin_micro.py).
#include <string.h>

#include "common.h"

namespace {

constexpr int XK_K = 256;
constexpr int XK_PX = 128;                    // pixels per tile: 4 wavefront rows of 32
constexpr int XK_CG = 128;                    // channels per workgroup: 2 wavefront columns of 64
constexpr int XK_WAVES = 8;
constexpr int XK_THREADS = XK_WAVES * 64;
constexpr int XK_UPT = 4;                     // units per pixel tile (64 reduction elements = 2 K blocks of 64 bytes each)
constexpr int XK_SLOTS = 7, XK_AHEAD = 6;
constexpr int XK_XB = 2 * XK_PX * 64;         // 16 KB per unit
constexpr int XK_TBUF = 32 * 64 * 2;          // per-wavefront store buffer: 32 pixels x 64 channels
constexpr int XK_OFF_TAB = XK_SLOTS * XK_XB;
constexpr int XK_OFF_T = XK_OFF_TAB + 2 * XK_K * 4;
constexpr int XK_BYTES = XK_OFF_T + XK_WAVES * XK_TBUF;
static_assert(XK_BYTES <= 160 * 1024, "LDS budget");
static_assert(XK_XB / 1024 == 2 * XK_WAVES, "two DMA pieces per wavefront and unit");

struct XkParams {
    const void* x;
    const void* w;
    void* out;
    double* stats;
    vince_bn_train bnin;      // only read by the BNIN instantiation
    uint32_t rows, Co, x_bytes, out_bytes;
    int ptiles, cgroups, replicas;
};

static __device__ __forceinline__ void buffer_store16(const uint4& v, uint32_t voff, v4i_t rsrc) {
    v4i_t d;
    __builtin_memcpy(&d, &v, 16);
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" ::"v"(d), "v"(voff), "s"(rsrc) : "memory");
}

template <bool BNIN>
__global__ __launch_bounds__(XK_THREADS) void conv_xk_kernel(const XkParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[XK_BYTES];
    unsigned char* const xsm = smem;
    float* const tab = (float*)(smem + XK_OFF_TAB);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup -> (channel group, pixel-tile lane): the channel groups of one lane on one XCD (consecutive workgroups go to the 8 XCDs
    // round-robin), so that the input tile comes from HBM once
    int cg, first;
    const int step = gridDim.x / p.cgroups;
    if (p.cgroups > 1 && gridDim.x % (VINCE_NUM_XCD * p.cgroups) == 0) {
        const int xcd = blockIdx.x % VINCE_NUM_XCD, i = blockIdx.x / VINCE_NUM_XCD;
        cg = i % p.cgroups;
        first = xcd * (step / VINCE_NUM_XCD) + i / p.cgroups;
    } else {
        cg = blockIdx.x % p.cgroups;
        first = blockIdx.x / p.cgroups;
    }
    const int c0 = cg * XK_CG;
    const int ntiles = first < p.ptiles ? (p.ptiles - first + step - 1) / step : 0;
    const int nunits = ntiles * XK_UPT;

    const v4i_t rsrc_x = make_rsrc(p.x, p.x_bytes);
    const v4i_t rsrc_o = make_rsrc(p.out, p.out_bytes);
    const uint32_t smem_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    constexpr uint32_t OOB = 0x80000000u;

    const int wp = wave >> 1, wc = wave & 1;                // 32-pixel row, 64-channel half
    const int khalf = lane >> 5;

    // ---- this wavefront's weights: fragment (j, q) = channels c0 + wc*64 + j*32 + (lane & 31), reduction elements
    // (q >> 1) * 32 + ((q & 1) * 2 + khalf) * 8 .. + 8
    uint4 wreg[2][16];
    {
        const bf16_t* __restrict__ wg = (const bf16_t*)p.w;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t co = (uint32_t)(c0 + wc * 64 + j * 32 + (lane & 31));
#pragma unroll
            for (int q = 0; q < 16; ++q)
                wreg[j][q] = co < p.Co ? *(const uint4*)(wg + (size_t)co * XK_K + (q >> 1) * 32 + ((q & 1) * 2 + khalf) * 8) : make_uint4(0, 0, 0, 0);
        }
    }
    if constexpr (BNIN) {
        // train-mode finalize as in bn_apply_kernel: every workgroup folds the statistic replicas, workgroup 0 publishes
        const vince_bn_train& fin = p.bnin;
        for (int c = tid; c < XK_K; c += XK_THREADS) {
            double s1 = 0, s2 = 0;
            for (int r = 0; r < fin.replicas; ++r) {
                s1 += fin.stats[((size_t)r * XK_K + c) * 2];
                s2 += fin.stats[((size_t)r * XK_K + c) * 2 + 1];
            }
            const double cnt = (double)fin.count;
            const double m = s1 / cnt;
            double var = s2 / cnt - m * m;
            if (var < 0) var = 0;
            const float mean = (float)m;
            const float invstd = (float)(1.0 / sqrt(var + (double)fin.eps));
            const float scv = fin.gamma[c] * invstd;
            const float shv = fin.beta[c] - mean * scv;
            tab[c] = scv;
            tab[XK_K + c] = shv;
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
    }
    // every compiler-tracked load above has returned before the first LDS-DMA is counted
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- the input ring.  Unit u = (pixel tile u / 4, reduction elements (u % 4) * 64 .. + 64) lives in slot u % 7 as two K blocks of
    // 128 rows x 64 bytes; a DMA piece = 16 rows of one K block; lane -> (row dr of the piece, slot), the slot holds logical K chunk
    // slot ^ ((row >> 2) & 3).  This wavefront's pieces: 2 * wave, 2 * wave + 1 -> K block wave >> 2, rows ((2 * wave) & 7) * 16 .. + 32
    const int dr = lane >> 2, dslot = lane & 3;
    const int dchunk = dslot ^ ((dr >> 2) & 3);
    const int pk = wave >> 2, prb = ((2 * wave) & 7) * 16;
    auto issue_u = [&](int u) {
        const int t = u / XK_UPT, part = u % XK_UPT;
        const uint32_t p0 = (uint32_t)(first + t * step) * XK_PX;
        const uint32_t sbase = __builtin_amdgcn_readfirstlane(smem_base + (uint32_t)(u % XK_SLOTS) * XK_XB + pk * (XK_PX * 64) + prb * 64);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const uint32_t m = p0 + (uint32_t)(prb + e * 16 + dr);
            // (units past the end are issued as zero fills so that the counts stay uniform)
#ifdef XK_NO_DMA
            const uint32_t off = OOB + (uint32_t)(u < nunits && m < p.rows);
#else
            const uint32_t off = (u < nunits && m < p.rows) ? (m * (uint32_t)XK_K + (uint32_t)(part * 64 + pk * 32 + dchunk * 8)) * 2u : OOB;
#endif
            lds_dma16(sbase + e * 1024, off, rsrc_x);
        }
    };
#pragma unroll
    for (int u = 0; u < XK_AHEAD; ++u) issue_u(u);
    __syncthreads();                                        // the table (the ring itself is published unit by unit below)

    const int sw = ((lane & 31) >> 2) & 3;
    const int row_off = (lane & 31) * 64;
    unsigned char* const tbuf = smem + XK_OFF_T + wave * XK_TBUF;
    float ssum[8], ssq[8];                                  // statistics of this lane's 8 channels (chunk lane & 7 of the wavefront's 64)
#pragma unroll
    for (int e = 0; e < 8; ++e) ssum[e] = ssq[e] = 0.f;

    for (int t = 0; t < ntiles; ++t) {
        f32x16_t acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
        for (int part = 0; part < XK_UPT; ++part) {
            const int u = t * XK_UPT + part;
            // this wavefront's pieces of unit u have landed: younger = 5 units x 2 pieces + the store groups of the last one (parts 2, 3)
            // or two (parts 0, 1) pixel tiles, as far as they exist
            const int groups = min(t, part <= 1 ? 2 : 1);
            if (groups == 0) wait_vmcnt<10>();
            else if (groups == 1) wait_vmcnt<14>();
            else wait_vmcnt<18>();
            __builtin_amdgcn_s_barrier();                   // B(u): everyone's have; and everyone is done with unit u - 1
            issue_u(u + XK_AHEAD);                          // into the slot unit u - 1 occupied
            const unsigned char* xfrag = xsm + (u % XK_SLOTS) * XK_XB + (wp * 32) * 64 + row_off;
#pragma unroll
            for (int ktl = 0; ktl < 2; ++ktl)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int slot = ((s2 * 2 + khalf) ^ sw) * 16;
                    uint4 xf = *(const uint4*)(xfrag + ktl * (XK_PX * 64) + slot);
                    if constexpr (BNIN) {
                        const int kb = part * 64 + ktl * 32 + (s2 * 2 + khalf) * 8;     // first input channel of this lane's fragment
                        bn_in_apply(xf, tab + kb, tab + XK_K + kb);
                    }
                    bf16x8_t bv;
                    __builtin_memcpy(&bv, &xf, 16);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        bf16x8_t av;
                        __builtin_memcpy(&av, &wreg[j][(part * 2 + ktl) * 2 + s2], 16);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[j], 0, 0, 0);
                    }
                }
        }
        // ---- epilogue: accumulators (lane = pixel, register quad = 4 consecutive channels, lane halves interleaved) -> whole 16-byte
        // channel chunks per lane -> store buffer -> 8 consecutive lanes write one 128-byte line
        const int row = lane & 31;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[j][8 * gp + e]),
                                                                    __float_as_uint(acc[j][8 * gp + 4 + e]), false, false);
                    v[e] = __uint_as_float(r[0]);
                    v[4 + e] = __uint_as_float(r[1]);
                }
                const int cpos = j * 4 + 2 * gp + khalf;
                *(uint4*)(tbuf + row * 128 + ((cpos ^ (row & 7)) * 16)) = Chunk<bf16_t>::pack(v);
            }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        const uint32_t pix0 = (uint32_t)(first + t * step) * XK_PX + (uint32_t)(wp * 32);
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int prow = (lane >> 3) + 8 * sidx, c = lane & 7;
            const uint4 val = *(const uint4*)(tbuf + prow * 128 + ((c ^ (prow & 7)) * 16));
            const uint32_t pix = pix0 + (uint32_t)prow;
            const bool ok = pix < p.rows;
#ifdef XK_NO_STORE
            buffer_store16(val, OOB, rsrc_o);
#else
            buffer_store16(val, ok ? (pix * p.Co + (uint32_t)(c0 + wc * 64 + c * 8)) * 2u : OOB, rsrc_o);    // always issued
#endif
#ifndef XK_NO_STATS
            if (ok)
#else
            if (ok && p.stats == (double*)16)
#endif
            {
                float f[8];
                Chunk<bf16_t>::unpack(val, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { ssum[e] += f[e]; ssq[e] += f[e] * f[e]; }
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the zero fills issued past the end
    if (p.stats) {
        // lanes l, l + 8, l + 16, ... hold the same channels: fold them, then one fp64 atomic per channel and wavefront
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) {
                ssum[e] += __shfl_xor(ssum[e], o, 64);
                ssq[e] += __shfl_xor(ssq[e], o, 64);
            }
        if (lane < 8) {
            double* dst = p.stats + (size_t)(blockIdx.x % (unsigned)p.replicas) * p.Co * 2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ch = c0 + wc * 64 + lane * 8 + e;
                unsafeAtomicAdd(dst + (size_t)ch * 2, (double)ssum[e]);
                unsafeAtomicAdd(dst + (size_t)ch * 2 + 1, (double)ssq[e]);
            }
        }
    }
}

int xk_num_cu() {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            n_cu = prop.multiProcessorCount;
    }
    return n_cu;
}

}  // namespace

// (declared in include/vince_hip.h; vince_conv_expand_stats in conv_xjoin.hip forwards its K = 256 case here with bn_in = NULL)
extern "C" int vince_conv_expand_stats_bn(int dtype, const void* x, const void* w, int64_t rows, int32_t K, int32_t Co, void* out,
                                          double* stats, int32_t replicas, const vince_bn_train* bn_in, void* stream) {
    VINCE_CHECK_ARG(dtype == VINCE_BF16, VINCE_E_DTYPE, "vince_conv_expand_stats_bn: bf16 only");
    VINCE_CHECK_ARG(x && w && out && rows > 0, VINCE_E_ARG, "vince_conv_expand_stats_bn: null pointer");
    VINCE_CHECK_ARG(K == XK_K, VINCE_E_UNSUPPORTED, "vince_conv_expand_stats_bn: K=%d (%d)", K, XK_K);
    VINCE_CHECK_ARG(Co > 0 && Co % XK_CG == 0, VINCE_E_SHAPE, "vince_conv_expand_stats_bn: Co=%d must be a multiple of %d", Co, XK_CG);
    VINCE_CHECK_ARG((((uintptr_t)x | (uintptr_t)w | (uintptr_t)out) & 15) == 0, VINCE_E_ALIGN,
                    "vince_conv_expand_stats_bn: pointers must be 16-byte aligned");
    const unsigned long long xb = (unsigned long long)rows * K * 2, ob = (unsigned long long)rows * Co * 2;
    VINCE_CHECK_ARG(xb < 0x7ff00000ull && ob < 0x7ff00000ull && rows < (1ll << 31), VINCE_E_UNSUPPORTED,
                    "vince_conv_expand_stats_bn: input or output beyond the 31-bit buffer offsets");
    if (replicas <= 0 || replicas > VINCE_STATS_REPLICAS) replicas = VINCE_STATS_REPLICAS;
    XkParams p;
    memset(&p, 0, sizeof(p));
    if (bn_in) {
        VINCE_CHECK_ARG(bn_in->stats && bn_in->count > 0 && bn_in->gamma && bn_in->beta && bn_in->scale && bn_in->shift && !bn_in->out_sum,
                        VINCE_E_ARG, "vince_conv_expand_stats_bn: bn_in needs stats, count, gamma, beta, scale and shift (and takes no out_sum)");
        VINCE_CHECK_ARG(!bn_in->running_mean == !bn_in->running_var, VINCE_E_ARG,
                        "vince_conv_expand_stats_bn: bn_in running_mean and running_var come together");
        p.bnin = *bn_in;
        if (p.bnin.replicas <= 0 || p.bnin.replicas > VINCE_STATS_REPLICAS) p.bnin.replicas = VINCE_STATS_REPLICAS;
    }
    p.x = x; p.w = w; p.out = out; p.stats = stats; p.replicas = replicas;
    p.rows = (uint32_t)rows; p.Co = (uint32_t)Co; p.x_bytes = (uint32_t)xb; p.out_bytes = (uint32_t)ob;
    p.ptiles = (int)((rows + XK_PX - 1) / XK_PX);
    p.cgroups = Co / XK_CG;
    long grid = xk_num_cu();
    const long items = (long)p.ptiles * p.cgroups;
    if (grid > items) grid = items;
    grid = grid / p.cgroups * p.cgroups;                    // every workgroup keeps one channel group
    if (grid < p.cgroups) grid = p.cgroups;
    VinceProfScope prof(VINCE_TAG_XSTATS, (double)rows * (K + Co) * 2, stream);
    if (bn_in) hipLaunchKernelGGL(conv_xk_kernel<true>, dim3((unsigned)grid), dim3(XK_THREADS), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(conv_xk_kernel<false>, dim3((unsigned)grid), dim3(XK_THREADS), 0, (hipStream_t)stream, p);
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}
