// Image-strip 3x3 convolution for layer1 (resnet.py:119-121, conv2 of the 64-wide bottlenecks: 64 -> 64 channels, 56 x 56, stride 1,
// pad 1) -- the "next kernel" of DESIGN.md section 7: the implicit GEMM fetches every input element nine times (once per tap) in
// 64-byte pieces of 16 different pixels per wave instruction, which is REQUEST-bound (tools/micro/feed_micro.hip); here
//   * a persistent workgroup owns whole images and walks them in strips of 4 output rows (224 pixels = 7 MFMA row blocks);
//   * the input rows live in an LDS RING of 10 image rows (each 56 pixels + one zero pixel either side, 128 bytes per pixel =
//     exactly one cache line): a strip reads 6 of them, the LOADER wavefront brings the next strip's 4 new rows meanwhile, every
//     DMA instruction a contiguous 1 KB (8 whole pixels); each input element crosses the L2 -> LDS path ONCE;
//   * all of W (64 x 9 x 64 bf16 = 72 KB) is resident; the nine taps are shifted fragment reads of the same rows -- the left / right
//     image edge falls on the zero pixels, the top / bottom edge on rows the buffer descriptor zero-filled;
//   * seven CONSUMER wavefronts each own one 32-pixel block x 64 channels (two MFMA 32x32x16 tiles) and finish it like
//     conv_xjoin's PLAIN mode: v_permlane32_swap to whole 16-byte channel chunks, a 2 KB per-wave transposition so that 4 lanes
//     write a 64-byte half line, per-lane BatchNorm statistics (a lane owns the same 8 channels for the whole launch).
// LDS: 72 KB + 72.5 KB + 7 x 2 KB = 158.5 KB of 160.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"

namespace {

constexpr int XS_C = 64;                   // Ci = Co
constexpr int XS_W = 56;                   // image width
constexpr int XS_ROWS = 4;                 // output rows per strip
constexpr int XS_PIX = XS_ROWS * XS_W;     // 224 = 7 x 32
constexpr int XS_CONSUMERS = XS_PIX / 32;  // 7: one 32-pixel block each
constexpr int XS_THREADS = (XS_CONSUMERS + 1) * 64;   // + the loader wavefront
constexpr int XS_RING = 10;                // image rows in the ring
constexpr int XS_RPX = XS_W + 2;           // pixels per ring row (zero pixel either side)
constexpr int XS_ROWB = XS_RPX * 128;      // 7424
constexpr int XS_WB = 9 * 2 * XS_C * 64;   // resident weights: [tap][32-element K block][co][64 B]
constexpr int XS_XB = XS_RING * XS_ROWB;
constexpr int XS_TBUF = 32 * 32 * 2;       // per-consumer transposition buffer (32 pixels x 32 channels)
constexpr int XS_BYTES = XS_WB + XS_XB + XS_CONSUMERS * XS_TBUF;
constexpr int XS_PPR = XS_W / 8;           // DMA pieces per image row (8 pixels each)
#ifndef XS_ABLATE
#define XS_ABLATE 0                        // measurement builds only: 1 no MFMA, 2 no output stores / statistics, 4 no loader DMA, 8 no fragment reads
#endif
#ifndef XS_VARIANT
#define XS_VARIANT 0
#endif
#ifndef XS_AHEAD
#define XS_AHEAD 4                         // fragment sets requested ahead of their MFMAs (16 reduction elements each)
#endif

struct XsParams {
    const void* x;
    const void* w;
    void* out;
    double* stats;
    uint32_t x_bytes, w_bytes;
    int N, H, strips_per_image, replicas;
    int tap_w[9];                          // weight tap index of kernel position (dh + 1) * 3 + (dw + 1)
    // BNRED (the input-gradient launch, vince_bn_reduce with mask_scale / mask_shift): the tensor being written is the gradient dz that a
    // BatchNorm + ReLU backward consumes next; `stats` then receives (sum g, sum g * xhat) of the STORED values, g = dz where
    // y * msc + msh > 0, xhat = (y - mean) * invstd -- what the implicit-GEMM gradient epilogue fuses for every other layer
    const void* br_y;
    const float* br_mean;
    const float* br_invstd;
    const float* br_msc;
    const float* br_msh;
    // BIAS (the BatchNorm-folded inference forward): out = [relu](conv + bias[co]) applied to the bf16-rounded convolution output, as
    // vince_conv_igemm's epilogue does; no statistics
    const float* bias;
    int relu;
};

template <bool BNRED, bool BIAS = false>
__global__ __launch_bounds__(XS_THREADS) void conv3x3_strip_kernel(const XsParams p) {
    static_assert(!(BNRED && BIAS), "bias + ReLU is a forward epilogue");
    __shared__ __attribute__((aligned(16))) unsigned char smem[XS_BYTES + ((BNRED || BIAS) ? 512 : 0)];
    unsigned char* const wsm = smem;
    unsigned char* const xsm = smem + XS_WB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const v4i_t rsrc_x = make_rsrc(p.x, p.x_bytes);
    const v4i_t rsrc_w = make_rsrc(p.w, p.w_bytes);
    const uint32_t smem_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    constexpr uint32_t OOB = 0x80000000u;

    // ---- resident weights (every wavefront helps, once): piece = 16 channel rows x 64 bytes of one (tap, K block); the 16-byte
    // slot holds logical K chunk slot ^ ((row >> 2) & 3) (the fragment-read swizzle of the implicit-GEMM kernels)
    {
        const int dr = lane >> 2, dslot = lane & 3;
        const int dchunk = dslot ^ ((dr >> 2) & 3);
        constexpr int PIECES = 9 * 2 * XS_C / 16;                 // 72
        for (int pc = wave; pc < PIECES; pc += XS_CONSUMERS + 1) {
            const int tk = pc / (XS_C / 16), rb = (pc % (XS_C / 16)) * 16;     // (tap, K block) index, first channel row
            const int tap = tk >> 1, kt = tk & 1;
            const uint32_t co = (uint32_t)(rb + dr);
            const uint32_t off = ((co * 9u + (uint32_t)p.tap_w[tap]) * (uint32_t)XS_C + (uint32_t)(kt * 32 + dchunk * 8)) * 2u;
            lds_dma16(__builtin_amdgcn_readfirstlane(smem_base + tk * (XS_C * 64) + rb * 64), off, rsrc_w);
        }
        if constexpr (BNRED) {
            if (tid < 128) ((float*)(smem + XS_BYTES))[tid] = tid < 64 ? p.br_msc[tid] : p.br_msh[tid - 64];
        }
        if constexpr (BIAS) {
            if (tid < 64) ((float*)(smem + XS_BYTES))[tid] = p.bias ? p.bias[tid] : 0.f;
        }
        // the zero pixels of every ring row
        for (int i = tid; i < XS_RING * 2 * 8; i += XS_THREADS) {
            const int row = i / 16, side = (i / 8) & 1, c = i & 7;
            *(uint4*)(xsm + row * XS_ROWB + (side ? (XS_RPX - 1) * 128 : 0) + c * 16) = make_uint4(0, 0, 0, 0);
        }
        wait_vmcnt<0>();
        __syncthreads();
    }

    const int n_images = p.N;
    const int spi = p.strips_per_image;

    // =============================== loader ===============================
    // image rows -1 .. H of one image are sequence rows 0 .. H + 1; sequence row r lives in ring slot r % 10; strip s reads
    // sequence rows 4 s .. 4 s + 5.  Before B(s) the rows of strip s have landed; right after it the 4 new rows of strip s + 1
    // are requested into the slots strip s - 1 has released.
    auto issue_row = [&](int img, int seq) {
        const int px = lane >> 3, pos = lane & 7;
        const int h = seq - 1;
        const uint32_t slot_base = smem_base + XS_WB + (uint32_t)(seq % XS_RING) * XS_ROWB + 128u;
        const bool live = h >= 0 && h < p.H;
#pragma unroll
        for (int pc = 0; pc < XS_PPR; ++pc) {
            const int pp = 1 + pc * 8 + px;                          // position in the ring row
            const int chunk = pos ^ ((pp >> 1) & 7);
            const uint32_t pix = ((uint32_t)img * (uint32_t)p.H + (uint32_t)h) * XS_W + (uint32_t)(pc * 8 + px);
            const uint32_t off = live ? pix * 128u + (uint32_t)chunk * 16u : OOB;
            if constexpr (!(XS_ABLATE & 4)) lds_dma16(__builtin_amdgcn_readfirstlane(slot_base + pc * 1024), off, rsrc_x);
        }
    };

    if (wave == XS_CONSUMERS) {
        for (int img = blockIdx.x; img < n_images; img += gridDim.x) {
            for (int r = 0; r < 6; ++r) issue_row(img, r);
            for (int s = 0; s < spi; ++s) {
                wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();                           // B(s)
                if (s + 1 < spi)
                    for (int r = 0; r < 4; ++r) issue_row(img, 4 * s + 6 + r);
            }
            // the consumers' last reads of this image: one more rendezvous before the ring is refilled from row 0
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // =============================== consumers ===============================
    // Consumer v owns the 32-pixel block v x all 64 channels (two MFMA 32x32x16 tiles sharing the pixel fragment: 3 LDS reads per 2
    // MFMAs).  Measured alternatives: four consumers of 2 x 2 tiles (4 reads per 4 MFMAs, one wavefront per SIMD, the loader folded
    // into the one with a single block) run 1.3-1.6x SLOWER (127 against 97 us at B=256) -- what a wavefront pays per fragment read is
    // ~80-100 cycles whatever the prefetch depth (2 .. 6 sets ahead measured equal), so the CU needs many wavefronts reading at once,
    // not fewer reads; NB stays a template parameter for that experiment.
    const int khalf = lane >> 5;
    const int sw = ((lane & 31) >> 2) & 3;
    const unsigned char* const wfrag = wsm + (lane & 31) * 64;
    bf16_t* __restrict__ out = (bf16_t*)p.out;
    unsigned char* const tbuf = smem + XS_WB + XS_XB + wave * XS_TBUF;
    const int trow = lane & 31;
    float ssum[2][8], ssq[2][8];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) ssum[q][e] = ssq[q][e] = 0.f;
    // BNRED: the constants of this lane's 2 x 8 channels (a lane stores the same channels for the whole launch)
    // (in the loop only the ReLU test needs constants -- sum g * xhat = invstd * (sum g * y - mean * sum g) is finished per lane at the
    // end -- and they sit in the last 512 bytes of LDS: the register file is full)
    const bf16_t* __restrict__ br_y = (const bf16_t*)p.br_y;
    const float* const ctab = (const float*)(smem + XS_BYTES);      // [2][64]: mask_scale, mask_shift

    auto consume = [&](auto nbc) {
        constexpr int NB = decltype(nbc)::value;
        int hr[NB], poff[NB][3], pswz[NB][3];               // per block: strip row of this lane's pixel; per dw: byte offset of the
#pragma unroll                                              // source pixel inside a ring row and its chunk swizzle
        for (int u = 0; u < NB; ++u) {
            const int idx = (NB * wave + u) * 32 + (lane & 31);
            hr[u] = idx / XS_W;
            const int wcol = idx - hr[u] * XS_W;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int pp = wcol + b;                    // w + 1 + dw
                poff[u][b] = pp * 128;
                pswz[u][b] = (pp >> 1) & 7;
            }
        }
        // The epilogue of strip s (accumulators -> bf16 chunks -> 2 KB transposition -> stores + statistics, ~300 VALU / LDS
        // instructions per block) rides on the MFMA loop of strip s + 1, one piece per step (every strip starts at a workgroup
        // barrier, so the consumers of a SIMD would otherwise run their MFMA loops together and then their epilogues together).
        f32x16_t pacc[NB][2];                               // accumulators of the previous strip
        uint32_t ppix0 = 0;                                 // its first output pixel (this wavefront's first block)
        uint4 opk[2][2], tval;
        uint4 yq[2];                                        // BNRED: the BatchNorm input at this lane's two output chunks of a pass
        auto epi_pack = [&](int u, int j, int gp) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(pacc[u][j][8 * gp + e]),
                                                                __float_as_uint(pacc[u][j][8 * gp + 4 + e]), false, false);
                v[e] = __uint_as_float(r[0]);
                v[4 + e] = __uint_as_float(r[1]);
            }
            opk[j][gp] = Chunk<bf16_t>::pack(v);
        };
        auto epi_write = [&](int pass) {                    // chunks of MFMA tile `pass` (32 channels) into the buffer
            asm volatile("" ::: "memory");
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                const int cpos = 2 * gp + khalf;
                *(uint4*)(tbuf + trow * 64 + ((cpos ^ ((trow >> 1) & 3)) * 16)) = opk[pass][gp];
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
        };
        auto epi_read = [&](int sidx) {
            const int prow = lane / 4 + 16 * sidx, c = lane % 4;
            tval = *(const uint4*)(tbuf + prow * 64 + ((c ^ ((prow >> 1) & 3)) * 16));
        };
        auto epi_yload = [&](int u, int pass) {             // BNRED: requested 6 pieces (MFMA steps) before the stores that use it
            if constexpr (BNRED) {
#pragma unroll
                for (int sidx = 0; sidx < 2; ++sidx) {
                    const int prow = lane / 4 + 16 * sidx, c = lane % 4;
                    yq[sidx] = *(const uint4*)(br_y + (size_t)(ppix0 + (uint32_t)(32 * u + prow)) * XS_C + (size_t)(pass * 32 + c * 8));
                }
            }
        };
        auto epi_store = [&](int u, int pass, int sidx) {
            const int prow = lane / 4 + 16 * sidx, c = lane % 4;
            const size_t off = (size_t)(ppix0 + (uint32_t)(32 * u + prow)) * XS_C + (size_t)(pass * 32 + c * 8);
            if constexpr (BIAS) {   // a lane stores the same 8 channels for the whole launch: their bias sits in the LDS table
                float f[8], bv[8];
                Chunk<bf16_t>::unpack(tval, f);
                *(float4*)&bv[0] = *(const float4*)(ctab + pass * 32 + c * 8);
                *(float4*)&bv[4] = *(const float4*)(ctab + pass * 32 + c * 8 + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    f[e] += bv[e];
                    if (p.relu) f[e] = fmaxf(f[e], 0.f);
                }
                *(uint4*)(out + off) = Chunk<bf16_t>::pack(f);
                return;
            }
            if constexpr (!(XS_ABLATE & 2)) *(uint4*)(out + off) = tval;
            float f[8];
            Chunk<bf16_t>::unpack(tval, f);
            if constexpr (BNRED) {
                float yy[8], csc[8], csh[8];
                Chunk<bf16_t>::unpack(yq[sidx], yy);
                *(float4*)&csc[0] = *(const float4*)(ctab + pass * 32 + c * 8);
                *(float4*)&csc[4] = *(const float4*)(ctab + pass * 32 + c * 8 + 4);
                *(float4*)&csh[0] = *(const float4*)(ctab + 64 + pass * 32 + c * 8);
                *(float4*)&csh[4] = *(const float4*)(ctab + 64 + pass * 32 + c * 8 + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float ge = (yy[e] * csc[e] + csh[e]) > 0.f ? f[e] : 0.f;
                    ssum[pass][e] += ge;
                    ssq[pass][e] += ge * yy[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { ssum[pass][e] += f[e]; ssq[pass][e] += f[e] * f[e]; }
            }
        };
        auto epi_fence = [&]() {
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
        };
        // 12 pieces per block, one per MFMA step from step 1 on (block 0: steps 1 .. 12, block 1: 13 .. 24 of 36)
        auto epi_piece = [&](int at) {
            const int u = (at - 1) / 12, k = (at - 1) % 12;
            if (at < 1 || u >= NB) return;
            switch (k) {
                case 0: epi_yload(u, 0); epi_pack(u, 0, 0); break;
                case 1: epi_pack(u, 0, 1); break;
                case 2: epi_pack(u, 1, 0); break;
                case 3: epi_pack(u, 1, 1); break;
                case 4: epi_write(0); break;
                case 5: epi_read(0); break;
                case 6: epi_store(u, 0, 0); epi_read(1); break;
                case 7: epi_store(u, 0, 1); epi_fence(); epi_yload(u, 1); break;
                case 8: epi_write(1); break;
                case 9: epi_read(0); break;
                case 10: epi_store(u, 1, 0); epi_read(1); break;
                default: epi_store(u, 1, 1); epi_fence(); break;
            }
        };
        constexpr bool LOADER = false;                      // (a consumer that also issues the row DMA: the four-consumer experiment)
        bool have_prev = false;
        for (int img = blockIdx.x; img < n_images; img += gridDim.x) {
            if constexpr (LOADER)
                for (int r = 0; r < 6; ++r) issue_row(img, r);
            for (int s = 0; s < spi; ++s) {
                if constexpr (LOADER) wait_vmcnt<0>();      // the rows of strip s (and this wavefront's own stores of strip s - 1)
                __builtin_amdgcn_s_barrier();               // B(s): rows 4 s .. 4 s + 5 are in the ring
                if constexpr (LOADER)
                    if (s + 1 < spi)
                        for (int r = 0; r < 4; ++r) issue_row(img, 4 * s + 6 + r);
                int rowb[NB][3];
#pragma unroll
                for (int u = 0; u < NB; ++u)
#pragma unroll
                    for (int a = 0; a < 3; ++a) rowb[u][a] = ((4 * s + hr[u] + a) % XS_RING) * XS_ROWB;   // sequence row 4 s + 1 + hr + dh
                f32x16_t acc[NB][2];
#pragma unroll
                for (int u = 0; u < NB; ++u)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[u][j][e] = 0.f;
                // 36 steps of 16 reduction elements: (tap, K block, half); fragments XS_AHEAD steps ahead of their MFMAs
                constexpr int AHEAD = XS_AHEAD, SETS = AHEAD + 1;
                uint4 xq[SETS][NB], wq[SETS][2];
                auto fetch = [&](int st, uint4 (&xf)[NB], uint4 (&wf)[2]) {
                    const int tap = st >> 2, kk = st & 3;   // kk = 16-element step within the 64 input channels
                    const int a = tap / 3, b = tap - a * 3;
                    const int chunk = kk * 2 + khalf;
#pragma unroll
                    for (int u = 0; u < NB; ++u) xf[u] = *(const uint4*)(xsm + rowb[u][a] + poff[u][b] + ((chunk ^ pswz[u][b]) * 16));
                    const int slot = (((kk & 1) * 2 + khalf) ^ sw) * 16;
                    const unsigned char* wb = wfrag + (tap * 2 + (kk >> 1)) * (XS_C * 64) + slot;
                    wf[0] = *(const uint4*)(wb);
                    wf[1] = *(const uint4*)(wb + 32 * 64);
                };
                auto mfma_loop = [&](auto with_epilogue) {
#pragma unroll
                    for (int st = 0; st < AHEAD; ++st) fetch(st, xq[st], wq[st]);
#pragma unroll
                    for (int st = 0; st < 36; ++st) {
                        if (st + AHEAD < 36) fetch(st + AHEAD, xq[(st + AHEAD) % SETS], wq[(st + AHEAD) % SETS]);
#pragma unroll
                        for (int u = 0; u < NB; ++u) {
                            bf16x8_t bv;
                            __builtin_memcpy(&bv, &xq[st % SETS][u], 16);
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                bf16x8_t av;
                                __builtin_memcpy(&av, &wq[st % SETS][j], 16);
                                if constexpr (XS_ABLATE & 1) { asm volatile("" ::"v"(av), "v"(bv)); }
                                else acc[u][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[u][j], 0, 0, 0);
                            }
                        }
                        if constexpr (decltype(with_epilogue)::value) epi_piece(st);
                    }
                };
                if (have_prev) mfma_loop(std::true_type{});     // (uniform)
                else mfma_loop(std::false_type{});
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    pacc[u][0] = acc[u][0];
                    pacc[u][1] = acc[u][1];
                }
                ppix0 = ((uint32_t)img * (uint32_t)p.H + (uint32_t)(4 * s)) * XS_W + (uint32_t)(NB * wave * 32);
                have_prev = true;
            }
            __builtin_amdgcn_s_barrier();                   // (see the loader: the ring restarts at row 0 for the next image)
        }
        if (have_prev) {                                    // the last strip's epilogue has nothing to hide behind
#pragma unroll
            for (int at = 1; at <= 12 * NB; ++at) epi_piece(at);
        }
    };
    consume(std::integral_constant<int, 1>{});
    if (p.stats) {
        if constexpr (BNRED) {   // (sum g, sum g * y) -> (sum g, sum g * xhat), xhat = (y - mean) * invstd
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int ch = q * 32 + (lane % 4) * 8 + e;
                    ssq[q][e] = (ssq[q][e] - p.br_mean[ch] * ssum[q][e]) * p.br_invstd[ch];
                }
        }
        // lanes l, l + 4, l + 8, ... hold the same channels: fold them, then one fp64 atomic per channel and wavefront
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int o = 4; o < 64; o <<= 1) {
                    ssum[q][e] += __shfl_xor(ssum[q][e], o, 64);
                    ssq[q][e] += __shfl_xor(ssq[q][e], o, 64);
                }
        if (lane < 4) {
            double* dst = p.stats + (size_t)(blockIdx.x % (unsigned)p.replicas) * XS_C * 2;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int ch = q * 32 + lane * 8 + e;
                    unsafeAtomicAdd(dst + (size_t)ch * 2, (double)ssum[q][e]);
                    unsafeAtomicAdd(dst + (size_t)ch * 2 + 1, (double)ssq[q][e]);
                }
        }
    }
}

}  // namespace

static int xs_num_cu() {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0, n = 0;
        n_cu = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            n_cu = n;
    }
    return n_cu;
}

struct XsBias { const float* bias; int relu; };
static int strip_common(int dtype, const void* x, const void* w, int32_t N, int32_t H, int32_t W, int32_t Ci, int32_t Co,
                        const int32_t* tap_map, void* out, double* stats, const vince_bn_reduce* bnred, int32_t replicas, void* stream,
                        const XsBias* bias_relu = nullptr) {
    VINCE_CHECK_ARG(dtype == VINCE_BF16, VINCE_E_DTYPE, "vince_conv3x3_strip: bf16 only");
    VINCE_CHECK_ARG(x && w && out && N > 0, VINCE_E_ARG, "vince_conv3x3_strip: null pointer");
    VINCE_CHECK_ARG(Ci == XS_C && Co == XS_C && W == XS_W && H > 0 && H % XS_ROWS == 0, VINCE_E_UNSUPPORTED,
                    "vince_conv3x3_strip: 64 -> 64 channels, width 56, height a multiple of 4 (got %d -> %d, %d x %d)", Ci, Co, H, W);
    VINCE_CHECK_ARG((((uintptr_t)x | (uintptr_t)w | (uintptr_t)out) & 15) == 0, VINCE_E_ALIGN, "vince_conv3x3_strip: pointers must be 16-byte aligned");
    const unsigned long long xb = (unsigned long long)N * H * W * Ci * 2;
    VINCE_CHECK_ARG(xb < 0x7ff00000ull, VINCE_E_UNSUPPORTED, "vince_conv3x3_strip: input beyond the 31-bit buffer offsets");
    if (replicas <= 0 || replicas > VINCE_STATS_REPLICAS) replicas = VINCE_STATS_REPLICAS;
    XsParams p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.w = w; p.out = out; p.stats = stats; p.replicas = replicas;
    p.x_bytes = (uint32_t)xb; p.w_bytes = (uint32_t)((unsigned long long)Co * 9 * Ci * 2);
    p.N = N; p.H = H; p.strips_per_image = H / XS_ROWS;
    for (int t = 0; t < 9; ++t) {
        p.tap_w[t] = tap_map ? tap_map[t] : t;
        VINCE_CHECK_ARG(p.tap_w[t] >= 0 && p.tap_w[t] < 9, VINCE_E_ARG, "vince_conv3x3_strip: tap_map[%d] out of range", t);
    }
    int grid = xs_num_cu();
    static const int grid_env = vince_knob("strip_grid", 0);   // (tests: several images per workgroup)
    if (grid_env > 0) grid = grid_env;
    if (grid > N) grid = N;
    VinceProfScope prof(VINCE_TAG_STRIP, 2.0 * N * H * W * Co * 9.0 * Ci, stream);
    if (bnred && bnred->y) {
        VINCE_CHECK_ARG(bnred->mean && bnred->invstd && bnred->sums && bnred->mask_scale && bnred->mask_shift && !bnred->mask_bits && !stats, VINCE_E_ARG,
                        "vince_conv3x3_strip_dgrad: bnred needs y, mean, invstd, sums and mask_scale / mask_shift (no mask bits), and excludes stats");
        p.stats = bnred->sums;
        p.br_y = bnred->y; p.br_mean = bnred->mean; p.br_invstd = bnred->invstd; p.br_msc = bnred->mask_scale; p.br_msh = bnred->mask_shift;
        hipLaunchKernelGGL(conv3x3_strip_kernel<true>, dim3((unsigned)grid), dim3(XS_THREADS), 0, (hipStream_t)stream, p);
    } else if (bias_relu) {
        VINCE_CHECK_ARG(!stats, VINCE_E_ARG, "vince_conv3x3_strip_bias: no statistics in the bias + ReLU epilogue");
        p.bias = bias_relu->bias;
        p.relu = bias_relu->relu;
        hipLaunchKernelGGL((conv3x3_strip_kernel<false, true>), dim3((unsigned)grid), dim3(XS_THREADS), 0, (hipStream_t)stream, p);
    } else {
        hipLaunchKernelGGL(conv3x3_strip_kernel<false>, dim3((unsigned)grid), dim3(XS_THREADS), 0, (hipStream_t)stream, p);
    }
    VINCE_CHECK_LAUNCH();
    return VINCE_OK;
}

// out[N][H][56][64] = conv3x3(x[N][H][56][64], w[64][9][64]) (stride 1, pad 1, bf16), per-channel (sum, sum of squares) of the stored
// output into stats (double[replicas][64][2], zeroed by the caller; may be null).  tap_map: 9 weight-tap indices by kernel position
// (dh + 1) * 3 + (dw + 1), or null for the identity (the forward convolution; the input gradient passes the flipped order).
extern "C" int vince_conv3x3_strip(int dtype, const void* x, const void* w, int32_t N, int32_t H, int32_t W, int32_t Ci, int32_t Co,
                                   const int32_t* tap_map, void* out, double* stats, int32_t replicas, void* stream) {
    return strip_common(dtype, x, w, N, H, W, Ci, Co, tap_map, out, stats, nullptr, replicas, stream);
}

// The same convolution with the epilogue of the BatchNorm-folded inference forward: out = [relu](conv(x, w) + bias[co]) -- bias and ReLU
// applied to the bf16-rounded convolution output exactly as vince_conv_igemm's epilogue does (bit-identical to it); bias may be NULL.
extern "C" int vince_conv3x3_strip_bias(int dtype, const void* x, const void* w, int32_t N, int32_t H, int32_t W, int32_t Ci, int32_t Co,
                                        const float* bias, int32_t relu, void* out, void* stream) {
    const XsBias b{bias, relu};
    return strip_common(dtype, x, w, N, H, W, Ci, Co, nullptr, out, nullptr, nullptr, 0, stream, &b);
}

// The INPUT GRADIENT of the same layer (autograd of resnet.py:119-121): dx = conv3x3(dy, W^T with the taps flipped) -- wt is the prepared
// [Ci][tap][Co] copy -- with vince_conv_igemm's fused BatchNorm-backward reduction (vince_bn_reduce with mask_scale / mask_shift: the
// plain BatchNorm + ReLU below this convolution) accumulated from the stored values.  bnred may be NULL (plain input gradient).
extern "C" int vince_conv3x3_strip_dgrad(int dtype, const void* dy, const void* wt, int32_t N, int32_t H, int32_t W, int32_t C, void* dx,
                                         const vince_bn_reduce* bnred, int32_t replicas, void* stream) {
    static const int32_t flipped[9] = {8, 7, 6, 5, 4, 3, 2, 1, 0};
    return strip_common(dtype, dy, wt, N, H, W, C, C, flipped, dx, nullptr, bnred, replicas, stream);
}
